"""TEST INFRASTRUCTURE: one rank of tests/test_gpu_multigpu_preflight.py::test_two_shards_are_one_batch_on_the_real_kernels (started by
torch.distributed.run; both ranks use cuda:0 and talk over gloo, because RCCL refuses two ranks on one device)."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist

    import parcels_amd as pa
    from case_utils import build_fieldset, load_golden

    name, outdir = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    case, _, _ = load_golden(name)
    order = np.load(os.path.join(outdir, "order.npy"))  # every rank builds the SAME arrays: late releases first, so that shard 1 holds none of them
    fs = build_fieldset(case)
    n = len(order)
    t0 = np.asarray(case["t0"], dtype=np.float64)[order]
    pset = pa.ParticleSet(fs, pclass=pa.get_default_particle(np.float64), x=np.asarray(case["x"])[order], y=np.asarray(case["y"])[order],
                          z=np.asarray(case["z"])[order], t=t0, shard="auto")
    assert 0 < len(pset) < n
    pf = pa.ParticleFile(os.path.join(outdir, "out.parquet"), outputdt=1.0e9, mode="w")  # collective (2 ranks); ONE Kernel.execute for the run
    kernels = [getattr(pa.kernels, k) for k in case["kernels"]]
    err = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            pset.execute(kernels, dt=float(case["dt"]), runtime=float(case["runtime"]), output_file=pf)
        except (pa.OutsideTimeInterval, pa.FieldOutOfBoundError) as e:
            err = type(e).__name__
    st = pset._last_stats or {}
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), err=np.array(err or ""), reran=st.get("reran", -1), keys=np.array(st.get("time_error_keys", []), dtype=np.int64),
             codes=np.array(st.get("codes_any_shard", []), dtype=np.int64), **{k: np.asarray(v) for k, v in pset._data.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
