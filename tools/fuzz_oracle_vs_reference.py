"""Offline sweep (build container only): the reference under oracle/ref_shim.py vs the C oracle on configurations of the GPU fuzz
generator (tests/test_gpu_fuzz.py), one subprocess per seed with a timeout (the reference can hang, DESIGN.md deviation 5).
Usage: python tools/fuzz_oracle_vs_reference.py LO HI [decorate]   (decorate: with the user-kernel tokens of test_gpu_fuzz.decorate_case)
PARCELS_ORACLE_CALL_WIDE=0: the oracle's per-particle time error (c_oracle.execute(call_wide_time_error=False)) instead of the reference's call-wide one."""
import sys, time, multiprocessing as mp, traceback
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
sys.dont_write_bytecode=True
def run(seed, q):
    try:
        import numpy as np, warnings
        warnings.simplefilter("ignore")
        import test_gpu_fuzz as f
        from oracle import make_golden as mg
        from case_utils import run_oracle, compare, stop_time_of_reference, is_curvilinear
        case,sort=f.draw_case(seed)
        if len(sys.argv) > 3 and sys.argv[3] == "decorate":
            before = list(case["kernels"])
            case = f.decorate_case(seed, case)
            if case["kernels"] == before:
                q.put((seed,"ok","undecorated",None)); return
        out, err, extras = mg.ref_run_case(case)
        tstop = stop_time_of_reference(case, out, err)
        import os
        got, gerr, _ = run_oracle(case, endtime=tstop, call_wide_time_error=os.environ.get("PARCELS_ORACLE_CALL_WIDE", "1") == "1")
        tol=f.tolerance(case)
        # the classes of tests/test_oracle_golden.py::test_oracle_matches_reference_on_random_configurations: bit-identical for fp64
        # rectilinear cases, 1e-13 where NumPy's SIMD sin / cos and libm may differ by an ulp (curvilinear meshes, sampled velocities),
        # 1e-11 for the stochastic kernels (libm vs NumPy log / sin / cos), the float32 class as drawn
        stochastic = any(k.startswith("AdvectionDiffusion") or k == "DiffusionUniformKh" for k in case["kernels"])
        if tol in (1e-10, 1e-11):
            tol = 1e-11 if stochastic else (1e-13 if (is_curvilinear(case) or case.get("sample_into")) else 0.0)
        scale=float(max(np.nanmax(np.abs(case["lon"])), np.nanmax(np.abs(case["lat"]))))
        try:
            assert gerr==err, f"error {gerr} vs {err}"
            compare(got, out, rtol=tol, atol_pos=tol*scale, check_state="errors" if tstop is not None else "all", label=f"seed {seed}")
            q.put((seed,"ok",str(err),case['kernels']))
        except AssertionError as e:
            q.put((seed,"MISMATCH",str(e)[:300],case['kernels']))
    except Exception as e:
        q.put((seed,"EXC",traceback.format_exc()[-400:],None))
if __name__=="__main__":
    lo,hi=int(sys.argv[1]),int(sys.argv[2])
    res=[]
    for seed in range(lo,hi):
        q=mp.Queue(); p=mp.Process(target=run,args=(seed,q)); p.start(); p.join(120)
        if p.is_alive():
            p.terminate(); print(seed,"TIMEOUT (reference hang?)",flush=True); continue
        try: r=q.get(timeout=5)
        except Exception: r=(seed,"NORESULT","",None)
        if r[1]!="ok": print(r,flush=True)
        res.append(r[1])
    import collections; print(lo,hi,collections.Counter(res))
