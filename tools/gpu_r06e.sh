#!/bin/bash
# round 6, fifth GPU call: (1) parity of listed failing samples on the dedicated kernels (+ the twe / fuzz-overshoot tests); (2) what that cost
# the hot kernels: C2 / C5 / C3 `prev` (commit before) vs `new`; C2 residency / batching variants w5 | b16 | w3b16; (3) a run that overshoots the
# last level at 1e6 particles: passes and seconds; (4) write-out sparse cadence; (5) bench.py --gpus 2 rehearsal (gloo, shared GPU)
out=gpurun_out/r06e; mkdir -p $out; OUT=$PWD/$out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_twe_passes.py tests/test_gpu_parity.py tests/test_gpu_fast_cgrid.py tests/test_gpu_fast_path.py tests/test_gpu_semantics.py -m gpu -q > $out/pytest_twe.log 2>&1; echo "pytest twe rc $?" | tee -a $out/summary.txt; tail -5 $out/pytest_twe.log | tee -a $out/summary.txt
PK_BASE_ABI=9 bash tools/ab_c2_variants.sh $out/ab_c2 "prev new w5 b16 w3b16" 2 | tee -a $out/summary.txt
bash tools/ab_c5_variants.sh $out/ab_c5 "prev new" 3 "--reps 3 --pairs-leg 0 --check 1e5" c5 | tee -a $out/summary.txt
bash tools/ab_c5_variants.sh $out/ab_c3 "prev new" 2 "--reps 3 --check 1e5" c3 | tee -a $out/summary.txt
timeout 600 python tools/bench_twe_passes.py 1e6 fast > $out/twe_passes_1e6.json 2> $out/twe_passes.err; echo "twe passes rc $?" | tee -a $out/summary.txt; tail -2 $out/twe_passes_1e6.json | cut -c1-600 | tee -a $out/summary.txt
PARCELS_HIP_LIB=$PWD/parcels_amd/libparcels_hip_prev.so timeout 600 python tools/bench_twe_passes.py 1e6 fast > $out/twe_passes_1e6_prev.json 2>> $out/twe_passes.err; tail -2 $out/twe_passes_1e6_prev.json | cut -c1-600 | tee -a $out/summary.txt
timeout 900 python tools/bench_writeout.py --particles 4e6 --steps 480 --every 48,240 > $out/writeout_4e6_sparse.json 2> $out/writeout_4e6_sparse.err; echo "writeout sparse rc $?" | tee -a $out/summary.txt
python - $out/writeout_4e6_sparse.json <<'PY' | tee -a $out/summary.txt
import json,sys
for f in sys.argv[1:]:
    try: d=json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(d["workload"])
    for c in d["cadences"]:
        print("  every", c["every_steps"], "wall", {k:round(v,3) for k,v in c["wall_s"].items()}, "per table ms", {k:round(v,1) for k,v in c["per_table_ms"].items()}, "hidden", round(c["output_hidden_frac"],3), "async<=inline", c["async_not_slower_than_inline"], "identical", c["byte_identical"], c["async_writer"])
    for k,v in d["encode_one_table_of_n_rows"].items():
        print("  encode", k, v if not isinstance(v,dict) else (round(v["seconds"]*1e3,1), "ms", round(v["GB_per_s"],2), "GB/s"))
PY
PARCELS_AMD_BENCH_REHEARSAL=1 timeout 900 python bench.py --gpus 2 --particles 2e6 --steps 8 --warmup 2 > $out/bench_gpus2_rehearsal.json 2> $out/bench_gpus2_rehearsal.err; echo "bench --gpus 2 rehearsal rc $?" | tee -a $out/summary.txt; tail -1 $out/bench_gpus2_rehearsal.json | cut -c1-900 | tee -a $out/summary.txt
