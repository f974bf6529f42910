#!/bin/bash
# round 6: the tree with PK_CG_NEAR / PK_CG_LEAN / PK_FAST_LEAN in the dedicated kernels: accuracy of the routines on the device, whole GPU suite,
# 4000 fuzz seeds, smoke, trace + PMC passes of every profiled kernel (profiles/$N_*), the default bench line        (N=r06s SEED0=500000 bash ...)
N=${N:-r06p}; out=gpurun_out/$N; mkdir -p $out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -I parcels_amd/csrc tools/lean_math_check.hip -o /tmp/lean_math_check 2> /dev/null && /tmp/lean_math_check | tee -a $out/summary.txt; echo "lean_math_check rc $?" | tee -a $out/summary.txt
timeout 1800 python -m pytest tests -m gpu -q -n 4 > $out/pytest_all.log 2>&1; echo "pytest all rc $?" | tee -a $out/summary.txt; grep -E "^FAILED" $out/pytest_all.log | cut -c1-200 | tee -a $out/summary.txt; tail -1 $out/pytest_all.log | tee -a $out/summary.txt
PARCELS_FUZZ_SEED0=${SEED0:-200000} PARCELS_FUZZ_SEEDS=4000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_configuration -n 4 > $out/fuzz.log 2>&1; echo "fuzz rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/fuzz.log | tee -a $out/summary.txt; grep -E "^FAILED" $out/fuzz.log | head | tee -a $out/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" | tee -a $out/summary.txt; tail -1 $out/smoke.log | tee -a $out/summary.txt
bash tools/gpu_round6_profiles.sh $N > $out/profiles.log 2>&1; echo "profiles rc $?" | tee -a $out/summary.txt
timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench_default.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "check", (d.get("check") or {}).get("passed"), "frac", d["roofline"]["frac"], "stale", d["roofline"].get("counters_stale"), "kernel", d["timed_reps"]["kernel_ms"]["median"])
for s in d.get("secondary") or []:
    print(s.get("kernels"), s.get("kernel_ms"), (s.get("roofline") or {}).get("frac"), (s.get("check") or {}).get("passed"), (s.get("roofline") or {}).get("counters_stale"), ((s.get("roofline") or {}).get("attainable") or {}).get("frac_of_peak"))
print(d.get("legs_wall_s"))
PY
echo finished | tee -a $out/summary.txt
