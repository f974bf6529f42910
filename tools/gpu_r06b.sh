#!/bin/bash
# round 6, second GPU call: (1) the whole GPU suite on this tree (LazyColumns rewrite, flag word, fused search, pk_api fixes); (2) C2 A/B
# base | ns2 (flag word only) | new (+ the fused lat / lon search); (3) C5 RK45: trace + SQ + FETCH + WRITE passes of base, dma, pxgdma
# (scratch 160 / 152 / 84 B: what reaches memory?); (4) bench.py default line (headline check, hash digest check)
out=gpurun_out/r06b; mkdir -p $out; OUT=$PWD/$out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_all.log 2>&1; echo "pytest all rc $?" | tee -a $out/summary.txt; tail -4 $out/pytest_all.log | tee -a $out/summary.txt
bash tools/ab_c2_variants.sh $out/ab_c2 "base ns2 new" 3 | tee -a $out/summary.txt
for v in base dma pxgdma; do
  export PARCELS_HIP_LIB=$PWD/parcels_amd/libparcels_hip_$v.so
  B="python $PWD/tools/bench_configs.py --config c5 --only rk45 --reps 1 --pairs-leg 0"
  (cd /tmp; rocprofv3 --kernel-trace --stats -d $OUT/rk45_${v}_trace -o t -- $B > $OUT/rk45_${v}_run.json 2> /dev/null
   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/rk45_${v}_pmc_sq -o p --output-format csv -- $B > /dev/null 2> /dev/null
   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/rk45_${v}_pmc_fetch -o p --output-format csv -- $B > /dev/null 2> /dev/null
   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/rk45_${v}_pmc_write -o p --output-format csv -- $B > /dev/null 2> /dev/null)
done
unset PARCELS_HIP_LIB
find $OUT -name "*.db" -delete 2>/dev/null
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench_default.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "check", d.get("check"), "frac", d["roofline"]["frac"])
for s in d.get("secondary") or []:
    print(s.get("kernels"), s.get("kernel_ms"), (s.get("roofline") or {}).get("frac"), (s.get("check") or {}).get("oracle_hash_table"), (s.get("check") or {}).get("passed"))
print(d.get("legs_wall_s"))
PY
