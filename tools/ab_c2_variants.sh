#!/bin/bash
# A/B of library variants on the headline workload (BASELINE config 2), same box, alternating; prints the median / min kernel ms of bench.py's
# 7 timed repetitions of 24 steps and the long run:   bash tools/ab_c2_variants.sh OUT "new ilp memc" [repeats]
out=${1:-gpurun_out/ab_c2}; vs=${2:-"new"}; reps=${3:-2}
mkdir -p $out
for r in $(seq 1 $reps); do
  for v in $vs; do
    if [ $v = new ]; then unset PARCELS_HIP_LIB; else export PARCELS_HIP_LIB=$PWD/parcels_amd/libparcels_hip_$v.so; fi
    if [ $v = base ] && [ -n "$PK_BASE_ABI" ]; then export PARCELS_HIP_ALLOW_ABI=$PK_BASE_ABI; else unset PARCELS_HIP_ALLOW_ABI; fi  # (the library of the previous round)
    timeout 300 python bench.py --steps 24 --warmup 2 --secondary 0 --no-cpu-baseline --user-kernels 0 > $out/c2_${v}_$r.json 2> $out/c2_${v}_$r.err
    python - $out/c2_${v}_$r.json $v $r <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); k=d["timed_reps"]["kernel_ms"]; lr=d.get("long_run") or {}
print(f"{sys.argv[2]:10s} rep {sys.argv[3]} C2 24 steps kernel ms median {k['median']:.3f} min {k['min']:.3f} max {k['max']:.3f}  value {d['value']:.4g}  long_run kernel ms {lr.get('kernel_ms')}", flush=True)
PY
  done
done | tee $out/summary.txt
