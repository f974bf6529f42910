#!/bin/bash
# A/B of library variants (tools/build_variant.sh NAME ... -> parcels_amd/libparcels_hip_NAME.so) on BASELINE config 5, same box, alternating:
#   bash tools/ab_c5_variants.sh OUT "base new h1" [repeats] ["extra bench_configs.py args"] [c3|c5]      ("new" = the library of `make`)
out=${1:-gpurun_out/ab_c5}; vs=${2:-"new"}; reps=${3:-2}; extra=${4:-"--reps 3"}; cfg=${5:-c5}
mkdir -p $out
for r in $(seq 1 $reps); do
  for v in $vs; do
    if [ $v = new ]; then unset PARCELS_HIP_LIB; else export PARCELS_HIP_LIB=$PWD/parcels_amd/libparcels_hip_$v.so; fi
    if [ $v = base ] && [ -n "$PK_BASE_ABI" ]; then export PARCELS_HIP_ALLOW_ABI=$PK_BASE_ABI; else unset PARCELS_HIP_ALLOW_ABI; fi  # (the library of the previous round)
    timeout 600 python tools/bench_configs.py --config $cfg $extra > $out/${cfg}_${v}_$r.json 2> $out/${cfg}_${v}_$r.err
    python - $out/${cfg}_${v}_$r.json $v $r <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(f"{sys.argv[2]:14s} rep {sys.argv[3]} {d['kernels']:22s} {d['kernel_ms']:8.3f} ms  steps {d['particle_steps']} attempts {d['attempts']}" + (f"  check {max((d['check'].get('max_rel_diff') or {}).values(), default=None)}" if d.get('check') else ""), flush=True)
PY
  done
done | tee $out/summary.txt
