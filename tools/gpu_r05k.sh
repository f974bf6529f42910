#!/bin/bash
# a STREAMED config-5 run (6 levels through a ring of 3: every level crosses PCIe) with and without the opt-in pair copies -- wall per simulated day
out=gpurun_out/r05k; mkdir -p $out
for v in 0 1; do
  PK_VELOCITY_PAIRS=$v timeout 600 python tools/bench_configs.py --config c5 --steps 120 --nt 6 --nslots 3 --reps 0 --pairs-leg 0 > $out/c5_stream_pairs$v.json 2> $out/c5_stream_pairs$v.err
  python - $out/c5_stream_pairs$v.json $v <<'PY' | tee -a $out/summary.txt
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print('pairs', sys.argv[2], d['kernels'], 'wall_s', round(d['wall_s'],3), 'kernel_ms', round(d['kernel_ms'],2), 'launches', d['launches'], 'steps', d['particle_steps'], {k:round(v,3) for k,v in d['stream_host_s'].items() if v is not None})
PY
done
