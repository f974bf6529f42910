#!/bin/bash
# Sweep PK_COPY_THREADS on the C3 streaming run (field levels through the 3-slot ring); prints wall / kernel / create seconds.
for t in "$@"; do
  PK_COPY_THREADS=$t timeout 300 python tools/bench_configs.py --config c3 --steps 120 --nt 6 --nslots 3 2>/dev/null | tail -n 1 > /tmp/sweep_$t.json
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads(open(f"/tmp/sweep_{t}.json").read())
print("threads", t, "wall_s", round(d["wall_s"], 3), "kernel_ms", round(d["kernel_ms"], 1), "create_s", round(d["device_create_s"], 2), flush=True)
PY
done
