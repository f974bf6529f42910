#!/bin/bash
# round 6: two writer threads (table k+1 compresses while table k is written): GPU tests that write files, then the write-out sweeps again
out=gpurun_out/r06j; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -k "output or file or write or resident or preflight or comm" > $out/pytest_files.log 2>&1; echo "pytest files rc $?" | tee -a $out/summary.txt; tail -3 $out/pytest_files.log | tee -a $out/summary.txt
timeout 900 python tools/bench_writeout.py --particles 1e7 --steps 96 --every 24 > $out/writeout_1e7.json 2> $out/writeout_1e7.err; echo "writeout 1e7 rc $?" | tee -a $out/summary.txt
timeout 900 python tools/bench_writeout.py --particles 4e6 --steps 20 --every 2,10 > $out/writeout_4e6.json 2> $out/writeout_4e6.err; echo "writeout 4e6 rc $?" | tee -a $out/summary.txt
python - $out/writeout_1e7.json $out/writeout_4e6.json <<'PY' | tee -a $out/summary.txt
import json,sys
for f in sys.argv[1:]:
    try: d=json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(d["workload"])
    for c in d["cadences"]:
        print("  every", c["every_steps"], "wall", {k:round(v,3) for k,v in c["wall_s"].items()}, "per table ms", {k:round(v,1) for k,v in c["per_table_ms"].items()}, "hidden", round(c["output_hidden_frac"],3), "async<=inline", c["async_not_slower_than_inline"], "identical", c["byte_identical"], c["async_writer"])
PY
timeout 900 python bench.py --secondary 0 --user-kernels 0 --no-cpu-baseline --long-run 0 --repeat-execute 0 > $out/bench_with_output.json 2> $out/bench_with_output.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench_with_output.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value", d["value"], "with_output", {k:v for k,v in (d.get("with_output") or {}).items() if k not in ("detail","note","workload")})
PY
