#!/bin/bash
# round 6, fourth GPU call: (1) RK45 with the new cell-cache placement (CM 60: 28 B scratch) against CM 2 (round 5's) -- time, then its
# counters; (2) the memory-side roof of its access pattern (tools/gather_roof); (3) write-out sweep incl. a sparse cadence; (4) the whole GPU
# suite; (5) profiles of the final binary (trace + PMC, summarised on the box); (6) the default bench line
out=gpurun_out/r06d; mkdir -p $out; OUT=$PWD/$out
export TMPDIR=/tmp
bash tools/ab_c5_variants.sh $out/ab_c5 "base cm2 new" 3 "--reps 3 --pairs-leg 0 --check 1e5 --only rk45" c5 | tee -a $out/summary.txt
for alu in 0 250 500 1000; do ./tools/gather_roof --alu $alu | tee -a $out/gather_roof.jsonl; done
./tools/gather_roof --p 0.0 | tee -a $out/gather_roof.jsonl
./tools/gather_roof --p 0.3 | tee -a $out/gather_roof.jsonl
timeout 600 python tools/bench_writeout.py --particles 4e6 --steps 480 --every 48,240 > $out/writeout_4e6_sparse.json 2> $out/writeout_4e6_sparse.err; echo "writeout sparse rc $?" | tee -a $out/summary.txt
timeout 900 python tools/bench_writeout.py --particles 1e7 --steps 96 --every 24 > $out/writeout_1e7.json 2> $out/writeout_1e7.err; echo "writeout 1e7 rc $?" | tee -a $out/summary.txt
python - $out/writeout_4e6_sparse.json $out/writeout_1e7.json <<'PY' | tee -a $out/summary.txt
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
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_all.log 2>&1; echo "pytest all rc $?" | tee -a $out/summary.txt; tail -4 $out/pytest_all.log | tee -a $out/summary.txt
bash tools/gpu_round6_profiles.sh r06f > $out/profiles.log 2>&1; echo "profiles rc $?" | tee -a $out/summary.txt
cp -r gpurun_out/r06_profiles $out/ 2>/dev/null
cp profiles/pmc_latest.json profiles/pmc_secondary_latest.json /tmp/ 2>/dev/null
timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench_default.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "check", (d.get("check") or {}).get("passed"), "frac", d["roofline"]["frac"], "stale", d["roofline"].get("counters_stale"))
print("with_output", {k:v for k,v in (d.get("with_output") or {}).items() if k not in ("detail","note","workload")})
for s in d.get("secondary") or []:
    print(s.get("kernels"), s.get("kernel_ms"), (s.get("roofline") or {}).get("frac"), (s.get("check") or {}).get("oracle_hash_table"), (s.get("check") or {}).get("passed"), (s.get("roofline") or {}).get("attainable",{}) if s.get("kernels")=="AdvectionRK45" else "")
print(d.get("legs_wall_s"))
PY
