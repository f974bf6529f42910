#!/bin/bash
# round 6, third GPU call: (1) the GPU tests that batch 2 found broken + the new ones (C-ABI exchange, held references, filtered snapshot);
# (2) write-out: tools/bench_writeout.py at two sizes (device write filter, 128K-row pages, positional writes); (3) kernels with the
# pinned scalars: C2 base | new, C5 / C3 base | nopin | new | pxgdma; (4) RK45 counters of base, new and pxgdma (SQ incl. scalar loads, FETCH, WRITE)
out=gpurun_out/r06c; mkdir -p $out; OUT=$PWD/$out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_comm_cabi.py tests/test_gpu_multigpu_preflight.py tests/test_gpu_resident_columns.py tests/test_gpu_parity.py tests/test_gpu_semantics.py tests/test_gpu_fast_cgrid.py tests/test_gpu_fast_path.py tests/test_gpu_streaming.py -m gpu -q > $out/pytest_subset.log 2>&1; echo "pytest subset rc $?" | tee -a $out/summary.txt; tail -8 $out/pytest_subset.log | tee -a $out/summary.txt
timeout 600 python tools/bench_writeout.py --particles 4e6 --steps 20 --every 2,10 > $out/writeout_4e6.json 2> $out/writeout_4e6.err; echo "writeout 4e6 rc $?" | tee -a $out/summary.txt
timeout 900 python tools/bench_writeout.py --particles 1e7 --steps 96 --every 24,96 > $out/writeout_1e7.json 2> $out/writeout_1e7.err; echo "writeout 1e7 rc $?" | tee -a $out/summary.txt
python - $out/writeout_4e6.json $out/writeout_1e7.json <<'PY' | tee -a $out/summary.txt
import json,sys
for f in sys.argv[1:]:
    try: d=json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(d["workload"])
    for c in d["cadences"]:
        print("  every", c["every_steps"], "wall", {k:round(v,3) for k,v in c["wall_s"].items()}, "per table ms", {k:round(v,1) for k,v in c["per_table_ms"].items()}, "hidden", round(c["output_hidden_frac"],3), "async<=inline", c["async_not_slower_than_inline"], "identical", c["byte_identical"], "GB/s", {k:round(v,2) for k,v in c["table_GB_per_s"].items()}, c["async_writer"])
    for k,v in d["encode_one_table_of_n_rows"].items():
        print("  encode", k, v if not isinstance(v,dict) else (round(v["seconds"]*1e3,1), "ms", round(v["GB_per_s"],2), "GB/s", v.get("waiting_for")))
PY
bash tools/ab_c2_variants.sh $out/ab_c2 "base new" 3 | tee -a $out/summary.txt
bash tools/ab_c5_variants.sh $out/ab_c5 "base nopin new pxgdma" 2 "--reps 3 --pairs-leg 0 --check 1e5" c5 | tee -a $out/summary.txt
bash tools/ab_c5_variants.sh $out/ab_c3 "base nopin new" 2 "--reps 3 --check 1e5" c3 | tee -a $out/summary.txt
for v in base new pxgdma; do
  if [ $v = new ]; then unset PARCELS_HIP_LIB; else export PARCELS_HIP_LIB=$PWD/parcels_amd/libparcels_hip_$v.so; fi
  if [ $v = base ]; then export PARCELS_HIP_ALLOW_ABI=8; else unset PARCELS_HIP_ALLOW_ABI; fi
  B="python $PWD/tools/bench_configs.py --config c5 --only rk45 --reps 1 --pairs-leg 0"
  (cd /tmp; rocprofv3 --kernel-trace --stats -d $OUT/rk45_${v}_trace -o t -- $B > $OUT/rk45_${v}_run.json 2> /dev/null
   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/rk45_${v}_pmc_sq -o p --output-format csv -- $B > /dev/null 2> /dev/null
   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/rk45_${v}_pmc_fetch -o p --output-format csv -- $B > /dev/null 2> /dev/null
   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/rk45_${v}_pmc_write -o p --output-format csv -- $B > /dev/null 2> /dev/null)
done
unset PARCELS_HIP_LIB PARCELS_HIP_ALLOW_ABI
find $OUT -name "*.db" -delete 2>/dev/null
