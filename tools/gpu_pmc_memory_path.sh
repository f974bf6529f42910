#!/bin/bash
# Runs ON THE GPU BOX: cache-path counters (L2 hit rate, vector-L1 traffic and stalls, address-unit busy) of tools/bench_configs.py, one
# rocprofv3 --pmc pass per counter group (a group with a counter this rocprofv3 does not know fails alone), summed per advection kernel.
# Every pass runs under its own timeout: a group that asks for more counters of one block than it has slots ("Request exceeds the
# capabilities of the hardware": four TA counters did, round 4) aborts inside rocprofv3 and then never returns.
#   usage: bash tools/gpu_pmc_memory_path.sh TAG config [bench_configs args]   ->  gpurun_out/TAG_mem/summary.txt (+ counters_available.txt)
TAG=$1; CFG=$2; shift; shift
OUT=$PWD/gpurun_out/${TAG}_mem; mkdir -p $OUT
export TMPDIR=/tmp
B="python $PWD/tools/bench_configs.py --config $CFG --reps 0 $*"
cd /tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_SCRATCH"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o p --output-format csv -- $B > /dev/null 2> $OUT/p$i.err || echo "group $i failed: $grp" >> $OUT/failed.txt
done
cd - > /dev/null
python - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(out, "p*", "**", "p_counter_collection.csv"), recursive=True):
    last = {}
    rows = list(csv.DictReader(open(f)))
    for r in rows:  # the LAST dispatch of every advection kernel is the timed launch (cold run only with --reps 0: then it is the only one)
        if "advect" in r["Kernel_Name"]:
            last[r["Kernel_Name"]] = max(last.get(r["Kernel_Name"], 0), int(r["Dispatch_Id"]))
    for r in rows:
        k = r["Kernel_Name"]
        if k in last and int(r["Dispatch_Id"]) == last[k]:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for k, c in acc.items():
        fo.write(k + "\n")
        for n in sorted(c):
            fo.write(f"  {n:40s} {c[n]:.6g}\n")
        if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            fo.write(f"  L2 hit rate                              {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}\n")
print(open(os.path.join(out, "summary.txt")).read())
PY
cat $OUT/failed.txt 2>/dev/null
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
du -sh $OUT
