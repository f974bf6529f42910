#!/bin/bash
# Runs ON THE GPU BOX: SQ / FETCH / WRITE PMC passes + kernel trace of tools/bench_configs.py (C3 / C5).  usage: TAG config [args]
TAG=$1; CFG=$2; shift; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
B="python $PWD/tools/bench_configs.py --config $CFG $*"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o t -- $B > $OUT/${TAG}_run.json 2> $OUT/${TAG}_trace.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmc_sq -o p --output-format csv -- $B > /dev/null 2> $OUT/${TAG}_pmc_sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR -d $OUT/${TAG}_pmc_sq2 -o p --output-format csv -- $B > /dev/null 2> $OUT/${TAG}_pmc_sq2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch -o p --output-format csv -- $B > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write -o p --output-format csv -- $B > /dev/null 2> $OUT/${TAG}_pmc_write.err
cd - > /dev/null
find $OUT/${TAG}_pmc_* -name "*.db" -delete 2>/dev/null
cat $OUT/${TAG}_run.json
