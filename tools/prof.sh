#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]  -- kernel trace + stats, then PMC passes (separate runs)
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu "$@" > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc1 -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > $OUT/pmc1.log 2>&1
echo "pmc1 rc=$?" >> $OUT/pmc1.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc2 -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > $OUT/pmc2.log 2>&1
echo "pmc2 rc=$?" >> $OUT/pmc2.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > $OUT/pmc3.log 2>&1
echo "pmc3 rc=$?" >> $OUT/pmc3.log
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > $OUT/pmc4.log 2>&1
echo "pmc4 rc=$?" >> $OUT/pmc4.log
# keep the merged payload small: stats + per-kernel summaries only
find $OUT -name "*.csv" -size +3M -delete
ls -R $OUT | head -50
