#!/bin/bash
# usage (on the GPU box): tools/prof_aux.sh <tag>  -- rocprofv3 kernel stats of the secondary kernels
# (sw, interval count / locate, span cover, index build) as tools/bench_aux.py drives them
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$1
mkdir -p $OUT
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_aux.py > $OUT/aux.log 2>&1
echo "rc=$?" >> $OUT/aux.log
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
find $OUT/trace -name "*kernel_trace.csv" -size +2M -delete
cat $OUT/aux.log | tail -12
head -30 $OUT/kernel_stats.csv
