#!/bin/bash
# usage (on the GPU box): tools/prof_trace.sh <tag> [bench.py args...]  -- kernel trace + stats only
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu "$@" > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
tail -2 $OUT/trace.log
