#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <workload>   -- one SQ counter pass over the wave kernel
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/pmcq_$1; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/p -- python3 bench.py --no-cpu --no-extra --workload $2 --no-secondary --no-e2e --copies 1 --one-at-a-time --steps 5 --warmup 2 > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wave_fast" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k:round(sum(v)/len(v)) for k,v in agg.items()})
PY
rm -rf $OUT/p
