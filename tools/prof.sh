#!/bin/bash
# usage (on the GPU box): tools/prof.sh <tag> <grid-filter> [bench.py args...]
# Kernel trace + stats of `python3 bench.py <args>` (one pass at a time, so that the per-kernel average
# is the launch duration bench.py's roofline block reports), then PMC passes in their own runs
# (never combined with trace domains), then gpurun_out/prof_<tag>/{kernel_stats.csv,wave_pmc.json}.
# <grid-filter>: only dispatches of the wave_fast kernels with this Grid_Size enter wave_pmc.json (0 = all).
set -o pipefail
TAG=$1; GRID=$2; shift; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $ROOT
B="python3 bench.py --no-cpu --no-secondary --one-at-a-time"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B "$@" > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 12 --warmup 2 --ramp-ms 5 "$@" > $OUT/pmc_fetch.log 2>&1
echo "pmc_fetch rc=$?" >> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 12 --warmup 2 --ramp-ms 5 "$@" > $OUT/pmc_write.log 2>&1
echo "pmc_write rc=$?" >> $OUT/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -- $B --steps 12 --warmup 2 --ramp-ms 5 "$@" > $OUT/pmc_sq1.log 2>&1
echo "pmc_sq1 rc=$?" >> $OUT/pmc_sq1.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- $B --steps 12 --warmup 2 --ramp-ms 5 "$@" > $OUT/pmc_sq2.log 2>&1
echo "pmc_sq2 rc=$?" >> $OUT/pmc_sq2.log
python3 - "$OUT" "$GRID" <<'PY'
import csv, glob, json, sys, collections
out, grid = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
grids = collections.Counter()
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
    for f in glob.glob(f"{out}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "wave_fast" not in r["Kernel_Name"]:
                continue
            grids[(r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size", "?"))] += 1
            if grid != "0" and r.get("Grid_Size") != grid:
                continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in sorted(agg.items())}
summ["_dispatches_seen"] = {f"{k[0]} grid={k[1]}": n for k, n in grids.items()}
json.dump(summ, open(out + "/wave_pmc.json", "w"), indent=1)
print(json.dumps(summ, indent=1)[:3000])
PY
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2
find $OUT/trace -name "*kernel_trace.csv" -delete
head -6 $OUT/kernel_stats.csv
