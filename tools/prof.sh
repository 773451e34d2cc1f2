#!/bin/bash
# usage (on the GPU box): tools/prof.sh <tag> [bench.py args...]
# Kernel trace + stats of `python3 bench.py <args>`, then PMC passes in their own runs
# (never combined with trace domains), then a small JSON summary under gpurun_out/prof_<tag>/.
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu "$@" > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu --steps 5 --warmup 2 "$@" > $OUT/pmc_fetch.log 2>&1
echo "pmc_fetch rc=$?" >> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu --steps 5 --warmup 2 "$@" > $OUT/pmc_write.log 2>&1
echo "pmc_write rc=$?" >> $OUT/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -- python3 bench.py --no-cpu --steps 5 --warmup 2 "$@" > $OUT/pmc_sq1.log 2>&1
echo "pmc_sq1 rc=$?" >> $OUT/pmc_sq1.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --no-cpu --steps 5 --warmup 2 "$@" > $OUT/pmc_sq2.log 2>&1
echo "pmc_sq2 rc=$?" >> $OUT/pmc_sq2.log
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
summ = {"kernel_stats": [], "pmc": {}}
for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
    summ["kernel_stats"] = [r for r in csv.DictReader(open(f))]
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
    for f in glob.glob(f"{out}/{d}/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            key = r["Kernel_Name"].split("(")[0][-40:] + " grid=" + r.get("Grid_Size", "?")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            summ["pmc"].setdefault(k, {}).update({c: {"mean": sum(x) / len(x), "n": len(x)} for c, x in v.items()})
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summ["pmc"], indent=1)[:6000])
PY
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
