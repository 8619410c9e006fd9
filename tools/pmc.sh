#!/bin/bash
# One rocprofv3 counter pass over a bench.py invocation.
# Usage: bash tools/pmc.sh <tag> "<counter list>" <bench.py arguments...>
set -u
TAG=$1; CNT=$2; shift 2
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc $CNT --output-format csv -d $OUT -- python3 $REPO/bench.py --skip-cpu-baseline --skip-secondary "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "cet::" in k:
        agg[(k.split("(")[0][-48:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:50s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
