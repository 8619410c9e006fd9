# HBM traffic counters of the receive chain on the reference grid (LS estimate + equalizer object):
#   tools/rxpmc.sh   (on the GPU box, from the repo root; results in gpurun_out/rxpmc/)
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/rxpmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $REPO/tools/rxchain.py > $OUT/$c.log 2>&1 || { tail -5 $OUT/$c.log; exit 1; }
  f=$(find $OUT/$c -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$c" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'ls_nn' in k or 'lmmse_grid' in k: acc[k.split('(')[0][:40]].append(float(r['Counter_Value']))
for k,v in acc.items(): print(sys.argv[2], k, 'launches', len(v), 'mean KiB', sum(v)/len(v))
PY
done
