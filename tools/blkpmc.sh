# HBM traffic counters of the block-fading equalizer for one or more library builds:
#   tools/blkpmc.sh [lib.so ...]   (on the GPU box, from the repo root; results in gpurun_out/blkpmc/)
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/blkpmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
libs="$@"; [ -z "$libs" ] && libs=channelestimationtransformer_amd/csrc/libcet.so
for lib in $libs; do
  tag=$(basename $lib .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    CET_LIB=$REPO/$lib timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $OUT/${tag}_$c -- python3 $REPO/bench.py --steps 3 --warmup 1 --skip-cpu-baseline --skip-secondary --workload lmmse --batch 8388608 --h-share 14 > $OUT/${tag}_$c.log 2>&1 || { tail -5 $OUT/${tag}_$c.log; exit 1; }
    f=$(find $OUT/${tag}_$c -name '*counter_collection.csv' | head -1)
    python3 - "$f" "$tag" "$c" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'lmmse_block' in r['Kernel_Name']]
v=[float(r['Counter_Value']) for r in rows]
print(sys.argv[2], sys.argv[3], 'launches', len(v), 'mean KiB', sum(v)/max(1,len(v)))
PY
  done
done
