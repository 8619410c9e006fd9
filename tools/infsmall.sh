# Transformer forward at small batches for one or more builds: tools/infsmall.sh [lib.so ...]
set -e
libs="$@"; [ -z "$libs" ] && libs=channelestimationtransformer_amd/csrc/libcet.so
for lib in $libs; do
  for b in 64 256 1024 1280; do
    CET_LIB=$PWD/$lib timeout -k 10 120 python bench.py --batch $b --steps 50 --warmup 5 --skip-cpu-baseline --skip-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib'.split('/')[-1], 'B', $b, round(d['roofline']['kernel_ms'],4), 'ms', round(d['value']))"
  done
done
