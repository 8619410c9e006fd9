# Transformer forward throughput of one or more builds of the library: tools/infbench.sh [lib.so ...]
set -e
libs="$@"; [ -z "$libs" ] && libs=channelestimationtransformer_amd/csrc/libcet.so
for lib in $libs; do
  for maps in "" "--no-attention-maps"; do
    CET_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 100 --warmup 50 --skip-cpu-baseline --skip-secondary $maps | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib'.split('/')[-1], '$maps', round(d['value']), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
  done
done
