# LMMSE throughput (per-RE and block fading) of one or more builds: tools/lmmsebench.sh [lib.so ...]
set -e
libs="$@"; [ -z "$libs" ] && libs=channelestimationtransformer_amd/csrc/libcet.so
for lib in $libs; do
  for hs in ${HS:-1 14}; do
    CET_LIB=$PWD/$lib timeout -k 10 120 python bench.py --workload lmmse --h-share $hs --steps 10 --warmup 2 --skip-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib'.split('/')[-1], 'h_share', $hs, round(d['value']/1e9,3), 'G RE/s', round(d['roofline']['achieved']), 'GB/s', round(d['roofline']['frac'],4))"
  done
done
