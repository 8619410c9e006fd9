# Recurrent-rollout throughput of one or more builds of the library: tools/rnnbench.sh [lib.so ...]
set -e
libs="$@"; [ -z "$libs" ] && libs=channelestimationtransformer_amd/csrc/libcet.so
for lib in $libs; do
  for k in rnn gru lstm; do
    CET_LIB=$PWD/$lib timeout -k 10 120 python bench.py --workload $k --pred-len 5 --steps 20 --warmup 3 --skip-cpu-baseline --skip-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib'.split('/')[-1], d['config']['workload'][:5], round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
  done
done
