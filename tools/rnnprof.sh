# Per-phase clocks of the recurrent rollout kernel (profiling build of the library).
set -e
for k in rnn gru lstm; do
  CET_LIB=$PWD/channelestimationtransformer_amd/csrc/libcet_prof.so timeout -k 10 120 python bench.py --workload $k --pred-len 5 --steps 1 --warmup 1 --skip-cpu-baseline --skip-secondary 2>&1 | grep "rnn prof" | tail -9
done
