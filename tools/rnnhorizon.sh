# Recurrent rollout throughput at the three horizons of the long-horizon configuration.
set -e
for k in lstm gru rnn; do
  for p in 5 25 100; do
    timeout -k 10 200 python bench.py --workload $k --pred-len $p --steps 10 --warmup 2 --skip-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', 'pred_len', $p, round(d['value']), 'pred/s', round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac'],4))"
  done
done
