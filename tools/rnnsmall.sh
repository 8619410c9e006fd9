# Small- and medium-batch rollout latency (hidden-split kernel, in chunks beyond one launch, against the
# batch-tiled one, and the automatic choice): tools/rnnsmall.sh [batch ...]
set -e
for k in gru lstm rnn; do
  for b in ${@:-16 64 256}; do
    for v in split tiled auto; do
      CET_RNN_KERNEL=${v/auto/} timeout -k 10 120 python bench.py --workload $k --batch $b --pred-len 5 --steps 50 --warmup 5 --skip-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k B', $b, '$v', round(d['roofline']['kernel_ms'],4), 'ms', round(d['value']), 'pred/s', 'dNMSE', d['nmse_check']['max_abs_delta_db'])"
    done
  done
done
