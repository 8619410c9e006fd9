# Headline step time against the length of the clock-settle phase and the number of timed steps.
set -e
for cfg in "0 5 20" "80 5 20" "300 5 20" "1000 5 20" "80 50 200" "1000 50 200" "80 5 20" "0 5 20"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --clock-settle-ms $1 --warmup $2 --steps $3 --skip-cpu-baseline --skip-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('settle $1 warmup $2 steps $3:', round(d['roofline']['kernel_ms'],4), 'ms  wall/step', round(d['ms_per_step'],4))"
done
