# Where the workgroup kernel overtakes the one-wave kernel: both pinned at several batch sizes.
set -e
for b in 256 512 768 1024 1280 1536 2048; do
  for k in one_wave workgroup; do
    CET_INFORMER_KERNEL=$k timeout -k 10 120 python bench.py --batch $b --steps 50 --warmup 5 --skip-cpu-baseline --skip-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', $b, '$k', round(d['roofline']['kernel_ms'],4), 'ms', round(d['value']))"
  done
done
