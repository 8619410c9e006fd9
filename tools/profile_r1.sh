#!/bin/bash
# Collects the rocprofv3 evidence for one round: kernel-trace statistics of the default bench
# run, and HBM traffic counters (separate passes) for the Transformer and LMMSE kernels.
# Usage (on the GPU box, from the repo root): bash tools/profile_r1.sh <tag>
set -u
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 20 --warmup 3 > $OUT/bench_trace.log 2>&1 || exit 1
# the per-dispatch trace is large (the default run also times stock PyTorch eager on the GPU); only
# its statistics are kept
find $OUT/trace -name '*_kernel_trace.csv' -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_informer_$c -- python3 $REPO/bench.py --steps 3 --warmup 1 --skip-cpu-baseline --skip-secondary > $OUT/pmc_informer_$c.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_informer_nomaps_$c -- python3 $REPO/bench.py --steps 3 --warmup 1 --skip-cpu-baseline --skip-secondary --no-attention-maps > $OUT/pmc_informer_nomaps_$c.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_lmmse_$c -- python3 $REPO/bench.py --workload lmmse --batch 8388608 --steps 3 --warmup 1 --skip-cpu-baseline > $OUT/pmc_lmmse_$c.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_lstm_$c -- python3 $REPO/bench.py --workload lstm --pred-len 5 --steps 3 --warmup 1 --skip-cpu-baseline > $OUT/pmc_lstm_$c.log 2>&1 || exit 1
done
cd $REPO && python3 tools/summarize_profile.py $OUT $TAG
