#!/bin/bash
# rocprofv3 evidence for the product kernels only (no stock-PyTorch comparison legs, no CPU
# baselines): kernel-trace statistics per workload and three SQ counter passes per workload
# (MFMA / VALU / LDS / wait counters; at most 8 SQ counters and 2 GRBM counters per pass).
# Round 3: adds the 12-head / 35-slot checkpoint (large launch shape of the Transformer kernel).
# Usage (on the GPU box, from the repo root): bash tools/profile_r3.sh <tag>
# then:  python tools/summarize_counters.py gpurun_out/prof_<tag> <tag>
set -u
TAG=${1:-r3}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PASS_A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
PASS_B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU"
PASS_C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM"
declare -A WL
WL[informer]=""
WL[informer_nomaps]="--no-attention-maps"
WL[informer12]="--workload informer12"
WL[lstm]="--workload lstm --pred-len 5"
WL[gru]="--workload gru --pred-len 5"
WL[rnn]="--workload rnn --pred-len 5"
WL[lmmse]="--workload lmmse --batch 8388608"
WL[lmmse_block]="--workload lmmse --batch 8388608 --h-share 14"
for w in informer informer_nomaps informer12 lstm gru rnn lmmse lmmse_block; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $REPO/bench.py --steps 200 --warmup 50 --skip-cpu-baseline --skip-secondary ${WL[$w]} > $OUT/trace_$w.log 2>&1 || { tail -5 $OUT/trace_$w.log; exit 1; }
  for p in A B C; do
    eval "CNT=\$PASS_$p"
    timeout -s KILL 150 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc_${w}_$p -- python3 $REPO/bench.py --steps 3 --warmup 1 --skip-cpu-baseline --skip-secondary ${WL[$w]} > $OUT/pmc_${w}_$p.log 2>&1 || { tail -5 $OUT/pmc_${w}_$p.log; exit 1; }
  done
  echo "profiled $w"
done
# the medium-batch recurrent kernels: kernel trace and the MFMA pass
WL[gru_b4096]="--workload gru --pred-len 5 --batch 4096"
WL[gru_b2048]="--workload gru --pred-len 5 --batch 2048"
WL[gru_b1024]="--workload gru --pred-len 5 --batch 1024"
for w in gru_b4096 gru_b2048 gru_b1024; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $REPO/bench.py --steps 200 --warmup 50 --skip-cpu-baseline --skip-secondary ${WL[$w]} > $OUT/trace_$w.log 2>&1 || { tail -5 $OUT/trace_$w.log; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc $PASS_A --output-format csv -d $OUT/pmc_${w}_A -- python3 $REPO/bench.py --steps 3 --warmup 1 --skip-cpu-baseline --skip-secondary ${WL[$w]} > $OUT/pmc_${w}_A.log 2>&1 || { tail -5 $OUT/pmc_${w}_A.log; exit 1; }
  echo "profiled $w"
done
for c in FETCH_SIZE WRITE_SIZE; do
  for w in informer informer_nomaps informer12 lstm lmmse lmmse_block; do
    timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${w}_$c -- python3 $REPO/bench.py --steps 3 --warmup 1 --skip-cpu-baseline --skip-secondary ${WL[$w]} > $OUT/pmc_${w}_$c.log 2>&1 || { tail -5 $OUT/pmc_${w}_$c.log; exit 1; }
  done
done
# summaries next to the raw passes (gpurun_out/ travels back; copy them into profiles/ afterwards);
# the per-dispatch traces are large and only needed for the summary
mkdir -p $OUT/summary
cd $REPO && python3 tools/summarize_counters.py $OUT $TAG $OUT/summary
find $OUT -name '*_kernel_trace.csv' -delete
