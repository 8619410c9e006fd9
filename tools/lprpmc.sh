# VALU issue counters of the per-RE equalizer (16x4, 8,388,608 REs) with two lanes per RE (the
# shipped mapping) and with one lane per RE (experiment build: make libcet_exp.so EXP=-DCET_EXP):
#   tools/lprpmc.sh channelestimationtransformer_amd/csrc/libcet_exp.so
# writes gpurun_out/lprpmc/lmmse_lpr.json (copy to profiles/).
set -u
REPO=$PWD
LIB=$REPO/${1:-channelestimationtransformer_amd/csrc/libcet_exp.so}
OUT=$REPO/gpurun_out/lprpmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for lpr in 2 1; do
  export CET_LMMSE_LPR=$lpr CET_LIB=$LIB
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$lpr -- python3 $REPO/bench.py --steps 20 --warmup 3 --skip-cpu-baseline --skip-secondary --workload lmmse --batch 8388608 > $OUT/trace_$lpr.log 2>&1 || { tail -5 $OUT/trace_$lpr.log; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_$lpr -- python3 $REPO/bench.py --steps 3 --warmup 1 --skip-cpu-baseline --skip-secondary --workload lmmse --batch 8388608 > $OUT/pmc_$lpr.log 2>&1 || { tail -5 $OUT/pmc_$lpr.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
N = 8388608
res = {"_how": "rocprofv3 --pmc (one pass) and --kernel-trace --stats of bench.py --workload lmmse --batch 8388608 "
               "--skip-secondary --skip-cpu-baseline on the experiment build (-DCET_EXP), CET_LMMSE_LPR=2 (two lanes per "
               "RE, shipped) and =1 (one lane per RE); means per launch of lmmse_tiled<16,4,LPR>. *_CYCLES of waves are "
               "quad-cycles summed over waves."}
for lpr in (2, 1):
    f = glob.glob(f"{out}/pmc_{lpr}/**/*counter_collection.csv", recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if "lmmse_tiled" not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    raw = {k: sum(v) / len(v) for k, v in acc.items()}
    st = glob.glob(f"{out}/trace_{lpr}/**/*kernel_stats.csv", recursive=True)[0]
    dur = [float(r["AverageNs"]) for r in csv.DictReader(open(st)) if "lmmse_tiled" in r["Name"]][0]
    name = [r["Name"] for r in csv.DictReader(open(st)) if "lmmse_tiled" in r["Name"]][0]
    res[f"lanes_per_RE_{lpr}"] = {
        "kernel": name, "avg_duration_us": dur / 1e3, "G_RE_per_s": N / dur,
        "raw_mean_per_launch": raw,
        "valu_wave_instructions_per_RE": raw["SQ_INSTS_VALU"] / N,
        "valu_share_of_wave_cycles": raw["SQ_ACTIVE_INST_VALU"] / raw["SQ_WAVE_CYCLES"],
        "waves": raw["SQ_WAVES"]}
json.dump(res, open(f"{out}/lmmse_lpr.json", "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "raw_mean_per_launch"} for k, v in res.items() if k[0] != "_"}, indent=1))
PY
find $OUT -name '*_kernel_trace.csv' -delete
