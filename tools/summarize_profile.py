"""Turns one tools/profile_r1.sh output directory into the files kept under profiles/.

Usage: python tools/summarize_profile.py gpurun_out/prof_<tag> <tag>
Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary of the default
bench run), profiles/<tag>_bench_under_rocprof.json (the JSON line of that run) and
profiles/<tag>_hbm_traffic.json (FETCH_SIZE / WRITE_SIZE passes -> bytes per launch).
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# pmc pass name -> (kernel substring, bench config, units per launch, algorithmic bytes per unit)
INFORMER_IO = 25 * 16 * 4 + 15 * 16 * 4 + 5 * 16 * 4  # x_enc + x_dec + pred, bytes per sequence
# self-attention maps of the 4- and 3-layer encoders of the stack: [B, 8, L, L] fp32 each
INFORMER_MAPS = 8 * sum(L * L for L in (25, 14, 8, 5, 5, 4, 3)) * 4
PASSES = {
    "informer": ("informer_kernel", "bench.py default (B=4096, output_attention=True)", 4096,
                 INFORMER_IO + INFORMER_MAPS),
    "informer_nomaps": ("informer_kernel", "bench.py --no-attention-maps (B=4096)", 4096, INFORMER_IO),
    "lmmse": ("lmmse", "bench.py --workload lmmse --batch 8388608 (16x4)", 8388608,
              (16 + 16 * 4 + 4) * 8 + 4 * 4),
    # recurrent rollout: input + predictions per sequence (the 3.25 MB weight set is read once
    # from HBM and then streamed from L2 every time step, which these counters do not see)
    "lstm": ("rnn_rollout_kernel", "bench.py --workload lstm --pred-len 5 (B=8192)", 8192,
             25 * 16 * 4 + 5 * 16 * 4),
}


def counter_mean(directory, kernel_substr, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    vals = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    if len(vals) > 1:
        vals = vals[1:]  # drop the warm-up launch
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
    log = os.path.join(src, "bench_trace.log")
    if os.path.exists(log):
        lines = [l for l in open(log).read().splitlines() if l.startswith("{")]
        if lines:
            with open(os.path.join(out, f"{tag}_bench_under_rocprof.json"), "w") as f:
                f.write(lines[-1] + "\n")
    traffic = {}
    for name, (kern, config, units, alg) in PASSES.items():
        fetch, n1 = counter_mean(os.path.join(src, f"pmc_{name}_FETCH_SIZE"), kern, "FETCH_SIZE")
        write, n2 = counter_mean(os.path.join(src, f"pmc_{name}_WRITE_SIZE"), kern, "WRITE_SIZE")
        if fetch is None or write is None:
            continue
        fb, wb = fetch * 1024 * 2, write * 1024
        traffic[name] = {
            "raw": {"FETCH_SIZE": {"launches": n1, "mean_raw_KiB": fetch},
                    "WRITE_SIZE": {"launches": n2, "mean_raw_KiB": write}},
            "fetch_bytes_corrected": fb, "write_bytes": wb, "traffic_bytes_per_launch": fb + wb,
            "config": config, "algorithmic_bytes_per_launch": alg * units, "units": units}
    traffic["_how"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                       "(tools/profile_r1.sh); bytes = FETCH_SIZE*1024*2 (gfx950 counts wide 128-B "
                       "reads as 64 B) + WRITE_SIZE*1024; mean over the timed launches")
    with open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps({k: v.get("traffic_bytes_per_launch") if isinstance(v, dict) else None
                      for k, v in traffic.items() if k != "_how"}))


if __name__ == "__main__":
    main()
