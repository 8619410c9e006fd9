"""Turns one tools/profile_r2.sh / profile_r3.sh output directory into the files kept under profiles/.

Usage: python tools/summarize_counters.py gpurun_out/prof_<tag> <tag> [output directory]
Writes
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats rows of the product kernels
                                    (one bench.py invocation per workload, no comparison legs)
  profiles/<tag>_mfma_util.json     per kernel: raw SQ counter means per launch and the derived
                                    shares (MFMA pipe busy, VALU issue, waits, LDS), with formulas
  profiles/<tag>_hbm_traffic.json   FETCH_SIZE / WRITE_SIZE passes -> bytes per launch
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 256 * 4

INFORMER_IO = 25 * 16 * 4 + 15 * 16 * 4 + 5 * 16 * 4
INFORMER_MAPS = 8 * sum(L * L for L in (25, 14, 8, 5, 5, 4, 3)) * 4
# workload -> (kernel substring, bench.py arguments, units per launch, algorithmic bytes per unit)
WORKLOADS = {
    "informer": ("informer_kernel<2, 1, 1, true, 4, 2>", "default (B=4096, output_attention=True)", 4096, INFORMER_IO + INFORMER_MAPS),
    "informer_nomaps": ("informer_kernel<2, 1, 1, true, 4, 2>", "--no-attention-maps (B=4096)", 4096, INFORMER_IO),
    "informer12": ("informer_kernel<3, 2, 1, false, 4, 3>", "--workload informer12 (B=4096, 12 heads, 35 slots)", 4096,
                   (35 + 25 + 5) * 16 * 4),
    "lstm": ("rnn_rollout_kernel", "--workload lstm --pred-len 5 (B=8192)", 8192, 25 * 16 * 4 + 5 * 16 * 4),
    "gru": ("rnn_rollout_kernel", "--workload gru --pred-len 5 (B=8192)", 8192, 25 * 16 * 4 + 5 * 16 * 4),
    "rnn": ("rnn_rollout_kernel", "--workload rnn --pred-len 5 (B=8192)", 8192, 25 * 16 * 4 + 5 * 16 * 4),
    "gru_b4096": ("rnn_mid_kernel", "--workload gru --pred-len 5 --batch 4096 (half tiles)", 4096, 25 * 16 * 4 + 5 * 16 * 4),
    "gru_b2048": ("rnn_mid_shared_kernel", "--workload gru --pred-len 5 --batch 2048 (half tiles on 2 workgroups)", 2048,
                  25 * 16 * 4 + 5 * 16 * 4),
    "gru_b1024": ("rnn_mid_shared_kernel", "--workload gru --pred-len 5 --batch 1024 (half tiles on 4 workgroups)", 1024,
                  25 * 16 * 4 + 5 * 16 * 4),
    "lmmse": ("lmmse_tiled", "--workload lmmse --batch 8388608 (16x4)", 8388608, 688),
    "lmmse_block": ("lmmse_block", "--workload lmmse --batch 8388608 --h-share 14", 8388608, 688 - 512 * 13 / 14),
}


def counter_means(directory, kernel_substr):
    """{counter: (mean per launch over the timed launches, launches)} of one pmc pass. Only the
    launches of the benchmark batch count (the largest grid of that kernel in the run; bench.py
    also launches it on the 192 stored sequences of its NMSE check)."""
    rows = []
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if kernel_substr in r["Kernel_Name"]]
    grid = max((int(r["Grid_Size"]) for r in rows), default=0)
    vals = {}
    for r in rows:
        if int(r["Grid_Size"]) == grid:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {}
    for k, v in vals.items():
        if len(v) > 1:
            v = v[1:]  # drop the warm-up launch
        out[k] = (sum(v) / len(v), len(v))
    return out


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out_dir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)

    # ---- kernel-trace statistics, product kernels only ------------------------------------------
    rows, header = [], None
    durations = {}
    for w, (kern, args, units, _alg) in WORKLOADS.items():
        for f in glob.glob(os.path.join(src, f"trace_{w}", "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                header = header or (["workload"] + list(r.keys()))
                if "cet::" in r["Name"]:
                    rows.append([w] + list(r.values()))
                    if kern in r["Name"] and w not in durations:
                        durations[w] = float(r["AverageNs"])
        # the per-dispatch trace, when still there: mean over the benchmark-size launches only
        for f in glob.glob(os.path.join(src, f"trace_{w}", "**", "*kernel_trace.csv"), recursive=True):
            d = [(int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]),
                  int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                 for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
            if d:
                g = max(x[0] for x in d)
                sel = [x[1] for x in d if x[0] == g][3:] or [x[1] for x in d if x[0] == g]
                durations[w] = sum(sel) / len(sel)
    if rows:
        with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(header)
            wr.writerows(rows)

    # ---- SQ counters --------------------------------------------------------------------------------
    util = {"_how": (
        "rocprofv3 --pmc, three passes per workload (tools/profile_r3.sh; round 2: profile_r2.sh), bench.py --skip-secondary "
        "--skip-cpu-baseline; values are means per kernel launch over the timed launches. "
        "SQ_*_CYCLES of waves (WAVE_CYCLES, WAIT_*, ACTIVE_INST_*) count quad-cycles summed over waves; "
        "SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES is summed over the "
        "shader engines' SQs. Derived: mfma_busy_share_of_kernel = MFMA_BUSY / (1024 SIMDs x kernel "
        "cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (sum over the 8 XCDs); mfma_cycles_per_mfma = "
        "MFMA_BUSY / INSTS_MFMA; valu_share / lds_share / wait shares = the wave-cycle counter over "
        "SQ_WAVE_CYCLES.")}
    for w, (kern, args, units, _alg) in WORKLOADS.items():
        c = {}
        for p in "ABC":
            c.update(counter_means(os.path.join(src, f"pmc_{w}_{p}"), kern))
        if not c:
            continue
        g = lambda k: c[k][0] if k in c else None  # noqa: E731
        d = {"config": "bench.py " + args, "units_per_launch": units,
             "raw_mean_per_launch": {k: v[0] for k, v in sorted(c.items())},
             "launches_averaged": min(v[1] for v in c.values())}
        if w in durations:
            d["avg_duration_us_kernel_trace"] = durations[w] / 1e3
        der = {}
        if g("GRBM_GUI_ACTIVE") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            kcyc = g("GRBM_GUI_ACTIVE") / 8
            der["kernel_cycles"] = kcyc
            der["mfma_busy_share_of_kernel"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (SIMDS * kcyc)
            if w in durations:
                der["shader_clock_ghz"] = kcyc / durations[w]
        if g("SQ_INSTS_MFMA"):
            der["mfma_cycles_per_mfma"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_INSTS_MFMA")
            der["mfma_instructions_per_unit"] = g("SQ_INSTS_MFMA") / units
            if g("SQ_INSTS_VALU"):
                der["valu_instructions_per_unit_incl_mfma"] = g("SQ_INSTS_VALU") / units
        wc = g("SQ_WAVE_CYCLES")
        if wc:
            for name, key in (("active_any_share", "SQ_ACTIVE_INST_ANY"), ("valu_share_incl_mfma", "SQ_ACTIVE_INST_VALU"),
                              ("lds_share", "SQ_ACTIVE_INST_LDS"), ("scalar_share", "SQ_ACTIVE_INST_SCA"),
                              ("wait_any_share", "SQ_WAIT_ANY"), ("wait_inst_any_share", "SQ_WAIT_INST_ANY"),
                              ("wait_inst_lds_share", "SQ_WAIT_INST_LDS")):
                if g(key) is not None:
                    der[name + "_of_wave_cycles"] = g(key) / wc
        if g("SQ_VALU_MFMA_COEXEC_CYCLES") is not None and g("SQ_VALU_MFMA_BUSY_CYCLES"):
            der["valu_mfma_coexec_over_mfma_busy"] = g("SQ_VALU_MFMA_COEXEC_CYCLES") / g("SQ_VALU_MFMA_BUSY_CYCLES")
        if g("SQ_LDS_IDX_ACTIVE"):
            der["lds_bank_conflict_share_of_lds_cycles"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        d["derived"] = der
        util[w] = d
    with open(os.path.join(out_dir, f"{tag}_mfma_util.json"), "w") as f:
        json.dump(util, f, indent=1)

    # ---- HBM traffic ------------------------------------------------------------------------------------
    traffic = {}
    for w, (kern, args, units, alg) in WORKLOADS.items():
        fs = counter_means(os.path.join(src, f"pmc_{w}_FETCH_SIZE"), kern).get("FETCH_SIZE")
        ws = counter_means(os.path.join(src, f"pmc_{w}_WRITE_SIZE"), kern).get("WRITE_SIZE")
        if not fs or not ws:
            continue
        fb, wb = fs[0] * 1024 * 2, ws[0] * 1024
        traffic[w] = {"raw": {"FETCH_SIZE": {"launches": fs[1], "mean_raw_KiB": fs[0]},
                              "WRITE_SIZE": {"launches": ws[1], "mean_raw_KiB": ws[0]}},
                      "fetch_bytes_corrected": fb, "write_bytes": wb, "traffic_bytes_per_launch": fb + wb,
                      "config": "bench.py " + args, "algorithmic_bytes_per_launch": alg * units, "units": units}
    traffic["_how"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                       "(tools/profile_r3.sh; round 2: profile_r2.sh); bytes = FETCH_SIZE*1024*2 (gfx950 counts wide 128-B "
                       "reads as 64 B) + WRITE_SIZE*1024; mean over the timed launches")
    if len(traffic) > 1:
        with open(os.path.join(out_dir, f"{tag}_hbm_traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)
    print(json.dumps({w: util[w]["derived"] for w in util if w != "_how"}, indent=1))


if __name__ == "__main__":
    main()
