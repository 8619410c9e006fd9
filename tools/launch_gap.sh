# Kernel durations and the gaps between back-to-back launches of the Transformer kernel, from the
# rocprofv3 kernel trace (timestamps): tools/launch_gap.sh  (on the GPU box, from the repo root)
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/gap
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for b in 2048 4096 8192; do
  timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/b$b -- python3 $REPO/bench.py --steps 30 --warmup 5 --batch $b --skip-cpu-baseline --skip-secondary --no-attention-maps > $OUT/b$b.log 2>&1 || { tail -5 $OUT/b$b.log; exit 1; }
  f=$(find $OUT/b$b -name '*kernel_trace.csv' | head -1)
  python3 - "$f" $b <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "informer_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = max(int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) for r in rows)
rows = [r for r in rows if (int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])) == big]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
gap = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
gap = [g for g in gap if g < 200000]  # back-to-back launches only
print("B", sys.argv[2], "launches", len(rows), "kernel us: mean %.1f min %.1f" % (sum(dur) / len(dur) / 1e3, min(dur) / 1e3),
      "gap us: mean %.1f min %.1f (n=%d)" % (sum(gap) / max(1, len(gap)) / 1e3, min(gap) / 1e3 if gap else 0, len(gap)))
PY
  find $OUT/b$b -name '*kernel_trace.csv' -delete
done
