"""Side-by-side phase intervals of two stamp traces of the Transformer kernel.

usage: python tools/stampdiff.py base.txt other.txt [wave]
The files are CET_PROF_DUMP outputs of two profiling builds (libcet_prof.so) for the same call;
`wave` picks the sampled wave (line) to compare, default 0 (a primary wave; odd lines are
partner waves). Prints one row per stamp: category, clocks in either build, difference.
"""
import sys

NAMES = ["embed", "qkv", "attn", "oproj", "add+ln", "ffn", "distil", "move", "proj", "other"]


def trace(path, wave):
    lines = [l for l in open(path) if l.startswith("B=")]
    per_call = 16  # sampled waves per call; take the last call in the file
    last = lines[-per_call:] if len(lines) >= per_call else lines
    toks = last[wave].split()[2:]
    t = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in toks]
    return [(t[i][0] - t[i - 1][0], t[i][1]) for i in range(1, len(t))]


def main():
    wave = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    a, b = trace(sys.argv[1], wave), trace(sys.argv[2], wave)
    print(f"stamps: {len(a)} / {len(b)}; total clocks {sum(d for d, _ in a)} / {sum(d for d, _ in b)}")
    tot = {}
    for i in range(min(len(a), len(b))):
        (da, ca), (db, cb) = a[i], b[i]
        print(f"{i:4d} {NAMES[ca]:7s} {NAMES[cb]:7s} {da:8d} {db:8d} {db - da:+7d}")
        k = NAMES[ca]
        tot[k] = tot.get(k, 0) + db - da
    print({k: v for k, v in tot.items()})


if __name__ == "__main__":
    main()
