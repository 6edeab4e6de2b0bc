"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel table.

    python tools/rocpd_stats.py gpurun_out/prof_a/bench_results.db [--md profiles/xxx.md] [--skip-first-ms 0]

Equivalent of `rocprofv3 --stats` CSV output: calls, total / average / min / max duration per
kernel name, share of the total GPU time.
"""
import argparse
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void sdmi::", "").replace("sdmi::", "")
    return name[:86]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--md", default="")
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, (end - start) from kernels").fetchall() if "name" in cols else []
    if not rows:
        print("columns of `kernels`:", cols)
        sys.exit(1)
    stats = {}
    for name, dur in rows:
        s = stats.setdefault(short(name), [0, 0, 1 << 62, 0])
        s[0] += 1
        s[1] += dur
        s[2] = min(s[2], dur)
        s[3] = max(s[3], dur)
    total = sum(s[1] for s in stats.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for name, s in sorted(stats.items(), key=lambda kv: -kv[1][1])[: args.top]:
        lines.append(f"| `{name}` | {s[0]} | {s[1] / 1e6:.3f} | {s[1] / s[0] / 1e3:.2f} | {s[2] / 1e3:.2f} | {s[3] / 1e3:.2f} | {100 * s[1] / total:.2f} |")
    lines.append(f"| **total** | {sum(s[0] for s in stats.values())} | {total / 1e6:.3f} | | | | 100 |")
    out = "\n".join(lines)
    print(out)
    if args.md:
        with open(args.md, "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
