#!/usr/bin/env python
"""Kernel-time table from a rocprofv3 rocpd database (ROCm 7 default output of `rocprofv3 --kernel-trace --stats`).
Usage: python tools/rocpd_summary.py <results.db> [out.md] ["title"] [top]"""
import re
import sqlite3
import sys


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else None
    title = sys.argv[3] if len(sys.argv) > 3 else src
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    cur = sqlite3.connect(src).cursor()
    rows = cur.execute("select name, count(*), sum(end - start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"# {title}", "", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for name, n, t in rows[:top]:
        nm = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")[:100]
        lines.append(f"| `{nm}` | {n} | {t / 1e6:.2f} | {t / n / 1e3:.1f} | {100.0 * t / tot:.2f} |")
    lines.append("")
    lines.append(f"Total kernel time {tot / 1e6:.1f} ms over {sum(r[1] for r in rows)} launches.")
    out = "\n".join(lines) + "\n"
    if dst:
        open(dst, "w").write(out)
    else:
        print(out)


if __name__ == "__main__":
    main()
