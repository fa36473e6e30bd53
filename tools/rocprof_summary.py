#!/usr/bin/env python
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats CSV into the markdown table kept under profiles/.
Usage: python tools/rocprof_summary.py <kernel_stats.csv> <out.md> "<title line>" [forwards]"""
import csv
import sys


def main():
    src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
    fwd = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    rows = list(csv.DictReader(open(src)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# {title}\n\n{fwd} forward(s) in the trace; durations in ms; `at::native::*` rows are the one-time synthetic weight init.\n\n")
        f.write("| kernel | calls | total ms | ms per forward | avg us | % |\n|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            name = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            t = int(r["TotalDurationNs"])
            f.write(f"| `{name[:90]}` | {r['Calls']} | {t / 1e6:.2f} | {t / 1e6 / fwd:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {100.0 * t / tot:.2f} |\n")
        f.write(f"\nTotal kernel time {tot / 1e6:.1f} ms ({tot / 1e6 / fwd:.1f} ms per forward).\n")


if __name__ == "__main__":
    main()
