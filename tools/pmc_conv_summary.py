#!/usr/bin/env python
"""Markdown summary of the rocprofv3 PMC passes over tools/pmc_conv_workload.py (conv vs dense persistent GEMM at equal M x N x K).
Usage: python tools/pmc_conv_summary.py <label>=<sq dir>,<misc dir> [<label>=<sq dir>,<misc dir> ...]
Per launch (the LAST of the three launches of each shape): duration, TFLOP/s, matrix-pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES over
4 SIMDs x 256 CUs x chip cycles), non-MFMA VALU and SALU instructions per wave and K-tile, LDS busy share, wait shares."""
import collections
import csv
import glob
import sys

SHAPES = {  # Grid_Size of the persistent launch -> (label, M, N, K)
}


def load(d, sub="gemm_pp"):
    rows, disp, durs = collections.defaultdict(dict), {}, {}
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
                disp[r["Dispatch_Id"]] = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                durs[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    order = sorted(rows, key=lambda x: int(x))
    return [(disp[i], durs.get(i, float("nan")), rows[i]) for i in order]


def main():
    # the workload launches, per round: conv L0, dense L0, conv L1, dense L1, conv L3, dense L3 (three rounds); take the last round
    geo = [("level 0: 64x64, 320 -> 320 (+res)", 524288, 320, 2880), ("level 1: 32x32, 640 -> 640 (+res)", 131072, 640, 5760),
           ("level 3: 8x8, 1280 -> 1280 (+res)", 8192, 1280, 11520)]
    for spec in sys.argv[1:]:
        label, dirs = spec.split("=")
        dsq, dmisc = dirs.split(",")
        sq, misc = load(dsq), load(dmisc)
        sq = [r for r in sq if "splitk_reduce" not in r[0]][-6:]
        misc = [r for r in misc if "splitk_reduce" not in r[0]][-6:]
        print(f"### {label}\n")
        print("| shape | kernel | duration | TFLOP/s | effective clock | matrix pipe busy | non-MFMA VALU / wave / K-tile | SALU / wave / K-tile | LDS busy | SQ_WAIT_ANY | SQ_WAIT_INST_LDS |")
        print("|---|---|---|---|---|---|---|---|---|---|---|")
        for i, ((k, dur, c), (k2, dur2, m)) in enumerate(zip(sq, misc)):
            name, M, N, K = geo[i // 2]
            waves = c.get("SQ_WAVES", 0) or 1
            nk_total = (M / 256) * (N / (320 if N % 320 == 0 and "5," in k else 256)) * (K / 64) * 8    # K-tiles x waves per tile
            chip_cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * 256 * chip_cyc) * (dur2 / dur) if chip_cyc else float("nan")
            valu = (c.get("SQ_INSTS_VALU", 0) - c.get("SQ_INSTS_MFMA", 0)) / nk_total
            salu = m.get("SQ_INSTS_SALU", 0) / nk_total
            lds = m.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * chip_cyc) if chip_cyc else float("nan")
            wait = c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
            wlds = m.get("SQ_WAIT_INST_LDS", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
            print(f"| {name} | `{k}` ({'conv' if i % 2 == 0 else 'dense, same M x N x K'}) | {dur:.0f} us | {2.0 * M * N * K / dur / 1e6:.0f} | "
                  f"{chip_cyc / dur2 / 1e3:.2f} GHz | {100 * busy:.1f} % | {valu:.0f} | {salu:.0f} | {100 * lds:.1f} % | {100 * wait:.1f} % | {100 * wlds:.1f} % |")
        print()


if __name__ == "__main__":
    main()
