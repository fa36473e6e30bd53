"""rocprofv3 counter CSVs (two passes: FETCH_SIZE, WRITE_SIZE) of tools/pmc_workload.py -> profiles/r<N>_flash_pmc_traffic.json + .md.
Unit and gfx950 correction come from the calibration kernel in the same trace (MI355X_MICROARCH.md, HBM section): the elementwise
multiply reads and writes 2 013 265 920 bytes each; whatever factor maps its counter values to those bytes is applied to the kernel."""
import csv
import glob
import json
import os
import statistics
import sys

KNOWN = 503316480 * 4.0


def load(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


def med(rows, pred):
    v = [x for k, x in rows if pred(k)]
    return (statistics.median(v), len(v)) if v else (None, 0)


def main(fetch_dir, write_dir, out_prefix, kernel_substr, kernel_label):
    fr, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    is_cal = lambda k: "elementwise" in k and "Mul" in k or "mul" in k.lower() and "vectorized" in k
    is_k = lambda k: kernel_substr in k
    cal_f, _ = med(fr, is_cal)
    cal_w, _ = med(wr, is_cal)
    k_f, nf = med(fr, is_k)
    k_w, nw = med(wr, is_k)
    if None in (cal_f, cal_w, k_f, k_w):
        print("missing rows:", cal_f, cal_w, k_f, k_w, sorted({k for k, _ in fr})[:20])
        sys.exit(1)
    ff, fw = KNOWN / cal_f, KNOWN / cal_w               # bytes per counter unit, per direction
    rd, wrb = k_f * ff, k_w * fw
    rec = {"kernel": kernel_label, "kv_len": 16384, "groups": 32, "hbm_bytes_per_launch": rd + wrb, "read_bytes": rd, "write_bytes": wrb,
           "fetch_counter_median": k_f, "write_counter_median": k_w, "launches": [nf, nw],
           "calibration": {"known_bytes_each_way": KNOWN, "fetch_counter": cal_f, "write_counter": cal_w,
                           "bytes_per_fetch_unit": ff, "bytes_per_write_unit": fw},
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) of tools/pmc_workload.py, scaled by the "
                     "calibration stream in the same trace (tools/pmc_traffic.py); algorithmic bytes 1.342e9"}
    json.dump(rec, open(out_prefix + ".json", "w"), indent=1)
    with open(out_prefix + ".md", "w") as f:
        f.write(f"# HBM traffic of `{kernel_label}` at the level-0 launch shape of BASELINE config 2 (rocprofv3 PMC)\n\n")
        f.write("| | FETCH_SIZE (counter units, median) | WRITE_SIZE | bytes read | bytes written |\n|---|---|---|---|---|\n")
        f.write(f"| calibration (2.013 GB each way) | {cal_f:.0f} | {cal_w:.0f} | {KNOWN:.4g} | {KNOWN:.4g} |\n")
        f.write(f"| {kernel_label} ({nf} / {nw} launches) | {k_f:.0f} | {k_w:.0f} | {rd:.4g} | {wrb:.4g} |\n\n")
        f.write(f"bytes per FETCH unit {ff:.1f}, per WRITE unit {fw:.1f}; traffic = {rd + wrb:.4g} B per launch vs 1.342e9 algorithmic "
                f"({(rd + wrb) / 1.342e9:.2f}x).\n")
    print(json.dumps(rec))


if __name__ == "__main__":
    main(*sys.argv[1:6])
