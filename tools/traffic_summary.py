"""Per-launch HBM-side traffic of ONE kernel of a bench run from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 128-B requests at 64 B -> doubled for wide coalesced
reads; both counters are in KiB.
usage: traffic_summary.py <fetch.csv> <write.csv> <out.json> <kernel substring> [algorithmic bytes] [grid size in threads]
(the optional grid size selects one launch shape among several of the same kernel, e.g. one convolution layer)"""
import csv
import json
import sys

fetch_csv, write_csv, out_json = sys.argv[1:4]
KERNEL = sys.argv[4] if len(sys.argv) > 4 else "gemm_nt4_kernel<65>"   # the kernel the vocabulary projection dispatches to
ALGO = int(sys.argv[5]) if len(sys.argv) > 5 and int(sys.argv[5]) > 0 else None
GRID = sys.argv[6] if len(sys.argv) > 6 else None


def mean_counter(path, name):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name and (GRID is None or r.get("Grid_Size", r.get("Grid_Size_X")) == GRID)]
    return sum(vals) / max(len(vals), 1), len(vals)


f, nf = mean_counter(fetch_csv, "FETCH_SIZE")
w, nw = mean_counter(write_csv, "WRITE_SIZE")
if ALGO is None:
    M, d, V, Vp = 32 * 1280, 512, 50771, 50816
    ALGO = (M * d + Vp * d) * 2 + M * Vp * 2 + (Vp // 64) * M * 4   # operands + bf16 E + fp32 row-sum partials
rec = {"kernel": KERNEL, "grid": GRID, "launches": [nf, nw], "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
       "read_bytes_corrected": f * 1024 * 2, "write_bytes": w * 1024, "traffic_bytes": f * 1024 * 2 + w * 1024,
       "algorithmic_bytes": ALGO, "note": "FETCH_SIZE doubled per the gfx950 correction; WRITE_SIZE uncalibrated"}
json.dump(rec, open(out_json, "w"), indent=1)
print(json.dumps(rec))
