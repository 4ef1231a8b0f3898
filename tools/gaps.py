"""Idle gaps of the last bench step in a rocprofv3 kernel_trace.csv: time between the end of a kernel and the start of the next
(kernels sorted by start; overlapping side-stream kernels are skipped): python tools/gaps.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
seg = sorted(rows[idx[-2] + 1: idx[-1] + 1], key=lambda r: int(r["Start_Timestamp"]))
t0 = int(seg[0]["Start_Timestamp"])
end = int(seg[0]["End_Timestamp"])
prev = seg[0]["Kernel_Name"]
gaps = []
busy = end - t0
for r in seg[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= end:
        gaps.append((s - end, prev[:40], r["Kernel_Name"][:40], (end - t0) / 1e3))
        busy += e - s
        end, prev = e, r["Kernel_Name"]
    elif e > end:
        busy += e - end
        end, prev = e, r["Kernel_Name"]
total = (end - t0) / 1e3
print(f"step span {total:.1f} us, sum of gaps {sum(g[0] for g in gaps)/1e3:.1f} us in {len(gaps)} gaps (mean {sum(g[0] for g in gaps)/max(len(gaps),1)/1e3:.2f} us)")
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  {g[0]/1e3:7.1f} us at {g[3]:9.1f} us  after {g[1]}  before {g[2]}")
