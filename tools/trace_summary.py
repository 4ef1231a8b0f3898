"""Per-kernel / per-grid breakdown of the last N steps of a rocprofv3 kernel_trace.csv of bench.py (steps are delimited
by the adam_kernel launch): python tools/trace_summary.py <kernel_trace.csv> [nsteps]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
lo, hi = idx[-nsteps - 1] + 1, idx[-1] + 1
acc = collections.OrderedDict()
for r in rows[lo:hi]:
    key = (r["Kernel_Name"][:46], r["Grid_Size_X"], r["Grid_Size_Y"])
    a = acc.setdefault(key, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in acc.values()) / nsteps
span = (int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / nsteps
print(f"kernel time / step {tot/1e6:.3f} ms, wall span / step {span/1e6:.3f} ms (profiled run)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if v[1] / nsteps < 4000:
        continue
    print(f"{k[0]:46s} grid=({k[1]},{k[2]}) n/step={v[0]/nsteps:5.1f} avg={v[1]/v[0]/1e3:8.1f} us  per-step={v[1]/nsteps/1e3:8.1f} us")
