#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
OUT=/root/repo/gpurun_out/prof_r04v
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python /root/repo/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $OUT.log 2>&1)
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
idx=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
lo,hi=idx[-2]+1,idx[-1]+1
t0=int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    n=r["Kernel_Name"]
    if "sort_" in n or "nt8p_kernel<0>" in n or "attn_fwd" in n or "nt8p_kernel<3>" in n:
        s=int(r["Start_Timestamp"])-t0; e=int(r["End_Timestamp"])-t0
        print(f"{s/1e3:9.1f} .. {e/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  {n[:44]}")
PY
rm -rf $OUT
