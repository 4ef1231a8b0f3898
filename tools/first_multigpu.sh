#!/bin/bash
# The first run on a multi-GPU MI355X node: one call, a scaling curve AND the reserve_cus knob.
#   1. the armed 2-GPU RCCL test (the product transport with N > 1 for the first time)
#   2. weak scaling at N = 1, 2, 4, 8 with the default (reserve_cus = 0)
#   3. N = 8 weak with 8 and 16 CUs left to the RCCL channels during the backward, and N = 8 strong (global batch 32)
# Every bench line is one JSON object (bench.py's contract); they are collected in gpurun_out/first_multigpu.jsonl.
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
N=${1:-8}
OUT=gpurun_out/first_multigpu.jsonl
mkdir -p gpurun_out; : > $OUT
python -m pytest tests/test_bench_dp_gpu.py -q -k "rccl_two_gpus" 2>&1 | tail -3
run() {  # run <gpus> <extra bench args...>
  local n=$1; shift
  if [ "$n" = 1 ]; then python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline "$@" | tail -1 >> $OUT
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) \
         bench.py --gpus $n --steps 50 --warmup 10 --no-cpu-baseline "$@" | tail -1 >> $OUT; fi
}
for n in 1 2 4 $N; do [ $n -le $N ] && run $n; done
run $N --reserve-cus 8
run $N --reserve-cus 16
run $N --scaling strong
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/first_multigpu.jsonl") if l.strip().startswith("{")]
base = next((r["value"] for r in rows if r["n_gpus"] == 1), None)
for r in rows:
    c = r["config"]
    print(f"N={r['n_gpus']} {r['scaling']:6s} reserve_cus={c.get('dp_reserve_cus')} transport={c.get('dp_transport')} "
          f"{r['ms_per_step']:8.3f} ms/step {r['value']/1e6:7.3f} M tokens/s" + (f"  x{r['value']/base:5.2f} vs N=1" if base else ""))
PY
