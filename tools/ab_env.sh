#!/bin/bash
# A/B of engine switches inside ONE gpurun call (boxes differ by +-4 %): alternates the arms, prints ms/step per run.
# usage: tools/ab_env.sh "<ENV=val ...>" "<ENV=val ...>" [reps] [bench args...]
A="$1"; B="$2"; REPS="${3:-2}"; shift 3 || true
for r in $(seq 1 "$REPS"); do
  for arm in "$A" "$B"; do
    out=$(env $arm python bench.py --no-cpu-baseline --steps 100 --warmup 20 "$@" 2>/dev/null | tail -1)
    echo "[$arm] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step=%.3f value=%.0f" % (d["ms_per_step"], d["value"]))')"
  done
done
