#!/bin/bash
# [r06] the gang stream-K head gradient on the secondary shapes (n_embd 1024 / 2048: gangs of 8 / 16 blocks), same call, alternated
for rep in 1 2; do
  for m in "dalle_coco --batch 16 --steps 30 --warmup 5" "1.3B --steps 8 --warmup 2"; do
    for o in 0 1; do
      out=$(DALLE_HIP_OPTIONS=tn_wide=$o python bench.py --no-cpu-baseline --model $m 2>/dev/null | tail -1)
      echo "[$m tn_wide=$o] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step=%.3f" % d["ms_per_step"])')"
    done
  done
done
