#!/bin/bash
# [r06] the weight-gradient changes after the 2-GiB fix (every arm now reads all 40 960 rows of dY), three arms alternated in one call:
#   a  tn_wide=0                       : 128x128 tiles everywhere (head: whole tiles + row-split tail; layers: FFN x 2 + attention pair)
#   b  tn_wide=1 DALLE_WGRAD_GROUP4=0  : + the head's gradient as a gang stream-K on 128x256 tiles
#   c  tn_wide=1                       : + the four gradients of a block as one grouped launch on 128x256 tiles (the product)
for rep in 1 2 3; do
  for arm in "DALLE_HIP_OPTIONS=tn_wide=0" "DALLE_HIP_OPTIONS=tn_wide=1 DALLE_WGRAD_GROUP4=0" "DALLE_HIP_OPTIONS=tn_wide=1"; do
    out=$(env $arm python bench.py --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | tail -1)
    echo "[$arm] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step=%.3f value=%.0f" % (d["ms_per_step"], d["value"]))')"
  done
done
