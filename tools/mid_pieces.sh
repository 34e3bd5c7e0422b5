#!/bin/bash
# Does cutting a launch help below 1 GiB per operand too?  f32 add at 2^27 / 2^26 elements as 1 / 2 / 4 launches, three alternating rounds.
for round in 1 2 3; do
  for cfg in "27 0" "27 24" "27 23" "26 0" "26 23" "26 22" "29 0" "29 24" "29 25"; do
    set -- $cfg
    r=$(SMHIP_PIECE_LOG2VEC=$2 timeout -k 10 120 python bench.py --log2n $1 --steps 100 --warmup 10 --configs none --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.2f us per step  %.2f %%  (%d launches)' % (r['step_kernel_ms']*1000, r['frac']*100, r['launches_per_step']))")
    echo "round $round  N=2^$1 piece=2^$2 vectors (0 = one launch): $r"
  done
done
