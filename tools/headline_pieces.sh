#!/bin/bash
# The headline (f32 add, N = 2^28) as one launch and as 2 / 4 / 8 pieces, five alternating rounds: is the piecewise form faster there too?
for round in 1 2 3 4 5; do
  for piece in 0 25 24 23; do
    r=$(SMHIP_PIECE_LOG2VEC=$piece timeout -k 10 120 python bench.py --steps 100 --warmup 10 --configs none --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us  %.2f %%' % (d['roofline']['kernel_ms']*1000, d['roofline']['frac']*100))")
    echo "round $round  piece=2^$piece vectors (0 = one launch): $r"
  done
done
