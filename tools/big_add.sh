#!/bin/bash
# The f32 add at N = 2^28 .. 2^31 as ONE launch and as pieces of 2^22 .. 2^26 vectors (SMHIP_PIECE_LOG2VEC; 0 = never split):
# which piece size recovers the large-array sag?   bash tools/big_add.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for lg in 28 30 31; do
  for piece in 0 22 23 24 25 26; do
    r=$(SMHIP_PIECE_LOG2VEC=$piece timeout -k 10 120 python bench.py --log2n $lg --steps 20 --warmup 3 --configs none --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f us  %.1f %%' % (d['roofline']['kernel_ms']*1000, d['roofline']['frac']*100))")
    echo "add f32 N=2^$lg  piece=2^$piece vectors (0 = one launch): $r"
  done
done
for lg in 30; do
  for piece in 0 24; do
    r=$(SMHIP_PIECE_LOG2VEC=$piece timeout -k 10 120 python bench.py --workload pow --log2n $lg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f us  %.1f %%' % (d['roofline']['kernel_ms']*1000, d['roofline']['frac']*100))")
    echo "pow f32 N=2^$lg (heavy tile kernel; two streams: by default split only above 2 GiB per operand) piece=$piece: $r"
  done
done
