#!/bin/bash
# The tile kernel's patches (SMHIP_TILE_QB: 64 x 512 B / 64 x 1024 B) under its two walks (SMHIP_TILE_ORDER: 1 diagonal,
# 0 row-major) over the shapes of tools/tile_shapes.py: the table behind the plan's choice (tile kernel).
#   bash tools/tile_variants.sh <tag> [fine]        on the GPU box; writes gpurun_out/<tag>/tile_variants.txt
set -o pipefail
tag=${1:-r03}; out=gpurun_out/$tag; mkdir -p $out
for wide in 0 1; do for order in 1 0; do
  echo "## patch row $((512 * (wide + 1))) B, order $order" >> $out/tile_variants.txt
  SMHIP_TILE_QB=$((512 * (wide + 1))) SMHIP_TILE_ORDER=$order timeout -k 10 110 python tools/tile_shapes.py - $2 >> $out/tile_variants.txt 2>&1 || exit 1
done; done
tail -80 $out/tile_variants.txt
