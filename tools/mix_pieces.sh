#!/bin/bash
# Which stream mixes gain from going out in pieces, and from which size?  1R+1W (a*s), 1R (sum), 2R (dot) at 2^28 .. 2^30 f32 elements,
# one launch / pieces of 2^25 / 2^24 vectors (the 2R+1W add: tools/big_add.sh, tools/headline_pieces.sh, tools/mid_pieces.sh).
for kind in scalar sum dot; do
  for lg in 28 29 30; do
    for piece in 0 25 24; do
      echo -n "piece=2^$piece  "; SMHIP_PIECE_LOG2VEC=$piece timeout -k 10 100 python tools/mix_pieces.py $kind $lg 2>&1 | tail -1
    done
  done
done
