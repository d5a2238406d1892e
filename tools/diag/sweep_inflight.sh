#!/bin/bash
# diagnostic: headline for several (inflight, batch, map-stream) settings on one box
for cfg in "4 8 -1" "5 8 -1" "6 8 -1" "3 8 -1" "4 8 0" "4 6 -1" "4 12 -1" "8 4 -1"; do
  set -- $cfg
  echo "== inflight $1 batch $2 map-stream $3"
  bash tools/diag/quick_headline.sh --inflight $1 --batch $2 --map-stream $3 || exit 1
done
