#!/bin/bash
# diagnostic: quick headline with and without an environment switch, alternating, on one box.  usage: env_ab.sh NAME=VALUE [passes]
KV=$1; N=${2:-2}
for p in $(seq 1 $N); do
  echo "== pass $p: without $KV"; bash tools/diag/quick_headline.sh || exit 1
  echo "== pass $p: with $KV"; env $KV bash tools/diag/quick_headline.sh || exit 1
done
