#!/bin/bash
# diagnostic: kernel table of one handle x 8 scans (kernels serialized) for several values of an env knob
# usage: vb_exp.sh ENVNAME v0 v1 ...   -> gpurun_out/prof_r03/exp_<ENVNAME>_<v>.md
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/prof_r03; mkdir -p $O
K=$1; shift
for v in "$@"; do
  export $K=$v
  rocprofv3 --kernel-trace --output-format csv -d $O/e_$v -- python3 bench.py --steps 12 --warmup 3 --repeats 1 --no-cpu --no-tracker --profile-steps 0 --prime-steps 0 --sequential-scans 0 --cached-plan-steps 0 --inflight 1 --batch 8 > $O/e_$v.json 2> $O/e_$v.err || exit 1
  python3 tools/summarize_prof.py steady $O/e_$v $O/exp_${K}_$v.md 12 > /dev/null
  rm -rf $O/e_$v
  echo "== $K=$v"; head -1 $O/exp_${K}_$v.md; grep -E "vb_accum|vb_merge|vb_scatter_det|vb_plan|feat_sector|icp_gn" $O/exp_${K}_$v.md
done
