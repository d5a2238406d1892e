#!/bin/bash
# who holds the machine in the headline configuration (GPU box): kernel trace of 4 handles x 8 scans, every instant split
# between the kernels resident then -> gpurun_out/prof_r02/r02_rocprof_share_4x8.md
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_r02; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/sh -- python3 bench.py --steps 16 --warmup 3 --repeats 1 --no-cpu --no-tracker --profile-steps 0 --prime-steps 0 --sequential-scans 0 --inflight 4 --batch 8 > $O/sh.json 2> $O/sh.err
python3 tools/summarize_prof.py share $O/sh $O/r02_rocprof_share_4x8.md 24
rm -rf $O/sh
