#!/bin/bash
# the rocprofv3 passes behind profiles/r03_*: run on the GPU box from the repo root; output under gpurun_out/prof_r03/
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_r03; mkdir -p $O
COMMON="--steps 12 --warmup 3 --repeats 1 --no-cpu --no-tracker --profile-steps 0 --prime-steps 0 --sequential-scans 0 --cached-plan-steps 0"
rocprofv3 --kernel-trace --output-format csv -d $O/b1 -- python3 bench.py $COMMON --inflight 1 --batch 1 > $O/b1.json 2> $O/b1.err
python3 tools/summarize_prof.py steady $O/b1 $O/r03_rocprof_steady_B1.md 8 > /dev/null
echo "steady B1 done"
rocprofv3 --kernel-trace --output-format csv -d $O/b44 -- python3 bench.py $COMMON --inflight 4 --batch 8 > $O/b44.json 2> $O/b44.err
python3 tools/summarize_prof.py steady $O/b44 $O/r03_rocprof_steady_4x8.md 12 > /dev/null
echo "steady 4x8 done"
rocprofv3 --kernel-trace --output-format csv -d $O/b18 -- python3 bench.py $COMMON --inflight 1 --batch 8 > $O/b18.json 2> $O/b18.err
python3 tools/summarize_prof.py steady $O/b18 $O/r03_rocprof_steady_1x8.md 12 > /dev/null
echo "steady 1x8 done"
PM="--steps 4 --warmup 2 --repeats 1 --no-cpu --no-tracker --profile-steps 0 --prime-steps 0 --sequential-scans 0 --cached-plan-steps 0 --inflight 1 --batch 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 bench.py $PM > $O/pf.json 2> $O/pf.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 bench.py $PM > $O/pw.json 2> $O/pw.err
python3 tools/summarize_prof.py pmc $O/pf $O/pw $O/r03_pmc_traffic.json 3 > $O/pmc_top.txt
echo "pmc done"
rocprofv3 --kernel-trace --output-format csv -d $O/trk -- python3 tools/diag/tracker_prof.py 200 > $O/trk.log 2> $O/trk.err
python3 tools/summarize_prof.py kernel $O/trk $O/r03_rocprof_tracker.md > /dev/null
echo "tracker done"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/node -- python3 tools/diag/node_run.py 100 > $O/node.log 2> $O/node.err
python3 tools/diag/node_run.py chain $O/node > $O/r03_tracker_node_chain.txt
echo "node chain done"
rocprofv3 --kernel-trace --output-format csv -d $O/sq -- python3 tools/diag/seq_run.py 24 > $O/seq.log 2> $O/seq.err
python3 tools/diag/gn_trace.py $O/sq > $O/r03_sequential_chain.txt
echo "sequential done"
# keep only the summaries (the raw traces are large)
rm -rf $O/b1 $O/b44 $O/b18 $O/pf $O/pw $O/trk $O/sq $O/node
ls -la $O
