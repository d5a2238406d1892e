"""The N > 1 control flow of bench.py over the nccl backend (= RCCL on ROCm) with a world of ONE rank, on the one GPU of the test box:
every collective bench.py issues for N > 1 — broadcast of the map, all_gather_into_tensor of the pose records per step, the MAX all_reduce
of the window time, the tracker-rate all_gather, the barriers — executes through RCCL.  It is the first and only execution of the RCCL
calls of this repository (no multi-GPU node has been available to any round: DESIGN 7); with one rank it proves the API use (tensor
shapes, dtypes, contiguity of the d_rec slices, process-group set-up and tear-down), not scaling."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_collectives_execute_over_rccl_with_one_rank(tmp_path):
    env = dict(os.environ, LVI_BENCH_RCCL_WORLD1="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547",
               HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="VERSION")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--repeats", "2", "--prime-steps", "2", "--profile-steps", "0",
           "--n-raw", "20001", "--keyframes", "8", "--kf-n-raw", "8001", "--map-points", "150000", "--pool", "4", "--inflight", "2", "--batch", "2",
           "--icp-iters", "6", "--no-cpu", "--tracker-seconds", "0.05", "--sequential-scans", "0", "--cached-plan-steps", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["results_ok"] is True, d.get("pose_err_vs_truth")
    assert d["tracker"]["per_rank"] and len(d["tracker"]["per_rank"]) == 1            # the tracker-rate all_gather ran
    # the library that served the collectives announces itself (NCCL_DEBUG=VERSION prints "NCCL version … " / RCCL's banner)
    assert ("NCCL version" in r.stdout + r.stderr) or ("RCCL" in r.stdout + r.stderr), (r.stderr[-800:], r.stdout[-300:])
