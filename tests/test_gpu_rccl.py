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


@pytest.mark.gpu
def test_native_multi_gpu_replay_one_process(tmp_path):
    """host/replay_multi.cpp — SURVEY 8(e)'s single-process form: one handle per GPU, scan i -> GPU i mod N, ncclCommInitAll communicators,
    one grouped ncclAllGather of the step's 32-byte pose records — built with hipcc against liblvi_hip.so and librccl and run with the GPUs
    the box has (one: the one-rank form of the collective).  Its records equal, bit for bit, the ones a Python-driven handle returns."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    from helpers import small_params
    pkg = graft.import_package()
    hip = pkg.load_hip()
    S = pkg.synth
    kw = small_params(Horizon_SCAN=4096, max_raw_points=9000, max_map_points=400000, icp_max_iters=10)
    g = pkg.LidarHotpath(hip, **kw)
    mc, ms = S.make_map(g, 6, 8001, seed=4711)
    n_raw, n_scans = 8001, 3
    poses = [S.loop_pose(0.3 + 0.4 * k, 0.01, -0.01) for k in range(n_scans)]
    scans = [S.make_scan(n_raw, poses[k], 200 + k) for k in range(n_scans)]
    guesses = np.stack([S.perturbed_guess(poses[k], 3 + k) for k in range(n_scans)]).astype(np.float32)
    g.map_set(mc, ms)
    ref = []
    for k in range(n_scans):
        g.scan_upload(scans[k]); g.scan_organize(); g.scan_extract(); g.scan_downsample()
        g.scan_match_async(guesses[k], 0)
        ref.append(g.get_pose_record())
    g.close()
    hdir = os.path.dirname(pkg.HIP_LIB_PATH)
    exe = tmp_path / "replay_multi"
    src = os.path.join(pkg.PKG_DIR, "host", "replay_multi.cpp")
    r = subprocess.run(["hipcc", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", str(exe), src, "-L" + hdir, "-llvi_hip", "-lrccl",
                        "-Wl,-rpath," + hdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    np.concatenate(scans).tofile(tmp_path / "scans.bin"); mc.tofile(tmp_path / "mc.bin"); ms.tofile(tmp_path / "ms.bin"); guesses.tofile(tmp_path / "g.bin")
    r = subprocess.run([str(exe), "8", "4096", str(tmp_path / "scans.bin"), str(n_scans), str(n_raw), str(tmp_path / "mc.bin"), str(len(mc)),
                        str(tmp_path / "ms.bin"), str(len(ms)), str(tmp_path / "g.bin"), "10"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stderr[-2000:], r.stdout[-500:])
    lines = r.stdout.strip().splitlines()
    assert any(ln.startswith("backend hip") and " gpus 1 " in ln for ln in lines), lines[:4]        # (RCCL prints its version banner first)
    recs = [ln.split() for ln in lines if ln.startswith("rec ")]
    assert len(recs) == n_scans
    for k, f in enumerate(recs):
        assert int(f[1]) == k and int(f[2]) == ref[k]["status"] and int(f[3]) == ref[k]["iters"]
        pose = np.array([float(v) for v in f[4:10]], np.float32)
        np.testing.assert_array_equal(pose.view(np.uint32), ref[k]["pose"].view(np.uint32))
