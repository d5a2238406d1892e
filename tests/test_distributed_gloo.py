"""CPU tier: the N>1 path (shard scans one-per-rank, gather pose records) with world_size 2 over gloo.
The ranks run the CPU oracle in place of the HIP library — what is under test here is the sharding
and the collective, which are the same code bench.py runs over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import __graft_entry__ as graft
from oracle import loader
pkg = graft.import_package()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = loader.load(pkg)
S = pkg.synth
P = dict(N_SCAN=4, Horizon_SCAN=2048, max_raw_points=8192, max_map_points=65536, icp_max_iters=10, icp_disable_break=1)
h = pkg.LidarHotpath(lib, **P)
# rank 0 owns the frozen map and broadcasts it (bench.py does the same over RCCL)
hdr = torch.zeros(2, dtype=torch.int64)
if rank == 0:
    mc, ms = S.make_map(h, 5, 5001, seed=17)
    hdr[0], hdr[1] = len(mc), len(ms)
dist.broadcast(hdr, 0)
tc = torch.zeros((int(hdr[0]), 4)); ts = torch.zeros((int(hdr[1]), 4))
if rank == 0:
    tc.copy_(torch.from_numpy(pkg._abi.pts_xyzi(mc))); ts.copy_(torch.from_numpy(pkg._abi.pts_xyzi(ms)))
dist.broadcast(tc, 0); dist.broadcast(ts, 0)
h.map_set(tc.numpy(), ts.numpy())
n = 5                                           # odd on purpose: the last step has an idle rank
poses = [S.loop_pose(0.3 + 0.9 * i, 0.01, -0.01) for i in range(n)]
scans = [S.make_scan(5001, poses[i], 100 + i) for i in range(n)]
guesses = [S.perturbed_guess(poses[i], i) for i in range(n)]
rec = pkg.replay.replay_scans(h, scans, guesses, rank, world, dist)
np.save(os.path.join(%(out)r, "rec_rank%%d.npy" %% rank), rec)
if rank == 0:
    ref = pkg.replay.replay_scans(h, scans, guesses, 0, 1, None)      # the same work unsharded
    np.save(os.path.join(%(out)r, "ref.npy"), ref)
    np.save(os.path.join(%(out)r, "truth.npy"), np.array(poses))
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_shard_helpers(pkg):
    R = pkg.replay
    assert R.shard(10, 1, 4) == [1, 5, 9] and R.shard(3, 3, 4) == []
    assert sorted(sum((R.shard(11, r, 3) for r in range(3)), [])) == list(range(11))
    rec = R.pack_record([1, 2, 3, 4, 5, 6], 2, 17)
    u = R.unpack_records(rec)
    assert u["status"][0] == 2 and u["iters"][0] == 17 and list(u["pose"][0]) == [1, 2, 3, 4, 5, 6]
    assert rec.nbytes == 32                                               # lvi_pose_record


def test_two_ranks_gloo(tmp_path):
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    r0, r1, ref = (np.load(tmp_path / f) for f in ("rec_rank0.npy", "rec_rank1.npy", "ref.npy"))
    np.testing.assert_array_equal(r0, r1)                                 # every rank holds the full gathered result
    np.testing.assert_array_equal(r0, ref)                                # sharding does not change any result
    truth = np.load(tmp_path / "truth.npy")
    u = __import__("numpy").asarray(r0)
    assert (u[:, 6].view(np.int32) == 0).all() and (u[:, 7].view(np.int32) == 10).all()
    assert np.abs(u[:, 3:6] - truth[:, 3:6]).max() < 0.06


ROLLING_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import __graft_entry__ as graft
from oracle import loader
pkg = graft.import_package()
R, S = pkg.replay, pkg.synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = loader.load(pkg)
B, NB, POOL, STEPS, DEPTH = 2, 2, 3, 5, 2                 # handles per rank, scans per launch sequence, odd step count, queue depth 2
P = dict(N_SCAN=4, Horizon_SCAN=2048, max_raw_points=8192, max_map_points=65536, icp_max_iters=6, icp_disable_break=1, batch_scans=NB)
hs = [pkg.LidarHotpath(lib, **P) for _ in range(B)]
hdr = torch.zeros(2, dtype=torch.int64)
if rank == 0:
    mc, ms = S.make_map(hs[0], 5, 5001, seed=17)
    hdr[0], hdr[1] = len(mc), len(ms)
dist.broadcast(hdr, 0)
tc = torch.zeros((int(hdr[0]), 4)); ts = torch.zeros((int(hdr[1]), 4))
if rank == 0:
    tc.copy_(torch.from_numpy(pkg._abi.pts_xyzi(mc))); ts.copy_(torch.from_numpy(pkg._abi.pts_xyzi(ms)))
dist.broadcast(tc, 0); dist.broadcast(ts, 0)
for h in hs:
    h.map_upload(tc.numpy(), ts.numpy()); h.map_build()
poses = [S.loop_pose(0.3 + 0.9 * (rank * POOL + k), 0.01, -0.01) for k in range(POOL)]
scans = [S.make_scan(5001, poses[k], 100 + rank * POOL + k) for k in range(POOL)]
guesses = [S.perturbed_guess(poses[k], rank * POOL + k) for k in range(POOL)]
per_step = B * NB
rec = torch.zeros((STEPS * per_step, 8), dtype=torch.float32)            # this rank's records ("device" memory of the CPU library)
gathered = {}

def issue(i, b, h):
    ks = [R.scan_index(i, b, z, B, NB, POOL) for z in range(NB)]
    h.batch_upload([scans[k] for k in ks])
    base = (i * B + b) * NB
    h.batch_run(np.stack([guesses[k] for k in ks]), rec[base].data_ptr(), rebuild_map=True)

def gather(j):
    gathered[j] = R.gather_records(rec[j * per_step:(j + 1) * per_step], world, dist).clone()

roll = R.RollingReplay(hs, issue, gather, depth=DEPTH)
for i in range(STEPS):
    roll.step(i)
    assert len(roll.pending) <= DEPTH
roll.flush()
assert roll.gathered == list(range(STEPS)), roll.gathered
out = %(out)r
np.save(os.path.join(out, "roll_rec_rank%%d.npy" %% rank), rec.numpy())
np.save(os.path.join(out, "roll_gather_rank%%d.npy" %% rank), torch.stack([gathered[j] for j in range(STEPS)]).numpy())
# the same scans one by one on a plain handle: what every record must equal
P1 = dict(P); P1["batch_scans"] = 1
h1 = pkg.LidarHotpath(lib, **P1)
h1.map_upload(tc.numpy(), ts.numpy())
ref = np.zeros((POOL, 8), np.float32)
for k in range(POOL):
    h1.map_build(); h1.scan_upload(scans[k]); h1.scan_organize(); h1.scan_extract(); h1.scan_downsample()
    r = h1.scan_match(guesses[k])
    ref[k] = R.pack_record(r["pose"], r["status"], r["iters"])
np.save(os.path.join(out, "roll_ref_rank%%d.npy" %% rank), ref)
dist.barrier()
dist.destroy_process_group()
"""


def test_rolling_batched_replay_two_ranks_gloo(pkg, tmp_path):
    """the loop bench.py times (replay.RollingReplay: B handles, S scans per launch sequence, queue depth 2, odd step count,
    all_gather of every step's records) under gloo with two ranks and the CPU library behind the same handle interface"""
    port = _free_port()
    script = tmp_path / "roll_worker.py"
    script.write_text(ROLLING_WORKER % dict(root=ROOT, out=str(tmp_path)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    R = pkg.replay
    B, NB, POOL, STEPS = 2, 2, 3, 5
    recs = [np.load(tmp_path / f"roll_rec_rank{r}.npy") for r in range(2)]
    refs = [np.load(tmp_path / f"roll_ref_rank{r}.npy") for r in range(2)]
    gath = [np.load(tmp_path / f"roll_gather_rank{r}.npy") for r in range(2)]
    np.testing.assert_array_equal(gath[0], gath[1])                      # every rank holds every step's records of every rank
    assert gath[0].shape == (STEPS, 2, B * NB, 8)
    for r in range(2):
        for i in range(STEPS):
            for b in range(B):
                for z in range(NB):
                    k = R.scan_index(i, b, z, B, NB, POOL)
                    row = recs[r][(i * B + b) * NB + z]
                    np.testing.assert_array_equal(row, refs[r][k])       # batched, rolling, sharded: the bits of the plain path
                    np.testing.assert_array_equal(gath[0][i, r, b * NB + z], row)
        assert (recs[r][:, 6].view(np.int32) == 0).all() and (recs[r][:, 7].view(np.int32) == 6).all()


def test_rolling_replay_bookkeeping(pkg):
    """depth 0 / 1 / 3 and the frame sharding of the tracker leg, without any library"""
    R = pkg.replay

    class Fake:
        def __init__(self):
            self.log = []

        def wait_mark(self, s): self.log.append(("wait", s))
        def mark(self, s): self.log.append(("mark", s))
        def sync(self): self.log.append(("sync",))

    for depth in (0, 1, 3):
        hs = [Fake(), Fake()]
        issued, gathered = [], []
        roll = R.RollingReplay(hs, lambda i, b, h: issued.append((i, b)), gathered.append, depth=depth)
        for i in range(7):
            roll.step(i)
            assert len(roll.pending) <= max(depth, 0)
            # a step's records are gathered only after every handle waited for that step's mark (or synchronised)
            assert gathered == list(range(max(0, i + 1 - depth)))
        roll.flush()
        assert gathered == list(range(7)) and issued == [(i, b) for i in range(7) for b in range(2)]
        if depth:
            assert hs[0].log[:3] == [("wait", 0), ("mark", 0), ("wait", 1 % depth)]
    assert R.shard_frames(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((R.shard_frames(40, r, 8) for r in range(8)), [])) == list(range(40))
    assert [R.scan_index(1, 1, z, 2, 4, 8) for z in range(4)] == [4, 5, 6, 7]


BENCH_WORKER = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import __graft_entry__ as graft
from oracle import loader
pkg = graft.import_package()
ora = loader.load(pkg)
pkg.load_hip = lambda: ora                      # the TEST puts the CPU oracle where bench.py loads the HIP library (bench.py itself never touches oracle/
                                                # outside its cpu_baseline leg); what runs is bench.py's own main(): broadcast, windows, gathers, tracker gather
os.environ["LVI_BENCH_DEVICE"] = "cpu"
import bench
sys.argv = ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2", "--prime-steps", "2", "--profile-steps", "1",
            "--n-raw", "5001", "--keyframes", "5", "--kf-n-raw", "5001", "--map-points", "20000", "--pool", "3", "--inflight", "2", "--batch", "2",
            "--icp-iters", "6", "--no-cpu", "--tracker-seconds", "0.05", "--sequential-scans", "0", "--cached-plan-steps", "2"]
bench.main()
"""


def test_bench_main_two_ranks_gloo(tmp_path):
    """bench.py's OWN main() for world = 2 — map broadcast, priming by step count, timed windows with the
    barrier fences, per-step gather of the pose records (odd step count), tracker-rate gather, the cached-plan secondary figure —
    over gloo with oracle-backed handles.  The RCCL form of the same collectives has never executed anywhere."""
    import json
    port = _free_port()
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER % dict(root=ROOT))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(tmp_path)))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-3000:] for o in outs)
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]      # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["metric"] == "scans_per_sec_100k_mid360" and d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["scans_per_step"] == 2 * 2 * 2 and d["config"]["map_plan"] == "per-rebuild"
    assert d["value"] > 0 and len(d["value_windows"]["scans_per_sec"]["all"]) == 2
    assert d["results_ok"] is True, d["pose_err_vs_truth"]
    assert len(d["tracker"]["per_rank"]) == 2 and d["tracker"]["value"] > 0
    assert d["value_cached_plan"]["scans_per_sec"] > 0
