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
