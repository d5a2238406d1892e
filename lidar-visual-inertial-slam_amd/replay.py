"""Batched replay of independent scans against a frozen local map (BASELINE config 5).

In production the path is sequential from scan to scan (mapOptimization.cpp:809,1594-1609); only a
replay against a frozen map shards.  Scan i goes to rank i mod world; every rank holds a replica of
the map; there is no exchange step inside the algorithm, only a gather of the 32-byte pose records
(`lvi_pose_record`: 6 x f32 pose, i32 status, i32 iterations) after each step — an RCCL
all_gather on GPUs ("nccl" backend), gloo in the CPU tests.
"""
import numpy as np

RECORD_FLOATS = 8


def shard(n_items, rank, world):
    """indices of the items this rank owns: i mod world == rank"""
    return list(range(rank, n_items, world))


def owner(i, world):
    return i % world


def pack_record(pose, status, iters):
    r = np.zeros(RECORD_FLOATS, np.float32)
    r[:6] = pose
    r[6:8] = np.array([status, iters], np.int32).view(np.float32)
    return r


def unpack_records(rec):
    rec = np.ascontiguousarray(rec, np.float32).reshape(-1, RECORD_FLOATS)
    return dict(pose=rec[:, :6].copy(), status=rec[:, 6].copy().view(np.int32), iters=rec[:, 7].copy().view(np.int32))


def gather_records(local, world, dist=None, force=False):
    """local: torch tensor (k, 8) of this rank's records for one step → (world, k, 8) on every rank
    (force: issue the collective also for a world of one rank — the RCCL rehearsal of tests/test_gpu_rccl.py)"""
    import torch
    if world == 1 and not force:
        return local.unsqueeze(0)
    k = local.shape[0]
    out = torch.empty((world * k,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)   # concatenated along dim 0
    dist.all_gather_into_tensor(out, local.contiguous())
    return out.view((world, k) + tuple(local.shape[1:]))


def replay_scans(handle, scans, guesses, rank=0, world=1, dist=None, device="cpu"):
    """Process this rank's share of `scans` on `handle` (map already set), gathering the pose records
    after every step.  Returns an (n_scans, 8) float32 array in scan order on every rank.
    Host-buffer path (used with the CPU oracle in the gloo tests and in the smoke test)."""
    import torch
    n = len(scans)
    steps = (n + world - 1) // world
    out = np.zeros((n, RECORD_FLOATS), np.float32)
    for s in range(steps):
        i = s * world + rank
        rec = pack_record(np.zeros(6), -999, 0)              # padding record of a rank with no scan in the last step
        if i < n:
            handle.scan_upload(scans[i]); handle.scan_organize(); handle.scan_extract(); handle.scan_downsample()
            r = handle.scan_match(guesses[i])
            rec = pack_record(r["pose"], r["status"], r["iters"])
        g = gather_records(torch.from_numpy(rec[None]).to(device), world, dist).cpu().numpy()
        for rk in range(world):
            j = s * world + rk
            if j < n:
                out[j] = g[rk, 0]
    return out


class RollingReplay:
    """The replay loop bench.py times, factored out so that the CPU tier can run it (gloo, oracle-backed handles).

    B handles per rank, each fed `depth` steps ahead: before handle b receives its scans of step i the host waits for
    the mark b recorded after step i - depth (lvi_lidar_wait_mark), so no queue runs dry while the host prepares the
    neighbours' scans, and never more than `depth` steps are in flight per handle.  After all handles got step i, the
    records of step i - depth are final on every handle and are gathered across ranks (`gather(j)`: RCCL / gloo
    all_gather of the step's 32-byte pose records; a no-op for one rank).  `issue(i, b, handle)` enqueues whatever a
    step means for one handle — one scan, or a batch of S scans in one launch sequence — without synchronising.
    depth 0 = synchronise every handle after every step (the non-rolling form).
    """

    def __init__(self, handles, issue, gather=None, depth=2):
        self.handles, self.issue, self.gather = list(handles), issue, gather
        self.depth = max(0, min(int(depth), 8))             # LVI_LIDAR_MARKS
        self.pending = []                                   # steps whose records have not been gathered yet
        self.gathered = []                                  # steps gathered, in order (tests read this)

    def step(self, i):
        for b, h in enumerate(self.handles):
            if self.depth:
                h.wait_mark(i % self.depth)
            self.issue(i, b, h)
            if self.depth:
                h.mark(i % self.depth)
        if not self.depth:
            for h in self.handles:
                h.sync()
            self._gather(i)
            return
        self.pending.append(i)
        if len(self.pending) > self.depth:
            self._gather(self.pending.pop(0))

    def flush(self):
        """finish everything in flight and gather what is left (inside the timed region)"""
        for h in self.handles:
            h.sync()
        while self.pending:
            self._gather(self.pending.pop(0))

    def _gather(self, j):
        self.gathered.append(j)
        if self.gather is not None:
            self.gather(j)


def scan_index(step, handle, slot, n_handles, batch, pool):
    """which scan of a rank's resident pool a (step, handle, batch slot) processes: consecutive scans, cycled"""
    return ((step * n_handles + handle) * batch + slot) % pool


def shard_frames(n_pairs, rank, world):
    """tracker leg: frame pair i -> rank i mod world"""
    return shard(n_pairs, rank, world)
