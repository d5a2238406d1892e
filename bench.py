#!/usr/bin/env python3
"""bench.py — scans/sec of the lidar scan-matching hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of 100k-point synthetic MID360 scans per rank,
reference-faithful per scan: organise → feature extraction (incl. per-ring VoxelGrid) → scan VoxelGrid →
re-voxelisation of the ≈5M-point raw local map + KNN index build (the reference rebuilds both
for every scan: mapOptimization.cpp:958-965, 1322-1323) → 10 fixed Gauss-Newton iterations
(convergence break disabled on GPU and CPU alike, SURVEY §8 d).  Inputs (raw scans, raw map)
are resident in HBM before the timed region; the only host traffic inside it is the launch
stream and the event waits of the rolling replay.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step processes --inflight handles x --batch scans per rank: the scans of one handle go through ONE launch
sequence (lvi_scan_batch_*, scan index in blockIdx.z).  Independent scans shard across ranks ("weak" scaling); the
frozen raw map is broadcast once over RCCL and the 32-byte pose records are all-gathered over RCCL per step.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
ROUND = "r03"                # profiles/<ROUND>_pmc_traffic.json carries this round's HBM counter traffic


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed windows of --steps steps each; `value` is the FIRST window (exactly --steps steps between fences), "
                         "the others give value_windows (median / min / max)")
    ap.add_argument("--n-raw", type=int, default=100001)
    ap.add_argument("--keyframes", type=int, default=250)
    ap.add_argument("--kf-n-raw", type=int, default=30001)
    ap.add_argument("--map-points", type=int, default=5_000_000)
    ap.add_argument("--frozen-map", action="store_true", help="reuse the DS map/index across scans (not the headline)")
    ap.add_argument("--map-source", choices=["resident", "assemble"], default="resident",
                    help="resident: the raw local map sits in HBM as one cloud (headline); assemble: it is fused per step from the "
                         "device-resident keyframe store (SURVEY f-4: what a node does instead of uploading 78 MB per scan)")
    ap.add_argument("--map-plan", choices=["per-rebuild", "cached"], default="per-rebuild",
                    help="per-rebuild (headline): every re-voxelisation of the raw map takes the bounding box (getMinMax3D) and the per-bin counts "
                         "again, per scan, as the reference's VoxelGrid::filter does; cached: once per map upload (lvi_lidar_params.map_plan_cache = 1)")
    ap.add_argument("--cached-plan-steps", type=int, default=40, help="secondary figure: steps timed once more with map_plan_cache = 1 (0 = skip)")
    ap.add_argument("--icp-iters", type=int, default=10)
    ap.add_argument("--pool", type=int, default=8, help="distinct scans per rank, cycled")
    ap.add_argument("--queue-depth", type=int, default=2,
                    help="steps kept enqueued per handle in the rolling form (0 = sync every handle after every step)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="independent handles per GPU (own streams, own DS map / index replicas); with --batch S each carries S scans per step")
    ap.add_argument("--batch", type=int, default=8,
                    help="scans per launch sequence (lvi_scan_batch_*): every kernel of the path carries the scan index in blockIdx.z. "
                         "1 = the single-scan entry points")
    ap.add_argument("--enqueue", choices=["graph", "eager"], default="eager",
                    help="--batch 1 only: one hipGraph launch per scan instead of eager launches")
    ap.add_argument("--map-stream", type=int, default=-1, help="lvi_lidar_params.map_on_main_stream: -1 auto (1 when >= 4 scans in flight), 0, 1")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-share-map", action="store_true", help="every handle keeps its own copy of the raw map (default: one copy per GPU, lvi_map_share)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-tracker", action="store_true")
    ap.add_argument("--tracker-seconds", type=float, default=0.6)
    ap.add_argument("--prime-steps", type=int, default=80,
                    help="untimed steps before the W warm-up steps, part of the set-up: measured on this runtime, one ~45 ms stall (runtime "
                         "pool growth) lands somewhere in the first ~0.2 s of back-to-back batched launches of a fresh process.  A COUNT, "
                         "not a duration: every rank must issue the same number of per-step collectives")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--sequential-scans", type=int, default=48,
                    help="secondary figure (rank 0): a sequential replay — raw stream -> pose -> keyframe -> next scan — through the C++ node "
                         "code (host/lvi_host.hpp) with the incremental local map and with the full per-scan assembly; 0 = skip")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    # LVI_BENCH_DEVICE=cpu: the CPU tier's rehearsal of THIS control flow (tests/test_distributed_gloo.py: two gloo ranks, the
    # library handle replaced by the test): tensors on the host, no CUDA calls.  Never set by the driver.
    on_gpu = os.environ.get("LVI_BENCH_DEVICE", "cuda") != "cpu"
    if on_gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda unavailable)")
    # rehearsal switches for a ONE-GPU box (the N > 1 code path with gloo, every rank on cuda:0); never set by the driver
    rehearse = os.environ.get("LVI_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    if on_gpu:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")

    def dev_sync():
        if on_gpu:
            torch.cuda.synchronize()
    # LVI_BENCH_RCCL_WORLD1=1 (tests/test_gpu_rccl.py, a one-GPU box): the N > 1 control flow with a world of ONE rank over the nccl
    # backend — every collective of this file (broadcast of the map, all_gather of the pose records per step, MAX all_reduce of the
    # window time, the tracker-rate gather, the barriers) executes through RCCL, trivially; never set by the driver
    use_dist = world > 1 or (os.environ.get("LVI_BENCH_RCCL_WORLD1") == "1" and on_gpu)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if rehearse or not on_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)          # RCCL
    pkg = graft.import_package()
    A, S, R = pkg._abi, pkg.synth, pkg.replay
    hip = pkg.load_hip()                      # raises when the HIP library is missing: no fallback

    B = max(1, args.inflight)
    NB = max(1, min(args.batch, 8))
    P = dict(N_SCAN=4, Horizon_SCAN=32768, max_raw_points=131072, max_map_points=args.map_points + 65536,
             icp_max_iters=args.icp_iters, icp_disable_break=1, batch_scans=NB, map_plan_cache=(1 if args.map_plan == "cached" else 0),
             map_on_main_stream=(1 if B * NB >= 4 else 0) if args.map_stream < 0 else args.map_stream,   # one stream per hardware queue once >= 4 scans are in flight
             max_keyframes=(args.keyframes + 8) if args.map_source == "assemble" else 0,
             max_keyframe_points=(args.map_points + 200000) if args.map_source == "assemble" else 0)
    dev_index = local_rank if on_gpu else 0
    hs = [pkg.LidarHotpath(hip, device=dev_index, **P) for _ in range(B)]
    g = hs[0]

    # ---------------------------------------------------------------- frozen local map: rank 0 builds, RCCL broadcasts
    t_setup = time.time()
    hdr = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == 0:
        mc, ms = S.make_map(g, args.keyframes, args.kf_n_raw, seed=4711, torch_device=dev, target_surf=args.map_points)
        hdr[0], hdr[1] = len(mc), len(ms)
    if use_dist:
        dist.broadcast(hdr, 0)
    nc, ns = int(hdr[0]), int(hdr[1])
    d_mc = torch.empty((nc, 4), dtype=torch.float32, device=dev)
    d_ms = torch.empty((ns, 4), dtype=torch.float32, device=dev)
    if rank == 0:
        d_mc.copy_(torch.from_numpy(A.pts_xyzi(mc)))
        d_ms.copy_(torch.from_numpy(A.pts_xyzi(ms)))
    if use_dist:
        dist.broadcast(d_mc, 0)
        dist.broadcast(d_ms, 0)
    dev_sync()
    keys = None
    if args.map_source == "assemble":
        # SURVEY f-4: the local map is not handed over as one raw cloud but fused on the device, per step, from the
        # keyframe store (here: the same points cut into --keyframes pieces, stored with the identity pose, so the
        # fused map — and therefore every result — is bit-identical to the resident-map run)
        hc, hsurf = d_mc.cpu().numpy(), d_ms.cpu().numpy()
        K = max(1, args.keyframes)
        cb = np.linspace(0, nc, K + 1).astype(np.int64); sb = np.linspace(0, ns, K + 1).astype(np.int64)
        for h in hs:
            for k in range(K):
                h.keyframe_add(hc[cb[k]:cb[k + 1]], hsurf[sb[k]:sb[k + 1]], np.zeros(6, np.float32))
        keys = np.arange(K, dtype=np.int32)
        for h in hs:
            h.map_assemble(keys)
            h.sync()
    else:
        # ONE copy of the frozen raw map per GPU: handle 0 holds it, the others read it in place (lvi_map_share) — the
        # map is a read-only cloud shared by every scan matched against it, and one copy stays in the memory-side cache
        for b, h in enumerate(hs):
            if b == 0 or args.no_share_map:
                h.map_upload_device(d_mc.data_ptr(), nc, d_ms.data_ptr(), ns)
            else:
                h.map_share(hs[0])
            h.map_build()
            h.sync()
    cnt_map = g.counts()

    # ---------------------------------------------------------------- measured copy ceiling of this GPU (report % of the vendor
    # figure AND of a measured device copy): 256 MB float4 tensor copied device-to-device, bytes read + written / time
    copy_gbs = None
    if rank == 0 and on_gpu:
        src = torch.empty((16 * 1024 * 1024, 4), dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize()
        copy_gbs = round(10 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del src, dst

    # ---------------------------------------------------------------- scan pool of this rank, resident in HBM
    poses, guesses, scans_host, d_scans = [], [], [], []
    for k in range(args.pool):
        sid = rank * args.pool + k
        pose = S.loop_pose(0.37 + 0.71 * sid, 0.01 * np.sin(sid), -0.02 * np.cos(sid))
        sc = S.make_scan(args.n_raw, pose, 12345 + sid, torch_device=dev)
        poses.append(pose); guesses.append(S.perturbed_guess(pose, sid)); scans_host.append(sc)
        d_scans.append(torch.from_numpy(sc.view(np.uint8).reshape(-1, 20).copy()).to(dev))
    n_windows = max(1, args.repeats)
    n_prime = max(0, args.prime_steps)                                # priming steps (records are written, never read)
    total = n_prime + args.warmup + n_windows * args.steps + args.profile_steps
    per_step = B * NB                                                 # scans per step and rank
    d_rec = torch.zeros((total * per_step, 8), dtype=torch.float32, device=dev)
    dev_sync()
    setup_s = time.time() - t_setup

    enq = [0.0]
    rebuild = not args.frozen_map

    def issue(i, b, h):
        """enqueue one handle's share of step i: NB consecutive scans of the pool, no synchronisation"""
        t_e = time.perf_counter()
        base = (i * B + b) * NB
        ks = [R.scan_index(i, b, z, B, NB, args.pool) for z in range(NB)]
        if NB == 1 and args.enqueue == "graph":
            # one C-ABI call per scan: D2D of the 2 MB scan + one hipGraph launch of the whole path
            h.scan_replay_enqueue(d_scans[ks[0]].data_ptr(), args.n_raw, guesses[ks[0]], d_rec[base].data_ptr(), rebuild_map=rebuild)
        elif NB == 1:
            if rebuild:
                if keys is not None:
                    h.map_assemble(keys)                              # fuse the keyframes, then the build (f-4)
                else:
                    h.map_build()
            h.scan_upload_device(d_scans[ks[0]].data_ptr(), args.n_raw)   # D2D, 2 MB
            h.scan_organize(); h.scan_extract(); h.scan_downsample()
            h.scan_match_async(guesses[ks[0]], d_rec[base].data_ptr())
        else:
            if rebuild and keys is not None:
                h.map_assemble(keys)                                  # fuses the raw map once per step and rebuilds every slot's DS map + index
            h.batch_bind_device([d_scans[k].data_ptr() for k in ks], [args.n_raw] * NB)      # scans are read in place
            h.batch_run(np.stack([guesses[k] for k in ks]), d_rec[base].data_ptr(), rebuild_map=rebuild and keys is None)
        enq[0] += time.perf_counter() - t_e

    def gather(j):
        if use_dist:
            R.gather_records(d_rec[j * per_step:(j + 1) * per_step], world, dist, force=True)   # RCCL all_gather: 32 B pose record per scan

    roll = R.RollingReplay(hs, issue, gather, depth=args.queue_depth)

    def fence():
        dev_sync()
        if use_dist:
            dist.barrier()
        dev_sync()

    step0 = 0
    if n_prime:
        for step0 in range(n_prime):
            roll.step(step0)
        step0 = n_prime
        roll.flush()
    for i in range(step0, step0 + args.warmup):
        roll.step(i)
    roll.flush()
    fence()
    win = []
    enq_first = 0.0
    for w in range(n_windows):
        enq[0] = 0.0
        t0 = time.perf_counter()
        first = step0 + args.warmup + w * args.steps
        for i in range(first, first + args.steps):
            roll.step(i)
        roll.flush()
        fence()
        el = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax[0])
        win.append(el)
        if w == 0:
            enq_first = enq[0]
    elapsed = win[0]
    enqueue_ms_per_scan = 1e3 * enq_first / (args.steps * per_step)
    rates = [world * per_step * args.steps / e for e in win]

    # ---------------------------------------------------------------- sanity of what was timed (every record of every window, vs ground truth)
    lo, hi = (step0 + args.warmup) * per_step, (step0 + args.warmup + n_windows * args.steps) * per_step
    rec = d_rec[lo:hi].cpu().numpy()
    status = rec[:, 6].copy().view(np.int32)
    iters = rec[:, 7].copy().view(np.int32)
    pool_idx = np.array([j % args.pool for j in range(lo, hi)])
    gt = np.array([poses[k] for k in pool_idx])
    err_t = float(np.abs(rec[:, 3:6] - gt[:, 3:6]).max())
    err_r = float(np.abs(rec[:, 0:3] - gt[:, 0:3]).max())
    ok = bool((status == 0).all() and (iters == args.icp_iters).all() and err_t < 0.05 and err_r < 0.01)
    # the same scan gives the same bits whichever handle, slot or step processed it
    for k in range(args.pool):
        rows = rec[pool_idx == k]
        if len(rows):
            ok = ok and bool((rows.view(np.uint32) == rows[0].view(np.uint32)).all())

    # ---------------------------------------------------------------- tracker leg (secondary metric: LK frames/sec), frame pairs sharded over the ranks
    # (runs before the profiled pass: measured on this runtime, once a pass with HIP timing events has run in the process,
    # every later 0.9 MB frame upload of the tracker takes ~1 ms instead of ~40 us)
    tracker_out = None
    if not args.no_tracker:
        try:
            tracker_out = bench_tracker(pkg, hip, dev_index, rank, world, args.tracker_seconds)
        except Exception as e:                      # noqa: BLE001 — the tracker leg must not hide the headline
            tracker_out = dict(error=str(e), value=0.0)
        if use_dist:                                # outside the try: every rank issues this collective, whatever its leg did
            tr = torch.tensor([float(tracker_out.get("value", 0.0))], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(tr) for _ in range(world)]
            dist.all_gather(allr, tr)
            tracker_out["per_rank"] = [round(float(x[0]), 1) for x in allr]
            tracker_out["value"] = round(float(sum(tracker_out["per_rank"])), 1)

    # ---------------------------------------------------------------- sequential mode (secondary figure, rank 0)
    seq_out = None
    if args.sequential_scans > 0 and rank == 0:
        try:
            seq_out = bench_sequential(pkg, hip, local_rank if on_gpu else 0, dev, args.sequential_scans, args.n_raw, args.keyframes, args.kf_n_raw)
        except Exception as e:                      # noqa: BLE001
            seq_out = dict(error=str(e))

    # ---------------------------------------------------------------- per-kernel timing with HIP events (same workload, same process)
    # Two passes, handle 0 records events around every launch on its stream:
    #   (1) ONE handle alone, the others idle: a kernel's own duration — what `rocprofv3 --kernel-trace` of `--inflight 1` shows
    #       (profiles/<round>_rocprof_steady_1x<batch>.md); the roofline is quoted on this;
    #   (2) as in the timed pass, the other handles launching beside it: durations then include what runs beside the kernel
    #       (reported next to (1) as `under_load`).
    stats, stats_load, kern_ms = [], [], 0.0
    if args.profile_steps > 0:
        g.prof_reset(); g.prof_enable(True)
        for i in range(step0 + args.warmup + n_windows * args.steps, step0 + args.warmup + n_windows * args.steps + args.profile_steps):
            roll.step(i)
        roll.flush()
        stats_load = g.prof_read()
        g.prof_reset()
        for k in range(args.profile_steps):
            issue(k, 0, g)                              # (step k's record slots: gathered long ago)
            g.sync()
        stats = g.prof_read()
        g.prof_enable(False)
    cnt = g.counts()
    Q = cnt["corner_ds"] + cnt["surf_ds"]
    for st_ in (stats, stats_load):
        for s in st_:
            s["avg_us"] = 1e3 * s["total_ms"] / max(s["launches"], 1)
            if s["name"] == "icp_gn":
                # the library books a nominal 128 * 0.25 * n_raw per scan; the real query count is known here (read back above)
                s["bytes_alg"] = 128.0 * Q * NB * s["launches"]
            s["gbs"] = round((s["bytes_alg"] / s["launches"]) / (s["avg_us"] * 1e-6) / 1e9, 1) if s["bytes_alg"] > 0 else None
        st_.sort(key=lambda s: -s["total_ms"])
    load_by_name = {s["name"]: s for s in stats_load}
    kern_ms = sum(s["total_ms"] for s in stats) / max(args.profile_steps, 1)
    traffic_tab, traffic_src = {}, None
    tpath = os.path.join(ROOT, "profiles", ROUND + "_pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic_tab = json.load(open(tpath)); traffic_src = os.path.relpath(tpath, ROOT)
        except Exception:
            traffic_tab = {}

    def roof(s, note):
        b = s["bytes_alg"] / s["launches"]
        gbs = b / (s["avg_us"] * 1e-6) / 1e9
        base = s["name"].split("/")[0] + "_kernel"
        # keyed "kernel [grid G, wg W]" (separate rocprofv3 --pmc passes of this same command, tools/summarize_prof.py pmc);
        # the profiled tags with bytes are the largest geometry of their kernel
        bases = [base] + (["vb_scatter_det_kernel"] if s["name"] == "vb_scatter/map" else [])      # (the map's partition is the deterministic form)
        cand = [(v["hbm_bytes_per_launch"], v.get("launches_per_step", 1.0)) for k, v in traffic_tab.items() if any(k.startswith(b + " ") for b in bases)]
        if s["name"].endswith("/map") and cand:
            cand = [max(cand)]                           # the map's geometry is the largest of the kernel's (ring / scan grids share the kernel)
        # the PMC passes run ONE scan per launch: a launch of this pass carries NB scans.  Several geometries of one kernel (icp_gn: the
        # 64- and the 256-feature form) are weighted by their launches per step
        tr = NB * sum(b * w for b, w in cand) / sum(w for _, w in cand) if cand else None
        ld = load_by_name.get(s["name"])
        under_load = None
        if ld and ld["launches"]:
            lg = (ld["bytes_alg"] / ld["launches"]) / (ld["avg_us"] * 1e-6) / 1e9
            under_load = dict(avg_launch_us=round(ld["avg_us"], 2), achieved=round(lg, 1), frac=round(lg / HBM_PEAK_GBS, 4),
                              note="the same launch with the other handles of the headline configuration running beside it (events bracket the launch on its "
                                   "stream: the duration includes what shares the machine with it)")
        return dict(bound="hbm", kernel=s["name"], achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                    under_load=under_load,
                    measured_copy_gbs=copy_gbs, frac_of_measured_copy=(round(gbs / copy_gbs, 4) if copy_gbs else None),
                    traffic=(round(tr) if tr is not None else None), traffic_over_algorithmic=(round(tr / b, 2) if tr else None), traffic_source=(traffic_src if tr is not None else "no PMC pass of this round's build committed yet: null, not a stale figure"),
                    bytes_alg_per_launch=b, scans_per_launch=NB, avg_launch_us=round(s["avg_us"], 2),
                    launches_per_step=s["launches"] / max(args.profile_steps, 1), note=note)

    roofline = None
    roofline_bw = None
    with_bytes = [s for s in stats if s["bytes_alg"] > 0]
    if with_bytes:
        # dominant kernel = largest total time per step among the kernels of the path
        roofline = roof(with_bytes[0], "HIP events on the launch stream, profiled pass of the same workload right after the timed windows with ONE handle "
                                       "(a launch sequence of %d scans) running alone, as rocprofv3 --kernel-trace of --inflight 1 shows it; "
                                       "the kernel with the largest total time per step.  " % NB +
                                       "icp_gn (one Gauss-Newton iteration: end of the previous iteration + searches + fits + rows) is a chain of dependent "
                                       "gathers on L2s that every launch finds cold plus ~2 000 instructions of small-matrix code per wavefront (clock64 phase "
                                       "stamps, DESIGN.md 5): neither HBM- nor VALU-bound, its HBM fraction is small by construction; bytes = 128 B x the real "
                                       "query count read back from the device")
        stream = max(with_bytes, key=lambda s: s["bytes_alg"] / s["launches"] if s["avg_us"] > 0 else 0)
        big = [s for s in with_bytes if s["bytes_alg"] / s["launches"] >= 0.5 * stream["bytes_alg"] / stream["launches"]]
        roofline_bw = roof(max(big, key=lambda s: s["total_ms"]), "the HBM-streaming kernel with the largest total time (map re-voxelisation)")

    # ---------------------------------------------------------------- whole-path roofline: SURVEY 8(d) formulas with the measured counts
    N, n = args.n_raw - 1, cnt["n"]
    C0, S1, Cd, Sd = cnt["corner"], cnt["surf"], cnt["corner_ds"], cnt["surf_ds"]
    Mraw, Mds = nc + ns, cnt_map["map_corner_ds"] + cnt_map["map_surf_ds"]
    path_bytes = dict(organize=20 * N + 24 * n, smooth_occlusion=16 * n, sector_pick_collect=8 * n + 16 * n + 16 * n,
                      ring_ds=16 * n + 16 * S1, scan_ds=16 * (C0 + S1) + 16 * (Cd + Sd),
                      map_ds=(16 * Mraw + 16 * Mds) if rebuild else 0, grid_build=36 * Mds if rebuild else 0,
                      gn_iterations=128 * Q * args.icp_iters)
    bytes_scan = float(sum(path_bytes.values()))
    path_gbs = bytes_scan * rates[0] / 1e9
    roofline_path = dict(bound="hbm", achieved=round(path_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(path_gbs / HBM_PEAK_GBS, 4),
                         frac_of_measured_copy=(round(path_gbs / copy_gbs, 4) if copy_gbs else None), bytes_per_scan=bytes_scan, stages=path_bytes,
                         note="SURVEY 8(d): compulsory bytes per scan (each input read once, each output written once, measured counts) x scans/s")

    out = dict(
        metric="scans_per_sec_100k_mid360", value=round(rates[0], 2), unit="scans/s",
        n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 4),
        higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
        config=dict(workload=("lidar_odometry scan-to-map, reference-faithful per scan: organise + LOAM feature extraction + voxel grids "
                              "+ re-voxelisation of the raw local map (%s) + KNN index build + %d GN iterations"
                              % ("bounding box and per-bin counts taken inside every rebuild, per scan" if args.map_plan == "per-rebuild" else
                                 "bounding box and per-bin counts taken ONCE per map upload: less than the reference does per scan", args.icp_iters))
                    if rebuild else "lidar_odometry scan-to-map with a frozen downsampled map (DS + index reused)",
                    map_plan=args.map_plan if rebuild else "n/a",
                    n_raw=args.n_raw, map_raw_points=nc + ns, map_ds_points=Mds,
                    scan_features=dict(corner=C0, surf=S1, corner_ds=Cd, surf_ds=Sd),
                    icp_iters=args.icp_iters, map_source=args.map_source, handles_per_gpu=B, scans_per_launch_sequence=NB,
                    scans_in_flight_per_gpu=B * NB * max(args.queue_depth, 1), scans_per_step=world * per_step,
                    enqueue=args.enqueue if NB == 1 else "batched", queue_depth=roll.depth,
                    handle_sync="per step" if not roll.depth else "per handle: before enqueueing a step, wait for the one issued queue_depth steps earlier",
                    sharding="independent scans sharded across ranks, RCCL all_gather of pose records per step"),
        value_windows=dict(n=n_windows, steps_each=args.steps, scans_per_sec=dict(median=round(float(np.median(rates)), 2), min=round(min(rates), 2),
                                                                                    max=round(max(rates), 2), all=[round(r, 1) for r in rates]),
                           window_s=[round(e, 4) for e in win],
                           note="`value` is window 0 alone (exactly --steps steps between fences); the others repeat it back to back"),
        roofline=roofline, roofline_streaming_kernel=roofline_bw, roofline_path=roofline_path,
        results_ok=ok, pose_err_vs_truth=dict(trans_m=err_t, rot_rad=err_r),
        kernel_time_ms_per_step=round(kern_ms, 4), host_enqueue_ms_per_scan=round(enqueue_ms_per_scan, 4),
        top_kernels=[dict(name=s["name"], launches_per_step=s["launches"] / max(args.profile_steps, 1), avg_us=round(s["avg_us"], 2), gbs=s["gbs"])
                     for s in stats[:14]],
        setup_s=round(setup_s, 1),
    )
    if tracker_out is not None:
        out["tracker"] = tracker_out
    if seq_out is not None:
        out["sequential"] = seq_out

    # ---------------------------------------------------------------- CPU baseline: the oracle on this host's cores (rank 0, N=1)
    if world == 1 and rank == 0 and not args.no_cpu:
        from oracle import loader
        ora = loader.load(pkg)
        Po = dict(P); Po["batch_scans"] = 1
        o = pkg.LidarHotpath(ora, **Po)
        o.map_upload(mc, ms)
        times = []
        ora_res = {}                                      # pool index -> the oracle's result for that scan (parity_vs_oracle)
        t_begin = time.perf_counter()
        k = 0
        while True:
            t1 = time.perf_counter()
            o.scan_upload(scans_host[k % args.pool]); o.scan_organize(); o.scan_extract(); o.scan_downsample()
            if rebuild or k == 0:
                o.map_build()
            r_o = o.scan_match(guesses[k % args.pool])
            ora_res.setdefault(k % args.pool, r_o)
            dt = time.perf_counter() - t1
            if k >= 2:                                # two warm-up scans (page faults, OpenMP pool; frozen mode: the map build)
                times.append(dt)
            k += 1
            if (time.perf_counter() - t_begin > args.cpu_seconds and len(times) >= 8) or len(times) >= 60:
                break
        cpu_rate = len(times) / sum(times)
        out["cpu_baseline"] = dict(value=round(cpu_rate, 4), unit="scans/s", cores=8, kind="port",
                                   median_ms_per_scan=round(1e3 * float(np.median(times)), 2),
                                   sample="%d scans of the same workload after 2 warm-up scans (%.1f s), CPU restatement of the reference "
                                          "(oracle/), OpenMP num_threads(8) only on the four loops the reference parallelises" % (len(times), sum(times)))
        out["speedup_vs_cpu"] = round(out["value"] / cpu_rate, 1)
        # parity of what was TIMED against the oracle on the very same scans and guesses (every timed record of a pool
        # scan the oracle leg reached; bar of BASELINE.json: 1e-4 m / 1e-4 rad, status and iteration count equal)
        dm, dr, same, npar = 0.0, 0.0, True, 0
        for j in range(len(rec)):
            kk = int(pool_idx[j])
            if kk not in ora_res:
                continue
            ro = ora_res[kk]
            dm = max(dm, float(np.abs(rec[j, 3:6] - ro["pose"][3:6]).max())); dr = max(dr, float(np.abs(rec[j, 0:3] - ro["pose"][0:3]).max()))
            same = same and int(status[j]) == int(ro["status"]) and int(iters[j]) == int(ro["iters"])
            npar += 1
        out["parity_vs_oracle"] = dict(max_dpose_m=dm, max_dpose_rad=dr, iters_equal=bool(same), n=npar, pool_scans_compared=len(ora_res),
                                       bar=dict(m=1e-4, rad=1e-4), note="PARITY UNPINNED: the checker is the CPU restatement (oracle/)")
        out["results_ok"] = bool(out["results_ok"] and npar > 0 and same and dm <= 1e-4 and dr <= 1e-4)
        if isinstance(out.get("tracker"), dict) and "value" in out["tracker"]:
            # the same LK step on the oracle (scalar, single-threaded restatement of OpenCV's calcOpticalFlowPyrLK; a real
            # OpenCV build would use SIMD and its thread pool — SURVEY 8d caveat)
            w, h = 1280, 720
            img0 = S.make_texture(w, h, 4242)
            img1 = S.warp_homography(img0, S.small_motion_homography(w, h, 100))
            ot = pkg.TrackerHotpath(ora, max_width=w, max_height=h)
            pts = ot.good_features(img0, 150, 0.01, 20.0)
            ot.push_image(img0)
            t1 = time.perf_counter()
            nfr = 4
            for i in range(nfr):
                ot.push_image(img1 if i % 2 == 0 else img0); ot.set_points(pts); ot.run_lk()
            out["tracker"]["cpu_oracle_frames_per_sec"] = round(nfr / (time.perf_counter() - t1), 2)
            ot.close()

    # ---------------------------------------------------------------- secondary: the same steps with the map plan cached per upload
    if rebuild and args.map_plan == "per-rebuild" and args.cached_plan_steps > 0 and args.map_source == "resident":
        for h in hs:
            h.close()
        P2 = dict(P); P2["map_plan_cache"] = 1
        hs2 = [pkg.LidarHotpath(hip, device=dev_index, **P2) for _ in range(B)]
        for b, h in enumerate(hs2):
            if b == 0 or args.no_share_map:
                h.map_upload_device(d_mc.data_ptr(), nc, d_ms.data_ptr(), ns)
            else:
                h.map_share(hs2[0])
            h.map_build(); h.sync()
        roll2 = R.RollingReplay(hs2, issue, lambda j: None, depth=args.queue_depth)
        w2 = max(args.warmup, 3) + min(n_prime, 40)            # fresh handles: the runtime's one-off stall is primed away again
        for i in range(w2):
            roll2.step(i)
        roll2.flush(); dev_sync()
        t0 = time.perf_counter()
        for i in range(args.cached_plan_steps):
            roll2.step(w2 + i)
        roll2.flush(); dev_sync()
        el2 = time.perf_counter() - t0
        out["value_cached_plan"] = dict(scans_per_sec=round(per_step * args.cached_plan_steps / el2, 2), steps=args.cached_plan_steps, this_rank_only=True,
                                        note="lvi_lidar_params.map_plan_cache = 1: bounding box, per-bin counts and partition offsets of the raw map computed once "
                                             "per upload (what round 2's headline measured); not the headline")
        for h in hs2:
            h.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def bench_sequential(pkg, hip, device, dev, n_scans, n_raw, n_keyframes, kf_n_raw):
    """The production shape of the path (strictly sequential): every scan is matched against the local map its
    predecessors built.  Scans (resident in HBM) go one by one through MapOptimizationNode of host/lvi_host.hpp — C++ over the
    C-ABI: updateInitialGuess, extractNearby, map (incremental lvi_map_update | full lvi_map_assemble), scan matching with the
    reference's break rule, saveFrame / keyframe push — with one host sync per scan (the pose decides what happens next).
    Two map sizes: the organic one (only the keyframes this run saves) and the BASELINE one (the node starts with the
    --keyframes keyframes of the headline's local map in its store, all inside the search radius: a ~5 M-point local map)."""
    import torch
    S, H = pkg.synth, pkg.host_api
    hl = pkg.load_host()
    poses = [S.loop_pose(0.3 + 0.027 * k, 0.003 * np.sin(k), -0.003 * np.cos(k)) for k in range(n_scans)]      # 0.26 m per scan: a keyframe every 4th
    d_scans = []
    for k in range(n_scans):
        sc = S.make_scan(n_raw, poses[k], 5000 + k, torch_device=dev)
        d_scans.append(torch.from_numpy(sc.view(np.uint8).reshape(-1, 20).copy()).to(dev))
    torch.cuda.synchronize()
    P = dict(N_SCAN=4, Horizon_SCAN=32768, max_raw_points=131072, max_map_points=6_500_000, max_keyframes=512, max_keyframe_points=6_500_000)
    seeds = []
    ex = pkg.LidarHotpath(hip, device=device, N_SCAN=4, Horizon_SCAN=32768, max_raw_points=131072, max_map_points=1 << 16)
    S.make_map(ex, n_keyframes, kf_n_raw, seed=4711, torch_device=dev, keyframes_out=seeds)
    ex.close()
    out = {}
    for size in ("organic_map", "baseline_map"):
        sec = {}
        ref_pose = None
        for name, inc in (("incremental_map", 1), ("full_assembly_per_scan", 0)):
            # density 0.05 m for the seeded run: every stored key pose survives extractNearby's pose downsampling → all keyframes fused
            m = H.SequentialMapper(hl, hip, pkg.default_params(hip, **P), device=device, incremental_map=inc,
                                   keyframe_density=(0.05 if size == "baseline_map" else 2.0))
            if size == "baseline_map":
                # seeded in loop order, ending with the keyframe next to the first scan: the node's pose after seeding is that key's
                d0 = [float(np.linalg.norm(np.asarray(pose[3:6], np.float64) - poses[0][3:6])) for (_, _, pose) in seeds]
                k0 = int(np.argmin(d0))
                order = list(range(k0 + 1, len(seeds))) + list(range(0, k0 + 1))
                for i in order:
                    c, s_, pose = seeds[i]
                    m.seed_keyframe(c, s_, pose, 0.0)
            t_scan, res = [], []
            # seeded: the stream continues in the map frame of the seeds (true poses); organic: it starts at the first scan
            for k in range(n_scans):
                t0 = time.perf_counter()
                r = m.scan_device(d_scans[k].data_ptr(), n_raw, 10.0 + 0.2 * k)
                t_scan.append(time.perf_counter() - t0)
                res.append(r)
            steady = t_scan[n_scans // 4:]
            # GPU time the local map costs per scan in this form (HIP events, a few more scans of the same stream replayed in place)
            m.handle.prof_reset(); m.handle.prof_enable(True)
            n_prof = 6
            for k in range(n_prof):
                m.scan_device(d_scans[n_scans - 1].data_ptr(), n_raw, 10.0 + 0.2 * (n_scans + k))
            st = m.handle.prof_read(); m.handle.prof_enable(False)
            is_map = lambda nm: nm.startswith(("inc_", "kf_assemble", "grid_")) or nm.endswith(("/map", "/inc"))
            map_us = sum(1e3 * x["total_ms"] for x in st if is_map(x["name"])) / n_prof
            all_us = sum(1e3 * x["total_ms"] for x in st) / n_prof
            sec[name] = dict(scans_per_sec=round(len(steady) / sum(steady), 1), ms_per_scan_median=round(1e3 * float(np.median(steady)), 3),
                             keyframes=res[-1]["n_keyframes"], keys_in_last_map=res[-1]["n_keys"], iters_last=res[-1]["iters"],
                             map_ds_points=m.handle.counts()["map_surf_ds"] + m.handle.counts()["map_corner_ds"], status_last=res[-1]["status"],
                             local_map_kernel_us_per_scan=round(map_us, 1), all_kernel_us_per_scan=round(all_us, 1))
            if ref_pose is None:
                ref_pose = res[-1]["pose"]
            else:
                sec["poses_bit_identical_between_forms"] = bool((res[-1]["pose"].view(np.uint32) == ref_pose.view(np.uint32)).all())
            m.close()
        out[size] = sec
    out["config"] = dict(scans=n_scans, n_raw=n_raw, spacing_m=0.26, seeded_keyframes=len(seeds),
                         note="reference break rule (<= 20 iterations), one host sync per scan; not the headline workload")
    return out


def bench_tracker(pkg, hip, device, rank, world, seconds):
    """LK frames/sec at 1280x720, 150 features, 4 pyramid levels (config 4 of BASELINE.json).  Frame pair i belongs to rank
    i mod world; every rank tracks its share for >= `seconds`, the per-rank rates are gathered and summed."""
    import torch
    S, R = pkg.synth, pkg.replay
    w, h = 1280, 720
    img0 = S.make_texture(w, h, 4242)
    n_frames = 5
    frames = [img0] + [S.warp_homography(img0, S.small_motion_homography(w, h, 100 + i)) for i in range(n_frames - 1)]
    t = pkg.TrackerHotpath(hip, device=device, max_width=w, max_height=h)
    pts = t.good_features(img0, 150, 0.01, 20.0)
    # pair p = (frame p mod 5, frame (p+1) mod 5); this rank's pairs, cycled until the time budget is spent
    mine = R.shard_frames(40 * world, rank, world)
    t.push_image(frames[0])
    n_done, t_total = 0, 0.0
    parts = [0.0, 0.0, 0.0]
    warm = 5
    it = 0
    while True:
        p = mine[it % len(mine)]
        f = frames[(p + 1) % n_frames]                      # the previous push is "cur", this one becomes "forw"
        t0 = time.perf_counter()
        t.push_image(f); t1 = time.perf_counter(); t.set_points(pts); t2 = time.perf_counter(); t.run_lk(); t.sync()
        t3 = time.perf_counter()
        if it >= warm:
            t_total += t3 - t0; n_done += 1
            parts[0] += t1 - t0; parts[1] += t2 - t1; parts[2] += t3 - t2
        it += 1
        if it >= warm and t_total >= seconds and n_done >= 40:
            break
    xy, st, _ = t.get_lk()
    rate_one = n_done / t_total
    # the same with three frame sequences in flight (own handle and stream each, as the lidar leg keeps several scans in flight):
    # a handle is synchronised only right before it receives its next frame, so upload, pyramid and LK of different sequences overlap
    ts = [t] + [pkg.TrackerHotpath(hip, device=device, max_width=w, max_height=h) for _ in range(2)]
    for q in ts[1:]:
        q.push_image(frames[0])
    for q in ts:
        q.sync()
    n_pipe, it2 = 0, 0
    t0 = time.perf_counter()
    while True:
        q = ts[it2 % len(ts)]
        q.sync()
        q.push_image(frames[(mine[(it2 // len(ts)) % len(mine)] + 1) % n_frames]); q.set_points(pts); q.run_lk()
        it2 += 1
        if it2 % 30 == 0 and time.perf_counter() - t0 >= seconds and it2 >= 120:
            break
    for q in ts:
        q.sync()
    t_pipe = time.perf_counter() - t0
    n_pipe = it2
    xy2, st2, _ = ts[1].get_lk()
    rate = n_pipe / t_pipe
    for q in ts[1:]:
        q.close()
    # Shi-Tomasi on the current frame (goodFeaturesToTrack, 150 corners, no mask), device-resident image
    t.set_mask(None); t.run_gftt(150); t.sync()
    t0 = time.perf_counter()
    n_g = 0
    while n_g < 20 or time.perf_counter() - t0 < 0.2:
        t.run_gftt(150); n_g += 1
    t.sync()
    gftt_us = 1e6 * (time.perf_counter() - t0) / n_g
    # per-kernel HIP events of the tracker chain (one frame pair + one GFTT), for the tracker roofline
    t.prof_reset(); t.prof_enable(True)
    for i in range(8):
        t.push_image(frames[(i + 1) % n_frames]); t.set_points(pts); t.run_lk()
        t.run_gftt(150)
    t.sync()
    ks = t.prof_read(); t.prof_enable(False)
    kt = {k["name"]: 1e3 * k["total_ms"] / 8.0 for k in ks}                  # us per frame, per kernel
    lk_us = sum(v for k, v in kt.items() if k.startswith(("pyrdown", "lk_", "clahe")))
    gf_us = sum(v for k, v in kt.items() if not k.startswith(("pyrdown", "lk_", "clahe")))
    lk_bytes, gf_bytes = 2.1e6, 1.85e6                       # per LK frame pair / per GFTT frame
    # the whole node callback (feature_tracker_node.cpp:37-231 through host/lvi_host.hpp, C++ over the C-ABI): CLAHE + pyramid + LK
    # + setMask + Shi-Tomasi + MEI undistortion + message assembly, every frame a publishing frame — what one camera frame costs
    node_fps = None
    try:
        H = pkg.host_api
        hl = pkg.load_host()
        tp = pkg.default_tracker_params(hip, max_width=w, max_height=h, max_cnt=150, min_dist=20.0)
        cam = dict(xi=1.40630886, k1=-0.03678799, k2=0.2610374, p1=0.00144626, p2=0.00035872, gamma1=1454.59041, gamma2=1451.94369, u0=0.5 * w, v0=0.5 * h)
        node = H.TrackerNode(hl, tp, h, w, 1000, equalize=True, cam=cam, device=device)
        tcb = []
        for i in range(700):
            t0 = time.perf_counter()
            r = node.image(frames[i % n_frames], 5.0 + 0.01 * i)
            tcb.append(time.perf_counter() - t0)
        # steady state: the track set is full after a few tens of frames (the first frames pick 150 corners each, later ones ~30);
        # 100-frame blocks of the last 600 frames, the median block is quoted
        blocks = [float(np.median(tcb[k:k + 100])) for k in range(100, 700, 100)]
        tn = float(np.median(blocks))
        node_fps = dict(frames_per_sec=round(1.0 / tn, 1), us_per_frame=round(1e6 * tn, 1), features_last=int(r["n_cur_pts"]),
                        us_per_frame_blocks=[round(1e6 * b, 1) for b in blocks], us_per_frame_first_50=round(1e6 * float(np.mean(tcb[10:60])), 1),
                        note="FeatureTrackerNode::img_callback, equalize = 1 (yaml), every frame published; includes the H2D of the frame and the read of the "
                             "results; median of six 100-frame blocks after the first 100 frames (the mean of frames 10 - 59, the figure of round 2, beside it)")
        node.close()
    except Exception as e:                      # noqa: BLE001
        node_fps = dict(error=str(e))
    total_rate = rate                               # this rank's share; main() gathers and sums the ranks
    per_rank = [rate]
    t.close()

    def rl(nbytes, us, note):
        if not us:
            return None
        gbs = nbytes / (us * 1e-6) / 1e9
        return dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 5), bytes_alg=nbytes,
                    kernel_us=round(us, 1), note=note)
    return dict(metric="lk_frames_per_sec_1280x720_150pts", value=round(total_rate, 1), unit="frames/s", per_rank=[round(r, 1) for r in per_rank],
                frames_timed=n_pipe, seconds_timed=round(t_pipe, 3), sharding="frame pair i -> rank i mod world",
                tracked=int(min(st.sum(), st2.sum())), features=int(len(pts)),
                note="three frame sequences in flight (own handle and stream each); includes the H2D of each new 0.92 MB frame",
                one_sequence=dict(frames_per_sec=round(rate_one, 1), note="one handle, host waits for every frame: the latency of upload + pyramid + LK + sync"),
                gftt_us_per_frame=round(gftt_us, 1),
                us_per_frame=dict(push_image=round(1e6 * parts[0] / n_done, 1), set_points=round(1e6 * parts[1] / n_done, 1), lk_and_sync=round(1e6 * parts[2] / n_done, 1)),
                kernel_us_per_frame={k: round(v, 2) for k, v in sorted(kt.items(), key=lambda kv: -kv[1])}, node_callback=node_fps,
                roofline_lk=rl(lk_bytes, lk_us, "SURVEY 8(d): 2.1 MB per LK frame pair / sum of the pyramid + LK kernel times (HIP events): latency-bound, 150 wavefronts"),
                roofline_gftt=rl(gf_bytes, gf_us, "SURVEY 8(d): 1.85 MB per GFTT frame / sum of the min-eig, compaction, sort and pick kernel times"))


if __name__ == "__main__":
    main()
