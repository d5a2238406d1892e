"""the sequential replay alone (organic map, incremental local map), for a rocprofv3 kernel trace: python tools/diag/seq_run.py [n_scans]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
import torch
pkg = graft.import_package(); hip = pkg.load_hip(); hl = pkg.load_host()
S, H = pkg.synth, pkg.host_api
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)
poses = [S.loop_pose(0.3 + 0.027 * k, 0.003 * np.sin(k), -0.003 * np.cos(k)) for k in range(n_scans)]
d_scans = []
for k in range(n_scans):
    sc = S.make_scan(100001, poses[k], 5000 + k, torch_device=dev)
    d_scans.append(torch.from_numpy(sc.view(np.uint8).reshape(-1, 20).copy()).to(dev))
torch.cuda.synchronize()
P = dict(N_SCAN=4, Horizon_SCAN=32768, max_raw_points=131072, max_map_points=6_500_000, max_keyframes=512, max_keyframe_points=6_500_000)
m = H.SequentialMapper(hl, hip, pkg.default_params(hip, **P), device=0, incremental_map=1, keyframe_density=2.0)
ts = []
for k in range(n_scans):
    t0 = time.perf_counter(); r = m.scan_device(d_scans[k].data_ptr(), 100001, 10.0 + 0.2 * k); ts.append(time.perf_counter() - t0)
print("ms per scan (last half):", 1e3 * np.median(ts[n_scans // 2:]), "iters", r["iters"], "keys", r["n_keys"])
m.close()
