"""the tracker NODE path alone (FeatureTrackerNode::img_callback through the C++ host mirror): N frames at 1280x720, every frame
published.  Under `rocprofv3 --kernel-trace --output-format csv -d D -- python3 tools/diag/node_run.py N` the trace is the node's
launch chain;  `python3 tools/diag/node_run.py chain D` prints one frame's chain (kernel, start offset, duration, gap before it)."""
import csv, glob, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def chain(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            r["Kernel_Name"] = "COPY " + r.get("Direction", "")
            rows.append(r)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    nm = lambda r: (re.search(r"(\w+_kernel)", r["Kernel_Name"]) or re.search(r"(.{1,40})", r["Kernel_Name"])).group(1)
    # a frame starts with the first pyrdown / clahe kernel after a gftt pick
    marks = [i for i, r in enumerate(rows) if "clahe" in r["Kernel_Name"] and (i == 0 or "clahe" not in rows[i - 1]["Kernel_Name"])]
    if len(marks) < 12:
        marks = [i for i, r in enumerate(rows) if "pyrdown" in r["Kernel_Name"] and "pyrdown" not in rows[i - 1]["Kernel_Name"]]
    a, b = marks[-6], marks[-5]
    t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
    busy = 0.0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{nm(r):28s} +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:7.2f}")
        busy += (e - s) / 1e3; prev_end = e
    period = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    print(f"frame period {period:.1f} us, kernel-busy {busy:.1f} us, {b - a} launches")
    per = [(int(rows[marks[i + 1]]["Start_Timestamp"]) - int(rows[marks[i]]["Start_Timestamp"])) / 1e3 for i in range(len(marks) // 2, len(marks) - 1)]
    print("median period of the last half:", sorted(per)[len(per) // 2])


if len(sys.argv) > 2 and sys.argv[1] == "chain":
    chain(sys.argv[2]); sys.exit(0)

import numpy as np
import __graft_entry__ as graft
pkg = graft.import_package(); hip = pkg.load_hip(); hl = pkg.load_host()
S, H = pkg.synth, pkg.host_api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w, h = 1280, 720
img0 = S.make_texture(w, h, 4242)
frames = [img0] + [S.warp_homography(img0, S.small_motion_homography(w, h, 100 + i)) for i in range(4)]
tp = pkg.default_tracker_params(hip, max_width=w, max_height=h, max_cnt=150, min_dist=20.0)
cam = dict(xi=1.40630886, k1=-0.03678799, k2=0.2610374, p1=0.00144626, p2=0.00035872, gamma1=1454.59041, gamma2=1451.94369, u0=0.5 * w, v0=0.5 * h)
node = H.TrackerNode(hl, tp, h, w, 1000, equalize=True, cam=cam, device=0)
ts = []
for i in range(n):
    t0 = time.perf_counter(); r = node.image(frames[i % 5], 5.0 + 0.01 * i); ts.append(time.perf_counter() - t0)
blocks = [1e6 * float(np.median(ts[i:i + 100])) for i in range(100, n - 99, 100)] or [1e6 * float(np.median(ts[n // 2:]))]
print("us per frame: best 100-frame block %.1f, median block %.1f, worst %.1f (%d blocks); features %d" % (min(blocks), float(np.median(blocks)), max(blocks), len(blocks), int(r["n_cur_pts"])))
node.close()
