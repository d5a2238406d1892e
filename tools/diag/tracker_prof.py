"""the tracker chain alone (push_image + LK on 150 features, then Shi-Tomasi), N frames at 1280x720 — the command rocprofv3 traces for
profiles/<round>_rocprof_tracker.md:   rocprofv3 --kernel-trace --output-format csv -d D -- python3 tools/diag/tracker_prof.py 200"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.import_package()
hip = pkg.load_hip()
S = pkg.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w, h = 1280, 720
img0 = S.make_texture(w, h, 4242)
frames = [img0] + [S.warp_homography(img0, S.small_motion_homography(w, h, 100 + i)) for i in range(4)]
t = pkg.TrackerHotpath(hip, max_width=w, max_height=h)
pts = t.good_features(img0, 150, 0.01, 20.0)
t.push_image(frames[0])
for i in range(n):
    t.push_image(frames[(i + 1) % 5]); t.set_points(pts); t.run_lk()
    t.set_mask(None); t.run_gftt(150)
t.sync()
print("frames", n, "features", len(pts))
