import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as graft
pkg = graft.import_package(); hip = pkg.load_hip(); S = pkg.synth
w, h = 1280, 720
img0 = S.make_texture(w, h, 4242)
frames = [img0] + [S.warp_homography(img0, S.small_motion_homography(w, h, 100 + i)) for i in range(4)]
t = pkg.TrackerHotpath(hip, max_width=w, max_height=h)
pts = t.good_features(img0, 150, 0.01, 20.0)
t.push_image(frames[0])
t.prof_enable(True)
for i in range(200):
    t.push_image(frames[(i + 1) % 5]); t.set_points(pts); t.run_lk()
t.sync()
ks = {k["name"]: 1e3 * k["total_ms"] / k["launches"] for k in t.prof_read()}
print({k: round(v, 2) for k, v in ks.items()})
