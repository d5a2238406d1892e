"""debug build only: phase times of the LK kernel (needs the debug counters written into next_xy / err)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
pkg = graft.import_package(); hip = pkg.load_hip(); S = pkg.synth
w, h = 1280, 720
img0 = S.make_texture(w, h, 4242)
frames = [img0] + [S.warp_homography(img0, S.small_motion_homography(w, h, 100 + i)) for i in range(4)]
t = pkg.TrackerHotpath(hip, max_width=w, max_height=h)
pts = t.good_features(img0, 150, 0.01, 20.0)
t.push_image(frames[0])
for i in range(4):
    t.push_image(frames[(i + 1) % 5]); t.set_points(pts); t.run_lk(); t.sync()
    xy, st, err = t.get_lk()
    tot = xy[:, 0]; ti = xy[:, 1] // 10000; tit = xy[:, 1] % 10000; it = err // 10000; ld = err % 10000
    print("10ns ticks per workgroup: total mean %.0f max %.0f; source phase (4 levels) %.0f; iteration phase %.0f of which region loads %.0f; iterations %.1f" % (tot.mean(), tot.max(), ti.mean(), tit.mean(), ld.mean(), it.mean()))
