"""phase stamps of the one-workgroup GFTT sort + pick at 1280x720: python tools/diag/gftt_cycles.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
pkg = graft.import_package(); hip = pkg.load_hip(); S = pkg.synth
w, h = 1280, 720
img = S.make_texture(w, h, 4242)
t = pkg.TrackerHotpath(hip, max_width=w, max_height=h)
t.push_image(img)
for quota in (150, 30):
    t.set_mask(None); t.run_gftt(quota); t.run_gftt(quota)
    c = t.debug_get(7, np.int64)
    print("quota", quota, dict(hist=int(c[0]), sort=int(c[1]), pick=int(c[2]), total=int(c[3]), bands=int(c[4]), cand=int(c[5]), nacc=int(c[6])))
