import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench, __graft_entry__ as g
pkg = g.import_package(); hip = pkg.load_hip(); S = pkg.synth
def trk(tag):
    print(tag, bench.bench_tracker(pkg, hip, 0)["value"], flush=True)
trk("A alone")
L = pkg.LidarHotpath(hip, N_SCAN=4, Horizon_SCAN=8192, max_raw_points=40000, max_map_points=400000)
scan = S.make_scan(20001, S.loop_pose(0.3), 1)
L.scan_upload(scan); L.scan_organize(); L.scan_extract(); L.sync()
trk("B after lidar work, no profiling")
L.prof_enable(True)
for i in range(3):
    L.scan_upload(scan); L.scan_organize(); L.scan_extract(); L.sync()
st = L.prof_read(); L.prof_enable(False)
trk("C after profiled pass")
L.close()
trk("D after closing the lidar handle")
