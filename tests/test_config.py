"""CPU tier: the defaults of the C-ABI parameter structs are the values of the reference's own configuration files
(config_pkg/config/params_lidar.yaml, params_camera.yaml).  The parsed values are committed as
tests/golden/reference_params.json (generator: tests/golden/make_params_fixture.py); where the reference tree is present —
the build container — the fixture itself is re-derived from the files with the harness' YAML loader and compared."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "reference_params.json")))
REF = "/root/reference/config_pkg/config"


@pytest.mark.parametrize("which", ["oracle", "hip"])
def test_lidar_defaults_are_the_reference_yaml(pkg, oracle, which):
    lib = oracle if which == "oracle" else pkg.load_hip()
    p = pkg.default_params(lib)
    for k, v in FIX["lidar"].items():
        got = getattr(p, k)
        assert got == pytest.approx(v, rel=1e-7), (k, got, v)
    assert p.icp_max_iters == 20 and p.icp_disable_break == 0          # scan2MapOptimization's loop bound (mapOptimization.cpp:1325)


@pytest.mark.parametrize("which", ["oracle", "hip"])
def test_tracker_defaults_are_the_reference_yaml(pkg, oracle, which):
    lib = oracle if which == "oracle" else pkg.load_hip()
    p = pkg.default_tracker_params(lib)
    t = FIX["tracker"]
    # max_width / max_height are CAPACITIES of the handle (benchmark frame 1280x720); the yaml's image must fit
    assert p.max_width >= t["max_width"] and p.max_height >= t["max_height"] and p.max_cnt == t["max_cnt"]
    assert p.min_dist == t["min_dist"]
    # cv:: defaults fixed by the call sites (feature_tracker.cpp:113, 166)
    assert (p.lk_win, p.lk_max_level, p.lk_max_iters) == (21, 3, 30) and p.lk_eps == 0.01 and p.gftt_quality == 0.01


def test_caller_loop_defaults_are_the_reference_yaml(pkg, oracle, tmp_path):
    """MapCallerParams of host/lvi_host.hpp through lvh_seq_params_default"""
    H = pkg.host_api
    out = tmp_path / "liblvi_host_oracle.so"
    H.build_host_library(str(out), os.path.dirname(oracle.path), "lvi_oracle", extra=("-fopenmp",))
    hl = H.HostLibrary(str(out))
    sp = H.SeqParams()
    hl.dll.lvh_seq_params_default(sp)
    for k, v in FIX["caller"].items():
        assert getattr(sp, k) == pytest.approx(v, rel=1e-7), k
    assert FIX["tracker_node"]["freq"] == 20 and FIX["tracker_node"]["F_threshold"] == 1.0 and FIX["tracker_node"]["equalize"] == 1


def test_yaml_loader_round_trip(pkg, oracle, tmp_path):
    """the loader on files of the two formats (written here from the fixture): ROS 2 parameter yaml and OpenCV FileStorage yaml"""
    lid = tmp_path / "params_lidar.yaml"
    lid.write_text("/**:\n  ros__parameters:\n" + "".join(f"    {k}: {json.dumps(v)}\n" for k, v in FIX["lidar"].items())
                   + "    useImuHeadingInitialization: false\n    mappingProcessInterval: 0.15\n    surroundingKeyframeDensity: 2.0\n    sensor: \"livox\"\n")
    lidar, caller, rest = pkg.config.load_lidar_yaml(str(lid))
    assert lidar == FIX["lidar"] and caller["use_imu_heading_initialization"] == 0 and caller["keyframe_density"] == 2.0 and rest["sensor"] == "livox"
    h = pkg.LidarHotpath(oracle, **lidar)                      # the overrides are accepted as they are
    assert h.params.Horizon_SCAN == 6000
    h.close()
    c = FIX["camera"]; n = FIX["tracker_node"]; t = FIX["tracker"]
    cam = tmp_path / "params_camera.yaml"
    cam.write_text("%YAML:1.0\n\nimage_topic: \"/camera/image_raw\"\npoint_cloud_topic: \"/lio_sam/deskew/cloud_deskewed\"\nmodel_type: MEI\n"
                   f"image_width: {t['max_width']}\nimage_height: {t['max_height']}\nmirror_parameters:\n   xi: {c['xi']}\n"
                   f"distortion_parameters:\n   k1: {c['k1']}\n   k2: {c['k2']}\n   p1: {c['p1']}\n   p2: {c['p2']}\n"
                   f"projection_parameters:\n   gamma1: {c['gamma1']}\n   gamma2: {c['gamma2']}\n   u0: {c['u0']}\n   v0: {c['v0']}\n"
                   "extrinsicRotation: !!opencv-matrix\n   rows: 3\n   cols: 3\n   dt: d\n   data: [1.0, 0.0, 0.0,\n          0.0, 1.0, 0.0,\n          0.0, 0.0, 1.0]\n"
                   f"max_cnt: {t['max_cnt']}\nmin_dist: {int(t['min_dist'])}\nfreq: {n['freq']}\nF_threshold: {n['F_threshold']}\nequalize: {n['equalize']}\nfisheye: 0\n")
    tr, mei, node = pkg.config.load_camera_yaml(str(cam))
    assert tr == FIX["tracker"] and mei == FIX["camera"] and node["freq"] == 20 and node["equalize"] == 1


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_fixture_matches_the_reference_files(pkg):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_params_fixture", os.path.join(HERE, "golden", "make_params_fixture.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    assert json.loads(json.dumps(m.parse(), sort_keys=True)) == FIX
