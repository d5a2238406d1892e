"""YAML loader of the harness: the reference's two configuration files → the C-ABI parameter structs.

    params_lidar.yaml   (ROS 2 parameter file: /**: ros__parameters: …)   → lvi_lidar_params fields + the caller-loop settings
    params_camera.yaml  (OpenCV FileStorage yaml: %YAML:1.0, !!opencv-matrix) → lvi_tracker_params fields, MEI camera, FREQ, …

Only keys the hot path reads are mapped (utility.h:156-313, feature_tracker/src/parameters.cpp:53-110)."""
import re

import yaml

LIDAR_KEYS = ("N_SCAN", "Horizon_SCAN", "downsampleRate", "lidarMinRange", "lidarMaxRange", "edgeThreshold", "surfThreshold",
              "edgeFeatureMinValidNum", "surfFeatureMinValidNum", "odometrySurfLeafSize", "mappingCornerLeafSize", "mappingSurfLeafSize",
              "z_tollerance", "rotation_tollerance", "imuRPYWeight", "numberOfCores")
CALLER_KEYS = dict(useImuHeadingInitialization="use_imu_heading_initialization", mappingProcessInterval="mapping_process_interval",
                   surroundingkeyframeAddingDistThreshold="keyframe_adding_dist", surroundingkeyframeAddingAngleThreshold="keyframe_adding_angle",
                   surroundingKeyframeDensity="keyframe_density", surroundingKeyframeSearchRadius="keyframe_search_radius")


def load_lidar_yaml(path):
    """→ (lidar params overrides for lidar.default_params, lvh_seq_params overrides for host_api.SequentialMapper, everything else)"""
    doc = yaml.safe_load(open(path))
    node = doc
    for k in ("/**", "ros__parameters"):
        node = node[k]
    lidar = {k: node[k] for k in LIDAR_KEYS if k in node}
    caller = {v: (int(node[k]) if isinstance(node[k], bool) else node[k]) for k, v in CALLER_KEYS.items() if k in node}
    rest = {k: v for k, v in node.items() if k not in lidar and k not in CALLER_KEYS}
    return lidar, caller, rest


def _opencv_yaml(path):
    txt = open(path).read()
    txt = re.sub(r"^%YAML[:\s]*1\.0\s*$", "", txt, flags=re.M)          # FileStorage header is not YAML 1.1 directive syntax
    txt = txt.replace("!!opencv-matrix", "")                            # plain mappings (rows / cols / dt / data)
    return yaml.safe_load(txt)


def load_camera_yaml(path):
    """→ (tracker params overrides, MEI camera dict or None, node settings: freq, F_threshold, equalize, fisheye, image size)"""
    d = _opencv_yaml(path)
    tracker = dict(max_width=int(d["image_width"]), max_height=int(d["image_height"]), max_cnt=int(d["max_cnt"]), min_dist=float(d["min_dist"]))
    cam = None
    if str(d.get("model_type", "")).upper() == "MEI":
        cam = dict(xi=float(d["mirror_parameters"]["xi"]), **{k: float(d["distortion_parameters"][k]) for k in ("k1", "k2", "p1", "p2")},
                   **{k: float(d["projection_parameters"][k]) for k in ("gamma1", "gamma2", "u0", "v0")})
    node = dict(freq=int(d["freq"]) or 100, F_threshold=float(d["F_threshold"]), equalize=int(d["equalize"]), fisheye=int(d["fisheye"]),
                image_width=int(d["image_width"]), image_height=int(d["image_height"]), image_topic=d["image_topic"], point_cloud_topic=d["point_cloud_topic"])
    return tracker, cam, node
