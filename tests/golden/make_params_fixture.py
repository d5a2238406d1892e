"""TEST INFRASTRUCTURE — run in the build container (the reference tree does not travel to the GPU box):
parses /root/reference/config_pkg/config/params_{lidar,camera}.yaml with the harness' own loader and commits the values
the hot path reads as tests/golden/reference_params.json (data, not source).    python tests/golden/make_params_fixture.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

REF = "/root/reference/config_pkg/config"


def parse():
    pkg = graft.import_package()
    lidar, caller, _ = pkg.config.load_lidar_yaml(os.path.join(REF, "params_lidar.yaml"))
    tracker, cam, node = pkg.config.load_camera_yaml(os.path.join(REF, "params_camera.yaml"))
    return dict(source=dict(lidar="config_pkg/config/params_lidar.yaml", camera="config_pkg/config/params_camera.yaml"),
                lidar=lidar, caller=caller, tracker=tracker, camera=cam, tracker_node=node)


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_params.json")
    json.dump(parse(), open(out, "w"), indent=1, sort_keys=True)
    print(open(out).read())
