"""MI355X-native hot path of valentinomario/LiDAR-Visual-Inertial-SLAM.

The product is ``csrc/liblvi_hip.so`` (hand-written HIP for gfx950 behind the C-ABI of
``include/lvi_hotpath.h``).  This Python package is harness plumbing: a ctypes binding,
seeded synthetic inputs, and the build recipe.  The directory name contains hyphens, so
import it through ``__graft_entry__.import_package()`` (module name
``lidar_visual_inertial_slam_amd``).
"""
import os

from . import _abi, config, host_api, replay, synth  # noqa: F401
from ._abi import Library, LviError, PT_DTYPE, LIVOX_DTYPE  # noqa: F401
from .lidar import LidarHotpath, default_params  # noqa: F401
from .tracker import TrackerHotpath, default_tracker_params  # noqa: F401

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
HIP_LIB_PATH = os.path.join(PKG_DIR, "csrc", "liblvi_hip.so")

_hip = None


def load_hip():
    """the product library; raises if it has not been built — there is no fallback"""
    global _hip
    if _hip is None:
        _hip = Library(HIP_LIB_PATH)
        if _hip.backend != "hip-gfx950":
            raise RuntimeError(f"{HIP_LIB_PATH} reports backend {_hip.backend!r}")
    return _hip


_host = None


def load_host():
    """host/liblvi_host_hip.so: the C++ host mirror over the product library (built by build.py)"""
    global _host
    if _host is None:
        _host = host_api.HostLibrary(host_api.HOST_HIP_LIB)
    return _host
