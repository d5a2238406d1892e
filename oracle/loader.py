"""TEST INFRASTRUCTURE — builds (if needed) and loads oracle/liblvi_oracle.so through the
package's generic ctypes binding.  Never imported by the product package."""
import os
import subprocess
from shutil import which

ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(ORACLE_DIR, "liblvi_oracle.so")

_lib = None


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".h"))]
    srcs.append(os.path.join(ORACLE_DIR, "..", "include", "lvi_hotpath.h"))
    stale = force or not os.path.exists(ORACLE_LIB) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs)
    if stale:
        r = subprocess.run(["make", "-C", ORACLE_DIR, "-B"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return ORACLE_LIB


def load(pkg):
    """pkg = the imported lidar_visual_inertial_slam_amd package (for its Library class)"""
    global _lib
    if _lib is None:
        if which("g++") and which("make"):
            build()
        _lib = pkg.Library(ORACLE_LIB)
        assert _lib.backend == "cpu-oracle"
    return _lib
