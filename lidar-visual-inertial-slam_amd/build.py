"""Build recipe of the product library: hipcc → csrc/liblvi_hip.so for gfx950, in-tree
(the .so travels to the GPU box with the snapshot; a JIT cache would not)."""
import os
import subprocess
from shutil import which

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
OUT = os.path.join(CSRC, "liblvi_hip.so")
SOURCES = ["lvi_sort.hip", "lvi_voxel.hip", "lvi_scan.hip", "lvi_icp.hip", "lvi_capi.hip", "lvi_tracker.hip"]
# -ffp-contract=off: voxel keys, KNN distances and the other bit-exact paths must never be fused into FMA
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    deps.append(os.path.join(CSRC, "..", "..", "include", "lvi_hotpath.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    hipcc = which("hipcc") or "/opt/rocm/bin/hipcc"
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed.append((src, out))
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(f"--- {s}\n{o}" for s, o in failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    return OUT


def build_host(verbose=False):
    """host/liblvi_host_hip.so: the C++ host mirror (host/lvi_host.hpp) flattened to C, linked against liblvi_hip.so"""
    from .host_api import HOST_HIP_LIB, build_host_library
    if verbose:
        print("g++ -shared host/lvi_seq_capi.cpp -llvi_hip ->", HOST_HIP_LIB, flush=True)
    return build_host_library(HOST_HIP_LIB, CSRC, "lvi_hip")


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
