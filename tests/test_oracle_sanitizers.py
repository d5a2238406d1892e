"""CPU tier: the oracle and the C++ host mirror built once with AddressSanitizer + UBSan and once with
ThreadSanitizer, driven through the lidar chain, the tracker chain and the f-row extras of host/replay_main.cpp.

TSan and OpenMP: this image's libgomp is not TSan-instrumented, so the implicit barrier at the end of a parallel region is
invisible to TSan and every access of the main thread AFTER a region "races" with the workers' accesses inside it.  Those
reports have one stack outside any `._omp_fn` clone.  A real race — two threads inside parallel regions, e.g. the per-index
flag writes of cornerOptimization / surfOptimization (mapOptimization.cpp:1010,1102; SURVEY App. B.9: std::vector<bool> in the
reference, one byte per flag here) — has both stacks inside `._omp_fn` frames; only those fail the test."""
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import make_small_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "lidar-visual-inertial-slam_amd", "host", "replay_main.cpp")] + \
      [os.path.join(ROOT, "oracle", f) for f in ("lvo_lidar.cpp", "lvo_tracker.cpp", "lvo_test_exports.cpp")]


@pytest.fixture(scope="module")
def inputs(pkg, oracle, tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    S = pkg.synth
    sc = make_small_scene(pkg, oracle, n_raw=8001, n_kf=6, Horizon_SCAN=4096)
    sc["scan"].tofile(d / "scan.bin"); sc["map_corner"].tofile(d / "mc.bin"); sc["map_surf"].tofile(d / "ms.bin")
    img0 = S.make_texture(240, 180, 9)
    img1 = S.warp_homography(img0, S.small_motion_homography(240, 180, 2, 3.0))
    img0.tofile(d / "a.bin"); img1.tofile(d / "b.bin")
    runs = [["lidar", "4096", str(d / "scan.bin"), str(len(sc["scan"])), str(d / "mc.bin"), str(len(sc["map_corner"])), str(d / "ms.bin"),
             str(len(sc["map_surf"]))] + ["%.9g" % v for v in sc["guess"]],
            ["track", "240", "180", str(d / "a.bin"), str(d / "b.bin"), "60", "12"],
            ["extras", "4096", str(d / "scan.bin"), str(len(sc["scan"])), "240", "180", str(d / "a.bin")]]
    return d, runs


def _build(out, flags):
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-fno-omit-frame-pointer", *flags, "-o", str(out), *SRC],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_address_and_undefined_behaviour_sanitizers(inputs):
    d, runs = inputs
    exe = d / "replay_asan"
    _build(exe, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    for args in runs:
        r = subprocess.run([str(exe)] + args, capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=1"))
        assert r.returncode == 0, (args[0], r.stderr[-3000:])
        assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (args[0], r.stderr[-3000:])
        assert "backend cpu-oracle" in r.stdout


def test_thread_sanitizer_parallel_regions(inputs):
    d, runs = inputs
    exe = d / "replay_tsan"
    _build(exe, ["-fsanitize=thread"])
    real = []
    for args in runs[:1] + runs[2:]:                     # the OpenMP loops are on the lidar side
        r = subprocess.run([str(exe)] + args, capture_output=True, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS="4", TSAN_OPTIONS="halt_on_error=0 exitcode=0 history_size=4"))
        assert "backend cpu-oracle" in r.stdout, r.stderr[-2000:]
        for block in r.stderr.split("WARNING: ThreadSanitizer: data race")[1:]:
            block = block.split("SUMMARY:")[0]
            # the first frame (#0) of each of the two access stacks
            tops = re.findall(r"(?:Write|Read|Previous write|Previous read|Atomic write|Atomic read|Previous atomic \w+) of size \d+ .*?\n\s+#0 ([^\n]*)", block)
            if len(tops) >= 2 and all("._omp_fn" in t for t in tops[:2]):
                real.append(block[:1500])
    assert not real, "data race between threads INSIDE parallel regions:\n" + "\n----\n".join(real[:3])


def test_openmp_thread_count_does_not_change_a_bit(pkg, oracle, inputs):
    """the complement of the TSan filter above: numberOfCores = 1 and 8 (the num_threads clause of the four loops the reference
    parallelises) give the same output text (poses printed to 9 digits)"""
    d, runs = inputs
    exe = d / "replay_plain"
    _build(exe, [])
    for args in (runs[0], runs[2]):
        outs = [subprocess.run([str(exe)] + args, capture_output=True, text=True, env=dict(os.environ, LVI_NUMBER_OF_CORES=str(t))).stdout for t in (1, 8)]
        assert outs[0] == outs[1] and "backend cpu-oracle" in outs[0]
