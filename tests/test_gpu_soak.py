"""GPU tier: the randomised parity soaks as suite regressions (tests/soak_*.py hold the case generators; by hand they run
longer).  A case may differ from "indices / labels exact, first-stage counts exact, status and iteration count equal, pose
within 1e-4" only for a cause the reference itself leaves open, and the cause is checked case by case (tests/soak_parity.py):
  ties       two equal curvature values in one sector (unstable std::sort, featureExtraction.cpp:171): verified per sector
  second_ds  second-stage grid count off by a few: verified that both libraries agree exactly on ONE first-stage input
  gn / knife iteration count / pose / one or two selected features differ in the staged path: verified that with bit-identical
             inputs (lvi_map_set + lvi_scan_to_map of the oracle's clouds) status, iteration count, every selected count and the
             pose agree
Anything else is "unexplained" and fails.  PARITY UNPINNED: the checker is the CPU restatement (oracle/)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_parity  # noqa: E402
import soak_tracker  # noqa: E402
import soak_voxel_knn  # noqa: E402
from helpers import small_params  # noqa: E402

pytestmark = pytest.mark.gpu

# the 15 cases of the round-1 soak (seed 2024, 200 cases) that differed, each with the class it was put in then, + 25 others
KNOWN = {2: "ties", 8: "ties", 19: "ties", 45: "ties", 47: "second_ds", 66: "gn", 69: "ties", 74: "gn", 80: "ties", 90: "gn", 111: "second_ds",
         113: "ties", 136: "second_ds", 175: "gn", 179: "gn"}


def test_lidar_soak_classified():
    only = sorted(set(KNOWN) | set(range(0, 50, 2)))
    s = soak_parity.run(200, 2024, only=only)
    assert s["executed"] == len(only)
    assert s["classes"]["unexplained"] == 0, s["report"]
    for r in s["report"]:
        assert r["kind"] in ("ties", "second_ds", "gn", "knife"), r
    # the class of a known case must not drift (advisor, round 2): a tie stays a tie, a second-stage count a second-stage count; a
    # Gauss-Newton knife edge may show as "gn" or "knife" (which iteration the one feature flips in follows the last bit of a pose);
    # a known case that no longer differs at all is fine too
    got = {r["case"]: r["kind"] for r in s["report"]}
    allowed = dict(ties=("ties",), second_ds=("second_ds",), gn=("gn", "knife"), knife=("gn", "knife"))
    for case, kind in KNOWN.items():
        if case in got:
            assert got[case] in allowed[kind], (case, kind, got[case])
    # the cascade allowance of the classifier (first difference <= 2 features, later iterations <= 6, final poses within 1e-5) was
    # measured on case 611 of the seed-31337 stream only; no case of THIS list may need it
    for r in s["report"]:
        assert not r.get("cascade", False), r
    # with identical inputs the GN path agrees to far better than the bar in every executed case
    assert s["worst_pose_diff_seam"] < 1e-4
    print("lidar soak:", s["executed"], "cases,", s["differing"], "differing:", s["classes"], "worst staged pose diff %.2e" % s["worst_pose_diff_staged"])


def test_tracker_soak_bit_exact(capsys):
    assert soak_tracker.main(60, 2025) == 0


def test_voxel_knn_soak(capsys):
    assert soak_voxel_knn.main(60, 31) == 0


def test_curvature_tie_sensitivity_report(pkg, oracle, hip):
    """the reference's order between equal curvature values is whatever std::sort leaves (unstable); the oracle
    inherits libstdc++'s introsort, the HIP path uses (value, index).  Noise-free synthetic scans DO hold equal values.  This test
    measures the sensitivity and pins its extent: every label / corner difference lies in a sector that holds a tie (or in the
    5-point spill behind it) and scans without a tied sector are exact"""
    A, S = pkg._abi, pkg.synth
    report = []
    for noise in (0.0, 0.02):
        for k in range(6):
            o = pkg.LidarHotpath(oracle, **small_params()); g = pkg.LidarHotpath(hip, **small_params())
            scan = S.make_scan(20001, S.loop_pose(0.4 + 0.9 * k, 0.0, 0.0), 100 + k, noise=noise)
            for h in (o, g):
                h.scan_upload(scan); h.scan_organize(); h.scan_extract()
            info = o.get_scan_info()
            n = info["n"]
            tied = soak_parity.tied_sectors(info, o.debug_get(A.DBG_CURVATURE, np.float32), 4)
            lo, lg = o.debug_get(A.DBG_LABEL, np.int32), g.debug_get(A.DBG_LABEL, np.int32)
            diff = np.nonzero(lo[5:n - 5] != lg[5:n - 5])[0] + 5
            in_tied = np.zeros(n, bool)
            for (_, _, sp, ep) in tied:
                in_tied[max(sp - 5, 0):ep + 6] = True             # a pick marks +-5 neighbours
            # a tie can only reorder picks inside its own sector and the 5-point spill into the next one of the same ring
            for (ring, j, sp, ep) in tied:
                if j < 5:
                    in_tied[ep:min(ep + 12, n)] = True
            assert in_tied[diff].all(), (noise, k, diff[~in_tied[diff]][:10])
            co, cg = o.debug_get(A.DBG_CORNER_INDEX, np.int32), g.debug_get(A.DBG_CORNER_INDEX, np.int32)
            if not tied:
                np.testing.assert_array_equal(co, cg)
                assert len(diff) == 0
            report.append((noise, k, len(tied), len(diff), int(len(co) != len(cg) or (co != cg).any())))
            o.close(); g.close()
    # Measured: equal curvature BITS also occur with range noise (a sector holds ~3 300 values d*d of f32 differences: a handful
    # of collisions per scan), and now and then one sits where it decides a pick — scan 104 of this list: 3 labels.  The extent
    # stays tiny and confined to the tied sectors (asserted above); nothing outside them may differ.
    for noise, k, n_tied, n_diff, corners_differ in report:
        assert n_diff <= 40, (noise, k, n_diff)
    assert sum(r[3] for r in report if r[0] > 0) <= 12
    print("tie sensitivity (noise, scan, tied sectors, differing labels, corner list differs):", report)
