"""Harness test-case file formats and the sweep's loader (src/test-data/testCases.ts:34-52,
src/test-data/saveTestCaseToFile.ts). CPU only."""
import os

import pyref as R
from webgpu_msm_bls12_377_amd.host import test_data as T


def test_round_trip_in_reference_format(tmp_path):
    pts = [{"x": R.G[0], "y": R.G[1], "z": 1}, {"x": R.FIXED_BASE[0], "y": R.FIXED_BASE[1], "z": 1}]
    ks = [1, R.R_ORDER - 1]
    os.makedirs(tmp_path / "points")
    os.makedirs(tmp_path / "scalars")
    T.save_points_to_file(pts, str(tmp_path / "points" / "16-power-points.txt"))
    T.save_scalars_to_file(ks, str(tmp_path / "scalars" / "16-power-scalars.txt"))
    tc = T.load_test_case(16, str(tmp_path))
    assert tc["baseAffinePoints"] == pts and tc["scalars"] == ks
    assert tc["expectedResult"]["x"] == T.EXPECTED_RESULTS[16][0]
    # the saver's own `"<dec>",` form parses too
    T.save_scalars_to_file(ks, str(tmp_path / "scalars" / "17-power-scalars.txt"), loader_format=False)
    with open(tmp_path / "scalars" / "17-power-scalars.txt") as f:
        assert T.parse_scalars_text(f.read()) == ks


def test_known_answers_are_curve_points():
    for power, pt in T.EXPECTED_RESULTS.items():
        assert R.on_curve(pt), power
    assert T.get_expected_result(15) == {"x": 0, "y": 0}
