"""CPU: the SemanticKITTI scan reader (pointnet12_amd/kitti.py, host-side parsing) against what the reference's own
``Semantic_KITTI_Utils.get`` returned for the same files (tests/golden/g9_kitti.npz, tools/make_golden_kitti.py)."""
import os

import numpy as np
import pytest

from conftest import golden
from pointnet12_amd import kitti


def files(tmp_path, g):
    fv, fl = os.path.join(tmp_path, "000003.bin"), os.path.join(tmp_path, "000003.label")
    g["bin"].tofile(fv)
    g["label"].tofile(fl)
    return fv, fl, {int(k): int(v) for k, v in zip(g["map_keys"], g["map_values"])}


@pytest.mark.parametrize("subset", ["all", "inview"])
def test_read_scan_golden(tmp_path, subset):
    g = golden("g9_kitti.npz")
    fv, fl, lmap = files(tmp_path, g)
    pts, lab = kitti.read_scan(fv, fl, lmap, subset)
    assert pts.dtype == np.float32 and lab.dtype == np.int32
    assert pts.shape == g[subset + "/points"].shape
    assert (pts.view(np.uint32) == g[subset + "/points"].view(np.uint32)).all()
    assert (lab == g[subset + "/labels"]).all()
    assert lab.min() >= 0 and lab.max() <= 18                      # class 0 dropped, the rest shifted down


def test_field_of_view_borders():
    """Strict inequalities on float32 angles; elevation is atan2(z, 3-D range) as the reference computes it."""
    def p(az_deg, el_sin, r=10.0):
        a = np.deg2rad(az_deg)
        return [r * np.cos(a), r * np.sin(a), r * el_sin, 0.5]
    pts = np.array([p(0, 0), p(39.99, 0), p(40.01, 0), p(-39.99, 0), p(-40.01, 0), p(179, 0),
                    p(0, 0.36), p(0, 0.38), p(0, -0.36), p(0, -0.38)], np.float32)
    # atan2(z, d) = 20 deg  <=>  z / d = tan(20 deg) = 0.364 with d = sqrt(x^2 + y^2 + z^2): z = 10 * s, d = 10 * sqrt(1 + s^2)
    # s = 0.36 -> z/d = 0.339 (inside), s = 0.38 -> 0.355 (inside too: the 3-D range widens the cone to s < 0.391)
    m = kitti.in_view(pts)
    assert m.tolist() == [True, True, False, True, False, False, True, True, True, True]
    assert not kitti.in_view(np.array([p(0, 0.40), p(0, -0.40), [np.nan, 0, 0, 0]], np.float32)).any()


def test_read_scan_errors(tmp_path):
    g = golden("g9_kitti.npz")
    fv, fl, lmap = files(tmp_path, g)
    g["label"][:-1].tofile(fl)
    with pytest.raises(ValueError):                                # kitti_utils.py:211
        kitti.read_scan(fv, fl, lmap)
    g["label"].tofile(fl)
    del lmap[10]
    with pytest.raises(KeyError):                                  # the reference's dict lookup raises as well
        kitti.read_scan(fv, fl, lmap)
    with pytest.raises(AssertionError):
        kitti.read_scan(fv, fl, lmap, "front")
