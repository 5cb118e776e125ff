"""CPU, build container only: the reference's unmodified model/pointnet2.py imports the product's pointnet_util (north_star's
drop-in boundary, model/pointnet2.py:5).  Skipped where /root/reference does not exist (the GPU box): nothing of the reference
travels, the file is read where it lies."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PN2_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "model", "pointnet2.py")), reason="the reference is not on this machine")
def test_reference_model_file_runs_on_the_product_modules():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_reference_import as C
    report = C.main()
    assert set(report) >= {"PointNet2ClsMsg", "PointNet2ClsSsg", "PointNet2PartSegSsg", "PointNet2PartSegMsg_one_hot", "PointNet2SemSeg"}
    assert report.get("checkpoint_keys", 156) == 156
