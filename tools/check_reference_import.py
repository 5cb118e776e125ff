#!/usr/bin/env python3
"""Build container only: the REFERENCE's unmodified ``model/pointnet2.py`` on top of ``pointnet12_amd.pointnet_util``.

north_star: the product is a drop-in behind ``from .pointnet_util import PointNetSetAbstractionMsg, PointNetSetAbstraction,
PointNetFeaturePropagation`` (model/pointnet2.py:5).  This loads the reference's own model file -- read where it lies under
/root/reference, never copied -- with that relative import resolved to the product module, builds its five networks, and
compares their ``state_dict`` (keys, shapes, ORDER) with the same networks built on the reference's own pointnet_util; the
shipped checkpoint must load ``strict=True`` into the product-backed ``PointNet2SemSeg``.  No GPU: construction only.

    PYTHONDONTWRITEBYTECODE=1 python tools/check_reference_import.py
"""
import importlib.util
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PN2_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True


def load_reference_models_over(util_module, pkg_name):
    """The reference's model/pointnet2.py as ``<pkg_name>.pointnet2`` with ``.pointnet_util`` = util_module."""
    pkg = types.ModuleType(pkg_name)
    pkg.__path__ = [os.path.join(REF, "model")]
    sys.modules[pkg_name] = pkg
    sys.modules[pkg_name + ".pointnet_util"] = util_module
    spec = importlib.util.spec_from_file_location(pkg_name + ".pointnet2", os.path.join(REF, "model", "pointnet2.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    import torch
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from pointnet12_amd import pointnet_util as product_util
    ref_util_spec = importlib.util.spec_from_file_location("pn2_reference_util", os.path.join(REF, "model", "pointnet_util.py"))
    ref_util = importlib.util.module_from_spec(ref_util_spec)
    ref_util_spec.loader.exec_module(ref_util)
    on_product = load_reference_models_over(product_util, "pn2_ref_on_product")
    on_reference = load_reference_models_over(ref_util, "pn2_ref_on_reference")
    nets = [("PointNet2ClsMsg", ()), ("PointNet2ClsSsg", ()), ("PointNet2PartSegSsg", (50,)), ("PointNet2PartSegMsg_one_hot", (50,)),
            ("PointNet2SemSeg", (19, 1))]
    report = {}
    for name, args in nets:
        torch.manual_seed(0)
        a = getattr(on_product, name)(*args)
        torch.manual_seed(0)
        b = getattr(on_reference, name)(*args)
        ka, kb = list(a.state_dict().items()), list(b.state_dict().items())
        assert [k for k, _ in ka] == [k for k, _ in kb], "%s: state_dict keys / order differ" % name
        assert all(x.shape == y.shape and x.dtype == y.dtype for (_, x), (_, y) in zip(ka, kb)), "%s: shapes differ" % name
        # same seed, same registration order -> same initial parameters (the product registers convs, then BatchNorms, as the reference)
        assert all(torch.equal(x, y) for (_, x), (_, y) in zip(ka, kb)), "%s: seeded initial values differ" % name
        report[name] = len(ka)
    ckpt = os.path.join(REF, "checkpoints", "pointnet2-inview-0.55884-0001.pth")
    if os.path.exists(ckpt):
        sd = torch.load(ckpt, map_location="cpu")
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        net = on_product.PointNet2SemSeg(19, 1)
        net.load_state_dict(sd, strict=True)
        report["checkpoint_keys"] = len(sd)
    print("reference model/pointnet2.py over pointnet12_amd.pointnet_util: ok", report)
    return report


if __name__ == "__main__":
    main()
