"""CPU oracle for the PointNet++ set-abstraction / feature-propagation path.

TEST INFRASTRUCTURE ONLY.  Importers allowed: ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``.  ``pointnet12_amd`` never imports this
package; its HIP path raises when the HIP library is missing instead of falling back.

``oracle.geometry``   numpy front-end of ``pn2_oracle.c`` (scalar C restatement of
                      ``model/pointnet_util.py:19-157,295-301`` with explicit rounding order)
``oracle.torch_ref``  torch-CPU restatement of the three modules and the two benchmark
                      networks, built on ``oracle.geometry`` indices

Parity pin: ``tools/make_golden.py`` imports the reference itself (possible only in the
development container), asserts this oracle bit-equal on every index tensor and on the raw
fp32 distance matrix, within 5e-6 on module outputs, and writes ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` re-checks the oracle against those fixtures everywhere.
"""
from . import geometry  # noqa: F401
