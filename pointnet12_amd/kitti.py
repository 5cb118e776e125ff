"""SemanticKITTI scans from disk into the resident scan store.

Host-side parsing of the two per-scan files the reference trains on -- ``velodyne/%06d.bin`` (float32 x, y, z,
intensity) and ``labels/%06d.label`` (uint32: semantic class in the low 16 bits, instance id above) -- restating
``Semantic_KITTI_Utils.get`` (data_utils/kitti_utils.py:183-227): map the raw class through ``learning_map``, drop
the points of class 0 and shift the rest down by one (:215-221), and for the ``inview`` subset keep the camera's
field of view (:223-227 with ``points_basic_filter`` :259-280).  This runs once per scan when the store is filled;
everything per batch happens on the device (loader.prepare_batch).

The filter keeps the reference's exact forms: the azimuth test is ``-40 deg < atan2(y, x) < 40 deg`` on float32
angles, and the elevation test takes ``atan2(z, d)`` with d the full 3-D range ``sqrt(x^2 + y^2 + z^2)`` (not the
ground range), bounds -20 deg .. 20 deg (:263-268, :237-249).
"""
import numpy as np

from . import loader


def in_view(points, h_fov=(-40, 40), v_fov=(-20, 20)):
    """Boolean mask of the points inside the horizontal / vertical field of view (kitti_utils.py:259-280; the box
    limits of :251-257 are +-10 000 m, i.e. never active, and are kept for NaN parity: a NaN coordinate fails them)."""
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    d = np.sqrt(x ** 2 + y ** 2 + z ** 2)
    az = np.arctan2(y, x)
    el = np.arctan2(z, d)
    keep = np.logical_and(az > (-h_fov[1] * np.pi / 180), az < (-h_fov[0] * np.pi / 180))
    keep = np.logical_and(keep, np.logical_and(el < (v_fov[1] * np.pi / 180), el > (v_fov[0] * np.pi / 180)))
    lim = 10000
    box = np.logical_and.reduce((x > -lim, x < lim, y > -lim, y < lim, z > -lim, z < lim, d > -lim, d < lim))
    return np.logical_and(keep, box)


def read_scan(fn_velo, fn_label, learning_map, subset="all"):
    """One scan: ``(points [M, 4] float32, labels [M] int32 in 0..num_classes-1)`` as ``Semantic_KITTI_Utils.get``
    returns them.  ``learning_map``: dict raw class -> training class (0 = ignored), the ``learning_map`` block of
    the dataset's ``semantic-kitti.yaml``."""
    if subset not in ("all", "inview"):
        raise AssertionError(subset)
    points = np.fromfile(fn_velo, dtype=np.float32).reshape(-1, 4)
    raw = np.fromfile(fn_label, dtype=np.uint32).reshape(-1)
    if raw.shape[0] != points.shape[0]:
        raise ValueError("Scan and Label don't contain same number of points")
    sem = raw & 0xFFFF
    lut = np.full(int(max(max(learning_map), int(sem.max()) if sem.size else 0)) + 1, -1, np.int64)
    for k, v in learning_map.items():
        lut[int(k)] = int(v)
    label = lut[sem]
    if (label < 0).any():
        raise KeyError(int(sem[label < 0][0]))                  # a raw class missing from the map (dict lookup raises)
    label = label.astype(np.int32)
    keep = label != 0                                            # drop class 0, shift the others down (:218-221)
    points, label = points[keep], label[keep] - 1
    if subset == "inview":
        m = in_view(points)
        points, label = points[m], label[m]
    return points, label


def load_scans(pairs, learning_map, subset="inview", device="cuda"):
    """``ScanStore`` over the ``(bin_path, label_path)`` pairs (a sequence's scans, e.g. every second one for
    training as SemKITTI_Loader.py:62-66 selects them)."""
    scans, labels = [], []
    for fn_velo, fn_label in pairs:
        p, l = read_scan(fn_velo, fn_label, learning_map, subset)
        scans.append(p)
        labels.append(l)
    return loader.ScanStore(scans, labels, device)
