"""S3DIS block files: pointnet12_amd/s3dis.py reads HDF5 files written by libhdf5 itself (the fixtures of
tools/make_golden_s3dis.py, the layout of indoor3d_sem_seg_hdf5_data) bit for bit, without h5py."""
import os

import numpy as np
import pytest

from pointnet12_amd import s3dis

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_s3dis")


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(ROOT, "expected.npz"))


@pytest.mark.parametrize("name,key", [("ply_data_all_0.h5", "0"), ("ply_data_all_1.h5", "1")])
def test_load_h5_reads_libhdf5_files_bit_exactly(expected, name, key):
    """Chunked + gzip (h5py's compression='gzip'), with and without the shuffle filter, chunk shapes that divide the dataset
    and chunk shapes that leave partial edge chunks."""
    data, label = s3dis.load_h5(os.path.join(ROOT, name))
    assert data.dtype == np.float32 and data.shape[1:] == (4096, 9) and label.dtype == np.uint8 and label.shape[1:] == (4096,)
    assert np.array_equal(data.view(np.uint32), expected["d" + key].view(np.uint32))
    assert np.array_equal(label, expected["l" + key])


def test_contiguous_layout_and_errors(expected, tmp_path):
    data, label = s3dis.load_h5(os.path.join(ROOT, "contiguous.h5"))
    assert np.array_equal(data, expected["d0"][:1, :8]) and np.array_equal(label, expected["l0"][:1, :8])
    with pytest.raises(KeyError):
        s3dis.read_datasets(os.path.join(ROOT, "contiguous.h5"), ("nope",))
    bad = tmp_path / "x.h5"
    bad.write_bytes(b"not hdf5 at all" * 100)
    with pytest.raises(ValueError):
        s3dis.load_h5(str(bad))


def test_recognize_all_data_splits_by_area(expected):
    """S3DISDataLoader.py:29-57: the files of all_files.txt concatenated, rooms of Area_5 held out."""
    tr_d, tr_l, te_d, te_l = s3dis.recognize_all_data(ROOT, test_area=5)
    alld = np.concatenate([expected["d0"], expected["d1"]])
    alll = np.concatenate([expected["l0"], expected["l1"]])
    assert np.array_equal(tr_d, alld[[0, 2, 4]]) and np.array_equal(tr_l, alll[[0, 2, 4]])
    assert np.array_equal(te_d, alld[[1, 3]]) and np.array_equal(te_l, alll[[1, 3]])
    ds = s3dis.S3DISDataLoader(tr_d, tr_l)
    assert len(ds) == 3 and ds[1][0].shape == (4096, 9) and np.array_equal(ds[1][1], alll[2])
    np.random.seed(0)
    aug = s3dis.S3DISDataLoader(tr_d, tr_l, data_augmentation=True)[0][0]
    assert aug.dtype == np.float32 and 0 < float(np.abs(aug - tr_d[0]).max()) <= 0.05 + 1e-6
