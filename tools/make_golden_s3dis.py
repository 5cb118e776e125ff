#!/usr/bin/env python3
"""tests/golden/g11_s3dis/*: two small S3DIS-layout block files WRITTEN BY libhdf5 itself (the library h5py wraps), the two list
files of the dataset directory, and the arrays they hold.

    python tools/make_golden_s3dis.py            # development container only: needs /opt/conda/lib/libhdf5.so

The files have the structure of indoor3d_sem_seg_hdf5_data/ply_data_all_*.h5 (reference data_utils/S3DISDataLoader.py:19-23
reads them with h5py): dataset ``data`` float32 [blocks, 4096, 9], dataset ``label`` uint8 [blocks, 4096], chunked and
gzip-compressed as h5py's ``compression='gzip'`` produces them (one file adds the shuffle filter and chunk shapes that do
not divide the dataset, the other is written the plain way; a third, tiny one is contiguous).  The block contents are
synthetic (no S3DIS data travels).  pointnet12_amd/s3dis.py must read them back bit for bit: tests/test_s3dis_cpu.py.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "g11_s3dis")


def main():
    lib = ctypes.CDLL("/opt/conda/lib/libhdf5.so")
    hid = ctypes.c_int64
    lib.H5open()
    gid = lambda name: hid.in_dll(lib, name).value
    for fn, res, args in [("H5Fcreate", hid, [ctypes.c_char_p, ctypes.c_uint, hid, hid]),
                          ("H5Screate_simple", hid, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
                          ("H5Pcreate", hid, [hid]), ("H5Pset_chunk", ctypes.c_int, [hid, ctypes.c_int, ctypes.c_void_p]),
                          ("H5Pset_deflate", ctypes.c_int, [hid, ctypes.c_uint]), ("H5Pset_shuffle", ctypes.c_int, [hid]),
                          ("H5Dcreate2", hid, [hid, ctypes.c_char_p, hid, hid, hid, hid, hid]),
                          ("H5Dwrite", ctypes.c_int, [hid, hid, hid, hid, hid, ctypes.c_void_p]),
                          ("H5Dclose", ctypes.c_int, [hid]), ("H5Sclose", ctypes.c_int, [hid]), ("H5Pclose", ctypes.c_int, [hid]),
                          ("H5Fclose", ctypes.c_int, [hid])]:
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = res, args
    DCPL = gid("H5P_CLS_DATASET_CREATE_ID_g")
    F32, U8 = gid("H5T_IEEE_F32LE_g"), gid("H5T_STD_U8LE_g")

    def write(path, arrays, chunks, level, shuffle):
        f = lib.H5Fcreate(path.encode(), 2, 0, 0)                      # H5F_ACC_TRUNC
        assert f >= 0
        for name, a in arrays.items():
            dims = (ctypes.c_uint64 * a.ndim)(*a.shape)
            sp = lib.H5Screate_simple(a.ndim, dims, None)
            pl = lib.H5Pcreate(DCPL)
            if chunks.get(name):
                c = (ctypes.c_uint64 * a.ndim)(*chunks[name])
                assert lib.H5Pset_chunk(pl, a.ndim, c) >= 0
                if shuffle:
                    assert lib.H5Pset_shuffle(pl) >= 0
                assert lib.H5Pset_deflate(pl, level[name]) >= 0
            t = F32 if a.dtype == np.float32 else U8
            d = lib.H5Dcreate2(f, name.encode(), t, sp, 0, pl, 0)
            assert d >= 0
            a = np.ascontiguousarray(a)
            assert lib.H5Dwrite(d, t, 0, 0, 0, a.ctypes.data_as(ctypes.c_void_p)) >= 0
            lib.H5Dclose(d); lib.H5Pclose(pl); lib.H5Sclose(sp)
        lib.H5Fclose(f)

    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(11)

    def blocks(n):                                                     # room-block-shaped: 1 m columns, colours, room-relative xyz
        xyz = rng.uniform(0, 1, (n, 4096, 3)) * np.array([1.0, 1.0, 3.0])
        rgb = rng.integers(0, 256, (n, 4096, 3)) / 255.0
        rel = rng.uniform(0, 1, (n, 4096, 3))
        d = np.concatenate([np.round(xyz, 3), rgb, np.round(rel, 3)], -1).astype(np.float32)
        return d, rng.integers(0, 13, (n, 4096)).astype(np.uint8)

    d0, l0 = blocks(3)
    d1, l1 = blocks(2)
    # h5py's layout for these files: gzip level 4 on data, level 1 on label (PointNet's data_prep_util.save_h5)
    write(os.path.join(OUT, "ply_data_all_0.h5"), {"data": d0, "label": l0}, {"data": (1, 1024, 9), "label": (1, 4096)},
          {"data": 4, "label": 1}, False)
    # chunk shapes that do not divide the dataset (edge chunks) + the shuffle filter
    write(os.path.join(OUT, "ply_data_all_1.h5"), {"data": d1, "label": l1}, {"data": (2, 1000, 5), "label": (1, 3000)},
          {"data": 4, "label": 1}, True)
    tiny = {"data": d0[:1, :8].copy(), "label": l0[:1, :8].copy()}
    write(os.path.join(OUT, "contiguous.h5"), tiny, {}, {}, False)
    with open(os.path.join(OUT, "all_files.txt"), "w") as f:
        f.write("indoor3d_sem_seg_hdf5_data/ply_data_all_0.h5\nindoor3d_sem_seg_hdf5_data/ply_data_all_1.h5\n")
    rooms = ["Area_1_office_1", "Area_5_hallway_2", "Area_1_office_1", "Area_5_hallway_2", "Area_6_lounge_1"]
    with open(os.path.join(OUT, "room_filelist.txt"), "w") as f:
        f.write("\n".join(rooms) + "\n")
    np.savez_compressed(os.path.join(OUT, "expected.npz"), d0=d0, l0=l0, d1=d1, l1=l1)
    sys.path.insert(0, ROOT)
    from pointnet12_amd import s3dis
    for name, (d, l) in (("ply_data_all_0.h5", (d0, l0)), ("ply_data_all_1.h5", (d1, l1)), ("contiguous.h5", (tiny["data"], tiny["label"]))):
        rd, rl = s3dis.load_h5(os.path.join(OUT, name))
        assert rd.dtype == np.float32 and rl.dtype == np.uint8 and np.array_equal(rd, d) and np.array_equal(rl, l), name
        print(name, os.path.getsize(os.path.join(OUT, name)), "bytes: read back bit-equal")


if __name__ == "__main__":
    main()
