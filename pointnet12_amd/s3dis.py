"""S3DIS room blocks from disk: the ``indoor3d_sem_seg_hdf5_data`` files the reference trains its
S3DIS networks on, restating ``data_utils/S3DISDataLoader.py``:

* ``load_h5`` (:19-23): ``data`` float32 [blocks, 4096, 9] (x, y, z, r, g, b, room-normalised x, y, z) and ``label`` uint8
  [blocks, 4096] of one ``ply_data_all_*.h5`` file.  The reference calls ``h5py``; this image has no h5py and the path must
  not depend on one, so the HDF5 container is parsed here directly -- the subset those files use and nothing else:
  superblock version 0/1, version-1 object headers (with continuation blocks), old-style groups (symbol table: version-1
  B-tree + local heap), datasets of fixed-point / IEEE float type with compact, contiguous or chunked layout (version-1
  chunk B-tree) and the deflate / shuffle filters (h5py's ``compression='gzip'``).  Anything else raises ``ValueError``
  naming what it met.
* ``recognize_all_data`` (:29-57): every file of ``all_files.txt`` concatenated, split by ``room_filelist.txt`` into the
  blocks of ``Area_<test_area>`` (test) and the rest (train).
* ``S3DISDataLoader`` (:59-77): the map-style dataset over (block, labels).  The reference's ``data_augmentation`` branch
  reads a variable that does not exist (``pcd``, :73) and raises NameError when enabled; here it jitters the block the way that
  branch evidently meant to (``jitter_point_cloud``: sigma 0.01, clip 0.05 -- data_utils/augmentation.py).

Host-side numpy; the blocks then go to the device through ``torch.as_tensor(...).cuda()`` like any other array.
"""
import os
import zlib

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"


class _H5:
    def __init__(self, buf):
        self.b = buf
        base = 0
        while buf[base:base + 8] != _SIG:                        # the superblock may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base + 8 > len(buf):
                raise ValueError("not an HDF5 file")
        ver = buf[base + 8]
        if ver not in (0, 1):
            raise ValueError("HDF5 superblock version %d (only the classic 0/1 layout is read here)" % ver)
        self.O, self.L = buf[base + 13], buf[base + 14]          # size of offsets / of lengths
        p = base + 24 + (4 if ver == 1 else 0)
        self.base = self._addr(p)
        p += 4 * self.O                                          # base, free-space, end-of-file, driver-info addresses
        self.root = self._addr(p + self.O)                       # root symbol table entry: link name offset, header address

    # ---- primitives
    def _u(self, p, n):
        return int.from_bytes(self.b[p:p + n], "little")

    def _addr(self, p):
        v = self._u(p, self.O)
        return None if v == (1 << (8 * self.O)) - 1 else v

    # ---- object header (version 1) -> list of (type, payload offset, size)
    def messages(self, addr):
        b, p = self.b, self.base + addr
        if b[p] != 1:
            raise ValueError("object header version %d (only version 1 is read here)" % b[p])
        n = self._u(p + 2, 2)
        size = self._u(p + 8, 4)
        blocks = [(p + 16, size)]
        out = []
        while blocks and len(out) < n:
            q, left = blocks.pop(0)
            end = q + left
            while q + 8 <= end and len(out) < n:
                mtype, msize = self._u(q, 2), self._u(q + 2, 2)
                body = q + 8
                if mtype == 0x10:                                # continuation: offset, length
                    blocks.append((self.base + self._addr(body), self._u(body + self.O, self.L)))
                out.append((mtype, body, msize))
                q = body + msize
        return out

    # ---- old-style group: name -> object header address
    def children(self, addr):
        stab = [m for m in self.messages(addr) if m[0] == 0x11]
        if not stab:
            raise ValueError("group without a symbol table message (new-style link messages are not read here)")
        btree, heap = self._addr(stab[0][1]), self._addr(stab[0][1] + self.O)
        hp = self.base + heap
        if self.b[hp:hp + 4] != b"HEAP":
            raise ValueError("bad local heap")
        data = self.base + self._addr(hp + 8 + 2 * self.L)
        names = {}

        def walk(node):
            p = self.base + node
            if self.b[p:p + 4] != b"TREE" or self.b[p + 4] != 0:
                raise ValueError("bad group B-tree node")
            level, used = self.b[p + 5], self._u(p + 6, 2)
            q = p + 8 + 2 * self.O
            for i in range(used):
                child = self._addr(q + self.L + i * (self.L + self.O))
                if level:
                    walk(child)
                    continue
                s = self.base + child
                if self.b[s:s + 4] != b"SNOD":
                    raise ValueError("bad symbol table node")
                for e in range(self._u(s + 6, 2)):
                    ent = s + 8 + e * (2 * self.O + 24)
                    off = data + self._u(ent, self.O)
                    names[self.b[off:self.b.index(b"\0", off)].decode()] = self._addr(ent + self.O)
        walk(btree)
        return names

    # ---- dataset
    def dataset(self, addr):
        shape = dtype = layout = None
        filters = []
        for mtype, p, size in self.messages(addr):
            b = self.b
            if mtype == 0x01:                                    # dataspace
                ver, rank = b[p], b[p + 1]
                q = p + (8 if ver == 1 else 4)
                shape = tuple(self._u(q + i * self.L, self.L) for i in range(rank))
            elif mtype == 0x03:                                  # datatype
                cls, bits0, nbytes = b[p] & 15, b[p + 1], self._u(p + 4, 4)
                order = ">" if bits0 & 1 else "<"
                if cls == 0:
                    dtype = np.dtype("%s%s%d" % (order, "i" if bits0 & 8 else "u", nbytes))
                elif cls == 1:
                    dtype = np.dtype("%sf%d" % (order, nbytes))
                else:
                    raise ValueError("HDF5 datatype class %d (only integers and IEEE floats are read here)" % cls)
            elif mtype == 0x08:                                  # data layout
                if b[p] != 3:
                    raise ValueError("data layout message version %d (only version 3 is read here)" % b[p])
                cls = b[p + 1]
                if cls == 0:
                    n = self._u(p + 2, 2)
                    layout = ("compact", p + 4, n)
                elif cls == 1:
                    layout = ("contiguous", self._addr(p + 2), self._u(p + 2 + self.O, self.L))
                elif cls == 2:
                    nd = b[p + 2]
                    dims = tuple(self._u(p + 3 + self.O + 4 * i, 4) for i in range(nd))
                    layout = ("chunked", self._addr(p + 3), dims)
                else:
                    raise ValueError("data layout class %d" % cls)
            elif mtype == 0x0B:                                  # filter pipeline
                ver, nf = b[p], b[p + 1]
                q = p + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid = self._u(q, 2)
                    if ver == 1 or fid >= 256:
                        nlen = self._u(q + 2, 2); q += 4
                    else:
                        nlen = 0; q += 2
                    ncd = self._u(q + 2, 2)
                    q += 4 + (((nlen + 7) & ~7) if ver == 1 else nlen)
                    cd = [self._u(q + 4 * i, 4) for i in range(ncd)]
                    q += 4 * ncd + (4 if ver == 1 and ncd & 1 else 0)
                    filters.append((fid, cd))
        if shape is None or dtype is None or layout is None:
            raise ValueError("not a dataset (dataspace / datatype / layout message missing)")
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if layout[0] == "compact":
            return np.frombuffer(self.b, dtype, count, layout[1]).reshape(shape).copy()
        if layout[0] == "contiguous":
            if layout[1] is None:                                # never written: the fill value (0) everywhere
                return np.zeros(shape, dtype)
            return np.frombuffer(self.b, dtype, count, self.base + layout[1]).reshape(shape).copy()
        btree, cdims = layout[1], layout[2]
        if cdims[-1] != dtype.itemsize or len(cdims) != len(shape) + 1:
            raise ValueError("chunk dimensions do not match the dataset")
        cshape = cdims[:-1]
        out = np.zeros(shape, dtype)
        if btree is None:
            return out
        nd = len(cdims)

        def unfilter(raw, mask):
            for i in range(len(filters) - 1, -1, -1):            # the pipeline is applied in order on write
                if mask & (1 << i):
                    continue
                fid, cd = filters[i]
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:                                   # shuffle: byte planes back into elements
                    es = cd[0] if cd else dtype.itemsize
                    a = np.frombuffer(raw, np.uint8)
                    n = len(a) // es
                    raw = a[:n * es].reshape(es, n).T.tobytes() + a[n * es:].tobytes()
                elif fid == 3:                                   # fletcher32: the checksum trails the data
                    raw = raw[:-4]
                else:
                    raise ValueError("HDF5 filter %d (only deflate, shuffle, fletcher32 are read here)" % fid)
            return raw

        def walk(node):
            p = self.base + node
            if self.b[p:p + 4] != b"TREE" or self.b[p + 4] != 1:
                raise ValueError("bad chunk B-tree node")
            level, used = self.b[p + 5], self._u(p + 6, 2)
            q = p + 8 + 2 * self.O
            ksz = 8 + 8 * nd
            for i in range(used):
                k = q + i * (ksz + self.O)
                child = self._addr(k + ksz)
                if level:
                    walk(child)
                    continue
                nbytes, mask = self._u(k, 4), self._u(k + 4, 4)
                off = [self._u(k + 8 + 8 * d, 8) for d in range(nd - 1)]
                raw = unfilter(bytes(self.b[self.base + child:self.base + child + nbytes]), mask)
                chunk = np.frombuffer(raw, dtype, int(np.prod(cshape))).reshape(cshape)
                sel_o = tuple(slice(o, min(o + c, s)) for o, c, s in zip(off, cshape, shape))
                sel_c = tuple(slice(0, s.stop - s.start) for s in sel_o)
                out[sel_o] = chunk[sel_c]
        walk(btree)
        return out


def read_datasets(h5_filename, names):
    """The named top-level datasets of an HDF5 file as numpy arrays (native byte order)."""
    with open(h5_filename, "rb") as f:
        h = _H5(f.read())
    kids = h.children(h.root)
    out = []
    for n in names:
        if n not in kids:
            raise KeyError("%s: no dataset %r (has: %s)" % (h5_filename, n, ", ".join(sorted(kids))))
        a = h.dataset(kids[n])
        out.append(a.astype(a.dtype.newbyteorder("=")))
    return out


def load_h5(h5_filename):
    """``(data [blocks, 4096, 9] float32, label [blocks, 4096] uint8)`` of one block file (S3DISDataLoader.py:19-23)."""
    data, label = read_datasets(h5_filename, ("data", "label"))
    return data, label


def get_data_files(list_filename):
    """S3DISDataLoader.py:15-16."""
    return [line.rstrip() for line in open(list_filename)]


def recognize_all_data(root, test_area=5):
    """``(train_data, train_label, test_data, test_label)``: all block files of ``root/all_files.txt`` (only the file name of
    each line is used, as in the reference), blocks of rooms whose name contains ``Area_<test_area>`` held out
    (S3DISDataLoader.py:29-57)."""
    all_files = get_data_files(os.path.join(root, "all_files.txt"))
    room_filelist = get_data_files(os.path.join(root, "room_filelist.txt"))
    data, label = [], []
    for name in all_files:
        d, l = load_h5(os.path.join(root, name.split("/")[-1]))
        data.append(d)
        label.append(l)
    data, label = np.concatenate(data, 0), np.concatenate(label, 0)
    tag = "Area_" + str(test_area)
    test = np.array([tag in room for room in room_filelist], bool)
    if len(test) != len(data):
        raise ValueError("room_filelist.txt names %d blocks, the files hold %d" % (len(test), len(data)))
    return data[~test], label[~test], data[test], label[test]


class S3DISDataLoader:
    """Map-style dataset over the blocks (S3DISDataLoader.py:59-77); usable with ``torch.utils.data.DataLoader``."""

    def __init__(self, data, labels, data_augmentation=False):
        self.data = data
        self.labels = labels
        self.data_augmentation = data_augmentation

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        pointcloud = self.data[index]
        label = self.labels[index]
        if self.data_augmentation:                               # jitter_point_cloud(sigma=0.01, clip=0.05), augmentation.py
            noise = np.clip(0.01 * np.random.randn(*pointcloud.shape), -0.05, 0.05)
            pointcloud = (pointcloud + noise).astype(np.float32)
        return pointcloud, label
