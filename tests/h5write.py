"""Minimal HDF5 writer (test infrastructure): superblock v0, one old-style root group (B-tree + local heap +
one symbol node), version-1 object headers, contiguous little-endian float32 datasets -- the subset
h5py's `create_dataset(name, shape, np.float32)` produces for the reference's dataset files
(data_generation.py:64-74).  h5py is not installed here, so dataset fixtures for the reader
(`formats.PaddedDataset`) are generated with this at test time; the reader itself is pinned on the real
h5py-written `weights.h5` of the reference (tests/test_host_logic.py)."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype: int, body: bytes) -> bytes:
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def _dataset_header(shape, data_addr, nbytes) -> bytes:
    space = struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)
    # IEEE float32 little-endian: class 1 version 1; bit field (0x20, 0x1f, 0); size 4; properties
    dtype = struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    layout = struct.pack("<BBQQ", 3, 1, data_addr, nbytes)
    msgs = _msg(0x0001, space) + _msg(0x0003, dtype) + _msg(0x0008, layout)
    return struct.pack("<BxHII4x", 1, 3, 1, len(msgs)) + msgs


class _Buf:
    def __init__(self):
        self.b = bytearray(96)                               # superblock (56) + root symbol table entry (40)

    def alloc(self, n: int) -> int:
        self.b += b"\0" * (-len(self.b) % 8)
        o = len(self.b)
        self.b += b"\0" * n
        return o

    def put(self, o: int, data: bytes):
        self.b[o:o + len(data)] = data


def _write_group(w: _Buf, children: dict):
    """-> (object header address, btree address, heap address) of an old-style group."""
    names = sorted(children)
    if len(names) > 8:
        raise ValueError("one symbol node per group: at most 8 members")
    heap_data, name_off = bytearray(b"\0" * 8), {}
    for n in names:
        name_off[n] = len(heap_data)
        heap_data += _pad8(n.encode("ascii") + b"\0")
    heap_data += b"\0" * 16
    struct.pack_into("<QQ", heap_data, len(heap_data) - 16, 1, 16)       # free block: next = 1 (none), size
    entries = b""
    for n in names:
        c = children[n]
        if isinstance(c, dict):
            hdr, bt, hp = _write_group(w, c)
            entries += struct.pack("<QQII", name_off[n], hdr, 1, 0) + struct.pack("<QQ", bt, hp)
        else:
            a = np.ascontiguousarray(c, "<f4")
            data = w.alloc(a.nbytes)
            w.put(data, a.tobytes())
            h = _dataset_header(a.shape, data, a.nbytes)
            hdr = w.alloc(len(h))
            w.put(hdr, h)
            entries += struct.pack("<QQII16x", name_off[n], hdr, 0, 0)
    snod = w.alloc(8 + 8 * 40)
    w.put(snod, b"SNOD" + struct.pack("<BxH", 1, len(names)) + entries)
    heap_data_addr = w.alloc(len(heap_data))
    w.put(heap_data_addr, bytes(heap_data))
    heap = w.alloc(32)
    w.put(heap, b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), len(heap_data) - 16, heap_data_addr))
    btree = w.alloc(24 + (2 * 16 + 1) * 8 + 2 * 16 * 8)
    last = name_off[names[-1]] if names else 0
    w.put(btree, b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod, last))
    msgs = _msg(0x0011, struct.pack("<QQ", btree, heap))
    hdr = w.alloc(16 + len(msgs))
    w.put(hdr, struct.pack("<BxHII4x", 1, 1, 1, len(msgs)) + msgs)
    return hdr, btree, heap


def write_h5(path: str, tree: dict) -> None:
    """tree: name -> float32 array, or name -> nested dict (a group).  Names are stored sorted, as the
    library keeps them."""
    w = _Buf()
    hdr, btree, heap = _write_group(w, tree)
    eof = len(w.b) + (-len(w.b) % 8)
    w.b += b"\0" * (eof - len(w.b))
    w.put(0, b"\x89HDF\r\n\x1a\n")
    w.put(8, struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0))
    w.put(16, struct.pack("<HHI", 4, 16, 0))
    w.put(24, struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF))
    w.put(56, struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", btree, heap))
    with open(path, "wb") as f:
        f.write(w.b)


def write_keras_dense(path: str, weights) -> None:
    """The group layout of a Keras `save_weights` / `model.save` HDF5 file for a Dense stack:
    /dense[_k]/dense[_k]/{kernel:0, bias:0}."""
    tree = {}
    for k, (W, b) in enumerate(weights):
        name = "dense" if k == 0 else f"dense_{k}"
        tree[name] = {name: {"kernel:0": W, "bias:0": b}}
    write_h5(path, {"model_weights": tree})
