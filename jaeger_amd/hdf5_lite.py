"""A small read-only HDF5 reader for Keras weight files (no h5py in the MI355X image).

Covers what ``model.save_weights`` / ``h5py`` write with default settings: superblock v0/v1,
old-style groups (symbol-table message -> v1 B-tree + local heap + SNOD nodes), version-1 object
headers with continuation blocks, simple dataspaces, fixed-point / IEEE-float datatypes, and
compact or contiguous dataset layouts.  Chunked / filtered datasets, v2 object headers ("OHDR") and
link messages are reported as unsupported rather than guessed at.

    with open(path, "rb") as fh: data = fh.read()
    f = H5File(data); f.datasets() -> {"/aa/aa/embeddings:0": np.ndarray, ...}

Format reference: the HDF5 File Format Specification v1/v2 (public); verified in the build
container against ``h5dump`` on the reference's ``WRes_1024.h5`` (tests/golden/make_golden.py).
"""

from __future__ import annotations

import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"


class H5Unsupported(ValueError):
    pass


class H5File:
    def __init__(self, data: bytes):
        self.d = memoryview(data)
        base = data.find(_SIG)
        if base != 0:
            raise H5Unsupported("not an HDF5 file (or has a user block)")
        ver = data[8]
        if ver not in (0, 1):
            raise H5Unsupported(f"superblock version {ver} (only 0/1: files written with libver='earliest')")
        self.O, self.L = data[13], data[14]          # size of offsets / lengths
        p = 24 if ver == 0 else 28                    # after group K values + flags (+ v1 indexed-storage K)
        p += 4 * self.O                               # base, free-space, end-of-file, driver addresses
        # root group symbol-table entry
        self.root = self._symbol_entry(p)

    # -- primitives ----------------------------------------------------------------------------
    def _u(self, off: int, n: int) -> int:
        return int.from_bytes(self.d[off:off + n], "little")

    def _symbol_entry(self, p: int) -> dict:
        O = self.O
        e = {"name_off": self._u(p, O), "header": self._u(p + O, O), "cache": self._u(p + 2 * O, 4)}
        if e["cache"] == 1:
            e["btree"], e["heap"] = self._u(p + 2 * O + 8, O), self._u(p + 3 * O + 8, O)
        return e

    def _messages(self, addr: int):
        """Yield (type, payload memoryview) of a version-1 object header incl. continuations."""
        if bytes(self.d[addr:addr + 4]) == b"OHDR":
            raise H5Unsupported("version-2 object header (file written with libver='latest')")
        if self.d[addr] != 1:
            raise H5Unsupported(f"object header version {self.d[addr]}")
        n_msgs, size = self._u(addr + 2, 2), self._u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        seen = 0
        while blocks and seen < n_msgs:
            p, remaining = blocks.pop(0)
            end = p + remaining
            while p + 8 <= end and seen < n_msgs:
                mtype, msize = self._u(p, 2), self._u(p + 2, 2)
                body = self.d[p + 8:p + 8 + msize]
                seen += 1
                if mtype == 0x10:                    # continuation
                    blocks.append((self._u(p + 8, self.O), self._u(p + 8 + self.O, self.L)))
                else:
                    yield mtype, body
                p += 8 + msize

    def _heap_name(self, heap: int, off: int) -> str:
        assert bytes(self.d[heap:heap + 4]) == b"HEAP"
        seg = self._u(heap + 8 + 2 * self.L, self.O)
        end = seg + off
        while self.d[end] != 0:
            end += 1
        return bytes(self.d[seg + off:end]).decode()

    def _group_entries(self, btree: int, heap: int):
        assert bytes(self.d[btree:btree + 4]) == b"TREE", "group B-tree signature"
        level, used = self.d[btree + 5], self._u(btree + 6, 2)
        p = btree + 8 + 2 * self.O
        for k in range(used):
            child = self._u(p + self.L, self.O)      # key k, child k, key k+1, ...
            p += self.L + self.O
            if level > 0:
                yield from self._group_entries(child, heap)
            else:
                assert bytes(self.d[child:child + 4]) == b"SNOD", "symbol-table node signature"
                n = self._u(child + 6, 2)
                q = child + 8
                for _ in range(n):
                    e = self._symbol_entry(q)
                    yield self._heap_name(heap, e["name_off"]), e
                    q += 2 * self.O + 24

    # -- objects ----------------------------------------------------------------------------------
    def _read_object(self, addr: int):
        """-> ("group", btree, heap) or ("dataset", ndarray)."""
        shape = dtype = None
        layout = None
        for mtype, body in self._messages(addr):
            if mtype == 0x11:
                return "group", self._u_mv(body, 0, self.O), self._u_mv(body, self.O, self.O)
            if mtype == 0x01:
                ver, rank, flags = body[0], body[1], body[2]
                p = 8 if ver == 1 else 4
                shape = tuple(self._u_mv(body, p + i * self.L, self.L) for i in range(rank))
            elif mtype == 0x03:
                cls, bits0, size = body[0] & 0x0F, body[1], self._u_mv(body, 4, 4)
                if bits0 & 1:
                    raise H5Unsupported("big-endian datatype")
                if cls == 1 and size in (2, 4, 8):
                    dtype = np.dtype(f"<f{size}")
                elif cls == 0 and size in (1, 2, 4, 8):
                    dtype = np.dtype(f"<{'i' if bits0 & 8 else 'u'}{size}")
                else:
                    dtype = None                      # strings etc.: not needed for weights
            elif mtype == 0x08:
                ver = body[0]
                if ver == 3:
                    lclass = body[1]
                    if lclass == 1:
                        layout = ("contiguous", self._u_mv(body, 2, self.O), self._u_mv(body, 2 + self.O, self.L))
                    elif lclass == 0:
                        n = self._u_mv(body, 2, 2)
                        layout = ("compact", bytes(body[4:4 + n]))
                    else:
                        layout = ("chunked",)
                else:
                    rank, lclass = body[1], body[2]
                    if lclass == 1:
                        layout = ("contiguous", self._u_mv(body, 8, self.O), None)
                    else:
                        layout = ("chunked",)
        if shape is None or layout is None:
            return "other", None
        if dtype is None:
            return "other", None
        if layout[0] == "chunked":
            raise H5Unsupported("chunked / filtered dataset")
        n = int(np.prod(shape)) if shape else 1
        if layout[0] == "compact":
            arr = np.frombuffer(layout[1], dtype, n)
        else:
            off = layout[1]
            if off == (1 << (8 * self.O)) - 1:        # undefined address: never written
                arr = np.zeros(n, dtype)
            else:
                arr = np.frombuffer(self.d[off:off + n * dtype.itemsize], dtype, n)
        return "dataset", arr.reshape(shape).copy()

    @staticmethod
    def _u_mv(mv, off: int, n: int) -> int:
        return int.from_bytes(mv[off:off + n], "little")

    def datasets(self) -> dict[str, np.ndarray]:
        """All numeric datasets, keyed by absolute path."""
        out: dict[str, np.ndarray] = {}

        def walk(prefix: str, btree: int, heap: int):
            for name, e in self._group_entries(btree, heap):
                path = f"{prefix}/{name}"
                if e["cache"] == 1:
                    walk(path, e["btree"], e["heap"])
                    continue
                kind, *rest = self._read_object(e["header"])
                if kind == "group":
                    walk(path, rest[0], rest[1])
                elif kind == "dataset":
                    out[path] = rest[0]

        if self.root["cache"] == 1:
            walk("", self.root["btree"], self.root["heap"])
        else:
            kind, *rest = self._read_object(self.root["header"])
            if kind != "group":
                raise H5Unsupported("root object is not an old-style group")
            walk("", rest[0], rest[1])
        return out


def read_datasets(path) -> dict[str, np.ndarray]:
    with open(path, "rb") as fh:
        return H5File(fh.read()).datasets()
