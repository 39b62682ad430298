"""``gs::Gaussians`` PLY I/O over libgsx (csrc/gsx_ply.cpp): the reference's load path
``read_ply_header -> header.count() -> read_ply_gaussians -> Gaussian::from`` (src/app.rs:1053-1096) and
``write_ply(writer, edits, mask)`` (src/app.rs:897-947)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .scene import GAUSSIAN_DTYPE


class PlyHeader:
    def __init__(self, raw: _lib.PlyHeader):
        self.raw = raw

    def count(self) -> int:
        """``PlyHeader::count`` (src/app.rs:1057)."""
        return int(self.raw.count)


class Gaussians:
    """``gs::Gaussians {gaussians: Vec<Gaussian>}``."""

    def __init__(self, gaussians: np.ndarray):
        self.gaussians = np.ascontiguousarray(gaussians, dtype=GAUSSIAN_DTYPE)

    @staticmethod
    def read_ply_header(data: bytes) -> PlyHeader:
        h = _lib.PlyHeader()
        buf = np.frombuffer(data, dtype=np.uint8)  # zero-copy view (bytes, bytearray, memoryview, ndarray): a garden-sized file is 1.4 GB
        _lib.check(_lib.load().gsx_ply_read_header(buf.ctypes.data, buf.size, C.byref(h)))
        return PlyHeader(h)

    @staticmethod
    def read_ply_gaussians(data: bytes, header: PlyHeader, start: int = 0, count: int | None = None, batch: int = 1 << 16):
        """Iterator of ``gs::Gaussian`` batches (the app streams them to the GPU in batches, scene.rs:341-380)."""
        n = header.count() - start if count is None else count
        buf = np.frombuffer(data, dtype=np.uint8)
        L = _lib.load()
        for s in range(start, start + n, batch):
            m = min(batch, start + n - s)
            out = np.empty(m, dtype=GAUSSIAN_DTYPE)  # every byte is written by the library (no padding in gs::Gaussian)
            _lib.check(L.gsx_ply_read_gaussians(buf.ctypes.data, buf.size, C.byref(header.raw), s, m, out.ctypes.data))
            yield out

    @classmethod
    def read_ply(cls, data: bytes) -> "Gaussians":
        """The whole file in ONE library call: a large range is converted on several host threads (csrc/gsx_ply.cpp)."""
        h = cls.read_ply_header(data)
        parts = list(cls.read_ply_gaussians(data, h, batch=max(h.count(), 1)))
        return cls(parts[0] if parts else np.zeros(0, GAUSSIAN_DTYPE))

    def write_ply(self, mask_words: np.ndarray | None = None, edits: np.ndarray | None = None) -> bytes:
        """``write_ply(writer, edits, mask)`` (app.rs:908-940): binary little-endian INRIA PLY of the (masked, edited) Gaussians."""
        return self.write_ply_array(mask_words, edits).tobytes()

    def write_ply_array(self, mask_words: np.ndarray | None = None, edits: np.ndarray | None = None) -> np.ndarray:
        """``write_ply`` into a uint8 array (no second copy of a multi-GB file)."""
        L = _lib.load()
        g = self.gaussians
        size = C.c_uint64()
        mw = None if mask_words is None else np.ascontiguousarray(mask_words, np.uint32)
        mp = None if mw is None else mw.ctypes.data_as(C.POINTER(C.c_uint32))
        ep = None
        if edits is not None:
            from .query import EDIT_DTYPE

            ed = np.ascontiguousarray(edits, EDIT_DTYPE)
            if ed.shape[0] != g.shape[0]:
                raise ValueError("one edit record per Gaussian")
            ep = ed.ctypes.data
        _lib.check(L.gsx_ply_write(g.ctypes.data, g.shape[0], mp, ep, None, 0, C.byref(size)))
        out = np.empty(size.value, np.uint8)
        _lib.check(L.gsx_ply_write(g.ctypes.data, g.shape[0], mp, ep, out.ctypes.data, out.size, C.byref(size)))
        return out
