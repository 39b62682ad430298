"""Selection, edit and query pods of the viewer — host-side mirror of the types the app builds and hands to the crate.

``GaussianEditFlag`` / ``GaussianEditPod`` (src/app.rs:1546-1564), ``QuerySelectionOp`` (src/tab/scene.rs:1605),
``QueryNonePod`` / ``QueryHitPod`` (scene.rs:1621-1637), ``QueryToolset`` with its Rect and Brush tools
(scene.rs:766-791, 1258-1264), ``gs::query::hit_pos_by_closest`` / ``hit_pos_by_alpha_range`` (scene.rs:660-679).
The arithmetic behind them is spec/RENDER_SPEC.md §7 [BUILD-SPEC] and runs in libgsx.so.
"""
from __future__ import annotations

import ctypes as C
import enum

import numpy as np

from . import _lib

EDIT_DTYPE = np.dtype([("flag", "<u4"), ("color", "<f4", (3,)), ("contrast", "<f4"), ("exposure", "<f4"),
                       ("gamma", "<f4"), ("alpha", "<f4")])
HIT_DTYPE = np.dtype([("index", "<u4"), ("depth", "<f4"), ("alpha", "<f4"), ("reserved", "<u4")])
assert EDIT_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 16


class GaussianEditFlag(enum.IntFlag):
    ENABLED = 1
    HIDDEN = 2
    OVERRIDE_COLOR = 4


class QuerySelectionOp(enum.IntEnum):
    Set = 0
    Add = 1
    Remove = 2


class QueryKind(enum.IntEnum):
    None_ = 0
    Hit = 1
    Rect = 2
    Brush = 3
    Texture = 4


class QueryToolsetTool(enum.IntEnum):
    Rect = 0
    Brush = 1


class GaussianEditPod:
    """``gs::GaussianEditPod::new(flag, color, contrast, exposure, gamma, alpha)``; ``default()`` is the identity."""

    def __init__(self, flag=0, color=(0.0, 1.0, 1.0), contrast=0.0, exposure=0.0, gamma=1.0, alpha=1.0):
        self.flag, self.color = int(flag), tuple(float(c) for c in color)
        self.contrast, self.exposure, self.gamma, self.alpha = float(contrast), float(exposure), float(gamma), float(alpha)

    @classmethod
    def default(cls) -> "GaussianEditPod":
        return cls()

    def raw(self) -> _lib.GaussianEdit:
        return _lib.GaussianEdit(self.flag, (C.c_float * 3)(*self.color), self.contrast, self.exposure, self.gamma, self.alpha)

    def record(self) -> np.ndarray:
        r = np.zeros((), EDIT_DTYPE)
        r["flag"], r["color"] = self.flag, self.color
        r["contrast"], r["exposure"], r["gamma"], r["alpha"] = self.contrast, self.exposure, self.gamma, self.alpha
        return r


def default_edits(n: int) -> np.ndarray:
    e = np.zeros(n, EDIT_DTYPE)
    e["color"] = (0.0, 1.0, 1.0)
    e["gamma"] = 1.0
    e["alpha"] = 1.0
    return e


class QueryPod:
    """What ``viewer.update_query`` takes: ``QueryNonePod::new().as_query()``, ``QueryHitPod::new(coords).as_query()``
    or the toolset's rect / brush / texture query."""

    def __init__(self, kind=QueryKind.None_, op=QuerySelectionOp.Set, p0=(0.0, 0.0), p1=(0.0, 0.0), radius=0.0):
        self.kind, self.op = QueryKind(kind), QuerySelectionOp(op)
        self.p0, self.p1, self.radius = (float(p0[0]), float(p0[1])), (float(p1[0]), float(p1[1])), float(radius)

    @classmethod
    def none(cls):
        return cls()

    @classmethod
    def hit(cls, coords):
        return cls(QueryKind.Hit, p0=coords)

    @classmethod
    def rect(cls, top_left, bottom_right, op=QuerySelectionOp.Set):
        return cls(QueryKind.Rect, op, top_left, bottom_right)

    @classmethod
    def brush(cls, start, end, radius, op=QuerySelectionOp.Set):
        return cls(QueryKind.Brush, op, start, end, radius)

    @classmethod
    def texture(cls, op=QuerySelectionOp.Set):
        return cls(QueryKind.Texture, op)

    def raw(self) -> _lib.Query:
        return _lib.Query(int(self.kind), int(self.op), (C.c_float * 2)(*self.p0), (C.c_float * 2)(*self.p1), self.radius, 0)


class QueryToolset:
    """``gs::QueryToolset``: turns pointer positions into per-frame queries (scene.rs:766-791).

    Immediate mode (``set_use_texture(False)``): every frame carries a live Rect / Brush-segment query.  Texture mode
    (the app's default, ``immediate = false``, app.rs:1452): the strokes are painted into a viewport-sized 8-bit texture
    — a filled rectangle for Rect, discs along the pointer path for Brush — and ``end()`` yields ONE texture query."""

    def __init__(self, size=(1, 1)):
        self.size = (int(size[0]), int(size[1]))
        self.use_texture = True
        self.brush_radius = 40.0
        self._tool = None
        self._op = QuerySelectionOp.Set
        self._start = self._pos = self._prev = (0.0, 0.0)
        self._ended = False
        self.texture = np.zeros((self.size[1], self.size[0]), np.uint8)

    def set_use_texture(self, on: bool) -> None:
        self.use_texture = bool(on)

    def update_brush_radius(self, r) -> None:
        self.brush_radius = float(r)

    def resize(self, size) -> None:
        self.size = (int(size[0]), int(size[1]))
        self.texture = np.zeros((self.size[1], self.size[0]), np.uint8)

    def start(self, tool: QueryToolsetTool, op: QuerySelectionOp, pos) -> None:
        self._tool, self._op = QueryToolsetTool(tool), QuerySelectionOp(op)
        self._start = self._pos = self._prev = (float(pos[0]), float(pos[1]))
        self._ended = False
        self.texture[:] = 0
        self._paint()

    def update_pos(self, pos) -> None:
        if self._tool is None:
            return
        self._prev, self._pos = self._pos, (float(pos[0]), float(pos[1]))
        self._paint()

    def end(self) -> None:
        self._ended = self._tool is not None

    def state(self):
        return None if self._tool is None else (self._tool, self._op, self._start, self._pos)

    def _paint(self) -> None:
        if not self.use_texture:
            return
        w, h = self.size
        if self._tool == QueryToolsetTool.Rect:
            self.texture[:] = 0
            x0, x1 = sorted((self._start[0], self._pos[0]))
            y0, y1 = sorted((self._start[1], self._pos[1]))
            # rasterisation rule: texel (i, j) is selected iff its centre (i + 0.5, j + 0.5) lies inside the rectangle
            i0, i1 = max(int(np.ceil(x0 - 0.5)), 0), min(int(np.floor(x1 - 0.5)), w - 1)
            j0, j1 = max(int(np.ceil(y0 - 0.5)), 0), min(int(np.floor(y1 - 0.5)), h - 1)
            if i0 <= i1 and j0 <= j1:
                self.texture[j0:j1 + 1, i0:i1 + 1] = 255
        else:
            r = self.brush_radius
            ax, ay = self._prev
            bx, by = self._pos
            j0, j1 = max(int(min(ay, by) - r) - 1, 0), min(int(max(ay, by) + r) + 2, h)
            i0, i1 = max(int(min(ax, bx) - r) - 1, 0), min(int(max(ax, bx) + r) + 2, w)
            if i0 >= i1 or j0 >= j1:
                return
            yy, xx = np.mgrid[j0:j1, i0:i1]
            px, py = xx + 0.5, yy + 0.5
            dx, dy = bx - ax, by - ay
            len2 = dx * dx + dy * dy
            t = np.clip(((px - ax) * dx + (py - ay) * dy) / len2, 0.0, 1.0) if len2 > 0 else 0.0
            d2 = (px - (ax + t * dx)) ** 2 + (py - (ay + t * dy)) ** 2
            self.texture[j0:j1, i0:i1][d2 <= r * r] = 255

    def query(self) -> QueryPod:
        """The query of this frame; after ``end()`` in texture mode, the one texture query, then None queries."""
        if self._tool is None:
            return QueryPod.none()
        if self.use_texture:
            if self._ended:
                self._tool = None
                return QueryPod.texture(self._op)
            return QueryPod.none()
        if self._ended:
            self._tool = None
            return QueryPod.none()
        if self._tool == QueryToolsetTool.Rect:
            return QueryPod.rect(self._start, self._pos, self._op)
        return QueryPod.brush(self._prev, self._pos, self.brush_radius, self._op)


def _cam_args(camera, size):
    w, h = int(size[0]), int(size[1])
    v = np.ascontiguousarray(camera.view(), np.float32).reshape(16)
    p = np.ascontiguousarray(camera.projection(w / h), np.float32).reshape(16)
    return v, p, w, h


def hit_pos_by_closest(coords, results: np.ndarray, camera, viewer_size):
    """``gs::query::hit_pos_by_closest(&pod, &results, &camera, viewer_size)`` -> (index, pos) or None."""
    L = _lib.load()
    r = np.ascontiguousarray(results, HIT_DTYPE)
    if r.size == 0:
        return None
    v, p, w, h = _cam_args(camera, viewer_size)
    c = np.ascontiguousarray(coords, np.float32).reshape(2)
    idx, pos = C.c_uint32(), np.zeros(3, np.float32)
    f32p = C.POINTER(C.c_float)
    _lib.check(L.gsx_query_hit_pos_by_closest(r.ctypes.data, r.size, v.ctypes.data_as(f32p), p.ctypes.data_as(f32p), w, h,
                                              c.ctypes.data_as(f32p), C.byref(idx), pos.ctypes.data_as(f32p)))
    return int(idx.value), pos


def hit_pos_by_alpha_range(coords, results: np.ndarray, camera, viewer_size, alpha_range: float = 0.05):
    """``gs::query::hit_pos_by_alpha_range(&pod, &mut results, &camera, viewer_size, 0.05)`` -> (index, alpha, pos) or None."""
    L = _lib.load()
    r = np.ascontiguousarray(results, HIT_DTYPE)
    if r.size == 0:
        return None
    v, p, w, h = _cam_args(camera, viewer_size)
    c = np.ascontiguousarray(coords, np.float32).reshape(2)
    idx, alpha, pos = C.c_uint32(), C.c_float(), np.zeros(3, np.float32)
    f32p = C.POINTER(C.c_float)
    _lib.check(L.gsx_query_hit_pos_by_alpha_range(r.ctypes.data, r.size, v.ctypes.data_as(f32p), p.ctypes.data_as(f32p), w, h,
                                                  c.ctypes.data_as(f32p), float(alpha_range), C.byref(idx), C.byref(alpha),
                                                  pos.ctypes.data_as(f32p)))
    return int(idx.value), float(alpha.value), pos
