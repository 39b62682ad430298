"""Mask shapes, the app's mask-operation grammar, and ``gs::MaskEvaluator`` over libgsx.

Restates the host-side pieces of the reference that feed K5:
  * ``GaussianSplattingMaskOp::parse`` — src/app.rs:1660-1783: operands are shape indices, operators by
    decreasing precedence ``!`` (complement) > ``^`` (symmetric difference) > ``-`` (difference) > ``&``
    (intersection) > ``|`` (union), all binary operators left-associative, parentheses, free whitespace;
    an empty string parses to ``None`` (the app then evaluates ``MaskOpTree::Reset``, scene.rs:2204-2205);
  * ``validate_shapes`` — src/app.rs:1786-1813: first out-of-range shape index is the error;
  * ``to_tree`` — src/app.rs:1816-1837, here flattened to the postfix program ``gsx_mask_evaluate`` takes.
The evaluation itself runs in a HIP kernel (csrc/kernels_mask.hip).
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field

import numpy as np

from . import _lib


class MaskShapeKind(enum.IntEnum):
    """``gs::MaskShapeKind`` (src/app.rs:1634)."""

    Box = 0
    Ellipsoid = 1


@dataclass
class MaskShape:
    """``gs::MaskShape {kind, pos, rotation, scale, color}`` (src/app.rs:1620-1640; colour is gizmo-only)."""

    kind: MaskShapeKind = MaskShapeKind.Box
    pos: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    rotation: np.ndarray = field(default_factory=lambda: np.array([0, 0, 0, 1], np.float32))  # quaternion x,y,z,w
    scale: np.ndarray = field(default_factory=lambda: np.ones(3, np.float32))


class MaskOpError(ValueError):
    pass


# --- syntax tree: nested tuples ("shape", i) | ("not", a) | (op, a, b) with op in | & - ^ ---
class MaskOp:
    """``GaussianSplattingMaskOp`` (src/app.rs:1634-1658)."""

    def __init__(self, tree):
        self.tree = tree

    @staticmethod
    def parse(text: str):
        """``GaussianSplattingMaskOp::parse`` (src/app.rs:1660-1783). Returns ``None`` for an empty string."""
        src = text.strip()
        if not src:
            return None
        pos = 0

        def ws():
            nonlocal pos
            while pos < len(src) and src[pos] == " ":  # nom `space0`: spaces and tabs
                pos += 1
            while pos < len(src) and src[pos] in " \t":
                pos += 1

        def factor():
            nonlocal pos
            ws()
            if pos < len(src) and src[pos].isdigit():
                start = pos
                while pos < len(src) and src[pos].isdigit():
                    pos += 1
                node = ("shape", int(src[start:pos]))
            elif pos < len(src) and src[pos] == "(":
                pos += 1
                node = union()
                ws()
                if pos >= len(src) or src[pos] != ")":
                    raise MaskOpError(f"Failed to parse mask operation: expected ')' at {pos}")
                pos += 1
            elif pos < len(src) and src[pos] == "!":
                pos += 1
                node = ("not", factor())
            else:
                raise MaskOpError(f"Failed to parse mask operation: unexpected input at {pos}")
            ws()
            return node

        def chain(sub, ch):
            def level():
                nonlocal pos
                node = sub()
                while True:
                    ws()
                    if pos < len(src) and src[pos] == ch:
                        pos += 1
                        node = (ch, node, sub())  # left fold, as `iter.fold(initial, ...)` in the reference
                    else:
                        return node
            return level

        sym = chain(factor, "^")
        diff = chain(sym, "-")
        inter = chain(diff, "&")
        union = chain(inter, "|")
        node = union()
        ws()
        if pos != len(src):
            raise MaskOpError(f"Failed to parse mask operation: trailing input at {pos}")
        return MaskOp(node)

    def validate_shapes(self, shape_count: int):
        """``validate_shapes`` (src/app.rs:1786-1813): returns the first offending index or ``None``."""

        def walk(t):
            if t[0] == "shape":
                return t[1] if t[1] >= shape_count else None
            for c in t[1:]:
                r = walk(c)
                if r is not None:
                    return r
            return None

        return walk(self.tree)

    def to_postfix(self):
        code = {"|": 1, "&": 2, "-": 3, "^": 4}
        out = []

        def walk(t):
            if t[0] == "shape":
                out.append((0, t[1]))
            elif t[0] == "not":
                walk(t[1])
                out.append((5, 0))
            else:
                walk(t[1])
                walk(t[2])
                out.append((code[t[0]], 0))

        walk(self.tree)
        return out


class _GsxMaskShape(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("pos", C.c_float * 3), ("quat", C.c_float * 4), ("scale", C.c_float * 3)]


class _GsxMaskOp(C.Structure):
    _fields_ = [("opcode", C.c_uint32), ("arg", C.c_uint32)]


def pack_program(op: MaskOp | None, shapes):
    ops = op.to_postfix() if op is not None else []
    c_ops = (_GsxMaskOp * max(len(ops), 1))(*[_GsxMaskOp(o, a) for o, a in ops])
    c_shapes = (_GsxMaskShape * max(len(shapes), 1))()
    for i, s in enumerate(shapes):
        c_shapes[i].kind = int(s.kind)
        c_shapes[i].pos[:] = [float(x) for x in np.asarray(s.pos, np.float32)]
        c_shapes[i].quat[:] = [float(x) for x in np.asarray(s.rotation, np.float32)]
        c_shapes[i].scale[:] = [float(x) for x in np.asarray(s.scale, np.float32)]
    return c_ops, len(ops), c_shapes, len(shapes)


class MaskEvaluator:
    """``gs::MaskEvaluator`` (src/tab/scene.rs:2034, 2124-2131, 2201-2209)."""

    def __init__(self, viewer):
        self._v = viewer

    def evaluate(self, op: MaskOp | None, key: str, shapes=()) -> None:
        """``evaluate(device, queue, &tree, mask_buffer, model_transform_buffer, gaussians_buffer)``;
        ``op is None`` is ``MaskOpTree::Reset``.  Raises ``GsxError`` for an out-of-range shape index."""
        c_ops, n_ops, c_shapes, n_shapes = pack_program(op, list(shapes))
        _lib.check(self._v._L.gsx_mask_evaluate(self._v._h, key.encode(), c_ops, n_ops, c_shapes, n_shapes))
