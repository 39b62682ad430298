"""Host-side mirror of the ``wgpu-3dgs-viewer`` (``gs::``) surface the app uses for the render path.

The reference's host language is Rust (no toolchain in this image), so this mirror over the C ABI is
Python; names, argument order and error behaviour follow the reference call sites so tests read like
the app's frame loop (src/tab/scene.rs:699-874, 2263-2326):

    viewer = MultiModelViewer(size=(1, 1))                      # MultiModelViewer::new_with   scene.rs:1969
    viewer.add_model("a", count)                                # new_empty + BindGroups::new   scene.rs:2111
    viewer.models["a"].gaussian_buffers.gaussians_buffer.update_range(start, gaussians)  # scene.rs:2083
    viewer.update_camera(camera, (w, h))                        # scene.rs:795
    viewer.update_model_transform("a", pos, quat, scale)        # scene.rs:796
    viewer.update_gaussian_transform(1.0, Splat, GaussianShDegree.new(3), False)  # scene.rs:803
    viewer.preprocessor.preprocess("a"); viewer.radix_sorter.sort("a")           # scene.rs:856, 865
    viewer.poll()                                               # device.poll(Maintain::Wait)    scene.rs:873
    viewer.renderer.render(keys_far_to_near)                    # render_with_pass loop          scene.rs:2302

Every call goes into libgsx.so (HIP kernels); nothing here computes pixels.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import Iterable, Sequence

import numpy as np

from . import _lib
from ._lib import GsxError, SpecParams  # noqa: F401  (re-exported)
from .scene import GAUSSIAN_DTYPE


class GaussianDisplayMode(enum.IntEnum):
    """``gs::GaussianDisplayMode`` (src/app.rs:1147, src/tab/transform.rs:106-146)."""

    Splat = 0
    Ellipse = 1
    Point = 2


class GaussianShDegree:
    """``gs::GaussianShDegree``: 0..=3; ``new`` returns ``None`` out of range (src/tab/transform.rs:139)."""

    def __init__(self, deg: int):
        self.deg = int(deg)

    @classmethod
    def new(cls, deg: int):
        return cls(deg) if 0 <= int(deg) <= 3 else None

    @classmethod
    def new_unchecked(cls, deg: int) -> "GaussianShDegree":
        return cls(deg)

    def degree(self) -> int:
        return self.deg


class ShKind(enum.IntEnum):
    """``GaussianSh{Single,Half,Norm8,None}Config`` (src/app.rs:386-403)."""

    Single = 0
    Half = 1
    Norm8 = 2
    Remove = 3


class Cov3dKind(enum.IntEnum):
    """``GaussianCov3d{Single,Half}Config`` (src/app.rs:405-418)."""

    Single = 0
    Half = 1


def _f32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class GaussiansBuffer:
    """``gs::GaussiansBuffer<G>``: the model's resident pod planes in HBM."""

    def __init__(self, viewer: "MultiModelViewer", key: str):
        self._v, self._key = viewer, key

    def len(self) -> int:
        n = C.c_uint64()
        _lib.check(self._v._L.gsx_model_len(self._v._h, self._key.encode(), C.byref(n)))
        return int(n.value)

    __len__ = len

    def update_range(self, start: int, gaussians: np.ndarray) -> None:
        """``gaussians_buffer.update_range(queue, start, &[gs::Gaussian])`` (src/tab/scene.rs:2083-2084)."""
        g = np.ascontiguousarray(gaussians, dtype=GAUSSIAN_DTYPE)
        _lib.check(self._v._L.gsx_model_upload_range(self._v._h, self._key.encode(), int(start), g.ctypes.data, g.shape[0]))

    def update_range_pod_device(self, start: int, n: int, d_pos: int, d_color: int, d_sh: int, d_cov3d: int) -> None:
        """Zero-copy upload of pod planes that already live in device memory (raw device pointers)."""
        _lib.check(self._v._L.gsx_model_upload_pod_device(self._v._h, self._key.encode(), int(start), int(n), d_pos,
                                                          d_color, d_sh, d_cov3d))

    def download_pod(self):
        n = self.len()
        pos = np.empty((n, 3), np.float32)
        color = np.empty(n, np.uint32)
        sh = np.empty((n, 45), np.float32)
        cov = np.empty((n, 6), np.float32)
        _lib.check(self._v._L.gsx_model_download_pod(self._v._h, self._key.encode(), _f32p(pos), _u32p(color), _f32p(sh), _f32p(cov)))
        return pos, color, sh, cov


class BufferHandle:
    """A cloned buffer handle (``buffer.clone()`` in the app, src/app.rs:769-780): owns a device-side snapshot of the buffer as
    of the clone; ``download()`` may run on any thread while the viewer keeps rendering (``gsx_buffer_*``)."""

    KINDS = {"mask": 0, "edits": 1, "selection": 2}

    def __init__(self, viewer: "MultiModelViewer", key: str, kind: str):
        self._L, self.kind = viewer._L, kind
        self._h = C.c_void_p()
        _lib.check(self._L.gsx_model_buffer_retain(viewer._h, key.encode(), self.KINDS[kind], C.byref(self._h)))

    def clone(self) -> "BufferHandle":
        other = object.__new__(BufferHandle)
        other._L, other.kind, other._h = self._L, self.kind, C.c_void_p(self._h.value)
        _lib.check(self._L.gsx_buffer_retain(self._h))
        return other

    def len(self) -> int:
        n = C.c_uint64()
        _lib.check(self._L.gsx_buffer_len(self._h, C.byref(n)))
        return int(n.value)

    def download(self) -> np.ndarray:
        if self.kind == "edits":
            from .query import EDIT_DTYPE

            out = np.zeros(self.len(), EDIT_DTYPE)
        else:
            out = np.empty(self.len(), np.uint32)
        _lib.check(self._L.gsx_buffer_download(self._h, out.ctypes.data, out.size))
        return out

    def release(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._L.gsx_buffer_release(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class MaskBuffer:
    """``gs::MaskBuffer``: one bit per Gaussian, 1 = kept (src/tab/scene.rs:1851, src/app.rs:806-807)."""

    def __init__(self, viewer: "MultiModelViewer", key: str):
        self._v, self._key = viewer, key

    def upload(self, words: np.ndarray | None) -> None:
        if words is None:
            _lib.check(self._v._L.gsx_model_upload_mask(self._v._h, self._key.encode(), None, 0))
            return
        w = np.ascontiguousarray(words, dtype=np.uint32)
        _lib.check(self._v._L.gsx_model_upload_mask(self._v._h, self._key.encode(), _u32p(w), w.size))

    def download(self) -> np.ndarray:
        n = (self._v.models[self._key].gaussian_buffers.gaussians_buffer.len() + 31) // 32
        w = np.empty(n, np.uint32)
        _lib.check(self._v._L.gsx_model_download_mask(self._v._h, self._key.encode(), _u32p(w), n))
        return w

    def clone(self) -> BufferHandle:
        """``mask_buffer.clone()`` (src/app.rs:775): a handle for a download on another thread."""
        return BufferHandle(self._v, self._key, "mask")


class SelectionBuffer:
    """``gs::SelectionBuffer``: one bit per Gaussian, 1 = selected (src/tab/scene.rs:1846-1850)."""

    def __init__(self, viewer: "MultiModelViewer", key: str):
        self._v, self._key = viewer, key

    def upload(self, words: np.ndarray | None) -> None:
        if words is None:
            _lib.check(self._v._L.gsx_model_upload_selection(self._v._h, self._key.encode(), None, 0))
            return
        w = np.ascontiguousarray(words, dtype=np.uint32)
        _lib.check(self._v._L.gsx_model_upload_selection(self._v._h, self._key.encode(), _u32p(w), w.size))

    def download(self) -> np.ndarray:
        n = (self._v.models[self._key].gaussian_buffers.gaussians_buffer.len() + 31) // 32
        w = np.empty(n, np.uint32)
        _lib.check(self._v._L.gsx_model_download_selection(self._v._h, self._key.encode(), _u32p(w), n))
        return w

    def clone(self) -> BufferHandle:
        return BufferHandle(self._v, self._key, "selection")


class GaussiansEditBuffer:
    """``gs::GaussiansEditBuffer``: one ``GaussianEditPod`` per Gaussian (src/tab/scene.rs:1816-1830, app.rs:789)."""

    def __init__(self, viewer: "MultiModelViewer", key: str):
        self._v, self._key = viewer, key

    def download(self) -> np.ndarray:
        from .query import EDIT_DTYPE

        n = self._v.models[self._key].gaussian_buffers.gaussians_buffer.len()
        out = np.zeros(n, EDIT_DTYPE)
        _lib.check(self._v._L.gsx_model_download_edits(self._v._h, self._key.encode(), out.ctypes.data, n))
        return out

    def clone(self) -> BufferHandle:
        """``gaussians_edit_buffer.clone()`` (src/app.rs:772): a handle for a download on another thread."""
        return BufferHandle(self._v, self._key, "edits")

    def upload(self, edits: np.ndarray | None) -> None:
        from .query import EDIT_DTYPE

        if edits is None:
            _lib.check(self._v._L.gsx_model_upload_edits(self._v._h, self._key.encode(), None, 0))
            return
        e = np.ascontiguousarray(edits, EDIT_DTYPE)
        _lib.check(self._v._L.gsx_model_upload_edits(self._v._h, self._key.encode(), e.ctypes.data, e.size))


class MultiModelViewerGaussianBuffers:
    """``gs::MultiModelViewerGaussianBuffers<G>`` (src/tab/scene.rs:2111-2112)."""

    def __init__(self, viewer: "MultiModelViewer", key: str):
        self.gaussians_buffer = GaussiansBuffer(viewer, key)
        self.mask_buffer = MaskBuffer(viewer, key)
        self.selection_buffer = SelectionBuffer(viewer, key)
        self.gaussians_edit_buffer = GaussiansEditBuffer(viewer, key)


class MultiModelViewerModel:
    """``gs::MultiModelViewerModel {gaussian_buffers, bind_groups}`` (src/tab/scene.rs:2133-2139)."""

    def __init__(self, viewer: "MultiModelViewer", key: str):
        self.gaussian_buffers = MultiModelViewerGaussianBuffers(viewer, key)


class _Preprocessor:
    def __init__(self, v):
        self._v = v

    def preprocess(self, key: str) -> None:
        """``preprocessor.preprocess(encoder, bind_group, gaussian_count)`` (src/tab/scene.rs:856-863)."""
        _lib.check(self._v._L.gsx_preprocess(self._v._h, key.encode()))


class _Postprocessor:
    def __init__(self, v):
        self._v = v

    def postprocess(self, key: str) -> None:
        """``postprocessor.postprocess(encoder, bg0, bg1, gaussian_count, indirect_args)`` (src/tab/scene.rs:601-611)."""
        _lib.check(self._v._L.gsx_postprocess(self._v._h, key.encode()))


class _RadixSorter:
    def __init__(self, v):
        self._v = v

    def sort(self, key: str) -> None:
        """``radix_sorter.sort(encoder, bind_group, indirect_args)`` (src/tab/scene.rs:865-869)."""
        _lib.check(self._v._L.gsx_sort(self._v._h, key.encode()))


class _Renderer:
    def __init__(self, v):
        self._v = v

    def render(self, model_render_keys: Sequence[str]) -> None:
        """The ``render_with_pass`` loop over ``model_render_keys`` far -> near (src/tab/scene.rs:2302-2314)."""
        keys = [k.encode() for k in model_render_keys]
        arr = (C.c_char_p * max(len(keys), 1))(*keys)
        _lib.check(self._v._L.gsx_render(self._v._h, arr, len(keys)))


class CommGroup:
    """``gsx_comm_group``: the in-process transport for `world` viewers of ONE process (one host thread each)."""

    def __init__(self, world: int, timeout_ms: int = 0):
        self._L = _lib.load()
        self._h = C.c_void_p()
        self.world = int(world)
        _lib.check(self._L.gsx_comm_group_create(self.world, int(timeout_ms), C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._L.gsx_comm_group_destroy(self._h)
            self._h = C.c_void_p()


class MultiModelViewer:
    """``gs::MultiModelViewer<G>`` over libgsx.so."""

    def __init__(self, size=(1, 1), device: int = 0, stream: int | None = None, sh: ShKind = ShKind.Single,
                 cov3d: Cov3dKind = Cov3dKind.Single):
        self._L = _lib.load()
        self._h = C.c_void_p()
        desc = _lib.ViewerDesc(_lib.GSX_ABI_VERSION, int(device), stream, int(size[0]), int(size[1]))
        _lib.check(self._L.gsx_viewer_create(C.byref(desc), C.byref(self._h)))
        self.sh, self.cov3d = ShKind(sh), Cov3dKind(cov3d)
        self.models: dict[str, MultiModelViewerModel] = {}
        self.preprocessor = _Preprocessor(self)
        self.radix_sorter = _RadixSorter(self)
        self.renderer = _Renderer(self)
        self.postprocessor = _Postprocessor(self)
        self.size = (int(size[0]), int(size[1]))

    # -- lifetime --
    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._L.gsx_viewer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- models --
    def add_model(self, key: str, count: int) -> MultiModelViewerModel:
        """``GaussianBuffers::new_empty(device, count)`` + ``BindGroups::new`` + ``models.insert`` (scene.rs:2111-2139)."""
        _lib.check(self._L.gsx_model_create(self._h, key.encode(), int(count), int(self.sh), int(self.cov3d)))
        self.models[key] = MultiModelViewerModel(self, key)
        return self.models[key]

    def remove_model(self, key: str) -> None:
        """``viewer.remove_model(&key)`` (src/tab/scene.rs:2176)."""
        _lib.check(self._L.gsx_model_remove(self._h, key.encode()))
        self.models.pop(key, None)

    # -- per-frame uniforms --
    def update_camera(self, camera, size) -> None:
        """``viewer.update_camera(queue, &impl CameraTrait, uvec2 size)`` (src/tab/scene.rs:795)."""
        w, h = int(size[0]), int(size[1])
        self.update_camera_with_matrices(camera.view(), camera.projection(w / h), (w, h))

    def update_camera_with_matrices(self, view, proj, size) -> None:
        v = np.ascontiguousarray(view, np.float32).reshape(16)
        p = np.ascontiguousarray(proj, np.float32).reshape(16)
        _lib.check(self._L.gsx_update_camera(self._h, _f32p(v), _f32p(p), int(size[0]), int(size[1])))
        self.size = (int(size[0]), int(size[1]))

    def update_model_transform(self, key: str, pos, quat, scale) -> None:
        """``viewer.update_model_transform(queue, key, pos, quat, scale)`` (src/tab/scene.rs:796-802)."""
        p = np.ascontiguousarray(pos, np.float32).reshape(3)
        q = np.ascontiguousarray(quat, np.float32).reshape(4)
        s = np.ascontiguousarray(scale, np.float32).reshape(3)
        _lib.check(self._L.gsx_update_model_transform(self._h, key.encode(), _f32p(p), _f32p(q), _f32p(s)))

    def update_gaussian_transform(self, size: float, display_mode: GaussianDisplayMode, sh_deg, no_sh0: bool) -> None:
        """``viewer.update_gaussian_transform(queue, size, display_mode, sh_deg, no_sh0)`` (src/tab/scene.rs:803-809)."""
        deg = sh_deg.degree() if isinstance(sh_deg, GaussianShDegree) else int(sh_deg)
        _lib.check(self._L.gsx_update_gaussian_transform(self._h, float(size), int(display_mode), deg, 1 if no_sh0 else 0))

    # -- selection / edits / queries (spec §7) --
    def update_query(self, pod) -> None:
        """``viewer.update_query(queue, &query_pod)`` (src/tab/scene.rs:785)."""
        raw = pod.raw()
        _lib.check(self._L.gsx_update_query(self._h, C.byref(raw)))

    def update_query_texture(self, texels: np.ndarray) -> None:
        """``viewer.update_query_texture_size`` + ``query_toolset.render(.., &query_texture)`` (scene.rs:740, 791)."""
        t = np.ascontiguousarray(texels, np.uint8)
        _lib.check(self._L.gsx_update_query_texture(self._h, t.ctypes.data, t.shape[1], t.shape[0]))

    def update_selection_highlight(self, rgba) -> None:
        """``viewer.update_selection_highlight(queue, vec4)`` / ``_with_pod`` (src/tab/scene.rs:816-829)."""
        c = np.ascontiguousarray(rgba, np.float32).reshape(4)
        _lib.check(self._L.gsx_update_selection_highlight(self._h, _f32p(c)))

    def update_selection_edit_with_pod(self, pod) -> None:
        """``viewer.update_selection_edit_with_pod(queue, &gs::GaussianEditPod)`` (src/tab/scene.rs:815, 821, 848)."""
        raw = pod.raw()
        _lib.check(self._L.gsx_update_selection_edit(self._h, C.byref(raw)))

    def show_unedited(self, key: str, on: bool) -> None:
        """Preprocess with the unedited model's bind group (src/tab/scene.rs:856-863)."""
        _lib.check(self._L.gsx_model_show_unedited(self._h, key.encode(), 1 if on else 0))

    def download_query_hits(self, key: str) -> np.ndarray:
        """``gs::query::download(&device, &queue, &count_buffer, &results_buffer)`` (src/tab/scene.rs:651-657)."""
        from .query import HIT_DTYPE

        n = C.c_uint64()
        out = np.zeros(65536, HIT_DTYPE)
        _lib.check(self._L.gsx_query_download_hits(self._h, key.encode(), out.ctypes.data, out.size, C.byref(n)))
        return out[: int(n.value)].copy()

    def set_spec_params(self, **kw) -> SpecParams:
        sp = SpecParams()
        self._L.gsx_spec_params_default(C.byref(sp))
        for k, val in kw.items():
            if not hasattr(sp, k):
                raise KeyError(k)
            setattr(sp, k, float(val))
        _lib.check(self._L.gsx_viewer_set_spec_params(self._h, C.byref(sp)))
        return sp

    def set_render_options(self, **kw) -> None:
        """``gsx_render_options``: progressive depth slabs (default on), first_slab_divisor, min_slab, growth."""
        o = _lib.RenderOptions()
        self._L.gsx_render_options_default(C.byref(o))
        for k, val in kw.items():
            if not hasattr(o, k):
                raise KeyError(k)
            setattr(o, k, float(val) if k == "spec_margin" else int(val))
        _lib.check(self._L.gsx_viewer_set_render_options(self._h, C.byref(o)))

    # -- frame execution --
    def poll(self) -> None:
        """``device.poll(wgpu::Maintain::Wait)`` (src/tab/scene.rs:614, 873)."""
        _lib.check(self._L.gsx_sync(self._h))

    def render_frame(self, model_render_keys: Sequence[str]) -> None:
        """preprocess + sort every key, then render: the whole per-frame protocol in one call."""
        keys = [k.encode() for k in model_render_keys]
        arr = (C.c_char_p * max(len(keys), 1))(*keys)
        _lib.check(self._L.gsx_render_frame(self._h, arr, len(keys)))

    # -- several GPUs: the index-sharded frame as one library call (include/gsx.h "multi-GPU"; no reference counterpart) --
    def comm_init_group(self, group: "CommGroup", rank: int) -> None:
        """Seat `rank` of an in-process group: one host thread + one viewer per GPU, collectives as peer copies."""
        _lib.check(self._L.gsx_viewer_comm_init_group(self._h, group._h, int(rank)))
        self._group = group  # keeps the group alive as long as the viewer

    def comm_init_rccl(self, world: int, rank: int, unique_id: bytes) -> None:
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(self._L.gsx_viewer_comm_init(self._h, int(world), int(rank), buf))

    def comm_init_custom(self, world: int, rank: int, all_to_all, all_gather) -> None:
        """A transport of the caller's own: two callables (d_send, d_recv, bytes, hip_stream) -> gsx_status that enqueue."""
        self._comm_fns = (_lib.COMM_FN(lambda ctx, s, r, b, st: int(all_to_all(s, r, b, st) or 0)),
                          _lib.COMM_FN(lambda ctx, s, r, b, st: int(all_gather(s, r, b, st) or 0)))
        _lib.check(self._L.gsx_viewer_comm_init_custom(self._h, int(world), int(rank), self._comm_fns[0], self._comm_fns[1], None))

    def comm_init_custom_v(self, world: int, rank: int, all_to_all_v, gather_v) -> None:
        """A transport of the caller's own that moves pieces of UNEQUAL size (``gsx_viewer_comm_init_custom_v``):
        ``all_to_all_v(d_send, send_offsets, send_bytes, d_recv, recv_offsets, recv_bytes, hip_stream)`` and
        ``gather_v(d_send, send_bytes, d_recv, recv_offsets, recv_bytes, root, hip_stream)`` -> gsx_status; the offset / size
        arguments arrive as lists of `world` ints.  Both enqueue on ``hip_stream``."""
        w = int(world)

        def a2a(ctx, s, so, sb, r, ro, rb, st):
            return int(all_to_all_v(s, [so[i] for i in range(w)], [sb[i] for i in range(w)], r, [ro[i] for i in range(w)], [rb[i] for i in range(w)], st) or 0)

        def gat(ctx, s, n, r, ro, rb, root, st):
            return int(gather_v(s, n, r, [ro[i] for i in range(w)], [rb[i] for i in range(w)], root, st) or 0)

        self._comm_fns = (_lib.COMM_A2A_V_FN(a2a), _lib.COMM_GATHER_V_FN(gat))
        _lib.check(self._L.gsx_viewer_comm_init_custom_v(self._h, w, int(rank), self._comm_fns[0], self._comm_fns[1], None))

    def comm_init_custom_v_native(self, world: int, rank: int, all_to_all_v_ptr, gather_v_ptr, ctx) -> None:
        """The same with two NATIVE functions (addresses or ctypes function objects of a shared library) and their context pointer:
        nothing of the transport runs in Python (tools/replay_transport.cpp)."""
        a = C.cast(all_to_all_v_ptr, _lib.COMM_A2A_V_FN)
        g = C.cast(gather_v_ptr, _lib.COMM_GATHER_V_FN)
        self._comm_fns = (a, g)
        _lib.check(self._L.gsx_viewer_comm_init_custom_v(self._h, int(world), int(rank), a, g, C.c_void_p(ctx)))

    def comm_destroy(self) -> None:
        _lib.check(self._L.gsx_viewer_comm_destroy(self._h))

    def debug_download_lane_framebuffer(self, lane: int, size=None) -> np.ndarray:
        """Tests: lane `lane`'s framebuffer once its stream has drained, WITHOUT completing the frames in flight (size: the viewport of
        the frame that lane holds, if the viewer's has changed since)."""
        w, h = size if size is not None else self.size
        out = np.empty((h, w, 4), np.float32)
        _lib.check(self._L.gsx_debug_download_lane_framebuffer(self._h, int(lane), out.ctypes.data, out.size))
        return out

    def shard_render_frame(self, key: str, shard_records_max: int, speculate: bool = True, margin: float = 0.25, radius: int = 3) -> None:
        """One whole index-sharded frame of this rank (``gsx_shard_render_frame``)."""
        _lib.check(self._L.gsx_shard_render_frame(self._h, key.encode(), int(shard_records_max), 1 if speculate else 0, float(margin), int(radius)))

    def shard_render_frame_keys(self, model_render_keys: Sequence[str], shard_records_max: Sequence[int], speculate: bool = True,
                                margin: float = 0.25, radius: int = 3) -> None:
        """Layered models, far -> near like ``renderer.render`` (``gsx_shard_render_frame_keys``)."""
        keys = [k.encode() for k in model_render_keys]
        arr = (C.c_char_p * len(keys))(*keys)
        mx = (C.c_uint32 * len(keys))(*[int(x) for x in shard_records_max])
        _lib.check(self._L.gsx_shard_render_frame_keys(self._h, arr, len(keys), mx, 1 if speculate else 0, float(margin), int(radius)))

    def shard_set_limits(self, key: str, limits: np.ndarray) -> None:
        """Per-tile depth-key limits (uint32 [tiles_y, tiles_x]) the next sharded frame uses instead of the last frame's."""
        a = np.ascontiguousarray(limits, np.uint32)
        _lib.check(self._L.gsx_shard_set_limits(self._h, key.encode(), a.ctypes.data))

    def shard_set_slot_records(self, key: str, records: int) -> None:
        _lib.check(self._L.gsx_shard_set_slot_records(self._h, key.encode(), int(records)))

    def shard_set_gather_root(self, root: int) -> None:
        """-1: every rank's framebuffer holds the whole frame after a sharded frame (default); r >= 0: only rank r's does."""
        _lib.check(self._L.gsx_shard_set_gather_root(self._h, int(root)))

    def shard_set_band_edges(self, world: int, edges) -> None:
        """Rank g owns the tile rows [edges[g], edges[g + 1]) in every following sharded frame (None: the library's own layout)."""
        if edges is None:
            _lib.check(self._L.gsx_shard_set_band_edges(self._h, int(world), None))
            return
        a = np.ascontiguousarray(edges, np.uint32)
        assert a.size == world + 1
        _lib.check(self._L.gsx_shard_set_band_edges(self._h, int(world), _u32p(a)))

    def shard_get_band_edges(self, world: int) -> np.ndarray:
        """The band layout the last sharded frame used (uint32 [world + 1] tile rows)."""
        out = np.empty(world + 1, np.uint32)
        _lib.check(self._L.gsx_shard_get_band_edges(self._h, int(world), _u32p(out)))
        return out

    def shard_set_balance(self, enabled: bool) -> None:
        """Balance the bands by the previous frame's per-row work (default, where the transport moves unequal pieces) or keep them equal."""
        _lib.check(self._L.gsx_shard_set_balance(self._h, 1 if enabled else 0))

    def shard_download_limits(self, key: str) -> np.ndarray:
        w, h = self.size
        out = np.empty(((h + 15) // 16, (w + 15) // 16), np.uint32)
        _lib.check(self._L.gsx_shard_download_limits(self._h, key.encode(), _u32p(out), out.size))
        return out

    def comm_info(self) -> dict:
        """What the communicator itself says (``gsx_viewer_comm_info``): transport, the ranks RCCL counted, its version."""
        ci = _lib.CommInfo()
        _lib.check(self._L.gsx_viewer_comm_info(self._h, C.byref(ci)))
        out = {n: int(getattr(ci, n)) for n, _ in _lib.CommInfo._fields_}
        out["transport"] = ("none", "rccl", "in-process group", "custom")[min(out["transport"], 3)]
        return out

    def shard_stats(self, reset: bool = False) -> dict:
        st = _lib.ShardStats()
        _lib.check(self._L.gsx_shard_get_stats(self._h, C.byref(st), 1 if reset else 0))
        return {n: int(getattr(st, n)) for n, _ in _lib.ShardStats._fields_}

    # -- readback --
    def download_framebuffer(self) -> np.ndarray:
        """float32 [H, W, 4]: premultiplied r,g,b and transmittance T."""
        w, h = self.size
        out = np.empty((h, w, 4), np.float32)
        _lib.check(self._L.gsx_download_framebuffer(self._h, _f32p(out), out.size))
        return out

    def download_rgba8(self, background=(0.0, 0.0, 0.0)) -> np.ndarray:
        w, h = self.size
        out = np.empty((h, w, 4), np.uint8)
        bg = np.ascontiguousarray(background, np.float32).reshape(3)
        _lib.check(self._L.gsx_download_rgba8(self._h, _f32p(bg), out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size))
        return out

    def framebuffer_device_ptr(self):
        p, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
        _lib.check(self._L.gsx_framebuffer_device_ptr(self._h, C.byref(p), C.byref(w), C.byref(h)))
        return p.value, int(w.value), int(h.value)

    # -- parity / introspection --
    def frame_stats(self, key: str) -> dict:
        st = _lib.FrameStats()
        _lib.check(self._L.gsx_model_frame_stats(self._h, key.encode(), C.byref(st)))
        return dict(n_gaussians=int(st.n_gaussians), n_visible=int(st.n_visible), n_tile_entries=int(st.n_tile_entries),
                    n_sorted=int(st.n_sorted), n_repair_tiles=int(st.n_repair_tiles), n_repair_sorted=int(st.n_repair_sorted),
                    speculated=bool(st.speculated), overflow_slabs=int(st.overflow_slabs))

    def download_projection(self, key: str) -> dict:
        n = self.models[key].gaussian_buffers.gaussians_buffer.len()
        out = dict(key=np.empty(n, np.uint32), rect=np.empty((n, 4), np.uint32), mean2d=np.empty((n, 2), np.float32),
                   conic_opacity=np.empty((n, 4), np.float32), rgb=np.empty((n, 3), np.float32))
        _lib.check(self._L.gsx_model_download_projection(self._h, key.encode(), _u32p(out["key"]), _u32p(out["rect"]),
                                                         _f32p(out["mean2d"]), _f32p(out["conic_opacity"]), _f32p(out["rgb"])))
        out["n_visible"] = int(np.count_nonzero(out["key"] != 0xFFFFFFFF))
        return out

    def download_sorted(self, key: str) -> np.ndarray:
        nv = C.c_uint64()
        _lib.check(self._L.gsx_model_download_sorted(self._h, key.encode(), None, 0, C.byref(nv)))
        idx = np.empty(max(int(nv.value), 1), np.uint32)
        _lib.check(self._L.gsx_model_download_sorted(self._h, key.encode(), _u32p(idx), idx.size, C.byref(nv)))
        return idx[: int(nv.value)]

    def download_tile_lists(self, key: str):
        st = self.frame_stats(key)
        w, h = self.size
        tiles = ((w + 15) // 16) * ((h + 15) // 16)
        off = np.empty(tiles + 1, np.uint32)
        lst = np.empty(max(st["n_tile_entries"], 1), np.uint32)
        _lib.check(self._L.gsx_model_download_tile_lists(self._h, key.encode(), _u32p(off), off.size, _u32p(lst), lst.size))
        return off, lst[: st["n_tile_entries"]]

    # -- timing --
    def launch_stats(self, reset: bool = False) -> dict:
        """How this viewer's (and its lanes') kernel launches reached the device: ``gsx_launch_stats`` (csrc/gsx_launch.h)."""
        ls = _lib.LaunchStats()
        _lib.check(self._L.gsx_viewer_launch_stats(self._h, C.byref(ls), 1 if reset else 0))
        return {n: int(getattr(ls, n)) for n, _ in ls._fields_}

    def set_pass_timing(self, enabled, passes=None) -> None:
        """``passes``: names from ``_lib.GSX_PASS_NAMES`` to bracket with events (default: all)."""
        code = 0
        if enabled:
            code = 1 if passes is None else sum(1 << (_lib.GSX_PASS_NAMES.index(p) + 1) for p in passes)
        _lib.check(self._L.gsx_set_pass_timing(self._h, code))

    def get_pass_timing(self) -> dict:
        ms = (C.c_float * _lib.GSX_PASS_COUNT)()
        launches = (C.c_uint32 * _lib.GSX_PASS_COUNT)()
        _lib.check(self._L.gsx_get_pass_timing(self._h, ms, launches))
        return {n: dict(ms=float(ms[i]), launches=int(launches[i])) for i, n in enumerate(_lib.GSX_PASS_NAMES)}


def set_launch_graphs(enabled) -> None:
    """Process-wide: frame-level entry points submit their launches as cached HIP graphs while their stream is busy (True / 1), always
    (2: tests) or never (False / 0, the default: the graphs save host time, not device time — csrc/gsx_launch.h)."""
    _lib.load().gsx_debug_set_launch_graphs(int(enabled))


def launch_count() -> int:
    """Kernel launches this process has asked libgsx for so far."""
    return int(_lib.load().gsx_debug_launch_count())


def device_bytes() -> int:
    """Device memory the library's buffers hold in this process right now (``gsx_debug_device_bytes``)."""
    return int(_lib.load().gsx_debug_device_bytes())


def render_keys_far_to_near(viewer_models_centers: dict, camera_pos) -> list:
    """``model_render_keys`` exactly as the app builds them (src/tab/scene.rs:533-558)."""
    from .camera import model_render_order

    return model_render_order(camera_pos, viewer_models_centers)
