"""Stage backend of ``parallel.ShardedViewer`` over libgsx.so (the product path; HIP kernels only).

PyTorch is used for the device buffers that RCCL moves (send / receive records, framebuffer strips) and
for stream identity; every byte in those buffers is produced and consumed by the HIP kernels in csrc/.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import numpy as np

from . import _lib
from .camera import ModelTransform
from .viewer import Cov3dKind, GaussianDisplayMode, GaussianShDegree, MultiModelViewer, ShKind

RECORD_FLOATS = 12


class HipStages:
    def __init__(self, device: int = 0, stream=None, use_torch: bool = False, sh: int = 0, cov3d: int = 0):
        self.device = device
        self.torch_stream = None
        if use_torch:
            import torch

            # a dedicated non-default stream shared by torch (RCCL staging) and libgsx, so launch order = data order
            self.torch_stream = torch.cuda.Stream(device=device)
            stream = self.torch_stream.cuda_stream
        self.viewer = MultiModelViewer(size=(1, 1), device=device, stream=stream, sh=ShKind(sh), cov3d=Cov3dKind(cov3d))
        self._size = (1, 1)
        self._send = None
        self._all = None

    def stream_ctx(self):
        if self.torch_stream is None:
            return contextlib.nullcontext()
        import torch

        return torch.cuda.stream(self.torch_stream)

    # -- scene --
    def load_shard(self, key: str, gaussians: np.ndarray, start: int, n_total: int) -> None:
        self.viewer.add_model(key, gaussians.shape[0])
        self.viewer.models[key].gaussian_buffers.gaussians_buffer.update_range(0, gaussians)
        self._n_local = gaussians.shape[0]

    def set_uniforms(self, key, camera, size, model_transform=None, gaussian_transform=None) -> None:
        mt = model_transform or ModelTransform()
        gt = gaussian_transform or (1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
        self.viewer.update_camera(camera, size)
        self.viewer.update_model_transform(key, mt.pos, mt.quat(), mt.scale)
        self.viewer.update_gaussian_transform(*gt)
        self._size = (int(size[0]), int(size[1]))

    # -- single GPU --
    def render_local(self, key: str) -> None:
        """Enqueue one whole frame; no host synchronisation (statistics are fetched lazily by ``stats``)."""
        self.viewer.render_frame([key])

    def stats(self, key: str) -> dict:
        return self.viewer.frame_stats(key)

    # -- multi GPU (stage split of include/gsx.h) --
    def project_and_pack(self, key: str, world: int):
        import torch

        v = self.viewer
        v.preprocessor.preprocess(key)
        cap = max(self._n_local * world, 1)  # worst case: every record touches every rank's rows
        if self._send is None or self._send.shape[0] < cap:
            self._send = torch.empty((cap, RECORD_FLOATS), dtype=torch.float32, device=f"cuda:{self.device}")
        counts = (C.c_uint64 * world)()
        _lib.check(v._L.gsx_shard_pack(v._h, key.encode(), world, self._send.data_ptr(), cap, counts))
        return self._send, [int(c) for c in counts]

    def alloc_records(self, n: int):
        import torch

        return torch.empty((n, RECORD_FLOATS), dtype=torch.float32, device=f"cuda:{self.device}")

    def render_records(self, key: str, recv, n: int, world: int, rank: int) -> dict:
        v = self.viewer
        _lib.check(v._L.gsx_shard_import(v._h, key.encode(), recv.data_ptr() if n else None, n, world, rank))
        v.radix_sorter.sort(key)
        v.renderer.render([key])

    def own_strip(self, world: int, rank: int):
        import torch

        v = self.viewer
        nbytes = C.c_uint64()
        _lib.check(v._L.gsx_shard_strip_bytes(v._h, world, C.byref(nbytes)))
        strip = torch.empty(nbytes.value // 4, dtype=torch.float32, device=f"cuda:{self.device}")
        _lib.check(v._L.gsx_shard_pack_strip(v._h, world, rank, strip.data_ptr(), nbytes.value))
        return strip

    def gather_buffer(self, strip, world: int):
        """One contiguous buffer of ``world`` strips that RCCL gathers into directly."""
        import torch

        if self._all is None or self._all.numel() != world * strip.numel():
            self._all = torch.empty(world * strip.numel(), dtype=torch.float32, device=strip.device)
        return self._all

    def assemble(self, gathered, world: int) -> None:
        v = self.viewer
        _lib.check(v._L.gsx_shard_unpack_strips(v._h, world, gathered.data_ptr(), gathered.numel() * 4))

    # -- common --
    def framebuffer(self) -> np.ndarray:
        return self.viewer.download_framebuffer()

    def poll(self) -> None:
        self.viewer.poll()

    def set_pass_timing(self, on: bool) -> None:
        self.viewer.set_pass_timing(on)

    def get_pass_timing(self) -> dict:
        return self.viewer.get_pass_timing()

    def close(self) -> None:
        self.viewer.close()
