"""Multi-GPU rendering: one process per MI355X, the Gaussian array sharded by splat index.

The reference is single-device (src/main.rs:85-98); this layer is new.  "over" blending is order
dependent per pixel, so per-GPU images of index shards cannot simply be summed.  The
path therefore has ONE real exchange step, sort-middle by screen rows:

  stage P  every rank projects / culls / colours its resident shard          (no communication)
  stage X  projected 48-byte records are routed to the rank(s) that own the tile rows they touch
           (tile row ty belongs to rank ty % world): RCCL all-to-all over xGMI, all 7 links busy
  stage C  every rank depth-sorts what it received, bins it into ITS tile rows and composites them;
           received records arrive ordered by (source rank, local index) = global index, so the
           stable sort breaks depth ties exactly like the single-GPU path -> bit-identical pixels
  stage M  the disjoint (rgb, T) tile-row strips are all-gathered and assembled into one framebuffer on rank 0.

``torch.distributed`` (backend nccl = RCCL) is plumbing only: it moves buffers the HIP kernels packed.
The stage implementation is injectable (``stages=``) so the routing / merge logic is covered on CPU
with gloo at world_size 2 (tests/test_parallel_cpu.py) using the oracle as a checker.
"""
from __future__ import annotations

import numpy as np

RECORD_FLOATS = 12  # mean.xy, rect.xy (bits), conic.abc, opacity, rgb, depth  = 48 bytes


def shard_range(n: int, rank: int, world: int):
    """Contiguous index shard [start, start+count) of rank; sizes differ by at most one."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def rows_owned(tiles_y: int, rank: int, world: int) -> np.ndarray:
    """Tile rows owned by a rank: ty % world == rank."""
    return np.arange(rank, tiles_y, world)


class ShardedViewer:
    """One rank's view of a scene rendered by ``world`` GPUs.  With world == 1 (and use_dist False) this is
    exactly the single-GPU ``MultiModelViewer`` protocol."""

    KEY = "shard"

    def __init__(self, device: int = 0, world: int = 1, rank: int = 0, use_dist: bool = False, stream=None,
                 stages=None, group=None, sh: int = 0, cov3d: int = 0):
        self.world, self.rank, self.use_dist, self.group = world, rank, use_dist, group
        if stages is None:
            from .hip_stages import HipStages  # the product path: libgsx.so, fails loudly if missing

            stages = HipStages(device=device, stream=stream, use_torch=use_dist, sh=sh, cov3d=cov3d)
        self.stages = stages
        self._stats = dict(n_gaussians=0, n_visible=0, n_tile_entries=0)

    # -- scene --
    def load_shard(self, gaussians: np.ndarray, start: int, n_total: int) -> None:
        self.stages.load_shard(self.KEY, gaussians, start, n_total)

    # -- frame --
    def render_frame(self, camera, size, model_transform=None, gaussian_transform=None):
        """One frame.  On rank 0 ``self.stages.framebuffer()`` afterwards holds the complete (rgb, T) image."""
        st = self.stages
        st.set_uniforms(self.KEY, camera, size, model_transform, gaussian_transform)
        if not self.use_dist:
            st.render_local(self.KEY)
            return
        with st.stream_ctx():
            self._render_frame_dist()

    def _render_frame_dist(self):
        import torch
        import torch.distributed as dist

        st = self.stages
        world, rank = self.world, self.rank
        # stage P + pack: records grouped by destination rank, ascending local index inside each group
        send, send_counts = st.project_and_pack(self.KEY, world)
        # stage X: counts first (tiny), then the records with exact split sizes
        sc = torch.as_tensor(send_counts, dtype=torch.int64, device=send.device)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        recv_counts = [int(x) for x in rc.tolist()]
        recv = st.alloc_records(sum(recv_counts))
        dist.all_to_all_single(recv, send[: sum(send_counts)], output_split_sizes=recv_counts,
                               input_split_sizes=[int(x) for x in send_counts], group=self.group)
        # stage C: sort + bin + composite this rank's tile rows
        st.render_records(self.KEY, recv, sum(recv_counts), world, rank)
        # stage M: the disjoint tile-row strips are all-gathered (every rank could present; rank 0 assembles)
        strip = st.own_strip(world, rank)  # flat [rows_per_rank * 16 * W * 4]: the rows this rank owns, packed
        gathered = st.gather_buffer(strip, world)  # flat [world * strip]
        dist.all_gather_into_tensor(gathered, strip, group=self.group)
        if rank == 0:
            st.assemble(gathered, world)

    def framebuffer(self) -> np.ndarray:
        return self.stages.framebuffer()

    def poll(self) -> None:
        self.stages.poll()

    def last_stats(self) -> dict:
        """Statistics of the last frame (synchronises)."""
        return dict(self.stages.stats(self.KEY))

    def set_pass_timing(self, on: bool) -> None:
        self.stages.set_pass_timing(on)

    def get_pass_timing(self) -> dict:
        return self.stages.get_pass_timing()

    def close(self) -> None:
        self.stages.close()
