"""GPU: the multi-GPU stage kernels (pack by destination, import, row-restricted binning, strip pack /
unpack) emulated on ONE device: `world` viewers each hold an index shard, records are exchanged by plain
device copies in the order RCCL's all-to-all would deliver them, and the assembled frame must equal the
single-viewer frame bit-for-bit.  The pack kernel is also checked against the oracle's routing."""
import ctypes as C

import numpy as np
import pytest

import oracle
from tests import common
from tests.oracle_stages import pack_by_destination
from wgpu_3dgs_viewer_app_amd import _lib, camera, parallel
from wgpu_3dgs_viewer_app_amd.hip_stages import HipStages

pytestmark = pytest.mark.gpu


def _single(g, cam, w, h):
    st = HipStages()
    st.load_shard("shard", g, 0, g.shape[0])
    st.set_uniforms("shard", cam, (w, h))
    st.render_local("shard")
    stats = st.stats("shard")
    fb = st.framebuffer()
    st.close()
    return fb, stats


@pytest.mark.parametrize("world", [2, 3, 8])
def test_emulated_world_matches_single_viewer(world):
    import torch

    n, w, h = 9000, 208, 152
    g = common.small_scene(n, 91)
    cam = camera.orbit_pose(57)
    ref, ref_stats = _single(g, cam, w, h)

    stages = []
    sends, counts = [], []
    for r in range(world):
        s0, c = parallel.shard_range(n, r, world)
        st = HipStages(use_torch=True)
        st.load_shard("shard", g[s0:s0 + c], s0, n)
        st.set_uniforms("shard", cam, (w, h))
        with st.stream_ctx():
            send, cnt = st.project_and_pack("shard", world)
            st.poll()
        # routing parity with the oracle on this shard
        f = common.oracle_frame(cam, w, h)
        pr = oracle.project(f, *oracle.convert(g[s0:s0 + c]))
        rsend, rcnt = pack_by_destination(pr, world)
        assert cnt == rcnt
        assert np.array_equal(send[: sum(cnt)].cpu().numpy().view(np.uint32), rsend.view(np.uint32)), "packed records differ"
        stages.append(st)
        sends.append(send[: sum(cnt)].clone())
        counts.append(cnt)
    torch.cuda.synchronize()

    strips, total_entries = [], 0
    for r, st in enumerate(stages):
        # what all_to_all_single delivers to rank r: from every source rank, its group for destination r
        chunks = []
        for src in range(world):
            off = sum(counts[src][:r])
            chunks.append(sends[src][off:off + counts[src][r]])
        recv = torch.cat(chunks) if chunks else sends[0][:0]
        with st.stream_ctx():
            st.render_records("shard", recv.contiguous(), recv.shape[0], world, r)
            stats = st.stats("shard")
            strips.append(st.own_strip(world, r).clone())
            st.poll()
        total_entries += stats["n_tile_entries"]
    torch.cuda.synchronize()
    assert total_entries == ref_stats["n_tile_entries"], "row ownership must partition the tile entries"

    st0 = stages[0]
    with st0.stream_ctx():
        gathered = st0.gather_buffer(strips[0], world)
        gathered.view(world, -1).copy_(torch.stack(strips))
        st0.assemble(gathered, world)
        st0.poll()
    fb = st0.framebuffer()
    for st in stages:
        st.close()
    assert np.array_equal(fb, ref), f"sharded frame differs from single-GPU frame: L-inf {np.abs(fb - ref).max()}"


def test_pack_capacity_error():
    import torch

    g = common.small_scene(500, 92)
    st = HipStages(use_torch=True)
    st.load_shard("shard", g, 0, 500)
    st.set_uniforms("shard", camera.orbit_pose(1), (128, 96))
    v = st.viewer
    v.preprocessor.preprocess("shard")
    small = torch.empty((4, 12), dtype=torch.float32, device="cuda")
    counts = (C.c_uint64 * 2)()
    with pytest.raises(_lib.GsxError):
        _lib.check(v._L.gsx_shard_pack(v._h, b"shard", 2, small.data_ptr(), 4, counts))
    st.close()
