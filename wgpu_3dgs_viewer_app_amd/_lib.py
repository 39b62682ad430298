"""ctypes binding of libgsx.so (include/gsx.h).  The library is the product; this file is plumbing.

The shared object is built in-tree by ``__graft_entry__.build()`` (``make -C csrc``).  If it is missing
the import of the product path FAILS LOUDLY — there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsx.so")

GSX_ABI_VERSION = 3
(GSX_OK, GSX_ERR_INVALID_ARG, GSX_ERR_OOM, GSX_ERR_HIP, GSX_ERR_RCCL, GSX_ERR_IO, GSX_ERR_PLY, GSX_ERR_NOT_FOUND,
 GSX_ERR_UNSUPPORTED, GSX_ERR_NO_DEVICE) = range(10)
GSX_PASS_NAMES = ("project", "depth_sort", "bin", "tile_sort", "composite", "project_geom", "shade")
GSX_PASS_COUNT = len(GSX_PASS_NAMES)

#: every symbol include/gsx.h declares (tests check the library exports exactly these)
EXPORTS = (
    "gsx_last_error_string", "gsx_abi_version", "gsx_spec_params_default", "gsx_viewer_create", "gsx_viewer_destroy",
    "gsx_viewer_set_spec_params", "gsx_model_create", "gsx_model_remove", "gsx_model_len", "gsx_model_upload_range",
    "gsx_model_upload_pod_device", "gsx_update_camera", "gsx_update_model_transform", "gsx_update_gaussian_transform",
    "gsx_model_upload_mask", "gsx_model_download_mask", "gsx_preprocess", "gsx_sort", "gsx_sync", "gsx_render",
    "gsx_render_frame", "gsx_download_framebuffer", "gsx_download_rgba8", "gsx_framebuffer_device_ptr",
    "gsx_model_frame_stats", "gsx_model_download_projection", "gsx_model_download_sorted",
    "gsx_model_download_tile_lists", "gsx_model_download_pod", "gsx_set_pass_timing", "gsx_get_pass_timing",
    "gsx_mask_evaluate", "gsx_ply_read_header", "gsx_ply_read_gaussians", "gsx_ply_write", "gsx_render_options_default", "gsx_viewer_set_render_options", "gsx_shard_layout", "gsx_viewer_set_external_framebuffer", "gsx_shard_pack", "gsx_shard_import", "gsx_shard_feedback_words", "gsx_shard_feedback", "gsx_shard_set_windows", "gsx_viewer_set_band", "gsx_resolve_rgba8_device",
    "gsx_render_more", "gsx_debug_set_radix_rank_mode",
    "gsx_shard_frame_begin", "gsx_shard_slot_records", "gsx_shard_pack_slots", "gsx_shard_import_slots", "gsx_shard_verify",
    "gsx_shard_wait_verdict", "gsx_shard_repair_count", "gsx_shard_post_counts", "gsx_shard_frame_end",
    "gsx_shard_next_windows", "gsx_shard_download_limits", "gsx_comm_unique_id", "gsx_viewer_comm_init", "gsx_viewer_comm_destroy",
    "gsx_comm_all_to_all", "gsx_comm_all_gather", "gsx_shard_render_frame", "gsx_shard_render_frame_keys",
    "gsx_comm_group_create", "gsx_comm_group_destroy", "gsx_viewer_comm_init_group", "gsx_viewer_comm_init_custom",
    "gsx_shard_set_limits", "gsx_shard_set_slot_records", "gsx_shard_get_stats", "gsx_shard_set_gather_root",
    "gsx_model_buffer_retain", "gsx_buffer_retain", "gsx_buffer_release", "gsx_buffer_len", "gsx_buffer_download",
    "gsx_gaussian_edit_default", "gsx_update_query", "gsx_update_query_texture", "gsx_update_selection_highlight",
    "gsx_update_selection_edit", "gsx_model_show_unedited", "gsx_postprocess", "gsx_model_upload_selection",
    "gsx_model_download_selection", "gsx_model_download_edits", "gsx_model_upload_edits", "gsx_query_download_hits",
    "gsx_query_hit_pos_by_closest", "gsx_query_hit_pos_by_alpha_range",
    "gsx_debug_set_launch_graphs", "gsx_debug_launch_count", "gsx_debug_device_bytes", "gsx_debug_download_lane_framebuffer", "gsx_viewer_launch_stats", "gsx_debug_tile_profile",
    "gsx_viewer_comm_init_custom_v", "gsx_shard_set_band_edges", "gsx_shard_get_band_edges", "gsx_shard_set_balance",
    "gsx_viewer_comm_info",
)


class GaussianEdit(C.Structure):
    """``gsx_gaussian_edit`` = gs::GaussianEditPod (32 bytes)."""
    _fields_ = [("flag", C.c_uint32), ("color", C.c_float * 3), ("contrast", C.c_float), ("exposure", C.c_float),
                ("gamma", C.c_float), ("alpha", C.c_float)]


class Query(C.Structure):
    """``gsx_query``."""
    _fields_ = [("kind", C.c_uint32), ("selection_op", C.c_uint32), ("p0", C.c_float * 2), ("p1", C.c_float * 2),
                ("radius", C.c_float), ("reserved", C.c_uint32)]


class LaunchStats(C.Structure):
    """``gsx_launch_stats``."""
    _fields_ = [(n, C.c_uint64) for n in ("graph_launches", "graph_nodes", "nodes_patched", "direct_launches", "graphs_built", "broken", "idle_direct_scopes")]


class SpecParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("max_std_dev", "cull_margin", "jacobian_clamp", "low_pass", "alpha_max",
                                        "alpha_min", "t_epsilon", "point_radius")]


class RenderOptions(C.Structure):
    _fields_ = [("progressive", C.c_uint32), ("first_slab_divisor", C.c_uint32), ("min_slab", C.c_uint32), ("growth", C.c_uint32),
                ("speculative", C.c_uint32), ("spec_margin", C.c_float), ("spec_radius", C.c_uint32), ("host_verify", C.c_uint32),
                ("frames_in_flight", C.c_uint32), ("slab_shading", C.c_uint32)]


class PlyHeader(C.Structure):
    _fields_ = [("count", C.c_uint64), ("header_bytes", C.c_uint64), ("vertex_bytes", C.c_uint32), ("is_ascii", C.c_uint32),
                ("offsets", C.c_int32 * 62)]


class ShardLayout(C.Structure):
    _fields_ = [("rows_per_rank", C.c_uint32), ("row_lo", C.c_uint32), ("row_hi", C.c_uint32), ("band_bytes", C.c_uint64),
                ("band_offset_bytes", C.c_uint64), ("padded_framebuffer_bytes", C.c_uint64)]


class ViewerDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("stream", C.c_void_p), ("width", C.c_uint32),
                ("height", C.c_uint32)]


class FrameStats(C.Structure):
    _fields_ = [("n_gaussians", C.c_uint64), ("n_visible", C.c_uint64), ("n_tile_entries", C.c_uint64), ("n_sorted", C.c_uint64),
                ("n_repair_tiles", C.c_uint64), ("n_repair_sorted", C.c_uint64), ("speculated", C.c_uint32), ("overflow_slabs", C.c_uint32)]


class ShardVerdict(C.Structure):
    """``gsx_shard_verdict``."""
    _fields_ = [("need_tiles", C.c_uint32), ("overflow", C.c_uint32), ("max_records", C.c_uint32), ("reserved", C.c_uint32)]


class ShardStats(C.Structure):
    """``gsx_shard_stats``."""
    _fields_ = [("frames", C.c_uint64), ("redo_frames", C.c_uint64), ("repair_frames", C.c_uint64), ("exchange_rounds", C.c_uint64),
                ("wire_bytes", C.c_uint64), ("verdict_wait_ns", C.c_uint64), ("last_slot_records", C.c_uint32),
                ("last_repair_slot_records", C.c_uint32), ("last_entries_sum", C.c_uint32), ("last_entries_max", C.c_uint32),
                ("last_work_permille", C.c_uint32), ("redo_fallbacks", C.c_uint32),
                ("last_repair_records", C.c_uint32), ("reserved0", C.c_uint32)]


class CommInfo(C.Structure):
    """``gsx_comm_info``."""
    _fields_ = [("transport", C.c_uint32), ("nranks", C.c_uint32), ("rank", C.c_uint32), ("lane_comms", C.c_uint32),
                ("device", C.c_int32), ("version", C.c_int32)]


#: gsx_comm_all_to_all_fn / gsx_comm_all_gather_fn: (ctx, d_send, d_recv, bytes, hip_stream) -> gsx_status
COMM_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
#: gsx_comm_all_to_all_v_fn(ctx, d_send, send_offsets, send_bytes, d_recv, recv_offsets, recv_bytes, hip_stream)
COMM_A2A_V_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64),
                            C.POINTER(C.c_uint64), C.c_void_p)
#: gsx_comm_gather_v_fn(ctx, d_send, send_bytes, d_recv, recv_offsets, recv_bytes, root, hip_stream)
COMM_GATHER_V_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32, C.c_void_p)


class GsxError(RuntimeError):
    """``gs::Error``: raised for every non-zero ``gsx_status``; ``status`` holds the code."""

    def __init__(self, status: int, message: str):
        super().__init__(f"gsx status {status}: {message}")
        self.status = status


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("GSX_LIB", LIB_PATH)  # development: A/B a differently built library on the same box
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the render path.")
    L = C.CDLL(path)
    vp, u32, u64, f32p, u32p, cp = C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_char_p
    sig = {
        "gsx_last_error_string": ([], C.c_char_p),
        "gsx_abi_version": ([], u32),
        "gsx_spec_params_default": ([C.POINTER(SpecParams)], None),
        "gsx_viewer_create": ([C.POINTER(ViewerDesc), C.POINTER(vp)], C.c_int32),
        "gsx_viewer_destroy": ([vp], None),
        "gsx_viewer_set_spec_params": ([vp, C.POINTER(SpecParams)], C.c_int32),
        "gsx_model_create": ([vp, cp, u64, C.c_int, C.c_int], C.c_int32),
        "gsx_model_remove": ([vp, cp], C.c_int32),
        "gsx_model_len": ([vp, cp, C.POINTER(u64)], C.c_int32),
        "gsx_model_upload_range": ([vp, cp, u64, vp, u64], C.c_int32),
        "gsx_model_upload_pod_device": ([vp, cp, u64, u64, vp, vp, vp, vp], C.c_int32),
        "gsx_update_camera": ([vp, f32p, f32p, u32, u32], C.c_int32),
        "gsx_update_model_transform": ([vp, cp, f32p, f32p, f32p], C.c_int32),
        "gsx_update_gaussian_transform": ([vp, C.c_float, C.c_int, u32, u32], C.c_int32),
        "gsx_model_upload_mask": ([vp, cp, u32p, u64], C.c_int32),
        "gsx_model_download_mask": ([vp, cp, u32p, u64], C.c_int32),
        "gsx_mask_evaluate": ([vp, cp, vp, u32, vp, u32], C.c_int32),
        "gsx_ply_read_header": ([vp, u64, C.POINTER(PlyHeader)], C.c_int32),
        "gsx_ply_read_gaussians": ([vp, u64, C.POINTER(PlyHeader), u64, u64, vp], C.c_int32),
        "gsx_ply_write": ([vp, u64, u32p, vp, vp, u64, C.POINTER(u64)], C.c_int32),
        "gsx_gaussian_edit_default": ([vp], None),
        "gsx_update_query": ([vp, vp], C.c_int32),
        "gsx_update_query_texture": ([vp, vp, u32, u32], C.c_int32),
        "gsx_update_selection_highlight": ([vp, f32p], C.c_int32),
        "gsx_update_selection_edit": ([vp, vp], C.c_int32),
        "gsx_model_show_unedited": ([vp, cp, u32], C.c_int32),
        "gsx_postprocess": ([vp, cp], C.c_int32),
        "gsx_model_upload_selection": ([vp, cp, u32p, u64], C.c_int32),
        "gsx_model_download_selection": ([vp, cp, u32p, u64], C.c_int32),
        "gsx_model_download_edits": ([vp, cp, vp, u64], C.c_int32),
        "gsx_model_upload_edits": ([vp, cp, vp, u64], C.c_int32),
        "gsx_query_download_hits": ([vp, cp, vp, u64, C.POINTER(u64)], C.c_int32),
        "gsx_query_hit_pos_by_closest": ([vp, u64, f32p, f32p, u32, u32, f32p, u32p, f32p], C.c_int32),
        "gsx_query_hit_pos_by_alpha_range": ([vp, u64, f32p, f32p, u32, u32, f32p, C.c_float, u32p, f32p, f32p], C.c_int32),
        "gsx_preprocess": ([vp, cp], C.c_int32),
        "gsx_sort": ([vp, cp], C.c_int32),
        "gsx_sync": ([vp], C.c_int32),
        "gsx_render": ([vp, C.POINTER(cp), u32], C.c_int32),
        "gsx_render_frame": ([vp, C.POINTER(cp), u32], C.c_int32),
        "gsx_download_framebuffer": ([vp, f32p, u64], C.c_int32),
        "gsx_download_rgba8": ([vp, f32p, C.POINTER(C.c_uint8), u64], C.c_int32),
        "gsx_resolve_rgba8_device": ([vp, f32p, u32, u32, vp], C.c_int32),
        "gsx_framebuffer_device_ptr": ([vp, C.POINTER(vp), C.POINTER(u32), C.POINTER(u32)], C.c_int32),
        "gsx_model_frame_stats": ([vp, cp, C.POINTER(FrameStats)], C.c_int32),
        "gsx_model_download_projection": ([vp, cp, u32p, u32p, f32p, f32p, f32p], C.c_int32),
        "gsx_model_download_sorted": ([vp, cp, u32p, u64, C.POINTER(u64)], C.c_int32),
        "gsx_model_download_tile_lists": ([vp, cp, u32p, u64, u32p, u64], C.c_int32),
        "gsx_model_download_pod": ([vp, cp, f32p, u32p, f32p, f32p], C.c_int32),
        "gsx_render_options_default": ([C.POINTER(RenderOptions)], None),
        "gsx_viewer_set_render_options": ([vp, C.POINTER(RenderOptions)], C.c_int32),
        "gsx_shard_layout": ([vp, u32, u32, C.POINTER(ShardLayout)], C.c_int32),
        "gsx_viewer_set_external_framebuffer": ([vp, vp, u64], C.c_int32),
        "gsx_shard_pack": ([vp, cp, u32, vp, vp, u64, C.POINTER(u64)], C.c_int32),
        "gsx_shard_import": ([vp, cp, vp, u64, u32, u32, vp], C.c_int32),
        "gsx_shard_set_windows": ([vp, cp, vp], C.c_int32),
        "gsx_viewer_set_band": ([vp, u32, u32], C.c_int32),
        "gsx_shard_feedback_words": ([vp, u32, C.POINTER(u32)], C.c_int32),
        "gsx_shard_feedback": ([vp, cp, u32, u32, vp], C.c_int32),
        "gsx_render_more": ([vp, C.POINTER(cp), u32], C.c_int32),
        "gsx_debug_set_radix_rank_mode": ([C.c_int32], None),
        "gsx_shard_frame_begin": ([vp, cp, u32, u32, u32, vp], C.c_int32),
        "gsx_shard_slot_records": ([vp, cp, u32, u32, C.POINTER(u32)], C.c_int32),
        "gsx_shard_wait_verdict": ([vp, cp, u32, C.POINTER(ShardVerdict)], C.c_int32),
        "gsx_shard_repair_count": ([vp, cp, u32, vp], C.c_int32),
        "gsx_shard_post_counts": ([vp, u32, vp, C.POINTER(u32)], C.c_int32),
        "gsx_shard_frame_end": ([vp, cp], C.c_int32),
        "gsx_shard_pack_slots": ([vp, cp, u32, u32, vp, u32], C.c_int32),
        "gsx_shard_import_slots": ([vp, cp, vp, u32, u32, u32, u32], C.c_int32),
        "gsx_shard_verify": ([vp, cp, u32, vp, C.POINTER(u32)], C.c_int32),
        "gsx_shard_next_windows": ([vp, cp, u32, vp, C.c_float, u32], C.c_int32),
        "gsx_shard_download_limits": ([vp, cp, u32p, u64], C.c_int32),
        "gsx_comm_unique_id": ([vp], C.c_int32),
        "gsx_viewer_comm_init": ([vp, u32, u32, vp], C.c_int32),
        "gsx_viewer_comm_destroy": ([vp], C.c_int32),
        "gsx_comm_all_to_all": ([vp, vp, vp, u64], C.c_int32),
        "gsx_comm_all_gather": ([vp, vp, vp, u64], C.c_int32),
        "gsx_shard_render_frame": ([vp, cp, u32, u32, C.c_float, u32], C.c_int32),
        "gsx_shard_render_frame_keys": ([vp, C.POINTER(cp), u32, u32p, u32, C.c_float, u32], C.c_int32),
        "gsx_comm_group_create": ([u32, u32, C.POINTER(vp)], C.c_int32),
        "gsx_comm_group_destroy": ([vp], None),
        "gsx_viewer_comm_init_group": ([vp, vp, u32], C.c_int32),
        "gsx_viewer_comm_init_custom": ([vp, u32, u32, COMM_FN, COMM_FN, vp], C.c_int32),
        "gsx_viewer_comm_init_custom_v": ([vp, u32, u32, COMM_A2A_V_FN, COMM_GATHER_V_FN, vp], C.c_int32),
        "gsx_shard_set_band_edges": ([vp, u32, u32p], C.c_int32),
        "gsx_shard_get_band_edges": ([vp, u32, u32p], C.c_int32),
        "gsx_shard_set_balance": ([vp, u32], C.c_int32),
        "gsx_shard_set_limits": ([vp, cp, vp], C.c_int32),
        "gsx_shard_set_slot_records": ([vp, cp, u32], C.c_int32),
        "gsx_shard_set_gather_root": ([vp, C.c_int32], C.c_int32),
        "gsx_shard_get_stats": ([vp, C.POINTER(ShardStats), u32], C.c_int32),
        "gsx_viewer_comm_info": ([vp, C.POINTER(CommInfo)], C.c_int32),
        "gsx_model_buffer_retain": ([vp, cp, C.c_int, C.POINTER(vp)], C.c_int32),
        "gsx_buffer_retain": ([vp], C.c_int32),
        "gsx_buffer_release": ([vp], None),
        "gsx_buffer_len": ([vp, C.POINTER(u64)], C.c_int32),
        "gsx_buffer_download": ([vp, vp, u64], C.c_int32),
        "gsx_debug_set_launch_graphs": ([C.c_int32], None),
        "gsx_debug_launch_count": ([], C.c_uint64),
        "gsx_debug_device_bytes": ([], C.c_uint64),
        "gsx_debug_tile_profile": ([vp, u32p, u64], C.c_int32),
        "gsx_debug_download_lane_framebuffer": ([vp, C.c_uint32, C.c_void_p, u64], C.c_int32),
        "gsx_viewer_launch_stats": ([vp, C.POINTER(LaunchStats), u32], C.c_int32),
        "gsx_set_pass_timing": ([vp, u32], C.c_int32),
        "gsx_get_pass_timing": ([vp, f32p, u32p], C.c_int32),
    }
    assert set(sig) == set(EXPORTS)
    for name, (args, res) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export what gsx.h declares
        fn.argtypes = args
        fn.restype = res
    if L.gsx_abi_version() != GSX_ABI_VERSION:
        raise ImportError(f"libgsx ABI {L.gsx_abi_version()} != binding {GSX_ABI_VERSION}")
    _lib = L
    return L


def check(status: int) -> None:
    if status != GSX_OK:
        raise GsxError(status, load().gsx_last_error_string().decode("utf-8", "replace"))
