//! gsx-sys — raw FFI over `include/gsx.h` (libgsx.so, the MI355X-native 3DGS render path).
//!
//! NOT COMPILED IN THIS REPOSITORY (no Rust toolchain in the build image): this is the binding source a maintainer of
//! LioQing/wgpu-3dgs-viewer-app would add, kept next to the header it mirrors.  The function list is generated from the
//! header by `tools/gen_rust_sys.py` (= what `bindgen` emits); `tests/test_oracle_cpu.py` checks that it covers every
//! exported symbol.  Safe wrappers with the crate's names live in `rust/gsx` (`gs::MultiModelViewer`, ...).
//!
//! build.rs (sketch): `println!("cargo:rustc-link-lib=dylib=gsx"); println!("cargo:rustc-link-search=native={}", env!("GSX_LIB_DIR"));`
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const GSX_ABI_VERSION: u32 = 3;
pub const GSX_TILE: u32 = 16;
pub const GSX_SH_COEFFS: usize = 15;
pub const GSX_RECORD_BYTES: u32 = 48;
pub const GSX_MASK_MAX_OPS: u32 = 64;
pub const GSX_MASK_MAX_SHAPES: u32 = 32;
pub const GSX_QUERY_MAX_HITS: u32 = 65536;
pub const GSX_EDIT_ENABLED: u32 = 1;
pub const GSX_EDIT_HIDDEN: u32 = 2;
pub const GSX_EDIT_OVERRIDE_COLOR: u32 = 4;

pub type gsx_status = i32;
pub const GSX_OK: gsx_status = 0;
pub const GSX_ERR_INVALID_ARG: gsx_status = 1;
pub const GSX_ERR_OOM: gsx_status = 2;
pub const GSX_ERR_HIP: gsx_status = 3;
pub const GSX_ERR_RCCL: gsx_status = 4;
pub const GSX_ERR_IO: gsx_status = 5; // gs::Error::Io
pub const GSX_ERR_PLY: gsx_status = 6;
pub const GSX_ERR_NOT_FOUND: gsx_status = 7;
pub const GSX_ERR_UNSUPPORTED: gsx_status = 8;
pub const GSX_ERR_NO_DEVICE: gsx_status = 9;

/// opaque: `gs::MultiModelViewer<G>` (src/tab/scene.rs:1930)
#[repr(C)]
pub struct gsx_viewer {
    _private: [u8; 0],
}

/// `gs::Gaussian` {rot, pos, color, sh, scale}: field for field, 224 bytes — `&[gs::Gaussian]` crosses the ABI as a pointer
#[repr(C)]
#[derive(Clone, Copy)]
pub struct gsx_gaussian {
    pub rot: [f32; 4], // glam Quat x, y, z, w
    pub pos: [f32; 3],
    pub color: [u8; 4], // UNORM8 r, g, b (0.5 + C0 f_dc), a (sigmoid(opacity))
    pub sh: [[f32; 3]; GSX_SH_COEFFS],
    pub scale: [f32; 3],
}

#[repr(i32)]
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum gsx_sh_kind { Single = 0, Half = 1, Norm8 = 2, None = 3 } // GaussianSh{Single,Half,Norm8,None}Config, src/app.rs:386-403
#[repr(i32)]
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum gsx_cov3d_kind { Single = 0, Half = 1 } // GaussianCov3d{Single,Half}Config, src/app.rs:405-418
#[repr(i32)]
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum gsx_display_mode { Splat = 0, Ellipse = 1, Point = 2 } // gs::GaussianDisplayMode, src/app.rs:1141-1165

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gsx_spec_params {
    pub max_std_dev: f32, pub cull_margin: f32, pub jacobian_clamp: f32, pub low_pass: f32,
    pub alpha_max: f32, pub alpha_min: f32, pub t_epsilon: f32, pub point_radius: f32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct gsx_render_options {
    pub progressive: u32, pub first_slab_divisor: u32, pub min_slab: u32, pub growth: u32,
    pub speculative: u32, pub spec_margin: f32, pub spec_radius: u32, pub host_verify: u32,
    pub frames_in_flight: u32, pub slab_shading: u32,
}
#[repr(C)]
pub struct gsx_viewer_desc { pub abi_version: u32, pub device: i32, pub stream: *mut c_void, pub width: u32, pub height: u32 }
#[repr(C)]
#[derive(Clone, Copy)]
pub struct gsx_mask_shape { pub kind: u32, pub pos: [f32; 3], pub quat_xyzw: [f32; 4], pub scale: [f32; 3] } // gs::MaskOpShapePod
#[repr(C)]
#[derive(Clone, Copy)]
pub struct gsx_mask_op { pub opcode: u32, pub arg: u32 } // postfix MaskOpTree: 0 Shape(arg) 1 Union 2 Intersection 3 Difference 4 SymmetricDifference 5 Complement
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct gsx_frame_stats {
    pub n_gaussians: u64, pub n_visible: u64, pub n_tile_entries: u64, pub n_sorted: u64,
    pub n_repair_tiles: u64, pub n_repair_sorted: u64, pub speculated: u32, pub overflow_slabs: u32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct gsx_gaussian_edit { pub flag: u32, pub color: [f32; 3], pub contrast: f32, pub exposure: f32, pub gamma: f32, pub alpha: f32 } // gs::GaussianEditPod
#[repr(C)]
#[derive(Clone, Copy)]
pub struct gsx_query { pub kind: u32, pub selection_op: u32, pub p0: [f32; 2], pub p1: [f32; 2], pub radius: f32, pub reserved: u32 }
#[repr(C)]
#[derive(Clone, Copy)]
pub struct gsx_query_hit { pub index: u32, pub depth: f32, pub alpha: f32, pub reserved: u32 } // gs::QueryHitResultPod
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct gsx_shard_layout_t {
    pub rows_per_rank: u32, pub row_lo: u32, pub row_hi: u32,
    pub band_bytes: u64, pub band_offset_bytes: u64, pub padded_framebuffer_bytes: u64,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct gsx_shard_verdict { pub need_tiles: u32, pub overflow: u32, pub max_records: u32, pub reserved: u32 }
#[repr(C)]
#[derive(Clone, Copy)]
pub struct gsx_ply_header { pub count: u64, pub header_bytes: u64, pub vertex_bytes: u32, pub is_ascii: u32, pub offsets: [i32; 62] }
/// opaque: an in-process group of viewers, one per GPU (gsx_comm_group_create)
#[repr(C)]
pub struct gsx_comm_group {
    _private: [u8; 0],
}
/// opaque: a ref-counted handle on a snapshot of one of a model's per-Gaussian buffers (gs:: buffers are `Clone`, src/app.rs:769-780)
#[repr(C)]
pub struct gsx_buffer {
    _private: [u8; 0],
}
#[repr(i32)]
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum gsx_buffer_kind { Mask = 0, Edits = 1, Selection = 2 }
/// the two collectives of a caller-supplied transport: they ENQUEUE on `hip_stream` and return 0 or a gsx_status
pub type gsx_comm_all_to_all_fn = Option<unsafe extern "C" fn(ctx: *mut c_void, d_send: *const c_void, d_recv: *mut c_void, bytes_per_peer: u64, hip_stream: *mut c_void) -> gsx_status>;
pub type gsx_comm_all_gather_fn = Option<unsafe extern "C" fn(ctx: *mut c_void, d_send: *const c_void, d_recv: *mut c_void, bytes_per_rank: u64, hip_stream: *mut c_void) -> gsx_status>;
/// ... and of a transport that moves pieces of unequal size (arrays of `world` byte offsets / sizes)
pub type gsx_comm_all_to_all_v_fn = Option<unsafe extern "C" fn(ctx: *mut c_void, d_send: *const c_void, send_offsets: *const u64, send_bytes: *const u64, d_recv: *mut c_void, recv_offsets: *const u64, recv_bytes: *const u64, hip_stream: *mut c_void) -> gsx_status>;
pub type gsx_comm_gather_v_fn = Option<unsafe extern "C" fn(ctx: *mut c_void, d_send: *const c_void, send_bytes: u64, d_recv: *mut c_void, recv_offsets: *const u64, recv_bytes: *const u64, root: i32, hip_stream: *mut c_void) -> gsx_status>;
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct gsx_shard_stats {
    pub frames: u64, pub redo_frames: u64, pub repair_frames: u64, pub exchange_rounds: u64,
    pub wire_bytes: u64, pub verdict_wait_ns: u64, pub last_slot_records: u32, pub last_repair_slot_records: u32,
    pub last_entries_sum: u32, pub last_entries_max: u32, pub last_work_permille: u32, pub redo_fallbacks: u32,
    pub last_repair_records: u32, pub reserved0: u32,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct gsx_comm_info { pub transport: u32, pub nranks: u32, pub rank: u32, pub lane_comms: u32, pub device: i32, pub version: i32 }
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct gsx_launch_stats {
    pub graph_launches: u64, pub graph_nodes: u64, pub nodes_patched: u64, pub direct_launches: u64, pub graphs_built: u64, pub broken: u64, pub idle_direct_scopes: u64,
}
pub type gsx_pass = u32; // 0 project, 1 depth sort, 2 bin, 3 tile sort, 4 composite, 5 project (geometry only), 6 shade
pub const GSX_PASS_COUNT: usize = 7;

#[link(name = "gsx")]
extern "C" {
    pub fn gsx_last_error_string() -> *const c_char;
    pub fn gsx_abi_version() -> u32;
    pub fn gsx_spec_params_default(out: *mut gsx_spec_params);
    pub fn gsx_viewer_create(desc: *const gsx_viewer_desc, out: *mut *mut gsx_viewer) -> gsx_status;
    pub fn gsx_viewer_destroy(v: *mut gsx_viewer);
    pub fn gsx_viewer_set_spec_params(v: *mut gsx_viewer, p: *const gsx_spec_params) -> gsx_status;
    pub fn gsx_render_options_default(out: *mut gsx_render_options);
    pub fn gsx_viewer_set_render_options(v: *mut gsx_viewer, o: *const gsx_render_options) -> gsx_status;
    pub fn gsx_model_create(v: *mut gsx_viewer, key: *const c_char, count: u64, sh: gsx_sh_kind, cov3d: gsx_cov3d_kind) -> gsx_status;
    pub fn gsx_model_remove(v: *mut gsx_viewer, key: *const c_char) -> gsx_status;
    pub fn gsx_model_len(v: *mut gsx_viewer, key: *const c_char, out_count: *mut u64) -> gsx_status;
    pub fn gsx_model_upload_range(v: *mut gsx_viewer, key: *const c_char, start: u64, src: *const gsx_gaussian, n: u64) -> gsx_status;
    pub fn gsx_model_upload_pod_device(v: *mut gsx_viewer, key: *const c_char, start: u64, n: u64, d_pos: *const f32, d_color: *const u32, d_sh: *const f32, d_cov3d: *const f32) -> gsx_status;
    pub fn gsx_update_camera(v: *mut gsx_viewer, view: *const f32, proj: *const f32, width: u32, height: u32) -> gsx_status;
    pub fn gsx_update_model_transform(v: *mut gsx_viewer, key: *const c_char, pos: *const f32, quat_xyzw: *const f32, scale: *const f32) -> gsx_status;
    pub fn gsx_update_gaussian_transform(v: *mut gsx_viewer, size: f32, mode: gsx_display_mode, sh_deg: u32, no_sh0: u32) -> gsx_status;
    pub fn gsx_model_upload_mask(v: *mut gsx_viewer, key: *const c_char, words: *const u32, n_words: u64) -> gsx_status;
    pub fn gsx_model_download_mask(v: *mut gsx_viewer, key: *const c_char, words: *mut u32, n_words: u64) -> gsx_status;
    pub fn gsx_mask_evaluate(v: *mut gsx_viewer, key: *const c_char, ops: *const gsx_mask_op, n_ops: u32, shapes: *const gsx_mask_shape, n_shapes: u32) -> gsx_status;
    pub fn gsx_preprocess(v: *mut gsx_viewer, key: *const c_char) -> gsx_status;
    pub fn gsx_sort(v: *mut gsx_viewer, key: *const c_char) -> gsx_status;
    pub fn gsx_sync(v: *mut gsx_viewer) -> gsx_status;
    pub fn gsx_render(v: *mut gsx_viewer, keys_far_to_near: *const *const c_char, n_keys: u32) -> gsx_status;
    pub fn gsx_render_frame(v: *mut gsx_viewer, keys_far_to_near: *const *const c_char, n_keys: u32) -> gsx_status;
    pub fn gsx_download_framebuffer(v: *mut gsx_viewer, rgbt: *mut f32, n_floats: u64) -> gsx_status;
    pub fn gsx_download_rgba8(v: *mut gsx_viewer, background_rgb: *const f32, rgba: *mut u8, n_bytes: u64) -> gsx_status;
    pub fn gsx_framebuffer_device_ptr(v: *mut gsx_viewer, out_ptr: *mut *mut c_void, out_w: *mut u32, out_h: *mut u32) -> gsx_status;
    pub fn gsx_model_frame_stats(v: *mut gsx_viewer, key: *const c_char, out: *mut gsx_frame_stats) -> gsx_status;
    pub fn gsx_model_download_projection(v: *mut gsx_viewer, key: *const c_char, depth_key: *mut u32, rect: *mut u32, mean2d: *mut f32, conic_opacity: *mut f32, rgb: *mut f32) -> gsx_status;
    pub fn gsx_model_download_sorted(v: *mut gsx_viewer, key: *const c_char, indices: *mut u32, capacity: u64, out_n_visible: *mut u64) -> gsx_status;
    pub fn gsx_model_download_tile_lists(v: *mut gsx_viewer, key: *const c_char, tile_offsets: *mut u32, n_offsets: u64, list: *mut u32, capacity: u64) -> gsx_status;
    pub fn gsx_model_download_pod(v: *mut gsx_viewer, key: *const c_char, pos: *mut f32, color: *mut u32, sh: *mut f32, cov3d: *mut f32) -> gsx_status;
    pub fn gsx_gaussian_edit_default(e: *mut gsx_gaussian_edit);
    pub fn gsx_update_query(v: *mut gsx_viewer, q: *const gsx_query) -> gsx_status;
    pub fn gsx_update_query_texture(v: *mut gsx_viewer, texels: *const u8, width: u32, height: u32) -> gsx_status;
    pub fn gsx_update_selection_highlight(v: *mut gsx_viewer, rgba: *const f32) -> gsx_status;
    pub fn gsx_update_selection_edit(v: *mut gsx_viewer, e: *const gsx_gaussian_edit) -> gsx_status;
    pub fn gsx_model_show_unedited(v: *mut gsx_viewer, key: *const c_char, on: u32) -> gsx_status;
    pub fn gsx_postprocess(v: *mut gsx_viewer, key: *const c_char) -> gsx_status;
    pub fn gsx_model_upload_selection(v: *mut gsx_viewer, key: *const c_char, words: *const u32, n_words: u64) -> gsx_status;
    pub fn gsx_model_download_selection(v: *mut gsx_viewer, key: *const c_char, words: *mut u32, n_words: u64) -> gsx_status;
    pub fn gsx_model_download_edits(v: *mut gsx_viewer, key: *const c_char, out: *mut gsx_gaussian_edit, n: u64) -> gsx_status;
    pub fn gsx_model_upload_edits(v: *mut gsx_viewer, key: *const c_char, edits: *const gsx_gaussian_edit, n: u64) -> gsx_status;
    pub fn gsx_query_download_hits(v: *mut gsx_viewer, key: *const c_char, out: *mut gsx_query_hit, capacity: u64, out_n: *mut u64) -> gsx_status;
    pub fn gsx_query_hit_pos_by_closest(hits: *const gsx_query_hit, n: u64, view: *const f32, proj: *const f32, width: u32, height: u32, coords: *const f32, out_index: *mut u32, out_pos: *mut f32) -> gsx_status;
    pub fn gsx_query_hit_pos_by_alpha_range(hits: *const gsx_query_hit, n: u64, view: *const f32, proj: *const f32, width: u32, height: u32, coords: *const f32, range: f32, out_index: *mut u32, out_alpha: *mut f32, out_pos: *mut f32) -> gsx_status;
    pub fn gsx_model_buffer_retain(v: *mut gsx_viewer, key: *const c_char, kind: gsx_buffer_kind, out: *mut *mut gsx_buffer) -> gsx_status;
    pub fn gsx_buffer_retain(b: *mut gsx_buffer) -> gsx_status;
    pub fn gsx_buffer_release(b: *mut gsx_buffer);
    pub fn gsx_buffer_len(b: *mut gsx_buffer, out_elements: *mut u64) -> gsx_status;
    pub fn gsx_buffer_download(b: *mut gsx_buffer, out: *mut c_void, n_elements: u64) -> gsx_status;
    pub fn gsx_shard_layout(v: *mut gsx_viewer, world: u32, rank: u32, out: *mut gsx_shard_layout_t) -> gsx_status;
    pub fn gsx_viewer_set_band(v: *mut gsx_viewer, row_lo: u32, row_hi: u32) -> gsx_status;
    pub fn gsx_viewer_set_external_framebuffer(v: *mut gsx_viewer, d_ptr: *mut c_void, bytes: u64) -> gsx_status;
    pub fn gsx_resolve_rgba8_device(v: *mut gsx_viewer, background_rgb: *const f32, y0: u32, y1: u32, d_rgba: *mut c_void) -> gsx_status;
    pub fn gsx_shard_pack(v: *mut gsx_viewer, key: *const c_char, world: u32, d_tile_window: *const u32, d_send: *mut c_void, capacity_records: u64, counts: *mut u64) -> gsx_status;
    pub fn gsx_shard_set_windows(v: *mut gsx_viewer, key: *const c_char, d_tile_window: *const u32) -> gsx_status;
    pub fn gsx_shard_import(v: *mut gsx_viewer, key: *const c_char, d_recv: *const c_void, n_records: u64, world: u32, rank: u32, d_tile_window: *const u32) -> gsx_status;
    pub fn gsx_shard_feedback_words(v: *mut gsx_viewer, world: u32, out_words: *mut u32) -> gsx_status;
    pub fn gsx_shard_feedback(v: *mut gsx_viewer, key: *const c_char, world: u32, rank: u32, d_out_u32: *mut c_void) -> gsx_status;
    pub fn gsx_render_more(v: *mut gsx_viewer, keys: *const *const c_char, n_keys: u32) -> gsx_status;
    pub fn gsx_shard_frame_begin(v: *mut gsx_viewer, key: *const c_char, world: u32, rank: u32, speculate: u32, d_limit_override: *const u32) -> gsx_status;
    pub fn gsx_shard_slot_records(v: *mut gsx_viewer, key: *const c_char, world: u32, shard_records_max: u32, out_records: *mut u32) -> gsx_status;
    pub fn gsx_shard_pack_slots(v: *mut gsx_viewer, key: *const c_char, world: u32, round: u32, d_send: *mut c_void, slot_records: u32) -> gsx_status;
    pub fn gsx_shard_import_slots(v: *mut gsx_viewer, key: *const c_char, d_recv: *const c_void, world: u32, rank: u32, round: u32, slot_records: u32) -> gsx_status;
    pub fn gsx_shard_verify(v: *mut gsx_viewer, key: *const c_char, world: u32, d_sat_all: *const c_void, out_seq: *mut u32) -> gsx_status;
    pub fn gsx_shard_wait_verdict(v: *mut gsx_viewer, key: *const c_char, seq: u32, out: *mut gsx_shard_verdict) -> gsx_status;
    pub fn gsx_shard_repair_count(v: *mut gsx_viewer, key: *const c_char, world: u32, d_out4: *mut c_void) -> gsx_status;
    pub fn gsx_shard_post_counts(v: *mut gsx_viewer, world: u32, d_counts_all: *const c_void, out_seq: *mut u32) -> gsx_status;
    pub fn gsx_shard_next_windows(v: *mut gsx_viewer, key: *const c_char, world: u32, d_sat_all: *const c_void, margin: f32, radius: u32) -> gsx_status;
    pub fn gsx_shard_frame_end(v: *mut gsx_viewer, key: *const c_char) -> gsx_status;
    pub fn gsx_shard_download_limits(v: *mut gsx_viewer, key: *const c_char, limits: *mut u32, n_words: u64) -> gsx_status;
    pub fn gsx_comm_unique_id(out_id: *mut u8) -> gsx_status;
    pub fn gsx_viewer_comm_init(v: *mut gsx_viewer, world: u32, rank: u32, id: *const u8) -> gsx_status;
    pub fn gsx_viewer_comm_destroy(v: *mut gsx_viewer) -> gsx_status;
    pub fn gsx_viewer_comm_info(v: *mut gsx_viewer, out: *mut gsx_comm_info) -> gsx_status;
    pub fn gsx_comm_all_to_all(v: *mut gsx_viewer, d_send: *const c_void, d_recv: *mut c_void, bytes_per_peer: u64) -> gsx_status;
    pub fn gsx_comm_all_gather(v: *mut gsx_viewer, d_send: *const c_void, d_recv: *mut c_void, bytes_per_rank: u64) -> gsx_status;
    pub fn gsx_shard_render_frame(v: *mut gsx_viewer, key: *const c_char, shard_records_max: u32, speculate: u32, margin: f32, radius: u32) -> gsx_status;
    pub fn gsx_shard_render_frame_keys(v: *mut gsx_viewer, keys_far_to_near: *const *const c_char, n_keys: u32, shard_records_max: *const u32, speculate: u32, margin: f32, radius: u32) -> gsx_status;
    pub fn gsx_comm_group_create(world: u32, timeout_ms: u32, out: *mut *mut gsx_comm_group) -> gsx_status;
    pub fn gsx_comm_group_destroy(g: *mut gsx_comm_group);
    pub fn gsx_viewer_comm_init_group(v: *mut gsx_viewer, g: *mut gsx_comm_group, rank: u32) -> gsx_status;
    pub fn gsx_viewer_comm_init_custom(v: *mut gsx_viewer, world: u32, rank: u32, all_to_all: gsx_comm_all_to_all_fn, all_gather: gsx_comm_all_gather_fn, ctx: *mut c_void) -> gsx_status;
    pub fn gsx_viewer_comm_init_custom_v(v: *mut gsx_viewer, world: u32, rank: u32, all_to_all_v: gsx_comm_all_to_all_v_fn, gather_v: gsx_comm_gather_v_fn, ctx: *mut c_void) -> gsx_status;
    pub fn gsx_shard_set_band_edges(v: *mut gsx_viewer, world: u32, edges: *const u32) -> gsx_status;
    pub fn gsx_shard_get_band_edges(v: *mut gsx_viewer, world: u32, out_edges: *mut u32) -> gsx_status;
    pub fn gsx_shard_set_balance(v: *mut gsx_viewer, enabled: u32) -> gsx_status;
    pub fn gsx_shard_set_limits(v: *mut gsx_viewer, key: *const c_char, limits: *const u32) -> gsx_status;
    pub fn gsx_shard_set_slot_records(v: *mut gsx_viewer, key: *const c_char, records: u32) -> gsx_status;
    pub fn gsx_shard_get_stats(v: *mut gsx_viewer, out: *mut gsx_shard_stats, reset: u32) -> gsx_status;
    pub fn gsx_shard_set_gather_root(v: *mut gsx_viewer, root: i32) -> gsx_status;
    pub fn gsx_ply_read_header(data: *const c_void, size: u64, out: *mut gsx_ply_header) -> gsx_status;
    pub fn gsx_ply_read_gaussians(data: *const c_void, size: u64, header: *const gsx_ply_header, start: u64, n: u64, out: *mut gsx_gaussian) -> gsx_status;
    pub fn gsx_ply_write(gaussians: *const gsx_gaussian, n: u64, mask_words: *const u32, edits: *const gsx_gaussian_edit, out: *mut c_void, capacity: u64, out_size: *mut u64) -> gsx_status;
    pub fn gsx_debug_set_radix_rank_mode(mode: i32);
    pub fn gsx_debug_set_launch_graphs(enabled: i32);
    pub fn gsx_debug_launch_count() -> u64;
    pub fn gsx_debug_device_bytes() -> u64;
    pub fn gsx_debug_download_lane_framebuffer(v: *mut gsx_viewer, lane: u32, rgbt: *mut f32, n_floats: u64) -> gsx_status;
    pub fn gsx_debug_tile_profile(v: *mut gsx_viewer, out4: *mut u32, n_tiles: u64) -> gsx_status;
    pub fn gsx_viewer_launch_stats(v: *mut gsx_viewer, out: *mut gsx_launch_stats, reset: u32) -> gsx_status;
    pub fn gsx_set_pass_timing(v: *mut gsx_viewer, enabled: u32) -> gsx_status;
    pub fn gsx_get_pass_timing(v: *mut gsx_viewer, ms: *mut f32, launches: *mut u32) -> gsx_status;
}
