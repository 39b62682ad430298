#!/usr/bin/env python3
"""Writes rust/gsx-sys/src/lib.rs from include/gsx.h: one `extern "C"` declaration per exported function, in header order
(what `bindgen include/gsx.h` would give a maintainer; there is no Rust toolchain in this image, so nothing here is compiled —
tests/test_oracle_cpu.py only checks that every declared symbol of the header appears).  Structs / enums are written by hand
below because their field comments matter more than their mechanics.  usage: python tools/gen_rust_sys.py"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "include", "gsx.h")).read()
code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
protos = re.findall(r"^(gsx_status|void|uint32_t|uint64_t|const char\*)\s+(gsx_\w+)\s*\(([^;]*?)\);", code, flags=re.M | re.S)

SCALAR = {"uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "float": "f32", "char": "c_char", "void": "c_void",
          "gsx_sh_kind": "gsx_sh_kind", "gsx_cov3d_kind": "gsx_cov3d_kind", "gsx_display_mode": "gsx_display_mode", "gsx_status": "gsx_status", "gsx_buffer_kind": "gsx_buffer_kind"}


def rust_type(ctype: str) -> str:
    t = ctype.strip()
    m = re.match(r"^(const\s+)?([\w ]+?)\s*((?:\*\s*(?:const\s*)?)*)$", t)
    assert m, ctype
    const, base, stars = bool(m.group(1)), m.group(2).strip(), m.group(3).replace(" ", "").replace("const", "")
    rt = SCALAR.get(base, base)
    for k, _ in enumerate(stars):
        inner_const = const if k == 0 else ("const*" in m.group(3).replace(" ", "") and k == len(stars) - 1)
        rt = ("*const " if (const if k == 0 else "const" in m.group(3)) else "*mut ") + rt
    return rt


def param(p: str):
    p = " ".join(p.split())
    if p == "void":
        return None
    m = re.match(r"^(.*?)(\w+)\s*(\[\w*\])?$", p)
    ctype, name, arr = m.group(1), m.group(2), m.group(3)
    if arr:  # array parameters decay to pointers
        ctype = ctype.strip() + "*"
    if name in ("type", "ref", "in", "box"):
        name += "_"
    return f"{name}: {rust_type(ctype)}"


out = []
for ret, name, args in protos:
    ps = [x for x in (param(a) for a in args.split(",")) if x]
    r = {"gsx_status": " -> gsx_status", "void": "", "uint32_t": " -> u32", "uint64_t": " -> u64", "const char*": " -> *const c_char"}[ret]
    out.append(f"    pub fn {name}({', '.join(ps)}){r};")

HEAD = '''//! gsx-sys — raw FFI over `include/gsx.h` (libgsx.so, the MI355X-native 3DGS render path).
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
'''
path = os.path.join(ROOT, "rust", "gsx-sys", "src", "lib.rs")
with open(path, "w") as f:
    f.write(HEAD + "\n".join(out) + "\n}\n")
print(f"{len(out)} functions -> {path}")
