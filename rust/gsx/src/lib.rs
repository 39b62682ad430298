//! gsx — the `gs::` facade over libgsx: the names and call shapes `LioQing/wgpu-3dgs-viewer-app` already uses for its
//! render path (`use wgpu_3dgs_viewer as gs;`), so that `src/tab/scene.rs` / `src/app.rs` change in imports only.
//!
//! NOT COMPILED IN THIS REPOSITORY (no Rust toolchain in the build image).  It mirrors, name for name, what is compiled and
//! tested here in two other host languages: `include/gsx.hpp` (C++ `gs::`, exercised by `tools/frame_driver.cpp`) and
//! `wgpu_3dgs_viewer_app_amd/viewer.py` (what the parity tests drive).  Every method cites the app call site it serves.
//!
//! What does NOT carry over: wgpu `Device` / `Queue` / `CommandEncoder` / bind groups.  libgsx enqueues on its own HIP stream;
//! the arguments stay in the signatures (ignored) so that call sites compile unchanged, and `device.poll(Maintain::Wait)`
//! becomes `viewer.poll()`.
use gsx_sys as sys;
use std::collections::HashMap;
use std::ffi::{CStr, CString};
use std::marker::PhantomData;

pub use sys::gsx_gaussian as Gaussian; // rot: Quat(xyzw), pos, color: U8Vec4, sh: [Vec3; 15], scale — field for field
pub use sys::gsx_gaussian_edit as GaussianEditPod;
pub use sys::gsx_query_hit as QueryHitResultPod;

/// `gs::Error` (app.rs:548, scene.rs:234): every non-zero status, with the library's thread-local message.
#[derive(Debug)]
pub enum Error {
    Io(String),
    Gsx { status: i32, message: String },
}
impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        match self {
            Error::Io(m) => write!(f, "io: {m}"),
            Error::Gsx { status, message } => write!(f, "gsx status {status}: {message}"),
        }
    }
}
fn check(status: sys::gsx_status) -> Result<(), Error> {
    if status == sys::GSX_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(sys::gsx_last_error_string()) }.to_string_lossy().into_owned();
    Err(if status == sys::GSX_ERR_IO { Error::Io(message) } else { Error::Gsx { status, message } })
}

/// The pod generic `G: gs::GaussianPod` of the app's 8-way dispatch (scene.rs:23-81, app.rs:352-418) becomes two enums.
pub trait GaussianPod {
    const SH: sys::gsx_sh_kind;
    const COV3D: sys::gsx_cov3d_kind;
}
macro_rules! pod {
    ($name:ident, $sh:ident, $cov:ident) => {
        pub struct $name;
        impl GaussianPod for $name {
            const SH: sys::gsx_sh_kind = sys::gsx_sh_kind::$sh;
            const COV3D: sys::gsx_cov3d_kind = sys::gsx_cov3d_kind::$cov;
        }
    };
}
pod!(GaussianPodWithShSingleCov3dSingleConfigs, Single, Single);
pod!(GaussianPodWithShSingleCov3dHalfConfigs, Single, Half);
pod!(GaussianPodWithShHalfCov3dSingleConfigs, Half, Single);
pod!(GaussianPodWithShHalfCov3dHalfConfigs, Half, Half);
pod!(GaussianPodWithShNorm8Cov3dSingleConfigs, Norm8, Single);
pod!(GaussianPodWithShNorm8Cov3dHalfConfigs, Norm8, Half); // the app's default (app.rs:398-417)
pod!(GaussianPodWithShNoneCov3dSingleConfigs, None, Single);
pod!(GaussianPodWithShNoneCov3dHalfConfigs, None, Half);

#[derive(Clone, Copy, PartialEq, Eq)]
pub enum GaussianDisplayMode { Splat = 0, Ellipse = 1, Point = 2 }
/// `GaussianShDegree::new` returns `None` above 3 (transform.rs:139).
#[derive(Clone, Copy)]
pub struct GaussianShDegree(u32);
impl GaussianShDegree {
    pub fn new(deg: u32) -> Option<Self> { (deg <= 3).then_some(Self(deg)) }
    pub fn degree(&self) -> u32 { self.0 }
}

/// What crosses the ABI of a camera: its two matrices (`CameraTrait`, app.rs:1236-1244, 1329-1343; glam column-major).
pub trait CameraTrait {
    fn view(&self) -> [f32; 16];
    fn projection(&self, aspect_ratio: f32) -> [f32; 16];
}

struct Handle(*mut sys::gsx_viewer);
impl Drop for Handle {
    fn drop(&mut self) { unsafe { sys::gsx_viewer_destroy(self.0) } }
}

/// `gs::GaussiansBuffer<G>`: `update_range(&queue, start, &[Gaussian])` (scene.rs:2083-2084), `len()` (scene.rs:608, 862).
pub struct GaussiansBuffer { v: *mut sys::gsx_viewer, key: CString }
impl GaussiansBuffer {
    pub fn update_range<Q>(&self, _queue: &Q, start: usize, gaussians: &[Gaussian]) {
        check(unsafe { sys::gsx_model_upload_range(self.v, self.key.as_ptr(), start as u64, gaussians.as_ptr(), gaussians.len() as u64) })
            .expect("update_range"); // infallible in the crate's signature: panics like the reference (scene.rs:2078-2081)
    }
    pub fn len(&self) -> usize {
        let mut n = 0u64;
        check(unsafe { sys::gsx_model_len(self.v, self.key.as_ptr(), &mut n) }).expect("len");
        n as usize
    }
}
/// What `buffer.clone()` gives in the app (app.rs:769-780: the clones are moved into spawned download threads): a handle on a
/// device-side snapshot of the buffer taken at clone time.  `Clone` adds a reference, `Drop` releases it, `download` may run on
/// any thread while the viewer renders (`gsx_buffer_*` in include/gsx.h).
pub struct BufferClone<T> { h: *mut sys::gsx_buffer, _t: std::marker::PhantomData<T> }
unsafe impl<T: Send> Send for BufferClone<T> {}
impl<T> Clone for BufferClone<T> {
    fn clone(&self) -> Self { unsafe { sys::gsx_buffer_retain(self.h) }; Self { h: self.h, _t: std::marker::PhantomData } }
}
impl<T> Drop for BufferClone<T> { fn drop(&mut self) { unsafe { sys::gsx_buffer_release(self.h) } } }
impl<T: Copy> BufferClone<T> {
    fn retain(v: *mut sys::gsx_viewer, key: &CString, kind: sys::gsx_buffer_kind) -> Result<Self, Error> {
        let mut h = std::ptr::null_mut();
        check(unsafe { sys::gsx_model_buffer_retain(v, key.as_ptr(), kind, &mut h) })?;
        Ok(Self { h, _t: std::marker::PhantomData })
    }
    pub async fn download<D, Q>(&self, _device: &D, _queue: &Q) -> Result<Vec<T>, Error> {
        let mut n = 0u64;
        check(unsafe { sys::gsx_buffer_len(self.h, &mut n) })?;
        let mut out: Vec<T> = Vec::with_capacity(n as usize);
        check(unsafe { sys::gsx_buffer_download(self.h, out.as_mut_ptr() as *mut std::os::raw::c_void, n) })?;
        unsafe { out.set_len(n as usize) };
        Ok(out)
    }
}
/// `gs::MaskBuffer` / `gs::SelectionBuffer` / `gs::GaussiansEditBuffer`: `download::<T>(&device, &queue).await` (app.rs:789, 806).
pub struct MaskBuffer { v: *mut sys::gsx_viewer, key: CString, words: usize }
impl MaskBuffer {
    pub async fn download<D, Q>(&self, _device: &D, _queue: &Q) -> Result<Vec<u32>, Error> {
        let mut w = vec![0u32; self.words];
        check(unsafe { sys::gsx_model_download_mask(self.v, self.key.as_ptr(), w.as_mut_ptr(), w.len() as u64) })?;
        Ok(w)
    }
    /// `mask_buffer.clone()` (app.rs:775)
    pub fn clone_handle(&self) -> Result<BufferClone<u32>, Error> { BufferClone::retain(self.v, &self.key, sys::gsx_buffer_kind::Mask) }
}
pub struct GaussiansEditBuffer { v: *mut sys::gsx_viewer, key: CString, n: usize }
impl GaussiansEditBuffer {
    pub async fn download<D, Q>(&self, _device: &D, _queue: &Q) -> Result<Vec<GaussianEditPod>, Error> {
        let mut e = Vec::with_capacity(self.n);
        check(unsafe { sys::gsx_model_download_edits(self.v, self.key.as_ptr(), e.as_mut_ptr(), self.n as u64) })?;
        unsafe { e.set_len(self.n) };
        Ok(e)
    }
    /// `gaussians_edit_buffer.clone()` (app.rs:772)
    pub fn clone_handle(&self) -> Result<BufferClone<GaussianEditPod>, Error> { BufferClone::retain(self.v, &self.key, sys::gsx_buffer_kind::Edits) }
}
/// `gs::MultiModelViewerGaussianBuffers<G>::new_empty(&device, count)` (scene.rs:2111-2112)
pub struct MultiModelViewerGaussianBuffers {
    pub gaussians_buffer: GaussiansBuffer,
    pub mask_buffer: MaskBuffer,
    pub gaussians_edit_buffer: GaussiansEditBuffer,
}
/// `gs::MultiModelViewerModel { gaussian_buffers, bind_groups }` (scene.rs:2133-2139); bind groups have no counterpart.
pub struct MultiModelViewerModel { pub gaussian_buffers: MultiModelViewerGaussianBuffers }

pub struct Preprocessor(*mut sys::gsx_viewer);
impl Preprocessor {
    /// `preprocessor.preprocess(&mut encoder, &bind_group, len)` (scene.rs:856-863): the model is named by its key.
    pub fn preprocess(&self, key: &str) { let k = CString::new(key).unwrap(); check(unsafe { sys::gsx_preprocess(self.0, k.as_ptr()) }).expect("preprocess") }
}
pub struct RadixSorter(*mut sys::gsx_viewer);
impl RadixSorter {
    /// `radix_sorter.sort(&mut encoder, &bind_group, &indirect_args)` (scene.rs:865-869)
    pub fn sort(&self, key: &str) { let k = CString::new(key).unwrap(); check(unsafe { sys::gsx_sort(self.0, k.as_ptr()) }).expect("sort") }
}
pub struct Renderer(*mut sys::gsx_viewer);
impl Renderer {
    /// the `render_with_pass` loop over `model_render_keys`, far -> near (scene.rs:2302-2314), as one call
    pub fn render(&self, model_render_keys: &[String]) {
        let keys: Vec<CString> = model_render_keys.iter().map(|k| CString::new(k.as_str()).unwrap()).collect();
        let ptrs: Vec<*const std::os::raw::c_char> = keys.iter().map(|k| k.as_ptr()).collect();
        check(unsafe { sys::gsx_render(self.0, ptrs.as_ptr(), ptrs.len() as u32) }).expect("render")
    }
}
pub struct Postprocessor(*mut sys::gsx_viewer);
impl Postprocessor {
    /// `postprocessor.postprocess(&mut encoder, bg0, bg1, len, &indirect_args)` (scene.rs:601-611)
    pub fn postprocess(&self, key: &str) { let k = CString::new(key).unwrap(); check(unsafe { sys::gsx_postprocess(self.0, k.as_ptr()) }).expect("postprocess") }
}

/// `gs::MultiModelViewer<G>` (scene.rs:1930): public fields as the app reads them (scene.rs:2115-2120, 2133-2139).
pub struct MultiModelViewer<G: GaussianPod> {
    handle: Handle,
    pub models: HashMap<String, MultiModelViewerModel>,
    pub preprocessor: Preprocessor,
    pub radix_sorter: RadixSorter,
    pub renderer: Renderer,
    pub postprocessor: Postprocessor,
    _pod: PhantomData<G>,
}
impl<G: GaussianPod> MultiModelViewer<G> {
    /// `MultiModelViewer::new_with(&device, target_format, depth_stencil, uvec2(1, 1))` (scene.rs:1969-1980); format and depth
    /// state do not apply (the result is an (rgb, T) float image).
    pub fn new_with<D, F, S>(_device: &D, _format: F, _depth_stencil: Option<S>, size: (u32, u32)) -> Result<Self, Error> {
        let desc = sys::gsx_viewer_desc { abi_version: sys::GSX_ABI_VERSION, device: 0, stream: std::ptr::null_mut(), width: size.0, height: size.1 };
        let mut v = std::ptr::null_mut();
        check(unsafe { sys::gsx_viewer_create(&desc, &mut v) })?;
        Ok(Self { handle: Handle(v), models: HashMap::new(), preprocessor: Preprocessor(v), radix_sorter: RadixSorter(v), renderer: Renderer(v),
                  postprocessor: Postprocessor(v), _pod: PhantomData })
    }
    /// `GaussianBuffers::new_empty(&device, count)` + `BindGroups::new(..)` + `models.insert(key, ..)` (scene.rs:2111-2139)
    pub fn insert_model(&mut self, key: &str, count: usize) -> Result<(), Error> {
        let k = CString::new(key).unwrap();
        check(unsafe { sys::gsx_model_create(self.handle.0, k.as_ptr(), count as u64, G::SH, G::COV3D) })?;
        let v = self.handle.0;
        self.models.insert(key.to_owned(), MultiModelViewerModel { gaussian_buffers: MultiModelViewerGaussianBuffers {
            gaussians_buffer: GaussiansBuffer { v, key: k.clone() },
            mask_buffer: MaskBuffer { v, key: k.clone(), words: (count + 31) / 32 },
            gaussians_edit_buffer: GaussiansEditBuffer { v, key: k, n: count },
        } });
        Ok(())
    }
    /// `viewer.remove_model(&key)` (scene.rs:2176)
    pub fn remove_model(&mut self, key: &str) {
        let k = CString::new(key).unwrap();
        check(unsafe { sys::gsx_model_remove(self.handle.0, k.as_ptr()) }).expect("remove_model");
        self.models.remove(key);
    }
    /// `viewer.update_camera(&queue, &impl CameraTrait, uvec2 size)` (scene.rs:795)
    pub fn update_camera<Q>(&mut self, _queue: &Q, camera: &impl CameraTrait, size: (u32, u32)) {
        let (view, proj) = (camera.view(), camera.projection(size.0 as f32 / size.1 as f32));
        check(unsafe { sys::gsx_update_camera(self.handle.0, view.as_ptr(), proj.as_ptr(), size.0, size.1) }).expect("update_camera")
    }
    /// `viewer.update_model_transform(&queue, key, pos, quat, scale)` (scene.rs:796-802); quat = glam x, y, z, w
    pub fn update_model_transform<Q>(&mut self, _queue: &Q, key: &str, pos: [f32; 3], quat: [f32; 4], scale: [f32; 3]) {
        let k = CString::new(key).unwrap();
        check(unsafe { sys::gsx_update_model_transform(self.handle.0, k.as_ptr(), pos.as_ptr(), quat.as_ptr(), scale.as_ptr()) }).expect("update_model_transform")
    }
    /// `viewer.update_gaussian_transform(&queue, size, display_mode, sh_deg, no_sh0)` (scene.rs:803-809)
    pub fn update_gaussian_transform<Q>(&mut self, _queue: &Q, size: f32, mode: GaussianDisplayMode, sh_deg: GaussianShDegree, no_sh0: bool) {
        let m = match mode { GaussianDisplayMode::Splat => sys::gsx_display_mode::Splat, GaussianDisplayMode::Ellipse => sys::gsx_display_mode::Ellipse,
                             GaussianDisplayMode::Point => sys::gsx_display_mode::Point };
        check(unsafe { sys::gsx_update_gaussian_transform(self.handle.0, size, m, sh_deg.degree(), no_sh0 as u32) }).expect("update_gaussian_transform")
    }
    /// `viewer.update_query(&queue, &pod)` (scene.rs:785)
    pub fn update_query<Q>(&mut self, _queue: &Q, query: &sys::gsx_query) { check(unsafe { sys::gsx_update_query(self.handle.0, query) }).expect("update_query") }
    /// `viewer.update_selection_highlight(&queue, vec4)` (scene.rs:816-829)
    pub fn update_selection_highlight<Q>(&mut self, _queue: &Q, rgba: [f32; 4]) {
        check(unsafe { sys::gsx_update_selection_highlight(self.handle.0, rgba.as_ptr()) }).expect("update_selection_highlight")
    }
    /// `viewer.update_selection_edit_with_pod(&queue, &pod)` (scene.rs:815, 821, 848)
    pub fn update_selection_edit_with_pod<Q>(&mut self, _queue: &Q, pod: &GaussianEditPod) {
        check(unsafe { sys::gsx_update_selection_edit(self.handle.0, pod) }).expect("update_selection_edit_with_pod")
    }
    /// `mask_evaluator.evaluate(&device, &queue, &MaskOpTree, &mask_buffer, &model_transform_buffer, &gaussians_buffer)` (scene.rs:2124-2131,
    /// 2201-2209): the tree in postfix order (`MaskOpTree::Reset` = no ops), evaluated on the GPU, no readback.
    pub fn evaluate_mask(&mut self, key: &str, ops: &[sys::gsx_mask_op], shapes: &[sys::gsx_mask_shape]) -> Result<(), Error> {
        let k = CString::new(key).unwrap();
        check(unsafe { sys::gsx_mask_evaluate(self.handle.0, k.as_ptr(), ops.as_ptr(), ops.len() as u32, shapes.as_ptr(), shapes.len() as u32) })
    }
    /// `queue.submit(..); device.poll(wgpu::Maintain::Wait)` (scene.rs:613-614, 872-873)
    pub fn poll(&self) { check(unsafe { sys::gsx_sync(self.handle.0) }).expect("poll") }
    /// what `SceneCallback::paint` blits instead of the `render_with_pass` loop
    pub fn download_rgba8(&self, background: [f32; 3], width: u32, height: u32) -> Result<Vec<u8>, Error> {
        let mut px = vec![0u8; 4 * width as usize * height as usize];
        check(unsafe { sys::gsx_download_rgba8(self.handle.0, background.as_ptr(), px.as_mut_ptr(), px.len() as u64) })?;
        Ok(px)
    }

    // ---- multi-GPU (no counterpart in the reference: src/main.rs:85-98 opens one wgpu device) ----
    /// One rank of an index-sharded viewer: `id` from `comm_unique_id()` on rank 0, carried to the other ranks by the host.
    pub fn comm_init(&mut self, world: u32, rank: u32, id: &[u8; 128]) -> Result<(), Error> { check(unsafe { sys::gsx_viewer_comm_init(self.handle.0, world, rank, id.as_ptr()) }) }
    /// One whole index-sharded frame of `key` (include/gsx.h "device-resident protocol"): afterwards every rank holds the frame.
    pub fn shard_render_frame(&mut self, key: &str, shard_records_max: u32, speculate: bool, margin: f32, radius: u32) -> Result<(), Error> {
        let k = CString::new(key).unwrap();
        check(unsafe { sys::gsx_shard_render_frame(self.handle.0, k.as_ptr(), shard_records_max, speculate as u32, margin, radius) })
    }
}
impl<G: GaussianPod> MultiModelViewer<G> {
    /// Layered models far -> near like `renderer.render` (scene.rs:2302-2314), every model index-sharded over the ranks.
    pub fn shard_render_frame_keys(&mut self, model_render_keys: &[String], shard_records_max: &[u32], speculate: bool, margin: f32, radius: u32) -> Result<(), Error> {
        let ks: Vec<CString> = model_render_keys.iter().map(|k| CString::new(k.as_str()).unwrap()).collect();
        let ps: Vec<*const std::os::raw::c_char> = ks.iter().map(|k| k.as_ptr()).collect();
        check(unsafe { sys::gsx_shard_render_frame_keys(self.handle.0, ps.as_ptr(), ps.len() as u32, shard_records_max.as_ptr(), speculate as u32, margin, radius) })
    }
    /// Where a sharded frame's bands go: `None` = every rank ends up with the whole frame (default), `Some(r)` = only rank `r`.
    pub fn shard_set_gather_root(&mut self, root: Option<u32>) -> Result<(), Error> { check(unsafe { sys::gsx_shard_set_gather_root(self.handle.0, root.map(|r| r as i32).unwrap_or(-1)) }) }
    /// One process, one thread + one viewer per GPU: seat `rank` of an in-process group (`sys::gsx_comm_group_create`).
    /// What the viewer's communicator really is (transport, ranks RCCL saw — `ncclCommCount` —, this rank, RCCL's version): a multi-process
    /// deployment asserts `nranks == world` before its first frame.
    pub fn comm_info(&mut self) -> Result<sys::gsx_comm_info, Error> {
        let mut info = sys::gsx_comm_info::default();
        check(unsafe { sys::gsx_viewer_comm_info(self.handle.0, &mut info) })?;
        Ok(info)
    }
    pub fn comm_init_group(&mut self, group: *mut sys::gsx_comm_group, rank: u32) -> Result<(), Error> { check(unsafe { sys::gsx_viewer_comm_init_group(self.handle.0, group, rank) }) }
}
pub fn comm_unique_id() -> Result<[u8; 128], Error> {
    let mut id = [0u8; 128];
    check(unsafe { sys::gsx_comm_unique_id(id.as_mut_ptr()) })?;
    Ok(id)
}

/// `gs::Gaussians { gaussians }`: `read_ply_header` / `PlyHeader::count` / `read_ply_gaussians` + `Gaussian::from(PlyGaussianPod)`
/// (app.rs:1053-1096) and `write_ply(writer, edits, mask)` (app.rs:897-947), over the library's host-side PLY code.
pub struct Gaussians { pub gaussians: Vec<Gaussian> }
impl Gaussians {
    pub fn read_ply(data: &[u8]) -> Result<Self, Error> {
        let mut h = std::mem::MaybeUninit::<sys::gsx_ply_header>::uninit();
        check(unsafe { sys::gsx_ply_read_header(data.as_ptr().cast(), data.len() as u64, h.as_mut_ptr()) })?;
        let h = unsafe { h.assume_init() };
        let mut g = Vec::with_capacity(h.count as usize);
        check(unsafe { sys::gsx_ply_read_gaussians(data.as_ptr().cast(), data.len() as u64, &h, 0, h.count, g.as_mut_ptr()) })?;
        unsafe { g.set_len(h.count as usize) };
        Ok(Self { gaussians: g })
    }
    pub fn write_ply(&self, edits: Option<&[GaussianEditPod]>, mask: Option<&[u32]>) -> Result<Vec<u8>, Error> {
        let (e, m) = (edits.map_or(std::ptr::null(), |e| e.as_ptr()), mask.map_or(std::ptr::null(), |m| m.as_ptr()));
        let mut size = 0u64;
        check(unsafe { sys::gsx_ply_write(self.gaussians.as_ptr(), self.gaussians.len() as u64, m, e, std::ptr::null_mut(), 0, &mut size) })?;
        let mut out = vec![0u8; size as usize];
        check(unsafe { sys::gsx_ply_write(self.gaussians.as_ptr(), self.gaussians.len() as u64, m, e, out.as_mut_ptr().cast(), size, &mut size) })?;
        Ok(out)
    }
}
