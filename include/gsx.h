/*
 * gsx.h — C ABI of libgsx.so, the MI355X-native 3D Gaussian Splatting render path.
 *
 * This header is the drop-in boundary for the one hot path of LioQing/wgpu-3dgs-viewer-app:
 * the per-splat render pipeline that the app drives through the Rust crate
 * `wgpu-3dgs-viewer 0.2.0` (Cargo.toml:25-31).  The reference has no C ABI; each entry point
 * below cites the reference call site (file:line under /root/reference/src) whose crate call it
 * replaces.  A Rust `gs::` facade over these symbols is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C, no C++ / torch types; all matrices column-major (glam `Mat4::to_cols_array()`).
 *  - every function returns gsx_status (0 = OK); gsx_last_error_string() gives a thread-local
 *    message.  Nothing aborts or throws across the ABI (reference: fallible calls return
 *    Result<_, gs::Error>, app.rs:548, scene.rs:234).
 *  - one viewer is used by one thread at a time (reference: "should not be used in multiple
 *    threads", scene.rs:1924-1930).
 *  - all device work is enqueued on the viewer's HIP stream; gsx_sync() is
 *    `device.poll(wgpu::Maintain::Wait)` (scene.rs:614, scene.rs:873).
 */
#ifndef GSX_H
#define GSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSX_ABI_VERSION 3u
#define GSX_TILE 16u /* screen tile edge in pixels (build-internal; the reference has no tiles) */
#define GSX_SH_COEFFS 15u /* SH degree 1..3 coefficients, each an RGB triple (gs::Gaussian::sh) */

typedef int32_t gsx_status;
enum {
    GSX_OK = 0,
    GSX_ERR_INVALID_ARG = 1,
    GSX_ERR_OOM = 2,
    GSX_ERR_HIP = 3,
    GSX_ERR_RCCL = 4,
    GSX_ERR_IO = 5, /* gs::Error::Io, scene.rs:234 */
    GSX_ERR_PLY = 6,
    GSX_ERR_NOT_FOUND = 7,
    GSX_ERR_UNSUPPORTED = 8,
    GSX_ERR_NO_DEVICE = 9
};

typedef struct gsx_viewer gsx_viewer; /* gs::MultiModelViewer<G>, scene.rs:1930 */

/* gs::Gaussian — the CPU-side Gaussian the app streams in (app.rs:1029-1031, scene.rs:2069-2085).
 * Field list follows wgpu-3dgs-viewer 0.2.0 (rot, pos, color, sh, scale). 224 bytes. */
typedef struct gsx_gaussian {
    float rot[4];    /* unit quaternion x,y,z,w (glam Quat) */
    float pos[3];
    uint8_t color[4]; /* r,g,b = clamp(0.5 + SH_C0*f_dc), a = sigmoid(opacity), UNORM8 */
    float sh[GSX_SH_COEFFS][3];
    float scale[3];  /* linear scale (exp already applied) */
} gsx_gaussian;

/* 8-way pod dispatch of the reference (scene.rs:23-81, app.rs:352-383). */
typedef enum gsx_sh_kind { GSX_SH_SINGLE = 0, GSX_SH_HALF = 1, GSX_SH_NORM8 = 2, GSX_SH_NONE = 3 } gsx_sh_kind;
typedef enum gsx_cov3d_kind { GSX_COV3D_SINGLE = 0, GSX_COV3D_HALF = 1 } gsx_cov3d_kind;

/* gs::GaussianDisplayMode (app.rs:1141-1165, transform.rs:106-146). */
typedef enum gsx_display_mode { GSX_DISPLAY_SPLAT = 0, GSX_DISPLAY_ELLIPSE = 1, GSX_DISPLAY_POINT = 2 } gsx_display_mode;

/* Constants of the render spec that the reference tree does not pin (SURVEY.md §8c [BUILD-SPEC]).
 * They are named and switchable so they can be reconciled against the real crate. */
typedef struct gsx_spec_params {
    float max_std_dev;    /* support cutoff: d^T Sigma^-1 d <= max_std_dev^2   (default 3.0) */
    float cull_margin;    /* clip-space frustum margin m: |x|,|y| <= m*w        (default 1.3) */
    float jacobian_clamp; /* view-space x/z,y/z clamp as a multiple of tan(fov/2) (default 1.3) */
    float low_pass;       /* added to the cov2d diagonal, px^2                  (default 0.3) */
    float alpha_max;      /* alpha = min(alpha_max, opacity*weight)             (default 1.0) */
    float alpha_min;      /* contributions with alpha < alpha_min are skipped   (default 0.0) */
    float t_epsilon;      /* front-to-back early termination: stop when T < eps (default 1e-4) */
    float point_radius;   /* GSX_DISPLAY_POINT dot radius in px at size 1       (default 2.0) */
} gsx_spec_params;

/* How gsx_render schedules one model's splats (build-internal; no reference counterpart).
 * progressive = 1: the depth-sorted splats are binned and composited front to back in growing depth slabs
 * [0, N_vis/first_slab_divisor), then `growth` times larger each; tiles whose every pixel has reached
 * T < t_epsilon are flagged and receive no further tile entries.  Pixels are identical to progressive = 0
 * (the per-pixel operation sequence is unchanged); only the work on hidden splats is skipped.
 * speculative = 1 (needs progressive): temporal occlusion speculation.  Every tile remembers the depth at which it
 * saturated in the model's previous frame; only the records in front of (1 + spec_margin) x that depth, maximised over
 * the tile's (2 spec_radius + 1)^2 neighbourhood, enter this frame's depth sort and binning.  The compositor verifies
 * the assumption on the device and a second round hands the tiles that are still open the records they were refused,
 * composited behind — pixels stay identical to speculative = 0 whatever the camera does; a wrong guess only costs
 * time.  After such a frame gsx_model_download_sorted returns the second round's (possibly empty) order, and the model
 * must go through gsx_preprocess + gsx_sort again before it is rendered once more (the app does so every frame;
 * gsx_render refuses otherwise: the frame's admission belongs to windows the frame has replaced).
 * host_verify: the (usually empty) second round is 8 kernel launches (round 6; ~20 before) that fall through at ~4 us of stream time each.
 * 0 (default): gsx_render never waits for the device; the second round is always enqueued.
 * 1: the device verification posts its verdict (how many tiles need the second round) into pinned host memory and
 * gsx_render waits for that one word before it returns; the second round is enqueued only when needed.  The next
 * frame's windows are enqueued before the wait.  The host can no longer run ahead of the device: it pays when the host
 * prepares a frame in a few tens of microseconds and repairs are rare (cfg4, 25 % of the frames repair: 1234 -> 1289 fps
 * from a lean Python loop, cfg3 1452 -> 1507) and costs dearly when they are not (cfg2, 94 %: 1527 -> 1002: the host
 * enqueues the second round while the device idles).
 * 2: ask only while repairs are rare — stop when six of the last eight verdicts needed the second round; the verdicts
 * keep being posted and the host looks at the latest one without waiting: eight repair-free ones in a row and it asks
 * again — AND only while the app waits for its frames anyway (it has called gsx_sync or a blocking readback since the
 * frame before, as the reference does twice per frame): a host that streams frames without waiting gets 0's behaviour
 * (round 6: the second round is 8 launches now; asking while streaming cost 6-10 %, profiles/r06_ab_host_verify.txt).
 * frames_in_flight = L > 1: gsx_render_frame deals consecutive frames round-robin to L lanes, each with its own stream and
 * per-frame buffers (records, sort and tile buffers, framebuffer, speculation windows: a lane speculates from ITS last
 * frame, L poses back); the Gaussian data is shared.  The device then overlaps the latency-bound tail of one frame with
 * the bandwidth-bound projection of the next: more frames per second, each taking longer from first to last kernel.
 * Frames still complete in order per lane and every frame is the same frame bit for bit.  Readback calls
 * (gsx_download_framebuffer, gsx_model_frame_stats, gsx_framebuffer_device_ptr ...) refer to the newest frame; a
 * framebuffer pointer stays valid until the same lane renders again (L frames later).  Calls that touch model data
 * (uploads, masks, selection / edits) are ordered after every frame in flight.  Frames with a selection, stored edits or the
 * highlight overlap like any other (the lanes read the viewer's selection and edit records; the per-frame preparation of those
 * runs only after one of its inputs changed, ordered between the frames in flight).  Frames with a query, a band
 * (gsx_viewer_set_band), an external framebuffer or a sharded model run on the viewer itself, one at a time. */
typedef struct gsx_render_options {
    uint32_t progressive;        /* default 1 */
    uint32_t first_slab_divisor; /* default 16 */
    uint32_t min_slab;           /* models with N_vis <= min_slab use one slab; default 131072 */
    uint32_t growth;             /* default 2 */
    uint32_t speculative;        /* default 1 */
    float spec_margin;           /* default 0.25 */
    uint32_t spec_radius;        /* default 3 (tiles) */
    uint32_t host_verify;        /* default 0 */
    uint32_t frames_in_flight;   /* default 1; 1 .. 4 — see below */
    uint32_t slab_shading;       /* default 1: a progressive frame WITHOUT windows (the first frame, a probe of the speculation tuner,
                                  * speculative = 0) projects geometry only and gives conic / colour records, depth slab by depth slab, to
                                  * exactly the records some block of tiles still takes — a few per cent of the visible ones on an opaque
                                  * scene; 0: such a frame projects every Gaussian in full (the reference's K1 + K3 vertex work for every
                                  * visible Gaussian).  Same pixels either way. */
} gsx_render_options;

typedef struct gsx_viewer_desc {
    uint32_t abi_version; /* GSX_ABI_VERSION */
    int32_t device;       /* HIP device ordinal */
    void* stream;         /* hipStream_t to enqueue on, or NULL to create one */
    uint32_t width, height; /* initial size; reference passes uvec2(1,1), scene.rs:1980 */
} gsx_viewer_desc;

/* ---- error / info ---- */
const char* gsx_last_error_string(void);
uint32_t gsx_abi_version(void);
void gsx_spec_params_default(gsx_spec_params* out);

/* ---- construction: gs::MultiModelViewer::new_with(device, format, depth_stencil, size), scene.rs:1969-1980 ---- */
gsx_status gsx_viewer_create(const gsx_viewer_desc* desc, gsx_viewer** out);
void gsx_viewer_destroy(gsx_viewer* v);
gsx_status gsx_viewer_set_spec_params(gsx_viewer* v, const gsx_spec_params* p);
void gsx_render_options_default(gsx_render_options* out);
gsx_status gsx_viewer_set_render_options(gsx_viewer* v, const gsx_render_options* o);

/* ---- models: MultiModelViewerGaussianBuffers::new_empty + BindGroups::new + models.insert,
 *      scene.rs:2111-2139; viewer.remove_model(&key), scene.rs:2176 ---- */
gsx_status gsx_model_create(gsx_viewer* v, const char* key, uint64_t count, gsx_sh_kind sh, gsx_cov3d_kind cov3d);
gsx_status gsx_model_remove(gsx_viewer* v, const char* key);
gsx_status gsx_model_len(gsx_viewer* v, const char* key, uint64_t* out_count); /* gaussians_buffer.len(), scene.rs:608 */

/* ---- upload: gaussians_buffer.update_range(queue, start, &[gs::Gaussian]), scene.rs:2083-2084.
 *      Converts Gaussian -> pod on the GPU (cov3d from rot/scale) into the resident SoA planes. ---- */
gsx_status gsx_model_upload_range(gsx_viewer* v, const char* key, uint64_t start, const gsx_gaussian* src, uint64_t n);
/* Zero-copy variant: pod-ready planes already in DEVICE memory (pos 3n, color n, sh 45n, cov3d 6n). */
gsx_status gsx_model_upload_pod_device(gsx_viewer* v, const char* key, uint64_t start, uint64_t n,
                                       const float* d_pos, const uint32_t* d_color, const float* d_sh,
                                       const float* d_cov3d);

/* ---- per-frame uniform setters (scene.rs:795-809) ---- */
/* viewer.update_camera(queue, &impl CameraTrait, uvec2 size), scene.rs:795 */
gsx_status gsx_update_camera(gsx_viewer* v, const float view[16], const float proj[16], uint32_t width, uint32_t height);
/* viewer.update_model_transform(queue, key, pos, quat, scale), scene.rs:796-802 */
gsx_status gsx_update_model_transform(gsx_viewer* v, const char* key, const float pos[3], const float quat_xyzw[4],
                                      const float scale[3]);
/* viewer.update_gaussian_transform(queue, size, display_mode, sh_deg, no_sh0), scene.rs:803-809 */
gsx_status gsx_update_gaussian_transform(gsx_viewer* v, float size, gsx_display_mode mode, uint32_t sh_deg,
                                         uint32_t no_sh0);

/* ---- mask (gs::MaskEvaluator result buffer, one bit per Gaussian, scene.rs:1851; app.rs:806-807) ---- */
gsx_status gsx_model_upload_mask(gsx_viewer* v, const char* key, const uint32_t* words, uint64_t n_words);
gsx_status gsx_model_download_mask(gsx_viewer* v, const char* key, uint32_t* words, uint64_t n_words);

/* gs::MaskEvaluator::evaluate(device, queue, &MaskOpTree, mask_buffer, model_transform_buffer, gaussians_buffer),
 * scene.rs:2124-2131, 2201-2209.  The tree (app.rs:1815-1837: Union | Intersection | Difference |
 * SymmetricDifference | Complement | Shape(&pod) | Reset) is passed in postfix order; n_ops == 0 is Reset
 * (every bit set).  A Gaussian is inside a shape when its WORLD position (model transform applied), taken
 * into the shape's frame (pos, rotation, scale), lies in the unit box |x|,|y|,|z| <= 1 or the unit ball.
 * Result: the model's mask buffer, bit = 1 kept / rendered.  One HIP kernel, no readback. */
typedef enum gsx_mask_shape_kind { GSX_MASK_BOX = 0, GSX_MASK_ELLIPSOID = 1 } gsx_mask_shape_kind;
typedef struct gsx_mask_shape { /* gs::MaskOpShapePod (gs::MaskShape {kind, pos, rotation, scale, color}) */
    uint32_t kind;
    float pos[3];
    float quat_xyzw[4];
    float scale[3];
} gsx_mask_shape;
typedef enum gsx_mask_opcode {
    GSX_MASK_OP_SHAPE = 0, /* push shape[arg] */
    GSX_MASK_OP_UNION = 1,
    GSX_MASK_OP_INTERSECTION = 2,
    GSX_MASK_OP_DIFFERENCE = 3, /* a - b, a pushed first */
    GSX_MASK_OP_SYMMETRIC_DIFFERENCE = 4,
    GSX_MASK_OP_COMPLEMENT = 5
} gsx_mask_opcode;
typedef struct gsx_mask_op { uint32_t opcode; uint32_t arg; } gsx_mask_op;
#define GSX_MASK_MAX_OPS 64u
#define GSX_MASK_MAX_SHAPES 32u
gsx_status gsx_mask_evaluate(gsx_viewer* v, const char* key, const gsx_mask_op* ops, uint32_t n_ops,
                             const gsx_mask_shape* shapes, uint32_t n_shapes);

/* ---- frame execution, split exactly like the reference's per-frame protocol (scene.rs:856-873, 2302-2314) ---- */
/* preprocessor.preprocess(encoder, bind_group, N): cull + SH colour + 3D->2D covariance + depth key. scene.rs:856-863 */
gsx_status gsx_preprocess(gsx_viewer* v, const char* key);
/* radix_sorter.sort(encoder, bind_group, indirect_args): depth radix sort of the surviving Gaussians. scene.rs:865-869 */
gsx_status gsx_sort(gsx_viewer* v, const char* key);
/* device.poll(Maintain::Wait), scene.rs:614 / scene.rs:873 */
gsx_status gsx_sync(gsx_viewer* v);
/* for key in model_render_keys (far -> near): renderer.render_with_pass(...), scene.rs:2302-2314.
 * Bins every model's sorted splats into 16x16 tiles and composites them; the result is the
 * premultiplied (r,g,b) + transmittance T framebuffer. */
gsx_status gsx_render(gsx_viewer* v, const char* const* keys_far_to_near, uint32_t n_keys);
/* preprocess + sort for every key, then render: one call per frame. */
gsx_status gsx_render_frame(gsx_viewer* v, const char* const* keys_far_to_near, uint32_t n_keys);

/* ---- readback (buffer.download(&device,&queue), app.rs:789, app.rs:806) ---- */
/* float32 [height][width][4] = premultiplied r,g,b and transmittance T. Synchronises. */
gsx_status gsx_download_framebuffer(gsx_viewer* v, float* rgbt, uint64_t n_floats);
/* resolve against a background colour into RGBA8 (what the egui target would hold). Synchronises. */
gsx_status gsx_download_rgba8(gsx_viewer* v, const float background_rgb[3], uint8_t* rgba, uint64_t n_bytes);
/* device pointer of the (rgb,T) framebuffer, for zero-copy consumers (RCCL merge, torch). */
gsx_status gsx_framebuffer_device_ptr(gsx_viewer* v, void** out_ptr, uint32_t* out_w, uint32_t* out_h);

/* ---- parity / introspection (no reference counterpart; used by tests and bench) ---- */
typedef struct gsx_frame_stats {
    uint64_t n_gaussians; /* N of the model */
    uint64_t n_visible;   /* N_vis after cull */
    uint64_t n_tile_entries; /* D = list entries binned by the last gsx_render: (Gaussian, tile) pairs for a frame that keeps
                                per-tile lists (progressive = 0, or a single-slab front model: all of them), (Gaussian, block of
                                tiles) pairs for the depth slabs of a progressive frame (fewer: see DESIGN.md, block lists) */
    uint64_t n_sorted;       /* records that entered the depth sort (= n_visible unless the frame was speculated) */
    uint64_t n_repair_tiles; /* speculated frame: tiles that needed the repair round */
    uint64_t n_repair_sorted; /* speculated frame: records that entered the repair round's depth sort */
    uint32_t speculated;     /* 1 if the last gsx_render of this model used last frame's windows */
    uint32_t overflow_slabs; /* depth slabs, over the model's lifetime, whose tile entries exceeded the pair buffers: their tails
                              * were composited pair-free on the device (complete pixels, slower) and the buffers grown */
} gsx_frame_stats;
gsx_status gsx_model_frame_stats(gsx_viewer* v, const char* key, gsx_frame_stats* out);
/* Per-Gaussian projection outputs of the last gsx_preprocess (host arrays of length N; any may be NULL):
 * depth_key (0xFFFFFFFF = culled), rect[4] = tile x0,y0,x1,y1 (exclusive max), mean2d[2],
 * conic_opacity[4], rgb[3]. */
gsx_status gsx_model_download_projection(gsx_viewer* v, const char* key, uint32_t* depth_key, uint32_t* rect,
                                         float* mean2d, float* conic_opacity, float* rgb);
/* Front-to-back order of the last gsx_sort: first n_visible entries are Gaussian indices. */
gsx_status gsx_model_download_sorted(gsx_viewer* v, const char* key, uint32_t* indices, uint64_t capacity,
                                     uint64_t* out_n_visible);
/* Tile lists of the last gsx_render for `key`: tile_offsets has tiles_x*tiles_y+1 entries, list holds D
 * Gaussian indices in front-to-back order per tile. */
gsx_status gsx_model_download_tile_lists(gsx_viewer* v, const char* key, uint32_t* tile_offsets,
                                         uint64_t n_offsets, uint32_t* list, uint64_t capacity);
/* Pod planes as resident in HBM (pos 3n, color n, sh 45n, cov3d 6n), for upload-conversion parity. */
gsx_status gsx_model_download_pod(gsx_viewer* v, const char* key, float* pos, uint32_t* color, float* sh,
                                  float* cov3d);

/* ---- selection, per-Gaussian edits, queries.  The app builds the pods and calls
 *      the crate (src/tab/scene.rs:740-835, 601-614, 651-657; app.rs:1479-1564); the arithmetic is the build's,
 *      spec/RENDER_SPEC.md §7 [BUILD-SPEC].  None of this costs anything while no selection / edit / query exists. ---- */
#define GSX_EDIT_ENABLED 1u        /* gs::GaussianEditFlag::ENABLED        (app.rs:1546-1552) */
#define GSX_EDIT_HIDDEN 2u         /* gs::GaussianEditFlag::HIDDEN */
#define GSX_EDIT_OVERRIDE_COLOR 4u /* gs::GaussianEditFlag::OVERRIDE_COLOR: color is RGB, otherwise an HSV edit */
/* gs::GaussianEditPod::new(flag, color, contrast, exposure, gamma, alpha) (app.rs:1553-1562); 32 bytes */
typedef struct gsx_gaussian_edit {
    uint32_t flag;
    float color[3]; /* HSV edit: hue shift [0,1], saturation factor, brightness factor; or the override RGB */
    float contrast, exposure, gamma, alpha;
} gsx_gaussian_edit;
void gsx_gaussian_edit_default(gsx_gaussian_edit* e); /* gs::GaussianEditPod::default(): flag 0, (0,1,1), 0, 0, 1, 1 */

typedef enum gsx_query_kind {  /* gs::QueryNonePod / QueryHitPod / the QueryToolset's rect, brush and texture queries */
    GSX_QUERY_NONE = 0, GSX_QUERY_HIT = 1, GSX_QUERY_RECT = 2, GSX_QUERY_BRUSH = 3, GSX_QUERY_TEXTURE = 4
} gsx_query_kind;
typedef enum gsx_selection_op { GSX_SELECTION_SET = 0, GSX_SELECTION_ADD = 1, GSX_SELECTION_REMOVE = 2 } gsx_selection_op;
typedef struct gsx_query {  /* coordinates in viewport pixels, origin top-left */
    uint32_t kind;         /* gsx_query_kind */
    uint32_t selection_op; /* gsx_selection_op (gs::QuerySelectionOp, scene.rs:1605), applied by gsx_postprocess */
    float p0[2];           /* Hit: the point; Rect: one corner; Brush: stroke start */
    float p1[2];           /* Rect: the opposite corner; Brush: stroke end */
    float radius;          /* Brush */
    uint32_t reserved;
} gsx_query;
typedef struct gsx_query_hit { uint32_t index; float depth; float alpha; uint32_t reserved; } gsx_query_hit; /* gs::QueryHitResultPod */
#define GSX_QUERY_MAX_HITS 65536u

/* viewer.update_query(queue, &pod) (scene.rs:785): the query every following gsx_preprocess evaluates. */
gsx_status gsx_update_query(gsx_viewer* v, const gsx_query* q);
/* The query texture the toolset paints its strokes into (viewer.update_query_texture_size + QueryToolset::render,
 * scene.rs:740, 791): width*height bytes from HOST memory, non-zero = selected; must match the viewport. */
gsx_status gsx_update_query_texture(gsx_viewer* v, const uint8_t* texels, uint32_t width, uint32_t height);
/* viewer.update_selection_highlight(queue, rgba) / _with_pod (scene.rs:816-829); alpha 0 = no highlight. */
gsx_status gsx_update_selection_highlight(gsx_viewer* v, const float rgba[4]);
/* viewer.update_selection_edit_with_pod(queue, &pod) (scene.rs:815, 821, 848): while ENABLED, every preprocess writes
 * it into the edit records of the selected Gaussians. */
gsx_status gsx_update_selection_edit(gsx_viewer* v, const gsx_gaussian_edit* e);
/* The "unedited model" bind group (scene.rs:856-863): on != 0 renders the model ignoring (and not touching) edits. */
gsx_status gsx_model_show_unedited(gsx_viewer* v, const char* key, uint32_t on);
/* postprocessor.postprocess(...) (scene.rs:601-611): applies the selection op of a Rect/Brush/Texture query evaluated
 * by the last gsx_preprocess(key) to the model's selection. */
gsx_status gsx_postprocess(gsx_viewer* v, const char* key);
/* Selection bitset, ceil(len/32) words, bit i = Gaussian i selected. */
gsx_status gsx_model_upload_selection(gsx_viewer* v, const char* key, const uint32_t* words, uint64_t n_words);
gsx_status gsx_model_download_selection(gsx_viewer* v, const char* key, uint32_t* words, uint64_t n_words);
/* gaussians_edit_buffer.download() (app.rs:789): len records; Gaussians never edited read as the default pod. */
gsx_status gsx_model_download_edits(gsx_viewer* v, const char* key, gsx_gaussian_edit* out, uint64_t n);
gsx_status gsx_model_upload_edits(gsx_viewer* v, const char* key, const gsx_gaussian_edit* edits, uint64_t n);
/* gs::query::download(count_buffer, results_buffer) (scene.rs:651-657): hits of the last Hit query on `key`, sorted by
 * (depth, index). */
gsx_status gsx_query_download_hits(gsx_viewer* v, const char* key, gsx_query_hit* out, uint64_t capacity, uint64_t* out_n);
/* gs::query::hit_pos_by_closest / hit_pos_by_alpha_range (scene.rs:660-679), host arithmetic: world position on the
 * pixel ray through `coords` at the chosen hit's depth.  view/proj column-major as in gsx_update_camera.
 * GSX_ERR_NOT_FOUND when there is no hit. */
gsx_status gsx_query_hit_pos_by_closest(const gsx_query_hit* hits, uint64_t n, const float view[16], const float proj[16],
                                        uint32_t width, uint32_t height, const float coords[2], uint32_t* out_index,
                                        float out_pos[3]);
gsx_status gsx_query_hit_pos_by_alpha_range(const gsx_query_hit* hits, uint64_t n, const float view[16], const float proj[16],
                                            uint32_t width, uint32_t height, const float coords[2], float range,
                                            uint32_t* out_index, float* out_alpha, float out_pos[3]);

/* ---- ref-counted buffer handles for readback off the owner's thread.  gs:: buffers are cheaply `Clone`: the export path clones
 *      every model's edit and mask buffer, moves the clones into two spawned threads and downloads there while the UI thread
 *      keeps rendering (app.rs:769-816, scene.rs:635-648).  gsx_model_buffer_retain (owner thread; enqueues a device-side
 *      snapshot on the viewer's stream: the contents as of this call, what a clone + download reads in wgpu's submission
 *      order) returns a handle; gsx_buffer_download runs on ANY thread, concurrently with frames, uploads, gsx_model_remove
 *      and even gsx_viewer_destroy — it touches only the handle.  gsx_buffer_retain adds a reference (a clone),
 *      gsx_buffer_release drops one; the snapshot is freed with the last.  Elements: GSX_BUFFER_EDITS -> gsx_gaussian_edit x
 *      len (never-edited Gaussians read as the default pod); GSX_BUFFER_MASK / _SELECTION -> uint32 x ceil(len / 32). ---- */
typedef struct gsx_buffer gsx_buffer;
typedef enum gsx_buffer_kind { GSX_BUFFER_MASK = 0, GSX_BUFFER_EDITS = 1, GSX_BUFFER_SELECTION = 2 } gsx_buffer_kind;
gsx_status gsx_model_buffer_retain(gsx_viewer* v, const char* key, gsx_buffer_kind kind, gsx_buffer** out);
gsx_status gsx_buffer_retain(gsx_buffer* b);
void gsx_buffer_release(gsx_buffer* b);
gsx_status gsx_buffer_len(gsx_buffer* b, uint64_t* out_elements);
gsx_status gsx_buffer_download(gsx_buffer* b, void* out, uint64_t n_elements);

/* ---- multi-GPU stage split.  No reference counterpart: the reference renders on one wgpu device
 *      (src/main.rs:85-98).  One process per GPU holds an index shard of the Gaussians; the screen is cut into
 *      `world` contiguous bands of tile rows, band g = rank g.  Per frame and rank:
 *        [gsx_shard_set_windows(key, windows)]            optional: lets the projection skip what cannot travel
 *        gsx_preprocess(key)                              project the resident shard
 *        gsx_shard_pack(key, world, windows, ..)          the records some tile's depth-key window [lo, hi) admits,
 *                                                         grouped by destination (first frame: no windows = all)
 *        [RCCL all-to-all of the 48-byte records, done by the caller]
 *        gsx_shard_import(.., windows); gsx_sort(key); gsx_render(&key, 1)   this rank's band, progressive slabs;
 *                                                         a tile bins exactly the records its window admits
 *        gsx_shard_feedback(..) -> per-tile saturation depth keys   one small all-gather: verification AND the
 *                                                                   next frame's windows (caller's policy)
 *        only if a tile with a bounded window is still open: gsx_shard_pack(key, world, windows2, ..) with
 *        windows2 = [hi, inf) for those tiles and [0, 0) elsewhere -> all-to-all -> gsx_shard_import(.., windows2)
 *        -> gsx_sort -> gsx_render_more(&key, 1)          (composited behind what the tiles hold)
 *        [RCCL all-gather of the bands straight into the caller's padded framebuffer]
 *      Pixels equal the single-GPU frame bit for bit whatever was predicted. ---- */
#define GSX_RECORD_BYTES 48u /* mean.xy rect.xy | conic.abc opacity | rgb depth */
typedef struct gsx_shard_layout_t {
    uint32_t rows_per_rank, row_lo, row_hi; /* tallest band of the layout; band of this rank: tile rows [row_lo, row_hi) */
    uint64_t band_bytes;                    /* this rank's band: rows * 16 * width * 16 (equal bands: rows_per_rank rows for every rank) */
    uint64_t band_offset_bytes;             /* of this rank's band inside the framebuffer */
    uint64_t padded_framebuffer_bytes;      /* whole tile rows of every band >= width * height * 16 */
} gsx_shard_layout_t;
gsx_status gsx_shard_layout(gsx_viewer* v, uint32_t world, uint32_t rank, gsx_shard_layout_t* out);
/* Screen-band rendering: this viewer composites only the tile rows [row_lo, row_hi) (clamped to the frame); Gaussians
 * whose tile rectangle misses the band are culled by gsx_preprocess, the other rows of the framebuffer are not touched.
 * With the whole scene resident on every GPU (10 M Gaussians = 8 GB of 288 GB) rank g renders band g of
 * gsx_shard_layout and the bands are all-gathered: no record exchange at all.  Default: every row. */
gsx_status gsx_viewer_set_band(gsx_viewer* v, uint32_t row_lo, uint32_t row_hi);
/* Render into caller-owned DEVICE memory (row-major [height][width] float4; may be padded below). NULL restores
 * the internal framebuffer. */
gsx_status gsx_viewer_set_external_framebuffer(gsx_viewer* v, void* d_ptr, uint64_t bytes);
/* Resolve the pixel rows [y0, y1) of the (rgb, T) framebuffer against a background colour into RGBA8 in caller-owned
 * DEVICE memory ((y1 - y0) * width uint32, R in the low byte) — the app's final blit to its Rgba8Unorm surface, per band:
 * what the screen-band mode all-gathers (4 bytes a pixel instead of 16).  Enqueued on the viewer's stream, no host
 * synchronisation.  Rows below the image (the padding of an external framebuffer) may be included. */
gsx_status gsx_resolve_rgba8_device(gsx_viewer* v, const float background_rgb[3], uint32_t y0, uint32_t y1, void* d_rgba);
/* Tile windows: d_tile_window (device, nullable, copied by the call) = one {uint32 lo, uint32 hi} pair per tile,
 * row-major [height/16 rounded up][width/16 rounded up]; a tile admits a record iff lo <= depth key < hi.
 * NULL = every tile admits everything.
 *
 * counts[world] (host) = records per destination; d_send (device) receives the travelling records grouped by
 * destination, ascending local index inside each group.  A visible record travels to rank g if its tile rectangle
 * holds, inside g's band, a tile that admits it.  Synchronises. */
gsx_status gsx_shard_pack(gsx_viewer* v, const char* key, uint32_t world, const uint32_t* d_tile_window, void* d_send,
                          uint64_t capacity_records, uint64_t* counts);
/* Optional, BEFORE gsx_preprocess: the windows [0, hi) the coming exchange will use (NULL clears).  The projection
 * pass then is geometry only, admits candidates against a max-pyramid of the window ends (a conservative superset of the
 * travellers), and only those get their conic / colour records; gsx_shard_pack(.., d_tile_window = NULL, ..) packs
 * from that candidate list with the exact windows given here — nothing else of the shard is looked at.  A later
 * gsx_shard_pack with explicit windows (the repair exchange) scans the whole shard and shades the travellers the
 * first round did not. */
gsx_status gsx_shard_set_windows(gsx_viewer* v, const char* key, const uint32_t* d_tile_window);
/* d_recv (device): n_records records ordered by (source rank, source index).  They become the model's active
 * record set for gsx_sort / gsx_render(_more); binning is restricted to this rank's band and to the tiles that
 * admit the record (the receiving side of gsx_shard_pack's predicate, so that every tile composites a gap-free
 * depth prefix in each round).  The model's own projection stays available for a second gsx_shard_pack. */
gsx_status gsx_shard_import(gsx_viewer* v, const char* key, const void* d_recv, uint64_t n_records, uint32_t world,
                            uint32_t rank, const uint32_t* d_tile_window);
/* Enqueues the write of *out_words = rows_per_rank * (width/16 rounded up) u32 into DEVICE memory: for every tile of
 * this rank's band, the depth key of the splat that saturated its last pixel this frame, 0 = still open (also for
 * padding rows below the frame).  An all-gather over ranks is the map of the whole (padded) frame. */
gsx_status gsx_shard_feedback_words(gsx_viewer* v, uint32_t world, uint32_t* out_words);
gsx_status gsx_shard_feedback(gsx_viewer* v, const char* key, uint32_t world, uint32_t rank, void* d_out_u32);
/* Second round of the same frame: like gsx_render but continues from the framebuffer / saturated-tile state. */
gsx_status gsx_render_more(gsx_viewer* v, const char* const* keys, uint32_t n_keys);

/* ---- multi-GPU, device-resident protocol.  Windows, verification, the repair round's windows, next frame's limits and every
 *      record count stay on the device; the exchange moves fixed-size SLOTS whose headers carry the counts (buffer of a round
 *      = `world` slots of (1 + T) records of GSX_RECORD_BYTES; record 0 of slot g = {u32 records the sender had for g, u32
 *      records sent = min(that, T)}); the host waits for ONE thing per frame, the verdict of round 0 — two words in pinned
 *      memory that the verification kernel derives from globally gathered data, so that every rank reads the same verdict
 *      and takes the same decision.  Per frame and rank (the STAGE calls below, for a caller that strings the protocol together
 *      itself: wgpu_3dgs_viewer_app_amd/parallel.py runs them with an injectable transport so that the tests can put `world` ranks
 *      on one GPU, or on CPU over gloo.  gsx_shard_render_frame runs the same stages without a host look inside the frame: its
 *      repair rounds are enqueued unasked and decide on the device, except the last model's, which — like a frame whose slots
 *      overflowed — is dealt with when the frame is retired; its verdicts go to a pinned ring that is read then):
 *        gsx_shard_frame_begin(key, world, rank, speculate, NULL)      windows [0, limit) from last frame's limits -> projection
 *        gsx_shard_slot_records(key, world, max_shard, &T)             round-0 slot size (2x what the last verdict reported)
 *        gsx_shard_pack_slots(key, world, 0, d_send, T)
 *        all-to-all of the slots, (1 + T) * 48 bytes per peer          gsx_comm_all_to_all
 *        gsx_shard_import_slots(key, d_recv, world, rank, 0, T)        import + depth sort + render this rank's band
 *        gsx_shard_feedback(key, world, rank, d_band) + all-gather     saturation depth key of every tile + slot statistics
 *        gsx_shard_verify(key, world, d_sat_all, &seq)                 repair windows on the device; posts the verdict
 *        gsx_shard_next_windows(..) ; all-gather of the bands          enqueued before the wait: what follows when all is well
 *        gsx_shard_wait_verdict(key, seq, &verdict)
 *          verdict.overflow: round 0 once more with T = max_shard (always fits), verify + wait again
 *          verdict.need_tiles > 0: gsx_shard_repair_count(d_4words) + all-gather + gsx_shard_post_counts(&seq) + wait -> T1
 *                                  (exact); pack_slots(.., 1, T1) -> all-to-all -> import_slots(.., 1, T1) -> feedback +
 *                                  all-gather -> next_windows + the band all-gather again (they replace the early ones)
 *        gsx_shard_frame_end(key)                                      the limits computed last become the next frame's
 *      Whatever the verdict, the frame that leaves the GPU is complete and equals the single-GPU frame bit for bit. ---- */
typedef struct gsx_shard_verdict {
    uint32_t need_tiles;   /* tiles of the frame with a bounded window that are still open (0 after gsx_shard_post_counts) */
    uint32_t overflow;     /* some rank had more records for a destination than the slot held: round 0 must be redone */
    uint32_t max_records;  /* records the busiest (rank, destination) pair wanted: next frame's slot hint / the repair round's T */
    uint32_t reserved;
} gsx_shard_verdict;
gsx_status gsx_shard_frame_begin(gsx_viewer* v, const char* key, uint32_t world, uint32_t rank, uint32_t speculate,
                                 const uint32_t* d_limit_override /* nullable: device u32 per tile, replaces the limits the last frame left */);
/* shard_records_max: the size of the model's LARGEST shard over all ranks (the caller's partition; ceil(N / world) for equal
 * shards).  Every rank gets the same answer: the policy uses only that and the last verdict's global figure. */
gsx_status gsx_shard_slot_records(gsx_viewer* v, const char* key, uint32_t world, uint32_t shard_records_max, uint32_t* out_records);
gsx_status gsx_shard_pack_slots(gsx_viewer* v, const char* key, uint32_t world, uint32_t round, void* d_send, uint32_t slot_records);
/* round: 0 or 1, | GSX_SHARD_BEHIND for a model of a LAYERED frame that is not the nearest one: its records are composited
 * behind what the nearer models of this frame left in the framebuffer (scene.rs:533-558, 2302-2314: models are painted far ->
 * near and never merged), and its feedback ignores the tiles those models had saturated already. */
#define GSX_SHARD_BEHIND 2u
gsx_status gsx_shard_import_slots(gsx_viewer* v, const char* key, const void* d_recv, uint32_t world, uint32_t rank, uint32_t round,
                                  uint32_t slot_records);
/* d_sat_all: the all-gathered gsx_shard_feedback bands: per rank gsx_shard_feedback_words() words = its band of saturation keys
 * (rows_per_rank tile rows) followed by 4 statistics words {records wanted for its busiest destination, slot overflowed, 0, 0} */
gsx_status gsx_shard_verify(gsx_viewer* v, const char* key, uint32_t world, const void* d_sat_all, uint32_t* out_seq);
gsx_status gsx_shard_wait_verdict(gsx_viewer* v, const char* key, uint32_t seq, gsx_shard_verdict* out);
/* repair round sizing: d_out4 (device, 4 u32) = {records this rank has for its busiest destination under the repair windows, 0,0,0};
 * all-gather those (16 bytes per rank), then gsx_shard_post_counts posts the global maximum as a verdict (max_records) */
gsx_status gsx_shard_repair_count(gsx_viewer* v, const char* key, uint32_t world, void* d_out4);
gsx_status gsx_shard_post_counts(gsx_viewer* v, uint32_t world, const void* d_counts_all, uint32_t* out_seq);
gsx_status gsx_shard_next_windows(gsx_viewer* v, const char* key, uint32_t world, const void* d_sat_all, float margin, uint32_t radius);
gsx_status gsx_shard_frame_end(gsx_viewer* v, const char* key);
/* parity / introspection: the per-tile limits the next frame will use (u32 per tile; 0xFFFFFFFF = unbounded). Synchronises. */
gsx_status gsx_shard_download_limits(gsx_viewer* v, const char* key, uint32_t* limits, uint64_t n_words);

/* ---- the collectives, inside the library over RCCL (xGMI): one communicator per viewer (with frames_in_flight = L sharded
 *      frames: L, the others created by the library on first use — a collective step), enqueued on the viewer's stream.
 *      RCCL is loaded at run time; missing / failing RCCL -> GSX_ERR_RCCL.  Bootstrap like any NCCL program: one rank calls
 *      gsx_comm_unique_id, the 128 bytes reach the other ranks by the host's own means, every rank calls gsx_viewer_comm_init
 *      (collective: returns when all `world` ranks have joined). ---- */
gsx_status gsx_comm_unique_id(uint8_t out_id[128]);
gsx_status gsx_viewer_comm_init(gsx_viewer* v, uint32_t world, uint32_t rank, const uint8_t id[128]);
gsx_status gsx_viewer_comm_destroy(gsx_viewer* v);
/* What the viewer's communicator IS, asked of the transport itself (a scaling record should show that RCCL saw N ranks, not that the
 * caller said N): transport 0 none / 1 RCCL / 2 in-process group / 3 the caller's functions; for RCCL nranks = ncclCommCount, rank =
 * ncclCommUserRank, device = ncclCommCuDevice of the first communicator, version = ncclGetVersion (e.g. 22707), lane_comms = the
 * further communicators frames in flight created; otherwise what the viewer was initialised with. */
typedef struct gsx_comm_info {
    uint32_t transport, nranks, rank, lane_comms;
    int32_t device, version;
} gsx_comm_info;
gsx_status gsx_viewer_comm_info(gsx_viewer* v, gsx_comm_info* out);
/* slot p of d_send goes to rank p, slot p of d_recv comes from rank p (grouped point-to-point: all xGMI links at once) */
gsx_status gsx_comm_all_to_all(gsx_viewer* v, const void* d_send, void* d_recv, uint64_t bytes_per_peer);
/* d_recv = world * bytes_per_rank; in place when d_send == d_recv + rank * bytes_per_rank */
gsx_status gsx_comm_all_gather(gsx_viewer* v, const void* d_send, void* d_recv, uint64_t bytes_per_rank);
/* One whole index-sharded frame of model `key` on this rank (needs a communicator: gsx_viewer_comm_init, _init_group or
 * _init_custom): the sequence above, into a padded framebuffer the library owns; after gsx_sync, gsx_download_framebuffer
 * returns the complete frame on every rank. */
gsx_status gsx_shard_render_frame(gsx_viewer* v, const char* key, uint32_t shard_records_max, uint32_t speculate, float margin, uint32_t radius);
/* The same for several LAYERED models — the reference paints `model_render_keys` far -> near and never merges models
 * (src/tab/scene.rs:533-558, 2302-2314), gsx_render does the same on one GPU.  Every model is index-sharded over the ranks
 * (shard_records_max[i] = largest shard of keys_far_to_near[i]); each keeps its own per-tile limits from frame to frame.  Per
 * model: exchange -> composite into this rank's band behind the nearer models -> verification -> repair exchange; one band
 * all-gather at the end.  The frame goes out model by model: the verdict of model i - 1 is read (a pinned word) before model i is
 * enqueued, and its repair is exchanged exactly sized where one is needed.  With frames_in_flight >= 2 a call alternates between what
 * is left of the frame before and the new one, and returns with half the new frame's models enqueued (any call that looks at a result
 * completes them).  GSX_SHARD_LAYER_PIPELINE=0: all models at once, the inner models' repair exchanges always enqueued with fixed-size
 * slots and decided on the device (measured slower).  Pixels equal gsx_render(keys_far_to_near) on one GPU bit for bit. */
gsx_status gsx_shard_render_frame_keys(gsx_viewer* v, const char* const* keys_far_to_near, uint32_t n_keys,
                                       const uint32_t* shard_records_max, uint32_t speculate, float margin, uint32_t radius);

/* ---- other transports under the same frame loop.  gsx_shard_render_frame only ever calls "all-to-all of equal slots" and
 *      "all-gather of equal pieces"; RCCL is one provider of the two.
 *      (a) in-process group: ONE host process drives several GPUs, one host thread and one viewer per GPU — what a
 *          single-process host such as the egui app (src/main.rs:85-98: one process, one event loop) would do.  The collectives are
 *          device-to-device copies ordered by HIP events (peer-to-peer over xGMI between devices), delivered in RCCL's order
 *          (by source rank); ranks meet at a host rendezvous inside every collective, which gives up after timeout_ms
 *          (0 = 60 s): every rank of the group then gets GSX_ERR_RCCL instead of a hang.  Also how the tests put `world`
 *          ranks on ONE GPU and run the real frame loop.
 *      (b) custom: the caller's own two functions (MPI, a test double ...).  They ENQUEUE on `hip_stream` like
 *          ncclSend / ncclRecv do and return 0 or a gsx_status; slot p of d_send goes to rank p, slot p of d_recv comes from
 *          rank p; the all-gather is in place when d_send == d_recv + rank * bytes_per_rank. ---- */
typedef struct gsx_comm_group gsx_comm_group;
gsx_status gsx_comm_group_create(uint32_t world, uint32_t timeout_ms, gsx_comm_group** out);
void gsx_comm_group_destroy(gsx_comm_group* g); /* after every viewer of the group called gsx_viewer_comm_destroy / _destroy */
gsx_status gsx_viewer_comm_init_group(gsx_viewer* v, gsx_comm_group* g, uint32_t rank);
typedef gsx_status (*gsx_comm_all_to_all_fn)(void* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_peer, void* hip_stream);
typedef gsx_status (*gsx_comm_all_gather_fn)(void* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* hip_stream);
gsx_status gsx_viewer_comm_init_custom(gsx_viewer* v, uint32_t world, uint32_t rank, gsx_comm_all_to_all_fn all_to_all,
                                       gsx_comm_all_gather_fn all_gather, void* ctx);
/* A transport that moves pieces of UNEQUAL size (arrays of `world` byte offsets / sizes; a size may be 0): what RCCL's grouped
 * ncclSend / ncclRecv and the in-process group do.  all_to_all_v: send_bytes[p] bytes at d_send + send_offsets[p] go to rank p, what
 * rank p sends lands at d_recv + recv_offsets[p] (recv_bytes[p] bytes: the ranks derive the sizes from gathered data, they agree).
 * gather_v: this rank's send_bytes bytes to every rank (root < 0) or to `root` only; rank p's piece lands at d_recv + recv_offsets[p];
 * d_send may be d_recv + recv_offsets[rank] (in place).  Both ENQUEUE on hip_stream and return 0 or a gsx_status.
 * With such a transport gsx_shard_render_frame balances the bands by the previous frame's per-row work and sizes the exchange slots
 * pair by pair; with the equal-piece functions of gsx_viewer_comm_init_custom the bands stay equal and the slots uniform. */
typedef gsx_status (*gsx_comm_all_to_all_v_fn)(void* ctx, const void* d_send, const uint64_t* send_offsets, const uint64_t* send_bytes,
                                               void* d_recv, const uint64_t* recv_offsets, const uint64_t* recv_bytes, void* hip_stream);
typedef gsx_status (*gsx_comm_gather_v_fn)(void* ctx, const void* d_send, uint64_t send_bytes, void* d_recv, const uint64_t* recv_offsets,
                                           const uint64_t* recv_bytes, int32_t root, void* hip_stream);
gsx_status gsx_viewer_comm_init_custom_v(gsx_viewer* v, uint32_t world, uint32_t rank, gsx_comm_all_to_all_v_fn all_to_all_v,
                                         gsx_comm_gather_v_fn gather_v, void* ctx);
/* Band layout of the sharded frames: rank g owns the tile rows [edges[g], edges[g + 1]) (edges[0] = 0, edges[world] >= tiles_y, non-decreasing;
 * a band may be empty).  gsx_shard_set_band_edges: these edges for every following frame and stage call (NULL: back to the
 * library's own — equal bands, or, in gsx_shard_render_frame over a transport that moves unequal pieces, bands balanced by the previous
 * frame's per-row work; gsx_shard_set_balance(0) keeps those equal too).  Every rank must set the same.  gsx_shard_get_band_edges: the
 * layout the last sharded frame used. */
gsx_status gsx_shard_set_band_edges(gsx_viewer* v, uint32_t world, const uint32_t* edges);
gsx_status gsx_shard_get_band_edges(gsx_viewer* v, uint32_t world, uint32_t* out_edges);
gsx_status gsx_shard_set_balance(gsx_viewer* v, uint32_t enabled);

/* A caller with its own exchange policy (and the tests): per-tile limits the NEXT sharded frame of `key` uses instead of the ones
 * the last frame left (host or device memory, u32 per tile, 0xFFFFFFFF = unbounded; copied by the call), and a fixed round-0 slot size
 * (records; 0 = the library's policy).  Every rank must set the same values. */
gsx_status gsx_shard_set_limits(gsx_viewer* v, const char* key, const uint32_t* limits);
gsx_status gsx_shard_set_slot_records(gsx_viewer* v, const char* key, uint32_t records);
/* What the sharded frames of this viewer did since the last reset (host-side bookkeeping; never synchronises). */
typedef struct gsx_shard_stats {
    uint64_t frames;          /* gsx_shard_render_frame calls completed */
    uint64_t redo_frames;     /* frames whose round 0 was redone with whole-shard slots (a slot overflowed) */
    uint64_t repair_frames;   /* frames that needed the repair exchange */
    uint64_t exchange_rounds; /* all-to-all rounds in total */
    uint64_t wire_bytes;      /* bytes this rank sent to OTHER ranks: slots (fixed size, whatever they hold), feedback and band gathers */
    uint64_t verdict_wait_ns; /* host time spent waiting for verdicts */
    uint32_t last_slot_records, last_repair_slot_records;
    uint32_t last_entries_sum, last_entries_max; /* list entries all ranks / the busiest rank binned in the last frame whose verdict was read
                                                    (balance of the bands: max * world / sum) */
    uint32_t last_work_permille;                 /* work of the busiest rank's tiles x world x 1000 / all ranks': what the bands are balanced
                                                    by (1000 = perfectly even) */
    uint32_t redo_fallbacks;                     /* redone frames whose exactly sized slots overflowed again and that were redone with
                                                    whole-shard slots */
    uint32_t last_repair_records;                /* most records any rank had for one destination in the repair round of the last frame
                                                    whose verdict was read */
    uint32_t reserved0;
} gsx_shard_stats;
gsx_status gsx_shard_get_stats(gsx_viewer* v, gsx_shard_stats* out, uint32_t reset);
/* Where a sharded frame's finished bands go.  root = -1 (default): an all-gather — after gsx_shard_render_frame every rank's
 * framebuffer holds the whole frame ((world - 1) / world of W x H x 16 bytes INTO every rank: 29 MB per rank and frame at
 * 1920x1080 on 8 ranks).  root >= 0: only that rank's does — north_star's "into one framebuffer"; every other rank sends its
 * band there (4 MB each in the same case, arriving on 7 links side by side) and the rest of its own framebuffer is undefined.
 * Every rank must set the same root (ranks that disagree fail the frame).  RCCL and the in-process group gather to the
 * root; a custom transport (gsx_viewer_comm_init_custom) has only its two functions and keeps delivering every band to every rank. */
gsx_status gsx_shard_set_gather_root(gsx_viewer* v, int32_t root);

/* ---- PLY I/O (host side; no GPU needed).  gs::Gaussians::read_ply_header / PlyHeader::count /
 *      read_ply_gaussians + gs::Gaussian::from(PlyGaussianPod) (app.rs:1053-1096) and write_ply (app.rs:897-947).
 *      INRIA 3DGS vertex: x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3 (62 f32, 248 B;
 *      properties are located by NAME, missing f_rest_* read as 0).  binary_little_endian and ascii. ---- */
typedef struct gsx_ply_header {
    uint64_t count;        /* element vertex N  (PlyHeader::count) */
    uint64_t header_bytes; /* offset of the first vertex */
    uint32_t vertex_bytes; /* stride of one binary vertex (0 for ascii) */
    uint32_t is_ascii;
    int32_t offsets[62];   /* byte offset (binary) / column (ascii) of each INRIA property, -1 if absent */
} gsx_ply_header;
/* data/size: the file contents (at least the header). */
gsx_status gsx_ply_read_header(const void* data, uint64_t size, gsx_ply_header* out);
/* Converts vertices [start, start+n) into gs::Gaussian records (rot normalised from w,x,y,z to x,y,z,w;
 * scale = exp; colour = clamp(0.5 + C0*f_dc), alpha = sigmoid(opacity) as UNORM8; f_rest [3][15] -> [15][3]). */
gsx_status gsx_ply_read_gaussians(const void* data, uint64_t size, const gsx_ply_header* header, uint64_t start,
                                  uint64_t n, gsx_gaussian* out);
/* Inverse conversion into a binary_little_endian INRIA PLY.  mask_words (nullable): only Gaussians whose bit
 * is set are written (write_ply's mask iterator).  edits (nullable, n records; write_ply's Option<&[GaussianEditPod]>,
 * app.rs:908-940): ENABLED+HIDDEN Gaussians are dropped, the other ENABLED edits are baked into f_dc / opacity
 * (spec §7 Export).  Call with out == NULL to get the size in *out_size. */
gsx_status gsx_ply_write(const gsx_gaussian* gaussians, uint64_t n, const uint32_t* mask_words,
                         const gsx_gaussian_edit* edits, void* out, uint64_t capacity, uint64_t* out_size);

/* ---- debug: which stable-rank path the radix sort uses.  -1 (default): per device, decided by a start-up probe; 0: wave
 *      ballot matching (documented hardware behaviour only); 1: one returning LDS add per element (relies on same-address
 *      LDS adds of one instruction being served in ascending lane order — what the probe checks).  Process-wide.  Both give
 *      the same order; tests/test_gpu_sort_stress.py and test_gpu_parity.py assert it at full size. ---- */
void gsx_debug_set_radix_rank_mode(int32_t mode);

/* ---- how a frame's kernel launches reach the device.  gsx_preprocess / gsx_sort / gsx_render / gsx_render_frame record their
 *      launches and submit them as cached HIP graphs whose nodes are patched to the frame's arguments (a dependent kernel boundary
 *      costs 1.75 us inside a graph, 3.3 us on a stream; csrc/gsx_launch.h).  The reference records a wgpu CommandEncoder per frame
 *      and submits it once (scene.rs:856-873): same shape.  What a graph saves is HOST time (cfg4: 112 -> 37 us per frame); the
 *      device runs the same kernels at the same pace, and a graph only leaves when its last launch is recorded — so an entry point
 *      that finds its stream idle (a host that waits for every frame) submits launch by launch: the first kernel starts at once.
 *      And the device gains nothing (a real kernel boundary costs the same in a graph as on a stream; a frame that is one graph launch
 *      starts a few microseconds later): 1607 vs 1633 frames/s on cfg4 with one frame in flight.  So the default is
 *      enabled = 0: every launch is submitted at once; 1 (or GSX_GRAPH=1 in the environment): as described — for hosts whose time is what
 *      counts; 2: record even when the stream is idle (tests).  Process-wide.  Either way the same kernels run with the same arguments in the same order:
 *      frames are bit-identical (tests/test_gpu_graph.py). ---- */
void gsx_debug_set_launch_graphs(int32_t enabled);
uint64_t gsx_debug_launch_count(void); /* kernel launches this process has asked for so far (recorded or submitted) */
uint64_t gsx_debug_device_bytes(void); /* device memory the library's buffers hold in this process right now (every viewer, every model) */
/* tests: the framebuffer of lane `lane` (0: the viewer itself, i: its i-th lane) as it is once that lane's stream has drained — WITHOUT
 * completing the frames in flight (every other readback does).  With frames_in_flight = L the frame of call k sits in lane k mod L until
 * call k + L; after call k + 1 it is retired, i.e. complete: this is how a test looks at a frame that was continued across calls. */
gsx_status gsx_debug_download_lane_framebuffer(gsx_viewer* v, uint32_t lane, float* rgbt, uint64_t n_floats);
/* development (viewers created under GSX_TILE_PROFILE=1; tools/tile_profile.py): what every tile of the last frame's first block-compositor
 * launch cost — per tile 4 words: start and duration in 10 ns ticks, chunks of 128 list entries walked | chunks in its list << 16, takers blended */
gsx_status gsx_debug_tile_profile(gsx_viewer* v, uint32_t* out4, uint64_t n_tiles);
typedef struct gsx_launch_stats {
    uint64_t graph_launches;  /* hipGraphLaunch calls */
    uint64_t graph_nodes;     /* kernel launches that left inside a graph */
    uint64_t nodes_patched;   /* graph nodes whose grid / arguments had changed since the graph last ran */
    uint64_t direct_launches; /* kernel launches of recorded segments too short for a graph, submitted one by one */
    uint64_t graphs_built;    /* graphs instantiated (a kernel sequence seen for the first time at its position) */
    uint64_t broken;          /* != 0: a graph call failed on this viewer, it launches directly since (frames stay right) */
    uint64_t idle_direct_scopes; /* entry points that found their stream idle and submitted launch by launch (the device was waiting) */
} gsx_launch_stats;
gsx_status gsx_viewer_launch_stats(gsx_viewer* v, gsx_launch_stats* out, uint32_t reset);

/* ---- timing: HIP events recorded on the viewer's stream around each pass of the last frame ---- */
typedef enum gsx_pass {
    GSX_PASS_PROJECT = 0,
    GSX_PASS_DEPTH_SORT = 1,
    GSX_PASS_BIN = 2,
    GSX_PASS_TILE_SORT = 3,
    GSX_PASS_COMPOSITE = 4,
    GSX_PASS_PROJECT_GEOM = 5, /* the geometry-only projection of a lazily shaded (speculated / sharded) frame: another kernel,
                                  other bytes — timed apart from GSX_PASS_PROJECT so that each average is one kernel's */
    GSX_PASS_SHADE = 6,        /* conic / colour records (SH colour, cov2d) for the Gaussians a lazily shaded frame admitted or a depth slab's
                                  blocks take: the deferred half of K1's arithmetic — k_shade_quads + the frame's colour ops, wherever in
                                  the frame they run (inside the depth sort of a speculated frame, after each slab's binning otherwise) */
    GSX_PASS_COUNT = 7
} gsx_pass;
/* enabled: 0 = off, 1 = every pass, otherwise a mask with bit (p + 1) set for each pass p to bracket with events
 * (an event pair costs a few microseconds of stream gap: time only what is being measured). */
gsx_status gsx_set_pass_timing(gsx_viewer* v, uint32_t enabled);
/* milliseconds of each pass accumulated over all models since the last call (resets accumulators);
 * synchronises. launches[i] = kernel launches of the dominant kernel of pass i. */
gsx_status gsx_get_pass_timing(gsx_viewer* v, float ms[GSX_PASS_COUNT], uint32_t launches[GSX_PASS_COUNT]);

#ifdef __cplusplus
}
#endif
#endif /* GSX_H */
