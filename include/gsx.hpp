// gsx.hpp — C++ host-side mirror of the `wgpu-3dgs-viewer` (gs::) surface the app uses for the render
// path, header-only over the C ABI of gsx.h.
//
// The reference's host language is Rust; there is no Rust toolchain in the build image, so this is the
// compiled-language facade (INTEGRATION.md shows the equivalent Rust binding).  Names, argument order and
// error behaviour follow the reference call sites (file:line under /root/reference/src):
//   gs::MultiModelViewer::new_with                       tab/scene.rs:1969-1980
//   viewer.models[key].gaussian_buffers.gaussians_buffer.update_range / len     tab/scene.rs:2083-2084, 608
//   viewer.update_camera / update_model_transform / update_gaussian_transform   tab/scene.rs:795-809
//   viewer.preprocessor.preprocess / radix_sorter.sort / renderer.render        tab/scene.rs:856-869, 2302-2314
//   device.poll(Maintain::Wait)                                                 tab/scene.rs:873
//   gs::CameraTrait {view, projection}, Mat4::look_at_rh / perspective_rh       app.rs:1236-1244
//   Quat::from_euler(EulerRot::ZYX, ..)                                         app.rs:1123-1130
// Fallible calls throw gs::Error (the reference returns Result<_, gs::Error>, app.rs:548).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "gsx.h"

namespace gs {

struct Error : std::runtime_error {
    gsx_status status;
    Error(gsx_status s, const char* msg) : std::runtime_error(msg), status(s) {}
};
inline void check(gsx_status s) {
    if (s != GSX_OK) throw Error(s, gsx_last_error_string());
}

using Gaussian = gsx_gaussian;  // {rot, pos, color, sh, scale}
using Vec3 = std::array<float, 3>;
using Quat = std::array<float, 4>;   // x, y, z, w
using Mat4 = std::array<float, 16>;  // column-major (glam to_cols_array)
struct UVec2 { uint32_t x, y; };

enum class GaussianDisplayMode : int { Splat = 0, Ellipse = 1, Point = 2 };  // app.rs:1147
enum class ShCompression : int { Single = 0, Half = 1, Norm8 = 2, Remove = 3 };  // app.rs:386-403
enum class Cov3dCompression : int { Single = 0, Half = 1 };                      // app.rs:405-418

class GaussianShDegree {  // transform.rs:139: `new` is None above 3
    uint32_t deg_;
    explicit GaussianShDegree(uint32_t d) : deg_(d) {}
public:
    static std::optional<GaussianShDegree> new_(uint32_t d) { return d <= 3 ? std::optional<GaussianShDegree>(GaussianShDegree(d)) : std::nullopt; }
    static GaussianShDegree new_unchecked(uint32_t d) { return GaussianShDegree(d); }
    uint32_t degree() const { return deg_; }
};

// ---- glam 0.29.2 restated in float32 ----
inline Vec3 sub(Vec3 a, Vec3 b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline float dot(Vec3 a, Vec3 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }
inline Vec3 normalize(Vec3 a) { float l = std::sqrt(dot(a, a)); return {a[0] / l, a[1] / l, a[2] / l}; }
inline Mat4 look_to_rh(Vec3 eye, Vec3 dir, Vec3 up) {
    Vec3 f = normalize(dir), s = normalize(cross(f, up)), u = cross(s, f);
    return {s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0, -dot(eye, s), -dot(eye, u), dot(eye, f), 1};
}
inline Mat4 look_at_rh(Vec3 eye, Vec3 center, Vec3 up) { return look_to_rh(eye, sub(center, eye), up); }
inline Mat4 perspective_rh(float fov_y, float aspect, float z_near, float z_far) {
    float s = std::sin(0.5f * fov_y), c = std::cos(0.5f * fov_y), h = c / s, w = h / aspect, r = z_far / (z_near - z_far);
    return {w, 0, 0, 0, 0, h, 0, 0, 0, 0, r, -1, 0, 0, r * z_near, 0};
}
inline Quat quat_mul(Quat a, Quat b) {
    return {a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
            a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]};
}
inline Quat quat_from_euler_zyx(float z, float y, float x) {  // Quat::from_euler(EulerRot::ZYX, z, y, x)
    Quat qz{0, 0, std::sin(0.5f * z), std::cos(0.5f * z)}, qy{0, std::sin(0.5f * y), 0, std::cos(0.5f * y)},
        qx{std::sin(0.5f * x), 0, 0, std::cos(0.5f * x)};
    return quat_mul(quat_mul(qz, qy), qx);
}

struct CameraTrait {  // gs::CameraTrait
    virtual ~CameraTrait() = default;
    virtual Mat4 view() const = 0;
    virtual Mat4 projection(float aspect_ratio) const = 0;
};
struct CameraOrbitControl : CameraTrait {  // app.rs:1208-1244
    Vec3 target{0, 0, 0}, pos{0, 0, -1};
    float z_near = 0.1f, z_far = 1e4f, vertical_fov = 1.0471975512f;
    Mat4 view() const override { return look_at_rh(pos, target, {0, 1, 0}); }
    Mat4 projection(float aspect) const override { return perspective_rh(vertical_fov, aspect, z_near, z_far); }
};

class MultiModelViewer;

class GaussiansBuffer {  // gs::GaussiansBuffer<G>
    gsx_viewer* v_;
    std::string key_;
public:
    GaussiansBuffer(gsx_viewer* v, std::string key) : v_(v), key_(std::move(key)) {}
    size_t len() const {
        uint64_t n = 0;
        check(gsx_model_len(v_, key_.c_str(), &n));
        return (size_t)n;
    }
    void update_range(size_t start, const Gaussian* gaussians, size_t n) {  // scene.rs:2083-2084
        check(gsx_model_upload_range(v_, key_.c_str(), start, gaussians, n));
    }
    void update_range(size_t start, const std::vector<Gaussian>& g) { update_range(start, g.data(), g.size()); }
};
// A cloned buffer (`edit_buffer.clone()` / `mask_buffer.clone()`, app.rs:769-780): copyable, shares one device-side snapshot
// taken at clone time; download() may run on any thread while the viewer renders (gsx_buffer_*).
template <class T>
class BufferClone {
    gsx_buffer* h_ = nullptr;
public:
    BufferClone(gsx_viewer* v, const std::string& key, gsx_buffer_kind kind) { check(gsx_model_buffer_retain(v, key.c_str(), kind, &h_)); }
    BufferClone(const BufferClone& o) : h_(o.h_) { if (h_) check(gsx_buffer_retain(h_)); }
    BufferClone(BufferClone&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    BufferClone& operator=(BufferClone o) noexcept { std::swap(h_, o.h_); return *this; }
    ~BufferClone() { gsx_buffer_release(h_); }
    std::vector<T> download() const {
        uint64_t n = 0;
        check(gsx_buffer_len(h_, &n));
        std::vector<T> out(n);
        check(gsx_buffer_download(h_, out.data(), n));
        return out;
    }
};
class MaskBuffer {  // gs::MaskBuffer (scene.rs:1851, app.rs:806-807): bit = kept
    gsx_viewer* v_;
    std::string key_;
public:
    MaskBuffer(gsx_viewer* v, std::string key) : v_(v), key_(std::move(key)) {}
    void upload(const std::vector<uint32_t>& words) { check(gsx_model_upload_mask(v_, key_.c_str(), words.data(), words.size())); }
    std::vector<uint32_t> download(size_t n_gaussians) {
        std::vector<uint32_t> w((n_gaussians + 31) / 32);
        check(gsx_model_download_mask(v_, key_.c_str(), w.data(), w.size()));
        return w;
    }
    BufferClone<uint32_t> clone() const { return BufferClone<uint32_t>(v_, key_, GSX_BUFFER_MASK); }  // app.rs:775
};
class SelectionBuffer {  // gs::SelectionBuffer (scene.rs:1846-1850): bit = selected
    gsx_viewer* v_;
    std::string key_;
public:
    SelectionBuffer(gsx_viewer* v, std::string key) : v_(v), key_(std::move(key)) {}
    void upload(const std::vector<uint32_t>& words) { check(gsx_model_upload_selection(v_, key_.c_str(), words.data(), words.size())); }
    void clear() { check(gsx_model_upload_selection(v_, key_.c_str(), nullptr, 0)); }
    std::vector<uint32_t> download(size_t n_gaussians) {
        std::vector<uint32_t> w((n_gaussians + 31) / 32);
        check(gsx_model_download_selection(v_, key_.c_str(), w.data(), w.size()));
        return w;
    }
    BufferClone<uint32_t> clone() const { return BufferClone<uint32_t>(v_, key_, GSX_BUFFER_SELECTION); }
};
using GaussianEditPod = gsx_gaussian_edit;  // gs::GaussianEditPod (app.rs:1553-1562)
class GaussiansEditBuffer {  // gs::GaussiansEditBuffer (scene.rs:1816-1830, app.rs:789)
    gsx_viewer* v_;
    std::string key_;
public:
    GaussiansEditBuffer(gsx_viewer* v, std::string key) : v_(v), key_(std::move(key)) {}
    std::vector<GaussianEditPod> download(size_t n_gaussians) {
        std::vector<GaussianEditPod> e(n_gaussians);
        check(gsx_model_download_edits(v_, key_.c_str(), e.data(), e.size()));
        return e;
    }
    void upload(const std::vector<GaussianEditPod>& e) { check(gsx_model_upload_edits(v_, key_.c_str(), e.data(), e.size())); }
    BufferClone<GaussianEditPod> clone() const { return BufferClone<GaussianEditPod>(v_, key_, GSX_BUFFER_EDITS); }  // app.rs:772
};
struct MultiModelViewerGaussianBuffers {
    GaussiansBuffer gaussians_buffer;
    MaskBuffer mask_buffer;
    SelectionBuffer selection_buffer;
    GaussiansEditBuffer gaussians_edit_buffer;
};
struct MultiModelViewerModel { MultiModelViewerGaussianBuffers gaussian_buffers; };

class MultiModelViewer {
    gsx_viewer* v_ = nullptr;
    struct Pre { gsx_viewer*& v; void preprocess(const std::string& key) { check(gsx_preprocess(v, key.c_str())); } };
    struct Sort { gsx_viewer*& v; void sort(const std::string& key) { check(gsx_sort(v, key.c_str())); } };
    struct Post { gsx_viewer*& v; void postprocess(const std::string& key) { check(gsx_postprocess(v, key.c_str())); } };  // scene.rs:601-611
    struct Ren {
        gsx_viewer*& v;
        void render(const std::vector<std::string>& model_render_keys) {  // far -> near, scene.rs:533-558
            std::vector<const char*> k;
            for (auto& s : model_render_keys) k.push_back(s.c_str());
            check(gsx_render(v, k.data(), (uint32_t)k.size()));
        }
    };
public:
    std::map<std::string, MultiModelViewerModel> models;
    Pre preprocessor{v_};
    Sort radix_sorter{v_};
    Ren renderer{v_};
    Post postprocessor{v_};
    ShCompression sh;
    Cov3dCompression cov3d;
    UVec2 size{1, 1};

    // MultiModelViewer::new_with(&device, format, depth_stencil, size): the target format / depth state of the
    // reference have no meaning for a float (rgb,T) framebuffer and are dropped.
    static MultiModelViewer new_with(int device, UVec2 size, ShCompression sh = ShCompression::Single,
                                     Cov3dCompression cov3d = Cov3dCompression::Single) {
        return MultiModelViewer(device, size, sh, cov3d);
    }
    MultiModelViewer(int device, UVec2 sz, ShCompression sh_, Cov3dCompression cov_) : sh(sh_), cov3d(cov_), size(sz) {
        gsx_viewer_desc d{GSX_ABI_VERSION, device, nullptr, sz.x, sz.y};
        check(gsx_viewer_create(&d, &v_));
    }
    MultiModelViewer(const MultiModelViewer&) = delete;
    MultiModelViewer(MultiModelViewer&& o) noexcept : v_(o.v_), models(std::move(o.models)), sh(o.sh), cov3d(o.cov3d), size(o.size) { o.v_ = nullptr; }
    ~MultiModelViewer() { if (v_) gsx_viewer_destroy(v_); }
    gsx_viewer* raw() { return v_; }

    MultiModelViewerModel& add_model(const std::string& key, size_t count) {  // new_empty + BindGroups::new + insert
        check(gsx_model_create(v_, key.c_str(), count, (gsx_sh_kind)sh, (gsx_cov3d_kind)cov3d));
        return models.emplace(key, MultiModelViewerModel{{GaussiansBuffer(v_, key), MaskBuffer(v_, key), SelectionBuffer(v_, key),
                                                            GaussiansEditBuffer(v_, key)}}).first->second;
    }
    void remove_model(const std::string& key) {  // scene.rs:2176
        check(gsx_model_remove(v_, key.c_str()));
        models.erase(key);
    }
    void update_camera(const CameraTrait& camera, UVec2 sz) {  // scene.rs:795
        Mat4 v = camera.view(), p = camera.projection((float)sz.x / (float)sz.y);
        check(gsx_update_camera(v_, v.data(), p.data(), sz.x, sz.y));
        size = sz;
    }
    void update_model_transform(const std::string& key, Vec3 pos, Quat quat, Vec3 scale) {  // scene.rs:796-802
        check(gsx_update_model_transform(v_, key.c_str(), pos.data(), quat.data(), scale.data()));
    }
    void update_gaussian_transform(float sz, GaussianDisplayMode mode, GaussianShDegree sh_deg, bool no_sh0) {  // scene.rs:803-809
        check(gsx_update_gaussian_transform(v_, sz, (gsx_display_mode)mode, sh_deg.degree(), no_sh0 ? 1u : 0u));
    }
    void update_query(const gsx_query& pod) { check(gsx_update_query(v_, &pod)); }  // scene.rs:785
    void update_query_texture(const std::vector<uint8_t>& texels, UVec2 sz) {  // scene.rs:740, 791
        check(gsx_update_query_texture(v_, texels.data(), sz.x, sz.y));
    }
    void update_selection_highlight(std::array<float, 4> rgba) { check(gsx_update_selection_highlight(v_, rgba.data())); }  // scene.rs:816-829
    void update_selection_edit_with_pod(const GaussianEditPod& pod) { check(gsx_update_selection_edit(v_, &pod)); }   // scene.rs:815
    void show_unedited(const std::string& key, bool on) { check(gsx_model_show_unedited(v_, key.c_str(), on ? 1u : 0u)); }  // scene.rs:856-863
    std::vector<gsx_query_hit> download_query_hits(const std::string& key) {  // gs::query::download, scene.rs:651-657
        std::vector<gsx_query_hit> h(GSX_QUERY_MAX_HITS);
        uint64_t n = 0;
        check(gsx_query_download_hits(v_, key.c_str(), h.data(), h.size(), &n));
        h.resize((size_t)n);
        return h;
    }
    void poll() { check(gsx_sync(v_)); }  // device.poll(Maintain::Wait)
    std::vector<float> download_framebuffer() {
        std::vector<float> fb((size_t)size.x * size.y * 4);
        check(gsx_download_framebuffer(v_, fb.data(), fb.size()));
        return fb;
    }
    gsx_frame_stats frame_stats(const std::string& key) {
        gsx_frame_stats st{};
        check(gsx_model_frame_stats(v_, key.c_str(), &st));
        return st;
    }
};

}  // namespace gs
