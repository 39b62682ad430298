// gsx_ply.cpp — INRIA 3DGS PLY reader / writer and the PLY-vertex <-> gs::Gaussian conversion (host only).
// Reference: gs::Gaussians::read_ply_header, PlyHeader::count, read_ply_gaussians, gs::Gaussian::from
// (src/app.rs:1053-1096) and Gaussians::write_ply (src/app.rs:897-947); the crate reads PLY through
// ply-rs 0.1.3 (Cargo.lock:2699-2702).  The on-disk layout is the one the app assumes when it multiplies
// the count by size_of::<PlyGaussianPod>() (src/tab/scene.rs:996-998): 62 little-endian f32 per vertex.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gsx.h"
#include "edit_math.h"

namespace gsx {
gsx_status ply_fail(gsx_status st, const char* fmt, ...);
}

namespace {

const char* kProps[62] = {
    "x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2",
    "f_rest_0", "f_rest_1", "f_rest_2", "f_rest_3", "f_rest_4", "f_rest_5", "f_rest_6", "f_rest_7", "f_rest_8",
    "f_rest_9", "f_rest_10", "f_rest_11", "f_rest_12", "f_rest_13", "f_rest_14", "f_rest_15", "f_rest_16", "f_rest_17",
    "f_rest_18", "f_rest_19", "f_rest_20", "f_rest_21", "f_rest_22", "f_rest_23", "f_rest_24", "f_rest_25", "f_rest_26",
    "f_rest_27", "f_rest_28", "f_rest_29", "f_rest_30", "f_rest_31", "f_rest_32", "f_rest_33", "f_rest_34", "f_rest_35",
    "f_rest_36", "f_rest_37", "f_rest_38", "f_rest_39", "f_rest_40", "f_rest_41", "f_rest_42", "f_rest_43", "f_rest_44",
    "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"};
enum { P_X = 0, P_FDC = 6, P_REST = 9, P_OPACITY = 54, P_SCALE = 55, P_ROT = 58 };
const float kShC0 = 0.28209479177387814f;

int type_size(const std::string& t) {
    if (t == "float" || t == "float32" || t == "int" || t == "int32" || t == "uint" || t == "uint32") return 4;
    if (t == "double" || t == "float64") return 8;
    if (t == "short" || t == "int16" || t == "ushort" || t == "uint16") return 2;
    if (t == "char" || t == "int8" || t == "uchar" || t == "uint8") return 1;
    return -1;
}

uint8_t unorm8(float x) {
    float c = std::min(std::max(x, 0.0f), 1.0f);
    return (uint8_t)std::floor(c * 255.0f + 0.5f);
}

void vertex_to_gaussian(const float v[62], gsx_gaussian* g) {
    // gs::Gaussian::from(PlyGaussianPod)
    float w = v[P_ROT], x = v[P_ROT + 1], y = v[P_ROT + 2], z = v[P_ROT + 3];
    float len = std::sqrt(((w * w + x * x) + y * y) + z * z);
    if (!(len > 0.0f)) { w = 1.0f; x = y = z = 0.0f; len = 1.0f; }
    g->rot[0] = x / len; g->rot[1] = y / len; g->rot[2] = z / len; g->rot[3] = w / len;
    for (int k = 0; k < 3; ++k) {
        g->pos[k] = v[P_X + k];
        g->scale[k] = std::exp(v[P_SCALE + k]);
        g->color[k] = unorm8(0.5f + kShC0 * v[P_FDC + k]);
    }
    g->color[3] = unorm8(1.0f / (1.0f + std::exp(-v[P_OPACITY])));
    for (int c = 0; c < 15; ++c)
        for (int ch = 0; ch < 3; ++ch) g->sh[c][ch] = v[P_REST + ch * 15 + c];  // channel-major on disk
}

void gaussian_to_vertex(const gsx_gaussian* g, float v[62]) {
    memset(v, 0, sizeof(float) * 62);
    for (int k = 0; k < 3; ++k) {
        v[P_X + k] = g->pos[k];
        v[P_SCALE + k] = std::log(g->scale[k]);
        v[P_FDC + k] = ((float)g->color[k] / 255.0f - 0.5f) / kShC0;
    }
    float a = std::min(std::max((float)g->color[3] / 255.0f, 1e-6f), 1.0f - 1e-6f);
    v[P_OPACITY] = std::log(a / (1.0f - a));
    v[P_ROT] = g->rot[3]; v[P_ROT + 1] = g->rot[0]; v[P_ROT + 2] = g->rot[1]; v[P_ROT + 3] = g->rot[2];
    for (int c = 0; c < 15; ++c)
        for (int ch = 0; ch < 3; ++ch) v[P_REST + ch * 15 + c] = g->sh[c][ch];
}

}  // namespace

extern "C" {

gsx_status gsx_ply_read_header(const void* data, uint64_t size, gsx_ply_header* out) {
    using gsx::ply_fail;
    if (!data || !out) return ply_fail(GSX_ERR_INVALID_ARG, "gsx_ply_read_header: null argument");
    const char* p = static_cast<const char*>(data);
    memset(out, 0, sizeof *out);
    for (int i = 0; i < 62; ++i) out->offsets[i] = -1;
    uint64_t pos = 0;
    auto next_line = [&](std::string& line) -> bool {
        if (pos >= size) return false;
        uint64_t e = pos;
        while (e < size && p[e] != '\n') ++e;
        if (e >= size) return false;
        line.assign(p + pos, p + e);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        pos = e + 1;
        return true;
    };
    std::string line;
    if (!next_line(line) || line != "ply") return ply_fail(GSX_ERR_PLY, "not a PLY file (missing 'ply' magic)");
    bool have_format = false, in_vertex = false, have_vertex = false;
    uint32_t stride = 0, column = 0;
    while (true) {
        if (!next_line(line)) return ply_fail(GSX_ERR_PLY, "PLY header is not terminated by end_header");
        if (line == "end_header") break;
        char a[64] = {0}, b[64] = {0}, c[64] = {0};
        int k = sscanf(line.c_str(), "%63s %63s %63s", a, b, c);
        if (k <= 0 || !strcmp(a, "comment") || !strcmp(a, "obj_info")) continue;
        if (!strcmp(a, "format")) {
            if (!strcmp(b, "binary_little_endian")) out->is_ascii = 0;
            else if (!strcmp(b, "ascii")) out->is_ascii = 1;
            else return ply_fail(GSX_ERR_PLY, "unsupported PLY format '%s'", b);
            have_format = true;
        } else if (!strcmp(a, "element")) {
            in_vertex = !strcmp(b, "vertex");
            if (in_vertex) {
                if (have_vertex) return ply_fail(GSX_ERR_PLY, "duplicate vertex element");
                out->count = strtoull(c, nullptr, 10);
                have_vertex = true;
            } else if (!have_vertex) {
                return ply_fail(GSX_ERR_PLY, "element '%s' precedes the vertex element (unsupported)", b);
            }
        } else if (!strcmp(a, "property") && in_vertex) {
            if (!strcmp(b, "list")) return ply_fail(GSX_ERR_PLY, "list property in the vertex element");
            int ts = type_size(b);
            if (ts < 0) return ply_fail(GSX_ERR_PLY, "unknown property type '%s'", b);
            for (int i = 0; i < 62; ++i)
                if (!strcmp(c, kProps[i])) {
                    if (ts != 4 || (strcmp(b, "float") && strcmp(b, "float32")))
                        return ply_fail(GSX_ERR_PLY, "property '%s' must be float32", c);
                    out->offsets[i] = out->is_ascii ? (int32_t)column : (int32_t)stride;
                }
            stride += (uint32_t)ts;
            ++column;
        }
    }
    if (!have_format || !have_vertex) return ply_fail(GSX_ERR_PLY, "PLY header lacks format or vertex element");
    const int required[] = {0, 1, 2, 6, 7, 8, 54, 55, 56, 57, 58, 59, 60, 61};
    for (int r : required)
        if (out->offsets[r] < 0) return ply_fail(GSX_ERR_PLY, "PLY vertex lacks property '%s'", kProps[r]);
    out->header_bytes = pos;
    out->vertex_bytes = out->is_ascii ? 0 : stride;
    if (out->is_ascii) out->vertex_bytes = column;  // columns per line for ascii
    return GSX_OK;
}

gsx_status gsx_ply_read_gaussians(const void* data, uint64_t size, const gsx_ply_header* h, uint64_t start, uint64_t n,
                                  gsx_gaussian* out) {
    using gsx::ply_fail;
    if (!data || !h || (n && !out)) return ply_fail(GSX_ERR_INVALID_ARG, "gsx_ply_read_gaussians: null argument");
    if (start > h->count || n > h->count - start) return ply_fail(GSX_ERR_INVALID_ARG, "gsx_ply_read_gaussians: range exceeds vertex count");
    const char* p = static_cast<const char*>(data);
    float v[62];
    if (!h->is_ascii) {
        const uint64_t need = h->header_bytes + (start + n) * (uint64_t)h->vertex_bytes;
        if (need > size) return ply_fail(GSX_ERR_IO, "PLY data truncated: need %llu bytes, have %llu", (unsigned long long)need, (unsigned long long)size);
        auto convert = [&](uint64_t i0, uint64_t i1) {
            float w[62];
            for (uint64_t i = i0; i < i1; ++i) {
                const char* row = p + h->header_bytes + (start + i) * (uint64_t)h->vertex_bytes;
                for (int k = 0; k < 62; ++k) {
                    if (h->offsets[k] >= 0) memcpy(&w[k], row + h->offsets[k], 4); else w[k] = 0.0f;
                }
                vertex_to_gaussian(w, out + i);
            }
        };
        // rows are independent: a large range is converted by several host threads (a 5.8 M-vertex file: 0.53 s on one core)
        const unsigned hw = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
        const uint64_t per = 1u << 17;
        const unsigned workers = (unsigned)std::min<uint64_t>(hw, n / per);
        if (workers <= 1) {
            convert(0, n);
            return GSX_OK;
        }
        // nothing may leave an extern "C" function by exception: a thread that cannot be created (pid limits) or an allocation
        // that fails ends the pool where it is, what was started is joined, and the calling thread converts the rest
        std::vector<std::thread> pool;
        unsigned started = 0;
        try {
            pool.reserve(workers);
            for (; started < workers; ++started) pool.emplace_back(convert, n * started / workers, n * (started + 1) / workers);
        } catch (...) {
        }
        if (started < workers) convert(n * started / workers, n);
        for (std::thread& t : pool) t.join();
        return GSX_OK;
    }
    // ascii: one vertex per line, whitespace separated
    uint64_t pos = h->header_bytes;
    std::vector<float> cols(h->vertex_bytes);
    for (uint64_t i = 0; i < start + n; ++i) {
        for (uint32_t c = 0; c < h->vertex_bytes; ++c) {
            while (pos < size && (p[pos] == ' ' || p[pos] == '\n' || p[pos] == '\r' || p[pos] == '\t')) ++pos;
            if (pos >= size) return ply_fail(GSX_ERR_IO, "ascii PLY data truncated at vertex %llu", (unsigned long long)i);
            char* end = nullptr;
            cols[c] = strtof(p + pos, &end);
            if (end == p + pos) return ply_fail(GSX_ERR_PLY, "ascii PLY: bad number at vertex %llu", (unsigned long long)i);
            pos = (uint64_t)(end - p);
        }
        if (i >= start) {
            for (int k = 0; k < 62; ++k) v[k] = h->offsets[k] >= 0 ? cols[h->offsets[k]] : 0.0f;
            vertex_to_gaussian(v, out + (i - start));
        }
    }
    return GSX_OK;
}

gsx_status gsx_ply_write(const gsx_gaussian* g, uint64_t n, const uint32_t* mask, const gsx_gaussian_edit* edits, void* out,
                         uint64_t capacity, uint64_t* out_size) {
    using gsx::ply_fail;
    if ((n && !g) || !out_size) return ply_fail(GSX_ERR_INVALID_ARG, "gsx_ply_write: null argument");
    auto keep = [&](uint64_t i) {
        if (mask && !((mask[i >> 5] >> (i & 31)) & 1u)) return false;
        return !(edits && (edits[i].flag & GSX_EDIT_ENABLED) && (edits[i].flag & GSX_EDIT_HIDDEN));
    };
    uint64_t kept = 0;
    for (uint64_t i = 0; i < n; ++i) kept += keep(i);
    std::string header = "ply\nformat binary_little_endian 1.0\nelement vertex " + std::to_string(kept) + "\n";
    for (int k = 0; k < 62; ++k) header += std::string("property float ") + kProps[k] + "\n";
    header += "end_header\n";
    *out_size = header.size() + kept * 248ull;
    if (!out) return GSX_OK;
    if (capacity < *out_size) return ply_fail(GSX_ERR_INVALID_ARG, "gsx_ply_write: buffer too small (%llu < %llu)", (unsigned long long)capacity, (unsigned long long)*out_size);
    char* p = static_cast<char*>(out);
    memcpy(p, header.data(), header.size());
    p += header.size();
    float v[62];
    for (uint64_t i = 0; i < n; ++i) {
        if (!keep(i)) continue;
        gaussian_to_vertex(g + i, v);
        if (edits && (edits[i].flag & GSX_EDIT_ENABLED)) {  // spec §7 Export: the colour ops baked into the DC colour and opacity
            float r = (float)g[i].color[0] / 255.0f, gg = (float)g[i].color[1] / 255.0f, b = (float)g[i].color[2] / 255.0f;
            float a = (float)g[i].color[3] / 255.0f;
            gsx::em_apply_edit(edits[i], r, gg, b, a);
            v[P_FDC] = (r - 0.5f) / kShC0;
            v[P_FDC + 1] = (gg - 0.5f) / kShC0;
            v[P_FDC + 2] = (b - 0.5f) / kShC0;
            a = std::min(std::max(a, 1e-6f), 1.0f - 1e-6f);
            v[P_OPACITY] = std::log(a / (1.0f - a));
        }
        memcpy(p, v, 248);
        p += 248;
    }
    return GSX_OK;
}

}  // extern "C"
