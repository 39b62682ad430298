// tools/frame_driver.cpp — replays the app's per-frame protocol (src/tab/scene.rs:265-571)
// through the C++ gs:: facade (include/gsx.hpp): streaming upload in batches of 1000 (scene.rs:358-375),
// uniforms, preprocess + sort per visible model, poll, render far -> near, readback.  Prints statistics and
// an FNV-1a checksum of the float framebuffer; tests/test_gpu_cpp_driver.py builds the same scene in Python
// and requires the identical checksum.
#include <cstdio>
#include <cstring>
#include <thread>

#include "../include/gsx.hpp"

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
static float u01(uint32_t& s) { return (float)(lcg(s) >> 8) * (1.0f / 16777216.0f); }

static std::vector<gs::Gaussian> make_scene(size_t n, uint32_t seed) {
    std::vector<gs::Gaussian> g(n);
    uint32_t s = seed;
    for (auto& x : g) {
        float q[4];
        float l = 0;
        for (float& c : q) { c = u01(s) * 2 - 1; l += c * c; }
        l = std::sqrt(l);
        for (int k = 0; k < 4; ++k) x.rot[k] = q[k] / l;
        for (int k = 0; k < 3; ++k) x.pos[k] = u01(s) * 6 - 3;
        for (int k = 0; k < 4; ++k) x.color[k] = (uint8_t)(lcg(s) >> 24);
        for (int c = 0; c < 15; ++c)
            for (int ch = 0; ch < 3; ++ch) x.sh[c][ch] = (u01(s) - 0.5f) * 0.3f;
        for (int k = 0; k < 3; ++k) x.scale[k] = 0.02f + 0.2f * u01(s);
    }
    return g;
}

int main(int argc, char** argv) {
    try {
        const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 20000;
        const uint32_t w = 320, h = 200;
        auto a = make_scene(n, 12345u), b = make_scene(n / 2, 777u);
        auto viewer = gs::MultiModelViewer::new_with(0, {1, 1});
        struct M { const char* key; std::vector<gs::Gaussian>* g; gs::Vec3 pos; gs::Vec3 rot_deg; gs::Vec3 scale; };
        M models[2] = {{"a", &a, {0, 0, 0}, {0, 0, 0}, {1, 1, 1}}, {"b", &b, {0.5f, 0.2f, 2.0f}, {10, 30, -20}, {1.1f, 0.9f, 1.0f}}};
        for (auto& m : models) {
            auto& model = viewer.add_model(m.key, m.g->size());
            for (size_t start = 0; start < m.g->size(); start += 1000)  // loader batches, scene.rs:358-375
                model.gaussian_buffers.gaussians_buffer.update_range(start, m.g->data() + start, std::min<size_t>(1000, m.g->size() - start));
        }
        gs::CameraOrbitControl cam;
        cam.pos = {2.0f, 1.5f, -6.0f};
        viewer.update_camera(cam, {w, h});
        const float d2r = 0.017453292519943295f;
        for (auto& m : models)
            viewer.update_model_transform(m.key, m.pos, gs::quat_from_euler_zyx(m.rot_deg[2] * d2r, m.rot_deg[1] * d2r, m.rot_deg[0] * d2r), m.scale);
        viewer.update_gaussian_transform(1.0f, gs::GaussianDisplayMode::Splat, *gs::GaussianShDegree::new_(3), false);
        for (auto& m : models) {
            viewer.preprocessor.preprocess(m.key);
            viewer.radix_sorter.sort(m.key);
        }
        viewer.poll();
        // far -> near by squared distance of the model origin to the camera (scene.rs:533-558)
        auto dist = [&](const M& m) { gs::Vec3 d = gs::sub(m.pos, cam.pos); return gs::dot(d, d); };
        std::vector<std::string> keys = dist(models[0]) >= dist(models[1]) ? std::vector<std::string>{"a", "b"} : std::vector<std::string>{"b", "a"};
        viewer.renderer.render(keys);
        auto fb = viewer.download_framebuffer();
        uint64_t hash = 1469598103934665603ull;
        const unsigned char* bytes = reinterpret_cast<const unsigned char*>(fb.data());
        for (size_t i = 0; i < fb.size() * 4; ++i) { hash ^= bytes[i]; hash *= 1099511628211ull; }
        if (argc > 2) {  // raw float32 [h][w][4] for the comparison with the Python mirror
            FILE* f = fopen(argv[2], "wb");
            if (!f || fwrite(fb.data(), sizeof(float), fb.size(), f) != fb.size()) return 5;
            fclose(f);
        }
        auto sa = viewer.frame_stats("a"), sb = viewer.frame_stats("b");
        printf("frame_driver n_a=%llu vis_a=%llu n_b=%llu vis_b=%llu order=%s,%s fnv=%016llx\n", (unsigned long long)sa.n_gaussians,
               (unsigned long long)sa.n_visible, (unsigned long long)sb.n_gaussians, (unsigned long long)sb.n_visible, keys[0].c_str(),
               keys[1].c_str(), (unsigned long long)hash);
        // a second frame through the selection protocol (scene.rs:785-835, 601-611): rect query -> postprocess -> selection bits
        gsx_query q{};
        q.kind = GSX_QUERY_RECT;
        q.selection_op = GSX_SELECTION_SET;
        q.p0[0] = 60.0f; q.p0[1] = 40.0f; q.p1[0] = 250.0f; q.p1[1] = 160.0f;
        viewer.update_query(q);
        viewer.update_selection_highlight({1.0f, 0.0f, 1.0f, 0.5f});
        for (auto& m : models) {
            viewer.preprocessor.preprocess(m.key);
            viewer.radix_sorter.sort(m.key);
        }
        viewer.renderer.render(keys);
        for (auto& m : models) viewer.postprocessor.postprocess(m.key);
        viewer.poll();
        size_t selected = 0;
        for (auto& m : models)
            for (uint32_t word : viewer.models.at(m.key).gaussian_buffers.selection_buffer.download(m.g->size())) selected += (size_t)__builtin_popcount(word);
        printf("frame_driver selected=%zu\n", selected);
        // the export path (app.rs:769-816): clone the selection / mask buffers, download the clones on spawned threads while this
        // thread renders one more frame that CHANGES the selection — the clones still hold what was there when they were taken
        {
            auto& bufs = viewer.models.at("a").gaussian_buffers;
            const std::vector<uint32_t> want_sel = bufs.selection_buffer.download(models[0].g->size());
            std::vector<uint32_t> got_sel, got_sel2, got_mask;
            {
                auto sel_clone = bufs.selection_buffer.clone();
                auto mask_clone = bufs.mask_buffer.clone();
                auto sel_clone2 = sel_clone;  // Clone of a clone
                std::thread t1([&] { got_sel = sel_clone.download(); got_mask = mask_clone.download(); });
                std::thread t2([&] { got_sel2 = sel_clone2.download(); });
                struct Join { std::thread& a; std::thread& b; ~Join() { a.join(); b.join(); } } join{t1, t2};
                q.selection_op = GSX_SELECTION_REMOVE;  // meanwhile: a frame that takes the rectangle OUT of the selection again
                viewer.update_query(q);
                for (auto& m : models) {
                    viewer.preprocessor.preprocess(m.key);
                    viewer.radix_sorter.sort(m.key);
                }
                viewer.renderer.render(keys);
                for (auto& m : models) viewer.postprocessor.postprocess(m.key);
                viewer.poll();
            }
            size_t left = 0;
            for (uint32_t word : bufs.selection_buffer.download(models[0].g->size())) left += (size_t)__builtin_popcount(word);
            bool mask_all_ones = !got_mask.empty();
            for (uint32_t wd : got_mask) mask_all_ones = mask_all_ones && wd == 0xFFFFFFFFu;
            const bool ok = left == 0 && got_sel == want_sel && got_sel2 == want_sel && mask_all_ones;
            printf("frame_driver clones=%s\n", ok ? "ok" : "MISMATCH");
            if (!ok) return 6;
        }
        // error convention: a missing model is a gs::Error, not a crash
        try { viewer.preprocessor.preprocess("missing"); return 2; } catch (const gs::Error& e) { if (e.status != GSX_ERR_NOT_FOUND) return 3; }
        if (gs::GaussianShDegree::new_(4)) return 4;
        return 0;
    } catch (const gs::Error& e) {
        fprintf(stderr, "gs::Error %d: %s\n", (int)e.status, e.what());
        return 1;
    }
}
