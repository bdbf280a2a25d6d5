// BASELINE.json configs[4] from C++: a camera orbit of diablo3_pose rendered in batches of views through
// render_views (tr_render_views), i.e. the reference's loop body (src/main.rs:57-116: clear_screen, camera,
// render_mesh_component) once per view, many views per call.
//
//   orbit_views <asset_dir> [views=512] [batch=16] [width=1920] [height=1080] [dump_view=-1] [out.rgba]
//
// eye_k = 2 (sin t, 0, cos t), t = 2 pi k / views.  With dump_view >= 0 that view's presented image
// is written as raw RGBA8 rows (row 0 = top), which tests/test_host_cpp.py compares with the oracle.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tinyrenderer/model_renderer.hpp"

using namespace tinyrenderer;

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <asset_dir> [views] [batch] [width] [height] [dump_view] [out.rgba]\n", argv[0]);
        return 2;
    }
    const std::string assets = argv[1];
    const int views = argc > 2 ? std::atoi(argv[2]) : 512;
    const int batch = argc > 3 ? std::atoi(argv[3]) : 16;
    const size_t W = argc > 4 ? (size_t)std::atoi(argv[4]) : 1920, H = argc > 5 ? (size_t)std::atoi(argv[5]) : 1080;
    const int dump_view = argc > 6 ? std::atoi(argv[6]) : -1;
    const char* out_path = argc > 7 ? argv[7] : nullptr;
    try {
        MeshComponent diablo = MeshComponent::from(assets + "/diablo_guy/diablo3_pose.obj", assets + "/diablo_guy/diablo3_pose_diffuse.png",
                                                   assets + "/diablo_guy/diablo3_pose_nm_tangent.png");
        ModelRendererState state(W, H, 0, batch < 2 ? 2 : batch);
        const double PI = 3.14159265358979323846;
        const Matrix4 perspective = make_perspective_matrix(10.0, 1.0, 0.35 * PI, (double)W / (double)H);  // main.rs:110-116
        std::vector<BuiltinShader> all;
        for (int k = 0; k < views; ++k) {
            const double t = 2.0 * PI * (double)k / (double)views;
            const Vector3 eye{2.0 * std::sin(t), 0.0, 2.0 * std::cos(t)};
            const Matrix4 view = look_at_rh(eye, {0.0, 0.0, 0.0}, {0.0, 1.0, 0.0});                      // main.rs:62-70
            all.push_back(BuiltinShader::make(perspective, view * diablo.local_to_world_matrix(), eye));   // main.rs:126-135
        }
        RgbaImage dumped;
        const auto t0 = std::chrono::steady_clock::now();
        for (int done = 0; done < views; done += batch) {
            const int n = std::min(batch, views - done);
            render_views(state, diablo, std::vector<BuiltinShader>(all.begin() + done, all.begin() + done + n));
            if (dump_view >= done && dump_view < done + n) {
                state.select_view(dump_view - done);
                dumped = state.present();
            }
        }
        state.synchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%d views of %zux%zu in batches of %d: %.1f us per view (%.0f views/s)\n", views, W, H, batch, us / views, 1e6 * views / us);
        if (out_path && !dumped.data.empty()) {
            FILE* f = std::fopen(out_path, "wb");
            if (!f) { std::fprintf(stderr, "cannot write %s\n", out_path); return 1; }
            std::fwrite(dumped.data.data(), 1, dumped.data.size(), f);
            std::fclose(f);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "orbit_views: %s\n", e.what());
        return 1;
    }
    return 0;
}
