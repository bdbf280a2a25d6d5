// The reference's frame loop (/root/reference/src/main.rs:10-309) on the HIP rasterizer, without
// the window: same scene (african_head + its inner eyes parented to it), same camera, same
// per-frame rotation 0.025 * time, same "Frametime (nosync)" print.
//
//   frame_loop <asset_dir> [frames=100] [width=853] [height=480] [out.rgba | out.png | -] [contexts=1] [batch=1]
//
// batch > 1: that many ITERATIONS of the loop are rendered as one batch of views of the two meshes
// (render_views_multi -> tr_render_views_multi: the head and the eye parented to it under the iteration's rotation,
// one plane slice per iteration) -- what a caller does that renders an animation ahead instead of frame by frame.
//
// contexts > 1: that many ModelRendererStates (one HIP stream each) take the frames in turn and the host
// waits only at the end, so that successive frames overlap on the GPU (what bench.py measures).
//
// With out.rgba the last frame is written as raw RGBA8 rows (row 0 = top), which is what
// tests/test_host_cpp.py compares with the oracle.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "tinyrenderer/model_renderer.hpp"

using namespace tinyrenderer;

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <asset_dir> [frames] [width] [height] [out.rgba]\n", argv[0]);
        return 2;
    }
    const std::string assets = argv[1];
    const int frames = argc > 2 ? std::atoi(argv[2]) : 100;
    const size_t frame_width = argc > 3 ? (size_t)std::atoi(argv[3]) : 853;   // main.rs:41-42
    const size_t frame_height = argc > 4 ? (size_t)std::atoi(argv[4]) : 480;
    const char* out_path = argc > 5 && std::strcmp(argv[5], "-") != 0 ? argv[5] : nullptr;
    const int n_contexts = argc > 6 ? std::max(1, std::atoi(argv[6])) : 1;
    try {
        // main.rs:12-39
        MeshComponent african_head = MeshComponent::from(assets + "/african_head/african_head.obj",
                                                         assets + "/african_head/african_head_diffuse.png",
                                                         assets + "/african_head/african_head_nm_tangent.png");
        MeshComponent eye_inner = MeshComponent::from(assets + "/african_head/african_head_eye_inner.obj",
                                                      assets + "/african_head/african_head_eye_inner_diffuse.png",
                                                      assets + "/african_head/african_head_eye_inner_nm_tangent.png");
        eye_inner.parent = &african_head;

        const int batch = argc > 7 ? std::max(1, std::atoi(argv[7])) : 1;
        const double PI = 3.14159265358979323846;
        if (batch > 1) {
            ModelRendererState state(frame_width, frame_height, 0, std::max(2, batch));
            const Vector3 camera_pos{0.0 * 2.0, 0.0 * 2.0, 1.0 * 2.0};                                                   // main.rs:62-70
            const Matrix4 camera_direction_matrix = look_at_rh(camera_pos, {0.0, 0.0, 0.0}, {0.0, 1.0, 0.0});
            const Matrix4 perspective_matrix =
                make_perspective_matrix(10.0, 1.0, 0.35 * PI, (double)frame_width / (double)frame_height);               // main.rs:110-116
            const auto loop_begin = std::chrono::steady_clock::now();
            int last_slice = 0;
            for (int time0 = 1; time0 <= frames; time0 += batch) {
                const int n = std::min(batch, frames - time0 + 1);
                std::vector<std::vector<BuiltinShader>> shaders;
                for (int time = time0; time < time0 + n; ++time) {
                    african_head.transform.set_rotation({0.025 * (double)time, 0.0, 0.0});                              // main.rs:92-94
                    shaders.push_back({BuiltinShader::make(perspective_matrix, camera_direction_matrix * african_head.local_to_world_matrix(), camera_pos),
                                       BuiltinShader::make(perspective_matrix, camera_direction_matrix * eye_inner.local_to_world_matrix(), camera_pos)});
                }
                render_views_multi(state, {&african_head, &eye_inner}, shaders);                                        // main.rs:272-273, n iterations
                last_slice = n - 1;
            }
            state.synchronize();
            const double loop_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - loop_begin).count();
            std::printf("%d frames of %zux%zu in batches of %d iterations: %.3f ms per frame (%.0f frames/s)\n", frames, frame_width,
                        frame_height, batch, loop_us / frames / 1000.0, 1e6 * frames / loop_us);
            if (out_path) {
                state.select_view(last_slice);
                const RgbaImage img = state.present();
                const std::string op = out_path;
                if (op.size() > 4 && op.compare(op.size() - 4, 4, ".png") == 0) {
                    write_rgba_img_to_file(op, img);
                } else if (FILE* f = std::fopen(out_path, "wb")) {
                    std::fwrite(img.data.data(), 1, img.data.size(), f);
                    std::fclose(f);
                } else {
                    std::fprintf(stderr, "cannot write %s\n", out_path);
                    return 1;
                }
            }
            return 0;
        }

        std::vector<std::unique_ptr<ModelRendererState>> states;
        for (int k = 0; k < n_contexts; ++k) {
            states.emplace_back(new ModelRendererState(frame_width, frame_height));
            clear_screen(*states.back());
        }

        double total_us = 0.0;
        const auto loop_begin = std::chrono::steady_clock::now();
        for (int time = 1; time <= frames; ++time) {  // main.rs:57-60
            ModelRendererState& model_renderer_state = *states[(size_t)(time % n_contexts)];
            const auto before = std::chrono::steady_clock::now();
            clear_screen(model_renderer_state);

            // main.rs:62-70
            const Vector3 camera_pos{0.0 * 2.0, 0.0 * 2.0, 1.0 * 2.0};
            const Matrix4 camera_direction_matrix = look_at_rh(camera_pos, {0.0, 0.0, 0.0}, {0.0, 1.0, 0.0});
            // main.rs:92-94
            african_head.transform.set_rotation({0.025 * (double)time, 0.0, 0.0});
            // main.rs:110-116
            const Matrix4 perspective_matrix =
                make_perspective_matrix(10.0, 1.0, 0.35 * PI, (double)frame_width / (double)frame_height);

            auto do_render_mesh_component = [&](const MeshComponent& mc) {  // main.rs:118-143
                const Matrix4 model_view_matrix = camera_direction_matrix * mc.local_to_world_matrix();
                render_mesh_component(model_renderer_state, mc, BuiltinShader::make(perspective_matrix, model_view_matrix, camera_pos));
            };
            do_render_mesh_component(african_head);  // main.rs:272-273
            do_render_mesh_component(eye_inner);
            model_renderer_state.resolve();
            if (n_contexts == 1) model_renderer_state.synchronize();

            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - before).count();
            total_us += us;
            if (n_contexts == 1 && (time <= 3 || time == frames))  // main.rs:289-295
                std::printf("\n\"Frametime (nosync)\": %.2fms (%.2f fps)\n", us / 1000.0, 1000000.0 / us);
        }
        for (auto& st : states) st->synchronize();
        const double loop_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - loop_begin).count();
        if (n_contexts == 1)
            std::printf("%d frames of %zux%zu: mean %.3f ms\n", frames, frame_width, frame_height, total_us / frames / 1000.0);
        else
            std::printf("%d frames of %zux%zu, %d contexts in flight: %.3f ms per frame (%.0f frames/s)\n", frames, frame_width,
                        frame_height, n_contexts, loop_us / frames / 1000.0, 1e6 * frames / loop_us);
        if (out_path) {
            const RgbaImage img = states[(size_t)(frames % n_contexts)]->present();
            const std::string op = out_path;
            if (op.size() > 4 && op.compare(op.size() - 4, 4, ".png") == 0) {
                write_rgba_img_to_file(op, img);
            } else if (FILE* f = std::fopen(out_path, "wb")) {
                std::fwrite(img.data.data(), 1, img.data.size(), f);
                std::fclose(f);
            } else {
                std::fprintf(stderr, "cannot write %s\n", out_path);
                return 1;
            }
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "frame_loop: %s\n", e.what());
        return 1;
    }
    return 0;
}
