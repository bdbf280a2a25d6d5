// Host side of the drop-in, C++17, header only: the reference's `model_renderer` module
// (/root/reference/src/model_renderer/mod.rs) over the C ABI of include/tinyrenderer_hip.h.
// `ModelRendererState`, `clear_screen`, `render_mesh_component`, `MeshComponent` and `Transform`
// keep the reference's names and call shape.  The two shader closures of render_mesh_component
// (mod.rs:225-230: host code, cannot run on a GPU) are replaced by `BuiltinShader`, the values the
// one closure pair the reference ships captures (src/main.rs:118-267).  Nothing here computes
// pixels: every call ends in the library, and without a gfx950 GPU construction throws.
#pragma once
#include <map>
#include <memory>

#include "helpers.hpp"

namespace tinyrenderer {

struct RendererError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// src/model_renderer/mod.rs:21-27,87-148 (the Cell<> fields become plain members)
class Transform {
  public:
    Vector3 position() const { return position_; }
    Vector3 rotation() const { return rotation_; }
    Vector3 scale() const { return scale_; }
    Matrix4 matrix() const { return matrix_; }
    void set_position(const Vector3& p) {  // mod.rs:118-125: pokes m14 / m24 / m34 only, no resync
        position_ = p;
        matrix_.at(0, 3) = p.x; matrix_.at(1, 3) = p.y; matrix_.at(2, 3) = p.z;
    }
    void set_rotation(const Vector3& r) { rotation_ = r; resync_matrix(); }
    void set_scale(const Vector3& s) { scale_ = s; resync_matrix(); }

  private:
    void resync_matrix() {  // mod.rs:140-147: T * R * S, left-associative
        matrix_ = make_translation_matrix(position_) * make_rotation_matrix(rotation_.x, rotation_.y, rotation_.z) *
                  make_scale_matrix(scale_);
    }
    Vector3 position_{0.0, 0.0, 0.0}, rotation_{0.0, 0.0, 0.0}, scale_{1.0, 1.0, 1.0};
    Matrix4 matrix_ = Matrix4::identity();
};

// src/model_renderer/mod.rs:12-19,37-85.  texture / normal_map are held already flipped
// vertically, like the reference's flip_vertically(load_nd_rgba_img_from_file(..)).
struct MeshComponent {
    Object mesh;
    Transform transform;
    RgbaImage texture, normal_map;
    const MeshComponent* parent = nullptr;

    MeshComponent() = default;
    MeshComponent(Object obj, RgbaImage tex, RgbaImage nm) : mesh(std::move(obj)), texture(std::move(tex)), normal_map(std::move(nm)) {}
    // MeshComponent::from((obj_path, texture_path, normal_map_path)), mod.rs:37-50
    static MeshComponent from(const std::string& obj_path, const std::string& texture_path, const std::string& normal_map_path) {
        return MeshComponent(load_model(obj_path), load_rgba_img_from_file(texture_path, /*flipped=*/true),
                             load_rgba_img_from_file(normal_map_path, /*flipped=*/true));
    }
    // mod.rs:67-84: fold identity * self * parent * grandparent ...
    Matrix4 local_to_world_matrix() const {
        Matrix4 acc = Matrix4::identity();
        for (const MeshComponent* c = this; c; c = c->parent) acc = acc * c->transform.matrix();
        return acc;
    }
};

// What the vertex + fragment closures of src/main.rs:146-267 capture.
struct BuiltinShader {
    Matrix4 perspective_matrix, model_view_matrix, inverse_model_view_matrix;
    Vector3 camera_pos;
    bool use_normal_map = true;         // main.rs:182
    double light_intensity = 1.0;       // main.rs:254
    double ambient_light = 1.0;         // main.rs:255
    Color vertex_color = WHITE;         // main.rs:170
    // main.rs:126-135: normal matrix = model_view.transpose().try_inverse() or identity
    static BuiltinShader make(const Matrix4& perspective, const Matrix4& model_view, const Vector3& camera_pos) {
        BuiltinShader s;
        s.perspective_matrix = perspective;
        s.model_view_matrix = model_view;
        auto inv = model_view.transpose().try_inverse();
        s.inverse_model_view_matrix = inv ? *inv : Matrix4::identity();
        s.camera_pos = camera_pos;
        return s;
    }
};

// src/model_renderer/mod.rs:29-35,150-173.  The buffers live on the GPU; the accessors read them
// back in the reference's layout (Array3<f64> of shape (W, H, C), x the slow axis).
class ModelRendererState {
  public:
    // max_views > 1: a state whose planes hold that many W x H slices, for render_views (tr_create_views)
    ModelRendererState(size_t frame_width, size_t frame_height, int device = 0, int max_views = 1)
        : w_(frame_width), h_(frame_height), max_views_(max_views) {
        const int rc = max_views > 1 ? tr_create_views(device, (int)frame_width, (int)frame_height, max_views, &ctx_)
                                     : tr_create(device, (int)frame_width, (int)frame_height, &ctx_);
        if (rc != TR_OK)
            throw RendererError("ModelRendererState::new: tr_create failed (" + std::to_string(rc) +
                                "); a gfx950 GPU is required, there is no CPU fallback");
    }
    ~ModelRendererState() {
        if (!ctx_) return;
        for (auto& kv : meshes_) tr_mesh_destroy(ctx_, kv.second);
        for (auto& kv : textures_) tr_texture_destroy(ctx_, kv.second);
        tr_destroy(ctx_);
    }
    ModelRendererState(const ModelRendererState&) = delete;
    ModelRendererState& operator=(const ModelRendererState&) = delete;

    size_t frame_width() const { return w_; }
    size_t frame_height() const { return h_; }
    int max_views() const { return max_views_; }
    tr_context* context() { return ctx_; }
    // the slice of a render_views batch that z_buffer / frame_buffer / present refer to
    void select_view(int view) { check(tr_select_view(ctx_, view), "tr_select_view"); }
    // colour arithmetic of the fragment closure (src/main.rs:173-267): fused (+-1 LSB, the default) or operation for operation
    void set_colour_exact(bool exact) { check(tr_set_colour_exact(ctx_, exact ? 1 : 0), "tr_set_colour_exact"); }

    // state.z_buffer.nd_img: (W, H, 1) f64, -inf where nothing was drawn
    std::vector<double> z_buffer() {
        std::vector<double> v(w_ * h_);
        check(tr_read_z_f64(ctx_, v.data(), TR_LAYOUT_REFERENCE), "tr_read_z_f64");
        return v;
    }
    // state.frame_buffer.nd_img: (W, H, 4) f64
    std::vector<double> frame_buffer() {
        std::vector<double> v(w_ * h_ * 4);
        check(tr_read_rgba_f64(ctx_, v.data(), TR_LAYOUT_REFERENCE), "tr_read_rgba_f64");
        return v;
    }
    // ndarray_to_image_rgba_and_flip(&frame_buffer) (helpers/mod.rs:204-222): RGBA8 rows, row 0 = top
    RgbaImage present() {
        RgbaImage img;
        img.width = (int)w_; img.height = (int)h_;
        img.data.resize(w_ * h_ * 4);
        check(tr_read_rgba8(ctx_, img.data.data(), 1), "tr_read_rgba8");
        return img;
    }
    // shade without reading back (the frame loop's end of frame when nothing is displayed)
    void resolve() { check(tr_resolve(ctx_, TR_OUT_RGBA8), "tr_resolve"); }
    void synchronize() { check(tr_synchronize(ctx_), "tr_synchronize"); }

    void check(int rc, const char* what) const {
        if (rc != TR_OK) throw RendererError(std::string(what) + ": " + tr_last_error(ctx_));
    }
    // device copies are made once per mesh / image object (keyed by address: keep them alive)
    tr_mesh* mesh_handle(const Object& o) {
        auto it = meshes_.find(&o);
        if (it != meshes_.end()) return it->second;
        tr_mesh* h = nullptr;
        check(tr_mesh_create(ctx_, o.vertices.data(), (int64_t)(o.vertices.size() / 3), o.normals.empty() ? nullptr : o.normals.data(),
                             (int64_t)(o.normals.size() / 3), o.tex_vertices.empty() ? nullptr : o.tex_vertices.data(),
                             (int64_t)(o.tex_vertices.size() / 3), o.triangles.empty() ? nullptr : o.triangles.data(),
                             (int64_t)o.n_triangles(), &h),
              "tr_mesh_create");
        meshes_[&o] = h;
        return h;
    }
    tr_texture* texture_handle(const RgbaImage& img) {
        if (img.data.empty()) return nullptr;
        auto it = textures_.find(&img);
        if (it != textures_.end()) return it->second;
        tr_texture* h = nullptr;
        check(tr_texture_create(ctx_, img.data.data(), img.width, img.height, &h), "tr_texture_create");
        textures_[&img] = h;
        return h;
    }

  private:
    tr_context* ctx_ = nullptr;
    size_t w_, h_;
    int max_views_ = 1;
    std::map<const Object*, tr_mesh*> meshes_;
    std::map<const RgbaImage*, tr_texture*> textures_;
};

// src/model_renderer/mod.rs:210-219
inline void clear_screen(ModelRendererState& state) { state.check(tr_clear(state.context()), "tr_clear"); }

// ---- EXTENSION shaders (tr_shade_ext in tinyrenderer_hip.h; the reference has only the shader of
// main.rs:146-267): Gouraud / Phong switch, specular map, shadow map.
class ShadowMap {  // the z plane of a state in which the scene was drawn from the light
  public:
    explicit ShadowMap(ModelRendererState& light_state) {
        light_state.check(tr_shadow_map_create(light_state.context(), &map_), "tr_shadow_map_create");
    }
    ShadowMap(const ShadowMap&) = delete;
    ShadowMap& operator=(const ShadowMap&) = delete;
    ~ShadowMap() { tr_shadow_map_destroy(map_); }
    // a new snapshot of `light_state` (same size) into this map: no allocation, no host wait
    void update(ModelRendererState& light_state) { light_state.check(tr_shadow_map_update(map_, light_state.context()), "tr_shadow_map_update"); }
    const tr_shadow_map* handle() const { return map_; }

  private:
    tr_shadow_map* map_ = nullptr;
};

struct ShadeExt {
    int model = TR_SHADE_REFERENCE;
    const RgbaImage* specular_map = nullptr;  // flipped like the other textures; channel 0 = exponent
    double shininess = 16.0, specular_strength = 0.6;
    const ShadowMap* shadow = nullptr;
    Matrix4 shadow_from_view = Matrix4::identity();  // light viewport * light pmv * inverse(viewport * pmv)
    double shadow_bias = 0.0, shadow_dim = 0.3;
};

inline tr_draw_params make_draw_params(ModelRendererState& state, const MeshComponent& mesh_component, const BuiltinShader& shader);

// One iteration of the reference's loop per shader -- clear_screen + render_mesh_component(state, &mesh, VS, FS)
// (src/main.rs:57-116,272-273) -- for ONE mesh under shaders.size() cameras, as a single batch (tr_render_views):
// view v lands in slice v of a state made with max_views; select_view(v) picks the slice the buffers read.
inline void render_views(ModelRendererState& state, const MeshComponent& mesh_component, const std::vector<BuiltinShader>& shaders,
                         uint32_t out_mask = TR_OUT_RGBA8) {
    std::vector<tr_draw_params> params;
    params.reserve(shaders.size());
    for (const BuiltinShader& sh : shaders) params.push_back(make_draw_params(state, mesh_component, sh));
    state.check(tr_render_views(state.context(), state.mesh_handle(mesh_component.mesh), params.data(), (int)params.size(), out_mask),
                "tr_render_views");
}

// The same for the loop's own scene of several meshes per iteration (main.rs:272-273: the head, then the eye parented to
// it): shaders[v][m] is mesh m's shader under view v (its model_view holds local_to_world_matrix with the parents folded
// in, mod.rs:67-84); one batch (tr_render_views_multi), one slice per view.
inline void render_views_multi(ModelRendererState& state, const std::vector<const MeshComponent*>& mesh_components,
                               const std::vector<std::vector<BuiltinShader>>& shaders, uint32_t out_mask = TR_OUT_RGBA8) {
    std::vector<tr_draw_params> params;
    std::vector<const tr_mesh*> meshes;
    for (const MeshComponent* mc : mesh_components) meshes.push_back(state.mesh_handle(mc->mesh));
    for (const auto& row : shaders) {
        if (row.size() != mesh_components.size()) throw RendererError("render_views_multi: one shader per mesh and view");
        for (size_t m = 0; m < row.size(); ++m) params.push_back(make_draw_params(state, *mesh_components[m], row[m]));
    }
    state.check(tr_render_views_multi(state.context(), (int)meshes.size(), meshes.data(), params.data(), (int)shaders.size(), out_mask),
                "tr_render_views_multi");
}

inline tr_draw_params make_draw_params(ModelRendererState& state, const MeshComponent& mesh_component, const BuiltinShader& shader) {
    tr_draw_params p{};
    const Matrix4 pmv = shader.perspective_matrix * shader.model_view_matrix;  // main.rs:152-153
    for (int i = 0; i < 16; ++i) {
        p.pmv[i] = pmv.m[i];
        p.normal_mat[i] = shader.inverse_model_view_matrix.m[i];
    }
    p.light_dir[0] = shader.camera_pos.x; p.light_dir[1] = shader.camera_pos.y; p.light_dir[2] = shader.camera_pos.z;
    p.ambient = shader.ambient_light;
    p.intensity = shader.light_intensity;
    for (int i = 0; i < 4; ++i) p.vertex_color[i] = shader.vertex_color[i];
    p.use_normal_map = shader.use_normal_map ? 1 : 0;
    p.diffuse = state.texture_handle(mesh_component.texture);
    p.normal_map = state.texture_handle(mesh_component.normal_map);
    return p;
}

// render_mesh_component(state, &mesh, VS, FS) (mod.rs:225-230) with the closure pair replaced by
// `shader`; takes the MeshComponent so the textures the closures capture (main.rs:140-141) come along.
// `ext` (no counterpart in the reference) selects an extension shader.
inline void render_mesh_component(ModelRendererState& state, const MeshComponent& mesh_component, const BuiltinShader& shader,
                                  const ShadeExt* ext = nullptr) {
    const tr_draw_params p = make_draw_params(state, mesh_component, shader);
    if (!ext) {
        state.check(tr_draw(state.context(), state.mesh_handle(mesh_component.mesh), &p), "tr_draw");
        return;
    }
    tr_shade_ext x{};
    x.model = ext->model;
    x.specular_map = ext->specular_map ? state.texture_handle(*ext->specular_map) : nullptr;
    x.shininess = ext->shininess;
    x.specular_strength = ext->specular_strength;
    x.shadow = ext->shadow ? ext->shadow->handle() : nullptr;
    for (int i = 0; i < 16; ++i) x.shadow_from_view[i] = ext->shadow_from_view.m[i];
    x.shadow_bias = ext->shadow_bias;
    x.shadow_dim = ext->shadow_dim;
    state.check(tr_draw_ext(state.context(), state.mesh_handle(mesh_component.mesh), &p, &x), "tr_draw_ext");
}

}  // namespace tinyrenderer
