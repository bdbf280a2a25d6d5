// Host side of the drop-in, C++17, header only: the reference's `helpers` module
// (/root/reference/src/helpers/mod.rs) over the C ABI of include/tinyrenderer_hip.h.
// Same names and argument meaning as the Rust functions.  Matrices are nalgebra's Matrix4<f64>:
// column-major storage, and every product is evaluated in nalgebra 0.30.1's association (one gemv
// per result column, accumulating left to right) so that the matrices handed to the GPU are the
// ones the reference would build.  [nalgebra-unpinned]: restated from the crate's published
// source; the reference holds no fixture for them, and they are inputs to both sides of every
// parity comparison.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../tinyrenderer_hip.h"

namespace tinyrenderer {

using Color = std::array<double, 4>;  // src/helpers/mod.rs:22-26
inline constexpr Color BLACK{0.0, 0.0, 0.0, 255.0};
inline constexpr Color RED{255.0, 0.0, 0.0, 255.0};
inline constexpr Color GREEN{0.0, 255.0, 0.0, 255.0};
inline constexpr Color BLUE{0.0, 0.0, 255.0, 255.0};
inline constexpr Color WHITE{255.0, 255.0, 255.0, 255.0};

struct Vector3 {
    double x = 0.0, y = 0.0, z = 0.0;
};

// nalgebra::Matrix4<f64>: m[c * 4 + r]
struct Matrix4 {
    double m[16];
    double& at(int r, int c) { return m[c * 4 + r]; }
    double at(int r, int c) const { return m[c * 4 + r]; }
    static Matrix4 identity() {
        Matrix4 o{};
        for (int i = 0; i < 4; ++i) o.at(i, i) = 1.0;
        return o;
    }
    // Matrix4::new(m11, m12, ...) takes its arguments row by row
    static Matrix4 from_rows(const double (&r)[16]) {
        Matrix4 o{};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) o.at(i, j) = r[i * 4 + j];
        return o;
    }
    Matrix4 transpose() const {
        Matrix4 o{};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) o.at(i, j) = at(j, i);
        return o;
    }
    // nalgebra `Matrix4::try_inverse` (do_inverse4: the MESA cofactor expansion on the column-major
    // slice); nullopt when the determinant is exactly zero.
    std::optional<Matrix4> try_inverse() const {
        const double* a = m;
        double inv[16];
        inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
        inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
        inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
        inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
        inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
        inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
        inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
        inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
        inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
        inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
        inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
        inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
        inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
        inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
        inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
        inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
        const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
        if (det == 0.0) return std::nullopt;
        const double inv_det = 1.0 / det;
        Matrix4 o{};
        for (int i = 0; i < 16; ++i) o.m[i] = inv[i] * inv_det;
        return o;
    }
};

// nalgebra `a * b`: C[r][j] = ((a[r][0] b[0][j] + a[r][1] b[1][j]) + a[r][2] b[2][j]) + a[r][3] b[3][j]
inline Matrix4 operator*(const Matrix4& a, const Matrix4& b) {
    Matrix4 c{};
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r) {
            double acc = a.at(r, 0) * b.at(0, j);
            for (int k = 1; k < 4; ++k) acc = a.at(r, k) * b.at(k, j) + acc;
            c.at(r, j) = acc;
        }
    return c;
}

// src/helpers/mod.rs:520-525
inline Matrix4 make_rotation_matrix(double pitch, double yaw, double roll) {
    const double cp = std::cos(pitch), sp = std::sin(pitch);
    const double cy = std::cos(yaw), sy = std::sin(yaw);
    const double cr = std::cos(roll), sr = std::sin(roll);
    const double rows[16] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, 0.0,
                             sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, 0.0,
                             -sp,     cp * sr,                cp * cr,                0.0,
                             0.0,     0.0,                    0.0,                    1.0};
    return Matrix4::from_rows(rows);
}
// src/helpers/mod.rs:527-532
inline Matrix4 make_translation_matrix(const Vector3& t) {
    Matrix4 o = Matrix4::identity();
    o.at(0, 3) = t.x; o.at(1, 3) = t.y; o.at(2, 3) = t.z;
    return o;
}
// src/helpers/mod.rs:534-539
inline Matrix4 make_scale_matrix(const Vector3& s) {
    Matrix4 o = Matrix4::identity();
    o.at(0, 0) = s.x; o.at(1, 1) = s.y; o.at(2, 2) = s.z;
    return o;
}
// src/helpers/mod.rs:542-556 (the reference calls it with near = 10, far = 1)
inline Matrix4 make_perspective_matrix(double near_plane_distance, double far_plane_distance, double vertical_fov,
                                       double aspect_ratio) {
    const double n = near_plane_distance, f = far_plane_distance;
    const double r = n * std::tan(vertical_fov / 2.0) * aspect_ratio;
    const double t = n * std::tan(vertical_fov / 2.0);
    const double rows[16] = {n / r, 0.0,   0.0,                      0.0,
                             0.0,   n / t, 0.0,                      0.0,
                             0.0,   0.0,   -1.0 * (f + n) / (f - n), (-2.0 * f * n) / (f - n),
                             0.0,   0.0,   -1.0,                     0.0};
    return Matrix4::from_rows(rows);
}

namespace detail {
inline Vector3 normalize3(const Vector3& v) {  // nalgebra: v / sqrt(0 + ((x x + y y) + z z))
    const double n = std::sqrt(0.0 + ((v.x * v.x + v.y * v.y) + v.z * v.z));
    return {v.x / n, v.y / n, v.z / n};
}
inline Vector3 cross3(const Vector3& a, const Vector3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
}  // namespace detail

// nalgebra `Matrix4::look_at_rh(&eye, &target, &up)` as used at src/main.rs:66-70
inline Matrix4 look_at_rh(const Vector3& eye, const Vector3& target, const Vector3& up) {
    const Vector3 d{target.x - eye.x, target.y - eye.y, target.z - eye.z};
    const Vector3 z = detail::normalize3({-d.x, -d.y, -d.z});
    const Vector3 x = detail::normalize3(detail::cross3(up, z));
    const Vector3 y = detail::normalize3(detail::cross3(z, x));
    const double rot[3][3] = {{x.x, x.y, x.z}, {y.x, y.y, y.z}, {z.x, z.y, z.z}};
    const double ne[3] = {-eye.x, -eye.y, -eye.z};
    Matrix4 o = Matrix4::identity();
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) o.at(r, c) = rot[r][c];
        o.at(r, 3) = (rot[r][0] * ne[0] + rot[r][1] * ne[1]) + rot[r][2] * ne[2];
    }
    return o;
}

// ---------------------------------------------------------------------------- loading surface
constexpr uint32_t NONE_INDEX = 0xFFFFFFFFu;

// The part of wavefront_obj::obj::Object that render_mesh_component consumes
// (src/model_renderer/mod.rs:237-276), in the layout tr_mesh_create takes.
struct Object {
    std::vector<double> vertices, tex_vertices, normals;  // xyz / uvw / xyz
    std::vector<uint32_t> triangles;                      // 9 per triangle: (v, vt, vn) x 3
    size_t n_triangles() const { return triangles.size() / 9; }
};

struct LoadError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// src/helpers/mod.rs:352-355
inline Object load_model(const std::string& path) {
    tr_obj* h = nullptr;
    if (tr_obj_load(path.c_str(), &h) != TR_OK) throw LoadError("load_model(" + path + "): " + tr_load_error());
    int64_t np = 0, nt = 0, nn = 0, ntri = 0;
    tr_obj_counts(h, &np, &nt, &nn, &ntri);
    Object o;
    if (np) o.vertices.assign(tr_obj_positions(h), tr_obj_positions(h) + 3 * np);
    if (nt) o.tex_vertices.assign(tr_obj_uvw(h), tr_obj_uvw(h) + 3 * nt);
    if (nn) o.normals.assign(tr_obj_normals(h), tr_obj_normals(h) + 3 * nn);
    if (ntri) o.triangles.assign(tr_obj_indices(h), tr_obj_indices(h) + 9 * ntri);
    tr_obj_free(h);
    return o;
}

// image::RgbaImage / MyRgbaImage: 8-bit RGBA rows
struct RgbaImage {
    int width = 0, height = 0;
    std::vector<uint8_t> data;  // row-major RGBA8
};

// src/helpers/mod.rs:93-96 (`image::open(path).into_rgba8()`): row 0 = top
inline RgbaImage load_rgba_img_from_file(const std::string& path, bool flipped = false) {
    tr_image* h = nullptr;
    if (tr_image_load_rgba8(path.c_str(), flipped ? 1 : 0, &h) != TR_OK)
        throw LoadError("load_rgba_img_from_file(" + path + "): " + tr_load_error());
    RgbaImage img;
    tr_image_size(h, &img.width, &img.height);
    const uint8_t* d = tr_image_data(h);
    img.data.assign(d, d + (size_t)img.width * img.height * 4);
    tr_image_free(h);
    return img;
}

// src/helpers/mod.rs:291-302: row y <- row H-1-y
inline RgbaImage flip_vertically(const RgbaImage& img) {
    RgbaImage o;
    o.width = img.width; o.height = img.height;
    o.data.resize(img.data.size());
    const size_t row = (size_t)img.width * 4;
    for (int y = 0; y < img.height; ++y)
        std::copy(img.data.begin() + (size_t)y * row, img.data.begin() + (size_t)(y + 1) * row,
                  o.data.begin() + (size_t)(img.height - 1 - y) * row);
    return o;
}

// _write_nd_rgba_img_to_file (src/helpers/mod.rs:114-120): PNG of an RGBA8 image, row 0 = top
inline void write_rgba_img_to_file(const std::string& path, const RgbaImage& img) {
    if (tr_image_write_png(path.c_str(), img.data.data(), img.width, img.height) != TR_OK)
        throw LoadError("write_rgba_img_to_file(" + path + "): " + tr_load_error());
}

// src/helpers/mod.rs:336-350: 1 / max(z, 0) where that is positive, else 0
inline std::vector<double> get_drawable_z_buffer(const std::vector<double>& z_buffer) {
    std::vector<double> out(z_buffer.size(), 0.0);
    for (size_t i = 0; i < z_buffer.size(); ++i) {
        const double val = std::fmax(z_buffer[i], 0.0);
        if (val > 0.0) out[i] = 1.0 / val;
    }
    return out;
}

// The one-triangle mesh of src/helpers/mod.rs:558-608 (UVs, no normals)
inline Object get_rando_mesh_object() {
    Object o;
    o.vertices = {0.0, -0.5, 0.0, 0.5, 0.5, 0.0, -0.5, 0.5, 0.0};
    o.tex_vertices = {0.5, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 1.0, 0.0};
    o.triangles = {0, 0, NONE_INDEX, 1, 1, NONE_INDEX, 2, 2, NONE_INDEX};
    return o;
}

}  // namespace tinyrenderer
