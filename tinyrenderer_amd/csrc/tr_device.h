// Device-side data layout and exact-arithmetic helpers for the gfx950 rasterizer.
//
// Numerics contract: every scalar is IEEE binary64, every
// + - * / sqrt is individually rounded, nothing is fused or re-associated.  This file
// is compiled with -ffp-contract=off and without fast-math; f64 divide and sqrt lower
// to correctly rounded sequences on gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tr {

// ---- limits of the 64-bit winner key: tri_seq:28 | half:1 | i:17 | j:18 ----
constexpr uint32_t kResMax = 131070u;          // scanline count limit per half (i <= res)
constexpr uint32_t kTriSeqMax = (1u << 28) - 1; // triangles per frame (all draws since clear)
constexpr uint64_t kKeyNone = 0xFFFFFFFFFFFFFFFFull;
constexpr int kKeyTriShift = 36, kKeyHalfShift = 35, kKeyIShift = 18;
constexpr uint32_t kNanSlots = 8;
constexpr uint32_t kNanEmpty = 0xFFFFFFFFu;
constexpr uint32_t kCounterSlots = 1024;  // 64-byte counter slots, summed on read
constexpr int kShadeBW = 32, kShadeBH = 2;  // pixels per shade workgroup = ONE wave: 128 B of RGBA8 per row; (measured at 4K: 32x8 in four
                                            // waves 46.3 us, 32x2 44.9, 64x1 45.0, 16x4 44.8: a one-wave workgroup fits any SIMD with a free slot)

struct Mat4 {  // column-major, m[c*4+r] (nalgebra storage order)
    double m[16];
};

// Per-triangle record read by the raster passes and by the shade pass to rebuild a sample:
// the y-sorted viewport points S0,S1,S2 and the long-side split point
// (reference: src/model_renderer/mod.rs:320-331,435-441).
struct __attribute__((aligned(128))) TriRaster {
    double s0[3], s1[3], s2[3], sp[3];
    uint32_t res[2];  // resolution of half 0 / half 1 (mod.rs:363-369); 0 = no scanlines
    uint16_t tx0, ty0, tx1, ty1;  // inclusive rect of the tile raster's tiles its samples can touch (tx0 > tx1: none)
    uint32_t flags;               // kTriNonFinite: x/y of a point is NaN/inf, no range pruning; kTriSmall
    uint32_t slot_shift;          // each scanline of this triangle occupies 1 << slot_shift slots of the
                                  // frame's slot sequence (load balance of the global raster, see k_setup)
    uint32_t pad[2];
};
constexpr uint32_t kTriNonFinite = 1u;
constexpr uint32_t kTriSmall = 2u;     // both halves have at most small_res + 1 scanlines (kSmallRes by default): rastered by k_small, owns no slots
constexpr uint32_t kTriFar = 4u;       // its centroid lies behind the draw's split depth (z_split): k_small / k_mid take such
                                       // triangles in a second depth launch, after the nearer ones have lowered the plane
// per-triangle flag byte of the k_small / k_mid sweeps (small_flag): path in the low bits, kFlagFar on top
constexpr uint8_t kFlagSmall = 1, kFlagMid = 2, kFlagPath = 3, kFlagFar = 4;
constexpr uint32_t kSmallRes = 8u;
constexpr int kTile = 16;             // granularity of a band's rows (tr_set_band) and the tile raster's smallest tile edge, in pixels
constexpr double kPruneMargin = 1.0 / 64.0;  // slack (pixels) of every conservative range test
static_assert(sizeof(TriRaster) == 128, "TriRaster must be 128 B");

// Per-triangle record read by the shade pass only (mod.rs:329,333-358,477-535).
struct __attribute__((aligned(256))) TriShade {
    double ux[3], uy[3];  // unsorted viewport x / y of the three corners
    double n[3][3];       // vertex-shader normal xyz per corner (main.rs:164)
    double uv[3][2];      // texture u,v per corner
    double t[3], b[3];    // tangent / bitangent xyz (mod.rs:346-351)
    uint32_t draw;        // index into the per-frame draw table
    uint32_t flags;       // bit0 has_uv (all three corners), bit1 has_tb
    double inv_res[2];    // 1.0 / res[half] of the triangle's TriRaster (mod.rs:376: `delta`): the same division, done once per
                          // triangle at setup instead of once per shaded pixel
};
static_assert(sizeof(TriShade) == 256, "TriShade must be 256 B");

// One draw call's fragment-shader constants (src/main.rs:173-267).
struct DrawDev {
    const uint32_t* tex;  // RGBA8, row-major, row 0 = bottom (already flipped)
    const uint32_t* nm;
    int32_t tw, th, nw, nh;
    double light[3];  // normalize(camera_pos), main.rs:249-251
    double ambient, intensity;
    double vcol[4];  // VertexShaderResult.color (WHITE), main.rs:170
    int32_t use_nm;
    int32_t pad;
};

// Batched views (tr_render_views): what differs from view to view of one mesh -- the two matrices of the vertex
// stage (main.rs:126-135,152-153) and the fragment constants (the light is the camera position, main.rs:249).
struct __attribute__((aligned(128))) ViewDev {
    Mat4 pmv;      // perspective * model_view
    Mat4 nmat;     // (model_view^T)^-1
    DrawDev draw;
    double z_split;  // viewport z of the mesh box's centre under this view, or NaN (kTriFar)
};
static_assert(sizeof(ViewDev) == 384, "ViewDev must be 384 B");

// EXTENSION shaders (no counterpart in the reference, whose only shader is main.rs:146-267): per
// draw, read by k_shade<true> only.  model 0 with no shadow map is the reference's formula.
struct DrawExt {
    int32_t model;          // 0 reference, 1 Gouraud (per-vertex diffuse), 2 Phong (adds a specular term)
    int32_t sw, sh;         // specular map size
    int32_t mw, mh;         // shadow map size
    int32_t pad;
    const uint32_t* spec;   // RGBA8 specular map (channel 0 = exponent) or nullptr
    const double* shadow;   // light-view z plane (min of -z per pixel, -inf = empty), row-major, or nullptr
    double shininess;       // exponent where there is no specular map (or no uv)
    double strength;        // weight of the specular term
    double bias, dim;       // shadow: in shadow iff dist > map + bias; lit terms are multiplied by dim there
    Mat4 to_light;          // this frame's viewport space -> the light frame's viewport space
};

// Small side table for NaN-depth samples (SURVEY.md 7.1 caveat i): a NaN sample wins a
// pixel iff it is the first sample to land on it in draw order.
struct NanTable {
    uint32_t pix[kNanSlots];        // pixel index or kNanEmpty
    uint64_t first_nan[kNanSlots];  // min order key over NaN samples on that pixel
    uint64_t first_any[kNanSlots];  // min order key over non-NaN samples on that pixel
};

// ------------------------------------------------------------ exact helpers
// nalgebra gemv association: out[r] = ((M[r][0]*x + M[r][1]*y) + M[r][2]*z) + M[r][3]*w
__device__ __forceinline__ void mat4_mul_vec4(const Mat4& a, double x, double y, double z, double w,
                                              double out[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double acc = a.m[r] * x;
        acc = a.m[4 + r] * y + acc;
        acc = a.m[8 + r] * z + acc;
        acc = a.m[12 + r] * w + acc;
        out[r] = acc;
    }
}
// nalgebra dot for static 3-vectors: (a+b)+c ; for 4-vectors: (a+c)+(b+d)
__device__ __forceinline__ double dot3(double ax, double ay, double az, double bx, double by,
                                       double bz) {
    double p0 = ax * bx, p1 = ay * by, p2 = az * bz;
    return (p0 + p1) + p2;
}
// (nalgebra's norm folds from 0.0: `0.0 + s` is s bit for bit here, s being a sum of squares and so never -0)
__device__ __forceinline__ void normalize3(double& x, double& y, double& z) {
    const double n = sqrt(dot3(x, y, z, x, y, z));
    x = x / n;
    y = y / n;
    z = z / n;
}
// Vector4 (x,y,z,0).normalize(): norm^2 = (x*x + z*z) + (y*y + 0*0); the w lane stays unused.
__device__ __forceinline__ void normalize4_w0(double& x, double& y, double& z) {
    double p0 = x * x, p1 = y * y, p2 = z * z, p3 = 0.0 * 0.0;
    p0 += p2;
    p1 += p3;
    const double n = sqrt(p0 + p1);
    x = x / n;
    y = y / n;
    z = z / n;
}

// Monotone u64 image of a non-NaN double, with -0 folded onto +0 (the reference's
// `current_z > dist` test cannot tell the zeros apart).
__device__ __forceinline__ uint64_t encode_depth(double d) {
    const uint64_t b = (uint64_t)__double_as_longlong(d + 0.0);
    // negative: every bit flipped, else the sign bit set (sign mask by an arithmetic shift: 4 instructions, no compare)
    return b ^ ((uint64_t)((int64_t)b >> 63) | 0x8000000000000000ull);
}

// Index of pixel (ix, iy) in the depth and winner-key planes.  Row-major by default: a scanline's samples are
// neighbours in memory.  `tiled` (frames of small-triangle meshes, W a multiple of 4 and H of 2): 4 x 2 pixels to a
// 64-byte segment, the unit of a memory-side atomic request -- the 32 triangles of a k_small wave cover a patch of a
// few pixels square, every row of which is a segment of its own in the row-major plane.
__device__ __forceinline__ uint32_t plane_index(uint32_t ix, uint32_t iy, uint32_t W, bool tiled) {
    return tiled ? ((((iy >> 1) * (W >> 2) + (ix >> 2)) << 3) | ((iy & 1u) << 2) | (ix & 3u)) : iy * W + ix;
}

// Rust `v.ceil() as i64` followed by `.max(2)` for v >= 0 or NaN; values above `limit`
// are reported as limit+1 so the caller can flag them.
__device__ __forceinline__ uint32_t res_from_extent(double m, uint32_t limit) {
    double c = ceil(m);
    if (!(c > 2.0)) return 2u;  // NaN, <= 2
    if (c > (double)limit) return limit + 1u;
    return (uint32_t)c;
}

// Pixel coordinate of `round(v) as i64` tested against [0, n-1]: returns -1 if outside.
// NaN rounds to 0 (Rust saturating cast), -0.0 is inside.
// round() (half away from zero) is evaluated as trunc(v + copysign(0.5 - 2^-54, v)): with the
// largest double below 0.5 the sum never reaches the next integer unless |frac(v)| >= 0.5, and a
// tie of the sum itself only occurs at |frac(v)| = 0.5, where it rounds to the even neighbour,
// which is the integer.  tests/test_round_trick.py checks it against round() on every
// half-integer neighbourhood up to 2^22 and 4.5e8 random values; the conversion to int truncates.
__device__ __forceinline__ int pixel_coord(double v, int n) {
    const double x = v + copysign(0.49999999999999994, v);
    if (x > -1.0 && x < (double)n) return (int)x;  // trunc(x) in [0, n-1]
    return x != x ? 0 : -1;
}
// The same coordinate for the sample loops, which test it against the frame themselves: the pixel is inside
// [0, n-1] exactly when (uint32_t)pixel_coord_raw(v) < n, and is then pixel_coord(v, n).  v_cvt_i32_f64 truncates,
// saturates at the ends of int32 (below -1: negative, i.e. huge as unsigned; n and above: >= n) and turns NaN
// into 0, which is the rule above without its three comparisons and two selects.  Checked against round() by the
// device selftest next to pixel_coord.
__device__ __forceinline__ int pixel_coord_raw(double v) {
    const double x = v + copysign(0.49999999999999994, v);
    int r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// Inclusive prefix sum of `v` over the 64 lanes of a wave in the VALU's data-parallel-primitive path: shifts within rows of 16
// lanes by 1, 2, 4 and 8, then the last lane of rows 0 / 2 broadcast into rows 1 / 3 and the last lane of the lower half into the
// upper half -- six adds, no LDS round trip (six `ds_bpermute` and their waits otherwise).
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}

// Minimum / maximum over each row of 16 lanes, left in every lane of the row (rotations within the row by 1, 2, 4 and 8).
__device__ __forceinline__ int row16_min(int x) {
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x121, 0xf, 0xf, false));  // row_ror:1
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x122, 0xf, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x124, 0xf, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false));
    return x;
}
__device__ __forceinline__ int row16_max(int x) {
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x121, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x122, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x124, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false));
    return x;
}

// The sample enumeration of one scanline end-point pair (segment_3d/mod.rs:26-33,41-48):
// given the half's apex a1 and the two far ends a2 (short edge) and b2 (split point),
// scanline i of `res` has A = a1 + t*(a2-a1), B = a1 + t*(b2-a1), t = i * (1/res).
struct Scanline {
    double ax, ay, az, dx, dy, dz;  // A and B-A
    uint32_t hres;                  // horizontal resolution (>= 2), or kBig+1 if too long
};
__device__ __forceinline__ Scanline make_scanline_pts(const double* a1, const double* a2, const double* b2, uint32_t i, double delta) {
    double t = (double)i * delta;
    double Ax = a1[0] + t * (a2[0] - a1[0]);
    double Ay = a1[1] + t * (a2[1] - a1[1]);
    double Az = a1[2] + t * (a2[2] - a1[2]);
    double Bx = a1[0] + t * (b2[0] - a1[0]);
    double By = a1[1] + t * (b2[1] - a1[1]);
    double Bz = a1[2] + t * (b2[2] - a1[2]);
    Scanline s;
    s.ax = Ax; s.ay = Ay; s.az = Az;
    s.dx = Bx - Ax; s.dy = By - Ay; s.dz = Bz - Az;
    // (ceil(max(|dx|,|dy|)) as i64 + 1).max(2)
    double c = ceil(fmax(fabs(s.dx), fabs(s.dy)));
    uint32_t h;
    if (c != c) h = 1u;                        // NaN as i64 = 0, +1
    else if (c > 262142.0) h = 262144u;        // beyond the key's j field; setup prevents this
    else h = (uint32_t)c + 1u;
    s.hres = h < 2u ? 2u : h;
    return s;
}
__device__ __forceinline__ Scanline make_scanline(const TriRaster& r, uint32_t half, uint32_t i, double delta) {
    return make_scanline_pts(half ? r.s2 : r.s0, r.s1, r.sp, i, delta);
}
__device__ __forceinline__ Scanline make_scanline(const TriRaster& r, uint32_t half, uint32_t i) {
    return make_scanline(r, half, i, 1.0 / (double)r.res[half]);
}


// One slot of the frame's scanline sequence as the raster passes consume it (global raster):
// A, B - A, 1/hres and the order key of sample 0.  keyh = (tri:28 | half:1 | i:17) << 18 | hres;
// hres == 0 marks an empty slot (padding, or a scanline too long for the key).
struct __attribute__((aligned(64))) ScanRec {
    double ax, ay, az, dx, dy, dz, inv;
    unsigned long long keyh;
};
static_assert(sizeof(ScanRec) == 64, "ScanRec must be 64 B");
constexpr unsigned long long kRecHresMask = 0x3FFFFull;


// Rust `v as u8` (and `v as usize`, as far as 32 bits): truncation toward zero, NaN and negatives to 0, saturation
// above -- which is what the hardware conversion does by itself (the C++ cast is undefined out of range, so the
// instruction is named).  tr_selftest_arithmetic compares it with the spelled-out rule.
__device__ __forceinline__ uint32_t f64_as_u32_sat(double v) {
    uint32_t r;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t f64_as_u8(double v) {
    const uint32_t r = f64_as_u32_sat(v);
    return r < 255u ? r : 255u;
}
__device__ __forceinline__ uint32_t f64_as_u8_spelled_out(double v) {
    if (!(v > 0.0)) return 0u;  // NaN, negative, zero
    if (v >= 255.0) return 255u;
    return (uint32_t)v;
}

// x / 255.0, correctly rounded, in 3 operations instead of the ~11 of a general f64 division.
// With r = RN(1/255): q = RN(x r), e = x - 255 q (exact in an fma: a multiple of ulp(q) below
// 2^9 ulp(q)), result = RN(q + e r) = RN(v + (v - q) eps) with v = x/255, |eps| <= 2^-53, i.e. v
// perturbed by less than 2^-104 |v|.  v = X 2^k / 255 with X < 2^53 is never closer than
// 2^-62 |v| to a rounding boundary (255 (2M+1) 2^(j-1) = X 2^k has an odd left and an even right
// side), so the perturbation cannot change the rounding.  Outside the normal range, for zeros
// (whose sign the fmas lose) and for non-finite x the plain division is used.  tests/test_round_trick.py checks it on the host.
__device__ __forceinline__ double div255(double x) {
    const double ax = fabs(x);
    if (__builtin_expect(!(ax < 1e150) || ax < 1e-150, 0)) return x / 255.0;
    constexpr double r = 1.0 / 255.0;
    const double q = x * r;
    const double e = __builtin_fma(-255.0, q, x);
    return __builtin_fma(e, r, q);
}

}  // namespace tr
