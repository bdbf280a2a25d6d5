// gfx950 kernels of the rasterizer.  Stage map (BASELINE.json north_star):
//   (1) k_transform      vertex transform into SoA            main.rs:146-172
//   (2) k_setup + k_scan triangle setup, cull, scanline count + prefix
//                                                              model_renderer/mod.rs:287-369,435-452
//   (3) k_raster<0/1>    sample enumeration, depth min, winner min
//                                                              mod.rs:360-390, segment_3d/mod.rs:19-115
//   (4) k_shade          deferred shade of the winning sample  mod.rs:477-535, main.rs:173-267
// Compiled with -ffp-contract=off: no FMA contraction anywhere.
#include "tr_device.h"

namespace tr {

// ============================================================ (1) transform
// One thread per unique position (ndc = (PMV*pos).xyz / w) and per unique normal
// (N = Nm * (n,0)).  The reference runs the vertex shader 3x per triangle with no sharing
// (mod.rs:272-276); the result is a pure function of the indexed attribute, so it is
// computed once per attribute here.  Output is SoA: ndc[0..n) x, [n..2n) y, [2n..3n) z.
__global__ __launch_bounds__(256) void k_transform(const double* __restrict__ pos, int n_pos,
                                                   const double* __restrict__ nrm, int n_nrm,
                                                   Mat4 pmv, Mat4 nmat, double* __restrict__ ndc,
                                                   double* __restrict__ nout) {
    // (n_pos, n_nrm <= 2^28: the thread id fits 32 bits, the element offsets are formed in 64)
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t np = (size_t)n_pos, nn = (size_t)n_nrm;
    if (id < np) {
        double g[4];
        mat4_mul_vec4(pmv, pos[3 * id], pos[3 * id + 1], pos[3 * id + 2], 1.0, g);
        ndc[id] = g[0] / g[3];
        ndc[np + id] = g[1] / g[3];
        ndc[2 * np + id] = g[2] / g[3];
    } else if (id < np + nn) {
        const size_t k = id - np;
        double g[4];
        mat4_mul_vec4(nmat, nrm[3 * k], nrm[3 * k + 1], nrm[3 * k + 2], 0.0, g);
        nout[k] = g[0];
        nout[nn + k] = g[1];
        nout[2 * nn + k] = g[2];
    }
}

// ================================================================ (2) setup
constexpr int kMaxBatchMeshes = 4;  // meshes per batch of views (tr_render_views_multi)
struct SetupArgs {
    const uint32_t* idx;   // 9 per triangle
    const double* pos;     // raw positions (face-normal fallback)
    const double* uvw;     // raw texture vertices
    const double* ndc;     // SoA from k_transform
    const double* nout;    // SoA from k_transform
    int n_pos, n_nrm, n_tris;
    uint32_t tri_base;     // first tri_seq of this draw
    uint32_t draw;         // draw index in the frame
    int W, H;
    Mat4 vpm;              // viewport matrix, mod.rs:314-318 (built on the host per draw)
    Mat4 nmat;             // for the face-normal fallback
    TriRaster* rast;
    TriShade* shade;
    uint32_t* half_cnt;    // slots per half, 2 per triangle (global-atomic raster only)
    uint32_t* tile_cnt;    // triangles per tile (tile raster only)
    int tiles_x, tiles_y;
    int tw_shift, th_shift; // log2 of the tile raster's tile width / height in pixels
    int y_lo, y_hi;        // band of pixel rows owned by this context
    int mode;              // bit0: fill half_cnt, bit1: count tiles, bit2: flag small triangles (k_small), bit4: flag mid ones (k_mid),
                           // bit3: no TriShade now (k_build_shade writes it for the triangles the winner pass flags)
    float slot_len;        // load balance of the slot sequence: a scanline gets slots until its mean length per slot is below this (k_setup)
    uint32_t mid_samples;  // mode bit 4: a triangle of at most 64 scanlines per half and about this many samples per half goes to k_mid
    uint32_t small_res;    // a triangle is small if neither half has more than small_res + 1 scanlines
    uint8_t* small_flag;   // per triangle: 1 = flagged kTriSmall (what k_small reads)
    uint8_t* small_grp;    // per 32 triangles: raised if one of them is small (k_small skips the other groups)
    uint8_t* need_shade;   // per triangle, draws with deferred shade records (mode bit 3): zeroed here, set by the winner pass
    uint32_t* flags;
    DrawDev params;
    DrawDev* draw_table;
    // fused front end (small draws): the vertex stage is evaluated per triangle corner instead of
    // read back, and the workgroup that finishes last scans the draw's slot counts
    const double* nrm;     // raw normals
    Mat4 pmv;              // perspective * model_view
    uint32_t* ticket;      // 256 words of 8 bytes: {scan_tag, slot total} of each workgroup of the fused launch
    uint32_t scan_tag;     // differs from launch to launch (never 0)
    uint32_t scan_expect;  // the tag the pollers wait for: scan_tag (TR_DEBUG_PREFIX_STALL: another one, so that the test sees the bail-out)
    uint32_t spin_limit;   // polls of one word before a workgroup gives up (2^24)
    uint32_t* half_off;    // prefix to write: half_off[scan_lo + 1 .. scan_lo + n_halves]
    uint32_t scan_lo, n_halves;
    // batched views (k_setup<false, true>): thread g = view * n_tris + k sets up triangle k of the mesh under
    // views[view]; its tri_seq is tri_base + g and its draw index draw + view
    const ViewDev* views;
    uint32_t n_views;
    int per_corner;        // k_setup<false, false>: evaluate the vertex stage per corner instead of reading ndc / nout
    const double* bounds;  // per 64 consecutive triangles of the mesh: min xyz, max xyz of their positions (k_mesh_bounds),
                           // or nullptr; a context that owns a band drops whole clusters that project outside its rows
    double z_split;        // viewport z of the mesh box's centre, or NaN: triangles whose centroid is farther get kTriFar
    uint8_t* cull;         // per cluster of this draw: 1 = misses the band (written by k_cull_clusters), or nullptr
    // batched views of SEVERAL meshes (tr_render_views_multi: the reference's loop draws its head and the eye parented to
    // it every iteration, main.rs:272-273): a view's triangles are those of mesh 0, then mesh 1, ... (n_tris = their sum);
    // triangle kv of a view belongs to the mesh m with mesh_off[m] <= kv < mesh_off[m + 1]; views[view * n_meshes + m]
    // holds that draw's matrices and constants, its draw-table entry is draw + view * n_meshes + m
    uint32_t n_meshes;
    uint32_t mesh_off[kMaxBatchMeshes + 1];
    const uint32_t* m_idx[kMaxBatchMeshes];
    const double* m_pos[kMaxBatchMeshes];
    const double* m_nrm[kMaxBatchMeshes];
    const double* m_uvw[kMaxBatchMeshes];
};

__device__ __forceinline__ void scan_in_block(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* off,
                                              uint32_t* flags);

__device__ __forceinline__ void setup_triangle(const SetupArgs& a, int k, bool fused, uint32_t& cnt0, uint32_t& cnt1,
                                               const Mat4& pmv, const Mat4& nmat, uint32_t draw, uint32_t seq, double z_split);

// FUSED = false: vertex stage read from k_transform's arrays, prefix by the k_scan_* kernels.
// FUSED = true (draws of up to 256 workgroups of 64 to 256 threads): one launch does transform + setup + scan.
// VIEWS (with FUSED = false): one launch sets up every (view, triangle) of a batch of views of one mesh; the vertex
// stage is evaluated per corner under the view's matrices (as in the fused launch), prefix by the k_scan_* kernels.
// Band contexts (multi-GPU), draws whose mesh has cluster boxes (k_mesh_bounds): eight lanes take the corners of a
// cluster's box through the vertex stage and the viewport; with w > 0 at all eight the projection of everything
// inside the box lies between the corners' extremes, so a box that misses the band's rows by more than a pixel of
// margin holds no triangle that could land a sample there: cull[cluster] = 1.  (One test per cluster here instead
// of one per wave of k_setup, where 56 lanes idled through it: 110 -> 40 us for a band of a 10^7-triangle mesh.)
__global__ __launch_bounds__(256) void k_cull_clusters(SetupArgs a, uint32_t n_clusters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cl = t >> 3, corner = t & 7u;
    double y = 0.0;
    bool ok = true;
    if (cl < n_clusters) {
        const double* b = a.bounds + 6 * (size_t)cl;
        double g[4], v[4];
        mat4_mul_vec4(a.pmv, b[(corner & 1u) ? 3 : 0], b[(corner & 2u) ? 4 : 1], b[(corner & 4u) ? 5 : 2], 1.0, g);
        mat4_mul_vec4(a.vpm, g[0] / g[3], g[1] / g[3], g[2] / g[3], 1.0, v);
        y = v[1] / v[3];
        ok = g[3] > 1e-12 && fabs(y) < 1e15;  // (NaN fails both)
    }
    double ymin = y, ymax = y;
    int all_ok = ok ? 1 : 0;
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        ymin = fmin(ymin, __shfl_xor(ymin, d, 64));
        ymax = fmax(ymax, __shfl_xor(ymax, d, 64));
        all_ok &= __shfl_xor(all_ok, d, 64);
    }
    const bool miss = ceil(ymin - 1.5) > (double)(a.y_hi - 1) || floor(ymax + 1.5) < (double)a.y_lo;
    const bool culled = all_ok && miss;  // (the same in the cluster's eight lanes)
    if (corner == 0u && cl < n_clusters) a.cull[cl] = culled ? 1 : 0;
    // k_setup writes nothing for a cluster dropped here; its small / mid flags, which the raster kernels sweep by
    // frame position, are cleared now (8 of the 64 bytes per lane)
    if (culled && cl < n_clusters && a.small_flag) {
        uint8_t* f = a.small_flag + a.tri_base + (size_t)cl * 64u + corner * 8u;
        const uint32_t left = (uint32_t)a.n_tris - (cl * 64u + corner * 8u);  // (may exceed 8, or be "negative")
        if (cl * 64u + corner * 8u + 8u <= (uint32_t)a.n_tris && ((uintptr_t)f & 7u) == 0) {
            *reinterpret_cast<uint2*>(f) = make_uint2(0u, 0u);
        } else if (cl * 64u + corner * 8u < (uint32_t)a.n_tris) {
            for (uint32_t i = 0; i < 8u && i < left; ++i) f[i] = 0;
        }
    }
}

template <bool FUSED, bool VIEWS = false>
__global__ __launch_bounds__(256) void k_setup(SetupArgs a) {
    if (VIEWS) {
        const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
        if (g >= a.n_views * (uint32_t)a.n_tris) return;
        const uint32_t view = g / (uint32_t)a.n_tris;
        const uint32_t kv = g - view * (uint32_t)a.n_tris;
        uint32_t m = 0;
#pragma unroll
        for (uint32_t q = 1; q < (uint32_t)kMaxBatchMeshes; ++q)
            if (q < a.n_meshes && kv >= a.mesh_off[q]) m = q;
        const int k = (int)(kv - a.mesh_off[m]);
        const uint32_t d = view * a.n_meshes + m;
        const ViewDev& vd = a.views[d];
        if (k == 0) a.draw_table[a.draw + d] = vd.draw;
        uint32_t c0 = 0, c1 = 0;
        if (a.n_meshes > 1u) {  // (the triangle's own mesh: indices, positions, normals, texture vertices)
            SetupArgs b = a;
            b.idx = a.m_idx[m]; b.pos = a.m_pos[m]; b.nrm = a.m_nrm[m]; b.uvw = a.m_uvw[m];
            setup_triangle(b, k, true, c0, c1, vd.pmv, vd.nmat, a.draw + d, a.tri_base + g, vd.z_split);
        } else {
            setup_triangle(a, k, true, c0, c1, vd.pmv, vd.nmat, a.draw + d, a.tri_base + g, vd.z_split);
        }
        a.half_cnt[2 * (size_t)(a.tri_base + g)] = c0;  // (setup_triangle leaves the counts to its caller when `fused`)
        a.half_cnt[2 * (size_t)(a.tri_base + g) + 1] = c1;
        // (the triangle's view, for k_mid: a scalar load there instead of a 32-bit division per half; after setup_triangle's
        // own store of the record, by the same thread)
        a.rast[a.tri_base + g].pad[0] = view;
        return;
    }
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) a.draw_table[a.draw] = a.params;
    uint32_t c0 = 0, c1 = 0;  // slots of my triangle's two halves
    // Band contexts (multi-GPU): the wave's 64 triangles are one cluster of the mesh, and k_cull_clusters found that its
    // box misses the band's rows: nothing of theirs is read, nothing written (the scan and k_build_shade consult the
    // same byte, the small / mid flags were cleared by k_cull_clusters) -- 7/8 of a 10^7-triangle mesh on an 8-GPU node.
    // (A fixed grid striding over the clusters kept was no faster: the hoisted matrices cost it a wave per SIMD.)
    if (!FUSED && a.cull && a.cull[(uint32_t)k >> 6]) return;
    if (k < a.n_tris) {
        // (per_corner: a draw whose shade records are deferred needs no normals here and its positions once per
        // corner: the vertex stage is evaluated in place, there is no k_transform launch and no ndc array)
        setup_triangle(a, k, FUSED || a.per_corner, c0, c1, a.pmv, a.nmat, a.draw, a.tri_base + (uint32_t)k, a.z_split);
        if (!FUSED && a.per_corner && (a.mode & 1)) {  // (setup_triangle leaves the counts to its caller when `fused`)
            a.half_cnt[2 * (size_t)(a.tri_base + (uint32_t)k)] = c0;
            a.half_cnt[2 * (size_t)(a.tri_base + (uint32_t)k) + 1] = c1;
        }
    }
    if (!FUSED) return;
    // ---- prefix of the draw's slot counts without a second pass over them: every workgroup scans its
    // 512 counts in registers, publishes its total as ONE 8-byte word {launch tag, total} (write-through,
    // `sc1`), and wave 0 adds up the words of the workgroups before it as they appear (`sc1` polls; the
    // tag is in the same granule as the value, so a word that shows the tag shows the total:
    // MI355X_MICROARCH.md, data-tagged granules).  No workgroup waits for a later one, and at most 255
    // small workgroups wait at all, far fewer than the GPU holds: every poll ends.  (The first
    // version let the last workgroup to arrive scan all counts from memory: 7-10 us behind the slowest
    // workgroup's 11 us; now the prefix is out about 2 us after a workgroup's own triangles.)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ uint32_t s_wtot[4];
    __shared__ unsigned long long s_base;
    const uint32_t pair = c0 + c1;  // (< 2^25)
    const uint32_t inc = wave_inclusive_sum(pair);  // (< 2^31)
    if (lane == 63) s_wtot[wave] = inc;
    __syncthreads();
    unsigned long long before = 0, tot = 0;
    const int n_waves = (int)(blockDim.x >> 6);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned long long t = w < n_waves ? s_wtot[w] : 0u;
        if (w < wave) before += t;
        tot += t;
    }
    uint32_t* off = a.half_off + a.scan_lo;
    unsigned long long* agg = reinterpret_cast<unsigned long long*>(a.ticket);
    if (threadIdx.x == 0) {
        const unsigned long long word = ((unsigned long long)a.scan_tag << 32) | (tot > 0xFFFFFFFFull ? 0xFFFFFFFFull : tot);
        __hip_atomic_store(&agg[blockIdx.x], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tot > 0xFFFFFFFFull) atomicOr(a.flags, 1u << 4);
    }
    if (wave == 0) {
        unsigned long long sum = lane == 0 ? (unsigned long long)off[0] : 0ull;  // (written before this launch)
        for (uint32_t j = (uint32_t)lane; j < blockIdx.x; j += 64u) {
            unsigned long long word = 0;
            uint32_t tries = 0;
            for (;;) {
                word = __hip_atomic_load(&agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(word >> 32) == a.scan_expect) break;
                if (++tries > a.spin_limit) {  // (2^24 tries = seconds: something is badly wrong; leave with TR_FLAG_PREFIX_STALL rather than hang)
                    atomicOr(a.flags, 1u << 7);
                    word = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            sum += word & 0xFFFFFFFFull;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
        if (lane == 0) s_base = sum;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    if (k < a.n_tris) {
        const uint32_t o = (uint32_t)(base + before) + (inc - pair);
        off[2 * (size_t)k + 1] = o + c0;  // off[j+1] = inclusive sum through j
        off[2 * (size_t)k + 2] = o + c0 + c1;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && base + tot > 0xFFFFFFFFull) atomicOr(a.flags, 1u << 4);
}

// The per-triangle record of the shade pass (mod.rs:329,333-358 and the vertex-shader outputs of main.rs:164-171):
// unsorted viewport x / y, the three normals, uvs, tangent and bitangent.  px/py/pz: clip_space_position after the
// divide; vx/vy: viewport x / y.
__device__ __forceinline__ TriShade make_shade_record(const SetupArgs& a, const uint32_t iv[3], const uint32_t it[3],
                                                      const uint32_t in[3], const double px[3], const double py[3],
                                                      const double pz[3], const double vx[3], const double vy[3],
                                                      bool fused, const Mat4& nmat, uint32_t draw, uint32_t res0, uint32_t res1) {
    TriShade S;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        S.ux[c] = vx[c];
        S.uy[c] = vy[c];
    }
    // normals: indexed (mod.rs:253-258) or the face normal (mod.rs:245-252)
    bool need_face = in[0] == 0xFFFFFFFFu || in[1] == 0xFFFFFFFFu || in[2] == 0xFFFFFFFFu;
    double fn[3] = {0, 0, 0};
    if (need_face) {
        double v[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            for (int d = 0; d < 3; ++d) v[c][d] = a.pos[3 * (size_t)iv[c] + d];
        double ax = v[1][0] - v[0][0], ay = v[1][1] - v[0][1], az = v[1][2] - v[0][2];
        normalize3(ax, ay, az);
        double bx = v[2][0] - v[1][0], by = v[2][1] - v[1][1], bz = v[2][2] - v[1][2];
        normalize3(bx, by, bz);
        double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
        normalize3(cx, cy, cz);
        double g[4];
        mat4_mul_vec4(nmat, cx, cy, cz, 0.0, g);
        fn[0] = g[0]; fn[1] = g[1]; fn[2] = g[2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (in[c] != 0xFFFFFFFFu && fused) {
            double g[4];
            mat4_mul_vec4(nmat, a.nrm[3 * (size_t)in[c]], a.nrm[3 * (size_t)in[c] + 1], a.nrm[3 * (size_t)in[c] + 2], 0.0, g);
            S.n[c][0] = g[0]; S.n[c][1] = g[1]; S.n[c][2] = g[2];
        } else if (in[c] != 0xFFFFFFFFu) {
            S.n[c][0] = a.nout[in[c]];
            S.n[c][1] = a.nout[(size_t)a.n_nrm + in[c]];
            S.n[c][2] = a.nout[2 * (size_t)a.n_nrm + in[c]];
        } else {
            S.n[c][0] = fn[0]; S.n[c][1] = fn[1]; S.n[c][2] = fn[2];
        }
    }
    uint32_t fl = 0;
    bool has_uv = it[0] != 0xFFFFFFFFu && it[1] != 0xFFFFFFFFu && it[2] != 0xFFFFFFFFu;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        S.uv[c][0] = has_uv ? a.uvw[3 * (size_t)it[c]] : 0.0;
        S.uv[c][1] = has_uv ? a.uvw[3 * (size_t)it[c] + 1] : 0.0;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { S.t[d] = 0.0; S.b[d] = 0.0; }
    if (has_uv) {
        fl |= 1u;
        // mod.rs:339-351; 2x2 try_inverse and the (2x2)*(2x3) product as nalgebra does them.
        double m11 = S.uv[1][0] - S.uv[0][0], m12 = S.uv[1][1] - S.uv[0][1];
        double m21 = S.uv[2][0] - S.uv[0][0], m22 = S.uv[2][1] - S.uv[0][1];
        double e0[3] = {px[1] - px[0], py[1] - py[0], pz[1] - pz[0]};
        double e1[3] = {px[2] - px[0], py[2] - py[0], pz[2] - pz[0]};
        double det = m11 * m22 - m21 * m12;
        if (det != 0.0) {
            double i11 = m22 / det, i12 = -m12 / det, i21 = -m21 / det, i22 = m11 / det;
            double t[3], b[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                t[d] = i12 * e1[d] + i11 * e0[d];
                b[d] = i22 * e1[d] + i21 * e0[d];
            }
            normalize4_w0(t[0], t[1], t[2]);
            normalize4_w0(b[0], b[1], b[2]);
#pragma unroll
            for (int d = 0; d < 3; ++d) { S.t[d] = t[d]; S.b[d] = b[d]; }
            fl |= 2u;
        }
    }
    S.draw = draw;
    S.flags = fl;
    S.inv_res[0] = 1.0 / (double)res0;  // (res 0: a half without scanlines, never read)
    S.inv_res[1] = 1.0 / (double)res1;
    return S;
}

__device__ __forceinline__ void setup_triangle(const SetupArgs& a, int k, bool fused, uint32_t& cnt0, uint32_t& cnt1,
                                               const Mat4& pmv, const Mat4& nmat, uint32_t draw, uint32_t seq, double z_split) {
    const uint32_t* ix = a.idx + 9 * (size_t)k;
    uint32_t iv[3] = {ix[0], ix[3], ix[6]};
    uint32_t it[3] = {ix[1], ix[4], ix[7]};
    uint32_t in[3] = {ix[2], ix[5], ix[8]};
    uint32_t res0 = 0, res1 = 0;

    double px[3], py[3], pz[3];  // clip_space_position (after divide), main.rs:154-162
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (fused) {  // k_transform's arithmetic, per corner
            double g[4];
            mat4_mul_vec4(pmv, a.pos[3 * (size_t)iv[c]], a.pos[3 * (size_t)iv[c] + 1], a.pos[3 * (size_t)iv[c] + 2], 1.0, g);
            px[c] = g[0] / g[3];
            py[c] = g[1] / g[3];
            pz[c] = g[2] / g[3];
        } else {
            px[c] = a.ndc[iv[c]];
            py[c] = a.ndc[(size_t)a.n_pos + iv[c]];
            pz[c] = a.ndc[2 * (size_t)a.n_pos + iv[c]];
        }
    }
    // mod.rs:293-312: cull iff all three corners are outside on their own.
    bool cull = true;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        bool out = px[c] < -1.0 || px[c] > 1.0 || py[c] < -1.0 || py[c] > 1.0 || pz[c] < -1.0 ||
                   pz[c] > 1.0;
        cull = cull && out;
    }
    TriRaster R;
    bool small = false, mid = false, far = false;
    R.res[0] = R.res[1] = 0;
    R.tx0 = 1; R.tx1 = 0; R.ty0 = 1; R.ty1 = 0; R.flags = 0; R.slot_shift = 0;
    R.pad[0] = R.pad[1] = 0;
    if (cull && (a.mode & 2)) a.rast[seq] = R;  // (only the tile raster's binning pass reads the record of every triangle)
    if (!cull) {
        // mod.rs:320-327 viewport transform; w of the product is 1 unless inputs are non-finite.
        // (the viewport matrix's last row is 0 0 0 1, so w is exactly 1 unless a coordinate is not finite -- 0 * inf --, and
        // x / 1.0 is x bit for bit: the nine divisions are done only by a wave that holds such a corner)
        double vx[3], vy[3], vz[3], vw[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double g[4];
            mat4_mul_vec4(a.vpm, px[c], py[c], pz[c], 1.0, g);
            vx[c] = g[0]; vy[c] = g[1]; vz[c] = g[2]; vw[c] = g[3];
        }
        if (!__all(vw[0] == 1.0 && vw[1] == 1.0 && vw[2] == 1.0)) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                vx[c] = vx[c] / vw[c];
                vy[c] = vy[c] / vw[c];
                vz[c] = vz[c] / vw[c];
            }
        }
        // A context that owns a band of rows (multi-GPU, tr_set_band): a triangle none of whose samples
        // can land in the band (every sample lies in the hull of the corners up to a few ulp and lands
        // in row round(y)) is dropped here, before any record is built or written.
        bool out_of_band = false;
        if (a.y_lo > 0 || a.y_hi < a.H) {
            const double ymin = fmin(fmin(vy[0], vy[1]), vy[2]), ymax = fmax(fmax(vy[0], vy[1]), vy[2]);
            const bool finite = fabs(vy[0]) < __builtin_huge_val() && fabs(vy[1]) < __builtin_huge_val() && fabs(vy[2]) < __builtin_huge_val();
            // (a triangle whose three corners share one y -- mod.rs:435-441: the split parameter is 0 / 0 -- has NaN in every sample's
            // coordinates and depth, and `as` casts send them all to pixel (0, 0): it belongs to the band that owns row 0, wherever
            // its corners are, and to no other)
            out_of_band = finite && (ymin == ymax ? a.y_lo > 0
                                                  : (ceil(ymin - kPruneMargin - 0.5) > (double)(a.y_hi - 1) || floor(ymax + kPruneMargin + 0.5) < (double)a.y_lo));
        }
        if (out_of_band) {
            if (a.mode & 2) a.rast[seq] = R;
        } else {
        // mod.rs:331 stable insertion sort by y descending, NaN compares Equal.
        int o[3] = {0, 1, 2};
        auto is_less = [&](int p, int q) { return vy[q] < vy[p]; };
#pragma unroll
        for (int i = 1; i >= 0; --i) {
            if (is_less(o[i + 1], o[i])) {
                int tmp = o[i];
                o[i] = o[i + 1];
                int dest = i + 1;
                for (int j = i + 2; j < 3; ++j) {
                    if (!is_less(o[j], tmp)) break;
                    o[j - 1] = o[j];
                    dest = j;
                }
                o[dest] = tmp;
            }
        }
        double s[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s[c][0] = vx[o[c]];
            s[c][1] = vy[o[c]];
            s[c][2] = vz[o[c]];
        }
        // mod.rs:435-441 long-side split point
        double tt = (s[1][1] - s[0][1]) / (s[2][1] - s[0][1]);
        if (tt < 0.0 || tt > 1.0) {
            atomicOr(a.flags, 1u << 1);  // the reference panics on the unwrap
        } else {
            double sp[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) sp[d] = s[0][d] + tt * (s[2][d] - s[0][d]);
            // mod.rs:363-369 for both halves: a = (apex -> S1), b = (apex -> split)
            double m0 = fmax(fmax(fmax(fabs(s[1][0] - s[0][0]), fabs(s[1][1] - s[0][1])),
                                  fabs(sp[0] - s[0][0])),
                             fabs(sp[1] - s[0][1]));
            double m1 = fmax(fmax(fmax(fabs(s[1][0] - s[2][0]), fabs(s[1][1] - s[2][1])),
                                  fabs(sp[0] - s[2][0])),
                             fabs(sp[1] - s[2][1]));
            uint32_t r0 = res_from_extent(m0, kResMax);
            uint32_t r1 = res_from_extent(m1, kResMax);
            if (r0 > kResMax || r1 > kResMax) {
                atomicOr(a.flags, 1u << 0);
            } else {
                res0 = r0;
                res1 = r1;
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                R.s0[d] = s[0][d];
                R.s1[d] = s[1][d];
                R.s2[d] = s[2][d];
                R.sp[d] = sp[d];
            }
        }
        // ---- shade record: now, or (mode bit 3) after the winner pass and only for triangles that may own a pixel
        if (!(a.mode & 8)) a.shade[seq] = make_shade_record(a, iv, it, in, px, py, pz, vx, vy, fused, nmat, draw, res0, res1);
        R.res[0] = res0;
        R.res[1] = res1;
        // Load balance of the global raster: a chunk is 64 slots.  A triangle whose scanlines are
        // long spreads them over 1 << shift slots each (the others stay empty), so that a chunk
        // holds about the same number of samples whatever the triangle size.  The longest
        // scanline of either half is S1 -> split; the mean one has about half its samples.
        if (res0) {
            double ext = fmax(fabs(R.sp[0] - R.s1[0]), fabs(R.sp[1] - R.s1[1]));
            double mean_len = ext * 0.5 + 2.0;
            uint32_t sh = 0;
            const double slot_len = (double)a.slot_len;
            while (sh < 6u && mean_len >= slot_len) { mean_len *= 0.5; ++sh; }  // NaN: stays 0
            R.slot_shift = sh;
        }
        // ---- conservative rect of tiles the triangle's samples can land in: every sample lies
        // in the hull of S0,S1,S2,split up to a few ulp; a sample at x lands in column round(x).
        R.tx0 = 1; R.tx1 = 0; R.ty0 = 1; R.ty1 = 0; R.flags = 0;
        if (res0) {
            double xs[4] = {R.s0[0], R.s1[0], R.s2[0], R.sp[0]};
            double ys[4] = {R.s0[1], R.s1[1], R.s2[1], R.sp[1]};
            bool finite = true;
            double xmin = xs[0], xmax = xs[0], ymin = ys[0], ymax = ys[0];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                finite = finite && fabs(xs[q]) < __builtin_huge_val() && fabs(ys[q]) < __builtin_huge_val();
                xmin = fmin(xmin, xs[q]); xmax = fmax(xmax, xs[q]);
                ymin = fmin(ymin, ys[q]); ymax = fmax(ymax, ys[q]);
            }
            if (!finite) {
                R.flags = kTriNonFinite;  // samples with a NaN coordinate land in column 0 / row 0
            } else {
                double cl = fmax(ceil(xmin - kPruneMargin - 0.5), 0.0);
                double ch = fmin(floor(xmax + kPruneMargin + 0.5), (double)(a.W - 1));
                double rl = fmax(ceil(ymin - kPruneMargin - 0.5), (double)a.y_lo);
                double rh = fmin(floor(ymax + kPruneMargin + 0.5), (double)(a.y_hi - 1));
                if (cl <= ch && rl <= rh) {
                    R.tx0 = (uint16_t)((int)cl >> a.tw_shift); R.tx1 = (uint16_t)((int)ch >> a.tw_shift);
                    R.ty0 = (uint16_t)((int)rl >> a.th_shift); R.ty1 = (uint16_t)((int)rh >> a.th_shift);
                }
            }
        }
        // (depth = -z: farther = smaller z.  NaN on either side: not far.)
        far = (R.s0[2] + R.s1[2]) + R.s2[2] < 3.0 * z_split;
        if (far) R.flags |= kTriFar;
        // A triangle of a few pixels (a few scanlines of a few samples each) is cheaper as ONE thread
        // per half that walks its samples than as slots of the frame-wide sequence: k_small takes it,
        // and it owns no slots.  (mode bit 2: the draw uses that path.)
        if ((a.mode & 4) && res0 && res0 <= a.small_res && res1 <= a.small_res && !(R.flags & kTriNonFinite)) {
            R.flags |= kTriSmall;
            small = true;
        }
        // A triangle of at most 64 scanlines per half and a few hundred samples is one WAVE per half in k_mid: its
        // scanlines are built in registers, its samples spread over the lanes; no slots, no records, no searches in
        // memory.  (mode bit 4: the draw uses that path -- meshes whose triangles cover some hundreds of pixels.)
        if ((a.mode & 16) && !small && res0 && res0 <= 63u && res1 <= 63u && !(R.flags & kTriNonFinite)) {
            const double ext = fmax(fabs(R.sp[0] - R.s1[0]), fabs(R.sp[1] - R.s1[1]));  // the longest scanline
            const double est = (double)((res0 > res1 ? res0 : res1) + 1u) * (ext * 0.5 + 2.0);
            if (est <= (double)a.mid_samples) {  // (NaN: no)
                mid = true;
                small = true;  // (owns no slots either)
            }
        }
        a.rast[seq] = R;
        if ((a.mode & 2) && res0) {
            if (R.flags & kTriNonFinite) {
                int ty_lo = a.y_lo >> a.th_shift, ty_hi = (a.y_hi - 1) >> a.th_shift;
                if (ty_lo == 0)
                    for (int tx = 0; tx < a.tiles_x; ++tx) atomicAdd(&a.tile_cnt[tx], 1u);
                for (int ty = (ty_lo > 1 ? ty_lo : 1); ty <= ty_hi; ++ty) atomicAdd(&a.tile_cnt[ty * a.tiles_x], 1u);
            } else {
                for (int ty = R.ty0; ty <= R.ty1; ++ty)
                    for (int tx = R.tx0; tx <= R.tx1; ++tx) atomicAdd(&a.tile_cnt[ty * a.tiles_x + tx], 1u);
            }
        }
        }  // !out_of_band
    }
    if (a.small_flag)  // (every draw writes it: k_small / k_mid may sweep draws that do not use them)
        a.small_flag[seq] = (uint8_t)((mid ? kFlagMid : (small ? kFlagSmall : 0)) | (far && small ? kFlagFar : 0));
    if (small && !mid) a.small_grp[seq >> 5] = 1;  // (zeroed by tr_clear; lanes of a group store the same byte)
    // deferred shade record: a triangle with slots (or of k_mid) always gets one (those kernels keep no per-triangle
    // state); one of a few pixels (k_small, the bulk of a high-polygon mesh) only if the winner pass flags it
    if (a.mode & 8) a.need_shade[seq] = (res0 && (!small || mid)) ? 1 : 0;
    if (a.mode & 1) {  // slots per half
        cnt0 = res0 && !small ? (res0 + 1u) << R.slot_shift : 0u;
        cnt1 = res1 && !small ? (res1 + 1u) << R.slot_shift : 0u;
        if (!fused) {  // (the fused launch scans them in registers)
            a.half_cnt[2 * (size_t)seq] = cnt0;
            a.half_cnt[2 * (size_t)seq + 1] = cnt1;
        }
    }
}

// Shade records after the fact (draws of many triangles, mode bit 3 of k_setup): one thread per triangle of the draw;
// those the winner pass flagged (a sample of theirs equalled the final depth of its pixel, or they hold a NaN depth)
// get their TriShade now, from the vertex stage evaluated per corner -- the arithmetic of k_transform, so the bits
// are those the eager path would have written.  A 10^7-triangle mesh has a few million such triangles: the others
// never cost 256 B of record traffic.
__device__ __forceinline__ void build_shade_record(const SetupArgs& a, uint32_t k);
__global__ __launch_bounds__(256) void k_build_shade(SetupArgs a, const uint8_t* __restrict__ need_shade) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (uint32_t)a.n_tris) return;
    if (a.cull && a.cull[k >> 6]) return;  // (a band context: the cluster misses its rows, the same byte for the whole wave)
    if (!need_shade[a.tri_base + k]) return;
    build_shade_record(a, k);
}
__device__ __forceinline__ void build_shade_record(const SetupArgs& a, uint32_t k) {
    const uint32_t seq = a.tri_base + k;
    const uint32_t* ix = a.idx + 9 * (size_t)k;
    const uint32_t iv[3] = {ix[0], ix[3], ix[6]};
    const uint32_t it[3] = {ix[1], ix[4], ix[7]};
    const uint32_t in[3] = {ix[2], ix[5], ix[8]};
    double px[3], py[3], pz[3], vx[3], vy[3], vw[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double g[4];
        mat4_mul_vec4(a.pmv, a.pos[3 * (size_t)iv[c]], a.pos[3 * (size_t)iv[c] + 1], a.pos[3 * (size_t)iv[c] + 2], 1.0, g);
        px[c] = g[0] / g[3];
        py[c] = g[1] / g[3];
        pz[c] = g[2] / g[3];
        mat4_mul_vec4(a.vpm, px[c], py[c], pz[c], 1.0, g);  // mod.rs:320-327
        vx[c] = g[0]; vy[c] = g[1]; vw[c] = g[3];
    }
    if (!__all(vw[0] == 1.0 && vw[1] == 1.0 && vw[2] == 1.0)) {  // (w is exactly 1 for finite corners: setup_triangle)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            vx[c] = vx[c] / vw[c];
            vy[c] = vy[c] / vw[c];
        }
    }
    const uint32_t res0 = a.rast[seq].res[0], res1 = a.rast[seq].res[1];  // (k_setup's, written a launch ago)
    a.shade[seq] = make_shade_record(a, iv, it, in, px, py, pz, vx, vy, true, a.nmat, a.draw, res0, res1);
}

// Bounding boxes of the mesh's clusters of 64 consecutive triangles (object space), once per mesh (tr_mesh_create):
// bounds[6 c .. 6 c + 5] = min x, y, z, max x, y, z over the positions the cluster's triangles index.
__global__ __launch_bounds__(256) void k_mesh_bounds(const uint32_t* __restrict__ idx, const double* __restrict__ pos,
                                                     uint32_t n_tris, double* __restrict__ bounds) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t0 = c * 64u;
    if (t0 >= n_tris) return;
    const uint32_t t1 = t0 + 64u < n_tris ? t0 + 64u : n_tris;
    double lo[3] = {__builtin_huge_val(), __builtin_huge_val(), __builtin_huge_val()};
    double hi[3] = {-__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val()};
    bool bad = false;
    for (uint32_t t = t0; t < t1; ++t) {
        // (a triangle of three identical positions is degenerate under every view: all of its samples are NaN and land on pixel
        // (0, 0), far from the cluster's box -- such a cluster is never culled, and setup_triangle gives the triangle to the band
        // that owns row 0.  Triangles that only PROJECT onto one screen row are view-dependent and stay the documented exception.)
        const uint32_t va = idx[9 * (size_t)t], vb = idx[9 * (size_t)t + 3], vc = idx[9 * (size_t)t + 6];
        bool same = true;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            same = same && pos[3 * (size_t)va + d] == pos[3 * (size_t)vb + d] && pos[3 * (size_t)vb + d] == pos[3 * (size_t)vc + d];
        bad = bad || same;
        for (int cnr = 0; cnr < 3; ++cnr) {
            const uint32_t v = idx[9 * (size_t)t + 3 * cnr];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double p = pos[3 * (size_t)v + d];
                bad = bad || !(fabs(p) < __builtin_huge_val());
                lo[d] = fmin(lo[d], p);
                hi[d] = fmax(hi[d], p);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {  // (a non-finite position, or a triangle of one point: a box nothing is culled against)
        bounds[6 * (size_t)c + d] = bad ? __builtin_nan("") : lo[d];
        bounds[6 * (size_t)c + 3 + d] = bad ? __builtin_nan("") : hi[d];
    }
}

// ================================================================= (2b) scan
// Exclusive prefix sum of half_cnt[lo..lo+n) into half_off[lo+1..lo+n], continuing from
// half_off[lo] (0 after a clear, else the previous draw's total).  Three phases over tiles of
// kScanTile entries; a single tile is handled by phase C alone.
constexpr int kScanThreads = 256;
constexpr int kScanPer = 16;
constexpr int kScanTile = kScanThreads * kScanPer;

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total_out) {
    // exclusive scan of one value per thread across a 256-thread block
    __shared__ uint32_t wsum[kScanThreads / 64];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_sum(v);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kScanThreads / 64; ++i) {
        if (i < w) base += wsum[i];
        tot += wsum[i];
    }
    __syncthreads();
    *total_out = tot;
    return base + inc - v;
}

// (Thread t of a tile takes entries [16 t, 16 t + 16) as four 16-byte loads when the arrays allow it: a lane that
// reads its 64 bytes one word at a time makes every load instruction of the wave touch 64 different lines.)
__device__ __forceinline__ void scan_load16(const uint32_t* __restrict__ cnt, uint32_t begin, uint32_t n, uint32_t v[kScanPer],
                                            const uint8_t* __restrict__ cull = nullptr) {
    // (a thread's 16 halves are 8 triangles of one cluster: a culled cluster's counts are zero and were never written)
    if (cull && begin < n && cull[begin >> 7]) {
#pragma unroll
        for (int i = 0; i < kScanPer; ++i) v[i] = 0u;
        return;
    }
    if (begin + kScanPer <= n && ((uintptr_t)(cnt + begin) & 15u) == 0) {
#pragma unroll
        for (int q = 0; q < kScanPer / 4; ++q) {
            const uint4 w = reinterpret_cast<const uint4*>(cnt + begin)[q];
            v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kScanPer; ++i) v[i] = (begin + i < n) ? cnt[begin + i] : 0u;
    }
}

// (cull: the draw's cluster bytes of a band context, or nullptr.  A tile is the 4 096 halves of 32 clusters; when
// k_cull_clusters dropped all of them, every count is zero and the tile is neither read here nor in k_scan_apply.)
__device__ __forceinline__ bool scan_tile_is_dead(const uint8_t* __restrict__ cull, uint32_t n_clusters, uint32_t tile) {
    if (!cull) return false;
    int alive = 0;
    if (threadIdx.x < (uint32_t)(kScanTile / 128)) {
        const uint32_t cl = tile * (uint32_t)(kScanTile / 128) + threadIdx.x;
        alive = (cl < n_clusters && cull[cl] == 0) ? 1 : 0;
    }
    return __syncthreads_or(alive) == 0;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_tile_sums(const uint32_t* __restrict__ cnt,
                                                                 uint32_t n,
                                                                 unsigned long long* tile_sum,
                                                                 const uint8_t* __restrict__ cull, uint32_t n_clusters) {
    uint32_t tile = blockIdx.x;
    if (scan_tile_is_dead(cull, n_clusters, tile)) {
        if (threadIdx.x == 0) tile_sum[tile] = 0ull;
        return;
    }
    uint32_t begin = tile * kScanTile + threadIdx.x * kScanPer;
    uint32_t v[kScanPer];
    scan_load16(cnt, begin, n, v, cull);
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) s += v[i];
    // block reduce (u64)
    __shared__ unsigned long long red[kScanThreads / 64];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < kScanThreads / 64; ++i) t += red[i];
        tile_sum[tile] = t;
    }
}

// single block: tile_sum -> exclusive tile bases (u64, in place), starting at *carry_in.  Thread t owns a run of
// consecutive tiles: one pass, one block scan of the runs' totals (20 rounds of four barriers each took 15 us for the
// 4 984 tiles of a 10^7-triangle draw).
__global__ __launch_bounds__(kScanThreads) void k_scan_tile_bases(unsigned long long* tile_sum,
                                                                  uint32_t n_tiles,
                                                                  const uint32_t* carry_in,
                                                                  uint32_t* flags) {
    __shared__ unsigned long long wtot[kScanThreads / 64];
    const uint32_t per = (n_tiles + kScanThreads - 1) / kScanThreads;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
    unsigned long long mine = 0;
    for (uint32_t i = lo; i < hi; ++i) mine += tile_sum[i];
    unsigned long long inc = mine;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    unsigned long long b = *carry_in, tot = *carry_in;
    for (int k = 0; k < kScanThreads / 64; ++k) {
        if (k < w) b += wtot[k];
        tot += wtot[k];
    }
    unsigned long long run = b + inc - mine;
    for (uint32_t i = lo; i < hi; ++i) {
        const unsigned long long v = tile_sum[i];
        tile_sum[i] = run;
        run += v;
    }
    if (threadIdx.x == 0 && tot > 0xFFFFFFFFull) atomicOr(flags, 1u << 4);
}

// per tile: exclusive scan with the tile base (tile_base == nullptr: single tile, base = off[0])
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const uint32_t* __restrict__ cnt,
                                                             uint32_t n, uint32_t* off,
                                                             const unsigned long long* tile_base,
                                                             uint32_t* flags,
                                                             const uint8_t* __restrict__ cull, uint32_t n_clusters) {
    uint32_t tile = blockIdx.x;
    uint32_t begin = tile * kScanTile + threadIdx.x * kScanPer;
    uint32_t v[kScanPer];
    uint32_t s = 0;
    if (scan_tile_is_dead(cull, n_clusters, tile)) {
        // every entry of the tile is the tile's base, and so is the entry in front of it (the previous tile's last):
        // whole aligned 16-byte groups, a lane beside its neighbour, plus the tile's own last entry
        const uint32_t base32 = (uint32_t)(tile_base ? tile_base[tile] : (unsigned long long)off[0]);
        const uint32_t t0 = tile * kScanTile;
        if (t0 + kScanTile <= n && ((uintptr_t)(off + t0) & 15u) == 0) {
#pragma unroll
            for (int q = 0; q < kScanPer / 4; ++q)
                reinterpret_cast<uint4*>(off + t0)[threadIdx.x + q * kScanThreads] = make_uint4(base32, base32, base32, base32);
            if (threadIdx.x == 0) off[t0 + kScanTile] = base32;
            return;
        }
#pragma unroll
        for (int i = 0; i < kScanPer; ++i) v[i] = 0u;
    } else {
        scan_load16(cnt, begin, n, v, cull);
    }
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) s += v[i];
    uint32_t tot;
    uint32_t ex = block_excl_scan(s, &tot);
    unsigned long long base = tile_base ? tile_base[tile] : (unsigned long long)off[0];
    if (!tile_base && threadIdx.x == 0 && base + tot > 0xFFFFFFFFull) atomicOr(flags, 1u << 4);
    uint32_t run = (uint32_t)base + ex;
    // off[j + 1] = inclusive sum through j: entries begin + 1 .. begin + 16, i.e. three words, three aligned 16-byte
    // stores and a last word when the array allows it
    uint32_t o[kScanPer];
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
        run += v[i];
        o[i] = run;
    }
    if (begin + kScanPer <= n && ((uintptr_t)(off + begin) & 15u) == 0) {
        off[begin + 1] = o[0]; off[begin + 2] = o[1]; off[begin + 3] = o[2];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            reinterpret_cast<uint4*>(off + begin + 4)[q] = make_uint4(o[3 + 4 * q], o[4 + 4 * q], o[5 + 4 * q], o[6 + 4 * q]);
        off[begin + 16] = o[15];
    } else {
#pragma unroll
        for (int i = 0; i < kScanPer; ++i)
            if (begin + i < n) off[begin + i + 1] = o[i];
    }
}

// single block: the whole scan of a small draw (one launch instead of three; the tail of the fused
// k_setup).  Thread t owns 8 consecutive entries in each of up to 4 rows of 2 048: a wave's loads and
// stores are contiguous 2 KB (a thread that owns one long run makes every lane of a load touch a line
// of its own: 3 us per 4 096 entries, measured), all rows are in flight before the first is needed,
// and one barrier per round serves the row scans.  Per-wave sums are 32-bit with the wrap detected (an
// entry is below 2^24, a thread's run below 2^27), the running base is 64-bit: a total beyond 2^32 - 1
// raises the overflow flag either way.
__device__ __forceinline__ void scan_in_block(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* off,
                                              uint32_t* flags) {
    constexpr int kRows = 4, kRun = 8;
    constexpr uint32_t kRow = (uint32_t)kScanThreads * kRun;
    // (two sets of wave totals, used in turn: one barrier per round is enough, and nothing waits for a
    // round's stores before the next round's loads are on their way)
    __shared__ uint32_t wtot[2][kRows][kScanThreads / 64];
    __shared__ uint32_t sOut[kRow * kRows + (kRow * kRows >> 5)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every thread keeps the running base itself (the same value in all of them); off[0] was written
    // before this launch (tr_clear or the previous draw)
    unsigned long long base = off[0];
    bool wrapped = false;
    int buf = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += kRow * kRows, buf ^= 1) {
        uint32_t v[kRows][kRun];
        uint32_t inc[kRows], own[kRows];
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const uint32_t begin = t0 + (uint32_t)r * kRow + threadIdx.x * kRun;
            uint32_t sum = 0;
            // (cnt and begin are 8-byte aligned: the counts of a triangle's two halves come as one load)
#pragma unroll
            for (int q = 0; q < kRun / 2; ++q) {
                uint2 two = make_uint2(0u, 0u);
                if (begin + 2 * q + 1 < n) two = *reinterpret_cast<const uint2*>(cnt + begin + 2 * q);
                else if (begin + 2 * q < n) two.x = cnt[begin + 2 * q];
                v[r][2 * q] = two.x;
                v[r][2 * q + 1] = two.y;
                sum += two.x + two.y;
            }
            own[r] = sum;
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            uint32_t x = own[r];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(x, d, 64);
                if (lane >= d) {
                    x += t;
                    wrapped |= x < t;
                }
            }
            inc[r] = x;
            if (lane == 63) wtot[buf][r][wave] = x;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            unsigned long long before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < kScanThreads / 64; ++w) {
                const unsigned long long t = wtot[buf][r][w];
                if (w < wave) before += t;
                all += t;
            }
            uint32_t run = (uint32_t)(base + before) + (inc[r] - own[r]);
            const uint32_t local = (uint32_t)r * kRow + threadIdx.x * kRun;
#pragma unroll
            for (int e = 0; e < kRun; ++e) {
                run += v[r][e];
                sOut[local + e + ((local + e) >> 5)] = run;  // (one pad word per 32: no bank conflicts either way)
            }
            base += all;
        }
        // the sums leave through LDS so that a wave stores 256 contiguous bytes per instruction (a thread
        // storing its own run writes 4 bytes of every 32: 3 us for 8 192 entries, measured)
        __syncthreads();
#pragma unroll 8
        for (uint32_t j = threadIdx.x; j < kRow * kRows; j += kScanThreads)
            if (t0 + j < n) off[t0 + j + 1] = sOut[j + (j >> 5)];  // off[j+1] = inclusive sum through j
        if (t0 + kRow * kRows < n) __syncthreads();  // (the next round rewrites sOut)
    }
    if (wrapped || (threadIdx.x == 0 && base > 0xFFFFFFFFull)) atomicOr(flags, 1u << 4);
}

__global__ __launch_bounds__(kScanThreads) void k_scan_small(const uint32_t* __restrict__ cnt, uint32_t n,
                                                             uint32_t* off, uint32_t* flags) {
    scan_in_block(cnt, n, off, flags);
}

// =============================================================== (3) raster
// The frame's scanlines form one sequence of SLOTS: half h of triangle t owns
// (res+1) << slot_shift consecutive slots starting at half_off[2t+h]; scanline i sits in slot
// i << slot_shift of that range, the others are empty.  k_scanlines turns every slot into a
// 64-byte ScanRec once per draw; the two raster passes then only stream records.
struct RasterArgs {
    const uint32_t* half_off;   // exclusive prefix of slots per half (+ total at the end)
    const TriRaster* rast;
    const uint8_t* small_flag;  // k_small: per triangle, 1 = its samples are this kernel's
    const uint8_t* small_grp;   // k_small: per 32 triangles (tri_seq >> 5), 1 = some triangle of the group is small
    ScanRec* rec;               // records of slots [0, rec_cap); slots beyond are rebuilt on the fly
    uint32_t rec_cap;
    unsigned long long* depth;  // encoded min depth per pixel, row-major
    unsigned long long* wkey;   // min order key among samples at the min depth
    uint32_t h_lo, h_hi;        // half range [h_lo, h_hi) to enumerate
    uint32_t count_h;           // samples of halves >= count_h are added to the counters
    int W, H;
    int y_lo, y_hi;             // band of rows owned by this context: [y_lo, y_hi)
    unsigned long long* counters;
    uint32_t* flags;
    NanTable* nan;
    uint8_t* blk_touched;       // per shade block (kShadeBW x kShadeBH pixels), at [2 * block]: 1 = a winner key was written into it
    int blocks_x;
    uint8_t* chunk_live;        // per 64-slot chunk of the frame: 1 = some sample may be a winner
    uint32_t live_cap;          // chunks beyond this are always treated as live
    int use_live;               // winner pass: skip chunks the depth pass found dead
    int rec_ready;              // depth pass: k_scanlines already built this range's records
    // winner pass: the OTHER half of the double-buffered per-frame state (the previous frame's, dead
    // since this frame's tr_clear) is zeroed on the side, so that the next tr_clear is a pointer swap
    unsigned long long* clean_state;
    uint32_t clean_words;
    NanTable* clean_nan;
    uint8_t* need_shade;        // k_small<1>, frames with deferred shade records: [tri] = 1 if the triangle may own a pixel (else nullptr)
    // batched views (tr_render_views): triangle t belongs to view t / view_tris, whose planes start view_pix
    // pixels and whose shade blocks start view_blk blocks after the previous view's; 0 = one view
    uint32_t view_tris, view_pix, view_blk;
    uint32_t tiled;             // the planes' layout this frame: 0 row-major, 1 tiles of 4 x 2 pixels (plane_index)
    uint32_t phase;             // k_small / k_mid depth pass: 0 every triangle, 1 the near ones only, 2 the far ones (kFlagFar) only
    uint32_t mid_shift;         // k_mid: log2 of the halves a wave looks up at once (0 .. 6)
    uint32_t box;               // k_raster depth pass: 1 = a chunk's samples are reduced in an LDS box of pixels first
    uint32_t* slots_seen;       // winner pass: the slots of the range it ran over, left in host memory the GPU can write
                                // (the context sizes the next frame's record buffer by it without ever waiting for it)
    int debug;                  // timing experiments only (TR_RASTER_DEBUG): 1 no atomics, 2 no loads either
    unsigned long long* trace;  // TR_RASTER_TRACE: 4 words per chunk (t_begin, t_records, t_end in 10 ns
                                // ticks, samples | block << 32)
    uint32_t trace_cap;
};

// Record of slot g, which lies in half `lo` (half_off[lo] <= g < half_off[lo + 1]).
__device__ __forceinline__ ScanRec record_of(const RasterArgs& a, uint32_t lo, uint32_t g) {
    ScanRec r;
    r.ax = r.ay = r.az = r.dx = r.dy = r.dz = r.inv = 0.0;
    r.keyh = 0ull;
    const uint32_t tri = lo >> 1, half = lo & 1u, s = g - a.half_off[lo];
    const TriRaster& t = a.rast[tri];
    const uint32_t sh = t.slot_shift;
    if (s & ((1u << sh) - 1u)) return r;  // padding slot
    const uint32_t i = s >> sh;
    Scanline sl = make_scanline(t, half, i);
    if (sl.hres > 262143u) {
        atomicOr(a.flags, 1u << 0);
        return r;
    }
    r.ax = sl.ax; r.ay = sl.ay; r.az = sl.az;
    r.dx = sl.dx; r.dy = sl.dy; r.dz = sl.dz;
    r.inv = 1.0 / (double)sl.hres;
    r.keyh = ((unsigned long long)tri << kKeyTriShift) | ((unsigned long long)half << kKeyHalfShift) |
             ((unsigned long long)i << kKeyIShift) | (unsigned long long)sl.hres;
    return r;
}

// Half holding slot g: the last lo in [lo, hi) with half_off[lo] <= g (given half_off[lo] <= g < half_off[hi]).
__device__ __forceinline__ uint32_t find_half(const uint32_t* __restrict__ half_off, uint32_t lo, uint32_t hi,
                                              uint32_t g) {
    while (hi - lo > 1u) {
        uint32_t mid = (lo + hi) >> 1;
        if (half_off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// The same search done by one wave with 64 probes per round (3 dependent loads for 10^5 halves
// instead of 17).  g is wave-uniform; every lane returns the result.
__device__ __forceinline__ uint32_t find_half_wave(const uint32_t* __restrict__ half_off, uint32_t lo, uint32_t hi,
                                                   uint32_t g) {
    const uint32_t lane = threadIdx.x & 63u;
    while (hi - lo > 1u) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t idx = lo + (lane + 1u) * step;  // probes lo+step, lo+2 step, ...
        const bool le = idx < hi && half_off[idx] <= g;
        const uint32_t n = (uint32_t)__popcll(__ballot(le));  // monotone: the first n probes hold
        const uint32_t nlo = lo + n * step;
        const uint32_t nhi = nlo + step;
        lo = nlo;
        hi = nhi < hi ? nhi : hi;
    }
    return lo;
}

__device__ __forceinline__ ScanRec build_record(const RasterArgs& a, uint32_t g) {
    return record_of(a, find_half(a.half_off, a.h_lo, a.h_hi, g), g);
}

constexpr int kHalfCache = 256;  // prefix entries a chunk can hold in LDS while it locates its halves

// Large draws of small triangles (BASELINE.json configs[3]): one thread per slot builds the
// records ahead of the depth pass, which would otherwise build them 64 at a time with one wave
// per workgroup.  A block owns 256 consecutive slots; two waves locate the halves of its first and
// last slot, the prefix entries between them go to LDS, each thread finishes its search there.
constexpr int kScanlineCache = 1024;
__global__ __launch_bounds__(256) void k_scanlines(RasterArgs a) {
    __shared__ uint32_t sOff[kScanlineCache + 1];
    __shared__ uint32_t sEnds[2];
    const uint32_t s_begin = a.half_off[a.h_lo];
    const uint32_t s_end = a.half_off[a.h_hi];
    unsigned long long n_scan = 0;
    for (unsigned long long b0 = (unsigned long long)s_begin + (unsigned long long)blockIdx.x * 256ull; b0 < s_end;
         b0 += (unsigned long long)gridDim.x * 256ull) {
        const uint32_t g_first = (uint32_t)b0;
        const uint32_t g_last = (uint32_t)(b0 + 255ull < s_end ? b0 + 255ull : (unsigned long long)s_end - 1ull);
        const int w = threadIdx.x >> 6;
        if (w < 2) {
            const uint32_t e = find_half_wave(a.half_off, a.h_lo, a.h_hi, w == 0 ? g_first : g_last);
            if ((threadIdx.x & 63) == 0) sEnds[w] = e;
        }
        __syncthreads();
        const uint32_t e_first = sEnds[0], e_last = sEnds[1];
        const uint32_t n_e = e_last - e_first + 1u;
        const bool cached = n_e <= (uint32_t)kScanlineCache;
        if (cached)
            for (uint32_t t = threadIdx.x; t <= n_e; t += 256u) sOff[t] = a.half_off[e_first + t];
        __syncthreads();
        const uint32_t g = g_first + threadIdx.x;
        if (g <= g_last) {
            uint32_t lo;
            if (cached) {
                uint32_t l = 0, h = n_e;  // sOff[l] <= g < sOff[h]
                while (h - l > 1u) {
                    uint32_t mid = (l + h) >> 1;
                    if (sOff[mid] <= g) l = mid; else h = mid;
                }
                lo = e_first + l;
            } else {
                lo = find_half(a.half_off, e_first, e_last + 1u, g);
            }
            const ScanRec r = record_of(a, lo, g);
            if (g < a.rec_cap) a.rec[g] = r;  // slots beyond the capacity are rebuilt by every raster pass
            n_scan += (lo >= a.count_h && (r.keyh & kRecHresMask)) ? 1 : 0;
        }
        __syncthreads();  // sOff / sEnds are rewritten by the next round
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) n_scan += __shfl_down(n_scan, d, 64);
    if ((threadIdx.x & 63) == 0 && n_scan)
        atomicAdd(&a.counters[(size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) % kCounterSlots) * 8 + 2], n_scan);
}

// Threads per chunk (template parameter of k_raster): 256 = four waves share a chunk's samples, the
// shortest launch when the GPU renders one frame at a time; 128 = less of a chunk's life is spent with
// waves waiting for its records, the better throughput when frames are in flight on several streams
// (measured at 4K: one context 6 840 / 6 430 frames/s, four contexts 9 750 / 10 450).
constexpr int kChunkSlots = 64;
// The depth pass's LDS boxes; the values are the best of a sweep on one MI355X box (`make variant EXTRA=-D...`,
// tools/ab_bench.sh): 2 048 pixels for 256 threads cost more in waves per SIMD than they save in atomics, 8-slot boxes send
// 5 % fewer atomic requests than 16-slot ones and are no faster.
#ifndef TR_BOX_PIX
#define TR_BOX_PIX 1024
#endif
#ifndef TR_BOX_SUB
#define TR_BOX_SUB 4
#endif
#ifndef TR_BOX_PIX_128
#define TR_BOX_PIX_128 512
#endif
constexpr int kBoxPixMax = TR_BOX_PIX;     // pixels a chunk can hold in its LDS boxes (8 bytes each): workgroups of 256 threads
constexpr int kBoxPix128 = TR_BOX_PIX_128; // ... of 128 threads (frames in flight: the LDS then bounds the waves per SIMD)
constexpr int kBoxSubShift = TR_BOX_SUB;   // 2^this consecutive slots of a chunk share one box
constexpr int kBoxSubs = kChunkSlots >> kBoxSubShift;
static_assert(kBoxPixMax <= 2048 && kBoxPix128 <= 2048, "the flush's division by a box's width is exact below 2048 entries only");


// One workgroup per chunk of 64 consecutive slots (chunks are aligned to the frame's slot
// sequence, so both passes and every draw cut it the same way).  The records go to LDS with one
// 16-byte load per thread; wave 0 scans the sample counts; then lanes own SAMPLES: consecutive
// lanes take consecutive samples of a scanline, i.e. consecutive pixels of a row, so the
// depth / winner atomics of one wave-instruction fall into contiguous 8-byte slots.
// VIEWS: the frame is a batch of views of one mesh (tr_render_views); a slot's samples land in the plane slice
// and the shade blocks of its triangle's view.
template <int PASS, bool PRECHECK, bool DBG, int kRasterThreads, bool VIEWS>
__device__ __forceinline__ void raster_body(const RasterArgs& a) {
    // (timing experiments and the per-chunk timeline live in the DBG instantiations only)
    const int dbg_mode = DBG ? a.debug : 0;
    unsigned long long* const dbg_trace = DBG ? a.trace : nullptr;
    __shared__ uint4 sRec4[kChunkSlots * 4];
    __shared__ __align__(16) uint32_t sPre[kChunkSlots + 4];  // exclusive prefix of the slots' sample counts, [64] = total
    __shared__ uint32_t sOff[PASS == 0 ? kHalfCache + 1 : 1];
    __shared__ uint32_t sEnds[2];
    __shared__ uint32_t sView[VIEWS ? kChunkSlots : 1];  // view of each slot's triangle
    const double* sRecD = reinterpret_cast<const double*>(sRec4);
    const unsigned long long* sRecU = reinterpret_cast<const unsigned long long*>(sRec4);
    // depth pass, box form: the minima of the chunk's samples per pixel of the chunk's screen box (empty = all ones)
    constexpr bool kBoxOn = PASS == 0 && PRECHECK && !VIEWS;
    constexpr int kBoxPix = kRasterThreads == 128 ? kBoxPix128 : kBoxPixMax;
    __shared__ unsigned long long sBox[kBoxOn ? kBoxPix : 1];
    __shared__ uint4 sBoxA[kBoxOn ? kBoxSubs : 1];        // per box: x0, y0, width | height << 16 (0: not boxed), first entry
    __shared__ uint32_t sBoxMagic[kBoxOn ? kBoxSubs : 1];  // per box: 2^21 / width + 1
    __shared__ uint32_t sBoxEnd[kBoxOn ? kBoxSubs : 1];    // per box: one past its last entry
    __shared__ uint32_t sBoxHdr[2];                        // entries in use, 1 = some samples of the chunk are not boxed
    __shared__ uint32_t sLive[2];                          // depth pass: a sample of the chunk reached the plane's value (per chunk parity)
    if (threadIdx.x == 0) { sLive[0] = 0u; sLive[1] = 0u; }
    const bool box_on = kBoxOn && a.box != 0;
    if (box_on)
        for (uint32_t p = threadIdx.x; p < (uint32_t)kBoxPix; p += (uint32_t)kRasterThreads) sBox[p] = ~0ull;

    // samples a thread keeps in flight, measured on MI355X at 4K: the winner pass is fastest with 3 (30.1 us; 2: 30.7,
    // 4: 31.2); the depth pass was fastest with 1 while every sample loaded the plane and issued its own atomic (58 us;
    // 63 with 2), and is with 2 now that its samples go to the LDS boxes (one context 8 230 against 8 090 frames/s)
#ifndef TR_WINNER_BATCH
#define TR_WINNER_BATCH 3
#endif
#ifndef TR_DEPTH_BATCH
#define TR_DEPTH_BATCH 2
#endif
    constexpr int kBatch = PASS == 0 ? TR_DEPTH_BATCH : TR_WINNER_BATCH;
    const int lane = threadIdx.x & 63;
    const uint32_t band_rows = (uint32_t)(a.y_hi - a.y_lo);
    const bool wave0 = threadIdx.x < 64;
    const uint32_t s_begin = a.half_off[a.h_lo];
    const uint32_t s_end = a.half_off[a.h_hi];
    const uint32_t s_count = PASS == 0 ? a.half_off[a.count_h > a.h_lo ? a.count_h : a.h_lo] : 0u;
    const uint32_t c_lo = s_begin / 64u;
    const uint32_t c_hi = (uint32_t)(((unsigned long long)s_end + 63ull) / 64ull);
    // (32-bit while they are being added to: flushed to the 64-bit counters before they can wrap)
    uint32_t cnt_samples = 0, cnt_inb = 0, cnt_scan = 0;
    auto flush_counts = [&]() {
        uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)cnt_samples, 0), ci = (uint32_t)__builtin_amdgcn_readlane((int)cnt_inb, 0), sc = cnt_scan;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sc += __shfl_down(sc, d, 64);
        if ((threadIdx.x & 63) == 0 && (cs | sc)) {
            // one 64-byte slot per block (mod kCounterSlots): same-address atomics serialise
            unsigned long long* slot = a.counters + (size_t)(blockIdx.x % kCounterSlots) * 8;
            atomicAdd(&slot[0], (unsigned long long)cs);
            atomicAdd(&slot[1], (unsigned long long)ci);
            if (sc) atomicAdd(&slot[2], (unsigned long long)sc);
        }
        cnt_samples = cnt_inb = cnt_scan = 0;
    };
    if (PASS == 1 && a.clean_state) {
        for (uint32_t w = blockIdx.x * (uint32_t)kRasterThreads + threadIdx.x; w < a.clean_words;
             w += gridDim.x * (uint32_t)kRasterThreads)
            a.clean_state[w] = 0ull;
        if (blockIdx.x == 0 && threadIdx.x < kNanSlots) {
            a.clean_nan->pix[threadIdx.x] = kNanEmpty;
            a.clean_nan->first_nan[threadIdx.x] = kKeyNone;
            a.clean_nan->first_any[threadIdx.x] = kKeyNone;
        }
    }
    bool nan_active = false;
    if (PASS == 1) nan_active = a.nan->pix[0] != kNanEmpty;
    if (PASS == 1 && a.slots_seen && blockIdx.x == 0 && threadIdx.x == 0) *a.slots_seen = s_end;

    uint32_t n_voted = 0;
    for (uint32_t chunk = c_lo + blockIdx.x; chunk < c_hi; chunk += gridDim.x) {
        if (PASS == 1 && chunk < a.live_cap) {
            // a chunk none of whose samples reached the depth plane's value cannot hold a winner
            // (a NaN-depth pixel needs every sample that lands on it: no skipping then)
            const bool dead = a.chunk_live[chunk] == 0;
            if (dead && a.use_live && !nan_active) continue;
        }
        bool any_live = false;
        unsigned long long tr_t0 = 0, tr_t1 = 0;
        if (dbg_trace && threadIdx.x == 0) tr_t0 = wall_clock64();
        // ---- record phase.  Depth pass: build the chunk's 64 records (two waves locate the halves
        // of its first and last slot with 64 probes per round, the prefix entries between them go
        // to LDS, 64 lanes finish their search there), keep them in LDS and store them for the
        // winner pass.  Winner pass: thread t brings 16 bytes (part t & 3) of slot t >> 2.
        if (PASS == 0 && !a.rec_ready) {
            const uint32_t g_first = chunk * 64u > s_begin ? chunk * 64u : s_begin;
            const uint32_t g_last = chunk * 64u + 63u < s_end - 1u ? chunk * 64u + 63u : s_end - 1u;
            if (g_first > g_last) continue;  // (uniform) the range ends exactly at this chunk's start
            for (int w = threadIdx.x >> 6; w < 2; w += kRasterThreads / 64) {  // (wave-uniform)
                const uint32_t e = find_half_wave(a.half_off, a.h_lo, a.h_hi, w == 0 ? g_first : g_last);
                if (lane == 0) sEnds[w] = e;
            }
            __syncthreads();
            const uint32_t e_first = sEnds[0], e_last = sEnds[1];
            const uint32_t n_e = e_last - e_first + 1u;
            const bool cached = n_e <= (uint32_t)kHalfCache;
            if (cached)
                for (uint32_t t = threadIdx.x; t <= n_e; t += (uint32_t)kRasterThreads) sOff[t] = a.half_off[e_first + t];
            __syncthreads();
            if (wave0) {
                const uint32_t g = chunk * 64u + (uint32_t)lane;
                ScanRec r;
                r.ax = r.ay = r.az = r.dx = r.dy = r.dz = r.inv = 0.0;
                r.keyh = 0ull;
                if (g >= g_first && g <= g_last) {
                    uint32_t lo;
                    if (cached) {
                        uint32_t l = 0, h = n_e;  // sOff[l] <= g < sOff[h]
                        while (h - l > 1u) {
                            uint32_t mid = (l + h) >> 1;
                            if (sOff[mid] <= g) l = mid; else h = mid;
                        }
                        lo = e_first + l;
                    } else {
                        lo = find_half(a.half_off, e_first, e_last + 1u, g);
                    }
                    r = record_of(a, lo, g);
                    if (g < a.rec_cap) a.rec[g] = r;
                    if (lo >= a.count_h && (r.keyh & kRecHresMask)) cnt_scan += 1;
                }
                ScanRec* dst = reinterpret_cast<ScanRec*>(sRec4) + lane;
                *dst = r;
            }
        } else {
            for (uint32_t t = threadIdx.x; t < 4u * (uint32_t)kChunkSlots; t += (uint32_t)kRasterThreads) {
            const uint32_t g = chunk * 64u + (t >> 2);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (g >= s_begin && g < s_end) {
                if (g < a.rec_cap) {
                    v = reinterpret_cast<const uint4*>(a.rec + g)[t & 3u];
                } else {
                    const ScanRec r = build_record(a, g);
                    const int part = (int)(t & 3u);
                    const double d0 = part == 0 ? r.ax : part == 1 ? r.az : part == 2 ? r.dy : r.inv;
                    const unsigned long long u1 = part == 0 ? (unsigned long long)__double_as_longlong(r.ay)
                                                 : part == 1 ? (unsigned long long)__double_as_longlong(r.dx)
                                                 : part == 2 ? (unsigned long long)__double_as_longlong(r.dz) : r.keyh;
                    const unsigned long long u0 = (unsigned long long)__double_as_longlong(d0);
                    v = make_uint4((uint32_t)u0, (uint32_t)(u0 >> 32), (uint32_t)u1, (uint32_t)(u1 >> 32));
                }
            }
            sRec4[t] = v;
            }
        }
        __syncthreads();
        if (wave0) {
            // (wave scans and row reductions of this prologue in the VALU's DPP path, not through `ds_bpermute`: they sit on every
            // chunk's serial stretch -- 4K depth pass 52.5 -> 50.3 us, winner pass 27.2 -> 26.0 us)
            const uint32_t h = (uint32_t)(sRecU[lane * 8 + 7] & kRecHresMask);
            const uint32_t inc = wave_inclusive_sum(h ? h + 1u : 0u);
            sPre[lane + 1] = inc;
            if (lane == 0) sPre[0] = 0;
            if (VIEWS) sView[lane] = (uint32_t)(sRecU[lane * 8 + 7] >> kKeyTriShift) / a.view_tris;  // (an empty slot: view 0, never read)
        }
        if (box_on && (threadIdx.x >> 6) == kRasterThreads / 64 - 1) {
            // the screen box of every group of 2^kBoxSubShift slots from its records' end points: a scanline's pixel
            // coordinates are monotone in j (every operation of x(j) = ax + (j * inv) * dx is), so samples 0 and hres
            // bound them.  Boxes get their entries in slot order while they fit; the samples of the others go to the plane
            // one by one.
            const double* r = sRecD + lane * 8;
            const uint32_t hres = (uint32_t)(sRecU[lane * 8 + 7] & kRecHresMask);
            int x_lo = 0x7fffffff, x_hi = -1, y_lo = 0x7fffffff, y_hi = -1;
            if (hres) {
                const double sp = (double)hres * r[6];
                const int xa = pixel_coord_raw(r[0]), xb = pixel_coord_raw(r[0] + sp * r[3]);
                const int ya = pixel_coord_raw(r[1]), yb = pixel_coord_raw(r[1] + sp * r[4]);
                const int xl = max(min(xa, xb), 0), xh = min(max(xa, xb), a.W - 1);
                const int yl = max(min(ya, yb), a.y_lo), yh = min(max(ya, yb), a.y_hi - 1);
                if (xl <= xh && yl <= yh) { x_lo = xl; x_hi = xh; y_lo = yl; y_hi = yh; }
            }
            if (kBoxSubShift == 4) {
                x_lo = row16_min(x_lo); y_lo = row16_min(y_lo); x_hi = row16_max(x_hi); y_hi = row16_max(y_hi);
            } else
#pragma unroll
            for (int d = 1; d < (1 << kBoxSubShift); d <<= 1) {
                x_lo = min(x_lo, __shfl_xor(x_lo, d, 64));
                y_lo = min(y_lo, __shfl_xor(y_lo, d, 64));
                x_hi = max(x_hi, __shfl_xor(x_hi, d, 64));
                y_hi = max(y_hi, __shfl_xor(y_hi, d, 64));
            }
            const bool leader = (lane & ((1 << kBoxSubShift) - 1)) == 0;
            const uint32_t w = x_hi >= x_lo ? (uint32_t)(x_hi - x_lo + 1) : 0u;
            const uint32_t h = y_hi >= y_lo ? (uint32_t)(y_hi - y_lo + 1) : 0u;
            const bool has = w != 0u && h != 0u;
            const bool fits = has && w <= 1024u && (unsigned long long)w * h <= (unsigned long long)kBoxPix;
            const uint32_t n = leader && fits ? w * h : 0u;
            const uint32_t incl = wave_inclusive_sum(n);
            const bool boxed = fits && incl <= (uint32_t)kBoxPix;
            const unsigned long long m_boxed = __ballot(leader && boxed);
            const unsigned long long m_open = __ballot(leader && has && !boxed);
            const uint32_t used = m_boxed ? (uint32_t)__builtin_amdgcn_readlane((int)incl, 63 - __builtin_clzll(m_boxed)) : 0u;
            if (leader) {
                const int b = lane >> kBoxSubShift;
                sBoxA[b] = make_uint4((uint32_t)x_lo, (uint32_t)y_lo, boxed ? (w | (h << 16)) : 0u, incl - n);
                sBoxMagic[b] = boxed ? (1u << 21) / w + 1u : 0u;
                sBoxEnd[b] = boxed ? incl : (incl < used ? incl : used);
            }
            if (lane == 0) {
                sBoxHdr[0] = used;
                sBoxHdr[1] = m_open ? 1u : 0u;
            }
        }
        __syncthreads();
        const uint32_t total = sPre[64];
        uint32_t box_n = 0;   // entries of sBox this chunk uses (uniform)
        bool box_open = true;  // some samples go to the plane directly (uniform)
        if (box_on) {
            box_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)sBoxHdr[0]);
            box_open = __builtin_amdgcn_readfirstlane((int)sBoxHdr[1]) != 0 || box_n == 0u;
        }
        if (dbg_trace && threadIdx.x == 0) tr_t1 = wall_clock64();
        // first samples of slots 8, 16, .., 56: the same for every lane (scalar registers); a sample
        // finds its group of 8 slots among them without touching LDS
        const uint32_t g1 = sPre[8], g2 = sPre[16], g3 = sPre[24], g4 = sPre[32], g5 = sPre[40], g6 = sPre[48], g7 = sPre[56];

        // ---- sample phase, kBatch samples per thread in flight: the depth-plane loads of a batch
        // are issued together and waited for once (a wave's wait on a load also waits for the
        // atomics it issued before it, so one sample at a time costs a full memory round trip
        // plus an atomic round trip per sample)
        for (uint32_t k0 = threadIdx.x; k0 < total; k0 += (uint32_t)(kRasterThreads * kBatch)) {
            uint32_t pix[kBatch], meta[kBatch], blk[kBatch];  // meta = slot in chunk << 18 | j
            unsigned long long enc[kBatch];
            int st[kBatch];  // 0: nothing to do, 1: finite depth, 2: NaN depth
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const uint32_t k = k0 + (uint32_t)(u * kRasterThreads);
                st[u] = 0; pix[u] = 0; meta[u] = 0; enc[u] = 0; blk[u] = 0;
                if (k < total) {
                    // slot of sample k = the last slot whose first sample is <= k: group of 8 from the
                    // uniform values, then the group's 8 prefix entries in two 16-byte LDS reads and a
                    // selection tree in registers (one LDS round trip instead of six dependent ones)
                    const bool c4 = g4 <= k;
                    const bool c2 = (c4 ? g6 : g2) <= k;
                    const bool c1 = (c4 ? (c2 ? g7 : g5) : (c2 ? g3 : g1)) <= k;
                    const uint32_t grp = (c4 ? 4u : 0u) + (c2 ? 2u : 0u) + (c1 ? 1u : 0u);
                    const uint4 ea = *reinterpret_cast<const uint4*>(&sPre[8u * grp]);
                    const uint4 eb = *reinterpret_cast<const uint4*>(&sPre[8u * grp + 4u]);
                    const bool d4 = eb.x <= k;
                    const uint32_t a0 = d4 ? eb.x : ea.x, a1 = d4 ? eb.y : ea.y, a2 = d4 ? eb.z : ea.z, a3 = d4 ? eb.w : ea.w;
                    const bool d2 = a2 <= k;
                    const uint32_t b0 = d2 ? a2 : a0, b1 = d2 ? a3 : a1;
                    const bool d1 = b1 <= k;
                    const uint32_t sidx = 8u * grp + (d4 ? 4u : 0u) + (d2 ? 2u : 0u) + (d1 ? 1u : 0u);
                    const uint32_t j = k - (d1 ? b1 : b0);
                    const double* r = sRecD + sidx * 8;
                    const double sp = (double)j * r[6];
                    const double x = r[0] + sp * r[3];
                    const double y = r[1] + sp * r[4];
                    const double z = r[2] + sp * r[5];
                    const int ix = pixel_coord_raw(x);
                    const int iy = pixel_coord_raw(y);
                    const bool inb = (uint32_t)ix < (uint32_t)a.W && (uint32_t)(iy - a.y_lo) < band_rows;
                    if (PASS == 0) {
                        // counted per wave (a vote and a scalar population count instead of two
                        // 64-bit adds in every lane); the totals are taken from lane 0 at the end
                        const bool counted = chunk * 64u + sidx >= s_count;
#ifndef TR_COUNT_ACT
                        cnt_samples += (uint32_t)__popcll(__ballot(counted));
                        cnt_inb += (uint32_t)__popcll(__ballot(counted && inb));
#endif
                    }
                    if (inb) {
                        const double dist = -z;
                        pix[u] = plane_index((uint32_t)ix, (uint32_t)iy, (uint32_t)a.W, a.tiled != 0);
                        meta[u] = (sidx << 18) | j;
                        if (PASS == 1) blk[u] = (uint32_t)(iy / kShadeBH) * (uint32_t)a.blocks_x + (uint32_t)(ix / kShadeBW);
                        if (VIEWS) {
                            const uint32_t vw = sView[sidx];
                            pix[u] += vw * a.view_pix;
                            if (PASS == 1) blk[u] += vw * a.view_blk;
                        }
                        if (dist != dist) {
                            st[u] = 2;
                        } else {
                            st[u] = 1;
                            enc[u] = encode_depth(dist);
                            if (PASS == 0 && dist == -__builtin_huge_val()) atomicOr(a.flags, 1u << 3);
                            if (kBoxOn && box_n) {
                                const uint4 bh = sBoxA[sidx >> kBoxSubShift];
                                const uint32_t bw = bh.z & 0xffffu, bx = (uint32_t)ix - bh.x, by = (uint32_t)iy - bh.y;
                                if (bx < bw && by < (bh.z >> 16)) {  // (else: not boxed, or a scanline that is not monotone)
                                    atomicMin(&sBox[bh.w + by * bw + bx], enc[u]);
                                    st[u] = 0;  // settled by the flush below
                                } else if (!box_open) {  // (every box of the chunk fitted, so nobody loads the plane: never seen)
                                    atomicMin(&a.depth[pix[u]], enc[u]);
                                    any_live = true;
                                    st[u] = 0;
                                }
                            }
                        }
                    }
                }
            }
            unsigned long long cur[kBatch];
            const bool want_load = (PASS == 1 || PRECHECK) && dbg_mode != 2 && !(kBoxOn && !box_open);
#pragma unroll
            for (int u = 0; u < kBatch; ++u)  // (lanes with nothing to do read pixel 0: no branch around the load)
                cur[u] = want_load ? a.depth[st[u] == 1 ? pix[u] : 0u] : 0ull;
            // every comparison against the loaded values first (one wait), then the atomics, none
            // of which reads a loaded register: the compiler needs no further wait between them
            bool act[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                if (PASS == 0) {
                    act[u] = st[u] == 1 && (!PRECHECK || enc[u] < cur[u]) && dbg_mode == 0;
                    any_live |= st[u] == 2 || (st[u] == 1 && (!PRECHECK || enc[u] <= cur[u]));
                } else {
                    act[u] = st[u] == 1 && cur[u] == enc[u];
                }
            }
            bool rare = false;
#ifdef TR_COUNT_ACT  // (diagnostic build: "samples" = wave-instructions with an atomic, "fragments" = lanes with one)
            if (PASS == 0) {
                const unsigned long long m = __ballot(act[0]);
                cnt_samples += m ? 1u : 0u;
                cnt_inb += (uint32_t)__popcll(m);
            }
#endif
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                if (PASS == 0) {
                    if (act[u]) atomicMin(&a.depth[pix[u]], enc[u]);
                    rare |= st[u] == 2;
                } else {
                    const unsigned long long key = (sRecU[(meta[u] >> 18) * 8 + 7] & ~kRecHresMask) | (unsigned long long)(meta[u] & (uint32_t)kRecHresMask);
                    if (act[u] && (!PRECHECK || key < a.wkey[pix[u]])) {
                        atomicMin(&a.wkey[pix[u]], key);
                        a.blk_touched[2u * blk[u]] = 1;  // the shade pass skips blocks nobody wrote a key into
                    }
                    rare |= nan_active && st[u] != 0;
                }
            }
            if (rare) {  // NaN depths (a degenerate triangle) and the pixels they touched
#pragma unroll 1
                for (int u = 0; u < kBatch; ++u) {
                    // (selects instead of dynamic indexing: the arrays stay in registers)
                    auto pick = [&](const auto* v) { return u == 0 ? v[0] : u == 1 ? v[kBatch > 1 ? 1 : 0] : u == 2 ? v[kBatch > 2 ? 2 : 0] : v[kBatch - 1]; };
                    const int stu = pick(st);
                    const uint32_t pu = pick(pix);
                    if (stu == 0) continue;
                    if (PASS == 0) {
                        if (stu != 2) continue;
                        bool placed = false;
                        for (uint32_t s = 0; s < kNanSlots && !placed; ++s) {
                            uint32_t old = atomicCAS(&a.nan->pix[s], kNanEmpty, pu);
                            placed = old == kNanEmpty || old == pu;
                        }
                        if (!placed) atomicOr(a.flags, 1u << 5);
                    } else {
                        const uint32_t mu = pick(meta);
                        const uint32_t su = mu >> 18, ju = mu & (uint32_t)kRecHresMask;
                        const unsigned long long key = (sRecU[su * 8 + 7] & ~kRecHresMask) | (unsigned long long)ju;
                        for (uint32_t s = 0; s < kNanSlots; ++s) {
                            if (a.nan->pix[s] == pu) {
                                if (stu == 2) atomicMin((unsigned long long*)&a.nan->first_nan[s], key);
                                else atomicMin((unsigned long long*)&a.nan->first_any[s], key);
                            }
                        }
                    }
                }
            }
        }
        if (kBoxOn && box_n) {
            // ---- flush: the boxes' minima go to the plane, one load + at most one atomic per touched PIXEL; consecutive
            // lanes = consecutive pixels of a box row.  Entries are left empty for the next chunk.
            __syncthreads();
            uint32_t ends[kBoxSubs > 1 ? kBoxSubs - 1 : 1] = {0u};
#pragma unroll
            for (int b = 0; b < kBoxSubs - 1; ++b) ends[b] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sBoxEnd[b]);
            constexpr int kFlush = 4;
            for (uint32_t p0 = threadIdx.x; p0 < box_n; p0 += (uint32_t)(kRasterThreads * kFlush)) {
                unsigned long long v[kFlush], cur[kFlush];
                uint32_t px[kFlush];
#pragma unroll
                for (int u = 0; u < kFlush; ++u) {
                    const uint32_t p = p0 + (uint32_t)(u * kRasterThreads);
                    v[u] = p < box_n ? sBox[p] : ~0ull;
                    px[u] = 0;
                    if (v[u] != ~0ull) {
                        sBox[p] = ~0ull;
                        uint32_t b = 0;
#pragma unroll
                        for (int i = 0; i < kBoxSubs - 1; ++i) b += p >= ends[i] ? 1u : 0u;
                        const uint4 bh = sBoxA[b];
                        const uint32_t bw = bh.z & 0xffffu, pl = p - bh.w;
                        const uint32_t by = (pl * sBoxMagic[b]) >> 21, bx = pl - by * bw;  // pl / bw, exact for pl < 2048 and bw <= 1024
                        px[u] = plane_index(bh.x + bx, bh.y + by, (uint32_t)a.W, a.tiled != 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < kFlush; ++u) cur[u] = v[u] != ~0ull && dbg_mode != 2 ? a.depth[px[u]] : ~0ull;
#pragma unroll
                for (int u = 0; u < kFlush; ++u) {
                    if (v[u] == ~0ull) continue;
                    any_live |= v[u] <= cur[u];
                    if (v[u] < cur[u] && dbg_mode == 0) atomicMin(&a.depth[px[u]], v[u]);
                }
            }
        }
        // One barrier (also the one before the next chunk rewrites LDS), and for the depth pass the chunk's "live" vote through
        // a word of LDS per chunk parity (`__syncthreads_or` is three barriers and three LDS operations; the word a chunk
        // uses is zeroed by thread 0 a chunk ahead, after it has read the word's previous value and barriers before anyone writes it again)
        int live = 0;
        const uint32_t par = n_voted++ & 1u;  // (votes this workgroup has held: uniform)
        if (PASS == 0) {
            if (threadIdx.x == 0) sLive[par ^ 1u] = 0u;
            if (__ballot(any_live) != 0ull && lane == 0) sLive[par] = 1u;
        }
        __syncthreads();
        if (PASS == 0 && threadIdx.x == 0) live = (int)sLive[par];
        if (dbg_trace && threadIdx.x == 0 && chunk - c_lo < a.trace_cap) {
            unsigned long long* t = dbg_trace + 4 * (size_t)(chunk - c_lo);
            t[0] = tr_t0; t[1] = tr_t1; t[2] = wall_clock64();
            t[3] = (unsigned long long)total | ((unsigned long long)blockIdx.x << 32);
        }
        if (PASS == 0 && (uint32_t)__builtin_amdgcn_readlane((int)cnt_samples, 0) > 0x40000000u) flush_counts();  // (uniform per wave)
        if (threadIdx.x == 0 && chunk < a.live_cap) {
            if (PASS == 0) {
                if (live) a.chunk_live[chunk] = 1;
            } else {
                a.chunk_live[chunk] = 0;  // consumed: the next frame starts from zeros
            }
        }
    }
    if (PASS == 0) {
        // cnt_samples / cnt_inb are per-wave totals already (every lane that took part in a vote holds
        // them; lane 0 takes part in every one its wave casts); cnt_scan is per lane
        flush_counts();
    }
}

template <int PASS, bool PRECHECK, bool DBG, int kRasterThreads, bool VIEWS = false>
__global__ __launch_bounds__(kRasterThreads) void k_raster(RasterArgs a) {
    raster_body<PASS, PRECHECK, DBG, kRasterThreads, VIEWS>(a);
}
// The two passes in the form frames are rendered with (one view; filter in front of the depth atomics, none in front of the
// winner's), each held to the registers of kOcc waves per SIMD.
#ifndef TR_DEPTH_OCC_128
#define TR_DEPTH_OCC_128 8
#endif
#ifndef TR_DEPTH_OCC_256
#define TR_DEPTH_OCC_256 8
#endif
#ifndef TR_WINNER_OCC_128
#define TR_WINNER_OCC_128 8
#endif
#ifndef TR_WINNER_OCC_256
#define TR_WINNER_OCC_256 8
#endif
template <int kRasterThreads, int kOcc>
__global__ __launch_bounds__(kRasterThreads) __attribute__((amdgpu_waves_per_eu(kOcc, kOcc))) void k_raster_depth(RasterArgs a) {
    raster_body<0, true, false, kRasterThreads, false>(a);
}
template <int kRasterThreads, int kOcc>
__global__ __launch_bounds__(kRasterThreads) __attribute__((amdgpu_waves_per_eu(kOcc, kOcc))) void k_raster_winner(RasterArgs a) {
    raster_body<1, false, false, kRasterThreads, false>(a);
}

// ---------------------------------------------------------------- small triangles
// The same two passes for triangles flagged kTriSmall (k_setup), which own no slots: one thread per
// triangle HALF walks its (res + 1) x (hres + 1) samples in the reference's order.  A high-polygon
// mesh (BASELINE.json configs[3]: 10^7 triangles of under a pixel) puts all of a half's samples on
// one or two pixels, so the thread merges consecutive samples of a pixel before it touches memory:
// one depth test + atomic per pixel instead of one per sample, and no scanline records at all.
// half_live[h] (depth pass: some sample was <= the plane when it arrived) lets the winner pass skip
// the halves that cannot hold a winner.
// A fixed grid strides over the halves: a context that owns a band of a 10^7-triangle frame keeps a fraction of the
// triangles, and a launch of one thread per half of the frame spent most of its time dispatching workgroups with
// nothing to do (0.25 ms of a 1 ms band); here such a wave reads 64 flag bytes and moves on.  (A compacted LIST of
// the small triangles, appended to by k_setup with one atomic per wave, was tried first: 160 000 atomics on one
// counter serialise at the memory side, k_setup went from 0.6 to 1.8 ms.)
template <int PASS, bool PRECHECK>
__global__ __launch_bounds__(256) void k_small(RasterArgs a, uint8_t* half_live) {
    unsigned long long cnt_samples = 0, cnt_inb = 0, cnt_scan = 0;
    bool nan_active = false;
    if (PASS == 1) nan_active = a.nan->pix[0] != kNanEmpty;
    // a wave takes the 64 halves of one GROUP of 32 consecutive triangles at a time; small_grp[g] (a byte k_setup
    // raises) says whether the group holds a small triangle at all, so a band context walks past the 7/8 of a
    // 10^7-triangle frame that lie outside its rows at one byte per 32 triangles
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t u_lo = a.h_lo >> 6, u_hi = (uint32_t)(((unsigned long long)a.h_hi + 63ull) >> 6);
    // (lane l looks up the group of the wave's l-th next turn: one load per 64 turns instead of a dependent load per
    // turn, which is what a band context's walk past its empty groups consisted of)
    for (uint32_t u0 = u_lo + wave_id; u0 < u_hi; u0 += 64u * n_waves) {
        const unsigned long long ul = (unsigned long long)u0 + (unsigned long long)lane * n_waves;
        unsigned long long turns = __ballot(ul < u_hi && a.small_grp[ul < u_hi ? (uint32_t)ul : u_lo] != 0);
      while (turns) {
        const uint32_t turn = (uint32_t)__ffsll((long long)turns) - 1u;
        turns &= turns - 1ull;
        const uint32_t u = u0 + turn * n_waves;
        const uint32_t h = u * 64u + lane;
        if (h < a.h_lo || h >= a.h_hi) continue;
        {   // (a byte per triangle: the other triangles' records are not touched)
            const uint32_t f = a.small_flag[h >> 1];
            if ((f & kFlagPath) != kFlagSmall) continue;
            // near triangles first, the far ones in a second launch: most of their samples then fail the plain test
            // against the plane instead of lowering it with an atomic that a nearer sample lowers again
            if (PASS == 0 && a.phase && ((f & kFlagFar) ? 2u : 1u) != a.phase) continue;
        }
        if (PASS == 1 && a.use_live && !nan_active && half_live[h] == 0) continue;
        const uint32_t tri = h >> 1, half = h & 1u;
        const TriRaster& t = a.rast[tri];
        const uint32_t res = t.res[half];
        const double delta = 1.0 / (double)res;
        const bool counted = h >= a.count_h;
        bool live = false;
        // (batched views: the triangle's view selects the plane slice and the shade blocks)
        const uint32_t vw = a.view_tris ? tri / a.view_tris : 0u;
        const uint32_t pix_base = vw * a.view_pix, blk_base = vw * a.view_blk;
        // depth pass: the minima of the last two pixels the half touched are still in registers (a
        // half of under a pixel alternates between two pixels from one scanline to the next);
        // winner pass: the last two pixels this thread wrote a key to
        uint32_t p0 = 0xFFFFFFFFu, p1 = 0xFFFFFFFFu;
        unsigned long long e0 = 0, e1 = 0;
        auto flush = [&](uint32_t pp, unsigned long long ee) {
            if (PASS == 0 && pp != 0xFFFFFFFFu) {
                // (an `sc1` load: past the CU's L1, whose lines the atomics of the triangles next door leave stale -- a
                // wave's neighbours land on the same pixels within microseconds, and every stale value costs an atomic
                // that changes nothing: 3.91 -> 3.74 ms for the 10^7-triangle frame.  k_raster's samples, a whole scanline
                // apart from their neighbours' in time, gain nothing from it.)
                const unsigned long long cur = PRECHECK ? __hip_atomic_load(&a.depth[pp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
                if (!PRECHECK || ee <= cur) live = true;
                if (!PRECHECK || ee < cur) atomicMin(&a.depth[pp], ee);
#ifdef TR_COUNT_ACT  // (diagnostic: "fragments" = lanes that issue an atomic, "samples" = wave-instructions that do)
                const unsigned long long m = __ballot(!PRECHECK || ee < cur);
                cnt_inb += (!PRECHECK || ee < cur) ? 1 : 0;
                cnt_samples += (m && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) ? 1 : 0;
#endif
            }
        };
        for (uint32_t i = 0; i <= res; ++i) {
            const Scanline sl = make_scanline(t, half, i, delta);
            const double inv = 1.0 / (double)sl.hres;
            if (counted) cnt_scan += 1;
            for (uint32_t j = 0; j <= sl.hres; ++j) {
                const double sp = (double)j * inv;
                const double x = sl.ax + sp * sl.dx;
                const double y = sl.ay + sp * sl.dy;
                const double z = sl.az + sp * sl.dz;
                const int ix = pixel_coord(x, a.W);
                const int iy = pixel_coord(y, a.H);
                const bool inb = ix >= 0 && iy >= a.y_lo && iy < a.y_hi;
#ifndef TR_COUNT_ACT
                if (PASS == 0 && counted) {
                    cnt_samples += 1;
                    cnt_inb += inb ? 1 : 0;
                }
#endif
                if (!inb) continue;
                const double dist = -z;
                const uint32_t pix = pix_base + plane_index((uint32_t)ix, (uint32_t)iy, (uint32_t)a.W, a.tiled != 0);
                const unsigned long long key = ((unsigned long long)tri << kKeyTriShift) | ((unsigned long long)half << kKeyHalfShift) |
                                               ((unsigned long long)i << kKeyIShift) | (unsigned long long)j;
                if (dist != dist) {  // NaN depth: the side table, as in k_raster
                    if (PASS == 0) {
                        live = true;
                        bool placed = false;
                        for (uint32_t s = 0; s < kNanSlots && !placed; ++s) {
                            uint32_t old = atomicCAS(&a.nan->pix[s], kNanEmpty, pix);
                            placed = old == kNanEmpty || old == pix;
                        }
                        if (!placed) atomicOr(a.flags, 1u << 5);
                    } else if (nan_active) {
                        for (uint32_t s = 0; s < kNanSlots; ++s)
                            if (a.nan->pix[s] == pix) atomicMin((unsigned long long*)&a.nan->first_nan[s], key);
                        if (a.need_shade) a.need_shade[tri] = 1;  // (a NaN depth wins outside the key plane)
                    }
                    continue;
                }
                const unsigned long long enc = encode_depth(dist);
                if (PASS == 0) {
                    if (dist == -__builtin_huge_val()) atomicOr(a.flags, 1u << 3);
                    if (pix == p1) {
                        e1 = enc < e1 ? enc : e1;
                    } else if (pix == p0) {
                        e0 = enc < e0 ? enc : e0;
                    } else {  // the older of the two goes to memory
                        flush(p0, e0);
                        p0 = p1; e0 = e1;
                        p1 = pix; e1 = enc;
                    }
                } else {
                    if (nan_active) {
                        for (uint32_t s = 0; s < kNanSlots; ++s)
                            if (a.nan->pix[s] == pix) atomicMin((unsigned long long*)&a.nan->first_any[s], key);
                    }
                    // (samples come in key order: after a hit on a pixel, later ones on it cannot win)
                    if (pix != p0 && pix != p1 && a.depth[pix] == enc) {
                        atomicMin(&a.wkey[pix], key);
                        if (a.need_shade) a.need_shade[tri] = 1;
                        a.blk_touched[2u * (blk_base + (uint32_t)(iy / kShadeBH) * (uint32_t)a.blocks_x + (uint32_t)(ix / kShadeBW))] = 1;
                        p0 = p1;
                        p1 = pix;
                    }
                }
            }
        }
        flush(p0, e0);
        flush(p1, e1);
        if (PASS == 0) half_live[h] = live ? 1 : 0;
      }
    }
    if (PASS == 0) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            cnt_samples += __shfl_down(cnt_samples, d, 64);
            cnt_inb += __shfl_down(cnt_inb, d, 64);
            cnt_scan += __shfl_down(cnt_scan, d, 64);
        }
        if ((threadIdx.x & 63) == 0 && (cnt_samples | cnt_scan)) {
            unsigned long long* slot = a.counters + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) % kCounterSlots) * 8;
            atomicAdd(&slot[0], cnt_samples);
            atomicAdd(&slot[1], cnt_inb);
            atomicAdd(&slot[2], cnt_scan);
        }
    }
}
template __global__ void k_small<0, true>(RasterArgs, uint8_t*);
template __global__ void k_small<0, false>(RasterArgs, uint8_t*);
template __global__ void k_small<1, false>(RasterArgs, uint8_t*);

// ---------------------------------------------------------------- mid-size triangles
// The same two passes for triangles k_setup flagged for this path (small_flag == 2: at most 64 scanlines per half,
// a few hundred samples; they own no slots): one WAVE per triangle half.  Lane i builds scanline i
// (make_scanline: mod.rs:376-379, segment_3d/mod.rs:26-48) and keeps it in the wave's corner of LDS, a wave scan of
// the scanlines' sample counts follows, and then the lanes own SAMPLES of the half in the reference's order --
// consecutive lanes are consecutive samples of a scanline, i.e. neighbouring pixels of a row.  Against the slot
// sequence this needs no prefix over the frame, no records in memory, no search through half_off and no barrier:
// a mesh whose triangles cover tens to hundreds of pixels (BASELINE.json configs[4]: diablo3_pose at 1080p, 9 samples
// to a scanline) fills a 64-slot chunk of the global raster with a few hundred samples and spends its time in the
// chunk's fixed costs.
constexpr int kMidWaves = 4;
template <int PASS, bool PRECHECK>
// (held to 64 VGPRs = 8 waves per SIMD: the depth pass takes 72 by itself; 33.2 -> 34.2 k views/s for the 1080p orbit.  The same
// cap on k_small<0>, 68 VGPRs, spills and changes nothing for the 10^7-triangle frame.)
__global__ __launch_bounds__(64 * kMidWaves) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_mid(RasterArgs a, uint8_t* half_live) {
    __shared__ double sRec[kMidWaves][64][7];      // A, B - A, 1 / hres of the half's scanlines
    __shared__ __align__(16) uint32_t sPre[kMidWaves][68];  // exclusive prefix of their sample counts ([64] = total; rows of 17 x 16 bytes)
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    unsigned long long cnt_samples = 0, cnt_inb = 0, cnt_scan = 0;
    bool nan_active = false;
    if (PASS == 1) nan_active = a.nan->pix[0] != kNanEmpty;
    auto lds_sync = [] {  // (one wave: its LDS operations execute in order; this only pins the compiler)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // A wave takes BLOCKS of 1 << mid_shift consecutive halves: that many lanes look their flags up at once and the
    // wave then visits the halves that are its business one after the other.  (One flag load per half and wave cost a
    // 10^7-triangle frame, few of whose triangles are of this path, 1.9 ms of dependent loads; large blocks leave an
    // orbit batch of 10^5 halves with too few wave-sized pieces of work to fill the GPU: the host sizes them.)
    const uint32_t bs = a.mid_shift, bn = 1u << bs;
    const uint32_t b_lo = a.h_lo >> bs, b_hi = (uint32_t)(((unsigned long long)a.h_hi + bn - 1ull) >> bs);
    for (uint32_t blk = b_lo + wave_id; blk < b_hi; blk += n_waves) {
      bool mine = false;
      if (lane < bn) {
          const uint32_t hl = blk * bn + lane;
          const uint32_t f = hl >= a.h_lo && hl < a.h_hi ? a.small_flag[hl >> 1] : 0u;
          mine = (f & kFlagPath) == kFlagMid && !(PASS == 0 && a.phase && ((f & kFlagFar) ? 2u : 1u) != a.phase);
          if (mine && PASS == 1 && a.use_live && !nan_active) mine = half_live[hl] != 0;
      }
      unsigned long long todo = __ballot(mine);
      while (todo) {
        const uint32_t h = blk * bn + (uint32_t)__builtin_ctzll(todo);  // (the same for every lane)
        todo &= todo - 1ull;
        const uint32_t tri = h >> 1, half = h & 1u;
        // (the half is the same for every lane: its record comes through the scalar cache into scalar registers -- k_setup wrote it
        // in an earlier launch; 24 fewer VGPRs and no vector-memory round trip ahead of the scanlines: 30.4 -> 30.9 k orbit views/s)
        const auto* tp = reinterpret_cast<const __attribute__((address_space(4))) TriRaster*>((uintptr_t)(a.rast + tri));
        const double t_a1[3] = {half ? tp->s2[0] : tp->s0[0], half ? tp->s2[1] : tp->s0[1], half ? tp->s2[2] : tp->s0[2]};
        const double t_a2[3] = {tp->s1[0], tp->s1[1], tp->s1[2]}, t_b2[3] = {tp->sp[0], tp->sp[1], tp->sp[2]};
        const uint32_t res = half ? tp->res[1] : tp->res[0];  // (2 .. 63)
        const bool counted = h >= a.count_h;
        const uint32_t vw = a.view_tris ? tp->pad[0] : 0u;  // (k_setup<.., VIEWS> left the view there: tri / view_tris)
        const uint32_t pix_base = vw * a.view_pix, blk_base = vw * a.view_blk;
        lds_sync();  // the previous half's readers are done with the wave's corner of LDS
        uint32_t cnt = 0;
        if (lane <= res) {
            const Scanline sl = make_scanline_pts(t_a1, t_a2, t_b2, lane, 1.0 / (double)res);
            double* r = sRec[w][lane];
            r[0] = sl.ax; r[1] = sl.ay; r[2] = sl.az; r[3] = sl.dx; r[4] = sl.dy; r[5] = sl.dz;
            r[6] = 1.0 / (double)sl.hres;
            cnt = sl.hres + 1u;  // (k_setup bounds res <= 63 and the half's estimated samples (mid_samples), not hres: a thin half may have hres in
                                 // the thousands; the 32-bit prefix and the key's 18-bit j field hold it)
        }
        // (the wave scan in the VALU's DPP path and the total from a lane read: no LDS round trips ahead of the sample rounds --
        // with `__shfl_up`, six dependent `ds_bpermute`, the orbit's winner pass took 207 instead of 189 us per batch of 32 views)
        const uint32_t inc = wave_inclusive_sum(cnt);
        sPre[w][lane + 1] = inc;
        if (lane == 0) sPre[w][0] = 0;
        lds_sync();
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (PASS == 0 && counted) {
            cnt_scan += lane == 0 ? res + 1u : 0u;
            cnt_samples += lane == 0 ? total : 0u;
        }
        bool live = false;
        // first samples of scanlines 8, 16, .., 56 (the inclusive sums of lanes 7, 15, ..): the same for every lane, scalar registers
        const uint32_t g1 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 7), g2 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 15),
                       g3 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 23), g4 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 31),
                       g5 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 39), g6 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 47),
                       g7 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 55);
        for (uint32_t k0 = 0; k0 < total; k0 += 64u) {
            const uint32_t k = k0 + lane;
            if (k >= total) continue;  // (only in the last round)
            // the last scanline whose first sample is <= k: its group of 8 from the uniform values, the group's 8 prefix entries
            // in two 16-byte LDS reads, a selection tree in registers -- one LDS round trip instead of six dependent ones (k_raster's search)
            const bool c4 = g4 <= k;
            const bool c2 = (c4 ? g6 : g2) <= k;
            const bool c1 = (c4 ? (c2 ? g7 : g5) : (c2 ? g3 : g1)) <= k;
            const uint32_t grp = (c4 ? 4u : 0u) + (c2 ? 2u : 0u) + (c1 ? 1u : 0u);
            const uint4 ea = *reinterpret_cast<const uint4*>(&sPre[w][8u * grp]);
            const uint4 eb = *reinterpret_cast<const uint4*>(&sPre[w][8u * grp + 4u]);
            const bool d4 = eb.x <= k;
            const uint32_t a0 = d4 ? eb.x : ea.x, a1 = d4 ? eb.y : ea.y, a2 = d4 ? eb.z : ea.z, a3 = d4 ? eb.w : ea.w;
            const bool d2 = a2 <= k;
            const uint32_t b0 = d2 ? a2 : a0, b1 = d2 ? a3 : a1;
            const bool d1 = b1 <= k;
            const uint32_t sidx = 8u * grp + (d4 ? 4u : 0u) + (d2 ? 2u : 0u) + (d1 ? 1u : 0u);
            const uint32_t j = k - (d1 ? b1 : b0);
            const double* r = sRec[w][sidx];
            const double sp = (double)j * r[6];
            const double x = r[0] + sp * r[3];
            const double y = r[1] + sp * r[4];
            const double z = r[2] + sp * r[5];
            const int ix = pixel_coord(x, a.W);
            const int iy = pixel_coord(y, a.H);
            const bool inb = ix >= 0 && iy >= a.y_lo && iy < a.y_hi;
            if (PASS == 0 && counted) cnt_inb += inb ? 1 : 0;
            if (!inb) continue;
            const double dist = -z;
            const uint32_t pix = pix_base + plane_index((uint32_t)ix, (uint32_t)iy, (uint32_t)a.W, a.tiled != 0);
            const unsigned long long key = ((unsigned long long)tri << kKeyTriShift) | ((unsigned long long)half << kKeyHalfShift) |
                                           ((unsigned long long)sidx << kKeyIShift) | (unsigned long long)j;
            if (dist != dist) {  // NaN depth: the side table, as in k_raster
                if (PASS == 0) {
                    live = true;
                    bool placed = false;
                    for (uint32_t s = 0; s < kNanSlots && !placed; ++s) {
                        uint32_t old = atomicCAS(&a.nan->pix[s], kNanEmpty, pix);
                        placed = old == kNanEmpty || old == pix;
                    }
                    if (!placed) atomicOr(a.flags, 1u << 5);
                } else if (nan_active) {
                    for (uint32_t s = 0; s < kNanSlots; ++s)
                        if (a.nan->pix[s] == pix) atomicMin((unsigned long long*)&a.nan->first_nan[s], key);
                }
                continue;
            }
            const unsigned long long enc = encode_depth(dist);
            if (PASS == 0) {
                if (dist == -__builtin_huge_val()) atomicOr(a.flags, 1u << 3);
                const unsigned long long cur = PRECHECK ? a.depth[pix] : ~0ull;  // (an `sc1` load as in k_small: no gain here)
                if (!PRECHECK || enc <= cur) live = true;
                if (!PRECHECK || enc < cur) atomicMin(&a.depth[pix], enc);
            } else {
                if (nan_active) {
                    for (uint32_t s = 0; s < kNanSlots; ++s)
                        if (a.nan->pix[s] == pix) atomicMin((unsigned long long*)&a.nan->first_any[s], key);
                }
                if (a.depth[pix] == enc) {
                    atomicMin(&a.wkey[pix], key);
                    a.blk_touched[2u * (blk_base + (uint32_t)(iy / kShadeBH) * (uint32_t)a.blocks_x + (uint32_t)(ix / kShadeBW))] = 1;
                }
            }
        }
        if (PASS == 0) {
            const bool any_live = __ballot(live) != 0ull;
            if (lane == 0) half_live[h] = any_live ? 1 : 0;
        }
      }  // halves of the block
    }
    if (PASS == 0) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            cnt_samples += __shfl_down(cnt_samples, d, 64);
            cnt_inb += __shfl_down(cnt_inb, d, 64);
            cnt_scan += __shfl_down(cnt_scan, d, 64);
        }
        if (lane == 0 && (cnt_samples | cnt_scan)) {
            unsigned long long* slot = a.counters + (size_t)((blockIdx.x * kMidWaves + w) % kCounterSlots) * 8;
            atomicAdd(&slot[0], cnt_samples);
            atomicAdd(&slot[1], cnt_inb);
            atomicAdd(&slot[2], cnt_scan);
        }
    }
}
template __global__ void k_mid<0, true>(RasterArgs, uint8_t*);
template __global__ void k_mid<0, false>(RasterArgs, uint8_t*);
template __global__ void k_mid<1, false>(RasterArgs, uint8_t*);

template __global__ void k_setup<false, false>(SetupArgs);
template __global__ void k_setup<true, false>(SetupArgs);
template __global__ void k_setup<false, true>(SetupArgs);
template __global__ void k_raster<0, true, false, 128, true>(RasterArgs);
template __global__ void k_raster<1, false, false, 128, true>(RasterArgs);
template __global__ void k_raster<0, true, false, 256, true>(RasterArgs);
template __global__ void k_raster<1, false, false, 256, true>(RasterArgs);
template __global__ void k_raster_depth<128, TR_DEPTH_OCC_128>(RasterArgs);
template __global__ void k_raster_depth<256, TR_DEPTH_OCC_256>(RasterArgs);
template __global__ void k_raster_winner<128, TR_WINNER_OCC_128>(RasterArgs);
template __global__ void k_raster_winner<256, TR_WINNER_OCC_256>(RasterArgs);
template __global__ void k_raster<0, true, false, 128>(RasterArgs);
template __global__ void k_raster<1, true, false, 128>(RasterArgs);
template __global__ void k_raster<0, false, false, 128>(RasterArgs);
template __global__ void k_raster<1, false, false, 128>(RasterArgs);
template __global__ void k_raster<0, true, false, 256>(RasterArgs);
template __global__ void k_raster<1, true, false, 256>(RasterArgs);
template __global__ void k_raster<0, false, false, 256>(RasterArgs);
template __global__ void k_raster<1, false, false, 256>(RasterArgs);
#ifdef TR_DIAG  // timing experiments (TR_RASTER_DEBUG) and the per-chunk timeline (TR_RASTER_TRACE)
template __global__ void k_raster<0, true, true, 128>(RasterArgs);
template __global__ void k_raster<1, true, true, 128>(RasterArgs);
template __global__ void k_raster<0, false, true, 128>(RasterArgs);
template __global__ void k_raster<1, false, true, 128>(RasterArgs);
template __global__ void k_raster<0, true, true, 256>(RasterArgs);
template __global__ void k_raster<1, true, true, 256>(RasterArgs);
template __global__ void k_raster<0, false, true, 256>(RasterArgs);
template __global__ void k_raster<1, false, true, 256>(RasterArgs);
#endif

// ================================================================ (4) shade
struct ShadeArgs {
    unsigned long long* wkey;
    unsigned long long* depth;   // global raster: cleaned together with wkey when self_clean
    int self_clean;              // write "none" back over every depth / key this pass consumes
    unsigned long long* key_out; // optional copy of the winner keys (TR_OUT_KEY)
    const TriRaster* rast;
    const TriShade* shade;
    const DrawDev* draws;
    const NanTable* nan;        // global-atomic raster only (nullptr in tile mode)
    const uint32_t* tile_off;   // tile mode: a tile with an empty list has no keys in wkey
    int tiles_x;
    int tw_shift, th_shift;  // tile mode: log2 of the tile size
    int W, H;
    uint32_t out_mask;
    uint32_t* rgba8;    // row-major, packed r | g<<8 | b<<16 | a<<24
    double* z_out;      // optional
    uint32_t* tri_out;  // optional
    double* rgba_f64;   // optional, 4 per pixel
    uint32_t* flags;
    unsigned long long* cov_counters;  // kCounterSlots x 8, [0] of each slot used
    // two bytes per shade block, side by side so that one scalar load brings both:
    // [2 b] the winner pass wrote a key into the block this frame, [2 b + 1] rgba8 holds something other than the background
    uint8_t* blk_flags;
    int skip_clean;        // blocks neither touched nor painted are left alone (RGBA8-only resolve, global raster)
    DrawDev draw0;         // copy of draws[0]
    const DrawExt* ext;    // k_shade<true>: per draw
    unsigned long long* trace;  // k_shade<.., true> (TR_SHADE_TRACE, tools/shade_trace.py): 4 words per wave
    // batched views (k_shade<false, false, true>): the grid is the views stacked in y; view v owns block rows
    // [v * view_blk_rows, (v + 1) * view_blk_rows), plane pixels from v * view_pix and draw-table entry v
    uint32_t view_blk_rows, view_pix;
    uint32_t view_meshes;  // batched views: draws (meshes) per view; 1: draw-table entry v is view v's
    uint32_t blk_y0;  // first block row of the launch (band contexts, boxed launches; 0 otherwise)
    uint32_t blk_x0;  // first block column of the launch (boxed launches)
    uint32_t launch_rows;  // block rows (per view) of this launch; a workgroup row beyond them is the one that takes block (0, 0)
    uint32_t tiled;        // layout of wkey / depth this frame (plane_index)
    uint32_t blocks_x;  // blocks per row of the FRAME (the launch's gridDim.x may be smaller)
};

// helpers/mod.rs:253-278 on an RGBA8 image (texel value as f64 is exact).  Channels r,g,b.
// The four texels and their weights (the weights use f64 x1, x2; the indices their saturated casts).
struct Taps {
    uint32_t c00, c10, c01, c11;
    double w00, w10, w01, w11;
};
__device__ __forceinline__ Taps sample_taps(const uint32_t* __restrict__ img, int W, int H, double x, double y, uint32_t* flags) {
    double x1 = floor(x), y1 = floor(y);
    // `x1 as usize`: NaN / negative -> 0, saturating above (any index >= 2^32 - 1 is out of range either way)
    uint32_t ix1 = f64_as_u32_sat(x1);
    uint32_t iy1 = f64_as_u32_sat(y1);
    uint32_t ix2 = ix1 == 0xFFFFFFFFu ? ix1 : ix1 + 1u;
    uint32_t iy2 = iy1 == 0xFFFFFFFFu ? iy1 : iy1 + 1u;
    if (ix2 > (uint32_t)(W - 1)) ix2 = (uint32_t)(W - 1);
    if (iy2 > (uint32_t)(H - 1)) iy2 = (uint32_t)(H - 1);
    double x2 = (double)ix2, y2 = (double)iy2;
    if (ix1 >= (uint32_t)W || iy1 >= (uint32_t)H) {  // the reference panics here
        atomicOr(flags, 1u << 2);
        if (ix1 >= (uint32_t)W) ix1 = (uint32_t)(W - 1);
        if (iy1 >= (uint32_t)H) iy1 = (uint32_t)(H - 1);
    }
    Taps t;
    if (W >= 2) {
        // ix2 is ix1 + 1 or, at the right edge, ix1 itself: the two texels of a row come as ONE 8-byte load from
        // column min(ix1, W - 2) (the texture unit spends its cycles per wave-instruction, not per byte: PMC)
        struct __attribute__((packed, aligned(4))) Pair { uint32_t lo, hi; };
        const uint32_t bx = ix1 < (uint32_t)(W - 1) ? ix1 : (uint32_t)(W - 2);
        // (32-bit byte offsets: a texture is at most 32768 x 32768 texels = 2^32 bytes, tr_texture_create; the base is the same in
        // every lane as a rule, so the load takes it from scalar registers and the address costs no 64-bit arithmetic)
        const char* base = reinterpret_cast<const char*>(img);
        const Pair r1 = *reinterpret_cast<const Pair*>(base + (iy1 * (uint32_t)W + bx) * 4u);
        const Pair r2 = *reinterpret_cast<const Pair*>(base + (iy2 * (uint32_t)W + bx) * 4u);
        t.c00 = ix1 == bx ? r1.lo : r1.hi;
        t.c10 = ix2 == bx ? r1.lo : r1.hi;
        t.c01 = ix1 == bx ? r2.lo : r2.hi;
        t.c11 = ix2 == bx ? r2.lo : r2.hi;
    } else {
        t.c00 = img[iy1 * (uint32_t)W + ix1];
        t.c10 = img[iy1 * (uint32_t)W + ix2];
        t.c01 = img[iy2 * (uint32_t)W + ix1];
        t.c11 = img[iy2 * (uint32_t)W + ix2];
    }
    t.w00 = (x2 - x) * (y2 - y);
    t.w10 = (x - x1) * (y2 - y);
    t.w01 = (x2 - x) * (y - y1);
    t.w11 = (x - x1) * (y - y1);
    return t;
}
// Two images of the same size sampled at the same point (a draw's normal map and its diffuse map, as a rule): the indices,
// the weights and the address arithmetic are the first image's, and the second image's texels are requested in the same
// breath, so that their round trip passes behind the arithmetic on the first's (4K african_head: shade pass 30.8 -> 28.8 us;
// no more registers than before, since the second image's indices and weights are not computed at all).  `cb` gets the second
// image's texels only: its weights are `ta`'s.  W >= 2 (the caller checks).
__device__ __forceinline__ void sample_taps_pair(const uint32_t* __restrict__ img_a, const uint32_t* __restrict__ img_b, int W, int H,
                                                 double x, double y, uint32_t* flags, Taps& ta, uint32_t cb[4]) {
    double x1 = floor(x), y1 = floor(y);
    uint32_t ix1 = f64_as_u32_sat(x1);
    uint32_t iy1 = f64_as_u32_sat(y1);
    uint32_t ix2 = ix1 == 0xFFFFFFFFu ? ix1 : ix1 + 1u;
    uint32_t iy2 = iy1 == 0xFFFFFFFFu ? iy1 : iy1 + 1u;
    if (ix2 > (uint32_t)(W - 1)) ix2 = (uint32_t)(W - 1);
    if (iy2 > (uint32_t)(H - 1)) iy2 = (uint32_t)(H - 1);
    double x2 = (double)ix2, y2 = (double)iy2;
    if (ix1 >= (uint32_t)W || iy1 >= (uint32_t)H) {  // the reference panics here (once per image: the flag is the same bit)
        atomicOr(flags, 1u << 2);
        if (ix1 >= (uint32_t)W) ix1 = (uint32_t)(W - 1);
        if (iy1 >= (uint32_t)H) iy1 = (uint32_t)(H - 1);
    }
    struct __attribute__((packed, aligned(4))) Pair { uint32_t lo, hi; };
    const uint32_t bx = ix1 < (uint32_t)(W - 1) ? ix1 : (uint32_t)(W - 2);  // (W >= 2: the caller checked)
    // (32-bit byte offsets, as in sample_taps)
    const uint32_t o1 = (iy1 * (uint32_t)W + bx) * 4u, o2 = (iy2 * (uint32_t)W + bx) * 4u;
    const char* base_a = reinterpret_cast<const char*>(img_a);
    const char* base_b = reinterpret_cast<const char*>(img_b);
    const Pair a1 = *reinterpret_cast<const Pair*>(base_a + o1);
    const Pair a2 = *reinterpret_cast<const Pair*>(base_a + o2);
    const Pair b1 = *reinterpret_cast<const Pair*>(base_b + o1);
    const Pair b2 = *reinterpret_cast<const Pair*>(base_b + o2);
    const bool l1 = ix1 == bx, l2 = ix2 == bx;
    ta.c00 = l1 ? a1.lo : a1.hi;
    ta.c10 = l2 ? a1.lo : a1.hi;
    ta.c01 = l1 ? a2.lo : a2.hi;
    ta.c11 = l2 ? a2.lo : a2.hi;
    cb[0] = l1 ? b1.lo : b1.hi;
    cb[1] = l2 ? b1.lo : b1.hi;
    cb[2] = l1 ? b2.lo : b2.hi;
    cb[3] = l2 ? b2.lo : b2.hi;
    ta.w00 = (x2 - x) * (y2 - y);
    ta.w10 = (x - x1) * (y2 - y);
    ta.w01 = (x2 - x) * (y - y1);
    ta.w11 = (x - x1) * (y - y1);
}
__device__ __forceinline__ void blend_taps(const Taps& t, double out[3]) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        double a = (double)((t.c00 >> (8 * ch)) & 0xFFu);
        double b = (double)((t.c10 >> (8 * ch)) & 0xFFu);
        double c = (double)((t.c01 >> (8 * ch)) & 0xFFu);
        double d = (double)((t.c11 >> (8 * ch)) & 0xFFu);
        out[ch] = a * t.w00 + b * t.w10 + c * t.w01 + d * t.w11;
    }
}
// The same sum with fused multiply-adds (k_shade's FAST colour arithmetic: within an ulp or two of blend_taps).
__device__ __forceinline__ void blend_taps_fused(const Taps& t, double out[3]) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        double a = (double)((t.c00 >> (8 * ch)) & 0xFFu);
        double b = (double)((t.c10 >> (8 * ch)) & 0xFFu);
        double c = (double)((t.c01 >> (8 * ch)) & 0xFFu);
        double d = (double)((t.c11 >> (8 * ch)) & 0xFFu);
        out[ch] = __builtin_fma(d, t.w11, __builtin_fma(c, t.w01, __builtin_fma(b, t.w10, a * t.w00)));
    }
}
__device__ __forceinline__ void sample_rgb(const uint32_t* __restrict__ img, int W, int H, double x,
                                           double y, double out[3], uint32_t* flags) {
    const Taps t = sample_taps(img, W, H, x, y, flags);
    blend_taps(t, out);
}
// v / |v| with the hardware's reciprocal square root and one Newton step (relative error ~1e-14) instead of a
// correctly rounded square root and three correctly rounded divisions.  A zero or infinite vector gives NaNs, as the
// exact form does.
__device__ __forceinline__ void normalize3_fast(double& x, double& y, double& z) {
    const double s = __builtin_fma(z, z, __builtin_fma(y, y, x * x));
    double r = __builtin_amdgcn_rsq(s);
    const double e = __builtin_fma(-(0.5 * s) * r, r, 0.5);
    r = __builtin_fma(r, e, r);
    x *= r; y *= r; z *= r;
}

// Loads from a wave-uniform address through the scalar cache (constant address space).  The data
// was written by an earlier kernel; the only writer in this kernel is the reader's own workgroup.
__device__ __forceinline__ uint32_t uniform_u32(const uint32_t* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>((uintptr_t)p);
}
__device__ __forceinline__ uint32_t uniform_u16(const uint8_t* p) {  // p 2-byte aligned
    const uintptr_t addr = (uintptr_t)p;
    const uint32_t w = *reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(addr & ~(uintptr_t)3);
    return (w >> ((uint32_t)(addr & 2u) * 8u)) & 0xFFFFu;
}

__device__ __forceinline__ DrawDev uniform_drawdev(const DrawDev* p) {  // (p is the same in every lane of the wave)
    static_assert(sizeof(DrawDev) % 8 == 0, "copied as 64-bit words");
    union { DrawDev d; unsigned long long w[sizeof(DrawDev) / 8]; } u;
    const __attribute__((address_space(4))) unsigned long long* q =
        reinterpret_cast<const __attribute__((address_space(4))) unsigned long long*>((uintptr_t)p);
#pragma unroll
    for (size_t i = 0; i < sizeof(DrawDev) / 8; ++i) u.w[i] = q[i];
    return u.d;
}



// x^e of the extension shaders' specular term.  A specular MAP holds bytes, so the exponent is almost
// always an integer in [0, 255]: eight squarings instead of the device library's pow (a few hundred
// instructions); within a few ulp of it, i.e. far inside the +-1 LSB these shaders are checked to.
__device__ __forceinline__ double pow_exponent(double x, double e) {
    if (e >= 0.0 && e <= 255.0 && e == floor(e)) {
        uint32_t n = (uint32_t)e;
        double r = 1.0, p = x;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            r = (n & 1u) ? r * p : r;
            p = p * p;
            n >>= 1;
        }
        return r;
    }
    return pow(x, e);
}

// EXT = false: the reference's shader, the tuned path of every parity test and of the benchmark.
// EXT = true: a frame in which some draw asked for an extension shader (DrawExt); draws without one
// still get the reference's formula, operation for operation.
// VIEWS: a batch of views of one mesh, stacked in y (tr_render_views); a wave's pixels belong to one view, whose
// fragment constants come through the scalar cache.
// FAST: the colour arithmetic AFTER the barycentrics and the texture coordinates may fuse multiply-adds and use the
// hardware's reciprocal square root (still f64): the presented colour stays within +-1 LSB of the reference's (the
// bar BASELINE.json sets for RGB) instead of equal to it.  The winning sample's position, the barycentrics, u and v --
// everything a texel ADDRESS or the depth plane depends on -- keep the reference's operations: sample_nd_img is
// discontinuous where x reaches W - 1 exactly (helpers/mod.rs:255-270), and a corner sample has u exactly its vertex's.
// STAGE: the records of the wave's distinct triangles (up to kStageTris of them: a 32 x 2 block of the reference's
// scenes lies in two or three) are copied to LDS with three wave-wide 16-byte loads and read from there, instead of
// ~24 loads per lane from global memory: the pass is bound by the texture unit's rate of wave-instructions (16 cycles
// each whatever the lanes' addresses: TCP_TOTAL_CACHE_ACCESSES = 18.4 M per 4K launch = 72 k cycles per CU of a 92 k-cycle
// launch).  Waves with more distinct triangles (meshes of triangles of a pixel) read them directly as before.
constexpr int kStageTris = 8;
constexpr int kStageStride = 512;  // bytes per staged triangle: TriRaster at 0, TriShade at 256 (its alignment)
template <bool EXT, bool TRACE, bool VIEWS, bool FAST, bool STAGE>
__device__ __forceinline__ void shade_body(const ShadeArgs& a) {
    __shared__ __align__(256) unsigned char sStage[STAGE ? kStageTris * kStageStride : 256];
    __shared__ uint32_t sStageTri[kStageTris];
    unsigned long long tr_t0 = 0;
    if (TRACE) tr_t0 = wall_clock64();
    auto trace_end = [&](uint32_t covered) {
        if (TRACE && (threadIdx.x & 63) == 0) {
            unsigned long long* t = a.trace + 4 * (size_t)((blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6));
            t[0] = tr_t0; t[1] = wall_clock64();
            t[2] = covered;
            t[3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
        }
    };
    // (a boxed launch that does not start at the frame's first block has one more row of workgroups, of which the first takes
    // block (0, 0): samples with NaN coordinates -- a triangle whose corners share one y -- land on pixel (0, 0) wherever the
    // frame's screen box lies, and what they leave there has to be painted, and repainted when it is gone)
    const bool extra = blockIdx.y >= a.launch_rows;
    if (extra && blockIdx.x != 0u) return;
    const uint32_t bx = extra ? 0u : blockIdx.x + a.blk_x0;
    const int x = (int)bx * kShadeBW + (threadIdx.x & (kShadeBW - 1));
    // (a context that owns a band of rows launches the block rows of its band only: blk_y0 is the first of them)
    // (a batch of views launches rows [blk_y0, blk_y0 + launch_rows) of every view, one view after the other)
    // (the view is the grid's z: rows x views may pass the 65 535 a grid dimension holds)
    const uint32_t view = VIEWS ? blockIdx.z : 0u;
    const uint32_t row = extra ? 0u : a.blk_y0 + blockIdx.y;
    const uint32_t by = (VIEWS ? view * a.view_blk_rows : 0u) + row;
    const int y = (int)row * kShadeBH + (threadIdx.x / kShadeBW);
    const bool live = x < a.W && y < a.H;
    const uint32_t pix = (VIEWS ? view * a.view_pix : 0u) + (uint32_t)y * (uint32_t)a.W + (uint32_t)x;
    // (the depth / key planes may be tiled this frame; the output planes never are)
    const uint32_t ppix = (VIEWS ? view * a.view_pix : 0u) + plane_index((uint32_t)x, (uint32_t)y, (uint32_t)a.W, a.tiled != 0);
    const uint32_t blk = by * a.blocks_x + bx;
    // (one scalar load for the whole kernel: the frame holds a NaN-depth sample -- a degenerate triangle's)
    const bool nan_frame = a.nan && uniform_u32(&a.nan->pix[0]) != kNanEmpty;
    if (a.skip_clean) {
        // Nothing landed in this block: its keys are all "none".  If the previous frame left it
        // at the background colour too there is nothing to do; otherwise repaint the background.
        // (A NaN-depth winner is written outside the winner pass: no skipping in such a frame.)
        // (scalar loads: a wave of an untouched block must not queue behind the vector memory traffic
        // of the waves that shade; measured, such waves lived 1.5 us and held half of the wave slots)
        // (both loads issued before either is needed: one round trip)
        const uint32_t f_both = uniform_u16(a.blk_flags + 2u * blk);
        const uint32_t f_touched = f_both & 0xFFu, f_painted = f_both >> 8;
        const bool touched = f_touched != 0 || nan_frame;
        const bool painted = f_painted != 0;
        if (!touched && !painted) { trace_end(0xFFFFFFFFu); return; }  // (nobody writes the flags on this path)
        // (the workgroup is one wave: every lane has read the flags before lane 0 rewrites them)
        if (!touched) {
            if (live) a.rgba8[pix] = 0xFF000000u;  // BLACK, helpers/mod.rs:22
            if (threadIdx.x == 0) a.blk_flags[2u * blk + 1u] = 0;
            trace_end(0xFFFFFFFEu);
            return;
        }
    }
    bool listed = live;
    if (a.tile_off && live) {
        int tile = (y >> a.th_shift) * a.tiles_x + (x >> a.tw_shift);
        listed = a.tile_off[tile + 1] > a.tile_off[tile];
    }
    // (requesting the key before the block flags are back, so that a covered wave saves a round trip, makes the idle waves
    // of the boxed launch wait for a load they do not need: 34.3 against 32.4 us)
    unsigned long long key = listed ? a.wkey[ppix] : kKeyNone;
    if (a.self_clean && key != kKeyNone) {  // leave the planes cleared for the next frame
        a.wkey[ppix] = kKeyNone;
        if (a.depth) a.depth[ppix] = kKeyNone;
    }
    if (nan_frame && live) {
        for (uint32_t s = 0; s < kNanSlots; ++s)
            if (a.nan->pix[s] == ppix && a.nan->first_nan[s] < a.nan->first_any[s]) key = a.nan->first_nan[s];
    }
    bool has = key != kKeyNone;
    unsigned long long nwin = __popcll(__ballot(has));
    if ((threadIdx.x & 63) == 0 && nwin)
        atomicAdd(&a.cov_counters[(size_t)((blk * 4 + (threadIdx.x >> 6)) % kCounterSlots) * 8], nwin);
    {
        static_assert(kShadeBW * kShadeBH == 64, "the shade workgroup is one wave: votes instead of barriers");
        const bool any = nwin != 0ull;
        if (threadIdx.x == 0) {
            if ((a.out_mask & 1u) && a.self_clean) {  // (both bytes in one store)
                *reinterpret_cast<uint16_t*>(a.blk_flags + 2u * blk) = any ? (uint16_t)0x0100u : (uint16_t)0u;
            } else {
                if (a.out_mask & 1u) a.blk_flags[2u * blk + 1u] = any ? 1 : 0;  // (no colour plane written: what it holds is unchanged)
                if (a.self_clean) a.blk_flags[2u * blk] = 0;
            }
        }
    }
    uint32_t tri = has ? (uint32_t)(key >> kKeyTriShift) : 0xFFFFFFFFu;
    uint32_t slot = 0;
    bool staged = false;
    if (STAGE && nwin) {
        // the wave's distinct triangles, in the order of their first lanes
        // (a test on runs of equal neighbours in front of the loop, so that waves over triangles of a pixel -- a
        // 10^7-triangle mesh -- do not walk it to its limit for nothing, changed nothing there and cost the 4K frame 1.4 us)
        unsigned long long todo = __ballot(has);
        uint32_t n_dist = 0;
        while (todo && n_dist < (uint32_t)kStageTris) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tri, leader);
            const bool mine = has && tri == t;
            if (mine) slot = n_dist;
            if ((int)threadIdx.x == leader) sStageTri[n_dist] = t;
            todo &= ~__ballot(mine);
            ++n_dist;
        }
        staged = todo == 0ull && n_dist != 0u;  // (uniform)
        if (staged) {
            __syncthreads();  // (one wave: orders the list's writes before its reads)
            // 16-byte chunks per triangle: the first 7 of TriRaster (points, split point, res) and the first 15 of TriShade
            // (all but its last 16 bytes, which are alignment)
            constexpr uint32_t kRastChunks = 7u, kChunks = 22u;
            const uint32_t n_chunks = n_dist * kChunks;
#pragma unroll
            for (uint32_t c0 = 0; c0 < (uint32_t)kStageTris * kChunks; c0 += 64u) {
                const uint32_t c = c0 + threadIdx.x;
                if (c < n_chunks) {
                    const uint32_t q = c / kChunks, part = c - q * kChunks;
                    const uint32_t t = sStageTri[q];
                    const uint4 v = part < kRastChunks ? reinterpret_cast<const uint4*>(a.rast + t)[part]
                                                       : reinterpret_cast<const uint4*>(a.shade + t)[part - kRastChunks];
                    reinterpret_cast<uint4*>(sStage + q * kStageStride)[part < kRastChunks ? part : part - kRastChunks + 16u] = v;
                }
            }
            __syncthreads();
        }
    }
    if (!live) { trace_end(0u); return; }
    double col[4] = {0.0, 0.0, 0.0, 255.0};  // BLACK, helpers/mod.rs:22
    double zval = -__builtin_huge_val();
    if (has) {
        uint32_t half = (uint32_t)(key >> kKeyHalfShift) & 1u;
        uint32_t i = (uint32_t)(key >> kKeyIShift) & 0x1FFFFu;
        uint32_t j = (uint32_t)key & 0x3FFFFu;
      auto shade_px = [&](const TriRaster& r, const TriShade& s) __attribute__((always_inline)) {
        Scanline sl = make_scanline(r, half, i, s.inv_res[half]);  // (delta = 1 / res from the record: one division less per pixel)
        double sp = (double)j * (1.0 / (double)sl.hres);
        double px = sl.ax + sp * sl.dx;
        double py = sl.ay + sp * sl.dy;
        double pz = sl.az + sp * sl.dz;
        zval = -pz;
        // the draw is the same for the whole wave almost always (always, in a one-draw frame): its
        // constants then come through the scalar cache into SGPRs instead of 112 B per lane through L1
        // helpers/mod.rs:326-333 barycentrics w.r.t. the UNSORTED viewport triangle
        double det = ((s.uy[1] - s.uy[2]) * (s.ux[0] - s.ux[2])) + ((s.ux[2] - s.ux[1]) * (s.uy[0] - s.uy[2]));
        double l1 = (((s.uy[1] - s.uy[2]) * (px - s.ux[2])) + ((s.ux[2] - s.ux[1]) * (py - s.uy[2]))) / det;
        double l2 = (((s.uy[2] - s.uy[0]) * (px - s.ux[2])) + ((s.ux[0] - s.ux[2]) * (py - s.uy[2]))) / det;
        double l3 = 1.0 - l1 - l2;
        // mod.rs:499-504
        double nx, ny, nz;
        if (FAST) {
            nx = __builtin_fma(l3, s.n[2][0], __builtin_fma(l2, s.n[1][0], l1 * s.n[0][0]));
            ny = __builtin_fma(l3, s.n[2][1], __builtin_fma(l2, s.n[1][1], l1 * s.n[0][1]));
            nz = __builtin_fma(l3, s.n[2][2], __builtin_fma(l2, s.n[1][2], l1 * s.n[0][2]));
            normalize3_fast(nx, ny, nz);
        } else {
            nx = l1 * s.n[0][0] + l2 * s.n[1][0] + l3 * s.n[2][0];
            ny = l1 * s.n[0][1] + l2 * s.n[1][1] + l3 * s.n[2][1];
            nz = l1 * s.n[0][2] + l2 * s.n[1][2] + l3 * s.n[2][2];
            normalize3(nx, ny, nz);
        }
        bool has_uv = s.flags & 1u, has_tb = s.flags & 2u;
        double u = 0.0, v = 0.0;
        if (has_uv) {  // mod.rs:510-514
            u = l1 * s.uv[0][0] + l2 * s.uv[1][0] + l3 * s.uv[2][0];
            v = l1 * s.uv[0][1] + l2 * s.uv[1][1] + l3 * s.uv[2][1];
        }
        // Waves whose pixels all belong to the frame's first draw (all of them in a one-draw frame)
        // take its constants from the kernel arguments, i.e. from SGPRs, instead of 112 B per lane
        // through the vector L1, which is the busiest unit of this kernel.
        auto finish = [&](const DrawDev& dp) __attribute__((always_inline)) {
            if (FAST && !EXT) {
                // main.rs:183-265 with fused arithmetic
                const bool use_nm = has_uv && has_tb && dp.use_nm;
                Taps tn, tt;
                // (a normal map and a diffuse map of one size -- the reference's assets all are: one set of indices and weights,
                // all four 8-byte texel loads in flight at once)
                // (not in the batched-view form: held to 72 VGPRs it spills 60 instead of 20 bytes a lane with the four loads in
                // flight, 34.6 against 35.3 k views/s for the 1080p orbit)
                const bool paired = !VIEWS && use_nm && dp.nw == dp.tw && dp.nh == dp.th && dp.nw >= 2;
                uint32_t ct[4] = {0u, 0u, 0u, 0u};
                if (paired) sample_taps_pair(dp.nm, dp.tex, dp.nw, dp.nh, u * (double)(dp.nw - 1), v * (double)(dp.nh - 1), a.flags, tn, ct);
                else if (use_nm) tn = sample_taps(dp.nm, dp.nw, dp.nh, u * (double)(dp.nw - 1), v * (double)(dp.nh - 1), a.flags);
                if (use_nm) {
                    double rgb[3];
                    blend_taps_fused(tn, rgb);
                    constexpr double k = 2.0 / 255.0;
                    const double mx = __builtin_fma(rgb[0], k, -1.0), my = __builtin_fma(rgb[1], k, -1.0), mz = __builtin_fma(rgb[2], k, -1.0);
                    double rx = __builtin_fma(nx, mz, __builtin_fma(s.b[0], my, s.t[0] * mx));
                    double ry = __builtin_fma(ny, mz, __builtin_fma(s.b[1], my, s.t[1] * mx));
                    double rz = __builtin_fma(nz, mz, __builtin_fma(s.b[2], my, s.t[2] * mx));
                    normalize3_fast(rx, ry, rz);
                    nx = rx; ny = ry; nz = rz;
                }
                double albedo[3];
                if (has_uv) {
                    // (maps of different sizes: requested here, not together with the normal map's texels -- a second set of indices
                    // and weights alive across the normal arithmetic took 84 instead of 70 VGPRs, 39.1 against 37.8 us at 4K)
                    if (paired) {
                        tt = tn;
                        tt.c00 = ct[0]; tt.c10 = ct[1]; tt.c01 = ct[2]; tt.c11 = ct[3];
                    } else
                        tt = sample_taps(dp.tex, dp.tw, dp.th, u * (double)(dp.tw - 1), v * (double)(dp.th - 1), a.flags);
                    blend_taps_fused(tt, albedo);
                } else {
    #pragma unroll
                    for (int c = 0; c < 3; ++c) albedo[c] = l1 * dp.vcol[c] + l2 * dp.vcol[c] + l3 * dp.vcol[c];
                }
                const double d = fmax(__builtin_fma(nz, dp.light[2], __builtin_fma(ny, dp.light[1], nx * dp.light[0])), 0.0);
                const double di = d * dp.intensity;
    #pragma unroll
                for (int c = 0; c < 3; ++c) col[c] = fmin(__builtin_fma(di, albedo[c], dp.ambient), 255.0);
                col[3] = 255.0;
                return;
            }
            // main.rs:183-218
            if (has_uv && has_tb && dp.use_nm) {
                double rgb[3];
                sample_rgb(dp.nm, dp.nw, dp.nh, u * (double)(dp.nw - 1), v * (double)(dp.nh - 1), rgb, a.flags);
                double mx = (div255(rgb[0]) * 2.0) - 1.0;
                double my = (div255(rgb[1]) * 2.0) - 1.0;
                double mz = (div255(rgb[2]) * 2.0) - 1.0;
                double rx = (s.t[0] * mx + s.b[0] * my) + nx * mz;
                double ry = (s.t[1] * mx + s.b[1] * my) + ny * mz;
                double rz = (s.t[2] * mx + s.b[2] * my) + nz * mz;
                normalize3(rx, ry, rz);
                nx = rx; ny = ry; nz = rz;
            }
            // main.rs:221-247
            double albedo[3];
            if (has_uv) {
                sample_rgb(dp.tex, dp.tw, dp.th, u * (double)(dp.tw - 1), v * (double)(dp.th - 1), albedo, a.flags);
            } else {  // mod.rs:518-525 colour interpolation
    #pragma unroll
                for (int c = 0; c < 3; ++c) albedo[c] = l1 * dp.vcol[c] + l2 * dp.vcol[c] + l3 * dp.vcol[c];
            }
            // main.rs:248-265
            double d = fmax(dot3(nx, ny, nz, dp.light[0], dp.light[1], dp.light[2]), 0.0);
            if (EXT) {
                const DrawExt& e = a.ext[s.draw];
                if (e.model != 0 || e.shadow) {
                    if (e.model == 1) {  // Gouraud: diffuse term at the vertices, interpolated
                        double g[3];
    #pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            double vx = s.n[c][0], vy = s.n[c][1], vz = s.n[c][2];
                            normalize3(vx, vy, vz);
                            g[c] = fmax(dot3(vx, vy, vz, dp.light[0], dp.light[1], dp.light[2]), 0.0);
                        }
                        d = l1 * g[0] + l2 * g[1] + l3 * g[2];
                    }
                    double spec = 0.0;
                    if (e.model == 2) {  // Phong: reflected light direction against the view axis
                        const double ndl = dot3(nx, ny, nz, dp.light[0], dp.light[1], dp.light[2]);
                        const double rz = (2.0 * ndl) * nz - dp.light[2];
                        double ex = e.shininess;
                        if (e.spec && has_uv) {
                            double srgb[3];
                            sample_rgb(e.spec, e.sw, e.sh, u * (double)(e.sw - 1), v * (double)(e.sh - 1), srgb, a.flags);
                            ex = srgb[0];
                        }
                        spec = pow_exponent(fmax(rz, 0.0), ex);
                    }
                    double lit = 1.0;
                    if (e.shadow) {
                        double q[4];
                        mat4_mul_vec4(e.to_light, px, py, pz, 1.0, q);
                        const double xl = q[0] / q[3], yl = q[1] / q[3], zl = q[2] / q[3];
                        const int ix = pixel_coord(xl, e.mw), iy = pixel_coord(yl, e.mh);
                        if (ix >= 0 && iy >= 0) {
                            const double zm = e.shadow[(size_t)iy * (size_t)e.mw + (size_t)ix];
                            if (zm != -__builtin_huge_val() && -zl > zm + e.bias) lit = e.dim;
                        }
                    }
                    const double term = lit * (d * dp.intensity + e.strength * spec);
    #pragma unroll
                    for (int c = 0; c < 3; ++c) col[c] = fmin(dp.ambient + (term * albedo[c]), 255.0);
                    col[3] = 255.0;
                    return;
                }
            }
    #pragma unroll
            for (int c = 0; c < 3; ++c) col[c] = fmin(dp.ambient + (d * dp.intensity * albedo[c]), 255.0);
            col[3] = 255.0;
        };
        // (a resolve for depth only, the snapshot of a shadow map: no shading, no colour plane)
        if (!(a.out_mask & 1u)) {
        } else if (VIEWS && a.view_meshes <= 1u) {
            finish(uniform_drawdev(a.draws + view));  // (the batch's draws are its views, in order)
        } else if (VIEWS) {
            // several meshes per view: a wave's pixels belong to one view, i.e. to at most view_meshes draws -- one round per
            // draw present in the wave, its constants through the scalar cache as above
            unsigned long long todo = __ballot(true);
            while (todo) {
                const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)s.draw, __builtin_ctzll(todo));
                const bool mine = s.draw == d;
                if (mine) finish(uniform_drawdev(a.draws + d));
                todo &= ~__ballot(mine);
            }
        } else if (__all(s.draw == 0u)) finish(a.draw0);  // kernel argument: scalar registers
        else finish(a.draws[s.draw]);
      };
        if (STAGE && staged)
            shade_px(*reinterpret_cast<const TriRaster*>(sStage + slot * kStageStride),
                     *reinterpret_cast<const TriShade*>(sStage + slot * kStageStride + 256));
        else
            shade_px(a.rast[tri], a.shade[tri]);
    }
    if (a.out_mask & 1u)
        a.rgba8[pix] = f64_as_u8(col[0]) | (f64_as_u8(col[1]) << 8) | (f64_as_u8(col[2]) << 16) |
                       (f64_as_u8(col[3]) << 24);
    if (a.out_mask & 16u) a.key_out[pix] = key;
    if (a.out_mask & 2u) a.z_out[pix] = zval;
    if (a.out_mask & 4u) a.tri_out[pix] = tri;
    if (a.out_mask & 8u) {
        double4 o = make_double4(col[0], col[1], col[2], col[3]);
        reinterpret_cast<double4*>(a.rgba_f64)[pix] = o;
    }
    if (TRACE) trace_end((uint32_t)__popcll(__ballot(has)));
}

// The exact kernels (records read per lane: staging them gains nothing where instruction issue bounds the pass --
// 40.5 against 39.7 us) ...
template <bool EXT, bool TRACE = false, bool VIEWS = false>
__global__ __launch_bounds__(kShadeBW * kShadeBH) void k_shade(ShadeArgs a) {
    shade_body<EXT, TRACE, VIEWS, false, false>(a);
}
// ... and the fused one, held to 64 VGPRs so that 8 waves fit a SIMD (no spills; 80 VGPRs and 6 waves otherwise:
// 32.0 against 33.9 us at 4K)
#ifndef TR_SHADE_VIEWS_OCC
#define TR_SHADE_VIEWS_OCC 7  // (the views form spills 44 bytes a lane at 64 VGPRs: 34.3 -> 35.3 k views/s for the 1080p orbit at 72)
#endif
template <bool VIEWS>
__global__ __launch_bounds__(kShadeBW * kShadeBH) __attribute__((amdgpu_waves_per_eu(VIEWS ? TR_SHADE_VIEWS_OCC : 8, VIEWS ? TR_SHADE_VIEWS_OCC : 8))) void k_shade_fused(ShadeArgs a) {
    shade_body<false, false, VIEWS, true, true>(a);
}
template __global__ void k_shade<false, false>(ShadeArgs);
template __global__ void k_shade<true, false>(ShadeArgs);
#ifdef TR_DIAG  // per-wave timeline (TR_SHADE_TRACE)
template __global__ void k_shade<false, true>(ShadeArgs);
#endif
template __global__ void k_shade<false, false, true>(ShadeArgs);
template __global__ void k_shade_fused<false>(ShadeArgs);
template __global__ void k_shade_fused<true>(ShadeArgs);

__global__ void k_put_ext(DrawExt e, DrawExt* table, uint32_t draw) {
    if (threadIdx.x == 0 && blockIdx.x == 0) table[draw] = e;
}

// helpers/mod.rs:214-218 on the device: image row H-1-y = framebuffer row y (the presented image)
__global__ __launch_bounds__(256) void k_flip_rows(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int W, int H) {
    const size_t n = (size_t)W * (size_t)H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t y = i / (size_t)W, x = i - y * (size_t)W;
        dst[((size_t)H - 1 - y) * (size_t)W + x] = src[i];
    }
}

// =============================================================== misc fills
__global__ __launch_bounds__(256) void k_fill_u64(unsigned long long* p, size_t n,
                                                  unsigned long long v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// One launch per tr_clear: zero the counter slots and status flags, empty the NaN side table,
// restart the scanline prefix.
__global__ __launch_bounds__(256) void k_clear_state(unsigned long long* state_blk, uint32_t n_words, NanTable* nan,
                                                     uint32_t* half_off) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) state_blk[i] = 0ull;
    if (blockIdx.x == 0) {
        int t = threadIdx.x;
        if (t == 0) half_off[0] = 0;
        if (t < (int)kNanSlots) {
            nan->pix[t] = kNanEmpty;
            nan->first_nan[t] = kKeyNone;
            nan->first_any[t] = kKeyNone;
        }
    }
}


// ============================================================ arithmetic self-test
// The exact shortcuts of tr_device.h against the operations they stand for, on the device:
// [0] the hardware's saturating f64 -> u32 conversion vs the spelled-out `as u8` / `as usize` rule,
// [1] div255 vs `/ 255.0`, [2] pixel_coord vs round(), [3] values tried per check.
__device__ __forceinline__ unsigned long long st_next(unsigned long long& s) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return s;
}
__device__ __forceinline__ double st_double(unsigned long long& s, int exp_span) {
    // random sign and mantissa, exponent uniform in [-exp_span, exp_span]
    unsigned long long b = st_next(s);
    long long e = 1023 + (long long)(st_next(s) % (unsigned long long)(2 * exp_span + 1)) - exp_span;
    b = (b & 0x800FFFFFFFFFFFFFull) | ((unsigned long long)e << 52);
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ int pixel_coord_by_round(double v, int n) {
    double r = round(v);
    if (r != r) return 0;
    if (r < 0.0 || r > (double)(n - 1)) return -1;
    return (int)r;
}
__global__ __launch_bounds__(256) void k_selftest(unsigned long long seed, uint32_t per_thread, unsigned long long* out) {
    unsigned long long s = seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x + 1));
    unsigned long long bad0 = 0, bad1 = 0, bad2 = 0;
    auto differ = [](double a, double b) {
        return __double_as_longlong(a) != __double_as_longlong(b) && !(a != a && b != b);
    };
    for (uint32_t it = 0; it < per_thread; ++it) {
        const int span = (it & 7) == 0 ? 600 : ((it & 7) == 1 ? 1022 : 40);  // mostly the shader's range, some extremes
        const double den = st_double(s, span);
        double num = st_double(s, span);
        if ((it & 15) == 2) num = den * (double)(st_next(s) % 1000);  // exact quotients
        if ((it & 15) == 3) num = 0.0;
        if ((it & 31) == 4) num = __longlong_as_double(0x7FF0000000000000ll);
        (void)den;
        // bilinear mixes of bytes, plus arbitrary values
        const double fx = (double)(st_next(s) >> 11) * 0x1p-53, fy = (double)(st_next(s) >> 11) * 0x1p-53;
        const unsigned long long t = st_next(s);
        const double mix = (double)(t & 255) * ((1.0 - fx) * (1.0 - fy)) + (double)((t >> 8) & 255) * (fx * (1.0 - fy)) +
                           (double)((t >> 16) & 255) * ((1.0 - fx) * fy) + (double)((t >> 24) & 255) * (fx * fy);
        {   // the hardware's saturating conversions against the spelled-out rule
            const double cv = (it & 3) == 0 ? mix : ((it & 3) == 1 ? num : st_double(s, 12));
            bad0 += f64_as_u8(cv) != f64_as_u8_spelled_out(cv) ? 1 : 0;
            const uint32_t want = (cv > 0.0) ? (cv >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)cv) : 0u;
            bad0 += f64_as_u32_sat(cv) != want ? 1 : 0;
        }
        bad1 += differ(div255(mix), mix / 255.0) ? 1 : 0;
        bad1 += differ(div255(num), num / 255.0) ? 1 : 0;
        // pixel coordinates: around half-integers and anywhere
        const double base = (double)((long long)(st_next(s) % 8200) - 50) * 0.5;
        const long long ulps = (long long)(st_next(s) % 9) - 4;
        const double near = __longlong_as_double(__double_as_longlong(base) + (base != 0.0 ? ulps : 0));
        bad2 += pixel_coord(near, 4096) != pixel_coord_by_round(near, 4096) ? 1 : 0;
        bad2 += pixel_coord(num, 4096) != pixel_coord_by_round(num, 4096) ? 1 : 0;
        bad2 += pixel_coord(fx * 4100.0 - 2.0, 4096) != pixel_coord_by_round(fx * 4100.0 - 2.0, 4096) ? 1 : 0;
        {   // the sample loops' form: inside exactly when the raw value is below n as an unsigned number
            const double vs[3] = {near, num, fx * 4100.0 - 2.0};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int raw = pixel_coord_raw(vs[q]);
                const int viaraw = (uint32_t)raw < 4096u ? raw : -1;
                bad2 += viaraw != pixel_coord_by_round(vs[q], 4096) ? 1 : 0;
            }
        }
    }
    if (bad0) atomicAdd(&out[0], bad0);
    if (bad1) atomicAdd(&out[1], bad1);
    if (bad2) atomicAdd(&out[2], bad2);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[3] = (unsigned long long)per_thread * gridDim.x * blockDim.x;
}

}  // namespace tr
