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
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n_pos) {
        double g[4];
        mat4_mul_vec4(pmv, pos[3 * id], pos[3 * id + 1], pos[3 * id + 2], 1.0, g);
        ndc[id] = g[0] / g[3];
        ndc[n_pos + id] = g[1] / g[3];
        ndc[2 * n_pos + id] = g[2] / g[3];
    } else if (id < n_pos + n_nrm) {
        int k = id - n_pos;
        double g[4];
        mat4_mul_vec4(nmat, nrm[3 * k], nrm[3 * k + 1], nrm[3 * k + 2], 0.0, g);
        nout[k] = g[0];
        nout[n_nrm + k] = g[1];
        nout[2 * n_nrm + k] = g[2];
    }
}

// ================================================================ (2) setup
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
    uint32_t* half_cnt;    // scanlines per half, 2 per triangle (global-atomic raster only)
    uint32_t* tile_cnt;    // triangles per tile (tile raster only)
    int tiles_x, tiles_y;
    int y_lo, y_hi;        // band of pixel rows owned by this context
    int mode;              // bit0: fill half_cnt, bit1: count tiles
    uint32_t* flags;
    DrawDev params;
    DrawDev* draw_table;
};

__global__ __launch_bounds__(256) void k_setup(SetupArgs a) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) a.draw_table[a.draw] = a.params;
    if (k >= a.n_tris) return;
    const uint32_t* ix = a.idx + 9 * (size_t)k;
    uint32_t iv[3] = {ix[0], ix[3], ix[6]};
    uint32_t it[3] = {ix[1], ix[4], ix[7]};
    uint32_t in[3] = {ix[2], ix[5], ix[8]};
    const uint32_t seq = a.tri_base + (uint32_t)k;
    uint32_t res0 = 0, res1 = 0;

    double px[3], py[3], pz[3];  // clip_space_position (after divide), main.rs:154-162
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        px[c] = a.ndc[iv[c]];
        py[c] = a.ndc[a.n_pos + iv[c]];
        pz[c] = a.ndc[2 * a.n_pos + iv[c]];
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
    TriShade S;
    R.res[0] = R.res[1] = 0;
    R.tx0 = 1; R.tx1 = 0; R.ty0 = 1; R.ty1 = 0; R.flags = 0;
    if (cull) a.rast[seq] = R;  // the binning pass reads res / rect of every triangle
    if (!cull) {
        // mod.rs:320-327 viewport transform; w of the product is 1 unless inputs are non-finite.
        double vx[3], vy[3], vz[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double g[4];
            mat4_mul_vec4(a.vpm, px[c], py[c], pz[c], 1.0, g);
            vx[c] = g[0] / g[3];
            vy[c] = g[1] / g[3];
            vz[c] = g[2] / g[3];
        }
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
        // ---- shade record
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
            mat4_mul_vec4(a.nmat, cx, cy, cz, 0.0, g);
            fn[0] = g[0]; fn[1] = g[1]; fn[2] = g[2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (in[c] != 0xFFFFFFFFu) {
                S.n[c][0] = a.nout[in[c]];
                S.n[c][1] = a.nout[a.n_nrm + in[c]];
                S.n[c][2] = a.nout[2 * a.n_nrm + in[c]];
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
        S.draw = a.draw;
        S.flags = fl;
        R.res[0] = res0;
        R.res[1] = res1;
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
                    R.tx0 = (uint16_t)((int)cl / kTile); R.tx1 = (uint16_t)((int)ch / kTile);
                    R.ty0 = (uint16_t)((int)rl / kTile); R.ty1 = (uint16_t)((int)rh / kTile);
                }
            }
        }
        a.rast[seq] = R;
        a.shade[seq] = S;
        if ((a.mode & 2) && res0) {
            if (R.flags & kTriNonFinite) {
                int ty_lo = a.y_lo / kTile, ty_hi = (a.y_hi - 1) / kTile;
                if (ty_lo == 0)
                    for (int tx = 0; tx < a.tiles_x; ++tx) atomicAdd(&a.tile_cnt[tx], 1u);
                for (int ty = (ty_lo > 1 ? ty_lo : 1); ty <= ty_hi; ++ty) atomicAdd(&a.tile_cnt[ty * a.tiles_x], 1u);
            } else {
                for (int ty = R.ty0; ty <= R.ty1; ++ty)
                    for (int tx = R.tx0; tx <= R.tx1; ++tx) atomicAdd(&a.tile_cnt[ty * a.tiles_x + tx], 1u);
            }
        }
    }
    if (a.mode & 1) {
        a.half_cnt[2 * (size_t)seq] = res0 ? res0 + 1u : 0u;
        a.half_cnt[2 * (size_t)seq + 1] = res1 ? res1 + 1u : 0u;
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
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
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

__global__ __launch_bounds__(kScanThreads) void k_scan_tile_sums(const uint32_t* __restrict__ cnt,
                                                                 uint32_t n,
                                                                 unsigned long long* tile_sum) {
    uint32_t tile = blockIdx.x;
    uint32_t begin = tile * kScanTile + threadIdx.x * kScanPer;
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < kScanPer; ++i)
        if (begin + i < n) s += cnt[begin + i];
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

// single block: tile_sum -> exclusive tile bases (u64, in place), starting at *carry_in
__global__ __launch_bounds__(kScanThreads) void k_scan_tile_bases(unsigned long long* tile_sum,
                                                                  uint32_t n_tiles,
                                                                  const uint32_t* carry_in,
                                                                  uint32_t* flags) {
    __shared__ unsigned long long run;
    __shared__ unsigned long long wtot[kScanThreads / 64];
    if (threadIdx.x == 0) run = *carry_in;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += kScanThreads) {
        uint32_t i = base + threadIdx.x;
        unsigned long long v = i < n_tiles ? tile_sum[i] : 0ull;
        unsigned long long inc = v;
        int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        unsigned long long b = run, tot = 0;
        for (int k = 0; k < kScanThreads / 64; ++k) {
            if (k < w) b += wtot[k];
            tot += wtot[k];
        }
        if (i < n_tiles) tile_sum[i] = b + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) run += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && run > 0xFFFFFFFFull) atomicOr(flags, 1u << 4);
}

// per tile: exclusive scan with the tile base (tile_base == nullptr: single tile, base = off[0])
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const uint32_t* __restrict__ cnt,
                                                             uint32_t n, uint32_t* off,
                                                             const unsigned long long* tile_base,
                                                             uint32_t* flags) {
    uint32_t tile = blockIdx.x;
    uint32_t begin = tile * kScanTile + threadIdx.x * kScanPer;
    uint32_t v[kScanPer];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
        v[i] = (begin + i < n) ? cnt[begin + i] : 0u;
        s += v[i];
    }
    uint32_t tot;
    uint32_t ex = block_excl_scan(s, &tot);
    unsigned long long base = tile_base ? tile_base[tile] : (unsigned long long)off[0];
    if (!tile_base && threadIdx.x == 0 && base + tot > 0xFFFFFFFFull) atomicOr(flags, 1u << 4);
    uint32_t run = (uint32_t)base + ex;
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
        run += v[i];
        if (begin + i < n) off[begin + i + 1] = run;  // off[j+1] = inclusive sum through j
    }
}

// single block: the whole scan for up to a few tiles (one launch instead of three)
__global__ __launch_bounds__(kScanThreads) void k_scan_small(const uint32_t* __restrict__ cnt, uint32_t n,
                                                             uint32_t* off, uint32_t* flags) {
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = off[0];
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n; t0 += kScanTile) {
        uint32_t begin = t0 + threadIdx.x * kScanPer;
        uint32_t v[kScanPer];
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < kScanPer; ++i) {
            v[i] = (begin + i < n) ? cnt[begin + i] : 0u;
            s += v[i];
        }
        uint32_t tot;
        uint32_t ex = block_excl_scan(s, &tot);
        unsigned long long base = carry;
        uint32_t run = (uint32_t)base + ex;
#pragma unroll
        for (int i = 0; i < kScanPer; ++i) {
            run += v[i];
            if (begin + i < n) off[begin + i + 1] = run;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            carry = base + tot;
            if (base + tot > 0xFFFFFFFFull) atomicOr(flags, 1u << 4);
        }
        __syncthreads();
    }
}

// =============================================================== (3) raster
// One workgroup per chunk of 64 consecutive scanlines of the frame-global scanline sequence.
// Lanes of wave 0 first own one scanline each (find its half by binary search in half_off, build A and
// B-A, count its samples), the wave scans the counts, then lanes own SAMPLES: consecutive
// lanes take consecutive samples of a scanline, i.e. consecutive pixels of a row, so the
// depth / winner atomics of one wave-instruction fall into contiguous 8-byte slots.
struct RasterArgs {
    const uint32_t* half_off;
    const TriRaster* rast;
    unsigned long long* depth;  // encoded min depth per pixel, row-major
    unsigned long long* wkey;   // min order key among samples at the min depth
    uint32_t h_lo, h_hi;        // half range [h_lo, h_hi) to enumerate
    int W, H;
    int y_lo, y_hi;             // band of rows owned by this context: [y_lo, y_hi)
    unsigned long long* counters;
    uint32_t* flags;
    NanTable* nan;
};

constexpr int kRasterThreads = 256;  // wave 0 builds the chunk's scanlines, all 4 waves enumerate samples

template <int PASS, bool PRECHECK>
__global__ __launch_bounds__(kRasterThreads) void k_raster(RasterArgs a) {
    __shared__ double sA[3][64];
    __shared__ double sD[3][64];
    __shared__ double sInv[64];
    __shared__ unsigned long long sKey[64];
    __shared__ uint32_t sPre[65];

    const int lane = threadIdx.x & 63;
    const bool wave0 = threadIdx.x < 64;
    const uint32_t s_begin = a.half_off[a.h_lo];
    const uint32_t s_end = a.half_off[a.h_hi];
    const uint32_t n_scan = s_end - s_begin;
    const uint32_t n_chunks = (n_scan + 63u) / 64u;
    unsigned long long cnt_samples = 0, cnt_inb = 0;
    bool nan_active = false;
    if (PASS == 1) nan_active = a.nan->pix[0] != kNanEmpty;

    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        // ---- scanline phase
        uint32_t g = s_begin + chunk * 64u + (uint32_t)lane;
        bool valid = wave0 && g < s_end;
        uint32_t cnt = 0;
        if (valid) {
            uint32_t lo = a.h_lo, hi = a.h_hi;  // half_off[lo] <= g < half_off[hi]
            while (hi - lo > 1u) {
                uint32_t mid = (lo + hi) >> 1;
                if (a.half_off[mid] <= g) lo = mid; else hi = mid;
            }
            uint32_t tri = lo >> 1, half = lo & 1u, i = g - a.half_off[lo];
            const TriRaster& r = a.rast[tri];
            Scanline sl = make_scanline(r, half, i);
            if (sl.hres > 262143u) {
                atomicOr(a.flags, 1u << 0);
            } else {
                cnt = sl.hres + 1u;
                sA[0][lane] = sl.ax; sA[1][lane] = sl.ay; sA[2][lane] = sl.az;
                sD[0][lane] = sl.dx; sD[1][lane] = sl.dy; sD[2][lane] = sl.dz;
                sInv[lane] = 1.0 / (double)sl.hres;
                sKey[lane] = ((unsigned long long)tri << kKeyTriShift) |
                             ((unsigned long long)half << kKeyHalfShift) |
                             ((unsigned long long)i << kKeyIShift);
            }
        }
        uint32_t inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (wave0) {
            sPre[lane + 1] = inc;
            if (lane == 0) sPre[0] = 0;
        }
        __syncthreads();
        const uint32_t total = sPre[64];

        // ---- sample phase
        for (uint32_t k = threadIdx.x; k < total; k += (uint32_t)kRasterThreads) {
            uint32_t sidx = 0;
#pragma unroll
            for (uint32_t step = 32; step; step >>= 1)
                if (sPre[sidx + step] <= k) sidx += step;
            uint32_t j = k - sPre[sidx];
            double sp = (double)j * sInv[sidx];
            double x = sA[0][sidx] + sp * sD[0][sidx];
            double y = sA[1][sidx] + sp * sD[1][sidx];
            double z = sA[2][sidx] + sp * sD[2][sidx];
            int ix = pixel_coord(x, a.W);
            int iy = pixel_coord(y, a.H);
            bool inb = ix >= 0 && iy >= a.y_lo && iy < a.y_hi;
            if (PASS == 0) {
                cnt_samples += 1;
                cnt_inb += inb ? 1 : 0;
            }
            if (!inb) continue;
            double dist = -z;
            uint32_t pix = (uint32_t)iy * (uint32_t)a.W + (uint32_t)ix;
            bool isnan_d = dist != dist;
            if (PASS == 0) {
                if (isnan_d) {
                    bool placed = false;
                    for (uint32_t s = 0; s < kNanSlots && !placed; ++s) {
                        uint32_t old = atomicCAS(&a.nan->pix[s], kNanEmpty, pix);
                        placed = old == kNanEmpty || old == pix;
                    }
                    if (!placed) atomicOr(a.flags, 1u << 5);
                } else {
                    if (dist == -__builtin_huge_val()) atomicOr(a.flags, 1u << 3);
                    unsigned long long e = encode_depth(dist);
                    if (!PRECHECK || e < a.depth[pix]) atomicMin(&a.depth[pix], e);
                }
            } else {
                unsigned long long key = sKey[sidx] | (unsigned long long)j;
                if (!isnan_d) {
                    unsigned long long e = encode_depth(dist);
                    if (a.depth[pix] == e && (!PRECHECK || key < a.wkey[pix])) atomicMin(&a.wkey[pix], key);
                }
                if (nan_active) {
                    for (uint32_t s = 0; s < kNanSlots; ++s) {
                        if (a.nan->pix[s] == pix) {
                            if (isnan_d) atomicMin((unsigned long long*)&a.nan->first_nan[s], key);
                            else atomicMin((unsigned long long*)&a.nan->first_any[s], key);
                        }
                    }
                }
            }
        }
        __syncthreads();  // LDS is rewritten by the next chunk
    }
    if (PASS == 0) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            cnt_samples += __shfl_down(cnt_samples, d, 64);
            cnt_inb += __shfl_down(cnt_inb, d, 64);
        }
        if (lane == 0 && cnt_samples) {
            // one 64-byte slot per block (mod kCounterSlots): same-address atomics serialise
            unsigned long long* slot = a.counters + (size_t)(blockIdx.x % kCounterSlots) * 8;
            atomicAdd(&slot[0], cnt_samples);
            atomicAdd(&slot[1], cnt_inb);
        }
    }
}

template __global__ void k_raster<0, true>(RasterArgs);
template __global__ void k_raster<1, true>(RasterArgs);
template __global__ void k_raster<0, false>(RasterArgs);
template __global__ void k_raster<1, false>(RasterArgs);

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
    int W, H;
    uint32_t out_mask;
    uint32_t* rgba8;    // row-major, packed r | g<<8 | b<<16 | a<<24
    double* z_out;      // optional
    uint32_t* tri_out;  // optional
    double* rgba_f64;   // optional, 4 per pixel
    uint32_t* flags;
    unsigned long long* cov_counters;  // kCounterSlots x 8, [0] of each slot used
};

// helpers/mod.rs:253-278 on an RGBA8 image (texel value as f64 is exact).  Channels r,g,b.
__device__ __forceinline__ void sample_rgb(const uint32_t* __restrict__ img, int W, int H, double x,
                                           double y, double out[3], uint32_t* flags) {
    double x1 = floor(x), y1 = floor(y);
    // `x1 as usize`: NaN / negative -> 0, saturating above
    uint32_t ix1 = (x1 > 0.0) ? (x1 >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)x1) : 0u;
    uint32_t iy1 = (y1 > 0.0) ? (y1 >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)y1) : 0u;
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
    uint32_t c00 = img[(size_t)iy1 * W + ix1];
    uint32_t c10 = img[(size_t)iy1 * W + ix2];
    uint32_t c01 = img[(size_t)iy2 * W + ix1];
    uint32_t c11 = img[(size_t)iy2 * W + ix2];
    double w00 = (x2 - x) * (y2 - y);
    double w10 = (x - x1) * (y2 - y);
    double w01 = (x2 - x) * (y - y1);
    double w11 = (x - x1) * (y - y1);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        double a = (double)((c00 >> (8 * ch)) & 0xFFu);
        double b = (double)((c10 >> (8 * ch)) & 0xFFu);
        double c = (double)((c01 >> (8 * ch)) & 0xFFu);
        double d = (double)((c11 >> (8 * ch)) & 0xFFu);
        out[ch] = a * w00 + b * w10 + c * w01 + d * w11;
    }
}

// Rust `v as u8`
__device__ __forceinline__ uint32_t f64_as_u8(double v) {
    if (!(v > 0.0)) return 0u;  // NaN, negative, zero
    if (v >= 255.0) return 255u;
    return (uint32_t)v;
}

constexpr int kShadeBW = 32, kShadeBH = 8;  // pixels per 256-thread block: rows of 128 B of RGBA8

__global__ __launch_bounds__(256) void k_shade(ShadeArgs a) {
    const int x = blockIdx.x * kShadeBW + (threadIdx.x & (kShadeBW - 1));
    const int y = blockIdx.y * kShadeBH + (threadIdx.x / kShadeBW);
    const bool live = x < a.W && y < a.H;
    const uint32_t pix = (uint32_t)y * (uint32_t)a.W + (uint32_t)x;
    bool listed = live;
    if (a.tile_off && live) {
        int tile = (y / kTile) * a.tiles_x + (x / kTile);
        listed = a.tile_off[tile + 1] > a.tile_off[tile];
    }
    unsigned long long key = listed ? a.wkey[pix] : kKeyNone;
    if (a.self_clean && key != kKeyNone) {  // leave the planes cleared for the next frame
        a.wkey[pix] = kKeyNone;
        a.depth[pix] = kKeyNone;
    }
    if (a.nan && live && a.nan->pix[0] != kNanEmpty) {
        for (uint32_t s = 0; s < kNanSlots; ++s)
            if (a.nan->pix[s] == pix && a.nan->first_nan[s] < a.nan->first_any[s]) key = a.nan->first_nan[s];
    }
    bool has = key != kKeyNone;
    unsigned long long nwin = __popcll(__ballot(has));
    if ((threadIdx.x & 63) == 0 && nwin)
        atomicAdd(&a.cov_counters[(size_t)(((blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) % kCounterSlots) * 8], nwin);
    if (!live) return;
    double col[4] = {0.0, 0.0, 0.0, 255.0};  // BLACK, helpers/mod.rs:22
    double zval = -__builtin_huge_val();
    uint32_t tri = 0xFFFFFFFFu;
    if (has) {
        tri = (uint32_t)(key >> kKeyTriShift);
        uint32_t half = (uint32_t)(key >> kKeyHalfShift) & 1u;
        uint32_t i = (uint32_t)(key >> kKeyIShift) & 0x1FFFFu;
        uint32_t j = (uint32_t)key & 0x3FFFFu;
        const TriRaster& r = a.rast[tri];
        Scanline sl = make_scanline(r, half, i);
        double sp = (double)j * (1.0 / (double)sl.hres);
        double px = sl.ax + sp * sl.dx;
        double py = sl.ay + sp * sl.dy;
        double pz = sl.az + sp * sl.dz;
        zval = -pz;
        const TriShade& s = a.shade[tri];
        const DrawDev& dp = a.draws[s.draw];
        // helpers/mod.rs:326-333 barycentrics w.r.t. the UNSORTED viewport triangle
        double det = ((s.uy[1] - s.uy[2]) * (s.ux[0] - s.ux[2])) + ((s.ux[2] - s.ux[1]) * (s.uy[0] - s.uy[2]));
        double l1 = (((s.uy[1] - s.uy[2]) * (px - s.ux[2])) + ((s.ux[2] - s.ux[1]) * (py - s.uy[2]))) / det;
        double l2 = (((s.uy[2] - s.uy[0]) * (px - s.ux[2])) + ((s.ux[0] - s.ux[2]) * (py - s.uy[2]))) / det;
        double l3 = 1.0 - l1 - l2;
        // mod.rs:499-504
        double nx = l1 * s.n[0][0] + l2 * s.n[1][0] + l3 * s.n[2][0];
        double ny = l1 * s.n[0][1] + l2 * s.n[1][1] + l3 * s.n[2][1];
        double nz = l1 * s.n[0][2] + l2 * s.n[1][2] + l3 * s.n[2][2];
        normalize3(nx, ny, nz);
        bool has_uv = s.flags & 1u, has_tb = s.flags & 2u;
        double u = 0.0, v = 0.0;
        if (has_uv) {  // mod.rs:510-514
            u = l1 * s.uv[0][0] + l2 * s.uv[1][0] + l3 * s.uv[2][0];
            v = l1 * s.uv[0][1] + l2 * s.uv[1][1] + l3 * s.uv[2][1];
        }
        // main.rs:183-218
        if (has_uv && has_tb && dp.use_nm) {
            double rgb[3];
            sample_rgb(dp.nm, dp.nw, dp.nh, u * (double)(dp.nw - 1), v * (double)(dp.nh - 1), rgb, a.flags);
            double mx = ((rgb[0] / 255.0) * 2.0) - 1.0;
            double my = ((rgb[1] / 255.0) * 2.0) - 1.0;
            double mz = ((rgb[2] / 255.0) * 2.0) - 1.0;
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
#pragma unroll
        for (int c = 0; c < 3; ++c) col[c] = fmin(dp.ambient + (d * dp.intensity * albedo[c]), 255.0);
        col[3] = 255.0;
    }
    a.rgba8[pix] = f64_as_u8(col[0]) | (f64_as_u8(col[1]) << 8) | (f64_as_u8(col[2]) << 16) |
                   (f64_as_u8(col[3]) << 24);
    if (a.out_mask & 16u) a.key_out[pix] = key;
    if (a.out_mask & 2u) a.z_out[pix] = zval;
    if (a.out_mask & 4u) a.tri_out[pix] = tri;
    if (a.out_mask & 8u) {
        double4 o = make_double4(col[0], col[1], col[2], col[3]);
        reinterpret_cast<double4*>(a.rgba_f64)[pix] = o;
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

}  // namespace tr
