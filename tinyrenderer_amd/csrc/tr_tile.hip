// Tile rasterizer (stages 2-3 of BASELINE.json north_star): triangles are binned to 16x16
// screen tiles, one wave owns one tile, keeps the tile's depth and winner planes in LDS and
// reproduces the reference's sample enumeration (model_renderer/mod.rs:360-390,
// segment_3d/mod.rs:19-115) restricted to the (scanline i, sample j) ranges that can land in
// the tile.  Range pruning is conservative (never drops a sample: every candidate is re-tested
// with the exact pixel rounding), so the result is identical to enumerating everything.
#include "tr_device.h"

namespace tr {

// ------------------------------------------------------------ (2) binning
// tile_cnt[t] was filled by k_setup.  One block: exclusive scan into tile_off[0..n_tiles],
// and tile_cnt is zeroed so it can serve as the fill cursor.
constexpr int kTileScanThreads = 1024;

__global__ __launch_bounds__(kTileScanThreads) void k_tile_scan(uint32_t* tile_cnt, uint32_t* tile_off,
                                                                 uint32_t n_tiles, uint32_t pair_cap,
                                                                 uint32_t* flags) {
    __shared__ uint32_t wsum[kTileScanThreads / 64];
    const uint32_t per = (n_tiles + kTileScanThreads - 1) / kTileScanThreads;
    const uint32_t begin = threadIdx.x * per;
    uint32_t s = 0;
    for (uint32_t i = 0; i < per; ++i)
        if (begin + i < n_tiles) s += tile_cnt[begin + i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int i = 0; i < kTileScanThreads / 64; ++i) {
        if (i < w) base += wsum[i];
        total += wsum[i];
    }
    uint32_t run = base + inc - s;
    for (uint32_t i = 0; i < per; ++i)
        if (begin + i < n_tiles) {
            tile_off[begin + i] = run;
            run += tile_cnt[begin + i];
            tile_cnt[begin + i] = 0;
        }
    if (threadIdx.x == 0) {
        tile_off[n_tiles] = total;
        if (total > pair_cap) atomicOr(flags, 1u << 6);  // TR_FLAG_BIN_OVERFLOW: host grows and redoes
    }
}

// One thread per triangle of the frame: append its tri_seq to the list of every tile of its rect.
__global__ __launch_bounds__(256) void k_bin_fill(const TriRaster* __restrict__ rast, uint32_t n_tris,
                                                  const uint32_t* __restrict__ tile_off, uint32_t* cursor,
                                                  uint32_t* pairs, uint32_t pair_cap, int tiles_x, int y_lo,
                                                  int y_hi) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_tris) return;
    const TriRaster& r = rast[k];
    if (!r.res[0]) return;
    auto put = [&](int tile) {
        uint32_t pos = tile_off[tile] + atomicAdd(&cursor[tile], 1u);
        if (pos < pair_cap) pairs[pos] = k;
    };
    if (r.flags & kTriNonFinite) {
        int ty_lo = y_lo / kTile, ty_hi = (y_hi - 1) / kTile;
        if (ty_lo == 0)
            for (int tx = 0; tx < tiles_x; ++tx) put(tx);
        for (int ty = (ty_lo > 1 ? ty_lo : 1); ty <= ty_hi; ++ty) put(ty * tiles_x);
    } else {
        for (int ty = r.ty0; ty <= r.ty1; ++ty)
            for (int tx = r.tx0; tx <= r.tx1; ++tx) put(ty * tiles_x + tx);
    }
}

// ------------------------------------------------------------ (3) tile raster
struct TileArgs {
    const uint32_t* tile_off;
    const uint32_t* pairs;
    uint32_t pair_cap;
    const TriRaster* rast;
    unsigned long long* wkey;  // winner key plane, row-major; written only for non-empty tiles
    int W, H;
    int tiles_x;
    int ty_lo;                 // first tile row of this context's band (blockIdx.y = 0)
    int y_lo, y_hi;
    unsigned long long* counters;  // kCounterSlots x 8
    uint32_t* flags;
};

// Scanlines i of one triangle half whose samples can have round(y) in [Y0+0.5.., Y1-0.5]:
// A.y(i) = a1y + (i/res)*dA and |B.y - A.y| <= g, so i/res must put A.y within [Y0-g, Y1+g].
__device__ __forceinline__ void scanline_range(const TriRaster& r, uint32_t half, double Y0, double Y1,
                                               int& ilo, int& ihi) {
    const uint32_t res = r.res[half];
    if (r.flags & kTriNonFinite) { ilo = 0; ihi = (int)res; return; }
    const double a1y = half ? r.s2[1] : r.s0[1];
    const double dA = r.s1[1] - a1y;
    const double g = fabs(r.s1[1] - r.sp[1]) + kPruneMargin;
    const double lo = Y0 - g, hi = Y1 + g;
    if (dA == 0.0) {
        if (a1y >= lo && a1y <= hi) { ilo = 0; ihi = (int)res; } else { ilo = 1; ihi = 0; }
        return;
    }
    double t0 = (lo - a1y) / dA, t1 = (hi - a1y) / dA;
    if (t0 > t1) { double t = t0; t0 = t1; t1 = t; }
    double f0 = floor(t0 * (double)res) - 1.0, f1 = ceil(t1 * (double)res) + 1.0;
    if (!(f0 > 0.0)) f0 = 0.0;                   // also catches NaN
    if (!(f1 < (double)res)) f1 = (double)res;
    if (f0 > f1) { ilo = 1; ihi = 0; return; }
    ilo = (int)f0;
    ihi = (int)f1;
}

// Samples j of a scanline whose x can round into [X0+0.5.., X1-0.5].
__device__ __forceinline__ void sample_range(const Scanline& s, bool prune, double X0, double X1, int& jlo,
                                             int& jhi) {
    const int hres = (int)s.hres;
    if (!prune) { jlo = 0; jhi = hres; return; }
    if (s.dx == 0.0) {
        if (s.ax >= X0 && s.ax <= X1) { jlo = 0; jhi = hres; } else { jlo = 1; jhi = 0; }
        return;
    }
    double t0 = (X0 - s.ax) / s.dx, t1 = (X1 - s.ax) / s.dx;
    if (t0 > t1) { double t = t0; t0 = t1; t1 = t; }
    double f0 = floor(t0 * (double)hres) - 1.0, f1 = ceil(t1 * (double)hres) + 1.0;
    if (!(f0 > 0.0)) f0 = 0.0;
    if (!(f1 < (double)hres)) f1 = (double)hres;
    if (f0 > f1) { jlo = 1; jhi = 0; return; }
    jlo = (int)f0;
    jhi = (int)f1;
}

// One wave per tile.  Phases: 0 depth min; 1 winner key among samples at the min depth;
// (only when a NaN depth was seen) 2 first key over non-NaN samples, 3 NaN samples that precede it.
__global__ __launch_bounds__(64) void k_tile_raster(TileArgs a) {
    __shared__ unsigned long long sDepth[kTile * kTile];
    __shared__ unsigned long long sKey[kTile * kTile];
    __shared__ uint32_t sPre[65];
    __shared__ int sIlo[64];
    __shared__ uint32_t sTri[64];
    __shared__ uint32_t sNan;

    const int lane = threadIdx.x;
    const int tx = blockIdx.x, ty = a.ty_lo + blockIdx.y;
    const int tile = ty * a.tiles_x + tx;
    const uint32_t list_lo = a.tile_off[tile];
    uint32_t list_hi = a.tile_off[tile + 1];
    if (list_hi > a.pair_cap) list_hi = a.pair_cap;  // overflow is flagged; the host redoes the frame
    if (list_hi <= list_lo) return;                   // empty tile: the shade pass knows from tile_off
    const uint32_t n_half = 2u * (list_hi - list_lo);

    const int px0 = tx * kTile, py0 = ty * kTile;
    const double X0 = (double)px0 - 0.5 - kPruneMargin, X1 = (double)(px0 + kTile - 1) + 0.5 + kPruneMargin;
    const double Y0 = (double)py0 - 0.5 - kPruneMargin, Y1 = (double)(py0 + kTile - 1) + 0.5 + kPruneMargin;

#pragma unroll
    for (int q = 0; q < kTile * kTile / 64; ++q) {
        sDepth[lane + 64 * q] = kKeyNone;
        sKey[lane + 64 * q] = kKeyNone;
    }
    if (lane == 0) sNan = 0;
    __syncthreads();

    unsigned long long cnt_inb = 0;
    int n_phase = 2;
    for (int phase = 0; phase < n_phase; ++phase) {
        for (uint32_t base = 0; base < n_half; base += 64u) {
            // ---- half phase: lane = one (triangle, half) of the tile's list
            uint32_t he = base + (uint32_t)lane;
            uint32_t cnt = 0;
            if (he < n_half) {
                uint32_t tri = a.pairs[list_lo + (he >> 1)];
                uint32_t half = he & 1u;
                const TriRaster& r = a.rast[tri];
                int ilo, ihi;
                scanline_range(r, half, Y0, Y1, ilo, ihi);
                if (ihi >= ilo) cnt = (uint32_t)(ihi - ilo + 1);
                sIlo[lane] = ilo;
                sTri[lane] = (tri << 1) | half;
            }
            uint32_t inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t t = __shfl_up(inc, d, 64);
                if (lane >= d) inc += t;
            }
            sPre[lane + 1] = inc;
            if (lane == 0) sPre[0] = 0;
            __syncthreads();
            const uint32_t total = sPre[64];

            // ---- segment phase: lane = one scanline restricted to the tile's columns
            for (uint32_t m = (uint32_t)lane; m < total; m += 64u) {
                uint32_t e = 0;
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1)
                    if (sPre[e + step] <= m) e += step;
                const uint32_t th = sTri[e];
                const uint32_t tri = th >> 1, half = th & 1u;
                const uint32_t i = (uint32_t)sIlo[e] + (m - sPre[e]);
                const TriRaster& r = a.rast[tri];
                const Scanline sl = make_scanline(r, half, i);
                if (sl.hres > 262143u) { atomicOr(a.flags, 1u << 0); continue; }
                int jlo, jhi;
                sample_range(sl, !(r.flags & kTriNonFinite), X0, X1, jlo, jhi);
                const double inv = 1.0 / (double)sl.hres;
                const unsigned long long key_hi = ((unsigned long long)tri << kKeyTriShift) |
                                                  ((unsigned long long)half << kKeyHalfShift) |
                                                  ((unsigned long long)i << kKeyIShift);
                for (int j = jlo; j <= jhi; ++j) {
                    double sp = (double)j * inv;
                    double x = sl.ax + sp * sl.dx;
                    double y = sl.ay + sp * sl.dy;
                    double z = sl.az + sp * sl.dz;
                    int ix = pixel_coord(x, a.W);
                    int iy = pixel_coord(y, a.H);
                    // inside this tile (which also implies inside the screen) and inside the band
                    if (ix < px0 || ix >= px0 + kTile || iy < py0 || iy >= py0 + kTile) continue;
                    if (iy < a.y_lo || iy >= a.y_hi) continue;
                    const int lp = (iy - py0) * kTile + (ix - px0);
                    const double dist = -z;
                    const bool isnan_d = dist != dist;
                    const unsigned long long key = key_hi | (unsigned long long)j;
                    if (phase == 0) {
                        cnt_inb += 1;
                        if (isnan_d) {
                            sNan = 1;
                        } else {
                            if (dist == -__builtin_huge_val()) atomicOr(a.flags, 1u << 3);
                            atomicMin(&sDepth[lp], encode_depth(dist));
                        }
                    } else if (phase == 1) {
                        if (!isnan_d && sDepth[lp] == encode_depth(dist)) atomicMin(&sKey[lp], key);
                    } else if (phase == 2) {
                        if (!isnan_d) atomicMin(&sDepth[lp], key);  // sDepth reused: first non-NaN key
                    } else {
                        if (isnan_d && key < sDepth[lp]) atomicMin(&sKey[lp], key);
                    }
                }
            }
            __syncthreads();
        }
        if (phase == 1 && sNan) {  // rare: a NaN-depth sample landed in this tile
            n_phase = 4;
#pragma unroll
            for (int q = 0; q < kTile * kTile / 64; ++q) sDepth[lane + 64 * q] = kKeyNone;
            __syncthreads();
        }
    }

    // ---- write the tile's winner keys (rows of 16 keys = 128 contiguous bytes)
#pragma unroll
    for (int q = 0; q < kTile * kTile / 64; ++q) {
        int lp = lane + 64 * q;
        int x = px0 + (lp & (kTile - 1)), y = py0 + (lp >> 4);
        if (x < a.W && y < a.H) a.wkey[(size_t)y * a.W + x] = sKey[lp];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt_inb += __shfl_down(cnt_inb, d, 64);
    if (lane == 0 && cnt_inb) atomicAdd(&a.counters[(size_t)(tile % kCounterSlots) * 8 + 1], cnt_inb);
}

}  // namespace tr
