// Tile rasterizer (stages 2-3 of BASELINE.json north_star): triangles are binned to 16x16
// screen tiles, one workgroup owns one tile, keeps the tile's depth and winner planes in LDS
// and reproduces the reference's sample enumeration (model_renderer/mod.rs:360-390,
// segment_3d/mod.rs:19-115) restricted to the (scanline i, sample j) ranges that can land in
// the tile.  Range pruning is conservative (never drops a sample: every candidate is re-tested
// with the exact pixel rounding), so the result is identical to enumerating everything.
#include "tr_device.h"

namespace tr {

// ------------------------------------------------------------ (2) binning
// tile_cnt[t] was filled by k_setup.  One block: exclusive scan into tile_off[0..n_tiles]
// (wavefront prefix sums + one cross-wave pass); tile_cnt is zeroed so it can serve as the
// fill cursor.  Arrays are padded to a multiple of 4*kTileScanThreads entries so every thread
// streams whole uint4s.
constexpr int kTileScanThreads = 1024;

__global__ __launch_bounds__(kTileScanThreads) void k_tile_scan(uint32_t* tile_cnt, uint32_t* tile_off,
                                                                 uint32_t* tile_list, uint32_t n_tiles,
                                                                 uint32_t n_padded, uint32_t pair_cap,
                                                                 uint32_t* flags) {
    __shared__ uint32_t wsum[kTileScanThreads / 64];
    __shared__ uint32_t wne[kTileScanThreads / 64];
    const uint32_t per4 = n_padded / (4 * kTileScanThreads);  // uint4s per thread
    uint4* cnt4 = reinterpret_cast<uint4*>(tile_cnt) + (size_t)threadIdx.x * per4;
    uint32_t s = 0, ne = 0;
    for (uint32_t i = 0; i < per4; ++i) {
        uint4 v = cnt4[i];
        s += v.x + v.y + v.z + v.w;
        ne += (v.x != 0) + (v.y != 0) + (v.z != 0) + (v.w != 0);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = s, inc_ne = ne;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, 64);
        uint32_t u = __shfl_up(inc_ne, d, 64);
        if (lane >= d) { inc += t; inc_ne += u; }
    }
    if (lane == 63) { wsum[w] = inc; wne[w] = inc_ne; }
    __syncthreads();
    uint32_t base = 0, total = 0, base_ne = 0, total_ne = 0;
#pragma unroll
    for (int i = 0; i < kTileScanThreads / 64; ++i) {
        if (i < w) { base += wsum[i]; base_ne += wne[i]; }
        total += wsum[i];
        total_ne += wne[i];
    }
    uint32_t run = base + inc - s;
    uint32_t pos = base_ne + inc_ne - ne;  // compaction cursor of non-empty tiles
    uint4* off4 = reinterpret_cast<uint4*>(tile_off) + (size_t)threadIdx.x * per4;
    const uint32_t first = threadIdx.x * per4 * 4;
    for (uint32_t i = 0; i < per4; ++i) {
        uint4 v = cnt4[i];
        uint4 o;
        o.x = run; run += v.x;
        o.y = run; run += v.y;
        o.z = run; run += v.z;
        o.w = run; run += v.w;
        off4[i] = o;
        cnt4[i] = make_uint4(0, 0, 0, 0);
        if (v.x) tile_list[pos++] = first + 4 * i;
        if (v.y) tile_list[pos++] = first + 4 * i + 1;
        if (v.z) tile_list[pos++] = first + 4 * i + 2;
        if (v.w) tile_list[pos++] = first + 4 * i + 3;
    }
    // padding entries hold count 0, so tile_off[n_tiles .. n_padded] all equal the total
    if (threadIdx.x == 0) {
        tile_off[n_padded] = total;
        tile_off[n_padded + 1] = total_ne;  // number of non-empty tiles
        if (total > pair_cap) atomicOr(flags, 1u << 6);  // TR_FLAG_BIN_OVERFLOW: host grows and redoes
    }
    (void)n_tiles;
}

// 16 lanes per triangle of the frame: lane q appends the triangle's tri_seq to the list of the
// q-th, (q+16)-th, ... tile of its rect.
constexpr int kFillLanes = 16;

__global__ __launch_bounds__(256) void k_bin_fill(const TriRaster* __restrict__ rast, uint32_t n_tris,
                                                  const uint32_t* __restrict__ tile_off, uint32_t* cursor,
                                                  uint32_t* pairs, uint32_t pair_cap, int tiles_x, int y_lo,
                                                  int y_hi, int th_shift) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (n_tris * 16 may pass 2^32)
    const uint32_t k = (uint32_t)(gid / kFillLanes);
    const int q = (int)(gid % kFillLanes);
    if (k >= n_tris) return;
    const TriRaster& r = rast[k];
    if (!r.res[0]) return;
    auto put = [&](int tile) {
        uint32_t pos = tile_off[tile] + atomicAdd(&cursor[tile], 1u);
        if (pos < pair_cap) pairs[pos] = k;
    };
    if (r.flags & kTriNonFinite) {
        int ty_lo = y_lo >> th_shift, ty_hi = (y_hi - 1) >> th_shift;
        if (ty_lo == 0)
            for (int tx = q; tx < tiles_x; tx += kFillLanes) put(tx);
        for (int ty = (ty_lo > 1 ? ty_lo : 1) + q; ty <= ty_hi; ty += kFillLanes) put(ty * tiles_x);
    } else {
        const int w = (int)r.tx1 - (int)r.tx0 + 1, h = (int)r.ty1 - (int)r.ty0 + 1;
        for (int t = q; t < w * h; t += kFillLanes) put((r.ty0 + t / w) * tiles_x + r.tx0 + t % w);
    }
}

// ------------------------------------------------------------ (3) tile raster
struct TileArgs {
    const uint32_t* tile_list;  // ids of non-empty tiles; count at tile_off[n_padded + 1]
    uint32_t n_padded;
    const uint32_t* tile_off;
    const uint32_t* pairs;
    uint32_t pair_cap;
    const TriRaster* rast;
    unsigned long long* wkey;  // winner key plane, row-major; written only for non-empty tiles
    int W, H;
    int tiles_x;
    int y_lo, y_hi;
    unsigned long long* counters;  // kCounterSlots x 8
    uint32_t* flags;
    int debug;                     // diagnostic early-outs (TR_TILE_DEBUG), 0 in production
};

// What the raster needs of a triangle, staged in LDS per batch (13 doubles).
struct LdsTri {
    double s0[3], s1[3], s2[3], sp[3];
    uint32_t res[2];
};

// Scanlines i of one triangle half whose samples can have round(y) in [Y0+0.5.., Y1-0.5]:
// A.y(i) = a1y + (i/res)*dA and |B.y - A.y| <= g, so i/res must put A.y within [Y0-g, Y1+g].
__device__ __forceinline__ void scanline_range(const LdsTri& r, bool prune, uint32_t half, double Y0, double Y1,
                                               int& ilo, int& ihi) {
    const uint32_t res = r.res[half];
    if (!prune) { ilo = 0; ihi = (int)res; return; }
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

// make_scanline on an LDS-staged triangle (same arithmetic as tr_device.h's).
__device__ __forceinline__ Scanline make_scanline_lds(const LdsTri& r, uint32_t half, uint32_t i) {
    const double* a1 = half ? r.s2 : r.s0;
    double delta = 1.0 / (double)r.res[half];
    double t = (double)i * delta;
    double Ax = a1[0] + t * (r.s1[0] - a1[0]);
    double Ay = a1[1] + t * (r.s1[1] - a1[1]);
    double Az = a1[2] + t * (r.s1[2] - a1[2]);
    double Bx = a1[0] + t * (r.sp[0] - a1[0]);
    double By = a1[1] + t * (r.sp[1] - a1[1]);
    double Bz = a1[2] + t * (r.sp[2] - a1[2]);
    Scanline s;
    s.ax = Ax; s.ay = Ay; s.az = Az;
    s.dx = Bx - Ax; s.dy = By - Ay; s.dz = Bz - Az;
    double c = ceil(fmax(fabs(s.dx), fabs(s.dy)));
    uint32_t h;
    if (c != c) h = 1u;
    else if (c > 262142.0) h = 262144u;
    else h = (uint32_t)c + 1u;
    s.hres = h < 2u ? 2u : h;
    return s;
}

constexpr int kTileWaves = 4;        // waves cooperating on one tile, each with private LDS planes
constexpr int kTileThreads = 64 * kTileWaves;
constexpr int kBatchTris = 32;       // triangles staged in LDS per batch
constexpr int kOwnCap = 512;         // samples expanded per window (owner table entries, per wave)
constexpr int kUnroll = 4;           // samples per lane per step of the sample loop
constexpr int kTileGrid = 256 * 3;   // persistent workgroups striding over the non-empty tiles

// Local pixel coordinate (0..span-1) of `round(v) as i64` inside the span of `span` pixels starting at pixel
// `lo`, or -1 when the rounded value falls outside the span or past `limit` (screen edge).
// round() is half away from zero; v - floor(v) is exact for |v| < 2^52, NaN casts to 0.
__device__ __forceinline__ int tile_coord(double v, int lo, int limit, int span) {
    double r;
    if (v != v) r = 0.0;
    else if (v < 0.0) r = (v > -0.5) ? 0.0 : -1.0;
    else {
        double f = floor(v);
        r = (v - f >= 0.5) ? f + 1.0 : f;
    }
    double l = r - (double)lo;
    if (!(l >= 0.0 && l <= (double)(span - 1)) || r > (double)(limit - 1)) return -1;
    return (int)l;
}

// Conservative sample range of a scanline inside [X0, X1] in f32: the estimate is off by far
// less than one sample (|j| < 2^18, a handful of f32 roundings), and the range is widened by
// one sample on each side; every candidate is re-tested exactly afterwards.
__device__ __forceinline__ void sample_range_fast(double ax, double dx, uint32_t hres_u, bool prune, double X0,
                                                  double X1, int& jlo, int& jhi) {
    const int hres = (int)hres_u;
    if (!prune) { jlo = 0; jhi = hres; return; }
    if (dx == 0.0) {
        if (ax >= X0 && ax <= X1) { jlo = 0; jhi = hres; } else { jlo = 1; jhi = 0; }
        return;
    }
    const float scale = (float)hres * __builtin_amdgcn_rcpf((float)dx);
    float t0 = (float)(X0 - ax) * scale, t1 = (float)(X1 - ax) * scale;
    if (t0 > t1) { float t = t0; t0 = t1; t1 = t; }
    float f0 = floorf(t0) - 1.0f, f1 = ceilf(t1) + 1.0f;
    if (!(f0 > 0.0f)) f0 = 0.0f;
    if (!(f1 < (float)hres)) f1 = (float)hres;
    if (f0 > f1) { jlo = 1; jhi = 0; return; }
    jlo = (int)f0;
    jhi = (int)f1;
}

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving
    // LDS accesses across the hand-off between lanes of the same wave.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One workgroup of kTileWaves waves per tile, persistent over the list of non-empty tiles.
//
// The tile's triangle list is walked in batches of kBatchTris staged in LDS; per batch
// (a) one lane per triangle half finds the scanlines that cross the tile's rows; the resulting
// scanline list is dealt to the waves in rounds of 64; per round a wave (b) builds, one lane per
// scanline, the end points and the sample range that crosses the tile's columns, and (c) flattens
// the samples of those scanlines through an owner table so every lane takes one sample per step.
//
// Each wave accumulates into PRIVATE depth / winner planes.  Because a private plane is touched
// by one wave only, its LDS instructions execute in program order, which keeps the depth
// minimum AND the winner key exact in a single enumeration:
//   old = ds_min_rtn(depth[p], enc);  cur = depth[p]           (cur = minimum after this step)
//   a lane with enc == cur and old > cur lowered the pixel in this step: it resets key[p];
//   every lane with enc == cur then does ds_min(key[p], key)   (ties keep the smallest order key)
// Exactly one lane per lowered pixel sees old > cur == enc (the first one the LDS serialises).
// The kTileWaves private results are merged per pixel by lexicographic (depth, key) minimum,
// which is exact because each holds its own minimum depth and the smallest key at that depth.
// NaN depths (SURVEY.md 7.1 caveat i) are rare and handled by two extra enumerations on the
// merged planes: first key over non-NaN samples, then NaN samples that precede it.
// TW x TH: the tile's size in pixels (powers of two).  The private planes (kTileWaves x TW x TH x 16 B: 16 KB at 16 x 16,
// 128 KB at 64 x 32 of the CU's 160 KB) are dynamic shared memory.
template <int TW, int TH>
__global__ __launch_bounds__(kTileThreads) void k_tile_raster(TileArgs a) {
    constexpr int kPix = TW * TH;
    extern __shared__ __align__(16) unsigned long long s_planes[];
    unsigned long long (*pDepth)[kPix] = reinterpret_cast<unsigned long long (*)[kPix]>(s_planes);
    unsigned long long (*pKey)[kPix] = reinterpret_cast<unsigned long long (*)[kPix]>(s_planes + (size_t)kTileWaves * kPix);
    __shared__ LdsTri sTri[kBatchTris];
    __shared__ double sDelta[2 * kBatchTris];  // 1/res per half (segment_3d/mod.rs:110)
    __shared__ uint32_t sSeq[kBatchTris];      // tri_seq | nonfinite<<31
    __shared__ uint32_t sPre[2 * kBatchTris + 1];
    __shared__ int sIlo[2 * kBatchTris];
    __shared__ uint32_t sNan;
    // per-wave segment records of the current round (SoA, one per lane) and owner table
    __shared__ double gA_[kTileWaves][3][64];
    __shared__ double gD_[kTileWaves][3][64];
    __shared__ double gInv_[kTileWaves][64];
    __shared__ unsigned long long gKey_[kTileWaves][64];
    __shared__ int gJlo_[kTileWaves][64];
    __shared__ uint32_t gPre_[kTileWaves][64];
    __shared__ unsigned char sOwn_[kTileWaves][kOwnCap];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    double (*gA)[64] = gA_[wv];
    double (*gD)[64] = gD_[wv];
    double* gInv = gInv_[wv];
    unsigned long long* gKey = gKey_[wv];
    int* gJlo = gJlo_[wv];
    uint32_t* gPre = gPre_[wv];
    unsigned char* sOwn = sOwn_[wv];

    const uint32_t n_list = a.tile_off[a.n_padded + 1];
    unsigned long long cnt_inb = 0, cnt_eval = 0, cnt_seg = 0, cnt_steps = 0;

    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const int tile = (int)a.tile_list[li];
        const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
        const uint32_t list_lo = a.tile_off[tile];
        uint32_t list_hi = a.tile_off[tile + 1];
        if (list_hi > a.pair_cap) list_hi = a.pair_cap;  // overflow is flagged; the host redoes the frame
        if (list_hi <= list_lo) continue;
        const uint32_t n_tri = list_hi - list_lo;
        if (a.debug == 1) continue;

        const int px0 = tx * TW, py0 = ty * TH;
        const double X0 = (double)px0 - 0.5 - kPruneMargin, X1 = (double)(px0 + TW - 1) + 0.5 + kPruneMargin;
        const double Y0 = (double)py0 - 0.5 - kPruneMargin, Y1 = (double)(py0 + TH - 1) + 0.5 + kPruneMargin;
        const int y_limit = a.y_hi < a.H ? a.y_hi : a.H;
        const long long t_start = (a.debug == 8) ? (long long)wall_clock64() : 0;

        __syncthreads();  // the previous tile's merge / write-out is done
        for (int p = tid; p < kTileWaves * kPix; p += kTileThreads) {
            pDepth[0][p] = kKeyNone;  // (the planes of the four waves are contiguous)
            pKey[0][p] = kKeyNone;
        }
        if (tid == 0) sNan = 0;

        // phase 0: depth + winner into the private planes of each wave; then merge into plane 0.
        // phases 2 / 3 (only if a NaN depth landed): plain atomic accumulations on plane 0.
        int phase = 0;
        while (true) {
            unsigned long long* myDepth = phase == 0 ? pDepth[wv] : pDepth[0];
            unsigned long long* myKey = phase == 0 ? pKey[wv] : pKey[0];
            for (uint32_t base = 0; base < n_tri; base += (uint32_t)kBatchTris) {
                const uint32_t nb = min((uint32_t)kBatchTris, n_tri - base);
                __syncthreads();  // every wave is done with the previous batch's staging arrays
                // ---- stage nb triangle records: 4 threads per triangle, 3 doubles each
                {
                    const uint32_t t = (uint32_t)tid >> 2, part = (uint32_t)tid & 3u;
                    if (t < nb) {
                        const uint32_t seq = a.pairs[list_lo + base + t];
                        const TriRaster& r = a.rast[seq];
                        const double* src = part == 0 ? r.s0 : part == 1 ? r.s1 : part == 2 ? r.s2 : r.sp;
                        double* dst = part == 0 ? sTri[t].s0 : part == 1 ? sTri[t].s1 : part == 2 ? sTri[t].s2 : sTri[t].sp;
                        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
                        if (part == 0) {
                            sTri[t].res[0] = r.res[0];
                            sTri[t].res[1] = r.res[1];
                            sSeq[t] = seq | ((r.flags & kTriNonFinite) ? 0x80000000u : 0u);
                        }
                    }
                }
                __syncthreads();
                // ---- (a) wave 0, lane = one (triangle, half): scanline range inside the tile's rows
                if (wv == 0) {
                    uint32_t cnt = 0;
                    if ((uint32_t)lane < 2u * nb) {
                        const uint32_t t = (uint32_t)lane >> 1, half = (uint32_t)lane & 1u;
                        int ilo, ihi;
                        scanline_range(sTri[t], !(sSeq[t] >> 31), half, Y0, Y1, ilo, ihi);
                        if (ihi >= ilo) cnt = (uint32_t)(ihi - ilo + 1);
                        sIlo[lane] = ilo;
                        sDelta[lane] = 1.0 / (double)sTri[t].res[half];
                    }
                    uint32_t inc = cnt;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        uint32_t t = __shfl_up(inc, d, 64);
                        if (lane >= d) inc += t;
                    }
                    sPre[lane + 1] = inc;
                    if (lane == 0) sPre[0] = 0;
                }
                __syncthreads();
                const uint32_t total = a.debug == 2 ? 0u : sPre[2 * nb];

                // ---- rounds of 64 scanlines, dealt round-robin to the waves (no block barrier inside)
                for (uint32_t seg0 = (uint32_t)wv * 64u; seg0 < total; seg0 += 64u * kTileWaves) {
                    const uint32_t n_seg = min(64u, total - seg0);
                    wave_lds_sync();  // this wave's readers of the previous round's records are done
                    // ---- (b) lane = one scanline: end points + sample range inside the tile's columns
                    uint32_t my_cnt = 0;
                    if ((uint32_t)lane < n_seg) {
                        const uint32_t m = seg0 + (uint32_t)lane;
                        uint32_t e = 0;
#pragma unroll
                        for (uint32_t step = kBatchTris; step; step >>= 1)
                            if (e + step < 2u * nb + 1u && sPre[e + step] <= m) e += step;
                        const uint32_t t = e >> 1, half = e & 1u;
                        const uint32_t i = (uint32_t)sIlo[e] + (m - sPre[e]);
                        const uint32_t seqf = sSeq[t];
                        const LdsTri& r = sTri[t];
                        // segment_3d/mod.rs:26-33 for both edges of the half at t = i * (1/res)
                        const double* a1 = half ? r.s2 : r.s0;
                        const double tt = (double)i * sDelta[e];
                        const double Ax = a1[0] + tt * (r.s1[0] - a1[0]);
                        const double Ay = a1[1] + tt * (r.s1[1] - a1[1]);
                        const double Az = a1[2] + tt * (r.s1[2] - a1[2]);
                        const double Bx = a1[0] + tt * (r.sp[0] - a1[0]);
                        const double By = a1[1] + tt * (r.sp[1] - a1[1]);
                        const double Bz = a1[2] + tt * (r.sp[2] - a1[2]);
                        const double dx = Bx - Ax, dy = By - Ay, dz = Bz - Az;
                        // segment_3d/mod.rs:41-48
                        const double c = ceil(fmax(fabs(dx), fabs(dy)));
                        uint32_t h;
                        if (c != c) h = 1u;
                        else if (c > 262142.0) h = 262144u;
                        else h = (uint32_t)c + 1u;
                        const uint32_t hres = h < 2u ? 2u : h;
                        int jlo = 1, jhi = 0;
                        if (hres > 262143u) atomicOr(a.flags, 1u << 0);
                        else sample_range_fast(Ax, dx, hres, !(seqf >> 31), X0, X1, jlo, jhi);
                        gA[0][lane] = Ax; gA[1][lane] = Ay; gA[2][lane] = Az;
                        gD[0][lane] = dx; gD[1][lane] = dy; gD[2][lane] = dz;
                        gInv[lane] = 1.0 / (double)hres;
                        gKey[lane] = ((unsigned long long)(seqf & 0x7FFFFFFFu) << kKeyTriShift) |
                                     ((unsigned long long)half << kKeyHalfShift) |
                                     ((unsigned long long)i << kKeyIShift);
                        gJlo[lane] = jlo;
                        if (jhi >= jlo) my_cnt = (uint32_t)(jhi - jlo + 1);
                        cnt_seg += 1;
                        cnt_eval += my_cnt;
                    }
                    // wave exclusive scan of the per-segment sample counts
                    uint32_t inc = my_cnt;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        uint32_t t = __shfl_up(inc, d, 64);
                        if (lane >= d) inc += t;
                    }
                    uint32_t n_samp = __shfl(inc, 63, 64);
                    if (a.debug == 3) n_samp = 0;

                    const uint32_t my_pre = inc - my_cnt;  // first flattened sample of my segment
                    gPre[lane] = my_pre;

                    // ---- (c) windows of kOwnCap flattened samples
                    for (uint32_t w0 = 0; w0 < n_samp; w0 += (uint32_t)kOwnCap) {
                        const uint32_t w1 = min(n_samp, w0 + (uint32_t)kOwnCap);
                        wave_lds_sync();  // previous window's readers are done with sOwn
                        {   // owner table: the segment's lane writes its index at each of its samples
                            uint32_t q0 = my_pre > w0 ? my_pre : w0;
                            uint32_t q1 = my_pre + my_cnt < w1 ? my_pre + my_cnt : w1;
                            for (uint32_t q = q0; q < q1; ++q) sOwn[q - w0] = (unsigned char)lane;
                        }
                        wave_lds_sync();
                        // kUnroll samples per lane per step, staged so that each LDS round trip
                        // (owner, record, depth min, depth read, key) is paid once per step.
                        for (uint32_t qb = w0 + (uint32_t)lane; qb < w1; qb += (uint32_t)(64 * kUnroll)) {
                            cnt_steps += 1;
                            uint32_t sg[kUnroll];
                            bool live[kUnroll];
#pragma unroll
                            for (int u = 0; u < kUnroll; ++u) {
                                const uint32_t q = qb + (uint32_t)(u * 64);
                                live[u] = q < w1;
                                sg[u] = live[u] ? sOwn[q - w0] : 0u;
                            }
                            int lp[kUnroll];
                            unsigned long long enc[kUnroll], key[kUnroll];
                            bool isnan_d[kUnroll];
#pragma unroll
                            for (int u = 0; u < kUnroll; ++u) {
                                const uint32_t q = qb + (uint32_t)(u * 64);
                                const uint32_t g = sg[u];
                                const int j = gJlo[g] + (int)(q - gPre[g]);
                                const double sp = (double)j * gInv[g];
                                const double x = gA[0][g] + sp * gD[0][g];
                                const double y = gA[1][g] + sp * gD[1][g];
                                const double z = gA[2][g] + sp * gD[2][g];
                                const int lx = tile_coord(x, px0, a.W, TW);
                                const int ly = tile_coord(y, py0, y_limit, TH);
                                live[u] = live[u] && lx >= 0 && ly >= 0 && py0 + ly >= a.y_lo;
                                lp[u] = live[u] ? ly * TW + lx : 0;
                                const double dist = -z;
                                isnan_d[u] = dist != dist;
                                enc[u] = encode_depth(dist);
                                key[u] = gKey[g] | (unsigned long long)j;
                                if (phase == 0 && live[u]) {
                                    cnt_inb += 1;
                                    if (isnan_d[u]) sNan = 1;
                                    else if (dist == -__builtin_huge_val()) atomicOr(a.flags, 1u << 3);
                                }
                            }
                            if (phase == 0) {
                                unsigned long long old[kUnroll], cur[kUnroll];
#pragma unroll
                                for (int u = 0; u < kUnroll; ++u) {
                                    live[u] = live[u] && !isnan_d[u];
                                    if (live[u]) old[u] = atomicMin(&myDepth[lp[u]], enc[u]);
                                }
#pragma unroll
                                for (int u = 0; u < kUnroll; ++u)
                                    if (live[u])
                                        cur[u] = __hip_atomic_load(&myDepth[lp[u]], __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                                // resets of every sample of the step come before any key min
#pragma unroll
                                for (int u = 0; u < kUnroll; ++u) {
                                    live[u] = live[u] && enc[u] == cur[u];
                                    if (live[u] && old[u] > cur[u])  // this sample lowered the pixel
                                        __hip_atomic_store(&myKey[lp[u]], kKeyNone, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
#pragma unroll
                                for (int u = 0; u < kUnroll; ++u)
                                    if (live[u]) atomicMin(&myKey[lp[u]], key[u]);
                            } else if (phase == 2) {
#pragma unroll
                                for (int u = 0; u < kUnroll; ++u)  // plane-0 depth reused: first non-NaN key
                                    if (live[u] && !isnan_d[u]) atomicMin(&myDepth[lp[u]], key[u]);
                            } else {
#pragma unroll
                                for (int u = 0; u < kUnroll; ++u)
                                    if (live[u] && isnan_d[u] && key[u] < myDepth[lp[u]]) atomicMin(&myKey[lp[u]], key[u]);
                            }
                        }
                    }
                }
            }
            __syncthreads();  // all samples of this phase have landed
            if (phase == 0) {
                // merge the private planes: lexicographic (depth, key) minimum per pixel -> plane 0
                for (int p = tid; p < kPix; p += kTileThreads) {
                    unsigned long long d = pDepth[0][p], k = pKey[0][p];
#pragma unroll
                    for (int w = 1; w < kTileWaves; ++w) {
                        const unsigned long long dw = pDepth[w][p], kw = pKey[w][p];
                        if (dw < d || (dw == d && kw < k)) { d = dw; k = kw; }
                    }
                    pKey[0][p] = k;
                    if (sNan) pDepth[0][p] = kKeyNone;
                }
                if (!sNan) break;
                phase = 2;
            } else if (phase == 2) {
                phase = 3;
            } else {
                break;
            }
        }
        __syncthreads();

        if (a.debug == 8 && tid == 0) {  // diagnostic: slowest tile (100 MHz ticks) and its shape
            unsigned long long dt = (unsigned long long)((long long)wall_clock64() - t_start);
            unsigned long long packed = (dt << 40) | ((unsigned long long)n_tri << 24) | (unsigned long long)tile;
            atomicMax(&a.counters[2], packed);
            atomicAdd(&a.counters[3], dt);
        }
        // ---- write the tile's winner keys (rows of TW keys: 128 contiguous bytes and more)
        for (int p = tid; p < kPix; p += kTileThreads) {
            const int x = px0 + (p & (TW - 1)), y = py0 + p / TW;
            if (x < a.W && y < a.H) a.wkey[(size_t)y * a.W + x] = pKey[0][p];
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        cnt_inb += __shfl_down(cnt_inb, d, 64);
        cnt_eval += __shfl_down(cnt_eval, d, 64);
        cnt_seg += __shfl_down(cnt_seg, d, 64);
        cnt_steps += __shfl_down(cnt_steps, d, 64);
    }
    if (lane == 0 && cnt_inb) {
        unsigned long long* slot = a.counters + (size_t)((blockIdx.x * kTileWaves + wv) % kCounterSlots) * 8;
        atomicAdd(&slot[1], cnt_inb);
        atomicAdd(&slot[5], cnt_eval);   // sample evaluations incl. pruning slack
        atomicAdd(&slot[6], cnt_seg);    // scanline segments built
        atomicAdd(&slot[7], cnt_steps);  // lane-steps of the sample loop
    }
}

template __global__ void k_tile_raster<16, 16>(TileArgs);
template __global__ void k_tile_raster<32, 16>(TileArgs);
template __global__ void k_tile_raster<32, 32>(TileArgs);
template __global__ void k_tile_raster<64, 32>(TileArgs);

}  // namespace tr
