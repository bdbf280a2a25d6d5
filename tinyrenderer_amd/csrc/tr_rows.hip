// Row raster (stage 3 of BASELINE.json north_star, third implementation): no global atomics.
//
// The reference's scanlines (model_renderer/mod.rs:376-389) run between two edges at the same
// parameter t, and the split point has the middle vertex's y (mod.rs:435-441), so a scanline is
// horizontal up to rounding: all its samples land in one pixel row (two when y sits on a
// rounding boundary).  Scanline records (ScanRec, one per slot of the frame's slot sequence, see
// tr_kernels.hip) are therefore binned by (pixel row, 256-pixel column segment); one wave owns a
// bin, keeps the bin's depth and winner planes in LDS and resolves depth minimum and first
// winner exactly in a single enumeration (segment_3d/mod.rs:19-115 sample order is kept by the
// order key).  Binning is conservative and every sample is re-tested with the exact pixel
// rounding, so the result is identical to enumerating everything in order.
#include "tr_device.h"

namespace tr {

constexpr int kSegW = 256;  // pixels per bin

struct RowArgs {
    const uint32_t* half_off;  // exclusive prefix of slots per half (+ total at [n_half])
    const TriRaster* rast;
    uint32_t n_half;           // 2 * triangles of the frame
    ScanRec* rec;              // one record per slot
    uint32_t rec_cap;
    uint32_t* bin_cnt;         // records per bin while counting, fill cursor afterwards
    const uint32_t* bin_off;   // exclusive prefix of bin_cnt (k_tile_scan); [n_padded] total, [n_padded + 1] non-empty bins
    const uint32_t* bin_list;  // ids of the non-empty bins
    uint32_t n_padded;
    uint32_t* refs;            // slot ids, grouped by bin
    uint32_t ref_cap;
    unsigned long long* wkey;  // winner key plane, row-major; only pixels that received a sample are written
    uint8_t* blk_touched;      // per shade block, at [2 * block]
    int blocks_x;
    int W, H, nseg;
    int y_lo, y_hi;
    unsigned long long* counters;  // kCounterSlots x 8: [0] samples, [1] fragments, [2] scanlines
    uint32_t* flags;
    unsigned long long* trace;  // TR_RASTER_TRACE: 4 words per bin visit (t_begin, t_records, t_end in 10 ns ticks,
    uint32_t trace_cap;         // samples | records << 32)
};

// Bins a record's samples can land in: rows [r0, r1] x segments [s0, s1] (empty if r0 > r1 or
// s0 > s1).  A sample coordinate is a + (j * (1/hres)) * d with the factor in [0, 1 + 2^-52]; a
// non-finite coordinate rounds to pixel 0 (NaN) or leaves the frame (inf): row / segment 0 is
// added for those.
__device__ __forceinline__ void pixel_span(double a, double d, int lo_clip, int hi_clip, int& p0, int& p1) {
    const double b = a + d;
    const bool regular = fabs(a) < 1073741824.0 && fabs(b) < 1073741824.0;  // 2^30: margin >> rounding error
    if (regular) {
        const double lo = fmin(a, b) - kPruneMargin, hi = fmax(a, b) + kPruneMargin;
        double f0 = ceil(lo - 0.5), f1 = floor(hi + 0.5);
        if (f0 < (double)lo_clip) f0 = (double)lo_clip;
        if (f1 > (double)hi_clip) f1 = (double)hi_clip;
        if (f0 > f1) { p0 = 1; p1 = 0; return; }
        p0 = (int)f0;
        p1 = (int)f1;
    } else {
        // NaN -> 0, +-inf -> outside, huge finite values sweep the axis in steps of many pixels
        const bool finite = fabs(a) < __builtin_huge_val() && fabs(b) < __builtin_huge_val();
        if (finite) { p0 = lo_clip; p1 = hi_clip; }  // anywhere
        else if (lo_clip == 0) { p0 = 0; p1 = 0; }
        else { p0 = 1; p1 = 0; }
    }
}

__device__ __forceinline__ void record_bins(const ScanRec& r, const RowArgs& a, int& r0, int& r1, int& s0, int& s1) {
    pixel_span(r.ay, r.dy, a.y_lo, a.y_hi - 1, r0, r1);
    int c0, c1;
    pixel_span(r.ax, r.dx, 0, a.W - 1, c0, c1);
    if (c0 > c1) { s0 = 1; s1 = 0; return; }
    s0 = c0 / kSegW;
    s1 = c1 / kSegW;
}

// Record of slot g, which lies in half `lo` (same arithmetic as the global raster's record_of).
__device__ __forceinline__ ScanRec row_record_of(const RowArgs& a, uint32_t lo, uint32_t g) {
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

// One thread per slot.  FILL = false: build the record, store it, count it in its bins (and in
// the frame counters).  FILL = true (after the bin prefix): append the slot id to its bins.
constexpr int kRowHalfCache = 1024;
template <bool FILL>
__global__ __launch_bounds__(256) void k_row_bin(RowArgs a) {
    __shared__ uint32_t sOff[FILL ? 1 : kRowHalfCache + 1];
    __shared__ uint32_t sEnds[2];
    const uint32_t s_end = a.half_off[a.n_half];
    if (s_end > a.rec_cap) {  // TR_FLAG_BIN_OVERFLOW: the host grows the record buffer and redoes the resolve
        if (!FILL && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.flags, 1u << 6);
        return;
    }
    unsigned long long n_scan = 0, n_samp = 0;
    for (unsigned long long b0 = (unsigned long long)blockIdx.x * 256ull; b0 < s_end;
         b0 += (unsigned long long)gridDim.x * 256ull) {
        const uint32_t g_first = (uint32_t)b0;
        const uint32_t g_last = (uint32_t)(b0 + 255ull < s_end ? b0 + 255ull : (unsigned long long)s_end - 1ull);
        const uint32_t g = g_first + threadIdx.x;
        ScanRec r;
        r.keyh = 0ull;
        if (!FILL) {
            // two waves locate the halves of the block's first and last slot (64 probes per round),
            // the prefix entries between them go to LDS, each thread finishes its search there
            const int w = threadIdx.x >> 6;
            if (w < 2) {
                uint32_t lo = 0, hi = a.n_half;
                const uint32_t target = w == 0 ? g_first : g_last;
                const uint32_t lane = threadIdx.x & 63u;
                while (hi - lo > 1u) {
                    const uint32_t step = (hi - lo + 63u) / 64u;
                    const uint32_t idx = lo + (lane + 1u) * step;
                    const bool le = idx < hi && a.half_off[idx] <= target;
                    const uint32_t n = (uint32_t)__popcll(__ballot(le));
                    const uint32_t nlo = lo + n * step, nhi = nlo + step;
                    lo = nlo;
                    hi = nhi < hi ? nhi : hi;
                }
                if (lane == 0) sEnds[w] = lo;
            }
            __syncthreads();
            const uint32_t e_first = sEnds[0], e_last = sEnds[1];
            const uint32_t n_e = e_last - e_first + 1u;
            const bool cached = n_e <= (uint32_t)kRowHalfCache;
            if (cached)
                for (uint32_t t = threadIdx.x; t <= n_e; t += 256u) sOff[t] = a.half_off[e_first + t];
            __syncthreads();
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
                    uint32_t l = e_first, h = e_last + 1u;
                    while (h - l > 1u) {
                        uint32_t mid = (l + h) >> 1;
                        if (a.half_off[mid] <= g) l = mid; else h = mid;
                    }
                    lo = l;
                }
                r = row_record_of(a, lo, g);
                a.rec[g] = r;
                const uint32_t hres = (uint32_t)(r.keyh & kRecHresMask);
                if (hres) {
                    n_scan += 1;
                    n_samp += hres + 1u;
                }
            }
        } else if (g <= g_last) {
            r = a.rec[g];
        }
        if (g <= g_last && (r.keyh & kRecHresMask)) {
            int r0, r1, s0, s1;
            record_bins(r, a, r0, r1, s0, s1);
            if (s0 <= s1) {
                for (int row = r0; row <= r1; ++row)
                    for (int s = s0; s <= s1; ++s) {
                        const uint32_t b = (uint32_t)row * (uint32_t)a.nseg + (uint32_t)s;
                        if (!FILL) {
                            atomicAdd(&a.bin_cnt[b], 1u);
                        } else {
                            const uint32_t pos = a.bin_off[b] + atomicAdd(&a.bin_cnt[b], 1u);
                            if (pos < a.ref_cap) a.refs[pos] = g;
                        }
                    }
            }
        }
        if (!FILL) __syncthreads();  // sOff / sEnds are rewritten by the next round
    }
    if (!FILL) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            n_scan += __shfl_down(n_scan, d, 64);
            n_samp += __shfl_down(n_samp, d, 64);
        }
        if ((threadIdx.x & 63) == 0 && n_scan) {
            unsigned long long* slot = a.counters + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) % kCounterSlots) * 8;
            atomicAdd(&slot[0], n_samp);
            atomicAdd(&slot[2], n_scan);
        }
    }
}
template __global__ void k_row_bin<false>(RowArgs);
template __global__ void k_row_bin<true>(RowArgs);

__device__ __forceinline__ void row_lds_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving
    // LDS accesses across the hand-off between lanes of the same wave.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave per non-empty bin (persistent grid over the bin list).  The planes belong to this
// wave alone, so its LDS instructions execute in program order, which keeps depth minimum AND
// winner key exact in one enumeration:
//   old = ds_min_rtn(depth[p], enc);  cur = depth[p]           (cur = minimum after this step)
//   the lane with enc == cur and old > cur lowered the pixel in this step: it resets key[p];
//   every lane with enc == cur then does ds_min(key[p], key)   (ties keep the smallest order key)
// A NaN depth wins its pixel iff it is the first sample to land on it (the reference's
// `current_z == -inf || current_z > dist` never replaces a NaN): bins that saw one re-enumerate
// twice (first key of any sample per pixel, then the NaN samples that hold it).
constexpr int kRowThreads = 64;
constexpr int kRowUnroll = 4;

__global__ __launch_bounds__(kRowThreads) void k_row_raster(RowArgs a) {
    __shared__ unsigned long long sDepth[kSegW];
    __shared__ unsigned long long sKey[kSegW];
    __shared__ unsigned long long sFirst[kSegW];
    __shared__ uint4 sRec4[64 * 4];
    __shared__ uint32_t sPre[65];
    const double* sRecD = reinterpret_cast<const double*>(sRec4);
    const unsigned long long* sRecU = reinterpret_cast<const unsigned long long*>(sRec4);
    const int lane = threadIdx.x;
    if (*a.flags & (1u << 6)) return;  // overflow: the host redoes the frame with larger buffers
    const uint32_t n_bins = a.bin_off[a.n_padded + 1];
    unsigned long long cnt_frag = 0;

    for (uint32_t bi = blockIdx.x; bi < n_bins; bi += gridDim.x) {
        const uint32_t b = a.bin_list[bi];
        const int row = (int)(b / (uint32_t)a.nseg);
        const int x0 = (int)(b % (uint32_t)a.nseg) * kSegW;
        const uint32_t off = a.bin_off[b];
        const uint32_t n = a.bin_off[b + 1] - off;
        unsigned long long tr_t0 = 0, tr_t1 = 0, tr_samples = 0;
        if (a.trace && lane == 0) tr_t0 = wall_clock64();
        if (lane == 0) a.bin_cnt[b] = 0;  // the fill cursor: the next frame counts from zero
        row_lds_sync();  // the previous bin's flush is done with the planes
        for (int p = lane; p < kSegW; p += kRowThreads) {
            sDepth[p] = kKeyNone;
            sKey[p] = kKeyNone;
        }
        bool saw_nan = false;
        for (int phase = 1; phase <= 3; ++phase) {
            if (phase == 2) {
                if (!__any(saw_nan)) break;
                row_lds_sync();
                for (int p = lane; p < kSegW; p += kRowThreads) sFirst[p] = kKeyNone;
            }
            for (uint32_t r0 = 0; r0 < n; r0 += 64u) {
                row_lds_sync();  // readers of the previous round's records are done
                // ---- lane = one record of the bin
                uint32_t cnt = 0;
                {
                    uint4 v[4] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
                    if (r0 + (uint32_t)lane < n) {
                        const uint32_t g = a.refs[off + r0 + (uint32_t)lane];
                        const uint4* src = reinterpret_cast<const uint4*>(a.rec + g);
                        v[0] = src[0]; v[1] = src[1]; v[2] = src[2]; v[3] = src[3];
                        const uint32_t hres = v[3].z & (uint32_t)kRecHresMask;  // low word of keyh
                        cnt = hres ? hres + 1u : 0u;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) sRec4[lane * 4 + q] = v[q];
                }
                uint32_t inc = cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    uint32_t t = __shfl_up(inc, d, 64);
                    if (lane >= d) inc += t;
                }
                sPre[lane + 1] = inc;
                if (lane == 0) sPre[0] = 0;
                row_lds_sync();
                const uint32_t total = sPre[64];
                if (a.trace && lane == 0 && phase == 1) { if (r0 == 0) tr_t1 = wall_clock64(); tr_samples += total; }
                // ---- lanes own samples; kRowUnroll per lane per step so each LDS round trip is paid once
                for (uint32_t k0 = (uint32_t)lane; k0 < total; k0 += (uint32_t)(kRowThreads * kRowUnroll)) {
                    int lp[kRowUnroll];
                    bool hit[kRowUnroll], isn[kRowUnroll];
                    unsigned long long enc[kRowUnroll], key[kRowUnroll];
#pragma unroll
                    for (int u = 0; u < kRowUnroll; ++u) {
                        const uint32_t k = k0 + (uint32_t)(u * kRowThreads);
                        hit[u] = false; isn[u] = false; lp[u] = 0; enc[u] = 0; key[u] = 0;
                        if (k < total) {
                            uint32_t sidx = 0;
#pragma unroll
                            for (uint32_t step = 32; step; step >>= 1)
                                if (sPre[sidx + step] <= k) sidx += step;
                            const uint32_t j = k - sPre[sidx];
                            const double* r = sRecD + sidx * 8;
                            const double sp = (double)j * r[6];
                            const double x = r[0] + sp * r[3];
                            const double y = r[1] + sp * r[4];
                            const double z = r[2] + sp * r[5];
                            const int ix = pixel_coord(x, a.W);
                            const int iy = pixel_coord(y, a.H);
                            if (iy == row && ix >= x0 && ix < x0 + kSegW) {
                                hit[u] = true;
                                lp[u] = ix - x0;
                                const double dist = -z;
                                isn[u] = dist != dist;
                                key[u] = (sRecU[sidx * 8 + 7] & ~kRecHresMask) | (unsigned long long)j;
                                if (!isn[u]) {
                                    enc[u] = encode_depth(dist);
                                    if (phase == 1 && dist == -__builtin_huge_val()) atomicOr(a.flags, 1u << 3);
                                }
                                if (phase == 1) cnt_frag += 1;
                            }
                        }
                    }
                    if (phase == 1) {
                        unsigned long long old[kRowUnroll], cur[kRowUnroll];
                        bool live[kRowUnroll];
#pragma unroll
                        for (int u = 0; u < kRowUnroll; ++u) {
                            saw_nan |= hit[u] && isn[u];
                            live[u] = hit[u] && !isn[u];
                            old[u] = 0; cur[u] = 0;
                            if (live[u]) old[u] = atomicMin(&sDepth[lp[u]], enc[u]);
                        }
#pragma unroll
                        for (int u = 0; u < kRowUnroll; ++u)
                            if (live[u])
                                cur[u] = __hip_atomic_load(&sDepth[lp[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        // resets of every sample of the step come before any key min
#pragma unroll
                        for (int u = 0; u < kRowUnroll; ++u) {
                            live[u] = live[u] && enc[u] == cur[u];
                            if (live[u] && old[u] > cur[u])  // this sample lowered the pixel
                                __hip_atomic_store(&sKey[lp[u]], kKeyNone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
#pragma unroll
                        for (int u = 0; u < kRowUnroll; ++u)
                            if (live[u]) atomicMin(&sKey[lp[u]], key[u]);
                    } else if (phase == 2) {
#pragma unroll
                        for (int u = 0; u < kRowUnroll; ++u)
                            if (hit[u]) atomicMin(&sFirst[lp[u]], key[u]);
                    } else {
#pragma unroll
                        for (int u = 0; u < kRowUnroll; ++u)
                            if (hit[u] && isn[u] && key[u] == sFirst[lp[u]])
                                __hip_atomic_store(&sKey[lp[u]], key[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        row_lds_sync();
        // ---- flush: pixels that received a sample get their winner key
        for (int p = lane; p < kSegW; p += kRowThreads) {
            const unsigned long long k = sKey[p];
            const int x = x0 + p;
            if (k != kKeyNone && x < a.W) {
                a.wkey[(size_t)row * (size_t)a.W + (size_t)x] = k;
                a.blk_touched[2u * ((uint32_t)(row / kShadeBH) * (uint32_t)a.blocks_x + (uint32_t)(x / kShadeBW))] = 1;
            }
        }
        if (a.trace && lane == 0 && bi < a.trace_cap) {
            unsigned long long* t = a.trace + 4 * (size_t)bi;
            t[0] = tr_t0; t[1] = tr_t1; t[2] = wall_clock64();
            t[3] = tr_samples | ((unsigned long long)n << 32);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt_frag += __shfl_down(cnt_frag, d, 64);
    if (lane == 0 && cnt_frag) atomicAdd(&a.counters[(size_t)(blockIdx.x % kCounterSlots) * 8 + 1], cnt_frag);
}

}  // namespace tr
