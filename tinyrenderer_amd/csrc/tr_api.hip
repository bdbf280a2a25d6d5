// C ABI of the gfx950 rasterizer (include/tinyrenderer_hip.h).  Host-side orchestration
// only: every per-vertex / per-sample / per-pixel operation runs in tr_kernels.hip.
// There is no CPU fallback: without a usable GPU tr_create fails with TR_ERR_NO_DEVICE.
#include "../../include/tinyrenderer_hip.h"
#include "tr_kernels.hip"
#include "tr_tile.hip"
#include "tr_rows.hip"

#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>
#include <new>
#include <string>
#include <vector>

using namespace tr;

struct tr_texture {
    uint32_t* dev = nullptr;
    int w = 0, h = 0;
};

struct tr_shadow_map {  // extension: a light-view z plane (tr_shadow_map_create / _update)
    double* dev = nullptr;
    int w = 0, h = 0;
    int device = 0;
    // The snapshot is a copy on the light context's stream; no host synchronisation: a context whose
    // frame uses the map makes its stream wait for `ready` before its shade pass and records an event
    // of its own behind that pass, which the next tr_shadow_map_update waits for before it overwrites.
    hipEvent_t ready = nullptr;
    std::mutex lock;  // (consumers may live on different host threads)
    std::vector<std::pair<const tr_context*, hipEvent_t>> consumed;
};

struct tr_mesh {
    double* pos = nullptr;
    double* nrm = nullptr;
    double* uvw = nullptr;
    uint32_t* idx = nullptr;
    double* bounds = nullptr;  // meshes of 65 536 triangles or more: a box per 64 consecutive triangles (k_mesh_bounds)
    int64_t n_pos = 0, n_nrm = 0, n_uvw = 0, n_tris = 0;
    bool any_uv = false;
    double box[6] = {0, 0, 0, 0, 0, 0};  // min xyz, max xyz of the positions (host, at creation)
    bool box_ok = false;                 // false: a position is not finite (the frame's screen box is then the whole frame)
};

// Pixel rectangle [x0, x1) x [y0, y1)
struct PixBox {
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    bool empty() const { return x0 >= x1 || y0 >= y1; }
    void add(const PixBox& o) {
        if (o.empty()) return;
        if (empty()) { *this = o; return; }
        x0 = std::min(x0, o.x0); y0 = std::min(y0, o.y0); x1 = std::max(x1, o.x1); y1 = std::max(y1, o.y1);
    }
};

enum { ST_CLEAR = 0, ST_TRANSFORM, ST_SETUP, ST_DEPTH, ST_WINNER, ST_SHADE, ST_COUNT };

struct tr_context {
    int device = 0;
    int W = 0, H = 0;
    int y_lo = 0, y_hi = 0;
    int shaded_lo = 0, shaded_hi = 0;  // rows the last shade launch covered (a band context shades its band's rows only)
    PixBox frame_box;   // where this frame's draws can land a sample (projected mesh boxes, a margin added); tr_clear empties it
    uint32_t last_shade_grid[3] = {0, 0, 0};  // workgroups of the last shade launch (tr_get_shade_launch)
    PixBox paint_box;   // where the colour plane may differ from the background after the last shade launch
    bool origin_maybe_painted = false;  // ... and pixel (0, 0) beside it: the last boxed launch's frame had triangles (NaN samples land there)
    bool depth_split = true;  // TR_DEPTH_SPLIT: draws that use k_small / k_mid split their triangles at the depth of the mesh box's centre
    bool split_any = false;   // a draw of this frame did
    bool plane_tiles = true;   // TR_PLANE_TILES: frames whose first draw uses k_small keep their planes in 4 x 2 tiles
    bool frame_tiled = false;  // this frame does
    bool colour_exact = false;  // TR_SHADE_EXACT / tr_set_colour_exact: k_shade's colour arithmetic operation for operation (0 LSB) instead of fused (+-1 LSB)
    bool shade_box = true;  // TR_SHADE_BOX: the cheap resolve launches the blocks of frame_box + paint_box only
    // batched views (tr_create_views / tr_render_views): the planes hold max_views slices of W x H pixels
    int max_views = 1;
    int n_views = 1;                 // slices the current frame covers (1 unless the frame is a tr_render_views batch)
    int sel_view = 0;                // slice the tr_read_* calls and tr_device_ptr_rgba8 refer to (tr_select_view)
    uint32_t view_tris = 0;          // triangles per view of the current batch (0: the frame is not a batch)
    size_t npix_all = 0;             // W * H * max_views
    uint32_t blk_per_view = 0, blk_rows_per_view = 0;
    ViewDev* view_table = nullptr;   // device, view_cap entries (one per (view, mesh) of a batch)
    size_t view_cap = 0;
    uint32_t view_meshes = 1;        // meshes (draws) per view of the current batch
    ViewDev* view_stage[2] = {nullptr, nullptr};  // page-locked host copies, used in turn
    hipEvent_t view_ev[2] = {nullptr, nullptr};   // "the upload from view_stage[i] has completed"
    int view_turn = 0;
    hipStream_t stream = nullptr;
    unsigned long long* planes = nullptr;  // depth[W*H] then wkey[W*H]
    unsigned long long* depth = nullptr;
    unsigned long long* wkey = nullptr;
    uint32_t* rgba8 = nullptr;
    double* z_out = nullptr;
    uint32_t* tri_out = nullptr;
    double* rgba_f64 = nullptr;
    unsigned long long* key_out = nullptr;
    // global raster: what the depth / winner planes currently hold
    bool planes_clean = false;      // every entry is "none"
    uint32_t depth_upto = 0;        // triangles whose depth pass is reflected in the depth plane
    bool keys_valid = false;        // winner pass done for all of them
    bool wkey_dirty = false;        // winner plane holds keys of an earlier winner pass
    bool cov_dirty = false;         // covered-pixel counters hold a previous resolve of this frame
    TriRaster* rast = nullptr;
    TriShade* shade = nullptr;
    uint32_t* half_cnt = nullptr;
    uint32_t* half_off = nullptr;
    ScanRec* rec = nullptr;          // scanline records of the frame's first rec_cap slots
    uint32_t rec_cap = 1u << 20;     // 64 MB; slots beyond are rebuilt by every raster pass (TR_SCAN_RECORDS)
    bool rec_auto = true;            // grow with the first draw of a frame unless TR_SCAN_RECORDS pins the size
    uint32_t* slots_seen = nullptr;  // page-locked word the winner pass leaves a frame's slot total in (never waited for)
    uint8_t* chunk_live = nullptr;
    uint32_t* flip_buf = nullptr;    // tr_read_rgba8(flip): the presented image, flipped on the device
    uint8_t* small_flag = nullptr;   // per triangle: rastered by k_small (tri_cap bytes)
    uint8_t* half_live = nullptr;    // small-triangle path: per half, may hold a winner (2 x tri_cap bytes)
    uint8_t* small_grp = nullptr;    // per 32 triangles: some triangle of the group is small (tri_cap / 32 + 1 bytes)
    bool debug_prefix_stall = false; // TR_DEBUG_PREFIX_STALL (tests): force the bounded poll of the fused set-up launch to give up
    uint8_t* cluster_cull = nullptr; // band contexts: a byte per 64-triangle cluster of the frame's big draws (k_cull_clusters)
    size_t cull_cap = 0, cull_used = 0;
    uint8_t* need_shade = nullptr;   // deferred shade records: per triangle, flagged by the winner pass (tri_cap bytes)
    uint32_t defer_tris = 65536;     // draws with at least this many triangles defer their shade records (TR_DEFER_SHADE)
    struct Deferred { const tr_mesh* mesh; SetupArgs sa; };
    std::vector<Deferred> deferred;  // such draws since tr_clear: k_build_shade runs for each before the shade pass
    bool frame_lost_mesh = false;    // the mesh of one of them was destroyed: the frame cannot be shaded again
    bool counted = false;            // in g_live_contexts
    int raster_threads = 0;          // TR_RASTER_THREADS=128|256 pins the raster workgroup size (0: by live contexts)
    bool small_path = true;          // TR_SMALL_TRIS=0 disables k_small
    uint32_t small_draw_tris = 65536; // draws with at least this many triangles use it (TR_SMALL_DRAW)
    uint32_t small_res = kSmallRes;  // TR_SMALL_RES
    bool small_any = false;          // some draw since tr_clear flagged small triangles
    bool mid_any = false;            // ... or mid-size ones (k_mid)
    int mid_path = -1;               // TR_MID_TRIS: 1 every draw may flag mid triangles, 0 none, -1 (default) launches of at
                                     // least 32 768 triangles that cover fewer than mid_px pixels of the frame on average
    uint32_t mid_px = 1024;          // TR_MID_PX
    uint32_t mid_samples = 4096;     // TR_MID_SAMPLES: largest estimated sample count of a half that goes to k_mid
    // row raster (mode 2)
    uint32_t* bin_cnt = nullptr;
    uint32_t* bin_off = nullptr;
    uint32_t* bin_list = nullptr;
    uint32_t* refs = nullptr;
    uint32_t n_bins_padded = 0;
    uint32_t ref_cap = 2u << 20;
    int nseg = 0;
    bool bin_dirty = false;          // row raster: a resolve ended with its overflow unsettled (fill cursors left behind)
    uint8_t* blk_flags = nullptr;    // per shade block [2 b]: touched this frame, [2 b + 1]: rgba8 not background
    uint32_t n_blk = 0;
    int blocks_x = 0;
    bool skip_clean = true;          // shade pass skips blocks that hold no key and are already background
    uint32_t counted_upto = 0;       // triangles whose samples are in the counters
    size_t tri_cap = 0;
    unsigned long long* tile_sum = nullptr;
    size_t tile_cap = 0;
    double* ndc = nullptr;
    double* nout = nullptr;
    size_t ndc_cap = 0, nout_cap = 0;
    DrawDev* draw_table = nullptr;
    DrawExt* ext_table = nullptr;    // extension shaders: allocated with the first tr_draw_ext that names one
    std::vector<tr_shadow_map*> frame_maps;  // shadow maps the draws since tr_clear refer to
    std::vector<const tr_texture*> frame_textures;  // textures the draws since tr_clear refer to
    bool frame_lost_texture = false; // one of them was destroyed: the frame cannot be shaded again
    bool frame_has_ext = false;      // some draw since tr_clear uses an extension: k_shade<true>
    DrawDev draw0;                   // host copy of the frame's first draw (a kernel argument of the shade pass)
    uint32_t draw_cap = 4096;
    unsigned long long* state_blk = nullptr;     // counters | cov_counters | flags, one allocation (current frame's half)
    // Per-frame state is double-buffered: the shade pass of a frame zeroes the OTHER half (the
    // previous frame's, dead since this frame's tr_clear), so the next tr_clear only swaps.
    unsigned long long* state_both = nullptr;    // 2 x kStateWords
    NanTable* nan_both = nullptr;                // 2
    int state_idx = 0;
    bool other_state_clean = false;
    unsigned long long* counters = nullptr;      // kCounterSlots x 8: [0] samples [1] fragments
    unsigned long long* cov_counters = nullptr;  // kCounterSlots x 8: [0] covered pixels
    uint32_t* flags = nullptr;
    // tile raster (mode 1)
    int mode = 0;                   // 0 = global-atomic raster (default: faster on MI355X so far), 1 = binned LDS tile raster
    int tiles_x = 0, tiles_y = 0;
    int tile_w = 16, tile_h = 16;   // TR_TILE=WxH: 16x16, 32x16, 32x32 or 64x32 (k_tile_raster<TW, TH>)
    int tw_shift = 4, th_shift = 4;
    uint32_t* tile_cnt = nullptr;   // triangles per tile / fill cursor
    uint32_t* tile_off = nullptr;   // n_tiles_padded + 4: offsets, total, number of non-empty tiles
    uint32_t* tile_list = nullptr;  // ids of non-empty tiles
    uint32_t* pairs = nullptr;      // tri_seq per (tile, slot)
    uint32_t pair_cap = 0;
    uint32_t n_tiles_padded = 0;    // tile arrays are padded to a multiple of 4*kTileScanThreads
    NanTable* nan = nullptr;
    uint32_t tri_total = 0;
    uint32_t draw_count = 0;
    uint32_t resolved_mask = 0;
    bool resolved = false;
    std::string err;
    // profiling
    bool precheck = true;          // plain-load filter in front of the depth atomics
#ifndef TR_BOX_DEFAULT
#define TR_BOX_DEFAULT 1
#endif
    bool raster_box = TR_BOX_DEFAULT != 0;  // k_raster depth pass: chunks reduce their samples per pixel in LDS first
    bool precheck_winner = false;  // ... and in front of the winner atomics
    bool skip_dead = true;         // winner pass skips chunks the depth pass found dead
    bool fused_front = true;       // small draws: transform + setup + scan in one launch (TR_FUSED_FRONT)
    uint32_t big_draw_tris = 65536;  // draws at least this large get their records from k_scanlines (TR_BIG_DRAW)
    uint32_t* ticket = nullptr;
    uint32_t scan_tag = 0;  // tag of the last fused k_setup launch (never 0 in a launch)
    bool prof = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev2 = nullptr, ev3 = nullptr;  // start / stop of a kernel launched with its own events (tr_profile_enable)
    bool prof_kernel = false;                 // the open profiling scope was timed by ev2 / ev3: its one kernel, as rocprofv3 times it
    double prof_ms[ST_COUNT] = {0};
    uint64_t prof_n[ST_COUNT] = {0};
};

static int resolve_frame(tr_context* c, uint32_t out_mask, bool colour);

namespace {

// Environment switches.  The release library reads the handful INTEGRATION.md documents (env_opt: capacities and path
// thresholds, none of which changes a result).  Everything else -- kernels with their memory operations taken out,
// per-chunk / per-wave timelines, forced poll failures, launch-shape experiments -- exists in the diagnostic build only
// (`make diag`: -DTR_DIAG -> libtinyrenderer_hip_diag.so), which the tools/ scripts and one test load by name.
inline const char* env_opt(const char* name) { return std::getenv(name); }
#ifdef TR_DIAG
inline const char* env_diag(const char* name) { return std::getenv(name); }
#else
inline const char* env_diag(const char*) { return nullptr; }
#endif

int fail(tr_context* c, int code, const std::string& msg) {
    if (c) {
        try { c->err = msg; }
        catch (...) { c->err.clear(); }  // (nothing may unwind through the C boundary)
    }
    return code;
}

int fail_hip(tr_context* c, hipError_t e, const char* what) {
    const int code = e == hipErrorOutOfMemory ? TR_ERR_OOM : TR_ERR_HIP;
    try { return fail(c, code, std::string(what) + ": " + hipGetErrorString(e)); }
    catch (...) { return code; }
}

#define TR_HIP(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail_hip(c, e_, #expr);                                                      \
    } while (0)

struct ProfScope {
    tr_context* c;
    int stage;
    ProfScope(tr_context* c_, int s) : c(c_), stage(s) {
        c->prof_kernel = false;
        if (c->prof) (void)hipEventRecord(c->ev0, c->stream);
    }
    ~ProfScope() {
        if (c->prof) {
            (void)hipEventRecord(c->ev1, c->stream);
            (void)hipEventSynchronize(c->ev1);
            float ms = 0.f;
            // (a stage that is ONE kernel is timed by that launch's own start / stop events: the kernel's duration as a
            // kernel trace reports it, without the 2-3 us between an event on the stream and the kernel's first wave)
            if (c->prof_kernel) (void)hipEventElapsedTime(&ms, c->ev2, c->ev3);
            else (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
            c->prof_ms[stage] += ms;
            c->prof_n[stage] += 1;
        }
        c->prof_kernel = false;
    }
};
// Launch of a stage's only kernel: with profiling on, through hipExtLaunchKernelGGL with the context's ev2 / ev3.
#define TR_LAUNCH_STAGE(c, kernel, grid, block, ...)                                                              \
    do {                                                                                                         \
        if ((c)->prof) {                                                                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, (c)->stream, (c)->ev2, (c)->ev3, 0, __VA_ARGS__);      \
            (c)->prof_kernel = true;                                                                             \
        } else {                                                                                                 \
            hipLaunchKernelGGL(kernel, grid, block, 0, (c)->stream, __VA_ARGS__);                                \
        }                                                                                                        \
    } while (0)

// nalgebra's 4x4 * 4x4 (one gemv per result column, left-to-right accumulation); used for the
// viewport matrix of src/model_renderer/mod.rs:314-318.
Mat4 host_mat4_mul(const Mat4& a, const Mat4& b) {
    Mat4 c;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r) {
            double acc = a.m[r] * b.m[j * 4];
            for (int k = 1; k < 4; ++k) acc = a.m[k * 4 + r] * b.m[j * 4 + k] + acc;
            c.m[j * 4 + r] = acc;
        }
    return c;
}

int ensure_tri_capacity(tr_context* c, size_t need) {
    if (need <= c->tri_cap) return TR_OK;
    size_t cap = c->tri_cap ? c->tri_cap : 16384;
    while (cap < need) cap *= 2;
    TriRaster* nr = nullptr;
    TriShade* ns = nullptr;
    uint32_t *ncnt = nullptr, *noff = nullptr;
    uint8_t *nlive = nullptr, *nsmall = nullptr, *nneed = nullptr, *ngrp = nullptr;
    TR_HIP(c, hipMalloc(&ngrp, cap / 32 + 1));
    TR_HIP(c, hipMemsetAsync(ngrp, 0, cap / 32 + 1, c->stream));

    TR_HIP(c, hipMalloc(&nneed, cap));
    TR_HIP(c, hipMalloc(&nlive, 2 * cap));
    TR_HIP(c, hipMalloc(&nsmall, cap));
    TR_HIP(c, hipMemsetAsync(nsmall, 0, cap, c->stream));
    TR_HIP(c, hipMalloc(&nr, cap * sizeof(TriRaster)));
    TR_HIP(c, hipMalloc(&ns, cap * sizeof(TriShade)));
    TR_HIP(c, hipMalloc(&ncnt, 2 * cap * sizeof(uint32_t)));
    TR_HIP(c, hipMalloc(&noff, (2 * cap + 1) * sizeof(uint32_t)));
    if (c->tri_total) {  // keep the draws already recorded this frame
        size_t n = c->tri_total;
        TR_HIP(c, hipMemcpyAsync(nr, c->rast, n * sizeof(TriRaster), hipMemcpyDeviceToDevice, c->stream));
        TR_HIP(c, hipMemcpyAsync(ns, c->shade, n * sizeof(TriShade), hipMemcpyDeviceToDevice, c->stream));
        TR_HIP(c, hipMemcpyAsync(ncnt, c->half_cnt, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        TR_HIP(c, hipMemcpyAsync(noff, c->half_off, (2 * n + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        TR_HIP(c, hipMemcpyAsync(nlive, c->half_live, 2 * n, hipMemcpyDeviceToDevice, c->stream));
        TR_HIP(c, hipMemcpyAsync(nsmall, c->small_flag, n, hipMemcpyDeviceToDevice, c->stream));
        TR_HIP(c, hipMemcpyAsync(nneed, c->need_shade, n, hipMemcpyDeviceToDevice, c->stream));
        TR_HIP(c, hipMemcpyAsync(ngrp, c->small_grp, n / 32 + 1, hipMemcpyDeviceToDevice, c->stream));
    } else if (c->half_off) {
        TR_HIP(c, hipMemcpyAsync(noff, c->half_off, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    } else {
        TR_HIP(c, hipMemsetAsync(noff, 0, sizeof(uint32_t), c->stream));
    }
    TR_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->rast); (void)hipFree(c->shade); (void)hipFree(c->half_cnt); (void)hipFree(c->half_off);
    (void)hipFree(c->half_live); (void)hipFree(c->small_flag); (void)hipFree(c->need_shade); (void)hipFree(c->small_grp);
    c->small_grp = ngrp;
    c->rast = nr; c->shade = ns; c->half_cnt = ncnt; c->half_off = noff; c->half_live = nlive; c->small_flag = nsmall;
    c->need_shade = nneed;
    c->tri_cap = cap;
    size_t tiles = (2 * cap + kScanTile - 1) / kScanTile;
    if (tiles > c->tile_cap) {
        (void)hipFree(c->tile_sum);
        c->tile_sum = nullptr;
        TR_HIP(c, hipMalloc(&c->tile_sum, tiles * sizeof(unsigned long long)));
        c->tile_cap = tiles;
    }
    return TR_OK;
}

// The per-(view, mesh) table of a batch of views and its two page-locked staging copies.
int ensure_view_capacity(tr_context* c, size_t need) {
    if (need <= c->view_cap) return TR_OK;
    if (c->stream) TR_HIP(c, hipStreamSynchronize(c->stream));  // (an upload from the old staging copies may be in flight)
    (void)hipFree(c->view_table);
    c->view_table = nullptr;
    for (int i = 0; i < 2; ++i) {
        if (c->view_stage[i]) (void)hipHostFree(c->view_stage[i]);
        c->view_stage[i] = nullptr;
    }
    c->view_cap = 0;
    TR_HIP(c, hipMalloc(&c->view_table, need * sizeof(ViewDev)));
    for (int i = 0; i < 2; ++i) TR_HIP(c, hipHostMalloc((void**)&c->view_stage[i], need * sizeof(ViewDev), hipHostMallocDefault));
    c->view_cap = need;
    return TR_OK;
}

int ensure_scratch(tr_context* c, double** p, size_t* cap, size_t need) {
    if (need <= *cap) return TR_OK;
    TR_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    TR_HIP(c, hipMalloc(p, need * sizeof(double)));
    *cap = need;
    return TR_OK;
}

constexpr uint32_t kStateWords = 2 * kCounterSlots * 8 + 8;  // u64 words of one half of the per-frame state

void use_state(tr_context* c, int idx) {
    c->state_idx = idx;
    c->state_blk = c->state_both + (size_t)idx * kStateWords;
    c->counters = c->state_blk;
    c->cov_counters = c->state_blk + kCounterSlots * 8;
    c->flags = (uint32_t*)(c->state_blk + 2 * kCounterSlots * 8);
    c->nan = c->nan_both + idx;
}

constexpr int kRasterWaves = 4096 * 4;  // waves of a raster launch, whatever the workgroup size
constexpr int kRowBinBlocks = 1024;
constexpr int kScanlineBlocks = 2048;
constexpr int kRowGrid = 256 * 15;  // waves: 15 per CU fit the LDS (10.3 KB each)
constexpr uint32_t kChunkLiveCap = 1u << 20;  // 64 M slots per frame get dead-chunk skipping

// (a wave per block of 1 << shift halves: single halves while there are fewer than a million, so that a batch of
// views fills the GPU; blocks of 16 for the meshes of millions of triangles, few of which are of this path)
uint32_t mid_shift_for(uint32_t n_halves) { return n_halves >= (4u << 20) ? 4u : (n_halves >= (1u << 20) ? 2u : 0u); }

// Tile raster: the instantiation for the context's tile size, and its dynamic LDS (the four waves' private depth + key planes).
using TileKernel = void (*)(TileArgs);
TileKernel tile_kernel(const tr_context* c) {
    if (c->tile_w == 64) return k_tile_raster<64, 32>;
    if (c->tile_w == 32 && c->tile_h == 32) return k_tile_raster<32, 32>;
    if (c->tile_w == 32) return k_tile_raster<32, 16>;
    return k_tile_raster<16, 16>;
}
size_t tile_lds_bytes(const tr_context* c) { return (size_t)kTileWaves * c->tile_w * c->tile_h * 2 * sizeof(unsigned long long); }

void fill_raster_args(tr_context* c, RasterArgs& ra, uint32_t h_lo, uint32_t h_hi) {
    ra.half_off = c->half_off;
    ra.rast = c->rast;
    ra.small_flag = c->small_flag;
    ra.small_grp = c->small_grp;
    ra.rec = c->rec;
    ra.rec_cap = c->rec_cap;
    ra.depth = c->depth;
    ra.wkey = c->wkey;
    ra.h_lo = h_lo;
    ra.h_hi = h_hi;
    ra.count_h = 2u * c->counted_upto;
    ra.W = c->W;
    ra.H = c->H;
    ra.y_lo = c->y_lo;
    ra.y_hi = c->y_hi;
    ra.counters = c->counters;
    ra.flags = c->flags;
    ra.nan = c->nan;
    ra.blk_touched = c->blk_flags;
    ra.blocks_x = c->blocks_x;
    ra.chunk_live = c->chunk_live;
    ra.live_cap = kChunkLiveCap;
    ra.use_live = 0;
    ra.rec_ready = 0;
    ra.phase = 0;
    ra.tiled = c->frame_tiled ? 1u : 0u;
    ra.clean_state = nullptr;
    ra.clean_words = 0;
    ra.clean_nan = nullptr;
    ra.need_shade = c->deferred.empty() ? nullptr : c->need_shade;
    ra.mid_shift = mid_shift_for(h_hi - h_lo);
    ra.box = c->raster_box ? 1u : 0u;
    ra.slots_seen = c->slots_seen;
    ra.view_tris = c->view_tris;
    ra.view_pix = (uint32_t)((size_t)c->W * c->H);
    ra.view_blk = c->blk_per_view;
    ra.debug = 0;
    ra.trace = nullptr;
    ra.trace_cap = 0;
}

// The fragment constants of one draw (src/main.rs:173-267) from its tr_draw_params.
void fill_draw_dev(DrawDev& d, const tr_draw_params* p) {
    std::memset(&d, 0, sizeof(d));
    d.tex = p->diffuse ? p->diffuse->dev : nullptr;
    d.tw = p->diffuse ? p->diffuse->w : 0;
    d.th = p->diffuse ? p->diffuse->h : 0;
    d.nm = p->normal_map ? p->normal_map->dev : nullptr;
    d.nw = p->normal_map ? p->normal_map->w : 0;
    d.nh = p->normal_map ? p->normal_map->h : 0;
    // src/main.rs:249-251: Vector3::new(camera_pos).normalize(); norm^2 = (x*x + y*y) + z*z
    {
        double x = p->light_dir[0], y = p->light_dir[1], z = p->light_dir[2];
        double n = std::sqrt(0.0 + ((x * x + y * y) + z * z));
        d.light[0] = x / n; d.light[1] = y / n; d.light[2] = z / n;
    }
    d.ambient = p->ambient;
    d.intensity = p->intensity;
    for (int i = 0; i < 4; ++i) d.vcol[i] = p->vertex_color[i];
    d.use_nm = p->use_normal_map && p->normal_map;
}

// src/model_renderer/mod.rs:314-318: scale((W-1)/2,(H-1)/2,1) * translate(1,1,0)
Mat4 viewport_matrix(int W, int H) {
    Mat4 sc = {{(double)(W - 1) / 2.0, 0, 0, 0, 0, (double)(H - 1) / 2.0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
    Mat4 trn = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 1.0, 1.0, 0.0, 1}};
    return host_mat4_mul(sc, trn);
}

// The scanline prefix of halves [lo, lo + n_halves), continuing from the previous draw's total.
int launch_scan(tr_context* c, uint32_t lo, uint32_t n_halves, const uint8_t* cull = nullptr) {
    const uint32_t n_clusters = (n_halves / 2u + 63u) / 64u;  // (of the draw whose halves these are, when `cull` is given)
    uint32_t tiles = (n_halves + kScanTile - 1) / kScanTile;
    if (tiles <= 8) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(kScanThreads), 0, c->stream, c->half_cnt + lo, n_halves,
                           c->half_off + lo, c->flags);
    } else {
        hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(kScanThreads), 0, c->stream, c->half_cnt + lo,
                           n_halves, c->tile_sum, cull, n_clusters);
        hipLaunchKernelGGL(k_scan_tile_bases, dim3(1), dim3(kScanThreads), 0, c->stream, c->tile_sum, tiles,
                           c->half_off + lo, c->flags);
        hipLaunchKernelGGL(k_scan_apply, dim3(tiles), dim3(kScanThreads), 0, c->stream, c->half_cnt + lo, n_halves,
                           c->half_off + lo, (const unsigned long long*)c->tile_sum, c->flags, cull, n_clusters);
    }
    TR_HIP(c, hipGetLastError());
    return TR_OK;
}

// Records of the frame's first slots: a triangle of a few pixels takes 6-8 slots; slots beyond the buffer cost a
// rebuild per pass.  Grown with the first draw of a frame unless TR_SCAN_RECORDS pins the size.
int grow_records_for(tr_context* c, uint64_t n_tris) {
    if (!(c->tri_total == 0 && c->rec_auto && c->mode != 1)) return TR_OK;
    uint64_t want = 8ull * n_tris;
    // (large triangles take a slot per scanline, a hundred and more each: the slot total an earlier frame of this context
    // left in `slots_seen` -- whichever was the last to get there, nobody waits for it -- plus a quarter)
    if (c->slots_seen) {
        const uint64_t seen = *(volatile uint32_t*)c->slots_seen;
        if (seen + seen / 4 > want) want = seen + seen / 4;
    }
    if (want > (64ull << 20)) want = 64ull << 20;  // 4 GB of records at most
    if (want > c->rec_cap) {
        TR_HIP(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->rec);
        c->rec = nullptr;
        TR_HIP(c, hipMalloc(&c->rec, want * sizeof(ScanRec)));
        c->rec_cap = (uint32_t)want;
    }
    return TR_OK;
}

bool shade_trace_on() {
    static const bool on = env_diag("TR_SHADE_TRACE") != nullptr;
    return on;
}

constexpr int kMaxDevices = 64;
std::atomic<int> g_live_contexts[kMaxDevices];  // per device (zero-initialised)

// ---- tr_read_rgba8_static: the one entry point that allocates nothing.  Its staging buffer (the flipped
// image on its way to the host, tr_read_rgba8's flip_buf) is this file-scope device array: every GPU the
// library runs on holds its own copy (33 177 600 B) at its own address until the process ends.  The name
// `g_static_stage` is only a host-side stand-in for it; the device address is looked up once per device
// with that device current (tr_create) and host code uses that pointer alone.
constexpr size_t kStaticStagePixels = (size_t)3840 * 2160;
__device__ __align__(256) uint32_t g_static_stage[kStaticStagePixels];
uint32_t* g_static_stage_ptr[kMaxDevices];  // per device; nullptr until tr_create ran on it
std::mutex g_static_stage_lock;
// One call at a time per GPU: every context of a GPU shares that GPU's array, so a second call that arrives while one
// is between its flip launch and the end of its copy is refused (TR_ERR_BUSY) instead of racing it.
std::atomic_flag g_static_stage_busy[kMaxDevices];  // (static storage: clear)

// (the caller has made `device` current)
hipError_t init_static_stage(int device) {
    if (device < 0 || device >= kMaxDevices) return hipSuccess;  // (the entry point reports such a device as unsupported)
    std::lock_guard<std::mutex> g(g_static_stage_lock);
    if (g_static_stage_ptr[device]) return hipSuccess;
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_stage));
    if (e != hipSuccess) return e;
    g_static_stage_ptr[device] = static_cast<uint32_t*>(p);
    return hipSuccess;
}
uint32_t* static_stage(int device) {
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> g(g_static_stage_lock);
    return g_static_stage_ptr[device];
}
inline std::atomic<int>& live_contexts(int device) { return g_live_contexts[device >= 0 && device < kMaxDevices ? device : 0]; }

template <int PASS>
void launch_k_raster(tr_context* c, bool single, bool precheck, bool dbg, int threads, dim3 grid, dim3 block, hipStream_t stream, const RasterArgs& ra) {
    if (ra.view_tris) {  // a batch of views: the tuned settings only (filter in front of the depth atomics, none in front of the winner's)
        if (threads == 128) hipLaunchKernelGGL((k_raster<PASS, PASS == 0, false, 128, true>), grid, block, 0, stream, ra);
        else hipLaunchKernelGGL((k_raster<PASS, PASS == 0, false, 256, true>), grid, block, 0, stream, ra);
        return;
    }
#define TR_LAUNCH(PC, DB, TH) hipLaunchKernelGGL((k_raster<PASS, PC, DB, TH>), grid, block, 0, stream, ra)
#ifdef TR_DIAG  // (the instantiations with the timing experiments and the per-chunk timeline exist in the diagnostic build only)
    if (dbg) {
        if (threads == 128) { if (precheck) TR_LAUNCH(true, true, 128); else TR_LAUNCH(false, true, 128); }
        else { if (precheck) TR_LAUNCH(true, true, 256); else TR_LAUNCH(false, true, 256); }
        return;
    }
#endif
    (void)dbg;
    // (`single`: this launch is all of its profiling stage -- timed by its own events when profiling is on)
    if (PASS == 0 && precheck) {
        if (single) {
            if (threads == 128) TR_LAUNCH_STAGE(c, (k_raster_depth<128, TR_DEPTH_OCC_128>), grid, block, ra);
            else TR_LAUNCH_STAGE(c, (k_raster_depth<256, TR_DEPTH_OCC_256>), grid, block, ra);
        } else {
            if (threads == 128) hipLaunchKernelGGL((k_raster_depth<128, TR_DEPTH_OCC_128>), grid, block, 0, stream, ra);
            else hipLaunchKernelGGL((k_raster_depth<256, TR_DEPTH_OCC_256>), grid, block, 0, stream, ra);
        }
        return;
    }
    if (PASS == 1 && !precheck) {
        if (single) {
            if (threads == 128) TR_LAUNCH_STAGE(c, (k_raster_winner<128, TR_WINNER_OCC_128>), grid, block, ra);
            else TR_LAUNCH_STAGE(c, (k_raster_winner<256, TR_WINNER_OCC_256>), grid, block, ra);
        } else {
            if (threads == 128) hipLaunchKernelGGL((k_raster_winner<128, TR_WINNER_OCC_128>), grid, block, 0, stream, ra);
            else hipLaunchKernelGGL((k_raster_winner<256, TR_WINNER_OCC_256>), grid, block, 0, stream, ra);
        }
        return;
    }
    if (threads == 128) { if (precheck) TR_LAUNCH(true, false, 128); else TR_LAUNCH(false, false, 128); }
    else { if (precheck) TR_LAUNCH(true, false, 256); else TR_LAUNCH(false, false, 256); }
#undef TR_LAUNCH
}

// Workgroups of a k_small launch over at most n_halves halves: one thread per half up to a grid that fills the GPU
// a few times over, the threads stride beyond that.
unsigned small_grid(uint32_t n_halves) {
    const uint32_t want = (n_halves + 255u) / 256u;
    return want < 1u ? 1u : (want > 8192u ? 8192u : want);
}

// Workgroups of a k_mid launch (kMidWaves waves each, one wave per half): enough to fill the GPU a few times over.
unsigned mid_grid(uint32_t n_halves) {
    const uint32_t want = ((n_halves >> mid_shift_for(n_halves)) + kMidWaves) / kMidWaves;
    return want < 1u ? 1u : (want > 16384u ? 16384u : want);
}

// Does a launch of n_launch triangles (a draw, or a batch of views of a mesh of n_tris triangles) use k_mid?  One wave
// per triangle half pays when there are enough halves to fill the GPU several times over and the triangles are
// small on the screen; a lone 2 492-triangle mesh is 4 984 waves of up to 64 serial rounds each (measured at 800x800:
// 9.3 k against 16.2 k frames/s through the slot sequence), 32 views of diablo3_pose are 321 000 short ones.
bool use_mid_path(const tr_context* c, uint64_t n_tris, uint64_t n_launch) {
    if (c->mode != 0 || n_tris == 0) return false;
    if (c->mid_path >= 0) return c->mid_path != 0;
    // (below 16 pixels per triangle on average nearly every triangle is one of k_small's)
    const uint64_t px_per_tri = (uint64_t)c->W * (uint64_t)c->H / n_tris;
    return n_launch >= 32768u && px_per_tri >= 16u && px_per_tri < (uint64_t)c->mid_px;
}

// Mean samples per slot the setup aims below when it spreads a triangle's scanlines over the slot sequence (k_setup), i.e.
// half of what a 64-slot chunk holds at most.  Measured at 4K (african_head): a context that is alone on its device renders
// fastest with chunks of <= 2 300 samples (36: 8 650 frames/s; 30: 8 570, 44: 8 310, 54: 8 210 -- one frame at a time the
// launch's tail decides); with frames in flight on several contexts the chunks' fixed costs do, and larger chunks win
// (36: 13 510 frames/s, 44: 13 680, 54: 13 830, 72: 13 780).
#ifndef TR_SLOT_LEN_MANY
#define TR_SLOT_LEN_MANY 54.0f
#endif
float slot_len_for(const tr_context* c) { return live_contexts(c->device).load() > 1 ? TR_SLOT_LEN_MANY : 36.0f; }

// passes 0 / 1 over the half range [h_lo, h_hi)
int launch_raster(tr_context* c, int pass, uint32_t h_lo, uint32_t h_hi, bool use_live = false, bool rec_ready = false) {
    RasterArgs ra;
    fill_raster_args(c, ra, h_lo, h_hi);
    ra.use_live = use_live && c->skip_dead ? 1 : 0;
    ra.rec_ready = rec_ready ? 1 : 0;
    if (const char* e = env_diag("TR_RASTER_DEBUG")) ra.debug = std::atoi(e);
    static const char* trace_path = env_diag("TR_RASTER_TRACE");  // debugging aid, see tools/raster_trace.py
    static unsigned long long* trace_buf = nullptr;
    constexpr uint32_t kTraceCap = 1u << 16;
    if (trace_path) {
        if (!trace_buf) TR_HIP(c, hipMalloc(&trace_buf, kTraceCap * 4 * sizeof(unsigned long long)));
        TR_HIP(c, hipMemsetAsync(trace_buf, 0, kTraceCap * 4 * sizeof(unsigned long long), c->stream));
        ra.trace = trace_buf;
        ra.trace_cap = kTraceCap;
    }
    // workgroup size: see k_raster; frames in flight are recognised by a second live context on the device
    int threads = c->raster_threads ? c->raster_threads : (live_contexts(c->device).load() > 1 ? 128 : 256);
    int grid_blocks = kRasterWaves * 64 / threads;
    if (const char* e = env_diag("TR_RASTER_BLOCKS")) grid_blocks = std::atoi(e);
    dim3 grid(grid_blocks), block(threads);
    struct TraceDump {
        tr_context* c; int pass; const char* path; unsigned long long* buf;
        ~TraceDump() {
            if (!path) return;
            std::vector<unsigned long long> h((size_t)kTraceCap * 4);
            if (hipStreamSynchronize(c->stream) != hipSuccess) return;
            if (hipMemcpy(h.data(), buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
            std::string fn = std::string(path) + (pass ? ".winner" : ".depth");
            if (FILE* f = std::fopen(fn.c_str(), "wb")) { std::fwrite(h.data(), sizeof(unsigned long long), h.size(), f); std::fclose(f); }
        }
    } dump{c, pass, trace_path, trace_buf};
    if (pass == 1) {
        ra.clean_state = c->state_both + (size_t)(1 - c->state_idx) * kStateWords;
        ra.clean_words = kStateWords;
        ra.clean_nan = c->nan_both + (1 - c->state_idx);
        c->other_state_clean = true;
    }
    if (pass == 0) {
        const bool dbg = ra.debug != 0 || ra.trace != nullptr;
        launch_k_raster<0>(c, !c->small_any && !c->mid_any && !rec_ready, c->precheck, dbg, threads, grid, block, c->stream, ra);
        // the sweeps of small / mid triangles in two launches when a draw of the frame split its triangles by depth
        // (kFlagFar): the near ones lower the plane, most samples of the far ones then fail the plain test
        const bool two = c->split_any && c->precheck;
        for (uint32_t ph = two ? 1u : 0u; ph <= (two ? 2u : 0u); ++ph) {
            ra.phase = ph;
            if (c->small_any) {
                const dim3 sg(small_grid(h_hi - h_lo));
                if (c->precheck) hipLaunchKernelGGL((k_small<0, true>), sg, dim3(256), 0, c->stream, ra, c->half_live);
                else hipLaunchKernelGGL((k_small<0, false>), sg, dim3(256), 0, c->stream, ra, c->half_live);
            }
            if (c->mid_any) {
                const dim3 mg(mid_grid(h_hi - h_lo));
                if (c->precheck) hipLaunchKernelGGL((k_mid<0, true>), mg, dim3(64 * kMidWaves), 0, c->stream, ra, c->half_live);
                else hipLaunchKernelGGL((k_mid<0, false>), mg, dim3(64 * kMidWaves), 0, c->stream, ra, c->half_live);
            }
        }
        ra.phase = 0;
        if (h_hi / 2u > c->counted_upto) c->counted_upto = h_hi / 2u;
    } else {
        if (c->small_any)  // (before k_raster<1>, whose tail zeroes the other half of the per-frame state)
            hipLaunchKernelGGL((k_small<1, false>), dim3(small_grid(h_hi - h_lo)), dim3(256), 0, c->stream, ra, c->half_live);
        if (c->mid_any)
            hipLaunchKernelGGL((k_mid<1, false>), dim3(mid_grid(h_hi - h_lo)), dim3(64 * kMidWaves), 0, c->stream, ra, c->half_live);
        // measured on MI355X (4K african_head): the winner pass is faster without the filter
        // (44 vs 52 us), the depth pass with it (71 vs 93 us)
        const bool dbg = ra.debug != 0 || ra.trace != nullptr;
        launch_k_raster<1>(c, !c->small_any && !c->mid_any, c->precheck_winner, dbg, threads, grid, block, c->stream, ra);
    }
    TR_HIP(c, hipGetLastError());
    return TR_OK;
}

// Tile mode: the (tile, triangle) pair list has a fixed capacity; if a frame needs more, the
// device raises bit 6, and the next host-synchronising call grows the list and redoes the resolve.
int settle_row_overflow(tr_context* c) {
    for (int attempt = 0; attempt < 4; ++attempt) {
        uint32_t fl = 0;
        TR_HIP(c, hipMemcpyAsync(&fl, c->flags, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
        TR_HIP(c, hipStreamSynchronize(c->stream));
        if (!(fl & (1u << 6))) return TR_OK;
        uint32_t slots = 0, refs = 0;
        TR_HIP(c, hipMemcpy(&slots, c->half_off + 2 * (size_t)c->tri_total, sizeof(slots), hipMemcpyDeviceToHost));
        bool fallback = false;
        if (slots > c->rec_cap) {
            uint64_t cap = (uint64_t)slots + slots / 4 + 1024;
            if (cap * sizeof(ScanRec) > (8ull << 30)) {
                fallback = true;
            } else {
                (void)hipFree(c->rec);
                c->rec = nullptr;
                TR_HIP(c, hipMalloc(&c->rec, cap * sizeof(ScanRec)));
                c->rec_cap = (uint32_t)cap;
            }
        } else {
            TR_HIP(c, hipMemcpy(&refs, c->bin_off + c->n_bins_padded, sizeof(refs), hipMemcpyDeviceToHost));
            uint64_t cap = (uint64_t)refs + refs / 4 + 1024;
            if (cap > (1ull << 30)) {
                fallback = true;
            } else {
                (void)hipFree(c->refs);
                c->refs = nullptr;
                TR_HIP(c, hipMalloc(&c->refs, cap * sizeof(uint32_t)));
                c->ref_cap = (uint32_t)cap;
            }
        }
        fl &= ~(1u << 6);
        TR_HIP(c, hipMemcpy(c->flags, &fl, sizeof(fl), hipMemcpyHostToDevice));
        TR_HIP(c, hipMemsetAsync(c->bin_cnt, 0, (size_t)c->n_bins_padded * sizeof(uint32_t), c->stream));
        TR_HIP(c, hipMemsetAsync(c->counters, 0, kCounterSlots * 8 * sizeof(unsigned long long), c->stream));
        uint32_t mask = c->resolved_mask;
        c->resolved = false;
        c->keys_valid = false;
        if (fallback) {
            // scanlines that sweep the whole frame (huge or non-finite coordinates) in numbers no bin
            // list should hold: this context continues with the global-atomic raster
            c->mode = 0;
            c->depth_upto = 0;
            c->counted_upto = 0;
            c->wkey_dirty = false;
            TR_HIP(c, hipMemsetAsync(c->planes, 0xFF, 2 * c->npix_all * sizeof(unsigned long long), c->stream));
            TR_HIP(c, hipMemset2DAsync(c->blk_flags, 2, 0, 1, c->n_blk, c->stream));
            // k_clear_state's job for the NaN table was done at tr_clear; the row raster never touched it
        }
        int rc = resolve_frame(c, mask, true);
        if (rc != TR_OK) return rc;
        if (fallback) return TR_OK;
    }
    return fail(c, TR_ERR_UNSUPPORTED, "row bin list kept overflowing");
}

int settle_bin_overflow(tr_context* c) {
    if (c->mode == 2 && c->resolved) {
        const int rc = settle_row_overflow(c);
        c->bin_dirty = rc != TR_OK;
        return rc;
    }
    if (c->mode != 1 || !c->resolved) return TR_OK;
    for (int attempt = 0; attempt < 4; ++attempt) {
        uint32_t fl = 0;
        TR_HIP(c, hipMemcpyAsync(&fl, c->flags, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
        TR_HIP(c, hipStreamSynchronize(c->stream));
        if (!(fl & (1u << 6))) return TR_OK;
        uint32_t need = 0;
        TR_HIP(c, hipMemcpy(&need, c->tile_off + (size_t)c->tiles_x * c->tiles_y, sizeof(need), hipMemcpyDeviceToHost));
        uint64_t cap = (uint64_t)need + need / 2 + 1024;
        if (cap > 0xFFFFFFF0ull) return fail(c, TR_ERR_SIZE, "tile bin list exceeds 2^32 entries");
        (void)hipFree(c->pairs);
        c->pairs = nullptr;
        TR_HIP(c, hipMalloc(&c->pairs, cap * sizeof(uint32_t)));
        c->pair_cap = (uint32_t)cap;
        fl &= ~(1u << 6);
        TR_HIP(c, hipMemcpy(c->flags, &fl, sizeof(fl), hipMemcpyHostToDevice));
        uint32_t mask = c->resolved_mask;
        c->resolved = false;
        // the fragment counter was bumped by the aborted attempt
        TR_HIP(c, hipMemsetAsync(c->counters, 0, kCounterSlots * 8 * sizeof(unsigned long long), c->stream));
        int rc = resolve_frame(c, mask, true);
        if (rc != TR_OK) return rc;
    }
    return fail(c, TR_ERR_UNSUPPORTED, "tile bin list kept overflowing");
}

int need_plane(tr_context* c, uint32_t plane, bool colour = true) {
    if (!(c->resolved && (c->resolved_mask & plane))) {
        // (colour == false only spares the shading of a frame nobody has resolved yet)
        int rc = resolve_frame(c, (c->resolved ? c->resolved_mask : 0u) | plane, colour || c->resolved);
        if (rc != TR_OK) return rc;
    }
    return settle_bin_overflow(c);
}

template <typename T>
int read_plane(tr_context* c, const T* dev, T* out, int layout, int comps) {
    size_t npix = (size_t)c->W * c->H;
    dev += (size_t)c->sel_view * npix * comps;  // (tr_select_view; slice 0 of an ordinary context)
    if (layout == TR_LAYOUT_ROW_MAJOR) {
        TR_HIP(c, hipMemcpyAsync(out, dev, npix * comps * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        TR_HIP(c, hipStreamSynchronize(c->stream));
        return TR_OK;
    }
    if (layout != TR_LAYOUT_REFERENCE) return fail(c, TR_ERR_INVALID, "unknown layout");
    std::unique_ptr<T[]> tmp(new (std::nothrow) T[npix * comps]);
    if (!tmp) return fail(c, TR_ERR_OOM, "host staging buffer for the reference layout");
    TR_HIP(c, hipMemcpyAsync(tmp.get(), dev, npix * comps * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    TR_HIP(c, hipStreamSynchronize(c->stream));
    for (int y = 0; y < c->H; ++y)
        for (int x = 0; x < c->W; ++x)
            for (int k = 0; k < comps; ++k)
                out[((size_t)x * c->H + y) * comps + k] = tmp[((size_t)y * c->W + x) * comps + k];
    return TR_OK;
}


}  // namespace

extern "C" {

const char* tr_version(void) {
#ifdef TR_DIAG
    return "tinyrenderer_hip abi=5 arch=gfx950 fp=f64 contract=off diag";
#else
    return "tinyrenderer_hip abi=5 arch=gfx950 fp=f64 contract=off";
#endif
}

const char* tr_last_error(const tr_context* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int tr_create(int device, int width, int height, tr_context** out_ctx) { return tr_create_views(device, width, height, 1, out_ctx); }

int tr_create_views(int device, int width, int height, int max_views, tr_context** out_ctx) {
    if (!out_ctx) return TR_ERR_INVALID;
    *out_ctx = nullptr;
    if (width < 1 || height < 1 || (int64_t)width * height > (int64_t)1 << 30) return TR_ERR_SIZE;
    // (pixel indices over all slices are 32-bit in the kernels; a batch's views are draws of one frame: at most 4096)
    if (max_views < 1 || max_views > 4096 || (int64_t)width * height * max_views >= (int64_t)1 << 31) return TR_ERR_SIZE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev)
        return TR_ERR_NO_DEVICE;
    tr_context* c = new (std::nothrow) tr_context();
    if (!c) return TR_ERR_OOM;
    c->device = device;
    c->W = width;
    c->H = height;
    c->y_lo = 0;
    c->y_hi = height;
    c->shaded_lo = 0;
    c->shaded_hi = height;
    c->paint_box.x0 = 0; c->paint_box.y0 = 0; c->paint_box.x1 = width; c->paint_box.y1 = height;
    if (const char* e = env_diag("TR_DEPTH_SPLIT")) c->depth_split = std::atoi(e) != 0;
    if (const char* e = env_diag("TR_PLANE_TILES")) c->plane_tiles = std::atoi(e) != 0;
    if (const char* e = env_diag("TR_SHADE_BOX")) c->shade_box = std::atoi(e) != 0;
    if (const char* e = env_opt("TR_SHADE_EXACT")) c->colour_exact = std::atoi(e) != 0;
    c->max_views = max_views;
    c->n_views = max_views;  // (an empty frame resolves every slice)
    c->npix_all = (size_t)width * height * (size_t)max_views;
    if (const char* e = env_opt("TR_RASTER_MODE"))
        c->mode = (std::strcmp(e, "tile") == 0 || std::strcmp(e, "1") == 0) ? 1
                  : (std::strcmp(e, "rows") == 0 || std::strcmp(e, "2") == 0) ? 2 : 0;
    if (const char* e = env_diag("TR_RASTER_PRECHECK")) c->precheck = std::atoi(e) != 0;
    if (const char* e = env_diag("TR_RASTER_PRECHECK_WINNER")) c->precheck_winner = std::atoi(e) != 0;
    if (const char* e = env_diag("TR_RASTER_SKIP_DEAD")) c->skip_dead = std::atoi(e) != 0;
    if (const char* e = env_diag("TR_RASTER_BOX")) c->raster_box = std::atoi(e) != 0;
    if (const char* e = env_opt("TR_FUSED_FRONT")) c->fused_front = std::atoi(e) != 0;
    if (const char* e = env_opt("TR_RASTER_THREADS")) {
        const int t = std::atoi(e);
        if (t == 128 || t == 256) c->raster_threads = t;
    }
    if (const char* e = env_diag("TR_SMALL_TRIS")) c->small_path = std::atoi(e) != 0;
    if (const char* e = env_diag("TR_SMALL_RES")) c->small_res = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = env_opt("TR_SMALL_DRAW")) c->small_draw_tris = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = env_opt("TR_MID_TRIS")) c->mid_path = std::atoi(e);
    if (const char* e = env_diag("TR_MID_PX")) c->mid_px = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = env_opt("TR_MID_SAMPLES")) c->mid_samples = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = env_diag("TR_BIG_DRAW")) c->big_draw_tris = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = env_diag("TR_DEBUG_PREFIX_STALL")) c->debug_prefix_stall = std::atoi(e) != 0;
    if (const char* e = env_opt("TR_DEFER_SHADE")) c->defer_tris = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = env_opt("TR_SCAN_RECORDS")) {
        c->rec_cap = (uint32_t)std::strtoul(e, nullptr, 10);
        c->rec_auto = false;
    }
    if (const char* e = env_opt("TR_ROW_REFS")) c->ref_cap = (uint32_t)std::strtoul(e, nullptr, 10);
    if (c->ref_cap < 1) c->ref_cap = 1;
    if (max_views > 1) c->mode = 0;  // (batches of views exist in the global raster only)
    auto bail = [&](int code) {
        tr_destroy(c);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(TR_ERR_NO_DEVICE);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return bail(TR_ERR_NO_DEVICE);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "tinyrenderer_hip: device %d is %s, kernels are built for gfx950 only\n",
                     device, prop.gcnArchName);
        return bail(TR_ERR_NO_DEVICE);
    }
    size_t npix = (size_t)width * height;
    // (device address of tr_read_rgba8_static's staging array, looked up once per device with that device current; a
    // failure is that call's to report -- TR_ERR_UNSUPPORTED -- not a reason to refuse a context that may never use it)
    (void)init_static_stage(device);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(TR_ERR_HIP);
    if (hipMalloc(&c->planes, 2 * c->npix_all * sizeof(unsigned long long)) != hipSuccess) return bail(TR_ERR_OOM);
    c->depth = c->planes;
    c->wkey = c->planes + c->npix_all;
    if (hipMalloc(&c->rgba8, c->npix_all * sizeof(uint32_t)) != hipSuccess) return bail(TR_ERR_OOM);
    if (max_views > 1) {
        for (int i = 0; i < 2; ++i)
            if (hipEventCreateWithFlags(&c->view_ev[i], hipEventDisableTiming) != hipSuccess) return bail(TR_ERR_HIP);
        if (ensure_view_capacity(c, (size_t)max_views) != TR_OK) return bail(TR_ERR_OOM);
    }
    (void)npix;
    if (hipMalloc(&c->draw_table, c->draw_cap * sizeof(DrawDev)) != hipSuccess) return bail(TR_ERR_OOM);
    if (hipMalloc(&c->state_both, 2 * (size_t)kStateWords * sizeof(unsigned long long)) != hipSuccess) return bail(TR_ERR_OOM);
    if (hipMalloc(&c->nan_both, 2 * sizeof(NanTable)) != hipSuccess) return bail(TR_ERR_OOM);
    use_state(c, 0);
    if (const char* e = env_opt("TR_TILE")) {
        int w = 0, h = 0;
        if (std::sscanf(e, "%dx%d", &w, &h) == 2 &&
            ((w == 16 && h == 16) || (w == 32 && h == 16) || (w == 32 && h == 32) || (w == 64 && h == 32))) {
            c->tile_w = w; c->tile_h = h;
        }
    }
    for (c->tw_shift = 0; (1 << c->tw_shift) < c->tile_w; ++c->tw_shift) {}
    for (c->th_shift = 0; (1 << c->th_shift) < c->tile_h; ++c->th_shift) {}
    if (hipFuncSetAttribute((const void*)tile_kernel(c), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bytes(c)) != hipSuccess) return bail(TR_ERR_HIP);
    c->tiles_x = (width + c->tile_w - 1) / c->tile_w;
    c->tiles_y = (height + c->tile_h - 1) / c->tile_h;
    {
        size_t n_tiles = (size_t)c->tiles_x * c->tiles_y;
        const size_t quantum = 4 * (size_t)kTileScanThreads;
        c->n_tiles_padded = (uint32_t)((n_tiles + quantum - 1) / quantum * quantum);
        c->pair_cap = 4u << 20;
        if (const char* e = env_opt("TR_TILE_PAIRS")) c->pair_cap = (uint32_t)std::strtoul(e, nullptr, 10);  // (tests: force a growth)
        if (c->pair_cap < 1) c->pair_cap = 1;
        if (hipMalloc(&c->tile_cnt, c->n_tiles_padded * sizeof(uint32_t)) != hipSuccess) return bail(TR_ERR_OOM);
        if (hipMalloc(&c->tile_off, ((size_t)c->n_tiles_padded + 4) * sizeof(uint32_t)) != hipSuccess) return bail(TR_ERR_OOM);
        if (hipMalloc(&c->tile_list, (size_t)c->n_tiles_padded * sizeof(uint32_t)) != hipSuccess) return bail(TR_ERR_OOM);
        if (hipMalloc(&c->pairs, (size_t)c->pair_cap * sizeof(uint32_t)) != hipSuccess) return bail(TR_ERR_OOM);
    }
    if (hipMalloc(&c->ticket, 256 * sizeof(unsigned long long)) != hipSuccess) return bail(TR_ERR_OOM);  // (k_setup<true>: one word per workgroup)
    if (hipMemsetAsync(c->ticket, 0, 256 * sizeof(unsigned long long), c->stream) != hipSuccess) return bail(TR_ERR_HIP);
    {
        c->nseg = (width + kSegW - 1) / kSegW;
        const size_t n_bins = (size_t)c->nseg * height;
        const size_t quantum = 4 * (size_t)kTileScanThreads;
        c->n_bins_padded = (uint32_t)((n_bins + quantum - 1) / quantum * quantum);
        if (hipMalloc(&c->bin_cnt, (size_t)c->n_bins_padded * sizeof(uint32_t)) != hipSuccess) return bail(TR_ERR_OOM);
        if (hipMalloc(&c->bin_off, ((size_t)c->n_bins_padded + 4) * sizeof(uint32_t)) != hipSuccess) return bail(TR_ERR_OOM);
        if (hipMalloc(&c->bin_list, (size_t)c->n_bins_padded * sizeof(uint32_t)) != hipSuccess) return bail(TR_ERR_OOM);
        if (hipMalloc(&c->refs, (size_t)c->ref_cap * sizeof(uint32_t)) != hipSuccess) return bail(TR_ERR_OOM);
        if (hipMemsetAsync(c->bin_cnt, 0, (size_t)c->n_bins_padded * sizeof(uint32_t), c->stream) != hipSuccess) return bail(TR_ERR_HIP);
    }
    if (const char* e = env_diag("TR_SHADE_SKIP_CLEAN")) c->skip_clean = std::atoi(e) != 0;
    c->blocks_x = (width + kShadeBW - 1) / kShadeBW;
    c->blk_rows_per_view = (uint32_t)((height + kShadeBH - 1) / kShadeBH);
    c->blk_per_view = (uint32_t)c->blocks_x * c->blk_rows_per_view;
    c->n_blk = c->blk_per_view * (uint32_t)max_views;
    if (hipMalloc(&c->blk_flags, 2 * (size_t)c->n_blk + 8) != hipSuccess)  // (read as aligned dwords)
        return bail(TR_ERR_OOM);
    // every block starts "painted": the first resolve writes the whole frame
    if (hipMemset2DAsync(c->blk_flags, 2, 0, 1, c->n_blk, c->stream) != hipSuccess) return bail(TR_ERR_HIP);
    if (hipMemset2DAsync(c->blk_flags + 1, 2, 1, 1, c->n_blk, c->stream) != hipSuccess) return bail(TR_ERR_HIP);
    if (hipMalloc(&c->chunk_live, kChunkLiveCap) != hipSuccess) return bail(TR_ERR_OOM);
    // (without it the record buffer is sized by the triangle count alone, as before: not an error)
    if (hipHostMalloc((void**)&c->slots_seen, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        c->slots_seen = nullptr;
    } else {
        *c->slots_seen = 0u;
    }
    if (hipMemsetAsync(c->chunk_live, 0, kChunkLiveCap, c->stream) != hipSuccess) return bail(TR_ERR_HIP);
    if (hipMalloc(&c->rec, (size_t)(c->rec_cap ? c->rec_cap : 1) * sizeof(ScanRec)) != hipSuccess) return bail(TR_ERR_OOM);
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return bail(TR_ERR_HIP);
    if (hipEventCreate(&c->ev2) != hipSuccess || hipEventCreate(&c->ev3) != hipSuccess) return bail(TR_ERR_HIP);
    if (ensure_tri_capacity(c, 16384) != TR_OK) return bail(TR_ERR_OOM);
    int rc = tr_clear(c);
    if (rc != TR_OK) return bail(rc);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(TR_ERR_HIP);
    *out_ctx = c;
    c->counted = true;
    live_contexts(c->device).fetch_add(1);
    return TR_OK;
}

void tr_destroy(tr_context* c) {
    if (!c) return;
    if (c->counted) live_contexts(c->device).fetch_sub(1);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->planes); (void)hipFree(c->rgba8); (void)hipFree(c->z_out); (void)hipFree(c->tri_out);
    (void)hipFree(c->rgba_f64); (void)hipFree(c->key_out); (void)hipFree(c->rast); (void)hipFree(c->shade); (void)hipFree(c->half_cnt);
    (void)hipFree(c->half_off); (void)hipFree(c->rec); (void)hipFree(c->flip_buf); (void)hipFree(c->chunk_live); if (c->slots_seen) (void)hipHostFree(c->slots_seen); (void)hipFree(c->half_live); (void)hipFree(c->small_flag); (void)hipFree(c->need_shade); (void)hipFree(c->small_grp); (void)hipFree(c->cluster_cull); (void)hipFree(c->blk_flags); (void)hipFree(c->ticket);
    (void)hipFree(c->bin_cnt); (void)hipFree(c->bin_off); (void)hipFree(c->bin_list); (void)hipFree(c->refs); (void)hipFree(c->tile_sum); (void)hipFree(c->ndc); (void)hipFree(c->nout);
    (void)hipFree(c->draw_table); (void)hipFree(c->ext_table); (void)hipFree(c->state_both); (void)hipFree(c->nan_both);
    (void)hipFree(c->tile_cnt); (void)hipFree(c->tile_off); (void)hipFree(c->tile_list); (void)hipFree(c->pairs);
    (void)hipFree(c->view_table);
    for (int i = 0; i < 2; ++i) {
        if (c->view_stage[i]) (void)hipHostFree(c->view_stage[i]);
        if (c->view_ev[i]) (void)hipEventDestroy(c->view_ev[i]);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev2) (void)hipEventDestroy(c->ev2);
    if (c->ev3) (void)hipEventDestroy(c->ev3);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int tr_clear(tr_context* c) {
    if (!c) return TR_ERR_INVALID;
    TR_HIP(c, hipSetDevice(c->device));
    ProfScope ps(c, ST_CLEAR);
    size_t npix = c->npix_all;
    {   // counters, covered-pixel counters, status flags, NaN table: the other half of the state,
        // which the last shade pass has already zeroed unless no resolve happened since the last clear
        const int other = 1 - c->state_idx;
        if (!c->other_state_clean) {
            hipLaunchKernelGGL(k_clear_state, dim3((kStateWords + 255) / 256), dim3(256), 0, c->stream,
                               c->state_both + (size_t)other * kStateWords, (uint32_t)kStateWords, c->nan_both + other,
                               c->half_off);
            TR_HIP(c, hipGetLastError());
        }
        use_state(c, other);
        c->other_state_clean = false;
    }
    if (c->mode == 2 && c->bin_dirty) {
        // row raster: the raster zeroes the fill cursors of the bins it consumes; a resolve that gave up on an
        // overflow (it could not grow its lists) leaves them behind
        TR_HIP(c, hipMemsetAsync(c->bin_cnt, 0, (size_t)c->n_bins_padded * sizeof(uint32_t), c->stream));
        c->bin_dirty = false;
    }
    if (c->mode == 1) {
        // tile raster: no plane is cleared; every non-empty tile is rewritten, empty ones are never read
        TR_HIP(c, hipMemsetAsync(c->tile_cnt, 0, (size_t)c->n_tiles_padded * sizeof(uint32_t), c->stream));
    } else if (!c->planes_clean) {
        // depth = "no sample yet", winner = none: both all-ones.  Skipped when the last resolve
        // already left the planes cleared (the shade pass wipes what it consumes).
        TR_HIP(c, hipMemsetAsync(c->planes, 0xFF, 2 * npix * sizeof(unsigned long long), c->stream));
        TR_HIP(c, hipMemset2DAsync(c->blk_flags, 2, 0, 1, c->n_blk, c->stream));  // "touched" goes with the keys
        c->planes_clean = true;
    }
    c->wkey_dirty = false;
    c->depth_upto = 0;
    c->counted_upto = 0;
    c->keys_valid = false;
    c->cov_dirty = false;
    c->tri_total = 0;
    c->draw_count = 0;
    c->resolved = false;
    c->resolved_mask = 0;
    c->frame_has_ext = false;
    c->frame_maps.clear();
    c->frame_textures.clear();
    c->frame_lost_texture = false;
    c->deferred.clear();
    c->frame_box = PixBox();
    c->cull_used = 0;
    c->frame_lost_mesh = false;
    if (c->small_any)  // the per-group summary of the small triangles restarts with the frame
        TR_HIP(c, hipMemsetAsync(c->small_grp, 0, c->tri_cap / 32 + 1, c->stream));
    c->small_any = false;
    c->mid_any = false;
    c->split_any = false;
    c->frame_tiled = false;
    c->view_tris = 0;
    c->view_meshes = 1;
    c->n_views = c->max_views;  // (an empty frame resolves every slice)
    return TR_OK;
}

int tr_mesh_create(tr_context* c, const double* positions, int64_t n_positions, const double* normals,
                   int64_t n_normals, const double* uvw, int64_t n_uvw, const uint32_t* idx,
                   int64_t n_triangles, tr_mesh** out_mesh) {
    if (!c || !out_mesh) return TR_ERR_INVALID;
    *out_mesh = nullptr;
    if (n_positions < 0 || n_normals < 0 || n_uvw < 0 || n_triangles < 0)
        return fail(c, TR_ERR_INVALID, "negative count");
    // (what the kernels can index: positions + normals are one launch of 32-bit thread ids, 3 * index + 2 and
    // 2 * n + index are formed in 64 bits; 2^28 - 1 triangles is the winner key's field)
    constexpr int64_t kAttrMax = (int64_t)1 << 28;
    if (n_triangles > (int64_t)kTriSeqMax || n_positions > kAttrMax || n_normals > kAttrMax || n_uvw > kAttrMax)
        return fail(c, TR_ERR_SIZE, "mesh too large: at most 2^28 positions / normals / texture vertices and 2^28 - 1 triangles");
    if ((n_positions && !positions) || (n_normals && !normals) || (n_uvw && !uvw) || (n_triangles && !idx))
        return fail(c, TR_ERR_INVALID, "null array");
    bool any_uv = false;
    for (int64_t k = 0; k < n_triangles; ++k)
        for (int cnr = 0; cnr < 3; ++cnr) {
            uint32_t v = idx[9 * k + 3 * cnr], t = idx[9 * k + 3 * cnr + 1], n = idx[9 * k + 3 * cnr + 2];
            if ((int64_t)v >= n_positions) return fail(c, TR_ERR_INVALID, "vertex index out of range");
            if (t != 0xFFFFFFFFu && (int64_t)t >= n_uvw) return fail(c, TR_ERR_INVALID, "texture index out of range");
            if (n != 0xFFFFFFFFu && (int64_t)n >= n_normals) return fail(c, TR_ERR_INVALID, "normal index out of range");
            any_uv = any_uv || t != 0xFFFFFFFFu;
        }
    TR_HIP(c, hipSetDevice(c->device));
    tr_mesh* m = new (std::nothrow) tr_mesh();
    if (!m) return fail(c, TR_ERR_OOM, "host allocation failed");
    m->n_pos = n_positions; m->n_nrm = n_normals; m->n_uvw = n_uvw; m->n_tris = n_triangles;
    m->any_uv = any_uv;
    {
        bool ok = n_positions > 0;
        double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (int64_t i = 0; i < n_positions && ok; ++i)
            for (int d = 0; d < 3; ++d) {
                const double v = positions[3 * i + d];
                if (!std::isfinite(v)) { ok = false; break; }
                if (i == 0 || v < lo[d]) lo[d] = v;
                if (i == 0 || v > hi[d]) hi[d] = v;
            }
        m->box_ok = ok;
        for (int d = 0; d < 3; ++d) { m->box[d] = lo[d]; m->box[3 + d] = hi[d]; }
    }
    auto up = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
        size_t alloc = bytes ? bytes : 8;
        hipError_t e = hipMalloc(dst, alloc);
        if (e != hipSuccess) return e;
        if (bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up((void**)&m->pos, positions, (size_t)n_positions * 3 * sizeof(double));
    if (e == hipSuccess) e = up((void**)&m->nrm, normals, (size_t)n_normals * 3 * sizeof(double));
    if (e == hipSuccess) e = up((void**)&m->uvw, uvw, (size_t)n_uvw * 3 * sizeof(double));
    if (e == hipSuccess) e = up((void**)&m->idx, idx, (size_t)n_triangles * 9 * sizeof(uint32_t));
    if (e == hipSuccess && n_triangles >= 65536) {
        // cluster boxes for band contexts (multi-GPU): one pass over the mesh, here rather than per frame
        const uint32_t n_clusters = (uint32_t)((n_triangles + 63) / 64);
        e = hipMalloc(&m->bounds, (size_t)n_clusters * 6 * sizeof(double));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_mesh_bounds, dim3((n_clusters + 255) / 256), dim3(256), 0, c->stream, m->idx, m->pos,
                               (uint32_t)n_triangles, m->bounds);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
    }
    if (e != hipSuccess) {
        tr_mesh_destroy(c, m);
        return fail(c, e == hipErrorOutOfMemory ? TR_ERR_OOM : TR_ERR_HIP, hipGetErrorString(e));
    }
    *out_mesh = m;
    return TR_OK;
}

int tr_mesh_destroy(tr_context* c, tr_mesh* m) {
    if (!c) return TR_ERR_INVALID;
    if (!m) return TR_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    // a draw whose shade records are built at resolve time (large draws) reads the mesh again then
    for (const auto& d : c->deferred)
        if (d.mesh == m) c->frame_lost_mesh = true;
    (void)hipFree(m->pos); (void)hipFree(m->nrm); (void)hipFree(m->uvw); (void)hipFree(m->idx); (void)hipFree(m->bounds);
    delete m;
    return TR_OK;
}

int tr_texture_create(tr_context* c, const uint8_t* rgba8, int width, int height, tr_texture** out_tex) {
    if (!c || !out_tex) return TR_ERR_INVALID;
    *out_tex = nullptr;
    if (!rgba8 || width < 1 || height < 1) return fail(c, TR_ERR_INVALID, "bad texture");
    if (width > 32768 || height > 32768) return fail(c, TR_ERR_SIZE, "texture larger than 32768");
    TR_HIP(c, hipSetDevice(c->device));
    tr_texture* t = new (std::nothrow) tr_texture();
    if (!t) return fail(c, TR_ERR_OOM, "host allocation failed");
    t->w = width;
    t->h = height;
    size_t bytes = (size_t)width * height * 4;
    hipError_t e = hipMalloc(&t->dev, bytes);
    if (e == hipSuccess) e = hipMemcpy(t->dev, rgba8, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(t->dev);
        delete t;
        return fail(c, e == hipErrorOutOfMemory ? TR_ERR_OOM : TR_ERR_HIP, hipGetErrorString(e));
    }
    *out_tex = t;
    return TR_OK;
}

int tr_texture_destroy(tr_context* c, tr_texture* t) {
    if (!c) return TR_ERR_INVALID;
    if (!t) return TR_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    // The draws recorded since tr_clear keep the texture's device address; a later plane request on this frame
    // would shade through it again.  What is resolved stays readable; any further resolve of this frame is refused.
    if (std::find(c->frame_textures.begin(), c->frame_textures.end(), t) != c->frame_textures.end()) c->frame_lost_texture = true;
    (void)hipFree(t->dev);
    delete t;
    return TR_OK;
}

int tr_draw(tr_context* c, const tr_mesh* mesh, const tr_draw_params* p) { return tr_draw_ext(c, mesh, p, nullptr); }

// Snapshot of the light context's z plane into m (allocated, same size).  A frame that has not been
// resolved yet is resolved for depth only: no shading, no colour plane (k_shade with TR_OUT_Z alone).
static int shadow_snapshot(tr_context* light, tr_shadow_map* m) {
    int rc = need_plane(light, TR_OUT_Z, /*colour=*/false);
    if (rc != TR_OK) return rc;
    {
        std::lock_guard<std::mutex> g(m->lock);
        for (auto& ce : m->consumed)  // shade passes that may still read the old contents
            TR_HIP(light, hipStreamWaitEvent(light->stream, ce.second, 0));
    }
    const size_t bytes = (size_t)m->w * m->h * sizeof(double);
    TR_HIP(light, hipMemcpyAsync(m->dev, light->z_out, bytes, hipMemcpyDeviceToDevice, light->stream));
    TR_HIP(light, hipEventRecord(m->ready, light->stream));
    return TR_OK;
}

int tr_shadow_map_create(tr_context* light, tr_shadow_map** out) {
    if (!light || !out) return TR_ERR_INVALID;
    *out = nullptr;
    if (light->max_views > 1) return fail(light, TR_ERR_UNSUPPORTED, "no shadow maps from a tr_create_views context");
    TR_HIP(light, hipSetDevice(light->device));
    tr_shadow_map* m = new (std::nothrow) tr_shadow_map();
    if (!m) return fail(light, TR_ERR_OOM, "shadow map");
    m->w = light->W; m->h = light->H; m->device = light->device;
    const size_t bytes = (size_t)m->w * m->h * sizeof(double);
    int rc = TR_OK;
    if (hipMalloc(&m->dev, bytes) != hipSuccess) rc = fail(light, TR_ERR_OOM, "shadow map");
    else if (hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess) rc = fail(light, TR_ERR_HIP, "shadow map event");
    else rc = shadow_snapshot(light, m);
    if (rc != TR_OK) {
        tr_shadow_map_destroy(m);
        return rc;
    }
    *out = m;
    return TR_OK;
}

int tr_shadow_map_update(tr_shadow_map* m, tr_context* light) {
    if (!m || !light) return TR_ERR_INVALID;
    if (m->device != light->device || m->w != light->W || m->h != light->H)
        return fail(light, TR_ERR_INVALID, "shadow map and light context differ in size or device");
    TR_HIP(light, hipSetDevice(light->device));
    return shadow_snapshot(light, m);
}

void tr_shadow_map_destroy(tr_shadow_map* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipFree(m->dev);  // (synchronises the device: no pass is reading the map any more)
    if (m->ready) (void)hipEventDestroy(m->ready);
    for (auto& ce : m->consumed) (void)hipEventDestroy(ce.second);
    delete m;
}

// Where a draw of `mesh` under `pmv` can land a sample: the mesh box's eight corners through the vertex stage and the
// viewport (the arithmetic of k_cull_clusters, on the host).  With w > 0 at all eight, everything inside the box
// projects between the corners' extremes and a triangle's samples lie inside the triangle; two pixels of margin cover
// the rounding.  Anything doubtful gives the whole frame.
static PixBox draw_pix_box(const tr_context* c, const tr_mesh* mesh, const double pmv[16]) {
    PixBox full;
    full.x1 = c->W; full.y1 = c->H;
    if (!mesh->box_ok) return full;
    const Mat4 vpm = viewport_matrix(c->W, c->H);
    double xmin = 0, xmax = 0, ymin = 0, ymax = 0;
    for (int k = 0; k < 8; ++k) {
        const double q[3] = {mesh->box[(k & 1) ? 3 : 0], mesh->box[(k & 2) ? 4 : 1], mesh->box[(k & 4) ? 5 : 2]};
        double g[4], v[4];
        for (int r = 0; r < 4; ++r) g[r] = pmv[r] * q[0] + pmv[4 + r] * q[1] + pmv[8 + r] * q[2] + pmv[12 + r];
        if (!(g[3] > 1e-9)) return full;
        const double n[3] = {g[0] / g[3], g[1] / g[3], g[2] / g[3]};
        for (int r = 0; r < 4; ++r) v[r] = vpm.m[r] * n[0] + vpm.m[4 + r] * n[1] + vpm.m[8 + r] * n[2] + vpm.m[12 + r];
        const double px = v[0] / v[3], py = v[1] / v[3];
        if (!std::isfinite(px) || !std::isfinite(py) || std::fabs(px) > 1e9 || std::fabs(py) > 1e9) return full;
        if (k == 0 || px < xmin) xmin = px;
        if (k == 0 || px > xmax) xmax = px;
        if (k == 0 || py < ymin) ymin = py;
        if (k == 0 || py > ymax) ymax = py;
    }
    PixBox b;
    b.x0 = (int)std::max(0.0, std::floor(xmin) - 2.0);
    b.y0 = (int)std::max(0.0, std::floor(ymin) - 2.0);
    b.x1 = (int)std::min((double)c->W, std::ceil(xmax) + 3.0);
    b.y1 = (int)std::min((double)c->H, std::ceil(ymax) + 3.0);
    return b;  // (may be empty: the mesh is off the frame)
}

// Viewport depth of the mesh box's centre under `pmv` (NaN: none): k_setup marks the triangles behind it kTriFar.
static double split_depth(const tr_context* c, const tr_mesh* mesh, const double pmv[16]) {
    if (!mesh->box_ok) return std::nan("");
    const double q[3] = {0.5 * (mesh->box[0] + mesh->box[3]), 0.5 * (mesh->box[1] + mesh->box[4]), 0.5 * (mesh->box[2] + mesh->box[5])};
    double g[4];
    for (int r = 0; r < 4; ++r) g[r] = pmv[r] * q[0] + pmv[4 + r] * q[1] + pmv[8 + r] * q[2] + pmv[12 + r];
    if (!(g[3] > 1e-9)) return std::nan("");
    const Mat4 vpm = viewport_matrix(c->W, c->H);
    const double n[3] = {g[0] / g[3], g[1] / g[3], g[2] / g[3]};
    const double vz = vpm.m[2] * n[0] + vpm.m[6] * n[1] + vpm.m[10] * n[2] + vpm.m[14];
    const double vw = vpm.m[3] * n[0] + vpm.m[7] * n[1] + vpm.m[11] * n[2] + vpm.m[15];
    const double z = vz / vw;
    return std::isfinite(z) ? z : std::nan("");
}

int tr_draw_ext(tr_context* c, const tr_mesh* mesh, const tr_draw_params* p, const tr_shade_ext* x) {
    if (!c || !mesh || !p) return TR_ERR_INVALID;
    if (mesh->any_uv && (!p->diffuse || (p->use_normal_map && !p->normal_map)))
        return fail(c, TR_ERR_INVALID, "mesh has texture coordinates but a texture is missing");
    if (c->max_views > 1) return fail(c, TR_ERR_INVALID, "a context made by tr_create_views renders through tr_render_views only");
    if (c->draw_count >= c->draw_cap) return fail(c, TR_ERR_SIZE, "more than 4096 draws since tr_clear");
    if ((uint64_t)c->tri_total + (uint64_t)mesh->n_tris > (uint64_t)kTriSeqMax)
        return fail(c, TR_ERR_SIZE, "more than 2^28-1 triangles since tr_clear");
    TR_HIP(c, hipSetDevice(c->device));
    c->resolved = false;
    c->resolved_mask = 0;
    if (x) {
        if (x->model < TR_SHADE_REFERENCE || x->model > TR_SHADE_PHONG) return fail(c, TR_ERR_INVALID, "unknown shading model");
        if (x->shadow && x->shadow->device != c->device) return fail(c, TR_ERR_INVALID, "shadow map lives on another device");
    }
    for (const tr_texture* t : {p->diffuse, p->normal_map, x ? x->specular_map : nullptr})
        if (t && std::find(c->frame_textures.begin(), c->frame_textures.end(), t) == c->frame_textures.end()) {
            try { c->frame_textures.push_back(t); }
            catch (...) { return fail(c, TR_ERR_OOM, "texture list"); }
        }
    const bool ext_draw = x && (x->model != TR_SHADE_REFERENCE || x->shadow);
    if (ext_draw || c->ext_table) {
        // (once the table exists every draw writes its entry: a frame may mix both kinds)
        if (!c->ext_table) {
            TR_HIP(c, hipMalloc(&c->ext_table, c->draw_cap * sizeof(DrawExt)));
            TR_HIP(c, hipMemsetAsync(c->ext_table, 0, c->draw_cap * sizeof(DrawExt), c->stream));
        }
        DrawExt e;
        std::memset(&e, 0, sizeof(e));
        if (ext_draw) {
            e.model = x->model;
            if (x->specular_map) { e.spec = x->specular_map->dev; e.sw = x->specular_map->w; e.sh = x->specular_map->h; }
            if (x->shadow) {
                e.shadow = x->shadow->dev; e.mw = x->shadow->w; e.mh = x->shadow->h;
                tr_shadow_map* sm = const_cast<tr_shadow_map*>(x->shadow);
                if (std::find(c->frame_maps.begin(), c->frame_maps.end(), sm) == c->frame_maps.end()) {
                    try { c->frame_maps.push_back(sm); }
                    catch (...) { return fail(c, TR_ERR_OOM, "shadow map list"); }
                }
            }
            e.shininess = x->shininess; e.strength = x->specular_strength;
            e.bias = x->shadow_bias; e.dim = x->shadow_dim;
            std::memcpy(e.to_light.m, x->shadow_from_view, sizeof(e.to_light.m));
            c->frame_has_ext = true;
        }
        hipLaunchKernelGGL(k_put_ext, dim3(1), dim3(64), 0, c->stream, e, c->ext_table, c->draw_count);
        TR_HIP(c, hipGetLastError());
    }
    if (mesh->n_tris == 0) {
        c->draw_count++;
        return TR_OK;
    }
    int rc = ensure_tri_capacity(c, (size_t)c->tri_total + (size_t)mesh->n_tris);
    if (rc != TR_OK) return rc;
    rc = grow_records_for(c, (uint64_t)mesh->n_tris);
    if (rc != TR_OK) return rc;
    // small draws (the reference's meshes): one launch does vertex stage + setup + prefix
    const bool fused = c->fused_front && c->mode != 1 && (uint64_t)mesh->n_tris <= 16384u;  // (at most 256 workgroups, of 64 threads at the least: k_setup<true>)
    // large draws: no shade record at setup time (k_build_shade writes it for the triangles the winner pass flags),
    // and so no normals and each position once per corner: no k_transform launch either
    const bool defer = c->mode == 0 && !fused && (uint64_t)mesh->n_tris >= (uint64_t)c->defer_tris;
    if (!fused && !defer) {
        rc = ensure_scratch(c, &c->ndc, &c->ndc_cap, (size_t)mesh->n_pos * 3);
        if (rc != TR_OK) return rc;
        rc = ensure_scratch(c, &c->nout, &c->nout_cap, (size_t)(mesh->n_nrm ? mesh->n_nrm : 1) * 3);
        if (rc != TR_OK) return rc;
    }

    Mat4 pmv, nmat;
    std::memcpy(pmv.m, p->pmv, sizeof(pmv.m));
    std::memcpy(nmat.m, p->normal_mat, sizeof(nmat.m));
    if (!fused && !defer) {
        ProfScope ps(c, ST_TRANSFORM);
        const int64_t n = mesh->n_pos + mesh->n_nrm;  // (<= 2^29: tr_mesh_create)
        hipLaunchKernelGGL(k_transform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, mesh->pos, (int)mesh->n_pos,
                           mesh->nrm, (int)mesh->n_nrm, pmv, nmat, c->ndc, c->nout);
        TR_HIP(c, hipGetLastError());
    }
    const uint32_t tri_base = c->tri_total;
    const uint32_t n_halves = 2u * (uint32_t)mesh->n_tris;
    c->frame_box.add(draw_pix_box(c, mesh, p->pmv));
    {
        ProfScope ps(c, ST_SETUP);
        SetupArgs sa;
        sa.idx = mesh->idx; sa.pos = mesh->pos; sa.uvw = mesh->uvw; sa.ndc = c->ndc; sa.nout = c->nout;
        sa.n_pos = (int)mesh->n_pos; sa.n_nrm = (int)mesh->n_nrm; sa.n_tris = (int)mesh->n_tris;
        sa.tri_base = tri_base; sa.draw = c->draw_count; sa.W = c->W; sa.H = c->H;
        sa.vpm = viewport_matrix(c->W, c->H);
        sa.nmat = nmat;
        sa.rast = c->rast; sa.shade = c->shade; sa.half_cnt = c->half_cnt; sa.flags = c->flags;
        sa.tile_cnt = c->tile_cnt; sa.tiles_x = c->tiles_x; sa.tiles_y = c->tiles_y; sa.tw_shift = c->tw_shift; sa.th_shift = c->th_shift;
        sa.small_res = 0;
        sa.small_flag = c->small_flag; sa.small_grp = c->small_grp;
        sa.y_lo = c->y_lo; sa.y_hi = c->y_hi; sa.mode = c->mode == 1 ? 2 : 1;  // (the row raster uses the global raster's slot sequence)
        if (c->mode == 0 && c->small_path && (uint64_t)mesh->n_tris >= (uint64_t)c->small_draw_tris) {
            sa.mode |= 4;  // triangles of a few pixels go to k_small and own no slots
            sa.small_res = c->small_res;
            c->small_any = true;
        }
        // the layout of the depth / key planes is settled by the frame's first draw (every pass of the frame shares it):
        // tiles of 4 x 2 pixels when that draw is a mesh of small triangles (plane_index, tr_device.h)
        if (tri_base == 0 && c->draw_count == 0)
            c->frame_tiled = c->plane_tiles && (sa.mode & 4) && (c->W & 3) == 0 && (c->H & 1) == 0;
        sa.mid_samples = c->mid_samples;
        sa.slot_len = slot_len_for(c);
        if (use_mid_path(c, (uint64_t)mesh->n_tris, (uint64_t)mesh->n_tris)) {
            sa.mode |= 16;  // triangles of up to 64 scanlines per half and a few hundred samples: one wave per half (k_mid)
            c->mid_any = true;
        }
        sa.need_shade = c->need_shade;
        if (defer) sa.mode |= 8;  // no TriShade now: k_build_shade writes it for the triangles the winner pass flags
        sa.per_corner = defer ? 1 : 0;
        sa.bounds = mesh->bounds;
        sa.z_split = std::nan("");
        if (c->depth_split && (sa.mode & (4 | 16))) {
            sa.z_split = split_depth(c, mesh, p->pmv);
            if (sa.z_split == sa.z_split) c->split_any = true;
        }
        sa.cull = nullptr;
        if (defer && mesh->bounds && (c->y_lo > 0 || c->y_hi < c->H)) {
            // a band context: whole clusters of the mesh that miss its rows are dropped before anything of theirs is read
            const size_t n_clusters = ((size_t)mesh->n_tris + 63) / 64;
            if (c->cull_used + n_clusters > c->cull_cap) {
                // (grown between frames only: an earlier draw of this frame may still hold a pointer into it)
                if (c->cull_used == 0) {
                    (void)hipFree(c->cluster_cull);
                    c->cluster_cull = nullptr; c->cull_cap = 0;
                    size_t cap = 4096;
                    while (cap < n_clusters) cap *= 2;
                    TR_HIP(c, hipMalloc(&c->cluster_cull, cap));
                    c->cull_cap = cap;
                }
            }
            if (c->cull_used + n_clusters <= c->cull_cap) {
                sa.cull = c->cluster_cull + c->cull_used;
                c->cull_used += n_clusters;
            }
        }
        DrawDev& d = sa.params;
        fill_draw_dev(d, p);
        sa.draw_table = c->draw_table;
        if (c->draw_count == 0) c->draw0 = d;
        sa.nrm = mesh->nrm; sa.pmv = pmv; sa.ticket = c->ticket; sa.half_off = c->half_off;
        if (++c->scan_tag == 0) c->scan_tag = 1;
        sa.scan_tag = c->scan_tag;
        sa.scan_expect = c->scan_tag;
        sa.spin_limit = 1u << 24;
        if (c->debug_prefix_stall) {  // (tests only: the pollers wait for a tag nobody publishes)
            sa.scan_expect = c->scan_tag ^ 0x40000000u;
            sa.spin_limit = 1u << 10;
        }
        sa.scan_lo = 2u * tri_base; sa.n_halves = n_halves;
        sa.views = nullptr; sa.n_views = 0;
        if (defer) {
            try { c->deferred.push_back({mesh, sa}); }
            catch (...) { return fail(c, TR_ERR_OOM, "deferred draw list"); }
        }
        if (fused) {
            // (measured: 128 threads per workgroup; 64 / 256 cost the orbit rehearsal 1 % / 4 % with four contexts)
            static const int st = env_diag("TR_SETUP_THREADS") ? std::atoi(env_diag("TR_SETUP_THREADS")) : 128;
            const int threads = (st == 64 || st == 256) ? st : 128;
            hipLaunchKernelGGL((k_setup<true>), dim3((unsigned)((mesh->n_tris + threads - 1) / threads)), dim3(threads), 0, c->stream, sa);
        }
        else {
            unsigned blocks = (unsigned)((mesh->n_tris + 255) / 256);
            if (sa.cull) {
                const uint32_t n_clusters = (uint32_t)(((size_t)mesh->n_tris + 63) / 64);
                hipLaunchKernelGGL(k_cull_clusters, dim3((n_clusters * 8u + 255u) / 256u), dim3(256), 0, c->stream, sa, n_clusters);
                TR_HIP(c, hipGetLastError());
            }
            hipLaunchKernelGGL((k_setup<false>), dim3(blocks), dim3(256), 0, c->stream, sa);
        }
        TR_HIP(c, hipGetLastError());
        if (c->mode != 1 && !fused) {
            // prefix of scanline counts, continuing from the previous draw's total
            rc = launch_scan(c, 2u * tri_base, n_halves, sa.cull);
            if (rc != TR_OK) return rc;
        }
    }
    c->keys_valid = false;
    if (c->mode == 0 && c->depth_upto == tri_base) {
        // depth plane is current up to this draw: add this draw's samples now.  (If an earlier
        // resolve wiped the planes, the depth pass of the whole frame is redone at tr_resolve.)
        ProfScope ps(c, ST_DEPTH);
        // many small triangles: build the records with one thread per slot first
        const bool big = (uint64_t)mesh->n_tris >= (uint64_t)c->big_draw_tris;
        if (big) {
            RasterArgs ra;
            fill_raster_args(c, ra, 2u * tri_base, 2u * tri_base + n_halves);
            hipLaunchKernelGGL(k_scanlines, dim3(kScanlineBlocks), dim3(256), 0, c->stream, ra);
            TR_HIP(c, hipGetLastError());
        }
        rc = launch_raster(c, 0, 2u * tri_base, 2u * tri_base + n_halves, false, big);
        if (rc != TR_OK) return rc;
        c->depth_upto = tri_base + (uint32_t)mesh->n_tris;
        c->planes_clean = false;
    }
    c->tri_total += (uint32_t)mesh->n_tris;
    c->draw_count++;
    return TR_OK;
}

int tr_select_view(tr_context* c, int view) {
    if (!c) return TR_ERR_INVALID;
    if (view < 0 || view >= c->n_views) return fail(c, TR_ERR_INVALID, "tr_select_view: no such view in the current frame");
    c->sel_view = view;
    return TR_OK;
}

// The reference renders one view per iteration of its loop (src/main.rs:57-116,272-273: clear_screen, camera,
// render_mesh_component).  Here n_views iterations over ONE mesh are a single frame of n_views plane slices: one
// setup launch over (view, triangle), one slot sequence, one depth pass, one winner pass, one shade launch.
static int render_views_impl(tr_context* c, int n_meshes, const tr_mesh* const* meshes, const tr_draw_params* params, int n_views, uint32_t out_mask);
static int render_views_guarded(tr_context* c, int n_meshes, const tr_mesh* const* meshes, const tr_draw_params* params, int n_views, uint32_t out_mask) {
    const int rc = render_views_impl(c, n_meshes, meshes, params, n_views, out_mask);
    if (rc != TR_OK && c) {
        // a batch that failed half way (capacity, upload, launch) must not read back as a resolved, empty frame: the
        // context is left as tr_clear leaves it, and keeps the failure's text
        std::string why;
        try { why = c->err; } catch (...) {}
        (void)tr_clear(c);
        try { c->err = why; } catch (...) {}
    }
    return rc;
}
int tr_render_views(tr_context* c, const tr_mesh* mesh, const tr_draw_params* params, int n_views, uint32_t out_mask) {
    return render_views_guarded(c, 1, &mesh, params, n_views, out_mask);
}
int tr_render_views_multi(tr_context* c, int n_meshes, const tr_mesh* const* meshes, const tr_draw_params* params, int n_views, uint32_t out_mask) {
    return render_views_guarded(c, n_meshes, meshes, params, n_views, out_mask);
}
static int render_views_impl(tr_context* c, int n_meshes, const tr_mesh* const* meshes, const tr_draw_params* params, int n_views, uint32_t out_mask) {
    if (!c || !meshes || !params) return TR_ERR_INVALID;
    if (c->max_views < 2) return fail(c, TR_ERR_INVALID, "tr_render_views needs a context made by tr_create_views");
    if (n_views < 1 || n_views > c->max_views) return fail(c, TR_ERR_SIZE, "tr_render_views: n_views outside [1, max_views]");
    if (n_meshes < 1 || n_meshes > kMaxBatchMeshes) return fail(c, TR_ERR_SIZE, "tr_render_views_multi: 1 to 4 meshes per batch");
    if ((uint64_t)n_views * (uint64_t)n_meshes > (uint64_t)c->draw_cap) return fail(c, TR_ERR_SIZE, "more than 4096 draws (views x meshes) in one batch");
    if (c->mode != 0) return fail(c, TR_ERR_UNSUPPORTED, "tr_render_views uses the global raster");
    uint64_t per_view = 0;
    for (int m = 0; m < n_meshes; ++m) {
        if (!meshes[m]) return fail(c, TR_ERR_INVALID, "null mesh");
        per_view += (uint64_t)meshes[m]->n_tris;
    }
    const uint64_t total = (uint64_t)n_views * per_view;
    if (total > (uint64_t)kTriSeqMax) return fail(c, TR_ERR_SIZE, "more than 2^28-1 triangles in one batch of views");
    const size_t n_draws = (size_t)n_views * (size_t)n_meshes;
    for (size_t d = 0; d < n_draws; ++d)
        if (meshes[d % (size_t)n_meshes]->any_uv && (!params[d].diffuse || (params[d].use_normal_map && !params[d].normal_map)))
            return fail(c, TR_ERR_INVALID, "mesh has texture coordinates but a texture is missing");
    int rc = tr_clear(c);
    if (rc != TR_OK) return rc;
    if (per_view == 0) {  // nothing to draw: every slice of the batch is background
        c->n_views = n_views;
        if (c->sel_view >= n_views) c->sel_view = 0;
        c->draw_count = (uint32_t)n_draws;
        return tr_resolve(c, out_mask);
    }
    rc = ensure_tri_capacity(c, (size_t)total);
    if (rc != TR_OK) return rc;
    rc = grow_records_for(c, total);
    if (rc != TR_OK) return rc;
    rc = ensure_view_capacity(c, n_draws);
    if (rc != TR_OK) return rc;
    for (size_t d = 0; d < n_draws; ++d)
        for (const tr_texture* t : {params[d].diffuse, params[d].normal_map})
            if (t && std::find(c->frame_textures.begin(), c->frame_textures.end(), t) == c->frame_textures.end()) {
                try { c->frame_textures.push_back(t); }
                catch (...) { return fail(c, TR_ERR_OOM, "texture list"); }
            }
    // the (view, mesh) constants: written into one of two page-locked buffers (the copy from the other may still be
    // in flight when batches are issued back to back), uploaded on the stream
    const int turn = c->view_turn;
    c->view_turn ^= 1;
    TR_HIP(c, hipEventSynchronize(c->view_ev[turn]));
    ViewDev* stage = c->view_stage[turn];
    for (size_t d = 0; d < n_draws; ++d) {
        const tr_mesh* mesh = meshes[d % (size_t)n_meshes];
        c->frame_box.add(draw_pix_box(c, mesh, params[d].pmv));
        std::memcpy(stage[d].pmv.m, params[d].pmv, sizeof(stage[d].pmv.m));
        std::memcpy(stage[d].nmat.m, params[d].normal_mat, sizeof(stage[d].nmat.m));
        fill_draw_dev(stage[d].draw, &params[d]);
        stage[d].z_split = c->depth_split ? split_depth(c, mesh, params[d].pmv) : std::nan("");
    }
    TR_HIP(c, hipMemcpyAsync(c->view_table, stage, n_draws * sizeof(ViewDev), hipMemcpyHostToDevice, c->stream));
    TR_HIP(c, hipEventRecord(c->view_ev[turn], c->stream));
    c->draw0 = stage[0].draw;
    const uint32_t n_halves = 2u * (uint32_t)total;
    {
        ProfScope ps(c, ST_SETUP);
        SetupArgs sa;
        std::memset(&sa, 0, sizeof(sa));
        const tr_mesh* mesh = meshes[0];
        sa.idx = mesh->idx; sa.pos = mesh->pos; sa.uvw = mesh->uvw; sa.nrm = mesh->nrm;
        sa.n_pos = (int)mesh->n_pos; sa.n_nrm = (int)mesh->n_nrm; sa.n_tris = (int)per_view;
        sa.n_meshes = (uint32_t)n_meshes;
        uint64_t off = 0;
        for (int m = 0; m < n_meshes; ++m) {
            sa.mesh_off[m] = (uint32_t)off;
            off += (uint64_t)meshes[m]->n_tris;
            sa.m_idx[m] = meshes[m]->idx; sa.m_pos[m] = meshes[m]->pos; sa.m_nrm[m] = meshes[m]->nrm; sa.m_uvw[m] = meshes[m]->uvw;
        }
        for (int m = n_meshes; m <= kMaxBatchMeshes; ++m) sa.mesh_off[m] = (uint32_t)off;
        sa.tri_base = 0; sa.draw = 0; sa.W = c->W; sa.H = c->H;
        sa.vpm = viewport_matrix(c->W, c->H);
        sa.rast = c->rast; sa.shade = c->shade; sa.half_cnt = c->half_cnt; sa.flags = c->flags;
        sa.tile_cnt = c->tile_cnt; sa.tiles_x = c->tiles_x; sa.tiles_y = c->tiles_y; sa.tw_shift = c->tw_shift; sa.th_shift = c->th_shift;
        sa.small_flag = c->small_flag; sa.small_grp = c->small_grp;
        sa.y_lo = 0; sa.y_hi = c->H; sa.mode = 1;
        if (c->small_path && per_view >= (uint64_t)c->small_draw_tris) {
            sa.mode |= 4;  // triangles of a few pixels go to k_small and own no slots
            sa.small_res = c->small_res;
            c->small_any = true;
        }
        sa.mid_samples = c->mid_samples;
        sa.slot_len = slot_len_for(c);
        if (use_mid_path(c, per_view, total)) {
            sa.mode |= 16;  // triangles of up to 64 scanlines per half and a few hundred samples: one wave per half (k_mid)
            c->mid_any = true;
        }
        sa.draw_table = c->draw_table;
        sa.half_off = c->half_off;
        sa.views = c->view_table; sa.n_views = (uint32_t)n_views;
        if (c->depth_split && (sa.mode & (4 | 16))) c->split_any = true;
        hipLaunchKernelGGL((k_setup<false, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, sa);
        TR_HIP(c, hipGetLastError());
        rc = launch_scan(c, 0u, n_halves);
        if (rc != TR_OK) return rc;
    }
    // (the frame's bookkeeping only now: everything that can fail before a sample is drawn has succeeded)
    c->n_views = n_views;
    if (c->sel_view >= n_views) c->sel_view = 0;
    c->draw_count = (uint32_t)n_draws;
    c->view_meshes = (uint32_t)n_meshes;
    c->view_tris = (uint32_t)per_view;
    c->tri_total = (uint32_t)total;
    {
        ProfScope ps(c, ST_DEPTH);
        const bool big = total >= (uint64_t)c->big_draw_tris;  // many slots: build the records with one thread per slot first
        if (big) {
            RasterArgs ra;
            fill_raster_args(c, ra, 0u, n_halves);
            hipLaunchKernelGGL(k_scanlines, dim3(kScanlineBlocks), dim3(256), 0, c->stream, ra);
            TR_HIP(c, hipGetLastError());
        }
        rc = launch_raster(c, 0, 0u, n_halves, false, big);
        if (rc != TR_OK) return rc;
        c->depth_upto = c->tri_total;
        c->planes_clean = false;
    }
    return tr_resolve(c, out_mask);
}

// Tile and row raster: a bin list that proved too small is grown and the resolve redone HERE (one read of the
// status word, i.e. a host wait per resolve in those two modes), so that what tr_resolve leaves in the planes
// is the frame -- a caller that consumes tr_device_ptr_rgba8 on the stream and clears without ever reading back
// would otherwise get a frame that drew nothing, and the aborted frame's fill cursors would leak into the next.
int tr_resolve(tr_context* c, uint32_t out_mask) {
    int rc = resolve_frame(c, out_mask, true);
    if (rc == TR_OK && c->mode != 0) rc = settle_bin_overflow(c);
    return rc;
}

// colour == false (shadow snapshots of an unresolved frame): winner selection and the planes asked
// for, no shading and no colour plane; RGBA8 is then not part of resolved_mask, so that a later read
// of the image resolves again.
static int resolve_frame(tr_context* c, uint32_t out_mask, bool colour) {
    if (!c) return TR_ERR_INVALID;
    if ((c->frame_lost_texture || c->frame_lost_mesh) && c->tri_total)
        return fail(c, TR_ERR_INVALID, "a texture or mesh used by this frame's draws was destroyed: the frame cannot be resolved again (tr_clear starts a new one)");
    TR_HIP(c, hipSetDevice(c->device));
    size_t npix = c->npix_all;
    out_mask &= (TR_OUT_RGBA8 | TR_OUT_Z | TR_OUT_TRI_ID | TR_OUT_RGBA_F64 | TR_OUT_KEY);
    if (colour || (out_mask & TR_OUT_RGBA_F64)) out_mask |= TR_OUT_RGBA8;
    else out_mask &= ~(uint32_t)TR_OUT_RGBA8;
    if ((out_mask & TR_OUT_Z) && !c->z_out) TR_HIP(c, hipMalloc(&c->z_out, npix * sizeof(double)));
    if ((out_mask & TR_OUT_TRI_ID) && !c->tri_out) TR_HIP(c, hipMalloc(&c->tri_out, npix * sizeof(uint32_t)));
    if ((out_mask & TR_OUT_RGBA_F64) && !c->rgba_f64) TR_HIP(c, hipMalloc(&c->rgba_f64, npix * 4 * sizeof(double)));
    if ((out_mask & TR_OUT_KEY) && !c->key_out) TR_HIP(c, hipMalloc(&c->key_out, npix * sizeof(unsigned long long)));
    if (c->tri_total && c->mode == 0) {
        if (c->depth_upto != c->tri_total) {  // planes were wiped by an earlier resolve: redo the depth pass
            ProfScope ps(c, ST_DEPTH);
            int rc = launch_raster(c, 0, 0u, 2u * c->tri_total);
            if (rc != TR_OK) return rc;
            c->depth_upto = c->tri_total;
            c->planes_clean = false;
            c->keys_valid = false;
        }
        if (!c->keys_valid) {
            ProfScope ps(c, ST_WINNER);
            if (c->wkey_dirty)  // draws were added after a resolve that kept its keys: start the winners over
                TR_HIP(c, hipMemsetAsync(c->wkey, 0xFF, npix * sizeof(unsigned long long), c->stream));
            // (that earlier winner pass also consumed the live flags of its chunks: no skipping then)
            int rc = launch_raster(c, 1, 0u, 2u * c->tri_total, !c->wkey_dirty);
            if (rc != TR_OK) return rc;
            c->keys_valid = true;
            c->wkey_dirty = true;
        }
    }
    if (c->tri_total && c->mode == 2 && !c->keys_valid) {
        RowArgs ra;
        ra.half_off = c->half_off; ra.rast = c->rast; ra.n_half = 2u * c->tri_total;
        ra.rec = c->rec; ra.rec_cap = c->rec_cap;
        ra.bin_cnt = c->bin_cnt; ra.bin_off = c->bin_off; ra.bin_list = c->bin_list; ra.n_padded = c->n_bins_padded;
        ra.refs = c->refs; ra.ref_cap = c->ref_cap;
        ra.wkey = c->wkey; ra.blk_touched = c->blk_flags; ra.blocks_x = c->blocks_x;
        ra.W = c->W; ra.H = c->H; ra.nseg = c->nseg; ra.y_lo = c->y_lo; ra.y_hi = c->y_hi;
        ra.counters = c->counters; ra.flags = c->flags;
        ra.trace = nullptr; ra.trace_cap = 0;
        if (c->resolved)  // a second raster of the same frame (more draws, or more planes): count afresh
            TR_HIP(c, hipMemsetAsync(c->counters, 0, kCounterSlots * 8 * sizeof(unsigned long long), c->stream));
        {
            ProfScope ps(c, ST_DEPTH);  // stage slot 3 = records + binning in row mode
            hipLaunchKernelGGL((k_row_bin<false>), dim3(kRowBinBlocks), dim3(256), 0, c->stream, ra);
            hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(kTileScanThreads), 0, c->stream, c->bin_cnt, c->bin_off,
                               c->bin_list, (uint32_t)(c->nseg * c->H), c->n_bins_padded, c->ref_cap, c->flags);
            hipLaunchKernelGGL((k_row_bin<true>), dim3(kRowBinBlocks), dim3(256), 0, c->stream, ra);
            TR_HIP(c, hipGetLastError());
        }
        {
            ProfScope ps(c, ST_WINNER);  // stage slot 4 = row raster
            int grid = kRowGrid;
            if (const char* e = env_diag("TR_ROW_GRID")) grid = std::atoi(e);
            ra.trace = nullptr; ra.trace_cap = 0;
            static const char* trace_path = env_diag("TR_RASTER_TRACE");
            static unsigned long long* trace_buf = nullptr;
            constexpr uint32_t kTraceCap = 1u << 16;
            if (trace_path) {
                if (!trace_buf) TR_HIP(c, hipMalloc(&trace_buf, kTraceCap * 4 * sizeof(unsigned long long)));
                TR_HIP(c, hipMemsetAsync(trace_buf, 0, kTraceCap * 4 * sizeof(unsigned long long), c->stream));
                ra.trace = trace_buf; ra.trace_cap = kTraceCap;
            }
            hipLaunchKernelGGL(k_row_raster, dim3(grid), dim3(kRowThreads), 0, c->stream, ra);
            TR_HIP(c, hipGetLastError());
            if (trace_path) {
                std::vector<unsigned long long> h((size_t)kTraceCap * 4);
                TR_HIP(c, hipStreamSynchronize(c->stream));
                TR_HIP(c, hipMemcpy(h.data(), trace_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::string fn = std::string(trace_path) + ".rows";
                if (FILE* f = std::fopen(fn.c_str(), "wb")) { std::fwrite(h.data(), sizeof(unsigned long long), h.size(), f); std::fclose(f); }
            }
        }
        c->keys_valid = true;
        c->planes_clean = false;
    }
    if (!c->resolved && c->mode == 1) {
        const uint32_t n_tiles = (uint32_t)(c->tiles_x * c->tiles_y);
        {
            ProfScope ps(c, ST_DEPTH);  // stage slot 3 = binning in tile mode
            hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(kTileScanThreads), 0, c->stream, c->tile_cnt, c->tile_off,
                               c->tile_list, n_tiles, c->n_tiles_padded, c->pair_cap, c->flags);
            if (c->tri_total)
                hipLaunchKernelGGL(k_bin_fill, dim3((unsigned)(((uint64_t)c->tri_total * kFillLanes + 255) / 256)), dim3(256), 0, c->stream, c->rast,
                                   c->tri_total, c->tile_off, c->tile_cnt, c->pairs, c->pair_cap, c->tiles_x, c->y_lo,
                                   c->y_hi, c->th_shift);
            TR_HIP(c, hipGetLastError());
        }
        if (c->tri_total) {
            ProfScope ps(c, ST_WINNER);  // stage slot 4 = tile raster in tile mode
            TileArgs ta;
            ta.tile_list = c->tile_list; ta.n_padded = c->n_tiles_padded;
            ta.tile_off = c->tile_off; ta.pairs = c->pairs; ta.pair_cap = c->pair_cap; ta.rast = c->rast;
            ta.wkey = c->wkey; ta.W = c->W; ta.H = c->H; ta.tiles_x = c->tiles_x;
            ta.y_lo = c->y_lo; ta.y_hi = c->y_hi;
            ta.counters = c->counters; ta.flags = c->flags;
            ta.debug = 0;
            if (const char* e = env_diag("TR_TILE_DEBUG")) ta.debug = std::atoi(e);
            int grid = kTileGrid;
            if (const char* e = env_diag("TR_TILE_GRID")) grid = std::atoi(e);
            hipLaunchKernelGGL(tile_kernel(c), dim3(grid), dim3(kTileThreads), tile_lds_bytes(c), c->stream, ta);
            TR_HIP(c, hipGetLastError());
            TR_HIP(c, hipGetLastError());
        }
    }
    if ((out_mask & TR_OUT_RGBA8) && c->mode == 0) {
        // shade records of the large draws, for the triangles the winner pass flagged
        for (auto& d : c->deferred) {
            ProfScope ps(c, ST_SETUP);
            SetupArgs ba = d.sa;
            ba.rast = c->rast; ba.shade = c->shade;  // (the arrays may have grown since the draw)
            unsigned blocks = (unsigned)((ba.n_tris + 255) / 256);
            hipLaunchKernelGGL(k_build_shade, dim3(blocks), dim3(256), 0, c->stream, ba, c->need_shade);
            TR_HIP(c, hipGetLastError());
        }
    }
    {
        ProfScope ps(c, ST_SHADE);
        ShadeArgs sa;
        sa.wkey = c->wkey; sa.depth = c->mode == 0 ? c->depth : nullptr; sa.key_out = c->key_out;
        // the cheap path (RGBA8 only) leaves the planes cleared so the next tr_clear is free
        sa.self_clean = (c->mode != 1 && out_mask == TR_OUT_RGBA8) ? 1 : 0;
        sa.rast = c->rast; sa.shade = c->shade; sa.draws = c->draw_table;
        sa.nan = c->mode == 0 ? c->nan : nullptr;
        sa.tile_off = c->mode == 1 ? c->tile_off : nullptr;
        sa.tiles_x = c->tiles_x; sa.tw_shift = c->tw_shift; sa.th_shift = c->th_shift;
        sa.W = c->W; sa.H = c->H; sa.out_mask = out_mask; sa.rgba8 = c->rgba8; sa.z_out = c->z_out;
        sa.tri_out = c->tri_out; sa.rgba_f64 = c->rgba_f64; sa.flags = c->flags; sa.cov_counters = c->cov_counters;
        sa.blk_flags = c->blk_flags;
        sa.draw0 = c->draw0;

        sa.skip_clean = (sa.self_clean && c->skip_clean) ? 1 : 0;
        if (c->cov_dirty)  // a second resolve of the same frame: tr_clear zeroes them otherwise
            TR_HIP(c, hipMemsetAsync(c->cov_counters, 0, kCounterSlots * 8 * sizeof(unsigned long long), c->stream));
        c->cov_dirty = true;
        sa.ext = c->ext_table;
        const bool views = c->max_views > 1;
        sa.view_blk_rows = c->blk_rows_per_view;
        sa.view_meshes = c->view_meshes;
        sa.view_pix = (uint32_t)((size_t)c->W * c->H);
        dim3 sgrid((c->W + kShadeBW - 1) / kShadeBW, c->blk_rows_per_view, (unsigned)(views ? c->n_views : 1));
        sa.blk_y0 = 0;
        sa.blk_x0 = 0;
        sa.blocks_x = sgrid.x;
        bool boxed = false, nothing = false;
        sa.launch_rows = c->blk_rows_per_view;
        sa.tiled = c->frame_tiled ? 1u : 0u;
        if (c->shade_box && sa.skip_clean && !shade_trace_on() && !c->frame_has_ext) {
            // the cheap resolve: a block outside the frame's screen box holds no key, and one outside the box the last
            // launch may have painted already shows the background -- such blocks (3/4 of the 4K head frame) would
            // read their two flag bytes and leave; they are not launched at all
            PixBox cur = c->frame_box;
            cur.y0 = std::max(cur.y0, c->y_lo);
            cur.y1 = std::min(cur.y1, c->y_hi);
            if (cur.empty()) cur = PixBox();
            PixBox go = cur;
            go.add(c->paint_box);
            // (a batch of fewer views than the context has slices leaves the other slices as they were: what they
            // may show stays in the box until a launch covers every slice again)
            if (views && c->n_views < c->max_views) c->paint_box.add(cur);
            else c->paint_box = cur;
            boxed = true;
            // (pixel (0, 0) -- where samples with NaN coordinates land, see below -- may hold something whenever the frame has
            // triangles, and may have to be wiped whenever the last boxed launch's frame had: an empty box alone does not
            // make an empty launch)
            const bool origin = c->tri_total > 0 || c->origin_maybe_painted;
            c->origin_maybe_painted = c->tri_total > 0;
            if (go.empty()) {
                if (origin) { sa.blk_x0 = 1; sa.blk_y0 = 0; sgrid.x = 1; sgrid.y = 0; sa.launch_rows = 0; }  // the workgroup of block (0, 0) only
                else nothing = true;
            } else {
                sa.blk_x0 = (uint32_t)(go.x0 / kShadeBW);
                sa.blk_y0 = (uint32_t)(go.y0 / kShadeBH);
                sgrid.x = (unsigned)((go.x1 + kShadeBW - 1) / kShadeBW) - sa.blk_x0;
                sgrid.y = (unsigned)((go.y1 + kShadeBH - 1) / kShadeBH) - sa.blk_y0;
                // (workgroups go to the eight XCDs in turn by their linear index: with a row of a multiple of 8 blocks --
                // or of any even number, in part -- a block COLUMN always lands on the same XCD, and the columns an object
                // covers then load the XCDs unevenly: shade launches of a 32-block-wide box took up to 2.5x as long as
                // full-frame ones beside other streams' kernels.  An odd row length walks every column round the XCDs.)
                if ((sgrid.x & 1u) == 0) {
                    if (sa.blk_x0 + sgrid.x < sa.blocks_x) sgrid.x += 1;
                    else if (sa.blk_x0 > 0) { sa.blk_x0 -= 1; sgrid.x += 1; }
                }
                sa.launch_rows = sgrid.y;
                // (the same box in every slice of the batch: the views are the grid's z)
            }
        } else {
            c->paint_box.x0 = 0; c->paint_box.y0 = 0; c->paint_box.x1 = c->W; c->paint_box.y1 = c->H;
            c->origin_maybe_painted = false;  // (inside paint_box)
        }
        if (!boxed && !views && !shade_trace_on() && out_mask == TR_OUT_RGBA8 && (c->y_lo > 0 || c->y_hi < c->H || c->shaded_lo > 0 || c->shaded_hi < c->H)) {
            // a band context (RGBA8 only: the other planes are rewritten whole): nothing lands outside [y_lo, y_hi); the
            // rows the previous launch covered are included so that a band that moved (balanced_bands re-cuts between frames) leaves no stale colour behind
            const int lo = c->y_lo < c->shaded_lo ? c->y_lo : c->shaded_lo, hi = c->y_hi > c->shaded_hi ? c->y_hi : c->shaded_hi;
            sa.blk_y0 = (uint32_t)(lo / kShadeBH);
            sgrid.y = (unsigned)((hi + kShadeBH - 1) / kShadeBH) - sa.blk_y0;
        }
        c->shaded_lo = c->y_lo;
        c->shaded_hi = c->y_hi;
        sa.trace = nullptr;
        c->last_shade_grid[0] = nothing ? 0u : sgrid.x; c->last_shade_grid[1] = nothing ? 0u : sgrid.y; c->last_shade_grid[2] = nothing ? 0u : sgrid.z;
        sa.launch_rows = sgrid.y;
        // Pixel (0, 0) is where every sample with NaN coordinates lands (a triangle whose corners share one y: mod.rs:435-441),
        // whatever the screen box of the frame's meshes: a boxed launch that does not hold the frame's first block gets one more
        // row of workgroups, the first of which takes that block (its flags say whether there is anything to paint or to
        // repaint; the others leave at once).  A band that does not own row 0 receives no such sample, but may have to wipe
        // what an earlier frame of this context, with other rows, left there.
        if (boxed && !nothing && (sa.blk_x0 != 0u || sa.blk_y0 != 0u)) sgrid.y += 1;
        static const char* shade_trace = env_diag("TR_SHADE_TRACE");  // debugging aid, tools/shade_trace.py
        (void)shade_trace;
        // the fused colour arithmetic (+-1 LSB) unless the caller asked for the reference's operations or for the f64 colour plane
        const bool fast = !c->colour_exact && !(out_mask & TR_OUT_RGBA_F64);
        if (views) {
            if (nothing) {}
            else if (fast) hipLaunchKernelGGL((k_shade_fused<true>), sgrid, dim3(kShadeBW * kShadeBH), 0, c->stream, sa);
            else hipLaunchKernelGGL((k_shade<false, false, true>), sgrid, dim3(kShadeBW * kShadeBH), 0, c->stream, sa);
        }
#ifdef TR_DIAG
        else if (shade_trace && !c->frame_has_ext) {
            const size_t words = (size_t)sgrid.x * sgrid.y * 4 * 4;
            static unsigned long long* buf = nullptr;
            static size_t buf_words = 0;
            if (buf_words < words) {
                (void)hipFree(buf);
                TR_HIP(c, hipMalloc(&buf, words * sizeof(unsigned long long)));
                buf_words = words;
            }
            TR_HIP(c, hipMemsetAsync(buf, 0, words * sizeof(unsigned long long), c->stream));
            sa.trace = buf;
            hipLaunchKernelGGL((k_shade<false, true>), sgrid, dim3(kShadeBW * kShadeBH), 0, c->stream, sa);
            TR_HIP(c, hipGetLastError());
            std::vector<unsigned long long> h(words);
            TR_HIP(c, hipStreamSynchronize(c->stream));
            TR_HIP(c, hipMemcpy(h.data(), buf, words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (FILE* f = std::fopen(shade_trace, "wb")) { std::fwrite(h.data(), sizeof(unsigned long long), words, f); std::fclose(f); }
        }
#endif
        else if (c->frame_has_ext) {
            for (tr_shadow_map* sm : c->frame_maps) TR_HIP(c, hipStreamWaitEvent(c->stream, sm->ready, 0));
            hipLaunchKernelGGL((k_shade<true>), sgrid, dim3(kShadeBW * kShadeBH), 0, c->stream, sa);
            for (tr_shadow_map* sm : c->frame_maps) {
                std::lock_guard<std::mutex> g(sm->lock);
                auto it = std::find_if(sm->consumed.begin(), sm->consumed.end(), [&](const auto& ce) { return ce.first == c; });
                if (it == sm->consumed.end()) {
                    hipEvent_t ev = nullptr;
                    TR_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                    try { sm->consumed.emplace_back(c, ev); }
                    catch (...) { (void)hipEventDestroy(ev); return fail(c, TR_ERR_OOM, "shadow map consumer list"); }
                    it = sm->consumed.end() - 1;
                }
                TR_HIP(c, hipEventRecord(it->second, c->stream));
            }
        }
        else if (nothing) {}
        else if (fast) TR_LAUNCH_STAGE(c, (k_shade_fused<false>), sgrid, dim3(kShadeBW * kShadeBH), sa);
        else TR_LAUNCH_STAGE(c, (k_shade<false>), sgrid, dim3(kShadeBW * kShadeBH), sa);
        TR_HIP(c, hipGetLastError());
    }
    if (c->mode != 1 && out_mask == TR_OUT_RGBA8) {
        c->planes_clean = true;
        c->depth_upto = 0;
        c->keys_valid = false;
        c->wkey_dirty = false;
    }
    c->resolved = true;
    c->resolved_mask = out_mask;
    return TR_OK;
}

int tr_read_z_f64(tr_context* c, double* out, int layout) {
    if (!c || !out) return TR_ERR_INVALID;
    int rc = need_plane(c, TR_OUT_Z);
    if (rc != TR_OK) return rc;
    return read_plane<double>(c, c->z_out, out, layout, 1);
}

int tr_read_triangle_id_u32(tr_context* c, uint32_t* out, int layout) {
    if (!c || !out) return TR_ERR_INVALID;
    int rc = need_plane(c, TR_OUT_TRI_ID);
    if (rc != TR_OK) return rc;
    return read_plane<uint32_t>(c, c->tri_out, out, layout, 1);
}

int tr_read_winner_key_u64(tr_context* c, uint64_t* out, int layout) {
    if (!c || !out) return TR_ERR_INVALID;
    int rc = need_plane(c, TR_OUT_KEY);
    if (rc != TR_OK) return rc;
    return read_plane<uint64_t>(c, (const uint64_t*)c->key_out, out, layout, 1);
}

int tr_read_rgba8(tr_context* c, uint8_t* out, int flip) {
    if (!c || !out) return TR_ERR_INVALID;
    int rc = need_plane(c, TR_OUT_RGBA8);
    if (rc != TR_OK) return rc;
    const size_t bytes = (size_t)c->W * c->H * 4;
    const uint32_t* src = c->rgba8 + (size_t)c->sel_view * c->W * c->H;
    if (flip) {  // src/helpers/mod.rs:214-218: image row = H-1-y, done on the device so that one copy delivers the image
        if (!c->flip_buf) TR_HIP(c, hipMalloc(&c->flip_buf, bytes));
        hipLaunchKernelGGL(k_flip_rows, dim3(2048), dim3(256), 0, c->stream, src, c->flip_buf, c->W, c->H);
        TR_HIP(c, hipGetLastError());
        src = c->flip_buf;
    }
    // (a buffer from tr_host_alloc is pinned: the copy then runs at the link's rate without staging)
    TR_HIP(c, hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, c->stream));
    TR_HIP(c, hipStreamSynchronize(c->stream));
    return TR_OK;
}

// tr_read_rgba8 for callers that cannot allocate: no hipMalloc / hipFree on any path, nothing is resolved
// (a resolve may grow buffers), the flipped image is staged in the per-device static array.
int tr_read_rgba8_static(tr_context* c, uint8_t* out, int flip) {
    if (!c || !out) return TR_ERR_INVALID;
    TR_HIP(c, hipSetDevice(c->device));
    if (!(c->resolved && (c->resolved_mask & TR_OUT_RGBA8)))
        return fail(c, TR_ERR_INVALID, "tr_read_rgba8_static: call tr_resolve with TR_OUT_RGBA8 first (this call resolves nothing)");
    const size_t npix = (size_t)c->W * c->H;
    if (npix > kStaticStagePixels)
        return fail(c, TR_ERR_SIZE, "tr_read_rgba8_static: frame larger than the static staging buffer (3840 x 2160 pixels)");
    if (c->mode != 0) {  // tile / row raster: a pending bin-list growth (which allocates) belongs to the other read calls
        uint32_t fl = 0;
        TR_HIP(c, hipMemcpyAsync(&fl, c->flags, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
        TR_HIP(c, hipStreamSynchronize(c->stream));
        if (fl & (1u << 6))
            return fail(c, TR_ERR_UNSUPPORTED, "tr_read_rgba8_static: bin list overflow pending; tr_get_status_flags settles it (and may allocate)");
    }
    const uint32_t* src = c->rgba8 + (size_t)c->sel_view * npix;
    struct Busy {  // (released on every way out)
        std::atomic_flag* f = nullptr;
        ~Busy() { if (f) f->clear(std::memory_order_release); }
    } busy;
    if (flip) {
        uint32_t* stage = static_stage(c->device);
        if (!stage) return fail(c, TR_ERR_UNSUPPORTED, "tr_read_rgba8_static: no staging array on this device");
        if (g_static_stage_busy[c->device].test_and_set(std::memory_order_acquire))
            return fail(c, TR_ERR_BUSY, "tr_read_rgba8_static: another call is using this GPU's staging array (all contexts of a GPU share it); call again");
        busy.f = &g_static_stage_busy[c->device];
        // every one of the W*H words copied below is written here first: the array's old contents
        // (the last call's image, of any size) never reach the caller
        hipLaunchKernelGGL(k_flip_rows, dim3(2048), dim3(256), 0, c->stream, src, stage, c->W, c->H);
        TR_HIP(c, hipGetLastError());
        src = stage;
    }
    TR_HIP(c, hipMemcpyAsync(out, src, npix * 4, hipMemcpyDeviceToHost, c->stream));
    TR_HIP(c, hipStreamSynchronize(c->stream));
    return TR_OK;
}

void* tr_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void tr_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int tr_read_rgba_f64(tr_context* c, double* out, int layout) {
    if (!c || !out) return TR_ERR_INVALID;
    int rc = need_plane(c, TR_OUT_RGBA_F64);
    if (rc != TR_OK) return rc;
    return read_plane<double>(c, c->rgba_f64, out, layout, 4);
}

int tr_get_counters(tr_context* c, uint64_t out[8]) {
    if (!c || !out) return TR_ERR_INVALID;
    TR_HIP(c, hipSetDevice(c->device));
    {
        int rc = settle_bin_overflow(c);
        if (rc != TR_OK) return rc;
    }
    constexpr size_t kWords = (size_t)kCounterSlots * 8;
    std::unique_ptr<unsigned long long[]> both(new (std::nothrow) unsigned long long[2 * kWords]);
    if (!both) return fail(c, TR_ERR_OOM, "host buffer for the counters");
    unsigned long long* tmp = both.get();
    unsigned long long* cov = tmp + kWords;
    TR_HIP(c, hipMemcpyAsync(tmp, c->counters, kWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    TR_HIP(c, hipMemcpyAsync(cov, c->cov_counters, kWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    TR_HIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (uint32_t s = 0; s < kCounterSlots; ++s) {
        out[0] += tmp[s * 8];
        out[1] += tmp[s * 8 + 1];
        out[2] += tmp[s * 8 + 2];
        out[5] += tmp[s * 8 + 5];
        out[6] += tmp[s * 8 + 6];
        out[7] += tmp[s * 8 + 7];
        out[4] += cov[s * 8];
    }
    if (!c->resolved) out[4] = 0;
    if (c->mode == 1) out[2] = 0;  // scanlines / samples are not counted by the tile raster
    out[3] = c->tri_total;
    if (env_diag("TR_TILE_DEBUG") && std::atoi(env_diag("TR_TILE_DEBUG")) == 8) {
        out[2] = tmp[2];
        out[3] = tmp[3];
    }
    return TR_OK;
}

int tr_get_status_flags(tr_context* c, uint32_t* out_flags) {
    if (!c || !out_flags) return TR_ERR_INVALID;
    TR_HIP(c, hipSetDevice(c->device));
    {
        int rc = settle_bin_overflow(c);
        if (rc != TR_OK) return rc;
    }
    TR_HIP(c, hipMemcpyAsync(out_flags, c->flags, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    TR_HIP(c, hipStreamSynchronize(c->stream));
    return TR_OK;
}

int tr_synchronize(tr_context* c) {
    if (!c) return TR_ERR_INVALID;
    TR_HIP(c, hipSetDevice(c->device));
    TR_HIP(c, hipStreamSynchronize(c->stream));
    return TR_OK;
}

int tr_get_stream(tr_context* c, void** out_hip_stream) {
    if (!c || !out_hip_stream) return TR_ERR_INVALID;
    *out_hip_stream = (void*)c->stream;
    return TR_OK;
}

int tr_device_ptr_rgba8(tr_context* c, void** out_dev_ptr) {
    if (!c || !out_dev_ptr) return TR_ERR_INVALID;
    *out_dev_ptr = (void*)(c->rgba8 + (size_t)c->sel_view * c->W * c->H);
    return TR_OK;
}

int tr_set_band(tr_context* c, int y_lo, int y_hi) {
    if (!c) return TR_ERR_INVALID;
    if (y_lo < 0 || y_hi > c->H || y_lo >= y_hi || (y_lo % kTile) != 0 || (y_hi != c->H && (y_hi % kTile) != 0))
        return fail(c, TR_ERR_INVALID, "band must be whole 16-pixel tile rows inside the frame");
    if (c->tri_total) return fail(c, TR_ERR_INVALID, "tr_set_band must follow tr_clear, before any draw");
    if (c->max_views > 1 && (y_lo != 0 || y_hi != c->H)) return fail(c, TR_ERR_UNSUPPORTED, "no bands on a tr_create_views context");
    c->y_lo = y_lo;
    c->y_hi = y_hi;
    return TR_OK;
}

int tr_get_shade_launch(tr_context* c, uint32_t out[4]) {
    if (!c || !out) return TR_ERR_INVALID;
    out[0] = c->last_shade_grid[0]; out[1] = c->last_shade_grid[1]; out[2] = c->last_shade_grid[2];
    out[3] = (uint32_t)(kShadeBW * kShadeBH);
    return TR_OK;
}

int tr_set_colour_exact(tr_context* c, int exact) {
    if (!c) return TR_ERR_INVALID;
    c->colour_exact = exact != 0;  // (from the next resolve on: what is resolved stays as it was shaded)
    return TR_OK;
}

int tr_set_raster_mode(tr_context* c, int mode) {
    if (!c || mode < 0 || mode > 2) return TR_ERR_INVALID;
    if (c->max_views > 1 && mode != 0) return fail(c, TR_ERR_UNSUPPORTED, "a tr_create_views context uses the global raster");
    if (c->tri_total) return fail(c, TR_ERR_INVALID, "tr_set_raster_mode must follow tr_clear, before any draw");
    if (mode != c->mode) {
        c->mode = mode;
        c->planes_clean = false;  // the modes leave different things in the planes
        return tr_clear(c);
    }
    return TR_OK;
}

int tr_selftest_arithmetic(int device, uint64_t n_values, uint64_t seed, uint64_t out[4]) {
    if (!out) return TR_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return TR_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return TR_ERR_NO_DEVICE;
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 4 * sizeof(unsigned long long)) != hipSuccess) return TR_ERR_OOM;
    int rc = TR_OK;
    const int blocks = 2048, threads = 256;
    uint64_t per = n_values / ((uint64_t)blocks * threads);
    if (per < 1) per = 1;
    if (per > 0xFFFFFFFFull) per = 0xFFFFFFFFull;
    if (hipMemset(d, 0, 4 * sizeof(unsigned long long)) != hipSuccess) rc = TR_ERR_HIP;
    if (rc == TR_OK) {
        hipLaunchKernelGGL(k_selftest, dim3(blocks), dim3(threads), 0, nullptr, (unsigned long long)(seed ? seed : 1), (uint32_t)per, d);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = TR_ERR_HIP;
    }
    unsigned long long h[4] = {0, 0, 0, 0};
    if (rc == TR_OK && hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) rc = TR_ERR_HIP;
    (void)hipFree(d);
    for (int i = 0; i < 4; ++i) out[i] = h[i];
    return rc;
}

int tr_profile_enable(tr_context* c, int enable) {
    if (!c) return TR_ERR_INVALID;
    c->prof = enable != 0;
    return TR_OK;
}

int tr_profile_read(tr_context* c, double ms[6], uint64_t counts[6], int reset) {
    if (!c || !ms || !counts) return TR_ERR_INVALID;
    for (int i = 0; i < ST_COUNT; ++i) {
        ms[i] = c->prof_ms[i];
        counts[i] = c->prof_n[i];
        if (reset) {
            c->prof_ms[i] = 0;
            c->prof_n[i] = 0;
        }
    }
    return TR_OK;
}

}  // extern "C"
