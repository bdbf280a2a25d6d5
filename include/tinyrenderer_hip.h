/*
 * tinyrenderer_hip.h -- C ABI of the MI355X (gfx950) rasterizer that stands
 * behind Davidster/tinyrenderer's `model_renderer` draw path.
 *
 * The reference has no FFI of its own: its boundary is the Rust function
 * `render_mesh_component(&mut ModelRendererState, &Object, &mut dyn Fn(VertexShaderArgs)->..,
 * &mut dyn Fn(FragmentShaderArgs)->..)` (src/model_renderer/mod.rs:225-230) plus
 * `ModelRendererState::new` (:151), `clear_screen` (:210) and the public
 * `frame_buffer` / `z_buffer` fields (:30-32).  Host closures cannot run on a
 * GPU, so the two closures are replaced by a descriptor of the reference's only
 * shader pair (src/main.rs:146-267).  Every entry point below names the
 * reference interface it replaces.
 *
 * Conventions: plain pointers and sizes only; every call returns 0 on success
 * or a negative TR_ERR_* code and never throws/aborts across the ABI;
 * `tr_last_error` gives text for the last failure on that context.  One
 * context per GPU per thread of use (not thread-safe).  The caller keeps
 * ownership of all host pointers; the library copies.  All matrices are
 * column-major f64 (nalgebra's storage order).
 */
#ifndef TINYRENDERER_HIP_H
#define TINYRENDERER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tr_context tr_context;
typedef struct tr_mesh tr_mesh;
typedef struct tr_texture tr_texture;

enum {
    TR_OK = 0,
    TR_ERR_INVALID = -1,      /* bad argument */
    TR_ERR_HIP = -2,          /* a HIP runtime call failed */
    TR_ERR_NO_DEVICE = -3,    /* no usable GPU / kernels not loadable */
    TR_ERR_OOM = -4,
    TR_ERR_UNSUPPORTED = -5,  /* scene hit a documented limit, see tr_get_status_flags */
    TR_ERR_SIZE = -6,         /* size beyond a documented maximum */
    TR_ERR_BUSY = -7          /* a shared resource is in use by another call (tr_read_rgba8_static); call again */
};

/* Bits of tr_get_status_flags(): conditions met on the device since tr_clear. */
enum {
    TR_FLAG_TRI_TOO_LARGE = 1u << 0, /* a triangle's sample grid exceeds 131070 scanlines: skipped
                                        (the reference has no clipping and would loop ~forever) */
    TR_FLAG_REF_PANIC_SPLIT = 1u << 1, /* split parameter outside [0,1]: the reference panics
                                          (model_renderer/mod.rs:441); triangle skipped */
    TR_FLAG_TEX_OOB = 1u << 2,       /* texture index out of range: the reference panics
                                        (helpers/mod.rs:262); index clamped */
    TR_FLAG_NEG_INF_DEPTH = 1u << 3, /* a sample with dist == -inf (reference re-opens the pixel;
                                        not reproduced) */
    TR_FLAG_SCANLINE_OVERFLOW = 1u << 4, /* more than 2^32-1 scanlines in one frame */
    TR_FLAG_NAN_PIXEL_OVERFLOW = 1u << 5, /* more NaN-depth pixels than the side table holds
                                             (global-atomic raster mode only) */
    TR_FLAG_BIN_OVERFLOW = 1u << 6, /* transient (tile / row raster): a bin list was too small; tr_resolve
                                       itself grows it, redoes the resolve and clears the bit before it
                                       returns (one host wait per resolve in those two modes) */
    TR_FLAG_PREFIX_STALL = 1u << 7  /* internal error, never seen in practice: a workgroup of the fused set-up launch
                                       gave up waiting for the scanline prefix of the workgroups before it (a bounded
                                       poll instead of a hang); the frame is incomplete */
};

/* Raster strategies (same results, bit for bit). */
enum {
    TR_RASTER_GLOBAL = 0, /* default: samples enumerated frame-wide, 64-bit atomicMin on HBM planes */
    TR_RASTER_TILE = 1,   /* triangles binned to 16x16 tiles, depth/winner tiles in LDS;
                             env TR_RASTER_MODE=tile selects it at tr_create */
    TR_RASTER_ROWS = 2    /* scanlines binned by (pixel row, 256-pixel segment), one wave per bin with
                             its planes in LDS, no global atomics; env TR_RASTER_MODE=rows */
};

/* Buffer layouts for the tr_read_* calls. */
enum {
    TR_LAYOUT_ROW_MAJOR = 0, /* element (x,y) at y*W + x        (this library's HBM layout)      */
    TR_LAYOUT_REFERENCE = 1  /* element (x,y) at x*H + y        (ndarray Array3 (W,H,C) layout,
                                src/helpers/mod.rs:391-427)                                     */
};

/* Replaces ModelRendererState::new(w,h), src/model_renderer/mod.rs:151-172.
 * Allocates the depth / winner / colour planes on `device` and clears them. */
int tr_create(int device, int width, int height, tr_context** out_ctx);
void tr_destroy(tr_context* ctx);
const char* tr_last_error(const tr_context* ctx);

/* Replaces clear_screen, src/model_renderer/mod.rs:210-219: colour (0,0,0,255),
 * z = -inf, no winner; also forgets the draws recorded since the last clear. */
int tr_clear(tr_context* ctx);

/* Replaces the wavefront_obj::obj::Object consumed at src/model_renderer/mod.rs:237-276
 * (geometry[0], Primitive::Triangle only).  idx holds, per triangle, 3 corners x
 * (vertex, texture, normal) 0-based indices; 0xFFFFFFFF = None (texture / normal only). */
int tr_mesh_create(tr_context* ctx, const double* positions, int64_t n_positions,
                   const double* normals, int64_t n_normals, const double* uvw, int64_t n_uvw,
                   const uint32_t* idx, int64_t n_triangles, tr_mesh** out_mesh);
/* Lifetime: a draw of 65 536 triangles or more builds its per-triangle shade records at resolve time (only for the
 * triangles that own a pixel) and reads the mesh again then: destroying such a mesh before the next tr_clear leaves
 * what is resolved readable and makes a further resolve of the frame fail with TR_ERR_INVALID. */
int tr_mesh_destroy(tr_context* ctx, tr_mesh* mesh);

/* Replaces MyRgbaImage after flip_vertically (src/model_renderer/mod.rs:40-41,
 * src/helpers/mod.rs:161-173,291-302): rgba8 is row-major, 4 bytes per texel, and ALREADY
 * flipped, i.e. row 0 is the bottom row of the decoded image (texel (x,y) = nd_img[[x,y,:]]). */
int tr_texture_create(tr_context* ctx, const uint8_t* rgba8, int width, int height,
                      tr_texture** out_tex);
/* Lifetime: the draws recorded since tr_clear refer to their textures until the next tr_clear.  Destroying one
 * through the context that drew with it is safe: what that frame has resolved stays readable, a further resolve
 * of the frame (more draws, or a plane not asked for before) fails with TR_ERR_INVALID.  A texture drawn with on
 * ANOTHER context of the same device must outlive that context's frame (tr_clear or tr_destroy first). */
int tr_texture_destroy(tr_context* ctx, tr_texture* tex);

/* The reference's shader pair (src/main.rs:146-267) as data. */
typedef struct tr_draw_params {
    double pmv[16];        /* perspective * model_view, src/main.rs:152-153                  */
    double normal_mat[16]; /* (model_view^T)^-1, src/main.rs:129-135,164                      */
    double light_dir[3];   /* camera_pos of src/main.rs:62-64; normalised by the shader :249  */
    double ambient;        /* src/main.rs:255 (1.0)                                           */
    double intensity;      /* src/main.rs:254 (1.0)                                           */
    double vertex_color[4];/* src/main.rs:170 (WHITE = 255,255,255,255)                       */
    int32_t use_normal_map;/* src/main.rs:182 (true)                                          */
    int32_t reserved;
    const tr_texture* diffuse;    /* mesh_component.texture,    src/main.rs:141 */
    const tr_texture* normal_map; /* mesh_component.normal_map, src/main.rs:140 */
} tr_draw_params;

/* Replaces render_mesh_component(state, mesh, VS, FS), src/model_renderer/mod.rs:225-475,
 * for the shader pair above: vertex transform, triangle setup, depth pass.  Winner selection and
 * shading are deferred to tr_resolve (the reference shades eagerly; the result is the same
 * because colour = shade(first sample in draw order whose depth equals the final minimum)). */
int tr_draw(tr_context* ctx, const tr_mesh* mesh, const tr_draw_params* params);

/* ---- EXTENSION shaders: no counterpart in the reference, whose only shader pair is the one above.
 * They exist because the build's brief (BASELINE.json north_star, configs[1]) names them; they are
 * defined by this library and its CPU restatement (oracle/, `ora_render_mesh_component_ext`) and are
 * checked GPU-against-CPU only.  With `ext` NULL, or model TR_SHADE_REFERENCE and no shadow map,
 * tr_draw_ext IS tr_draw.  For a pixel with unit normal n (normal-mapped as in the reference), light
 * direction l, albedo a:
 *   diff  = max(n.l, 0)                       (Gouraud: max(n_c.l, 0) of the three unit vertex normals,
 *                                              interpolated with the pixel's barycentrics)
 *   spec  = pow(max((2 (n.l) n - l).z, 0), e)  (Phong only; e = channel 0 of the specular map at the
 *                                              pixel's uv, sampled like the other textures, else `shininess`)
 *   lit   = shadow_dim if the pixel's position, taken to the light frame by `shadow_from_view`
 *           (column-major; perspective-divided; rounded to a pixel like every sample), lies on a
 *           non-empty pixel of the shadow map whose z is smaller than the pixel's by more than
 *           `shadow_bias`; else 1
 *   rgb_c = min(ambient + (lit * (diff * intensity + specular_strength * spec)) * a_c, 255)
 * tr_shadow_map_create snapshots the z plane of a context in which the scene was drawn from the
 * light (any size); the map is independent of that context afterwards and may serve any number of
 * frames on the same device.  A light frame that has not been resolved is resolved for depth only
 * (no shading).  tr_shadow_map_update takes a new snapshot into an existing map of the same size (a
 * light that moves every frame: no allocation).  Neither call waits on the host: the copy runs on the
 * light context's stream, a context whose frame uses the map makes its shade pass wait for it, and an
 * update waits for the shade passes that still read the old contents.  tr_shadow_map_destroy must not
 * be called before every context that drew with the map has been cleared (tr_clear) or destroyed: until
 * then a read of another plane may resolve that frame again. */
enum { TR_SHADE_REFERENCE = 0, TR_SHADE_GOURAUD = 1, TR_SHADE_PHONG = 2 };
typedef struct tr_shadow_map tr_shadow_map;
typedef struct tr_shade_ext {
    int32_t model;                  /* TR_SHADE_* */
    int32_t reserved;
    const tr_texture* specular_map; /* Phong exponent map or NULL */
    double shininess;
    double specular_strength;
    const tr_shadow_map* shadow;    /* NULL = no shadow test */
    double shadow_from_view[16];    /* light_viewport * light_pmv * inverse(viewport * pmv) of this draw */
    double shadow_bias;
    double shadow_dim;
} tr_shade_ext;
int tr_shadow_map_create(tr_context* light_ctx, tr_shadow_map** out_map);
int tr_shadow_map_update(tr_shadow_map* map, tr_context* light_ctx);
void tr_shadow_map_destroy(tr_shadow_map* map);
int tr_draw_ext(tr_context* ctx, const tr_mesh* mesh, const tr_draw_params* params, const tr_shade_ext* ext);

/* No reference analogue (the reference shades inside the z-test): selects the winning sample per
 * pixel over all draws since tr_clear and runs the deferred shade.  out_mask selects which
 * planes are written besides RGBA8 (TR_OUT_*).  With TR_OUT_RGBA8 alone (the fast path) the
 * global raster's depth / winner planes are wiped as they are consumed, which makes the next
 * tr_clear free; asking for another plane later re-runs the raster passes from the recorded
 * draws. */
enum { TR_OUT_RGBA8 = 1, TR_OUT_Z = 2, TR_OUT_TRI_ID = 4, TR_OUT_RGBA_F64 = 8, TR_OUT_KEY = 16 };
int tr_resolve(tr_context* ctx, uint32_t out_mask);

/* Replace reads of state.z_buffer.nd_img / state.frame_buffer.nd_img
 * (src/model_renderer/mod.rs:30-32) and ndarray_to_image_rgba_and_flip
 * (src/helpers/mod.rs:204-222).  All synchronise the context's stream.
 *   z:        f64 per pixel, -inf where nothing was drawn.
 *   tri id:   u32 per pixel = index of the winning triangle counted over all draws since
 *             tr_clear (draw order, then mesh order); 0xFFFFFFFF where nothing was drawn.
 *   key:      u64 per pixel = tri_seq<<36 | half<<35 | scanline i<<18 | sample j; ~0 = none.
 *   rgba8:    `val as u8` of the f64 colour; flip != 0 writes row H-1-y (the presented image).
 *   rgba f64: 4 f64 per pixel. */
int tr_read_z_f64(tr_context* ctx, double* out, int layout);
int tr_read_triangle_id_u32(tr_context* ctx, uint32_t* out, int layout);
int tr_read_winner_key_u64(tr_context* ctx, uint64_t* out, int layout);
int tr_read_rgba8(tr_context* ctx, uint8_t* out, int flip);
int tr_read_rgba_f64(tr_context* ctx, double* out, int layout);
/* No counterpart in the reference.  tr_read_rgba8 for callers that CANNOT allocate device memory (serving loops,
 * harnesses that forbid hipMalloc, code that must not allocate between the captures or replays of a HIP graph; the
 * call itself synchronises the context's stream, so it is not for use INSIDE a capture): it never calls hipMalloc /
 * hipFree, resolves nothing (the frame must have been resolved with TR_OUT_RGBA8: TR_ERR_INVALID otherwise) and
 * stages the flipped image in a static device array of 3840 x 2160 pixels (33 177 600 B per GPU, held until the
 * process ends) instead of a buffer allocated on first use; larger frames: TR_ERR_SIZE; a GPU on which the array's
 * address could not be looked up: TR_ERR_UNSUPPORTED.  Pass `out` from tr_host_alloc for a copy the runtime does not
 * stage either.  All contexts of a GPU share that array and it keeps the last call's contents (every word handed to
 * the caller is rewritten first); one call at a time per GPU uses it: a call that arrives while another context's
 * call on the same GPU is in progress returns TR_ERR_BUSY and may simply be made again. */
int tr_read_rgba8_static(tr_context* ctx, uint8_t* out, int flip);
/* Page-locked host memory for the read calls above (no counterpart in the reference): a device-to-host
 * copy into it runs at the link's rate instead of being staged (4K RGBA8: 0.7 ms instead of ~3 ms).
 * tr_host_alloc returns NULL on failure (or without a GPU). */
void* tr_host_alloc(size_t bytes);
void tr_host_free(void* p);

/* Counters since tr_clear: [1] in-bounds z-tested samples ("fragments",
 * src/model_renderer/mod.rs:380-389; valid after tr_resolve in tile mode), [3] triangles
 * submitted, [4] pixels with a winner (valid after tr_resolve); [0] samples enumerated and
 * [2] scanlines are not counted by TR_RASTER_TILE (0 there); [5..7] reserved. */
int tr_get_counters(tr_context* ctx, uint64_t out[8]);
int tr_get_status_flags(tr_context* ctx, uint32_t* out_flags);

/* Batched views.  The reference renders one view per iteration of its loop -- clear_screen, camera, then
 * render_mesh_component per mesh (src/main.rs:57-116,272-273).  tr_render_views is n_views such iterations over
 * ONE mesh in one call: params[v] is what tr_draw takes for view v (its pmv, normal matrix, camera position as the
 * light, textures).  It begins a new frame (an implicit tr_clear), renders every view into its own W x H slice of
 * the planes -- one setup launch over (view, triangle), one depth pass, one winner pass, one shade launch for the
 * whole batch -- and resolves with out_mask (TR_OUT_*).  Per view the results are those of tr_clear + tr_draw +
 * tr_resolve on a W x H context, except that a winner key's / triangle id's tri_seq counts over the batch:
 * view v's triangle k is v * n_triangles + k.  tr_select_view picks the slice the tr_read_* calls,
 * tr_read_rgba8_static and tr_device_ptr_rgba8 refer to (view 0 after tr_create_views); the slices are contiguous
 * in view order, so tr_device_ptr_rgba8 of view 0 addresses all n_views images.  tr_get_counters sums over the
 * batch.  A context made by tr_create_views (max_views in [2, 4096], W * H * max_views < 2^31) renders through
 * tr_render_views / tr_render_views_multi only (tr_draw: TR_ERR_INVALID) and uses the global raster; no bands, no extension shaders. */
int tr_create_views(int device, int width, int height, int max_views, tr_context** out_ctx);
int tr_render_views(tr_context* ctx, const tr_mesh* mesh, const tr_draw_params* params, int n_views, uint32_t out_mask);
int tr_select_view(tr_context* ctx, int view);
/* The same for the reference's own scene: its loop draws SEVERAL meshes per iteration -- the head and the eye parented
 * to it (src/main.rs:272-273; the parent fold of src/model_renderer/mod.rs:67-84 is in each draw's matrices).
 * meshes[0 .. n_meshes) (1 to 4) are drawn in that order in every view; params[v * n_meshes + m] is what tr_draw takes
 * for mesh m under view v.  Per view the results are those of tr_clear + n_meshes x tr_draw + tr_resolve: tri_seq
 * continues across the meshes as in tr_draw, and over the batch view v's triangle k of mesh m is
 * v * T + (triangles of meshes before m) + k with T the triangles of all meshes.  n_views x n_meshes <= 4096. */
int tr_render_views_multi(tr_context* ctx, int n_meshes, const tr_mesh* const* meshes, const tr_draw_params* params,
                          int n_views, uint32_t out_mask);

/* Multi-GPU band sharding (no reference analogue; exact because a pixel's result is a pure min
 * over the samples landing on it): restrict this context to pixel rows [y_lo, y_hi), whole
 * 16-pixel tile rows.  Call after tr_clear and before the first draw; rows outside the band read
 * back as cleared.  tr_set_raster_mode switches strategy under the same rule.
 * The bands of a frame stack to the whole frame bit for bit, including the one sample that does not land where its
 * triangle lies: a triangle whose corners share one y on the screen has NaN in every sample (mod.rs:435-441) and lands
 * on pixel (0, 0), so it is drawn by the band that owns row 0 and by no other.  One exception: in a draw of >= 65 536
 * triangles whole clusters of 64 are dropped by their boxes before any triangle is looked at; a cluster that holds a
 * triangle of three identical positions (degenerate under every view) is never dropped, but a triangle that only
 * PROJECTS onto one screen row under the frame's matrices, inside a dropped cluster, does not reach pixel (0, 0). */
int tr_set_band(tr_context* ctx, int y_lo, int y_hi);
int tr_set_raster_mode(tr_context* ctx, int mode);

/* Colour arithmetic of the fragment stage (src/main.rs:173-267, src/model_renderer/mod.rs:499-504).  By default
 * everything after the barycentrics and the texture coordinates -- normal interpolation and normalisation, the
 * bilinear sums of helpers::sample_nd_img (src/helpers/mod.rs:263-270), the TBN product, the Lambert term -- may fuse
 * multiply-adds and use the hardware's reciprocal square root (still binary64): the presented RGBA8 is then within
 * +-1 LSB of the reference's, the bar for colour.  exact != 0 (or TR_SHADE_EXACT=1 in the environment at tr_create)
 * evaluates those steps operation for operation as the reference does: 0 LSB, f64 colour bit for bit.  Depth, winner
 * keys, triangle ids and texel addresses are exact either way; a resolve that asks for TR_OUT_RGBA_F64 and frames
 * with extension shaders always use the exact arithmetic.  May be called at any time and holds from the next
 * tr_resolve / tr_render_views on (a frame already resolved keeps the colours it was shaded with; tr_resolve again
 * shades it the other way). */
int tr_set_colour_exact(tr_context* ctx, int exact);

/* Plumbing for callers that keep data on the GPU (no reference analogue). */
int tr_synchronize(tr_context* ctx);
int tr_get_stream(tr_context* ctx, void** out_hip_stream);
int tr_device_ptr_rgba8(tr_context* ctx, void** out_dev_ptr); /* u32[W*H], row-major */
/* Per-stage GPU time in ms accumulated since the last call with reset != 0, measured with
 * hipEvents on the context's stream when profiling is enabled:
 * [0] clear [1] transform [2] setup(+scan) [3] binning (tile mode) / depth pass (global mode)
 * [4] tile raster (tile mode) / winner pass (global mode) [5] shade;
 * counts[i] = number of launches accumulated.  A stage that is ONE kernel launch (the depth pass, the winner pass and
 * the shade pass of a frame of ordinary triangles) is timed by that launch's own start / stop events
 * (hipExtLaunchKernelGGL): the kernel's duration as a kernel trace reports it, without the 2-3 us that lie between an
 * event recorded on the stream and the kernel's first wave. */
int tr_profile_enable(tr_context* ctx, int enable);
/* Shape of the last shade launch: out[0..2] = workgroups in x, y, z (z = views of a batch; all 0: nothing was
 * launched), out[3] = pixels per workgroup.  The RGBA8-only resolve launches the blocks of the frame's screen box
 * only: bench.py prices the bytes of the blocks actually launched with it (roofline.launched_bytes). */
int tr_get_shade_launch(tr_context* ctx, uint32_t out[4]);
int tr_profile_read(tr_context* ctx, double ms[6], uint64_t counts[6], int reset);

/* ---- loading surface (host only; no context or GPU needed) ------------------------------
 * Replace helpers::load_model (src/helpers/mod.rs:352-355: wavefront_obj::obj::parse, first
 * object; only the triangles of geometry[0] are kept because that is all the renderer draws,
 * src/model_renderer/mod.rs:237; polygons are fan-triangulated, indices are 0-based, 0xFFFFFFFF
 * = None) and helpers::load_rgba_img_from_file (+ flip_vertically) (src/helpers/mod.rs:93-96,
 * 291-302: image::open(..).into_rgba8()).  PNG: 8-bit grey / grey+alpha / RGB / RGBA / palette,
 * non-interlaced.  TGA: types 2, 3, 10, 11.  The arrays returned by the accessors have the
 * layout tr_mesh_create / tr_texture_create take and live until the object is freed.
 * tr_load_error() gives text for the calling thread's last loader failure. */
typedef struct tr_obj tr_obj;
typedef struct tr_image tr_image;
int tr_obj_load(const char* path, tr_obj** out);
int tr_obj_parse(const char* text, int64_t len, tr_obj** out);
int tr_obj_counts(const tr_obj* obj, int64_t* n_positions, int64_t* n_uvw, int64_t* n_normals,
                  int64_t* n_triangles);
const double* tr_obj_positions(const tr_obj* obj);
const double* tr_obj_uvw(const tr_obj* obj);
const double* tr_obj_normals(const tr_obj* obj);
const uint32_t* tr_obj_indices(const tr_obj* obj);
void tr_obj_free(tr_obj* obj);
int tr_image_load_rgba8(const char* path, int flip_vertically, tr_image** out);
int tr_image_size(const tr_image* img, int* width, int* height);
const uint8_t* tr_image_data(const tr_image* img);
void tr_image_free(tr_image* img);
/* helpers::_write_nd_rgba_img_to_file (src/helpers/mod.rs:114-120): 8-bit RGBA PNG of rgba8[height][width][4],
 * row 0 = top (what tr_read_rgba8(.., flip = 1) returns). */
int tr_image_write_png(const char* path, const uint8_t* rgba8, int width, int height);
const char* tr_load_error(void);

/* Device self-test of the exact arithmetic shortcuts the kernels use (no counterpart in the reference):
 * runs them against the operations they stand for on about n_values pseudo-random operands.
 * out[0] the hardware's saturating f64 -> u32 conversion vs Rust's `as` rule, out[1] x/255 as multiply + fmas vs `/`, out[2] pixel
 * rounding vs round(): numbers of mismatches (all must be 0); out[3] operands tried per check. */
int tr_selftest_arithmetic(int device, uint64_t n_values, uint64_t seed, uint64_t out[4]);

/* Library identity: returns a static string with the build's target arch and ABI version. */
const char* tr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TINYRENDERER_HIP_H */
