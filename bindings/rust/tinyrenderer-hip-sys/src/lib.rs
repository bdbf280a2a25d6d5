//! Raw binding of `include/tinyrenderer_hip.h`: one `extern "C"` item per export, same order as the header.
//! Every call returns 0 or a negative `TR_ERR_*`; nothing unwinds across the boundary.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

macro_rules! opaque {
    ($($name:ident),*) => { $(#[repr(C)] pub struct $name { _private: [u8; 0] })* };
}
opaque!(tr_context, tr_mesh, tr_texture, tr_shadow_map, tr_obj, tr_image);

pub const TR_OK: c_int = 0;
pub const TR_ERR_INVALID: c_int = -1;
pub const TR_ERR_HIP: c_int = -2;
pub const TR_ERR_NO_DEVICE: c_int = -3;
pub const TR_ERR_OOM: c_int = -4;
pub const TR_ERR_UNSUPPORTED: c_int = -5;
pub const TR_ERR_SIZE: c_int = -6;
pub const TR_ERR_BUSY: c_int = -7;

pub const TR_FLAG_TRI_TOO_LARGE: u32 = 1 << 0;
pub const TR_FLAG_REF_PANIC_SPLIT: u32 = 1 << 1;
pub const TR_FLAG_TEX_OOB: u32 = 1 << 2;
pub const TR_FLAG_NEG_INF_DEPTH: u32 = 1 << 3;
pub const TR_FLAG_SCANLINE_OVERFLOW: u32 = 1 << 4;
pub const TR_FLAG_NAN_PIXEL_OVERFLOW: u32 = 1 << 5;
pub const TR_FLAG_BIN_OVERFLOW: u32 = 1 << 6;
pub const TR_FLAG_PREFIX_STALL: u32 = 1 << 7;

pub const TR_RASTER_GLOBAL: c_int = 0;
pub const TR_RASTER_TILE: c_int = 1;
pub const TR_RASTER_ROWS: c_int = 2;

pub const TR_LAYOUT_ROW_MAJOR: c_int = 0;
/// element (x, y) at x * H + y: ndarray's `Array3<f64>` of shape (W, H, C), src/helpers/mod.rs:391-427
pub const TR_LAYOUT_REFERENCE: c_int = 1;

pub const TR_OUT_RGBA8: u32 = 1;
pub const TR_OUT_Z: u32 = 2;
pub const TR_OUT_TRI_ID: u32 = 4;
pub const TR_OUT_RGBA_F64: u32 = 8;
pub const TR_OUT_KEY: u32 = 16;

pub const TR_SHADE_REFERENCE: i32 = 0;
pub const TR_SHADE_GOURAUD: i32 = 1;
pub const TR_SHADE_PHONG: i32 = 2;

/// The shader pair of src/main.rs:146-267 as data.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct tr_draw_params {
    /// perspective * model_view, column-major (`Matrix4::as_slice()`), src/main.rs:152-153
    pub pmv: [f64; 16],
    /// model_view.transpose().try_inverse(), src/main.rs:129-135
    pub normal_mat: [f64; 16],
    /// camera_pos, src/main.rs:62-64 (normalised by the shader, :249)
    pub light_dir: [f64; 3],
    pub ambient: f64,
    pub intensity: f64,
    /// WHITE, src/main.rs:170
    pub vertex_color: [f64; 4],
    pub use_normal_map: i32,
    pub reserved: i32,
    pub diffuse: *const tr_texture,
    pub normal_map: *const tr_texture,
}

/// EXTENSION shaders (no counterpart in the reference).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct tr_shade_ext {
    pub model: i32,
    pub reserved: i32,
    pub specular_map: *const tr_texture,
    pub shininess: f64,
    pub specular_strength: f64,
    pub shadow: *const tr_shadow_map,
    /// light_viewport * light_pmv * (viewport * pmv)^-1, column-major
    pub shadow_from_view: [f64; 16],
    pub shadow_bias: f64,
    pub shadow_dim: f64,
}

extern "C" {
    // ModelRendererState::new / clear_screen, src/model_renderer/mod.rs:151-172, 210-219
    pub fn tr_create(device: c_int, width: c_int, height: c_int, out_ctx: *mut *mut tr_context) -> c_int;
    pub fn tr_destroy(ctx: *mut tr_context);
    pub fn tr_last_error(ctx: *const tr_context) -> *const c_char;
    pub fn tr_clear(ctx: *mut tr_context) -> c_int;
    // the wavefront_obj Object consumed at mod.rs:237-276; MyRgbaImage after flip_vertically, mod.rs:40-41
    pub fn tr_mesh_create(ctx: *mut tr_context, positions: *const f64, n_positions: i64, normals: *const f64, n_normals: i64,
        uvw: *const f64, n_uvw: i64, idx: *const u32, n_triangles: i64, out_mesh: *mut *mut tr_mesh) -> c_int;
    pub fn tr_mesh_destroy(ctx: *mut tr_context, mesh: *mut tr_mesh) -> c_int;
    pub fn tr_texture_create(ctx: *mut tr_context, rgba8: *const u8, width: c_int, height: c_int, out_tex: *mut *mut tr_texture) -> c_int;
    pub fn tr_texture_destroy(ctx: *mut tr_context, tex: *mut tr_texture) -> c_int;
    // render_mesh_component, mod.rs:225-475
    pub fn tr_draw(ctx: *mut tr_context, mesh: *const tr_mesh, params: *const tr_draw_params) -> c_int;
    // extension shaders
    pub fn tr_shadow_map_create(light_ctx: *mut tr_context, out_map: *mut *mut tr_shadow_map) -> c_int;
    pub fn tr_shadow_map_update(map: *mut tr_shadow_map, light_ctx: *mut tr_context) -> c_int;
    pub fn tr_shadow_map_destroy(map: *mut tr_shadow_map);
    pub fn tr_draw_ext(ctx: *mut tr_context, mesh: *const tr_mesh, params: *const tr_draw_params, ext: *const tr_shade_ext) -> c_int;
    pub fn tr_resolve(ctx: *mut tr_context, out_mask: u32) -> c_int;
    // state.z_buffer.nd_img / state.frame_buffer.nd_img (mod.rs:30-32), ndarray_to_image_rgba_and_flip (helpers/mod.rs:204-222)
    pub fn tr_read_z_f64(ctx: *mut tr_context, out: *mut f64, layout: c_int) -> c_int;
    pub fn tr_read_triangle_id_u32(ctx: *mut tr_context, out: *mut u32, layout: c_int) -> c_int;
    pub fn tr_read_winner_key_u64(ctx: *mut tr_context, out: *mut u64, layout: c_int) -> c_int;
    pub fn tr_read_rgba8(ctx: *mut tr_context, out: *mut u8, flip: c_int) -> c_int;
    pub fn tr_read_rgba_f64(ctx: *mut tr_context, out: *mut f64, layout: c_int) -> c_int;
    pub fn tr_read_rgba8_static(ctx: *mut tr_context, out: *mut u8, flip: c_int) -> c_int;
    pub fn tr_host_alloc(bytes: usize) -> *mut c_void;
    pub fn tr_host_free(p: *mut c_void);
    pub fn tr_get_counters(ctx: *mut tr_context, out: *mut u64) -> c_int;
    pub fn tr_get_status_flags(ctx: *mut tr_context, out_flags: *mut u32) -> c_int;
    // batched views: the loop of src/main.rs:57-116,272-273, n_views iterations in one call
    pub fn tr_create_views(device: c_int, width: c_int, height: c_int, max_views: c_int, out_ctx: *mut *mut tr_context) -> c_int;
    pub fn tr_render_views(ctx: *mut tr_context, mesh: *const tr_mesh, params: *const tr_draw_params, n_views: c_int, out_mask: u32) -> c_int;
    pub fn tr_select_view(ctx: *mut tr_context, view: c_int) -> c_int;
    // the loop's own scene, several meshes per iteration (main.rs:272-273): params[v * n_meshes + m]
    pub fn tr_render_views_multi(ctx: *mut tr_context, n_meshes: c_int, meshes: *const *const tr_mesh, params: *const tr_draw_params, n_views: c_int, out_mask: u32) -> c_int;
    // multi-GPU bands, strategy, plumbing
    pub fn tr_set_band(ctx: *mut tr_context, y_lo: c_int, y_hi: c_int) -> c_int;
    pub fn tr_set_raster_mode(ctx: *mut tr_context, mode: c_int) -> c_int;
    // colour arithmetic of the fragment closure (main.rs:173-267): fused (+-1 LSB, default) or operation for operation (0 LSB)
    pub fn tr_set_colour_exact(ctx: *mut tr_context, exact: c_int) -> c_int;
    pub fn tr_synchronize(ctx: *mut tr_context) -> c_int;
    pub fn tr_get_stream(ctx: *mut tr_context, out_hip_stream: *mut *mut c_void) -> c_int;
    pub fn tr_device_ptr_rgba8(ctx: *mut tr_context, out_dev_ptr: *mut *mut c_void) -> c_int;
    pub fn tr_profile_enable(ctx: *mut tr_context, enable: c_int) -> c_int;
    pub fn tr_get_shade_launch(ctx: *mut tr_context, out: *mut u32) -> c_int;
    pub fn tr_profile_read(ctx: *mut tr_context, ms: *mut f64, counts: *mut u64, reset: c_int) -> c_int;
    // loading surface: helpers::load_model (helpers/mod.rs:352-355), load_rgba_img_from_file + flip_vertically (:93-96, 291-302)
    pub fn tr_obj_load(path: *const c_char, out: *mut *mut tr_obj) -> c_int;
    pub fn tr_obj_parse(text: *const c_char, len: i64, out: *mut *mut tr_obj) -> c_int;
    pub fn tr_obj_counts(obj: *const tr_obj, n_positions: *mut i64, n_uvw: *mut i64, n_normals: *mut i64, n_triangles: *mut i64) -> c_int;
    pub fn tr_obj_positions(obj: *const tr_obj) -> *const f64;
    pub fn tr_obj_uvw(obj: *const tr_obj) -> *const f64;
    pub fn tr_obj_normals(obj: *const tr_obj) -> *const f64;
    pub fn tr_obj_indices(obj: *const tr_obj) -> *const u32;
    pub fn tr_obj_free(obj: *mut tr_obj);
    pub fn tr_image_load_rgba8(path: *const c_char, flip_vertically: c_int, out: *mut *mut tr_image) -> c_int;
    pub fn tr_image_size(img: *const tr_image, width: *mut c_int, height: *mut c_int) -> c_int;
    pub fn tr_image_data(img: *const tr_image) -> *const u8;
    pub fn tr_image_free(img: *mut tr_image);
    pub fn tr_image_write_png(path: *const c_char, rgba8: *const u8, width: c_int, height: c_int) -> c_int;
    pub fn tr_load_error() -> *const c_char;
    pub fn tr_selftest_arithmetic(device: c_int, n_values: u64, seed: u64, out: *mut u64) -> c_int;
    pub fn tr_version() -> *const c_char;
}
