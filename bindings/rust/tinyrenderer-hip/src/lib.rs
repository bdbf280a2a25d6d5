//! Drop-in for the draw path of `src/model_renderer/mod.rs`: `ModelRendererState` (:29-35, 150-173), `clear_screen`
//! (:210-219) and `render_mesh_component` (:225-230) keep their names and call shape; the work happens in
//! `libtinyrenderer_hip.so`.  The reference's two shader closures (`&mut dyn Fn(VertexShaderArgs) -> ..`,
//! `&mut dyn Fn(FragmentShaderArgs) -> ..`) are host code and cannot run on a GPU: `BuiltinShader` carries the values
//! the closures of `src/main.rs:146-267` capture instead.  Errors panic, as every `.unwrap()` of the reference does
//! (`mod.rs:39-41, 441, 447, 452`).
use std::collections::HashMap;
use std::ffi::CStr;

use tinyrenderer_hip_sys as sys;

pub type Matrix4 = nalgebra::Matrix4<f64>;
pub type Point3 = nalgebra::Point3<f64>;
/// `helpers::NDRgbaImage` / `NDGrayImage`: `Array3<f64>` of shape (W, H, C), src/helpers/mod.rs:28-29
pub type NDImage = ndarray::Array3<f64>;

pub const WHITE: [f64; 4] = [255.0, 255.0, 255.0, 255.0]; // helpers/mod.rs:21
const NONE_INDEX: u32 = 0xFFFF_FFFF;

/// The values the vertex and fragment closures of src/main.rs:146-267 capture.
pub struct BuiltinShader<'a> {
    pub perspective_matrix: Matrix4,                 // main.rs:110-116
    pub model_view_matrix: Matrix4,                  // main.rs:126-127
    /// `model_view.transpose().try_inverse()` or identity, main.rs:129-135
    pub normal_matrix: Matrix4,
    pub camera_pos: Point3,                          // main.rs:62-64, the light of :249-252
    /// `mesh_component.texture.nd_img` / `.normal_map.nd_img` (already flipped by `MeshComponent::from`, mod.rs:40-41)
    pub texture: &'a NDImage,
    pub normal_map: &'a NDImage,
    pub use_normal_map: bool,                        // main.rs:182
    pub light_intensity: f64,                        // main.rs:254
    pub ambient_light: f64,                          // main.rs:255
}

pub struct ModelRendererState {
    ctx: *mut sys::tr_context,
    frame_width: usize,
    frame_height: usize,
    // device copies of what the caller's borrowed data describe, made on first use and kept for the state's life;
    // keyed by the address of the borrowed value (meshes and textures are long-lived in the reference's loop)
    meshes: HashMap<usize, *mut sys::tr_mesh>,
    textures: HashMap<usize, *mut sys::tr_texture>,
}

fn error_text(ctx: *const sys::tr_context) -> String {
    unsafe { CStr::from_ptr(sys::tr_last_error(ctx)) }.to_string_lossy().into_owned()
}

fn check(ctx: *const sys::tr_context, rc: i32, what: &str) {
    if rc != sys::TR_OK {
        panic!("{what} failed: {rc} {}", error_text(ctx));
    }
}

fn to16(m: &Matrix4) -> [f64; 16] {
    let mut out = [0.0; 16];
    out.copy_from_slice(m.as_slice()); // nalgebra stores column-major, which is what the ABI takes
    out
}

impl ModelRendererState {
    /// src/model_renderer/mod.rs:151-172: frame buffer BLACK, z buffer -inf.
    pub fn new(frame_width: usize, frame_height: usize) -> ModelRendererState {
        Self::on_device(0, frame_width, frame_height)
    }

    /// One state per GPU (one process per GPU is the intended multi-GPU shape).
    pub fn on_device(device: i32, frame_width: usize, frame_height: usize) -> ModelRendererState {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sys::tr_create(device, frame_width as i32, frame_height as i32, &mut ctx) };
        if rc != sys::TR_OK {
            panic!("tr_create failed: {rc} (a gfx950 GPU is required; there is no CPU fallback)");
        }
        ModelRendererState { ctx, frame_width, frame_height, meshes: HashMap::new(), textures: HashMap::new() }
    }

    pub fn frame_width(&self) -> usize { self.frame_width }
    pub fn frame_height(&self) -> usize { self.frame_height }

    /// `state.frame_buffer.nd_img` (mod.rs:31): `Array3<f64>` (W, H, 4).  Resolves the frame if need be.
    pub fn frame_buffer(&mut self) -> NDImage {
        let mut v = vec![0f64; self.frame_width * self.frame_height * 4];
        let rc = unsafe { sys::tr_read_rgba_f64(self.ctx, v.as_mut_ptr(), sys::TR_LAYOUT_REFERENCE) };
        check(self.ctx, rc, "tr_read_rgba_f64");
        ndarray::Array3::from_shape_vec((self.frame_width, self.frame_height, 4), v).unwrap()
    }

    /// `state.z_buffer.nd_img` (mod.rs:32): `Array3<f64>` (W, H, 1), -inf where nothing was drawn.
    pub fn z_buffer(&mut self) -> NDImage {
        let mut v = vec![0f64; self.frame_width * self.frame_height];
        let rc = unsafe { sys::tr_read_z_f64(self.ctx, v.as_mut_ptr(), sys::TR_LAYOUT_REFERENCE) };
        check(self.ctx, rc, "tr_read_z_f64");
        ndarray::Array3::from_shape_vec((self.frame_width, self.frame_height, 1), v).unwrap()
    }

    /// `ndarray_to_image_rgba_and_flip(&state.frame_buffer)` (helpers/mod.rs:204-222) without the f64 detour: the
    /// `as u8` truncation and the vertical flip happen on the device.  This is what `main.rs:297-302` shows.
    pub fn present(&mut self) -> image::RgbaImage {
        let mut v = vec![0u8; self.frame_width * self.frame_height * 4];
        let rc = unsafe { sys::tr_read_rgba8(self.ctx, v.as_mut_ptr(), 1) };
        check(self.ctx, rc, "tr_read_rgba8");
        image::RgbaImage::from_raw(self.frame_width as u32, self.frame_height as u32, v).unwrap()
    }

    /// Conditions the reference would have panicked or hung on (`TR_FLAG_*`), since the last clear.
    /// Colour arithmetic of the fragment closure (src/main.rs:173-267): `false` (the library's default) lets it fuse
    /// multiply-adds (presented RGBA8 within 1 LSB of this crate's reference), `true` evaluates it operation for
    /// operation as `main.rs` does.  Depth and visibility are exact either way.
    pub fn set_colour_exact(&mut self, exact: bool) {
        let rc = unsafe { sys::tr_set_colour_exact(self.ctx, exact as i32) };
        check(self.ctx, rc, "tr_set_colour_exact");
    }

    pub fn status_flags(&mut self) -> u32 {
        let mut f = 0u32;
        let rc = unsafe { sys::tr_get_status_flags(self.ctx, &mut f) };
        check(self.ctx, rc, "tr_get_status_flags");
        f
    }

    /// The `Object` as `render_mesh_component` consumes it (mod.rs:237-276): geometry[0], `Primitive::Triangle`
    /// only, 0-based (vertex, texture, normal) indices per corner, `None` as 0xFFFFFFFF.
    fn mesh_handle(&mut self, mesh: &wavefront_obj::obj::Object) -> *mut sys::tr_mesh {
        let key = mesh as *const _ as usize;
        if let Some(&h) = self.meshes.get(&key) {
            return h;
        }
        let positions: Vec<f64> = mesh.vertices.iter().flat_map(|v| [v.x, v.y, v.z]).collect();
        let normals: Vec<f64> = mesh.normals.iter().flat_map(|n| [n.x, n.y, n.z]).collect();
        let uvw: Vec<f64> = mesh.tex_vertices.iter().flat_map(|t| [t.u, t.v, t.w]).collect();
        let mut idx: Vec<u32> = Vec::new();
        if let Some(geometry) = mesh.geometry.first() {
            for shape in &geometry.shapes {
                if let wavefront_obj::obj::Primitive::Triangle(a, b, c) = shape.primitive {
                    for (v, t, n) in [a, b, c] {
                        idx.push(v as u32);
                        idx.push(t.map_or(NONE_INDEX, |t| t as u32));
                        idx.push(n.map_or(NONE_INDEX, |n| n as u32));
                    }
                }
            }
        }
        let mut h = std::ptr::null_mut();
        let rc = unsafe {
            sys::tr_mesh_create(self.ctx, positions.as_ptr(), (positions.len() / 3) as i64, normals.as_ptr(),
                (normals.len() / 3) as i64, uvw.as_ptr(), (uvw.len() / 3) as i64, idx.as_ptr(), (idx.len() / 9) as i64, &mut h)
        };
        check(self.ctx, rc, "tr_mesh_create");
        self.meshes.insert(key, h);
        h
    }

    /// `MyRgbaImage.nd_img` (W, H, 4) f64, already flipped -> RGBA8 rows with row 0 = nd y 0
    /// (`out[(y*W + x)*4 + c] = nd_img[[x, y, c]] as u8`: exact, texels are integers 0..=255).
    fn texture_handle(&mut self, nd_img: &NDImage) -> *mut sys::tr_texture {
        let key = nd_img as *const _ as usize;
        if let Some(&h) = self.textures.get(&key) {
            return h;
        }
        let (w, h_img) = (nd_img.shape()[0], nd_img.shape()[1]);
        let mut rgba8 = vec![0u8; w * h_img * 4];
        for y in 0..h_img {
            for x in 0..w {
                for c in 0..4 {
                    rgba8[(y * w + x) * 4 + c] = nd_img[[x, y, c]] as u8;
                }
            }
        }
        let mut h = std::ptr::null_mut();
        let rc = unsafe { sys::tr_texture_create(self.ctx, rgba8.as_ptr(), w as i32, h_img as i32, &mut h) };
        check(self.ctx, rc, "tr_texture_create");
        self.textures.insert(key, h);
        h
    }
}

impl Drop for ModelRendererState {
    fn drop(&mut self) {
        unsafe {
            for (_, m) in self.meshes.drain() {
                sys::tr_mesh_destroy(self.ctx, m);
            }
            for (_, t) in self.textures.drain() {
                sys::tr_texture_destroy(self.ctx, t);
            }
            sys::tr_destroy(self.ctx);
        }
    }
}

/// src/model_renderer/mod.rs:210-219.
pub fn clear_screen(model_renderer_state: &mut ModelRendererState) {
    let rc = unsafe { sys::tr_clear(model_renderer_state.ctx) };
    check(model_renderer_state.ctx, rc, "tr_clear");
}

fn draw_params(state: &mut ModelRendererState, shader: &BuiltinShader) -> sys::tr_draw_params {
    sys::tr_draw_params {
        pmv: to16(&(shader.perspective_matrix * shader.model_view_matrix)), // main.rs:152-153
        normal_mat: to16(&shader.normal_matrix),
        light_dir: [shader.camera_pos.x, shader.camera_pos.y, shader.camera_pos.z],
        ambient: shader.ambient_light,
        intensity: shader.light_intensity,
        vertex_color: WHITE, // main.rs:170
        use_normal_map: shader.use_normal_map as i32,
        reserved: 0,
        diffuse: state.texture_handle(shader.texture),
        normal_map: state.texture_handle(shader.normal_map),
    }
}

/// `render_mesh_component(state, &mesh, VS, FS)` (src/model_renderer/mod.rs:225-230) for the reference's shader
/// pair.  Vertex transform, triangle setup and the depth pass are queued on the GPU; winner selection and shading
/// happen when the frame is read (`present`, `frame_buffer`, `z_buffer`).
pub fn render_mesh_component(model_renderer_state: &mut ModelRendererState, mesh: &wavefront_obj::obj::Object, shader: &BuiltinShader) {
    let m = model_renderer_state.mesh_handle(mesh);
    let p = draw_params(model_renderer_state, shader);
    let rc = unsafe { sys::tr_draw(model_renderer_state.ctx, m, &p) };
    check(model_renderer_state.ctx, rc, "tr_draw");
}

/// A state whose planes hold `max_views` W x H slices: `render_views` runs the loop body of src/main.rs:57-116,272-273
/// (clear_screen + render_mesh_component) for one mesh under `shaders.len()` cameras as one batch.
pub struct ViewBatchState {
    inner: ModelRendererState,
    max_views: usize,
}

impl ViewBatchState {
    pub fn new(frame_width: usize, frame_height: usize, max_views: usize) -> ViewBatchState {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sys::tr_create_views(0, frame_width as i32, frame_height as i32, max_views as i32, &mut ctx) };
        if rc != sys::TR_OK {
            panic!("tr_create_views failed: {rc}");
        }
        ViewBatchState {
            inner: ModelRendererState { ctx, frame_width, frame_height, meshes: HashMap::new(), textures: HashMap::new() },
            max_views,
        }
    }

    pub fn render_views(&mut self, mesh: &wavefront_obj::obj::Object, shaders: &[BuiltinShader]) {
        assert!(!shaders.is_empty() && shaders.len() <= self.max_views, "between 1 and max_views views per batch");
        let m = self.inner.mesh_handle(mesh);
        let params: Vec<sys::tr_draw_params> = shaders.iter().map(|s| draw_params(&mut self.inner, s)).collect();
        let rc = unsafe { sys::tr_render_views(self.inner.ctx, m, params.as_ptr(), params.len() as i32, sys::TR_OUT_RGBA8) };
        check(self.inner.ctx, rc, "tr_render_views");
    }

    /// The presented image of view `view` of the last batch.
    pub fn present(&mut self, view: usize) -> image::RgbaImage {
        let rc = unsafe { sys::tr_select_view(self.inner.ctx, view as i32) };
        check(self.inner.ctx, rc, "tr_select_view");
        self.inner.present()
    }
}
