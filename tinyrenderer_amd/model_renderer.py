"""Host-side mirror of the reference's `model_renderer` module (src/model_renderer/mod.rs).

`ModelRendererState`, `clear_screen`, `render_mesh_component`, `MeshComponent` and `Transform`
keep the reference's names and call shape; the two shader closures of
`render_mesh_component` (host code that cannot run on a GPU) are replaced by `BuiltinShader`,
a descriptor of the one shader pair the reference ships (src/main.rs:146-267).  All work is
done by the C-ABI library through `_ffi`; nothing here computes pixels.
"""
import ctypes as C

import numpy as np

from . import _ffi
from . import helpers as H


class Transform:
    """src/model_renderer/mod.rs:21-27,87-148."""

    def __init__(self):
        self._position = (0.0, 0.0, 0.0)
        self._rotation = (0.0, 0.0, 0.0)  # euler angles
        self._scale = (1.0, 1.0, 1.0)
        self._matrix = np.eye(4, dtype=np.float64)

    def position(self):
        return self._position

    def rotation(self):
        return self._rotation

    def scale(self):
        return self._scale

    def matrix(self):
        return self._matrix.copy()

    def set_position(self, new_position):
        # mod.rs:118-125: pokes m14/m24/m34 only, no resync
        self._position = tuple(float(v) for v in new_position)
        self._matrix[0, 3], self._matrix[1, 3], self._matrix[2, 3] = self._position

    def set_rotation(self, new_rotation):
        self._rotation = tuple(float(v) for v in new_rotation)
        self._resync_matrix()

    def set_scale(self, new_scale):
        self._scale = tuple(float(v) for v in new_scale)
        self._resync_matrix()

    def _resync_matrix(self):
        # mod.rs:140-147: T * R * S, left-associative
        r = self._rotation
        self._matrix = H.mat4_mul(
            H.mat4_mul(H.make_translation_matrix(self._position), H.make_rotation_matrix(r[0], r[1], r[2])),
            H.make_scale_matrix(self._scale),
        )


class MeshComponent:
    """src/model_renderer/mod.rs:12-19,37-85.  `texture` / `normal_map` are (H, W, 4) uint8 images
    already flipped vertically like the reference's `flip_vertically(load_nd_rgba_img_from_file(..))`."""

    def __init__(self, mesh, texture, normal_map, parent=None):
        self.mesh = mesh
        self.transform = Transform()
        self.texture = texture
        self.normal_map = normal_map
        self.parent = parent

    @classmethod
    def from_paths(cls, obj_path, texture_path, normal_map_path):
        """`MeshComponent::from((obj_path, texture_path, normal_map_path))`, mod.rs:37-50."""
        obj = H.load_model(obj_path)
        texture = H.flip_vertically(H.load_rgba_img_from_file(texture_path))
        normal_map = H.flip_vertically(H.load_rgba_img_from_file(normal_map_path))
        return cls(obj, texture, normal_map)

    def local_to_world_matrix(self):
        # mod.rs:67-84: fold identity * self * parent * grandparent ...
        acc = np.eye(4, dtype=np.float64)
        cur = self
        while cur is not None:
            acc = H.mat4_mul(acc, cur.transform._matrix)
            cur = cur.parent
        return acc


class BuiltinShader:
    """The vertex + fragment closure pair of src/main.rs:146-267 as data."""

    def __init__(self, perspective_matrix, model_view_matrix, camera_pos, use_normal_map=True,
                 light_intensity=1.0, ambient_light=1.0, vertex_color=H.WHITE, normal_matrix=None):
        self.perspective_matrix = np.asarray(perspective_matrix, dtype=np.float64)
        self.model_view_matrix = np.asarray(model_view_matrix, dtype=np.float64)
        if normal_matrix is None:
            # main.rs:129-135: model_view.transpose().try_inverse() or identity
            normal_matrix = H.try_inverse4(self.model_view_matrix.T)
            if normal_matrix is None:
                print(f"Failed to transform normal for: {self.model_view_matrix}")
                normal_matrix = np.eye(4, dtype=np.float64)
        self.normal_matrix = np.asarray(normal_matrix, dtype=np.float64)
        self.camera_pos = tuple(float(v) for v in camera_pos)
        self.use_normal_map = bool(use_normal_map)
        self.light_intensity = float(light_intensity)
        self.ambient_light = float(ambient_light)
        self.vertex_color = tuple(float(v) for v in vertex_color)

    def pmv(self):
        # main.rs:152-153: (perspective * model_view) * position
        return H.mat4_mul(self.perspective_matrix, self.model_view_matrix)


class ShadowMap:
    """EXTENSION (no counterpart in the reference): the z plane of a state in which the scene was
    drawn from the light, snapshotted on the GPU (tr_shadow_map_create).  `keep_host_copy` also keeps
    it as a (W, H) array in the reference layout, which is what the CPU restatement consumes."""

    def __init__(self, light_state, keep_host_copy=False):
        self._lib = light_state._lib
        h = C.c_void_p()
        _ffi.check(light_state._ctx, self._lib.tr_shadow_map_create(light_state._ctx, C.byref(h)), "tr_shadow_map_create")
        self._handle = h
        self.width, self.height = light_state.frame_width, light_state.frame_height
        self.z_host = light_state.z_buffer[..., 0] if keep_host_copy else None

    def update(self, light_state):
        """A new snapshot of `light_state` (same size) into this map: no allocation, no host wait."""
        _ffi.check(light_state._ctx, self._lib.tr_shadow_map_update(self._handle, light_state._ctx), "tr_shadow_map_update")
        if self.z_host is not None:
            self.z_host = light_state.z_buffer[..., 0]

    def z_reference(self):
        if self.z_host is None:
            raise ValueError("ShadowMap was created without keep_host_copy")
        return self.z_host

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.tr_shadow_map_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostShadow:
    """A shadow map that only lives on the host (for the CPU restatement): z in the reference layout."""

    def __init__(self, z_wh):
        self.z_host = np.ascontiguousarray(z_wh, dtype=np.float64)
        self.width, self.height = self.z_host.shape

    def z_reference(self):
        return self.z_host


class ShadeExt:
    """EXTENSION shaders (tr_shade_ext in include/tinyrenderer_hip.h; the reference has only the
    shader of main.rs:146-267): Gouraud / Phong switch, specular map, shadow map."""

    def __init__(self, model=_ffi.TR_SHADE_REFERENCE, specular_map=None, shininess=16.0, specular_strength=0.6,
                 shadow=None, shadow_from_view=None, shadow_bias=0.0, shadow_dim=0.3):
        self.model = int(model)
        self.specular_map = specular_map  # (H, W, 4) uint8, flipped like the other textures
        self.shininess = float(shininess)
        self.specular_strength = float(specular_strength)
        self.shadow = shadow
        self.shadow_from_view = (np.eye(4) if shadow_from_view is None else np.asarray(shadow_from_view, dtype=np.float64))
        self.shadow_bias = float(shadow_bias)
        self.shadow_dim = float(shadow_dim)


def viewport_matrix(frame_width, frame_height):
    """src/model_renderer/mod.rs:314-318: scale((W-1)/2, (H-1)/2, 1) * translate(1, 1, 0)."""
    return H.mat4_mul(H.make_scale_matrix(((frame_width - 1) / 2.0, (frame_height - 1) / 2.0, 1.0)),
                      H.make_translation_matrix((1.0, 1.0, 0.0)))


def shadow_from_view(view_shader, frame_width, frame_height, light_shader, light_width, light_height):
    """Matrix taking a viewport-space position of the camera frame to the light frame's viewport
    space (extension): VP_l * PMV_l * inverse(VP * PMV)."""
    cam = viewport_matrix(frame_width, frame_height) @ view_shader.pmv()
    lit = viewport_matrix(light_width, light_height) @ light_shader.pmv()
    return lit @ np.linalg.inv(cam)


class ModelRendererState:
    """src/model_renderer/mod.rs:29-35,150-173.  `frame_buffer` / `z_buffer` are read back from
    the GPU on access, in the reference's (W, H, C) layout."""

    def __init__(self, frame_width, frame_height, device=0, raster_mode=None, max_views=1, colour_exact=None):
        """`colour_exact`: True = the fragment arithmetic operation for operation as the reference's (0 LSB), False = fused
        (+-1 LSB, the library's default), None = the library's default or TR_SHADE_EXACT (tr_set_colour_exact).
        `max_views` > 1: a state whose planes hold that many W x H slices, for `render_views`
        (tr_create_views; the reference renders one view per loop iteration, main.rs:57-116)."""
        self._lib = _ffi.load()
        self.frame_width = int(frame_width)
        self.frame_height = int(frame_height)
        self.max_views = int(max_views)
        ctx = C.c_void_p()
        if self.max_views > 1:
            rc = self._lib.tr_create_views(int(device), self.frame_width, self.frame_height, self.max_views, C.byref(ctx))
        else:
            rc = self._lib.tr_create(int(device), self.frame_width, self.frame_height, C.byref(ctx))
        if rc != _ffi.TR_OK:
            raise _ffi.TrError(
                f"tr_create failed: {_ffi.TR_ERR.get(rc, rc)} (a gfx950 GPU is required; there is no CPU fallback)")
        self._ctx = ctx
        if raster_mode is not None:
            _ffi.check(ctx, self._lib.tr_set_raster_mode(ctx, int(raster_mode)), "tr_set_raster_mode")
        if colour_exact is not None:
            self.set_colour_exact(colour_exact)
        self._meshes = {}
        self._textures = {}
        self._keepalive = []

    # -- resources ----------------------------------------------------------------
    def _mesh_handle(self, obj):
        key = id(obj)
        if key not in self._meshes:
            h = C.c_void_p()
            v, t, n, tri = obj.vertices, obj.tex_vertices, obj.normals, obj.triangles
            rc = self._lib.tr_mesh_create(
                self._ctx, v.ctypes.data, v.shape[0], n.ctypes.data if n.size else None, n.shape[0],
                t.ctypes.data if t.size else None, t.shape[0], tri.ctypes.data if tri.size else None,
                tri.shape[0], C.byref(h))
            _ffi.check(self._ctx, rc, "tr_mesh_create")
            self._meshes[key] = h
            self._keepalive.append(obj)
        return self._meshes[key]

    def _texture_handle(self, img):
        if img is None:
            return None
        key = id(img)
        if key not in self._textures:
            img8 = np.ascontiguousarray(img, dtype=np.uint8)
            if img8.ndim != 3 or img8.shape[2] != 4:
                raise ValueError("texture must be (H, W, 4) uint8")
            h = C.c_void_p()
            rc = self._lib.tr_texture_create(self._ctx, img8.ctypes.data, img8.shape[1], img8.shape[0], C.byref(h))
            _ffi.check(self._ctx, rc, "tr_texture_create")
            self._textures[key] = h
            self._keepalive.append(img)
        return self._textures[key]

    # -- buffers --------------------------------------------------------------------
    def _read(self, fn, dtype, comps, layout):
        W, Hh = self.frame_width, self.frame_height
        shape = (W, Hh, comps) if layout == _ffi.TR_LAYOUT_REFERENCE else (Hh, W, comps)
        out = np.empty(shape, dtype=dtype)
        _ffi.check(self._ctx, fn(self._ctx, out.ctypes.data, layout), fn.__name__)
        return out

    @property
    def z_buffer(self):
        """`state.z_buffer.nd_img`: (W, H, 1) float64, -inf where nothing was drawn."""
        return self._read(self._lib.tr_read_z_f64, np.float64, 1, _ffi.TR_LAYOUT_REFERENCE)

    @property
    def frame_buffer(self):
        """`state.frame_buffer.nd_img`: (W, H, 4) float64."""
        return self._read(self._lib.tr_read_rgba_f64, np.float64, 4, _ffi.TR_LAYOUT_REFERENCE)

    def triangle_ids(self, layout=_ffi.TR_LAYOUT_REFERENCE):
        return self._read(self._lib.tr_read_triangle_id_u32, np.uint32, 1, layout)[..., 0]

    def winner_keys(self, layout=_ffi.TR_LAYOUT_REFERENCE):
        return self._read(self._lib.tr_read_winner_key_u64, np.uint64, 1, layout)[..., 0]

    def z_row_major(self):
        return self._read(self._lib.tr_read_z_f64, np.float64, 1, _ffi.TR_LAYOUT_ROW_MAJOR)[..., 0]

    def present(self, pinned=False):
        """`ndarray_to_image_rgba_and_flip(&frame_buffer)` (helpers/mod.rs:204-222): (H, W, 4) uint8, row 0 = top.
        `pinned`: deliver into a page-locked buffer owned by this state (tr_host_alloc) and return a
        view of it, valid until the next call -- the copy then runs at the PCIe rate."""
        shape = (self.frame_height, self.frame_width, 4)
        if not pinned:
            out = np.empty(shape, dtype=np.uint8)
            _ffi.check(self._ctx, self._lib.tr_read_rgba8(self._ctx, out.ctypes.data, 1), "tr_read_rgba8")
            return out
        n = shape[0] * shape[1] * 4
        if getattr(self, "_pinned", None) is None:
            p = self._lib.tr_host_alloc(n)
            if not p:
                raise _ffi.TrError("tr_host_alloc failed")
            self._pinned = p
        _ffi.check(self._ctx, self._lib.tr_read_rgba8(self._ctx, self._pinned, 1), "tr_read_rgba8")
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(self._pinned)).reshape(shape)

    def select_view(self, view):
        """The slice of a `render_views` batch that `z_buffer`, `frame_buffer`, `present()` ... refer to."""
        _ffi.check(self._ctx, self._lib.tr_select_view(self._ctx, int(view)), "tr_select_view")

    def set_band(self, y_lo, y_hi):
        """Restrict this context to pixel rows [y_lo, y_hi) (whole 16-pixel tile rows); call after
        clear_screen and before the first draw."""
        _ffi.check(self._ctx, self._lib.tr_set_band(self._ctx, int(y_lo), int(y_hi)), "tr_set_band")

    def set_colour_exact(self, exact):
        """tr_set_colour_exact: the reference's colour arithmetic operation for operation (True) or fused (False)."""
        _ffi.check(self._ctx, self._lib.tr_set_colour_exact(self._ctx, 1 if exact else 0), "tr_set_colour_exact")

    def resolve(self, out_mask=_ffi.TR_OUT_RGBA8):
        _ffi.check(self._ctx, self._lib.tr_resolve(self._ctx, out_mask), "tr_resolve")

    def synchronize(self):
        _ffi.check(self._ctx, self._lib.tr_synchronize(self._ctx), "tr_synchronize")

    def counters(self):
        out = (C.c_uint64 * 8)()
        _ffi.check(self._ctx, self._lib.tr_get_counters(self._ctx, out), "tr_get_counters")
        names = ["samples", "fragments", "scanlines", "triangles", "covered_pixels", "tile_sample_evals",
                 "tile_segments", "tile_lane_steps"]
        return {k: int(out[i]) for i, k in enumerate(names)}

    def status_flags(self):
        f = C.c_uint32()
        _ffi.check(self._ctx, self._lib.tr_get_status_flags(self._ctx, C.byref(f)), "tr_get_status_flags")
        return int(f.value)

    def close(self):
        if getattr(self, "_ctx", None):
            for h in self._meshes.values():
                self._lib.tr_mesh_destroy(self._ctx, h)
            for h in self._textures.values():
                self._lib.tr_texture_destroy(self._ctx, h)
            if getattr(self, "_pinned", None):
                self._lib.tr_host_free(self._pinned)
                self._pinned = None
            self._lib.tr_destroy(self._ctx)
            self._ctx = None
            self._meshes, self._textures, self._keepalive = {}, {}, []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def clear_screen(model_renderer_state):
    """src/model_renderer/mod.rs:210-219."""
    s = model_renderer_state
    _ffi.check(s._ctx, s._lib.tr_clear(s._ctx), "tr_clear")


def make_draw_params(state, shader, texture, normal_map):
    p = _ffi.DrawParams()
    pmv = shader.pmv()
    p.pmv[:] = [float(v) for v in pmv.T.reshape(16)]  # column-major
    p.normal_mat[:] = [float(v) for v in shader.normal_matrix.T.reshape(16)]
    p.light_dir[:] = list(shader.camera_pos)
    p.ambient = shader.ambient_light
    p.intensity = shader.light_intensity
    p.vertex_color[:] = list(shader.vertex_color)
    p.use_normal_map = 1 if shader.use_normal_map else 0
    p.diffuse = state._texture_handle(texture)
    p.normal_map = state._texture_handle(normal_map)
    return p


def render_views(model_renderer_state, mesh_component, shaders, out_mask=_ffi.TR_OUT_RGBA8):
    """One iteration of the reference's loop per shader -- `clear_screen` + `render_mesh_component(state,
    &mesh, VS, FS)` (src/main.rs:57-116,272-273) -- for ONE mesh under len(shaders) cameras, as a single batch
    (tr_render_views): view v lands in slice v of a state made with `max_views`; `select_view(v)` picks the slice
    the buffers read."""
    s = model_renderer_state
    n = len(shaders)
    arr = (_ffi.DrawParams * n)()
    for v, sh in enumerate(shaders):
        arr[v] = make_draw_params(s, sh, mesh_component.texture, mesh_component.normal_map)
    rc = s._lib.tr_render_views(s._ctx, s._mesh_handle(mesh_component.mesh), arr, n, out_mask)
    _ffi.check(s._ctx, rc, "tr_render_views")


def render_views_multi(model_renderer_state, mesh_components, shaders, out_mask=_ffi.TR_OUT_RGBA8):
    """`render_views` for the reference's own scene: every iteration of its loop draws several meshes (main.rs:272-273:
    the head, then the eye parented to it).  `shaders[v][m]` is the shader of mesh m under view v (its model_view holds
    the mesh's local_to_world_matrix, parents folded in: mod.rs:67-84).  One batch, one slice per view
    (tr_render_views_multi)."""
    s = model_renderer_state
    n_views, n_meshes = len(shaders), len(mesh_components)
    arr = (_ffi.DrawParams * (n_views * n_meshes))()
    for v, row in enumerate(shaders):
        assert len(row) == n_meshes
        for m, sh in enumerate(row):
            arr[v * n_meshes + m] = make_draw_params(s, sh, mesh_components[m].texture, mesh_components[m].normal_map)
    handles = (C.c_void_p * n_meshes)(*[s._mesh_handle(mc.mesh) for mc in mesh_components])
    rc = s._lib.tr_render_views_multi(s._ctx, n_meshes, handles, arr, n_views, out_mask)
    _ffi.check(s._ctx, rc, "tr_render_views_multi")


def make_shade_ext(state, ext):
    x = _ffi.ShadeExt()
    x.model = ext.model
    x.specular_map = state._texture_handle(ext.specular_map)
    x.shininess = ext.shininess
    x.specular_strength = ext.specular_strength
    if ext.shadow is not None:
        if not isinstance(ext.shadow, ShadowMap):
            raise TypeError("the GPU path needs a ShadowMap (tr_shadow_map_create)")
        x.shadow = ext.shadow._handle
    x.shadow_from_view[:] = [float(v) for v in ext.shadow_from_view.T.reshape(16)]  # column-major
    x.shadow_bias = ext.shadow_bias
    x.shadow_dim = ext.shadow_dim
    return x


def render_mesh_component(model_renderer_state, mesh_component, shader, ext=None):
    """`render_mesh_component(state, &mesh, VS, FS)` (src/model_renderer/mod.rs:225-230) with the
    closure pair replaced by `shader` (a BuiltinShader).  Takes the MeshComponent so the
    shader's texture / normal map (captured by the reference's closures, main.rs:140-141) come
    along.  `ext` (a ShadeExt) selects an extension shader; the reference has none."""
    s = model_renderer_state
    p = make_draw_params(s, shader, mesh_component.texture, mesh_component.normal_map)
    if ext is None:
        rc = s._lib.tr_draw(s._ctx, s._mesh_handle(mesh_component.mesh), C.byref(p))
        _ffi.check(s._ctx, rc, "tr_draw")
    else:
        x = make_shade_ext(s, ext)
        rc = s._lib.tr_draw_ext(s._ctx, s._mesh_handle(mesh_component.mesh), C.byref(p), C.byref(x))
        _ffi.check(s._ctx, rc, "tr_draw_ext")
