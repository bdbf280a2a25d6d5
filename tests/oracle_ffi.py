"""Test-infrastructure binding of oracle/liboracle.so (the CPU restatement of the reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class OraMesh(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("uvw", C.c_void_p), ("idx", C.c_void_p),
                ("n_pos", C.c_int64), ("n_nrm", C.c_int64), ("n_uvw", C.c_int64), ("n_tris", C.c_int64)]


class OraShader(C.Structure):
    _fields_ = [("perspective", C.c_double * 16), ("model_view", C.c_double * 16),
                ("normal_matrix", C.c_double * 16), ("camera_pos", C.c_double * 3),
                ("light_intensity", C.c_double), ("ambient_light", C.c_double),
                ("use_normal_map", C.c_int32), ("_pad", C.c_int32),
                ("texture", C.c_void_p), ("normal_map", C.c_void_p)]


class OraShaderExt(C.Structure):  # extension shaders (no counterpart in the reference)
    _fields_ = [("model", C.c_int32), ("_pad", C.c_int32), ("specular_map", C.c_void_p),
                ("shininess", C.c_double), ("specular_strength", C.c_double),
                ("shadow", C.c_void_p), ("shadow_w", C.c_int64), ("shadow_h", C.c_int64),
                ("shadow_from_view", C.c_double * 16), ("shadow_bias", C.c_double), ("shadow_dim", C.c_double)]


_lib = None


def load(build_if_missing=True):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and build_if_missing:
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    lib = C.CDLL(LIB_PATH)
    vp, i64, pd = C.c_void_p, C.c_int64, C.POINTER(C.c_double)
    lib.ora_state_new.restype = vp
    lib.ora_state_new.argtypes = [i64, i64]
    lib.ora_state_free.argtypes = [vp]
    lib.ora_clear_screen.argtypes = [vp]
    lib.ora_image_from_rgba8.restype = vp
    lib.ora_image_from_rgba8.argtypes = [vp, i64, i64]
    lib.ora_image_flip_vertically.restype = vp
    lib.ora_image_flip_vertically.argtypes = [vp]
    lib.ora_image_free.argtypes = [vp]
    lib.ora_image_data.restype = pd
    lib.ora_image_data.argtypes = [vp]
    lib.ora_sample_nd_img.restype = C.c_int
    lib.ora_sample_nd_img.argtypes = [vp, C.c_double, C.c_double, pd]
    lib.ora_render_mesh_component.restype = C.c_int
    lib.ora_render_mesh_component.argtypes = [vp, C.POINTER(OraMesh), C.POINTER(OraShader)]
    lib.ora_render_mesh_component_ext.restype = C.c_int
    lib.ora_render_mesh_component_ext.argtypes = [vp, C.POINTER(OraMesh), C.POINTER(OraShader), C.POINTER(OraShaderExt)]
    for n in ("ora_z_buffer", "ora_frame_buffer"):
        getattr(lib, n).restype = pd
        getattr(lib, n).argtypes = [vp]
    for n in ("ora_win_draw", "ora_win_tri", "ora_win_half", "ora_win_i", "ora_win_j"):
        getattr(lib, n).restype = C.POINTER(C.c_int64)
        getattr(lib, n).argtypes = [vp]
    lib.ora_counters.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.ora_present_rgba8_flip.argtypes = [vp, vp]
    lib.ora_segment_points.restype = i64
    lib.ora_segment_points.argtypes = [pd, pd, i64, pd, i64]
    lib.ora_barycentric.argtypes = [pd, pd, pd, pd, pd]
    lib.ora_check_unit_param.restype = i64
    lib.ora_check_unit_param.argtypes = [i64, i64]
    lib.ora_mat4_mul.argtypes = [pd, pd, pd]
    _lib = lib
    return lib


def _cm16(m):
    return [float(v) for v in np.asarray(m, dtype=np.float64).T.reshape(16)]


class OracleImage:
    """MyRgbaImage: built from a decoded (H, W, 4) uint8 image (row 0 = top), then flipped by
    the oracle's own flip_vertically when `flip` is set (mirrors model_renderer/mod.rs:40-41)."""

    def __init__(self, rgba_hw4_top_first, flip=True):
        lib = load()
        img = np.ascontiguousarray(rgba_hw4_top_first, dtype=np.uint8)
        self.h, self.w = img.shape[0], img.shape[1]
        raw = lib.ora_image_from_rgba8(img.ctypes.data, self.w, self.h)
        if flip:
            self.ptr = lib.ora_image_flip_vertically(raw)
            lib.ora_image_free(raw)
        else:
            self.ptr = raw

    def sample(self, x, y):
        out = (C.c_double * 4)()
        rc = load().ora_sample_nd_img(self.ptr, x, y, out)
        return rc, list(out)

    def __del__(self):
        try:
            load().ora_image_free(self.ptr)
        except Exception:
            pass


class OracleState:
    def __init__(self, w, h):
        self.lib = load()
        self.w, self.h = int(w), int(h)
        self.ptr = self.lib.ora_state_new(self.w, self.h)

    def clear_screen(self):
        self.lib.ora_clear_screen(self.ptr)

    def render_mesh_component(self, obj, shader, texture, normal_map, ext=None, specular_map=None):
        """obj: helpers.Object; shader: model_renderer.BuiltinShader; texture/normal_map: OracleImage;
        ext: model_renderer.ShadeExt (extension shaders) with specular_map as an OracleImage."""
        m = OraMesh()
        v, t, n, tri = obj.vertices, obj.tex_vertices, obj.normals, obj.triangles
        m.positions, m.normals, m.uvw, m.idx = v.ctypes.data, n.ctypes.data, t.ctypes.data, tri.ctypes.data
        m.n_pos, m.n_nrm, m.n_uvw, m.n_tris = v.shape[0], n.shape[0], t.shape[0], tri.shape[0]
        s = OraShader()
        s.perspective[:] = _cm16(shader.perspective_matrix)
        s.model_view[:] = _cm16(shader.model_view_matrix)
        s.normal_matrix[:] = _cm16(shader.normal_matrix)
        s.camera_pos[:] = list(shader.camera_pos)
        s.light_intensity = shader.light_intensity
        s.ambient_light = shader.ambient_light
        s.use_normal_map = 1 if shader.use_normal_map else 0
        s.texture = texture.ptr if texture is not None else None
        s.normal_map = normal_map.ptr if normal_map is not None else None
        if ext is None:
            return self.lib.ora_render_mesh_component(self.ptr, C.byref(m), C.byref(s))
        x = OraShaderExt()
        x.model = ext.model
        x.specular_map = specular_map.ptr if specular_map is not None else None
        x.shininess, x.specular_strength = ext.shininess, ext.specular_strength
        z = None
        if ext.shadow is not None:
            z = np.ascontiguousarray(ext.shadow.z_reference(), dtype=np.float64)
            x.shadow, x.shadow_w, x.shadow_h = z.ctypes.data, z.shape[0], z.shape[1]
        x.shadow_from_view[:] = _cm16(ext.shadow_from_view)
        x.shadow_bias, x.shadow_dim = ext.shadow_bias, ext.shadow_dim
        return self.lib.ora_render_mesh_component_ext(self.ptr, C.byref(m), C.byref(s), C.byref(x))

    def _arr(self, fn, dtype, comps):
        p = fn(self.ptr)
        n = self.w * self.h * comps
        a = np.ctypeslib.as_array(p, shape=(n,)).astype(dtype, copy=True)
        return a.reshape(self.w, self.h, comps)

    def z_buffer(self):
        return self._arr(self.lib.ora_z_buffer, np.float64, 1)

    def frame_buffer(self):
        return self._arr(self.lib.ora_frame_buffer, np.float64, 4)

    def winners(self):
        """dict of (W,H) int64 arrays: draw, tri, half, i, j (-1 = none)."""
        return {k: self._arr(getattr(self.lib, "ora_win_" + k), np.int64, 1)[..., 0]
                for k in ("draw", "tri", "half", "i", "j")}

    def counters(self):
        out = (C.c_int64 * 7)()
        self.lib.ora_counters(self.ptr, out)
        names = ["samples", "fragments", "zpass", "scanlines", "culled", "tex_panic", "point_err"]
        return {k: int(out[i]) for i, k in enumerate(names)}

    def present(self):
        out = np.empty((self.h, self.w, 4), dtype=np.uint8)
        self.lib.ora_present_rgba8_flip(self.ptr, out.ctypes.data)
        return out

    def __del__(self):
        try:
            self.lib.ora_state_free(self.ptr)
        except Exception:
            pass
