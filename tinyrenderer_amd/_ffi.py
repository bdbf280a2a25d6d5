"""ctypes binding of the C-ABI library (include/tinyrenderer_hip.h).

The library is the product; this module only declares its entry points.  There is no
fallback: if the shared object is missing or no gfx950 GPU is present the calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TINYRENDERER_HIP_LIB: an alternative build of the same library, for A/B timing; TINYRENDERER_HIP_DIAG=1: the
# diagnostic build, `make diag`, which alone reads the experiment / trace switches -- INTEGRATION.md 4.1)
LIB_PATH = os.environ.get("TINYRENDERER_HIP_LIB") or os.path.join(
    _HERE, "libtinyrenderer_hip_diag.so" if os.environ.get("TINYRENDERER_HIP_DIAG") == "1" else "libtinyrenderer_hip.so")

TR_OK = 0
TR_ERR = {
    -1: "TR_ERR_INVALID",
    -2: "TR_ERR_HIP",
    -3: "TR_ERR_NO_DEVICE",
    -4: "TR_ERR_OOM",
    -5: "TR_ERR_UNSUPPORTED",
    -6: "TR_ERR_SIZE",
    -7: "TR_ERR_BUSY",
}
TR_FLAG_TRI_TOO_LARGE = 1 << 0
TR_FLAG_REF_PANIC_SPLIT = 1 << 1
TR_FLAG_TEX_OOB = 1 << 2
TR_FLAG_NEG_INF_DEPTH = 1 << 3
TR_FLAG_SCANLINE_OVERFLOW = 1 << 4
TR_FLAG_NAN_PIXEL_OVERFLOW = 1 << 5
TR_FLAG_BIN_OVERFLOW = 1 << 6
TR_FLAG_PREFIX_STALL = 1 << 7
TR_RASTER_GLOBAL, TR_RASTER_TILE, TR_RASTER_ROWS = 0, 1, 2
TR_LAYOUT_ROW_MAJOR = 0
TR_LAYOUT_REFERENCE = 1
TR_OUT_RGBA8, TR_OUT_Z, TR_OUT_TRI_ID, TR_OUT_RGBA_F64, TR_OUT_KEY = 1, 2, 4, 8, 16
TR_SHADE_REFERENCE, TR_SHADE_GOURAUD, TR_SHADE_PHONG = 0, 1, 2  # extension shaders (tr_draw_ext)

# Every symbol include/tinyrenderer_hip.h declares (checked by tests/test_abi.py).
EXPORTS = [
    "tr_create", "tr_destroy", "tr_last_error", "tr_clear", "tr_mesh_create", "tr_mesh_destroy",
    "tr_texture_create", "tr_texture_destroy", "tr_draw", "tr_resolve", "tr_read_z_f64",
    "tr_read_triangle_id_u32", "tr_read_winner_key_u64", "tr_read_rgba8", "tr_read_rgba_f64",
    "tr_get_counters", "tr_get_status_flags", "tr_synchronize", "tr_get_stream",
    "tr_device_ptr_rgba8", "tr_profile_enable", "tr_profile_read", "tr_version", "tr_set_band",
    "tr_set_raster_mode", "tr_obj_load", "tr_obj_parse", "tr_obj_counts", "tr_obj_positions", "tr_obj_uvw",
    "tr_obj_normals", "tr_obj_indices", "tr_obj_free", "tr_image_load_rgba8", "tr_image_size", "tr_image_data",
    "tr_image_free", "tr_load_error", "tr_image_write_png", "tr_selftest_arithmetic",
    "tr_draw_ext", "tr_shadow_map_create", "tr_shadow_map_update", "tr_shadow_map_destroy", "tr_host_alloc", "tr_host_free",
    "tr_read_rgba8_static", "tr_create_views", "tr_render_views", "tr_select_view", "tr_set_colour_exact", "tr_render_views_multi", "tr_get_shade_launch",
]


class DrawParams(C.Structure):
    _fields_ = [
        ("pmv", C.c_double * 16),
        ("normal_mat", C.c_double * 16),
        ("light_dir", C.c_double * 3),
        ("ambient", C.c_double),
        ("intensity", C.c_double),
        ("vertex_color", C.c_double * 4),
        ("use_normal_map", C.c_int32),
        ("reserved", C.c_int32),
        ("diffuse", C.c_void_p),
        ("normal_map", C.c_void_p),
    ]


class ShadeExt(C.Structure):
    """tr_shade_ext: extension shaders, no counterpart in the reference."""
    _fields_ = [
        ("model", C.c_int32),
        ("reserved", C.c_int32),
        ("specular_map", C.c_void_p),
        ("shininess", C.c_double),
        ("specular_strength", C.c_double),
        ("shadow", C.c_void_p),
        ("shadow_from_view", C.c_double * 16),
        ("shadow_bias", C.c_double),
        ("shadow_dim", C.c_double),
    ]


_lib = None


def load():
    """Load the shared library (does not touch the GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C tinyrenderer_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback"
        )
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32
    pd = C.POINTER(C.c_double)
    sig = {
        "tr_create": (i32, [i32, i32, i32, C.POINTER(vp)]),
        "tr_destroy": (None, [vp]),
        "tr_last_error": (C.c_char_p, [vp]),
        "tr_clear": (i32, [vp]),
        "tr_mesh_create": (i32, [vp, vp, i64, vp, i64, vp, i64, vp, i64, C.POINTER(vp)]),
        "tr_mesh_destroy": (i32, [vp, vp]),
        "tr_texture_create": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
        "tr_texture_destroy": (i32, [vp, vp]),
        "tr_draw": (i32, [vp, vp, C.POINTER(DrawParams)]),
        "tr_resolve": (i32, [vp, u32]),
        "tr_read_z_f64": (i32, [vp, vp, i32]),
        "tr_read_triangle_id_u32": (i32, [vp, vp, i32]),
        "tr_read_winner_key_u64": (i32, [vp, vp, i32]),
        "tr_read_rgba8": (i32, [vp, vp, i32]),
        "tr_read_rgba_f64": (i32, [vp, vp, i32]),
        "tr_read_rgba8_static": (i32, [vp, vp, i32]),
        "tr_create_views": (i32, [i32, i32, i32, i32, C.POINTER(vp)]),
        "tr_render_views": (i32, [vp, vp, C.POINTER(DrawParams), i32, u32]),
        "tr_select_view": (i32, [vp, i32]),
        "tr_render_views_multi": (i32, [vp, i32, C.POINTER(vp), C.POINTER(DrawParams), i32, u32]),
        "tr_get_counters": (i32, [vp, C.POINTER(C.c_uint64)]),
        "tr_get_status_flags": (i32, [vp, C.POINTER(u32)]),
        "tr_synchronize": (i32, [vp]),
        "tr_get_stream": (i32, [vp, C.POINTER(vp)]),
        "tr_device_ptr_rgba8": (i32, [vp, C.POINTER(vp)]),
        "tr_profile_enable": (i32, [vp, i32]),
        "tr_get_shade_launch": (i32, [vp, C.POINTER(u32)]),
        "tr_profile_read": (i32, [vp, pd, C.POINTER(C.c_uint64), i32]),
        "tr_version": (C.c_char_p, []),
        "tr_set_band": (i32, [vp, i32, i32]),
        "tr_set_raster_mode": (i32, [vp, i32]),
        "tr_set_colour_exact": (i32, [vp, i32]),
        "tr_obj_load": (i32, [C.c_char_p, C.POINTER(vp)]),
        "tr_obj_parse": (i32, [C.c_char_p, i64, C.POINTER(vp)]),
        "tr_obj_counts": (i32, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
        "tr_obj_positions": (pd, [vp]),
        "tr_obj_uvw": (pd, [vp]),
        "tr_obj_normals": (pd, [vp]),
        "tr_obj_indices": (C.POINTER(C.c_uint32), [vp]),
        "tr_obj_free": (None, [vp]),
        "tr_image_load_rgba8": (i32, [C.c_char_p, i32, C.POINTER(vp)]),
        "tr_image_size": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "tr_image_data": (C.POINTER(C.c_uint8), [vp]),
        "tr_image_free": (None, [vp]),
        "tr_load_error": (C.c_char_p, []),
        "tr_image_write_png": (i32, [C.c_char_p, vp, i32, i32]),
        "tr_selftest_arithmetic": (i32, [i32, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
        "tr_draw_ext": (i32, [vp, vp, C.POINTER(DrawParams), C.POINTER(ShadeExt)]),
        "tr_shadow_map_create": (i32, [vp, C.POINTER(vp)]),
        "tr_shadow_map_update": (i32, [vp, vp]),
        "tr_shadow_map_destroy": (None, [vp]),
        "tr_host_alloc": (vp, [C.c_size_t]),
        "tr_host_free": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class TrError(RuntimeError):
    pass


def check(ctx, rc, what):
    if rc != TR_OK:
        msg = ""
        if ctx:
            msg = load().tr_last_error(ctx).decode(errors="replace")
        raise TrError(f"{what} failed: {TR_ERR.get(rc, rc)} {msg}")
