"""Host-side mirror of the reference's `helpers` module (src/helpers/mod.rs).

Same names, argument meaning and conventions as the Rust functions so scenes written
against the reference read the same here.  Matrices are 4x4 numpy float64 arrays indexed
[row, col]; every product is evaluated scalar by scalar in the association nalgebra 0.30.1
uses (gemv per column, left-to-right accumulation), never through BLAS.
"""
import math
import os

import numpy as np

# src/helpers/mod.rs:22-26
BLACK = (0.0, 0.0, 0.0, 255.0)
RED = (255.0, 0.0, 0.0, 255.0)
GREEN = (0.0, 255.0, 0.0, 255.0)
BLUE = (0.0, 0.0, 255.0, 255.0)
WHITE = (255.0, 255.0, 255.0, 255.0)

NONE_INDEX = 0xFFFFFFFF


# ----------------------------------------------------------------- matrices
def mat4_mul(a, b):
    """nalgebra `a * b` for 4x4: C[r][j] = ((a[r][0]b[0][j] + a[r][1]b[1][j]) + a[r][2]b[2][j]) + a[r][3]b[3][j]."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    c = np.zeros((4, 4), dtype=np.float64)
    for j in range(4):
        for r in range(4):
            acc = float(a[r, 0]) * float(b[0, j])
            for k in range(1, 4):
                acc = float(a[r, k]) * float(b[k, j]) + acc
            c[r, j] = acc
    return c


def mat4_mul_vec4(a, v):
    a = np.asarray(a, dtype=np.float64)
    out = []
    for r in range(4):
        acc = float(a[r, 0]) * float(v[0])
        for k in range(1, 4):
            acc = float(a[r, k]) * float(v[k]) + acc
        out.append(acc)
    return np.array(out, dtype=np.float64)


def make_rotation_matrix(pitch, yaw, roll):
    """src/helpers/mod.rs:520-525."""
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    cr, sr = math.cos(roll), math.sin(roll)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, 0.0],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, 0.0],
            [-sp, cp * sr, cp * cr, 0.0],
            [0.0, 0.0, 0.0, 1.0],
        ],
        dtype=np.float64,
    )


def make_translation_matrix(t):
    """src/helpers/mod.rs:527-532."""
    m = np.eye(4, dtype=np.float64)
    m[0, 3], m[1, 3], m[2, 3] = float(t[0]), float(t[1]), float(t[2])
    return m


def make_scale_matrix(s):
    """src/helpers/mod.rs:534-539."""
    m = np.eye(4, dtype=np.float64)
    m[0, 0], m[1, 1], m[2, 2] = float(s[0]), float(s[1]), float(s[2])
    return m


def make_perspective_matrix(near_plane_distance, far_plane_distance, vertical_fov, aspect_ratio):
    """src/helpers/mod.rs:542-556 (the reference calls it with near=10, far=1)."""
    n = float(near_plane_distance)
    f = float(far_plane_distance)
    r = n * math.tan(vertical_fov / 2.0) * aspect_ratio
    t = n * math.tan(vertical_fov / 2.0)
    return np.array(
        [
            [n / r, 0.0, 0.0, 0.0],
            [0.0, n / t, 0.0, 0.0],
            [0.0, 0.0, -1.0 * (f + n) / (f - n), (-2.0 * f * n) / (f - n)],
            [0.0, 0.0, -1.0, 0.0],
        ],
        dtype=np.float64,
    )


def _normalize3(v):
    x, y, z = float(v[0]), float(v[1]), float(v[2])
    n = math.sqrt(0.0 + ((x * x + y * y) + z * z))
    return (x / n, y / n, z / n)


def _cross3(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def look_at_rh(eye, target, up):
    """nalgebra `Matrix4::look_at_rh` as used at src/main.rs:66-70 (IsometryMatrix3 form).

    Third-party arithmetic restated from the crate's published algorithm; the matrix is an
    input to both the oracle and the GPU path, so it never enters a parity comparison.
    """
    d = (target[0] - eye[0], target[1] - eye[1], target[2] - eye[2])
    z = _normalize3((-d[0], -d[1], -d[2]))
    x = _normalize3(_cross3(up, z))
    y = _normalize3(_cross3(z, x))
    # rotation = face_towards(-dir, up).inverse() = transpose of [x y z] columns
    rot = [[x[0], x[1], x[2]], [y[0], y[1], y[2]], [z[0], z[1], z[2]]]
    ne = (-float(eye[0]), -float(eye[1]), -float(eye[2]))
    m = np.eye(4, dtype=np.float64)
    for r in range(3):
        for c in range(3):
            m[r, c] = rot[r][c]
        m[r, 3] = (rot[r][0] * ne[0] + rot[r][1] * ne[1]) + rot[r][2] * ne[2]
    return m


def try_inverse4(m):
    """nalgebra `Matrix4::try_inverse` (the MESA cofactor expansion on the column-major slice).

    Returns None when the determinant is exactly zero (src/main.rs:129-135 falls back to the
    identity in that case).
    """
    a = [float(v) for v in np.asarray(m, dtype=np.float64).T.reshape(16)]  # column-major slice
    inv = [0.0] * 16
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10]
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10]
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9]
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9]
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10]
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10]
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9]
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9]
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6]
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6]
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5]
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5]
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6]
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6]
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5]
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5]
    det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12]
    if det == 0.0:
        return None
    inv_det = 1.0 / det
    out = np.array([v * inv_det for v in inv], dtype=np.float64).reshape(4, 4).T  # back to [row, col]
    return out


# ------------------------------------------------------------------ loaders
class Object:
    """The part of `wavefront_obj::obj::Object` that render_mesh_component consumes
    (src/model_renderer/mod.rs:237-276): vertices, tex_vertices, normals and the triangles
    of geometry[0] as (v, vt, vn) index triples, 0-based, NONE_INDEX for an absent part."""

    def __init__(self, vertices, tex_vertices, normals, triangles, name=""):
        self.name = name
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        self.tex_vertices = np.ascontiguousarray(tex_vertices, dtype=np.float64).reshape(-1, 3)
        self.normals = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
        self.triangles = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 9)


def parse_obj(text):
    """`wavefront_obj::obj::parse(..).objects.remove(0)` for what the renderer reads: first
    object, geometry[0] (faces before the first `usemtl` switch after faces began), polygons
    fan-triangulated, indices made 0-based (negative = relative to the end)."""
    verts, tex, nrm, tris = [], [], [], []
    name = ""
    seen_object = False
    geometry_closed = False
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        parts = line.split()
        tag, args = parts[0], parts[1:]
        if tag == "o":
            if seen_object or verts or tris:
                break  # only the first object is kept (helpers/mod.rs:353-354)
            seen_object = True
            name = " ".join(args)
        elif tag == "v":
            verts.append([float(args[0]), float(args[1]), float(args[2])])
        elif tag == "vt":
            u = float(args[0])
            v = float(args[1]) if len(args) > 1 else 0.0
            w = float(args[2]) if len(args) > 2 else 0.0
            tex.append([u, v, w])
        elif tag == "vn":
            nrm.append([float(args[0]), float(args[1]), float(args[2])])
        elif tag == "usemtl":
            if tris:
                geometry_closed = True  # later faces belong to geometry[1..], never drawn
        elif tag == "f" and not geometry_closed:
            corners = []
            for a in args:
                f = a.split("/")
                vi = int(f[0])
                ti = int(f[1]) if len(f) > 1 and f[1] else None
                ni = int(f[2]) if len(f) > 2 and f[2] else None
                vi = vi - 1 if vi > 0 else len(verts) + vi
                if ti is not None:
                    ti = ti - 1 if ti > 0 else len(tex) + ti
                if ni is not None:
                    ni = ni - 1 if ni > 0 else len(nrm) + ni
                corners.append((vi, NONE_INDEX if ti is None else ti, NONE_INDEX if ni is None else ni))
            for k in range(1, len(corners) - 1):
                tris.append([*corners[0], *corners[k], *corners[k + 1]])
    return Object(
        np.array(verts, dtype=np.float64).reshape(-1, 3),
        np.array(tex, dtype=np.float64).reshape(-1, 3),
        np.array(nrm, dtype=np.float64).reshape(-1, 3),
        np.array(tris, dtype=np.uint32).reshape(-1, 9),
        name,
    )


def _native():
    from . import _ffi

    return _ffi, _ffi.load()


def _object_from_native(lib, h):
    import ctypes as C

    n = [C.c_int64() for _ in range(4)]
    lib.tr_obj_counts(h, *[C.byref(x) for x in n])
    n_pos, n_uvw, n_nrm, n_tri = (int(x.value) for x in n)

    def arr(ptr, count, dtype):
        if count == 0:
            return np.zeros((0,), dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)

    return Object(arr(lib.tr_obj_positions(h), 3 * n_pos, np.float64), arr(lib.tr_obj_uvw(h), 3 * n_uvw, np.float64),
                  arr(lib.tr_obj_normals(h), 3 * n_nrm, np.float64), arr(lib.tr_obj_indices(h), 9 * n_tri, np.uint32))


def load_model(path):
    """src/helpers/mod.rs:352-355, through the library's native OBJ parser (`tr_obj_load`)."""
    import ctypes as C

    ffi, lib = _native()
    h = C.c_void_p()
    rc = lib.tr_obj_load(os.fsencode(path), C.byref(h))
    if rc != ffi.TR_OK:
        raise ffi.TrError(f"tr_obj_load({path}): {ffi.TR_ERR.get(rc, rc)} {lib.tr_load_error().decode(errors='replace')}")
    try:
        return _object_from_native(lib, h)
    finally:
        lib.tr_obj_free(h)


def parse_obj_native(text):
    """`wavefront_obj::obj::parse` on a string through the library (`tr_obj_parse`)."""
    import ctypes as C

    ffi, lib = _native()
    data = text.encode()
    h = C.c_void_p()
    rc = lib.tr_obj_parse(data, len(data), C.byref(h))
    if rc != ffi.TR_OK:
        raise ffi.TrError(f"tr_obj_parse: {ffi.TR_ERR.get(rc, rc)} {lib.tr_load_error().decode(errors='replace')}")
    try:
        return _object_from_native(lib, h)
    finally:
        lib.tr_obj_free(h)


def load_rgba_img_from_file(path, flip=False):
    """src/helpers/mod.rs:93-96: `image::open(path)?.into_rgba8()` -> (H, W, 4) uint8, row 0 = top
    (row 0 = bottom with flip=True, i.e. followed by `flip_vertically`).  Decoded by the library's
    own PNG / TGA reader (`tr_image_load_rgba8`): 8-bit L / LA / RGB / RGBA expand exactly as
    `into_rgba8` does (L -> l,l,l,255; LA -> l,l,l,a; RGB -> a = 255); no gamma is applied."""
    import ctypes as C

    ffi, lib = _native()
    h = C.c_void_p()
    rc = lib.tr_image_load_rgba8(os.fsencode(path), 1 if flip else 0, C.byref(h))
    if rc != ffi.TR_OK:
        raise ffi.TrError(f"tr_image_load_rgba8({path}): {ffi.TR_ERR.get(rc, rc)} {lib.tr_load_error().decode(errors='replace')}")
    try:
        w, hh = C.c_int(), C.c_int()
        lib.tr_image_size(h, C.byref(w), C.byref(hh))
        data = np.ctypeslib.as_array(lib.tr_image_data(h), shape=(hh.value * w.value * 4,))
        return data.reshape(hh.value, w.value, 4).copy()
    finally:
        lib.tr_image_free(h)


def write_rgba_img_to_file(path, img):
    """`_write_nd_rgba_img_to_file` (src/helpers/mod.rs:114-120) for an (H, W, 4) uint8 image, row 0 = top."""
    ffi, lib = _native()
    img8 = np.ascontiguousarray(img, dtype=np.uint8)
    if img8.ndim != 3 or img8.shape[2] != 4:
        raise ValueError("image must be (H, W, 4) uint8")
    rc = lib.tr_image_write_png(os.fsencode(path), img8.ctypes.data, img8.shape[1], img8.shape[0])
    if rc != ffi.TR_OK:
        raise ffi.TrError(f"tr_image_write_png({path}): {ffi.TR_ERR.get(rc, rc)} {lib.tr_load_error().decode(errors='replace')}")


def get_drawable_z_buffer(z_buffer):
    """src/helpers/mod.rs:336-350: 1 / max(z, 0) where that is positive, else 0 (same shape)."""
    z = np.asarray(z_buffer, dtype=np.float64)
    val = np.fmax(z, 0.0)  # f64::max: NaN -> 0
    out = np.zeros_like(val)
    pos = val > 0.0
    out[pos] = 1.0 / val[pos]
    return out


def flip_vertically(img):
    """src/helpers/mod.rs:291-302 on an (H, W, C) image: row y <- row H-1-y."""
    return np.ascontiguousarray(img[::-1])


def get_rando_mesh_object():
    """The one-triangle mesh of src/helpers/mod.rs:558-608 (UVs, no normals)."""
    return Object(
        [[0.0, -0.5, 0.0], [0.5, 0.5, 0.0], [-0.5, 0.5, 0.0]],
        [[0.5, 0.0, 0.0], [1.0, 1.0, 0.0], [0.0, 1.0, 0.0]],
        np.zeros((0, 3)),
        [[0, 0, NONE_INDEX, 1, 1, NONE_INDEX, 2, 2, NONE_INDEX]],
    )


ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
