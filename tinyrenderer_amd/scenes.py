"""The reference's scene parameters (src/main.rs:12-43,57-116) as reusable builders.

These are the literals of `main`: camera at (0,0,2) looking at the origin, perspective
(near=10, far=1, fov=0.35*pi, aspect=W/H), head rotation 0.025*time about the first euler
angle, eye mesh parented to the head.  BASELINE.json's configs re-parameterise them.
"""
import math
import os

from . import helpers as H
from .model_renderer import BuiltinShader, MeshComponent

AFRICAN_HEAD = ("african_head/african_head.obj", "african_head/african_head_diffuse.png",
                "african_head/african_head_nm_tangent.png")
AFRICAN_HEAD_EYE_INNER = ("african_head/african_head_eye_inner.obj",
                          "african_head/african_head_eye_inner_diffuse.png",
                          "african_head/african_head_eye_inner_nm_tangent.png")
DIABLO = ("diablo_guy/diablo3_pose.obj", "diablo_guy/diablo3_pose_diffuse.png",
          "diablo_guy/diablo3_pose_nm_tangent.png")
# specular maps (grey: the Phong exponent) for the EXTENSION shaders; the reference loads none of them
SPECULAR = {"head": "african_head/african_head_spec.png", "diablo": "diablo_guy/diablo3_pose_spec.png"}


def load_specular(name, asset_dir=None):
    d = asset_dir or H.ASSET_DIR
    return H.flip_vertically(H.load_rgba_img_from_file(os.path.join(d, SPECULAR[name])))


def load_component(triple, asset_dir=None):
    d = asset_dir or H.ASSET_DIR
    return MeshComponent.from_paths(*(os.path.join(d, p) for p in triple))


def camera(eye=(0.0, 0.0, 2.0)):
    """main.rs:62-70: camera_pos and the look-at (view) matrix."""
    return tuple(eye), H.look_at_rh(eye, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))


def orbit_eye(k, n_views, radius=2.0):
    """eye_k = 2*(sin t, 0, cos t), t = 2*pi*k/n."""
    t = 2.0 * math.pi * k / n_views
    return (radius * math.sin(t), 0.0, radius * math.cos(t))


def perspective(frame_width, frame_height):
    """main.rs:110-116."""
    return H.make_perspective_matrix(10.0, 1.0, 0.35 * math.pi, frame_width / frame_height)


def shader_for(mesh_component, frame_width, frame_height, eye=(0.0, 0.0, 2.0), use_normal_map=True):
    """main.rs:118-142: model_view = view * local_to_world; normal matrix = (MV^T)^-1."""
    cam_pos, view = camera(eye)
    mv = H.mat4_mul(view, mesh_component.local_to_world_matrix())
    return BuiltinShader(perspective(frame_width, frame_height), mv, cam_pos, use_normal_map=use_normal_map)


def subdivide(obj, levels):
    """split every triangle by its edge midpoints `levels` times
    (4**levels times the triangles); positions, normals and uvw of a midpoint are (a + b) * 0.5.
    Midpoints are not shared between neighbouring triangles (the renderer never shares vertices)."""
    import numpy as np

    v, t, n, tri = obj.vertices, obj.tex_vertices, obj.normals, obj.triangles.astype(np.int64)
    for _ in range(levels):
        T = tri.shape[0]
        out = np.empty((4 * T, 9), dtype=np.int64)
        new_attrs = []
        for k, arr in enumerate((v, t, n)):  # index column k of each corner: v, vt, vn
            idx = tri[:, [k, 3 + k, 6 + k]]  # (T, 3) corner indices a, b, c
            none = idx[:, 0] == H.NONE_INDEX
            if arr.shape[0] == 0 or none.all():
                new_attrs.append(arr)
                for c in range(4):
                    out[c::4, [k, 3 + k, 6 + k]] = H.NONE_INDEX
                continue
            safe = np.where(idx == H.NONE_INDEX, 0, idx)
            a, b, c_ = arr[safe[:, 0]], arr[safe[:, 1]], arr[safe[:, 2]]
            mab, mbc, mca = (a + b) * 0.5, (b + c_) * 0.5, (c_ + a) * 0.5
            base = arr.shape[0]
            arr2 = np.concatenate([arr, mab, mbc, mca], axis=0)
            iab = base + np.arange(T)
            ibc = base + T + np.arange(T)
            ica = base + 2 * T + np.arange(T)
            ia, ib, ic = idx[:, 0], idx[:, 1], idx[:, 2]
            corners = [(ia, iab, ica), (iab, ib, ibc), (ica, ibc, ic), (iab, ibc, ica)]
            for c, (p, q, r) in enumerate(corners):
                out[c::4, k] = np.where(none, H.NONE_INDEX, p)
                out[c::4, 3 + k] = np.where(none, H.NONE_INDEX, q)
                out[c::4, 6 + k] = np.where(none, H.NONE_INDEX, r)
            new_attrs.append(arr2)
        v, t, n = new_attrs
        tri = out
    return H.Object(v, t, n, tri.astype(np.uint32), obj.name)
