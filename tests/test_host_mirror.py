"""Host-side mirror of the reference's helpers / model_renderer surface (no GPU needed)."""
import math
import os

import numpy as np

from tinyrenderer_amd import helpers as H
from tinyrenderer_amd import scenes
from tinyrenderer_amd.model_renderer import BuiltinShader, MeshComponent, Transform


def test_mat4_mul_association_matches_oracle(oracle):
    import ctypes as C

    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((4, 4)), rng.standard_normal((4, 4))
    out = (C.c_double * 16)()
    oracle.load().ora_mat4_mul((C.c_double * 16)(*a.T.reshape(16)), (C.c_double * 16)(*b.T.reshape(16)), out)
    got = H.mat4_mul(a, b)
    assert np.array_equal(np.array(out).reshape(4, 4).T, got)  # bit-identical, not just close
    assert np.allclose(got, a @ b)


def test_transform_semantics():
    # model_renderer/mod.rs:118-147
    t = Transform()
    t.set_rotation((0.3, -0.2, 0.1))
    t.set_scale((2.0, 3.0, 4.0))
    want = H.mat4_mul(H.mat4_mul(H.make_translation_matrix((0, 0, 0)), H.make_rotation_matrix(0.3, -0.2, 0.1)),
                      H.make_scale_matrix((2.0, 3.0, 4.0)))
    assert np.array_equal(t.matrix(), want)
    t.set_position((1.0, 2.0, 3.0))  # pokes the last column only
    want[0, 3], want[1, 3], want[2, 3] = 1.0, 2.0, 3.0
    assert np.array_equal(t.matrix(), want)
    t.set_rotation((0.0, 0.0, 0.0))  # resync now includes the stored position
    assert np.array_equal(t.matrix()[:3, 3], [1.0, 2.0, 3.0])


def test_local_to_world_child_first_fold():
    # model_renderer/mod.rs:67-84: identity * self * parent
    obj = H.get_rando_mesh_object()
    parent = MeshComponent(obj, None, None)
    child = MeshComponent(obj, None, None, parent=parent)
    parent.transform.set_rotation((0.4, 0.0, 0.0))
    child.transform.set_position((0.1, 0.2, 0.3))
    want = H.mat4_mul(H.mat4_mul(np.eye(4), child.transform.matrix()), parent.transform.matrix())
    assert np.array_equal(child.local_to_world_matrix(), want)


def test_default_camera_is_a_translation():
    # main.rs:62-70: eye (0,0,2) looking at the origin -> view = translate(0,0,-2)
    eye, view = scenes.camera()
    want = H.make_translation_matrix((0.0, 0.0, -2.0))
    assert eye == (0.0, 0.0, 2.0) and np.array_equal(view, want)


def test_look_at_and_inverse_general():
    eye = scenes.orbit_eye(517, 4096)
    view = H.look_at_rh(eye, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    assert np.allclose(view[:3, :3] @ view[:3, :3].T, np.eye(3), atol=1e-14)
    assert np.allclose(view @ np.array([*eye, 1.0]), [0, 0, 0, 1], atol=1e-14)
    inv = H.try_inverse4(view)
    assert np.allclose(inv, np.linalg.inv(view), atol=1e-13)
    assert H.try_inverse4(np.zeros((4, 4))) is None
    sh = BuiltinShader(scenes.perspective(16, 9), view, eye)
    assert np.allclose(sh.normal_matrix, np.linalg.inv(view.T), atol=1e-13)


def test_obj_loader_counts_and_indices():
    # helpers/mod.rs:352-355; counts from the files' own comments
    head = H.load_model(os.path.join(H.ASSET_DIR, scenes.AFRICAN_HEAD[0]))
    assert (len(head.vertices), len(head.tex_vertices), len(head.normals), len(head.triangles)) == (1258, 1339, 1258, 2492)
    assert head.triangles[0].tolist() == [23, 0, 23, 24, 1, 24, 25, 2, 25]  # "f 24/1/24 25/2/25 26/3/26"
    assert head.vertices[0].tolist() == [-0.000581696, -0.734665, -0.623267]
    assert head.tex_vertices[0].tolist() == [0.532, 0.923, 0.0]
    diablo = H.load_model(os.path.join(H.ASSET_DIR, scenes.DIABLO[0]))
    assert (len(diablo.vertices), len(diablo.tex_vertices), len(diablo.normals), len(diablo.triangles)) == (2519, 3263, 2519, 5022)


def test_obj_parser_variants():
    text = """
# comment
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0.5 0.25
vn 0 0 1
f 1 2 3 4
f 1/1 2/1 3/1
f 1//1 2//1 -1//1
o second
v 9 9 9
f 1 2 5
"""
    o = H.parse_obj(text)
    N = H.NONE_INDEX
    assert len(o.vertices) == 4  # the second object is dropped
    assert o.triangles.tolist() == [
        [0, N, N, 1, N, N, 2, N, N], [0, N, N, 2, N, N, 3, N, N],  # quad -> fan
        [0, 0, N, 1, 0, N, 2, 0, N],
        [0, N, 0, 1, N, 0, 3, N, 0]]
    assert o.tex_vertices.tolist() == [[0.5, 0.25, 0.0]]


def test_image_loader_and_flip():
    p = os.path.join(H.ASSET_DIR, scenes.AFRICAN_HEAD_EYE_INNER[1])
    img = H.load_rgba_img_from_file(p)
    assert img.shape == (256, 256, 4) and img.dtype == np.uint8 and (img[..., 3] == 255).all()
    f = H.flip_vertically(img)
    assert np.array_equal(f[0], img[-1]) and np.array_equal(f[-1], img[0])
