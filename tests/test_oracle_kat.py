"""Known-answer tests that pin the oracle (CPU restatement) to the reference source.

The reference has no tests or golden vectors, so each case below is
derived by hand from the cited Rust lines."""
import ctypes as C
import math

import numpy as np

from tinyrenderer_amd import helpers as H
from tinyrenderer_amd import scenes
from tinyrenderer_amd.model_renderer import BuiltinShader, MeshComponent


def _seg(oracle, p1, p2, res=-1, cap=64):
    lib = oracle.load()
    out = (C.c_double * (3 * cap))()
    a = (C.c_double * 3)(*p1)
    b = (C.c_double * 3)(*p2)
    n = lib.ora_segment_points(a, b, res, out, cap)
    return n, np.array(out[: 3 * max(0, min(n, cap))]).reshape(-1, 3)


def test_kat1_segment_default_resolution(oracle):
    # segment_3d/mod.rs:41-48,103-115: (0,0,0)->(4,0,0): res = ceil(4)+1 = 5, 6 points at x = i*0.2*4
    n, pts = _seg(oracle, (0, 0, 0), (4, 0, 0))
    assert n == 6
    assert pts[:, 0].tolist() == [i * (1.0 / 5) * 4.0 for i in range(6)]
    assert [int(math.floor(abs(x) + 0.5)) for x in pts[:, 0]] == [0, 1, 2, 2, 3, 4]
    # tiny segment: resolution floor of 2 -> 3 points
    n, pts = _seg(oracle, (1, 1, 0), (1.2, 1.1, 5))
    assert n == 3 and pts[1].tolist() == [1 + 0.5 * (1.2 - 1), 1 + 0.5 * (1.1 - 1), 0 + 0.5 * 5]
    # explicit resolution below 2 is an Err (segment_3d/mod.rs:50-55)
    n, _ = _seg(oracle, (0, 0, 0), (1, 0, 0), res=1)
    assert n == -1


def test_kat2_unit_parameter_never_exceeds_one(oracle):
    # segment_3d/mod.rs:20-25,110-111: t = n*(1/n) must stay in [0,1] or the iterator panics.
    # Checked for every resolution the GPU key can hold (i <= 131070, j <= 262143) and beyond.
    assert oracle.load().ora_check_unit_param(2, 1 << 20) == 0


def test_kat3_barycentric(oracle):
    # helpers/mod.rs:326-333
    lib = oracle.load()
    p1, p2, p3 = (0.0, 0.0), (4.0, 0.0), (0.0, 4.0)

    def bary(pt):
        out = (C.c_double * 3)()
        lib.ora_barycentric((C.c_double * 2)(*p1), (C.c_double * 2)(*p2), (C.c_double * 2)(*p3),
                            (C.c_double * 2)(*pt), out)
        return list(out)

    assert bary(p1) == [1.0, 0.0, 0.0]
    assert bary(p2) == [0.0, 1.0, 0.0]
    assert bary(p3) == [0.0, 0.0, 1.0]
    c = bary((4.0 / 3.0, 4.0 / 3.0))
    assert all(abs(v - 1.0 / 3.0) < 1e-15 for v in c)


def test_kat4_sample_nd_img(oracle):
    # helpers/mod.rs:253-278 on a 4x3 image with distinct texels
    Wt, Ht = 4, 3
    img = np.zeros((Ht, Wt, 4), dtype=np.uint8)
    for y in range(Ht):
        for x in range(Wt):
            img[y, x] = (10 * x + y, 100 + x, 200 - y, 255)
    im = oracle.OracleImage(img, flip=False)  # nd_img[x][y] = img[y][x]
    rc, v = im.sample(1.0, 1.0)  # integer coords inside: exactly the texel
    assert rc == 0 and v == [float(c) for c in img[1, 1]]
    rc, v = im.sample(1.5, 1.0)
    assert rc == 0 and v[0] == 0.5 * img[1, 1, 0] + 0.5 * img[1, 2, 0]
    rc, v = im.sample(float(Wt - 1), 1.0)  # x == W-1: x1 == x2 -> all weights 0 -> 0
    assert rc == 0 and v == [0.0, 0.0, 0.0, 0.0]
    rc, v = im.sample(-1e-9, 0.0)  # floor = -1 as f64, index saturates to 0: two unit weights
    assert rc == 0 and v[0] == img[0, 0, 0] * ((1.0 - -1e-9) * 1.0) + img[0, 1, 0] * ((-1e-9 - -1.0) * 1.0)
    rc, _ = im.sample(float(Wt), 0.0)  # index W: the reference panics
    assert rc == 1
    # flip_vertically: helpers/mod.rs:291-302
    imf = oracle.OracleImage(img, flip=True)
    rc, v = imf.sample(2.0, 0.0)
    assert v == [float(c) for c in img[Ht - 1, 2]]


def test_kat5_perspective_matrix():
    # helpers/mod.rs:548-555 with the reference's near=10, far=1
    m = H.make_perspective_matrix(10.0, 1.0, 0.35 * math.pi, 1.0)
    assert m[2, 2] == -1.0 * (1.0 + 10.0) / (1.0 - 10.0) and abs(m[2, 2] - 11.0 / 9.0) < 1e-15
    assert m[2, 3] == (-2.0 * 1.0 * 10.0) / (1.0 - 10.0) and abs(m[2, 3] - 20.0 / 9.0) < 1e-15
    assert m[3, 2] == -1.0 and m[3, 3] == 0.0
    t = 10.0 * math.tan(0.35 * math.pi / 2.0)
    assert m[0, 0] == 10.0 / t and m[1, 1] == 10.0 / t
    # nearer points get larger NDC z, i.e. smaller dist = -z
    near = H.mat4_mul_vec4(m, (0, 0, -1.0, 1.0))
    far = H.mat4_mul_vec4(m, (0, 0, -3.0, 1.0))
    assert near[2] / near[3] > far[2] / far[3]


def _rando_draw(W, Hh):
    obj = H.get_rando_mesh_object()
    tex = np.full((8, 8, 4), 255, dtype=np.uint8)
    tex[..., 0] = np.arange(8, dtype=np.uint8)[None, :] * 30
    nm = np.zeros((8, 8, 4), dtype=np.uint8)
    nm[..., :] = (128, 128, 255, 255)
    mc = MeshComponent(obj, tex, nm)
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.0, 2.0))
    return mc, sh


def test_kat6_rando_triangle_bruteforce(oracle):
    """get_rando_mesh_component (helpers/mod.rs:558-608) with identity matrices: coverage and z
    recomputed by a direct Python transcription of SURVEY.md 7.1 (independent of the C++)."""
    W = Hh = 24
    mc, sh = _rando_draw(W, Hh)
    st = oracle.OracleState(W, Hh)
    st.clear_screen()
    rc = st.render_mesh_component(mc.mesh, sh, oracle.OracleImage(mc.texture[::-1]), oracle.OracleImage(mc.normal_map[::-1]))
    assert rc == 0
    z = st.z_buffer()[..., 0]

    # brute force: viewport points, y-desc stable sort, split, two halves
    sx = (W - 1) / 2.0
    sy = (Hh - 1) / 2.0
    ndc = [(0.0, -0.5, 0.0), (0.5, 0.5, 0.0), (-0.5, 0.5, 0.0)]
    vp = [((sx * x + 0.0 * y) + 0.0 * zz + sx * 1.0, (0.0 * x + sy * y) + 0.0 * zz + sy * 1.0, zz) for x, y, zz in ndc]
    S = sorted(vp, key=lambda p: -p[1])  # python's sort is stable, like the reference's
    tt = (S[1][1] - S[0][1]) / (S[2][1] - S[0][1])
    split = tuple(S[0][d] + tt * (S[2][d] - S[0][d]) for d in range(3))
    zb = np.full((W, Hh), -np.inf)

    def rnd(v):
        return int(math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1))

    def seg(p1, p2, t):
        return tuple(p1[d] + t * (p2[d] - p1[d]) for d in range(3))

    n_samples = 0
    for a1, a2, b2 in ((S[0], S[1], split), (S[2], S[1], split)):
        res = max(2, math.ceil(max(abs(a2[0] - a1[0]), abs(a2[1] - a1[1]), abs(b2[0] - a1[0]), abs(b2[1] - a1[1]))))
        for i in range(res + 1):
            t = i * (1.0 / res)
            A, B = seg(a1, a2, t), seg(a1, b2, t)
            hres = max(2, math.ceil(max(abs(B[0] - A[0]), abs(B[1] - A[1]))) + 1)
            for j in range(hres + 1):
                p = seg(A, B, j * (1.0 / hres))
                n_samples += 1
                x, y = rnd(p[0]), rnd(p[1])
                if 0 <= x <= W - 1 and 0 <= y <= Hh - 1 and (zb[x, y] == -np.inf or zb[x, y] > -p[2]):
                    zb[x, y] = -p[2]
    assert st.counters()["samples"] == n_samples
    assert np.array_equal(np.isfinite(z), np.isfinite(zb))
    assert np.array_equal(z[np.isfinite(z)], zb[np.isfinite(zb)])
    # no normals in the mesh -> face-normal fallback (mod.rs:245-258); shading ran and wrote colour
    fb = st.frame_buffer()
    assert (fb[np.isfinite(z)][:, 3] == 255.0).all() and (fb[~np.isfinite(z)] == np.array(H.BLACK)).all()


def test_kat7_horizontal_degenerate_triangle_writes_nan_at_origin(oracle):
    """S0.y == S2.y: the split parameter is 0/0 = NaN, passes the range check
    (segment_3d/mod.rs:20), every sample is NaN, rounds to pixel (0,0) with NaN depth, and a NaN
    that lands on a clear pixel is stored and never replaced (mod.rs:388-390)."""
    W = Hh = 16
    obj = H.Object([[-0.5, 0.25, 0.0], [0.5, 0.25, 0.0], [0.0, 0.25, 0.0]], [[0, 0, 0], [1, 0, 0], [0.5, 1, 0]],
                   np.zeros((0, 3)), [[0, 0, H.NONE_INDEX, 1, 1, H.NONE_INDEX, 2, 2, H.NONE_INDEX]])
    mc, sh = _rando_draw(W, Hh)
    st = oracle.OracleState(W, Hh)
    st.clear_screen()
    rc = st.render_mesh_component(obj, sh, oracle.OracleImage(mc.texture[::-1]), oracle.OracleImage(mc.normal_map[::-1]))
    assert rc == 0
    z = st.z_buffer()[..., 0]
    assert np.isnan(z[0, 0])
    assert np.isinf(z.reshape(-1)[1:]).all()
    # the NaN fragment shades to white: d = max(NaN,0) = 0, 1 + 0*1*NaN = NaN, NaN.min(255) = 255
    assert st.frame_buffer()[0, 0].tolist() == [255.0, 255.0, 255.0, 255.0]
    # ... and a later valid triangle cannot take the pixel back
    mc2, sh2 = _rando_draw(W, Hh)
    big = H.Object([[-1, -1, 0.5], [1, -1, 0.5], [-1, 1, 0.5]], np.zeros((0, 3)), np.zeros((0, 3)),
                   [[0, H.NONE_INDEX, H.NONE_INDEX, 1, H.NONE_INDEX, H.NONE_INDEX, 2, H.NONE_INDEX, H.NONE_INDEX]])
    rc = st.render_mesh_component(big, sh2, oracle.OracleImage(mc.texture[::-1]), oracle.OracleImage(mc.normal_map[::-1]))
    assert rc == 0
    z = st.z_buffer()[..., 0]
    assert np.isnan(z[0, 0]) and np.isfinite(z[1, 1])


def test_counts_match_survey(oracle):
    """Workload sizes of BASELINE.md section 2, counted there by an independent throw-away script."""
    mc = scenes.load_component(scenes.AFRICAN_HEAD)
    mc.transform.set_rotation((0.025, 0.0, 0.0))
    W, Hh = 853, 480
    sh = scenes.shader_for(mc, W, Hh)
    st = oracle.OracleState(W, Hh)
    st.clear_screen()
    assert st.render_mesh_component(mc.mesh, sh, oracle.OracleImage(mc.texture[::-1]), oracle.OracleImage(mc.normal_map[::-1])) == 0
    c = st.counters()
    assert (c["samples"], c["scanlines"], c["culled"]) == (469508, 61088, 0)
    assert int(np.isfinite(st.z_buffer()).sum()) == 76368
