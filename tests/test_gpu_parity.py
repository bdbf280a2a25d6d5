"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on the same
inputs.  Bars: depth, winner key and triangle id bit-exact; presented
RGBA8 within +-1 LSB (the f64 colours are expected to be bit-identical too and the count of
differing f64 channels is reported)."""
import numpy as np
import pytest

import scene_util as SU
from tinyrenderer_amd import scenes

pytestmark = pytest.mark.gpu


def _head_scene(W, H, rot=0.025, eye=(0.0, 0.0, 2.0), with_eye=False):
    head = SU.component("head")
    head.transform.set_rotation((rot, 0.0, 0.0))
    draws = [(head, scenes.shader_for(head, W, H, eye=eye))]
    if with_eye:
        eye_mc = SU.component("eye_inner")
        eye_mc.parent = head
        draws.append((eye_mc, scenes.shader_for(eye_mc, W, H, eye=eye)))
    return draws


MODES = [pytest.param(1, id="tile"), pytest.param(0, id="global"), pytest.param(2, id="rows")]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("W,H", [(64, 64), (128, 96), (200, 200), (853, 480), (333, 517)])
def test_head_small(oracle, W, H, mode):
    draws = _head_scene(W, H)
    o = SU.oracle_render(oracle, W, H, draws)
    g = SU.gpu_render(W, H, draws, mode=mode)
    info = SU.assert_parity(o, g)
    assert g["counters"]["fragments"] == o["counters"]["fragments"]
    if mode != 1:  # the tile raster does not enumerate (or count) out-of-bounds samples
        assert g["counters"]["samples"] == o["counters"]["samples"]
        assert g["counters"]["scanlines"] == o["counters"]["scanlines"]
    assert info["f64_colour_mismatches"] == 0


@pytest.mark.parametrize("mode", MODES)
def test_config1_head_800(oracle, mode):
    """BASELINE.json configs[0]: african_head 800x800."""
    draws = _head_scene(800, 800)
    o = SU.oracle_render(oracle, 800, 800, draws)
    g = SU.gpu_render(800, 800, draws, mode=mode)
    info = SU.assert_parity(o, g)
    assert o["counters"]["samples"] == 1087161 and info["covered"] == 211671
    assert g["counters"]["covered_pixels"] == 211671
    assert info["f64_colour_mismatches"] == 0


@pytest.mark.parametrize("mode", MODES)
def test_config2_diablo_1024(oracle, mode):
    """BASELINE.json configs[1] (without the shadow pass, which has no reference)."""
    mc = SU.component("diablo")
    mc.transform.set_rotation((0.025, 0.0, 0.0))
    draws = [(mc, scenes.shader_for(mc, 1024, 1024))]
    o = SU.oracle_render(oracle, 1024, 1024, draws)
    g = SU.gpu_render(1024, 1024, draws, mode=mode)
    info = SU.assert_parity(o, g)
    assert o["counters"]["samples"] == 1294206 and info["covered"] == 173184
    assert info["f64_colour_mismatches"] == 0


@pytest.mark.parametrize("mode", MODES)
def test_two_draws_head_and_eye(oracle, mode):
    """The reference's own frame: head then eye_inner parented to it (main.rs:272-273), 853x480."""
    draws = _head_scene(853, 480, rot=0.05, with_eye=True)
    o = SU.oracle_render(oracle, 853, 480, draws)
    g = SU.gpu_render(853, 480, draws, mode=mode)
    SU.assert_parity(o, g)
    n_head = draws[0][0].mesh.triangles.shape[0]
    assert (g["seq"] >= n_head).any(), "no eye pixel won: the second draw was not exercised"


def test_orbit_views(oracle):
    """A few views of config 5's camera orbit at reduced size."""
    mc = SU.component("diablo")
    mc.transform.set_rotation((0.0, 0.0, 0.0))
    for k in (0, 517, 2048, 3333):
        eye = scenes.orbit_eye(k, 4096)
        draws = [(mc, scenes.shader_for(mc, 320, 180, eye=eye))]
        o = SU.oracle_render(oracle, 320, 180, draws)
        g = SU.gpu_render(320, 180, draws)
        SU.assert_parity(o, g)


def test_determinism():
    """Race detector stand-in: repeated renders hash identically."""
    draws = _head_scene(640, 360)
    ref = None
    for _ in range(5):
        g = SU.gpu_render(640, 360, draws)
        h = (g["z"].tobytes(), g["key"].tobytes(), g["present"].tobytes())
        if ref is None:
            ref = h
        assert h == ref


def _synthetic_component(obj, seed=3):
    import numpy as np
    from tinyrenderer_amd.model_renderer import MeshComponent

    rng = np.random.default_rng(seed)
    tex = rng.integers(0, 256, size=(32, 48, 4), dtype=np.uint8)
    nm = rng.integers(0, 256, size=(16, 16, 4), dtype=np.uint8)
    return MeshComponent(obj, tex, nm)


@pytest.mark.parametrize("nm_hw,tex_hw", [((2, 2), (2, 2)), ((3, 5), (3, 5)), ((1, 1), (1, 1)), ((5, 1), (5, 1)), ((1, 6), (1, 6)),
                                          ((64, 64), (64, 64)), ((7, 7), (4, 9)), ((2, 2), (1, 1))])
def test_texture_map_sizes_paired_and_not(oracle, nm_hw, tex_hw):
    """`sample_nd_img` (helpers/mod.rs:253-278) on maps of every awkward size, through both texel paths of the fused shade
    kernel: a normal map and a diffuse map of ONE size share indices and weights and are fetched together (width >= 2),
    maps of different sizes -- and one-texel-wide ones -- are fetched one after the other.  The head's uv reach 0 and 1
    exactly at some corners, i.e. the last column / row and their clamped neighbours."""
    from tinyrenderer_amd.model_renderer import MeshComponent

    rng = np.random.default_rng(nm_hw[0] * 131 + nm_hw[1] * 17 + tex_hw[0] * 7 + tex_hw[1])
    head = SU.component("head")
    mc = MeshComponent(head.mesh, rng.integers(0, 256, size=tex_hw + (4,), dtype=np.uint8),
                       rng.integers(0, 256, size=nm_hw + (4,), dtype=np.uint8))
    mc.transform.set_rotation((0.4, 0.1, 0.0))
    W, Hh = 240, 170
    draws = [(mc, scenes.shader_for(mc, W, Hh))]
    o = SU.oracle_render(oracle, W, Hh, draws)
    g = SU.gpu_render(W, Hh, draws)
    info = SU.assert_parity(o, g)
    assert info["f64_colour_mismatches"] == 0 and info["covered"] > 5000


def test_rando_triangle_face_normal_fallback(oracle):
    """helpers/mod.rs:558-608: one triangle, UVs, no normals -> face normal (mod.rs:245-258)."""
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader

    mc = _synthetic_component(H.get_rando_mesh_object())
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.3, 0.2, 2.0))
    for W, Hh in ((24, 24), (97, 61)):
        o = SU.oracle_render(oracle, W, Hh, [(mc, sh)])
        g = SU.gpu_render(W, Hh, [(mc, sh)])
        info = SU.assert_parity(o, g)
        assert info["f64_colour_mismatches"] == 0 and info["covered"] > 0


def test_mesh_without_uv_uses_vertex_colour(oracle):
    """No texture coordinates: albedo = interpolated vertex colour (main.rs:243-244), no TBN."""
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader

    N = H.NONE_INDEX
    obj = H.Object([[-0.8, -0.7, 0.1], [0.9, -0.2, -0.3], [0.1, 0.8, 0.4], [-0.3, 0.2, 0.9]],
                   np.zeros((0, 3)), [[0, 0, 1], [0, 1, 0], [1, 0, 0]],
                   [[0, N, 0, 1, N, 1, 2, N, 2], [0, N, 2, 2, N, 1, 3, N, N]])
    mc = _synthetic_component(obj)
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.5, 2.0))  # vertex colour is WHITE as in main.rs:170
    o = SU.oracle_render(oracle, 80, 60, [(mc, sh)])
    g = SU.gpu_render(80, 60, [(mc, sh)])
    info = SU.assert_parity(o, g)
    assert info["f64_colour_mismatches"] == 0 and info["covered"] > 0


def test_nan_depth_degenerate_triangle(oracle):
    """SURVEY.md 8(c) KAT 7: a horizontal degenerate triangle stores NaN at pixel (0,0) when it
    arrives on a clear pixel, and loses when a valid sample got there first."""
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader

    N = H.NONE_INDEX
    degenerate = H.Object([[-0.5, 0.25, 0.0], [0.5, 0.25, 0.0], [0.0, 0.25, 0.0]],
                          [[0, 0, 0], [1, 0, 0], [0.5, 1, 0]], np.zeros((0, 3)),
                          [[0, 0, N, 1, 1, N, 2, 2, N]])
    big = H.Object([[-1, -1, 0.5], [1, -1, 0.5], [-1, 1, 0.5]], np.zeros((0, 3)), np.zeros((0, 3)),
                   [[0, N, N, 1, N, N, 2, N, N]])
    d, b = _synthetic_component(degenerate), _synthetic_component(big)
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.0, 2.0))
    for order in ((d, b), (b, d)):
        for mode in (2, 1, 0):
            for W, Hh in ((16, 16), (50, 37)):
                draws = [(mc, sh) for mc in order]
                o = SU.oracle_render(oracle, W, Hh, draws)
                g = SU.gpu_render(W, Hh, draws, mode=mode)
                SU.assert_parity(o, g)
                assert np.isnan(o["z"][0, 0]) == (order[0] is d)


def test_non_finite_corners(oracle):
    """Corners the perspective divide sends to infinity or NaN (clip w = 0: a vertex on the camera plane).  The reference has
    no clipping: such a triangle is kept when another corner is inside (mod.rs:293-312), its viewport points are NaN
    (0 * inf in the viewport product's w: the one case in which that w is not exactly 1, and k_setup divides by it), its
    scanlines and samples follow the NaN rules (sort: partial_cmp -> Equal, `as` casts -> 0), and it may leave NaN at pixel (0, 0)."""
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader

    N = H.NONE_INDEX
    # clip = (x, y, z, z): w = z
    P = np.array([[1.0, 0, 0, 0], [0, 1.0, 0, 0], [0, 0, 1.0, 0], [0, 0, 1.0, 0]])
    wild = H.Object([[0.3, 0.2, 0.0], [0.1, 0.1, 0.5], [-0.2, 0.1, 0.5],        # first corner: (inf, inf, NaN)
                     [0.0, 0.0, 0.0], [0.2, -0.1, 0.5], [-0.1, -0.2, 0.5],      # first corner: 0 / 0 throughout
                     [0.2, 0.1, 0.5], [-0.3, -0.2, 0.0], [0.1, -0.3, 0.5]],     # middle corner: (-inf, -inf, NaN)
                    [[0.1, 0.2, 0], [0.9, 0.3, 0], [0.4, 0.95, 0]], [[0, 0, 1]],
                    [[0, 0, 0, 1, 1, 0, 2, 2, 0], [3, 0, 0, 4, 1, 0, 5, 2, 0], [6, 0, 0, 7, 1, 0, 8, 2, 0]])
    tame = H.Object([[-0.4, -0.4, 0.5], [0.4, -0.4, 0.5], [-0.4, 0.4, 0.5]], np.zeros((0, 3)), np.zeros((0, 3)),
                    [[0, N, N, 1, N, N, 2, N, N]])
    w_, t_ = _synthetic_component(wild), _synthetic_component(tame)
    sh = BuiltinShader(P, np.eye(4), (0.0, 0.0, 2.0))
    for order in ((w_, t_), (t_, w_), (w_,)):
        for mode in (2, 1, 0):
            for W, Hh in ((24, 24), (64, 48)):
                draws = [(mc, sh) for mc in order]
                o = SU.oracle_render(oracle, W, Hh, draws)
                g = SU.gpu_render(W, Hh, draws, mode=mode)
                SU.assert_parity(o, g)
    assert o["counters"]["culled"] == 0


@pytest.mark.parametrize("mode", MODES)
def test_nan_pixel_outside_the_frames_screen_box(oracle, mode):
    """The samples of a triangle whose corners share one y land on pixel (0, 0) wherever the triangle lies: the cheap resolve
    launches the shade blocks of the meshes' screen box only, and that box need not hold pixel (0, 0) -- which has to be
    painted all the same, and repainted once the triangle is gone (found by tests/manual/fuzz_frames.py)."""
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader, ModelRendererState, clear_screen, render_mesh_component

    N = H.NONE_INDEX
    degenerate = H.Object([[0.1, 0.25, 0.0], [0.6, 0.25, 0.0], [0.3, 0.25, 0.0]],
                          [[0, 0, 0], [1, 0, 0], [0.5, 1, 0]], np.zeros((0, 3)), [[0, 0, N, 1, 1, N, 2, 2, N]])
    small = H.Object([[0.5, 0.5, 0.2], [0.8, 0.5, 0.2], [0.5, 0.8, 0.2]], np.zeros((0, 3)), np.zeros((0, 3)), [[0, N, N, 1, N, N, 2, N, N]])
    d, s_ = _synthetic_component(degenerate), _synthetic_component(small)
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.0, 2.0))
    W, Hh = 200, 150
    with_nan = SU.oracle_render(oracle, W, Hh, [(d, sh), (s_, sh)])
    without = SU.oracle_render(oracle, W, Hh, [(s_, sh)])
    assert np.isnan(with_nan["z"][0, 0]) and np.isneginf(without["z"][0, 0])
    assert not np.array_equal(with_nan["present"][Hh - 1, 0], without["present"][Hh - 1, 0])  # (the NaN pixel is not black)
    st = ModelRendererState(W, Hh, raster_mode=mode)
    try:
        for _ in range(2):
            clear_screen(st)
            render_mesh_component(st, d, sh)
            render_mesh_component(st, s_, sh)
            st.resolve(_ffi.TR_OUT_RGBA8)
            SU.check_present(st, with_nan["present"], "frame with the NaN pixel")
            clear_screen(st)
            render_mesh_component(st, s_, sh)
            st.resolve(_ffi.TR_OUT_RGBA8)
            SU.check_present(st, without["present"], "the frame after it")
            assert st.status_flags() == 0
    finally:
        st.close()


@pytest.mark.parametrize("mode", MODES)
def test_nan_pixel_of_a_band_whose_rows_miss_the_meshes(oracle, mode):
    """The same pixel in a band context: the band owns row 0, the meshes' screen box misses the band's rows altogether (an
    empty launch box is not an empty launch), and afterwards a band that does not own row 0 has to read back a cleared
    pixel (0, 0) -- both found by tests/manual/fuzz_frames.py with FUZZ_BANDS=1."""
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader, ModelRendererState, clear_screen, render_mesh_component

    N = H.NONE_INDEX
    degenerate = H.Object([[0.1, 0.5, 0.0], [0.6, 0.5, 0.0], [0.3, 0.5, 0.0]],
                          [[0, 0, 0], [1, 0, 0], [0.5, 1, 0]], np.zeros((0, 3)), [[0, 0, N, 1, 1, N, 2, 2, N]])
    small = H.Object([[0.5, 0.6, 0.2], [0.8, 0.6, 0.2], [0.5, 0.9, 0.2]], np.zeros((0, 3)), np.zeros((0, 3)), [[0, N, N, 1, N, N, 2, N, N]])
    d, s_ = _synthetic_component(degenerate), _synthetic_component(small)
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.0, 2.0))
    W, Hh = 120, 160
    o = SU.oracle_render(oracle, W, Hh, [(d, sh), (s_, sh)])
    assert np.isnan(o["z"][0, 0])
    black = np.array([0, 0, 0, 255], dtype=np.uint8)
    st = ModelRendererState(W, Hh, raster_mode=mode)
    try:
        for _ in range(2):
            clear_screen(st)
            st.set_band(0, 48)                      # rows 0..47: the meshes lie above row 100
            render_mesh_component(st, d, sh)
            render_mesh_component(st, s_, sh)
            st.resolve(_ffi.TR_OUT_RGBA8)
            want = np.broadcast_to(black, o["present"].shape).copy()
            want[Hh - 48:] = o["present"][Hh - 48:]  # (presented rows are flipped)
            assert not np.array_equal(want[Hh - 1, 0], black)
            SU.check_present(st, want, "band [0, 48)")
            clear_screen(st)
            st.set_band(96, Hh)                      # does not own row 0: pixel (0, 0) reads back cleared
            render_mesh_component(st, d, sh)
            render_mesh_component(st, s_, sh)
            st.resolve(_ffi.TR_OUT_RGBA8)
            want = np.broadcast_to(black, o["present"].shape).copy()
            want[: Hh - 96] = o["present"][: Hh - 96]
            SU.check_present(st, want, "band [96, H)")
            assert st.status_flags() == 0
    finally:
        st.close()


def test_offscreen_and_culled(oracle):
    """Triangles partly / fully outside the frustum: per-corner cull (mod.rs:293-312) and the
    bounds test (mod.rs:380-386); no clipping exists in the reference."""
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader

    N = H.NONE_INDEX
    obj = H.Object([[-1.6, -0.2, 0.0], [0.4, 1.7, 0.2], [0.3, -0.4, -0.1],      # partly outside
                    [1.2, 1.3, 0.0], [1.9, 1.2, 0.0], [1.5, 1.8, 0.0],          # all outside: culled
                    [0.0, 0.0, 1.5], [0.5, 0.0, 0.5], [0.0, 0.5, 0.5]],          # one corner z-outside
                   [[0.1, 0.2, 0], [0.9, 0.3, 0], [0.4, 0.95, 0]], [[0, 0, 1]],
                   [[0, 0, 0, 1, 1, 0, 2, 2, 0], [3, 0, 0, 4, 1, 0, 5, 2, 0], [6, 0, 0, 7, 1, 0, 8, 2, 0]])
    mc = _synthetic_component(obj)
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.0, 2.0))
    o = SU.oracle_render(oracle, 120, 90, [(mc, sh)])
    assert o["counters"]["culled"] == 1
    for mode in (2, 1, 0):
        g = SU.gpu_render(120, 90, [(mc, sh)], mode=mode)
        SU.assert_parity(o, g)
        assert g["counters"]["fragments"] == o["counters"]["fragments"] < o["counters"]["samples"]
        if mode != 1:
            assert g["counters"]["samples"] == o["counters"]["samples"]


def test_empty_frame_and_empty_mesh(oracle):
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader

    empty = H.Object(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 9), dtype=np.uint32))
    mc = _synthetic_component(empty)
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.0, 2.0))
    for draws in ([], [(mc, sh)]):
        g = SU.gpu_render(33, 17, draws)
        assert np.isneginf(g["z"]).all() and (g["seq"] == -1).all()
        assert (g["present"] == np.array([0, 0, 0, 255], dtype=np.uint8)).all()
        assert (g["fb"] == np.array([0.0, 0.0, 0.0, 255.0])).all()


def _random_scene(rng, n_tris, W, H, spread=1.3):
    """Random triangle soup in NDC (identity matrices): slivers, tiny and huge triangles, shared
    vertices, some partly off-screen -- the shapes that stress the tile raster's range pruning."""
    from tinyrenderer_amd import helpers as H_
    from tinyrenderer_amd.model_renderer import BuiltinShader

    n_v = max(3, n_tris)
    v = rng.uniform(-spread, spread, size=(n_v, 3))
    v[:, 2] = rng.uniform(-0.9, 0.9, size=n_v)
    # a few exact duplicates / axis-aligned edges / pixel-centre coordinates
    v[: n_v // 8, 0] = np.round(v[: n_v // 8, 0] * 8) / 8
    v[n_v // 8: n_v // 4, 1] = v[0, 1]
    vt = rng.uniform(0.0, 1.0, size=(n_v, 3))
    vn = rng.normal(size=(n_v, 3))
    tri = np.zeros((n_tris, 9), dtype=np.uint32)
    for k in range(n_tris):
        scale = rng.choice([0.02, 0.1, 0.5, 3.0])
        a = rng.integers(0, n_v)
        near = np.argsort(np.abs(v[:, :2] - v[a, :2]).sum(axis=1))[: max(3, int(n_v * scale / 3.0))]
        b, c = rng.choice(near, 2, replace=False)
        tri[k] = [a, a, a, b, b, b, c, c, c]
    obj = H_.Object(v, vt, vn, tri)
    mc = _synthetic_component(obj, seed=int(rng.integers(1 << 30)))
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.2, -0.1, 2.0))
    return [(mc, sh)]


@pytest.mark.parametrize("seed,n_tris,W,H", [(1, 40, 97, 61), (2, 300, 160, 160), (3, 2000, 256, 144),
                                              (4, 12, 641, 367), (5, 800, 48, 300)])
def test_fuzz_random_triangles(oracle, seed, n_tris, W, H):
    rng = np.random.default_rng(seed)
    draws = _random_scene(rng, n_tris, W, H)
    o = SU.oracle_render(oracle, W, H, draws)
    for mode in (2, 1, 0):
        g = SU.gpu_render(W, H, draws, mode=mode)
        info = SU.assert_parity(o, g)
        assert g["counters"]["fragments"] == o["counters"]["fragments"]
        assert info["f64_colour_mismatches"] == 0 and info["covered"] > 0


def test_band_sharding_is_exact(oracle):
    """bands of tile rows rendered by separate contexts stack to the full frame."""
    W, Hh = 400, 300
    draws = _head_scene(W, Hh)
    full = SU.gpu_render(W, Hh, draws)
    o = SU.oracle_render(oracle, W, Hh, draws)
    SU.assert_parity(o, full)
    from tinyrenderer_amd import sharding

    for world in (2, 3):
        z = np.full_like(full["z"], -np.inf)
        key = np.full_like(full["key"], np.uint64(0xFFFFFFFFFFFFFFFF))
        present = np.zeros_like(full["present"])
        frags = 0
        for r in range(world):
            y_lo, y_hi = sharding.band_rows(Hh, r, world)
            g = SU.gpu_render(W, Hh, draws, band=(y_lo, y_hi))
            z[:, y_lo:y_hi] = g["z"][:, y_lo:y_hi]          # reference layout is (W, H)
            key[:, y_lo:y_hi] = g["key"][:, y_lo:y_hi]
            present[Hh - y_hi: Hh - y_lo] = g["present"][Hh - y_hi: Hh - y_lo]  # presented rows are flipped
            frags += g["counters"]["fragments"]
            outside = np.ones(Hh, dtype=bool)
            outside[y_lo:y_hi] = False
            assert np.isneginf(g["z"][:, outside]).all()
        assert np.array_equal(z.view(np.uint64), full["z"].view(np.uint64))
        assert np.array_equal(key, full["key"]) and np.array_equal(present, full["present"])
        assert frags == full["counters"]["fragments"]


def _assert_bands_stitch_to(o, W, Hh, draws, edges, mode):
    """Every band [edges[k], edges[k + 1]) rendered by itself; the stitched planes must be the oracle's whole frame."""
    z = np.full_like(o["z"], -np.inf)
    key = np.full_like(o["key"], np.uint64(0xFFFFFFFFFFFFFFFF))
    present = np.zeros_like(o["present"])
    frags = 0
    for y_lo, y_hi in zip(edges[:-1], edges[1:]):
        g = SU.gpu_render(W, Hh, draws, mode=mode, band=(y_lo, y_hi))
        assert g["flags"] == 0
        z[:, y_lo:y_hi] = g["z"][:, y_lo:y_hi]
        key[:, y_lo:y_hi] = g["key"][:, y_lo:y_hi]
        present[Hh - y_hi: Hh - y_lo] = g["present"][Hh - y_hi: Hh - y_lo]
        frags += g["counters"]["fragments"]
        outside = np.ones(Hh, dtype=bool)
        outside[y_lo:y_hi] = False
        assert np.isneginf(g["z"][:, outside]).all()
    same = (z.view(np.uint64) == o["z"].view(np.uint64)) | (np.isnan(z) & np.isnan(o["z"]))
    assert same.all(), f"{(~same).sum()} depth mismatches at {np.argwhere(~same)[:4].tolist()}"
    assert np.array_equal(key, o["key"]) and np.array_equal(present, o["present"])
    assert frags == o["counters"]["fragments"], (frags, o["counters"]["fragments"])


def test_bands_and_degenerate_triangles(oracle):
    """A triangle whose corners share one y has NaN in every sample (the split parameter is 0 / 0, mod.rs:435-441) and
    `as` casts send them all to pixel (0, 0): in a band-sharded frame it belongs to the band that owns row 0, wherever its
    corners lie, and to no other (found by tests/manual/fuzz_bands.py: the band test of the corners' rows used to drop it
    from every band).  The deterministic case, then three of the fuzz's scenes, in every raster mode."""
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader

    N = H.NONE_INDEX
    degenerate = H.Object([[-0.5, 0.25, 0.0], [0.5, 0.25, 0.0], [0.0, 0.25, 0.0]],
                          [[0, 0, 0], [1, 0, 0], [0.5, 1, 0]], np.zeros((0, 3)), [[0, 0, N, 1, 1, N, 2, 2, N]])
    big = H.Object([[-1, -1, 0.5], [1, -1, 0.5], [-1, 1, 0.5]], np.zeros((0, 3)), np.zeros((0, 3)), [[0, N, N, 1, N, N, 2, N, N]])
    d, b = _synthetic_component(degenerate), _synthetic_component(big)
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.0, 2.0))
    W, Hh = 50, 70
    for order in ((d, b), (b, d), (d,)):
        draws = [(mc, sh) for mc in order]
        o = SU.oracle_render(oracle, W, Hh, draws)
        assert np.isnan(o["z"][0, 0]) == (order[0] is d)
        for mode in (0, 1, 2):
            _assert_bands_stitch_to(o, W, Hh, draws, [0, 16, 48, Hh], mode)
    for seed in (5005, 5006, 5016):
        rng = np.random.default_rng(seed)
        n_tris = int(rng.choice([3, 17, 60, 250, 900]))
        W = int(rng.integers(8, 500))
        Hh = int(rng.integers(33, 400))
        draws = _random_scene(rng, n_tris, W, Hh, spread=float(rng.choice([0.6, 1.0, 1.3, 2.5])))
        o = SU.oracle_render(oracle, W, Hh, draws)
        rows = (Hh + 15) // 16
        edges = [0, 16 * (rows // 3), 16 * (2 * rows // 3), Hh]
        edges = sorted(set(min(e, Hh) for e in edges))
        for mode in (0, 1, 2):
            _assert_bands_stitch_to(o, W, Hh, draws, edges, mode)


def test_bands_of_a_large_draw_with_a_one_point_triangle(oracle):
    """Draws of >= 65 536 triangles drop whole clusters of 64 by their boxes before a band context looks at a triangle
    (k_cull_clusters).  A triangle of three identical positions is degenerate under every view -- its samples are NaN and
    land on pixel (0, 0), far from its cluster's box --, so its cluster is never dropped (k_mesh_bounds) and the band that owns
    row 0 gets the triangle."""
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import MeshComponent

    W, Hh = 640, 480
    head = SU.component("head")
    fine = scenes.subdivide(head.mesh, 3)
    tri = fine.triangles.copy()
    k = int(np.argmax(fine.vertices[tri[:, 0], 1]))  # a triangle at the top of the head: rows far from row 0
    tri[k, 3] = tri[k, 0]
    tri[k, 6] = tri[k, 0]
    mc = MeshComponent(H.Object(fine.vertices, fine.tex_vertices, fine.normals, tri, fine.name), head.texture, head.normal_map)
    mc.transform.set_rotation((0.025, 0.0, 0.0))
    draws = [(mc, scenes.shader_for(mc, W, Hh))]
    o = SU.oracle_render(oracle, W, Hh, draws)
    assert np.isnan(o["z"][0, 0]), "the one-point triangle's samples should own pixel (0, 0)"
    for mode in (0, 1):
        _assert_bands_stitch_to(o, W, Hh, draws, [0, 160, 320, Hh], mode)


@pytest.mark.parametrize("mode", MODES)
def test_frame_reuse_and_late_plane_requests(oracle, mode):
    """One context over several frames: the RGBA8-only resolve wipes the global raster's planes
    (free clear), a later request for z / keys re-runs the passes, draws after a resolve extend
    the frame, and a clear really clears."""
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

    W, Hh = 320, 200
    draws = _head_scene(W, Hh, with_eye=True)
    o2 = SU.oracle_render(oracle, W, Hh, draws)
    o1 = SU.oracle_render(oracle, W, Hh, draws[:1])
    st = ModelRendererState(W, Hh, raster_mode=mode)
    try:
        for _ in range(2):
            clear_screen(st)
            render_mesh_component(st, *draws[0])
            st.resolve(_ffi.TR_OUT_RGBA8)                      # fast path
            SU.check_present(st, o1["present"])
            z = st.z_buffer[..., 0]                            # late request for another plane
            assert np.array_equal(z.view(np.uint64), o1["z"].view(np.uint64))
            assert np.array_equal(st.winner_keys(), o1["key"])
            render_mesh_component(st, *draws[1])               # extend the same frame
            st.resolve(_ffi.TR_OUT_RGBA8)
            SU.check_present(st, o2["present"])
            assert np.array_equal(st.winner_keys(), o2["key"])
            assert np.array_equal(st.z_buffer[..., 0].view(np.uint64), o2["z"].view(np.uint64))
        clear_screen(st)
        st.resolve(_ffi.TR_OUT_RGBA8)
        assert (st.present() == np.array([0, 0, 0, 255], dtype=np.uint8)).all()
        assert np.isneginf(st.z_buffer).all()
    finally:
        st.close()


@pytest.mark.parametrize("mode", MODES)
def test_config4_subdivided_mesh_bands(oracle, mode):
    """BASELINE.json configs[3] at reduced scale: african_head subdivided 3x (159 488 triangles
    of a few pixels), rendered whole and as 2 tile-row bands that must stack to the same frame."""
    from tinyrenderer_amd import sharding
    from tinyrenderer_amd.model_renderer import MeshComponent

    W, Hh = 960, 544
    head = SU.component("head")
    fine = MeshComponent(scenes.subdivide(head.mesh, 3), head.texture, head.normal_map)
    assert fine.mesh.triangles.shape[0] == 2492 * 64
    fine.transform.set_rotation((0.025, 0.0, 0.0))
    draws = [(fine, scenes.shader_for(fine, W, Hh))]
    o = SU.oracle_render(oracle, W, Hh, draws)
    g = SU.gpu_render(W, Hh, draws, mode=mode)
    info = SU.assert_parity(o, g)
    assert info["f64_colour_mismatches"] == 0
    key = np.full_like(g["key"], np.uint64(0xFFFFFFFFFFFFFFFF))
    for r in range(2):
        y_lo, y_hi = sharding.band_rows(Hh, r, 2)
        b = SU.gpu_render(W, Hh, draws, mode=mode, band=(y_lo, y_hi))
        key[:, y_lo:y_hi] = b["key"][:, y_lo:y_hi]
    assert np.array_equal(key, g["key"])


def test_unfused_front_end_and_draw_sizes_around_the_fused_limit(oracle, monkeypatch):
    """TR_FUSED_FRONT=0: k_transform + k_setup<false> + the one-launch scan (k_scan_small, several rounds
    for the head subdivided once) instead of the fused launch; and, fused again, draws just below and
    above its limit of 64 workgroups (16 384 triangles), the latter through the three-kernel scan."""
    from tinyrenderer_amd.model_renderer import MeshComponent

    W, Hh = 400, 300
    head = SU.component("head")
    once = MeshComponent(scenes.subdivide(head.mesh, 1), head.texture, head.normal_map)  # 9 968 triangles
    once.transform.set_rotation((0.025, 0.0, 0.0))
    scenes_ = [_head_scene(W, Hh, with_eye=True), [(once, scenes.shader_for(once, W, Hh))]]
    refs = [SU.oracle_render(oracle, W, Hh, d) for d in scenes_]
    monkeypatch.setenv("TR_FUSED_FRONT", "0")
    for d, o in zip(scenes_, refs):
        g = SU.gpu_render(W, Hh, d, mode=0)
        info = SU.assert_parity(o, g)
        assert info["f64_colour_mismatches"] == 0
        for k in ("samples", "fragments", "scanlines"):
            assert g["counters"][k] == o["counters"][k], k
    monkeypatch.delenv("TR_FUSED_FRONT")
    # 16 384 and 16 385 triangles: a prefix of the twice-subdivided head (39 872 triangles)
    import copy

    twice = scenes.subdivide(head.mesh, 2)
    for n in (16384, 16385):
        part = copy.copy(twice)
        part.triangles = twice.triangles[:n]
        mc = MeshComponent(part, head.texture, head.normal_map)
        mc.transform.set_rotation((0.025, 0.0, 0.0))
        d = [(mc, scenes.shader_for(mc, W, Hh))]
        o = SU.oracle_render(oracle, W, Hh, d)
        g = SU.gpu_render(W, Hh, d, mode=0)
        info = SU.assert_parity(o, g)
        assert info["f64_colour_mismatches"] == 0
        for k in ("samples", "fragments", "scanlines"):
            assert g["counters"][k] == o["counters"][k], (n, k)


def test_scanline_record_overflow_falls_back(oracle, monkeypatch):
    """Global raster: slots beyond the record buffer (TR_SCAN_RECORDS) are rebuilt by every pass;
    results and counters do not depend on the capacity."""
    W, Hh = 400, 300
    draws = _head_scene(W, Hh, with_eye=True)
    o = SU.oracle_render(oracle, W, Hh, draws)
    for cap in ("0", "1000", "20011"):
        monkeypatch.setenv("TR_SCAN_RECORDS", cap)
        g = SU.gpu_render(W, Hh, draws, mode=0)
        info = SU.assert_parity(o, g)
        assert info["f64_colour_mismatches"] == 0
        for k in ("samples", "fragments", "scanlines"):
            assert g["counters"][k] == o["counters"][k], (cap, k)


def test_raster_workgroup_sizes(oracle, monkeypatch):
    """k_raster with 128 and 256 threads per chunk (chosen by the number of live contexts; pinned
    here with TR_RASTER_THREADS) and with a second context alive: same frame, same counters."""
    from tinyrenderer_amd.model_renderer import ModelRendererState

    W, Hh = 400, 300
    draws = _head_scene(W, Hh, with_eye=True)
    o = SU.oracle_render(oracle, W, Hh, draws)
    for threads in ("128", "256"):
        monkeypatch.setenv("TR_RASTER_THREADS", threads)
        g = SU.gpu_render(W, Hh, draws, mode=0)
        info = SU.assert_parity(o, g)
        assert info["f64_colour_mismatches"] == 0
        for k in ("samples", "fragments", "scanlines"):
            assert g["counters"][k] == o["counters"][k], (threads, k)
    monkeypatch.delenv("TR_RASTER_THREADS")
    other = ModelRendererState(64, 64)
    try:
        g = SU.gpu_render(W, Hh, draws, mode=0)
        info = SU.assert_parity(o, g)
        assert info["f64_colour_mismatches"] == 0
        for k in ("samples", "fragments", "scanlines"):
            assert g["counters"][k] == o["counters"][k], k
    finally:
        other.close()


def test_counters_over_an_extended_frame(oracle):
    """draw, fast resolve (wipes the planes), draw, resolve: the depth pass of the whole frame is
    redone, but every sample is counted once."""
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

    W, Hh = 320, 200
    draws = _head_scene(W, Hh, with_eye=True)
    o = SU.oracle_render(oracle, W, Hh, draws)
    st = ModelRendererState(W, Hh, raster_mode=0)
    try:
        clear_screen(st)
        render_mesh_component(st, *draws[0])
        st.resolve(_ffi.TR_OUT_RGBA8)
        render_mesh_component(st, *draws[1])
        st.resolve(_ffi.TR_OUT_RGBA8)
        c = st.counters()
        for k in ("samples", "fragments", "scanlines"):
            assert c[k] == o["counters"][k], k
        SU.check_present(st, o["present"])
    finally:
        st.close()


@pytest.mark.parametrize("mode", MODES)
def test_large_and_tiny_triangles_mixed(oracle, mode):
    """Triangles from a few pixels to most of a 4K-wide frame in one draw: exercises the slot
    padding that balances the global raster's chunks (slot_shift 0..5)."""
    from tinyrenderer_amd import helpers as H_
    from tinyrenderer_amd.model_renderer import BuiltinShader

    W, Hh = 2048, 256
    rng = np.random.default_rng(11)
    n_v = 60
    v = np.zeros((n_v, 3))
    v[:, 0] = rng.uniform(-0.98, 0.98, n_v)
    v[:, 1] = rng.uniform(-0.98, 0.98, n_v)
    v[:, 2] = rng.uniform(-0.9, 0.9, n_v)
    vt = rng.uniform(0.02, 0.98, (n_v, 3))
    vn = rng.normal(size=(n_v, 3))
    tris = []
    for k in range(24):  # big
        a, b, c = rng.choice(n_v, 3, replace=False)
        tris.append([a, a, a, b, b, b, c, c, c])
    order = np.argsort(v[:, 0])
    for k in range(n_v - 2):  # slivers / small
        a, b, c = order[k], order[k + 1], order[k + 2]
        tris.append([a, a, a, b, b, b, c, c, c])
    obj = H_.Object(v.reshape(-1), vt.reshape(-1), vn.reshape(-1), np.array(tris, dtype=np.uint32).reshape(-1))
    mc = _synthetic_component(obj, seed=5)
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.3, 0.2, 1.0))
    draws = [(mc, sh)]
    o = SU.oracle_render(oracle, W, Hh, draws)
    g = SU.gpu_render(W, Hh, draws, mode=mode)
    info = SU.assert_parity(o, g)
    assert info["f64_colour_mismatches"] == 0 and info["covered"] > 10000
    if mode != 1:
        for k in ("samples", "fragments", "scanlines"):
            assert g["counters"][k] == o["counters"][k], k


def test_background_blocks_are_repainted_between_frames(oracle):
    """The RGBA8-only resolve leaves blocks alone that hold no key and are already background
    colour: frames with different coverage on one context must still come out whole."""
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

    W, Hh = 400, 240
    head, eye_mc = SU.component("head"), SU.component("eye_inner")
    eye_mc.parent = None
    views = [[(head, scenes.shader_for(head, W, Hh, eye=(0.0, 0.0, 2.0)))],
             [(eye_mc, scenes.shader_for(eye_mc, W, Hh, eye=(1.5, 0.3, 4.0)))],
             [],
             [(head, scenes.shader_for(head, W, Hh, eye=(-2.0, 0.5, 3.0)))],
             [(eye_mc, scenes.shader_for(eye_mc, W, Hh, eye=(0.0, 0.0, 2.0)))]]
    st = ModelRendererState(W, Hh, raster_mode=0)
    try:
        for k, draws in enumerate(views):
            clear_screen(st)
            for d in draws:
                render_mesh_component(st, *d)
            st.resolve(_ffi.TR_OUT_RGBA8)
            want = SU.oracle_render(oracle, W, Hh, draws)["present"]
            SU.check_present(st, want, str(f"view {k}"))
            if k == 3:  # a late plane request in between must not confuse the block flags
                assert np.array_equal(st.winner_keys(), SU.oracle_render(oracle, W, Hh, draws)["key"])
                SU.check_present(st, want)
    finally:
        st.close()
        eye_mc.parent = head


def test_row_raster_grows_its_buffers(oracle, monkeypatch):
    """Row raster: a record buffer or bin list that is too small raises TR_FLAG_BIN_OVERFLOW on the
    device; the next synchronising call grows it and redoes the resolve."""
    W, Hh = 400, 300
    draws = _head_scene(W, Hh, with_eye=True)
    o = SU.oracle_render(oracle, W, Hh, draws)
    for env in ({"TR_SCAN_RECORDS": "1000"}, {"TR_ROW_REFS": "500"}, {"TR_SCAN_RECORDS": "1", "TR_ROW_REFS": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = SU.gpu_render(W, Hh, draws, mode=2)
        info = SU.assert_parity(o, g)
        assert info["f64_colour_mismatches"] == 0
        for k in ("samples", "fragments", "scanlines"):
            assert g["counters"][k] == o["counters"][k], (env, k)
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("mode,env", [(2, {"TR_ROW_REFS": "500"}), (2, {"TR_SCAN_RECORDS": "1000"}), (1, {"TR_TILE_PAIRS": "300"})])
def test_bin_overflow_is_settled_by_resolve_itself(oracle, monkeypatch, mode, env):
    """A GPU-resident loop never reads back: tr_resolve, consume tr_device_ptr_rgba8 on the stream, tr_clear.
    A bin list that overflows (tile / row raster) must therefore be grown and the frame redone inside
    tr_resolve, and nothing of the aborted attempt may reach the next, smaller frame."""
    import ctypes as C

    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    W, Hh = 400, 300
    hip = C.CDLL("libamdhip64.so.7")
    lib = _ffi.load()
    big = _head_scene(W, Hh, with_eye=True)
    eye_mc = SU.component("eye_inner")
    small = [(eye_mc, scenes.shader_for(eye_mc, W, Hh, eye=(0.3, 0.1, 1.2)))]
    st = ModelRendererState(W, Hh, raster_mode=mode)
    try:
        for draws in (big, small, big):
            clear_screen(st)
            for d in draws:
                render_mesh_component(st, *d)
            st.resolve(_ffi.TR_OUT_RGBA8)
            st.synchronize()
            # the device plane as a consumer on the stream would see it: no library read in between
            dev = C.c_void_p()
            assert lib.tr_device_ptr_rgba8(st._ctx, C.byref(dev)) == 0
            got = np.empty((Hh, W, 4), dtype=np.uint8)
            assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), dev, C.c_size_t(got.nbytes), 2) == 0  # device to host
            want = SU.oracle_render(oracle, W, Hh, draws)["present"][::-1]  # (the plane is not flipped)
            assert np.array_equal(got, want)
        assert st.status_flags() == 0
    finally:
        st.close()


def test_texture_destroyed_under_a_recorded_frame(oracle):
    """Advisor finding, round 1: tr_texture_destroy freed memory the frame's recorded draws still pointed at, and a
    late plane request shaded through it.  Now: what is resolved stays readable, a further resolve is refused,
    the next frame is unaffected."""
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

    W, Hh = 160, 120
    draws = _head_scene(W, Hh)
    o = SU.oracle_render(oracle, W, Hh, draws)
    st = ModelRendererState(W, Hh)
    try:
        clear_screen(st)
        render_mesh_component(st, *draws[0])
        st.resolve(_ffi.TR_OUT_RGBA8)
        tex = st._textures.pop(id(draws[0][0].texture))
        assert st._lib.tr_texture_destroy(st._ctx, tex) == 0
        SU.assert_image(st.present(), o["present"])       # resolved: still there
        with pytest.raises(_ffi.TrError, match="destroyed"):
            st.z_buffer                                          # would shade again through the freed texture
        clear_screen(st)                                         # a new frame (the texture is uploaded again)
        render_mesh_component(st, *draws[0])
        assert np.array_equal(st.z_buffer[..., 0].view(np.uint64), o["z"].view(np.uint64))
        SU.check_present(st, o["present"])
    finally:
        st.close()


def test_mid_triangle_path(oracle, monkeypatch):
    """k_mid (one wave per triangle half: triangles of at most 64 scanlines per half and a bounded sample count, taken by
    launches of 32 768 triangles or more whose triangles are small on the screen): forced onto every draw here
    (TR_MID_TRIS=1), with several sample caps so that meshes split between k_mid and the slot sequence.  Whole frames
    and counters, bands, two draws, a frame extended after a resolve, NaN depths, random triangle soups."""
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader, ModelRendererState, clear_screen, render_mesh_component

    monkeypatch.setenv("TR_MID_TRIS", "1")
    for cap in ("4096", "200", "40"):
        monkeypatch.setenv("TR_MID_SAMPLES", cap)
        for W, Hh in ((200, 160), (853, 480)):
            draws = _head_scene(W, Hh, with_eye=True)
            o = SU.oracle_render(oracle, W, Hh, draws)
            g = SU.gpu_render(W, Hh, draws, mode=0)
            info = SU.assert_parity(o, g)
            assert info["f64_colour_mismatches"] == 0
            for k in ("samples", "fragments", "scanlines"):
                assert g["counters"][k] == o["counters"][k], (cap, W, k)
            top = SU.gpu_render(W, Hh, draws, mode=0, band=(0, 64))
            assert np.array_equal(top["key"][:, :64], o["key"][:, :64]) and (top["key"][:, 64:] == np.uint64(2**64 - 1)).all()
    monkeypatch.setenv("TR_MID_SAMPLES", "4096")
    monkeypatch.setenv("TR_SMALL_DRAW", "1")  # with the small-triangle path as well: three paths in one frame
    W, Hh = 160, 120
    draws = _head_scene(W, Hh, with_eye=True)
    o2 = SU.oracle_render(oracle, W, Hh, draws)
    st = ModelRendererState(W, Hh, raster_mode=0)
    try:
        clear_screen(st)
        render_mesh_component(st, *draws[0])
        st.resolve(_ffi.TR_OUT_RGBA8)
        render_mesh_component(st, *draws[1])
        st.resolve(_ffi.TR_OUT_RGBA8)
        SU.check_present(st, o2["present"])
        assert np.array_equal(st.winner_keys(), o2["key"])
        c = st.counters()
        for k in ("samples", "fragments", "scanlines"):
            assert c[k] == o2["counters"][k], k
    finally:
        st.close()
    monkeypatch.delenv("TR_SMALL_DRAW")
    N = H.NONE_INDEX
    degenerate = H.Object([[-0.5, 0.25, 0.0], [0.5, 0.25, 0.0], [0.0, 0.25, 0.0]],
                          [[0, 0, 0], [1, 0, 0], [0.5, 1, 0]], np.zeros((0, 3)), [[0, 0, N, 1, 1, N, 2, 2, N]])
    big = H.Object([[-1, -1, 0.5], [1, -1, 0.5], [-1, 1, 0.5]], np.zeros((0, 3)), np.zeros((0, 3)),
                   [[0, N, N, 1, N, N, 2, N, N]])
    d, b = _synthetic_component(degenerate), _synthetic_component(big)
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.0, 2.0))
    for order in ((d, b), (b, d)):
        for W, Hh in ((8, 8), (16, 16), (50, 37)):
            frame = [(mc, sh) for mc in order]
            SU.assert_parity(SU.oracle_render(oracle, W, Hh, frame), SU.gpu_render(W, Hh, frame, mode=0))
    rng = np.random.default_rng(78)
    for n_tris, W, Hh in ((300, 96, 80), (1500, 64, 64), (60, 24, 16), (400, 200, 120)):
        draws = _random_scene(rng, n_tris, W, Hh, spread=1.1)
        o = SU.oracle_render(oracle, W, Hh, draws)
        g = SU.gpu_render(W, Hh, draws, mode=0)
        SU.assert_parity(o, g)
        assert g["counters"]["samples"] == o["counters"]["samples"]


def test_deferred_shade_records(oracle, monkeypatch):
    """Draws of 65 536 triangles or more write no TriShade in k_setup; k_build_shade writes it before the shade pass
    for the triangles the winner pass flagged.  Forced onto every draw here (TR_DEFER_SHADE=1 with the unfused
    front end), with and without the small-triangle path: two draws, NaN depths, a frame extended after a resolve,
    a late plane request, and a mesh destroyed under its recorded frame."""
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader, ModelRendererState, clear_screen, render_mesh_component

    monkeypatch.setenv("TR_DEFER_SHADE", "1")
    monkeypatch.setenv("TR_FUSED_FRONT", "0")
    for small in ("1", "1000000000"):
        monkeypatch.setenv("TR_SMALL_DRAW", small)
        for W, Hh in ((64, 48), (320, 200)):
            draws = _head_scene(W, Hh, with_eye=True)
            o = SU.oracle_render(oracle, W, Hh, draws)
            g = SU.gpu_render(W, Hh, draws, mode=0)
            info = SU.assert_parity(o, g)
            assert info["f64_colour_mismatches"] == 0
        N = H.NONE_INDEX
        degenerate = H.Object([[-0.5, 0.25, 0.0], [0.5, 0.25, 0.0], [0.0, 0.25, 0.0]],
                              [[0, 0, 0], [1, 0, 0], [0.5, 1, 0]], np.zeros((0, 3)), [[0, 0, N, 1, 1, N, 2, 2, N]])
        big = H.Object([[-1, -1, 0.5], [1, -1, 0.5], [-1, 1, 0.5]], np.zeros((0, 3)), np.zeros((0, 3)),
                       [[0, N, N, 1, N, N, 2, N, N]])
        d, b = _synthetic_component(degenerate), _synthetic_component(big)
        sh = BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.0, 2.0))
        for order in ((d, b), (b, d)):
            for W, Hh in ((8, 8), (50, 37)):
                frame = [(mc, sh) for mc in order]
                SU.assert_parity(SU.oracle_render(oracle, W, Hh, frame), SU.gpu_render(W, Hh, frame, mode=0))
        W, Hh = 160, 120
        draws = _head_scene(W, Hh, with_eye=True)
        o1, o2 = SU.oracle_render(oracle, W, Hh, draws[:1]), SU.oracle_render(oracle, W, Hh, draws)
        st = ModelRendererState(W, Hh, raster_mode=0)
        try:
            clear_screen(st)
            render_mesh_component(st, *draws[0])
            st.resolve(_ffi.TR_OUT_RGBA8)
            SU.check_present(st, o1["present"])
            assert np.array_equal(st.frame_buffer.view(np.uint64), o1["fb"].view(np.uint64))  # late plane: shaded again
            render_mesh_component(st, *draws[1])
            st.resolve(_ffi.TR_OUT_RGBA8)
            SU.check_present(st, o2["present"])
            # the head's mesh goes away under the recorded frame: resolved planes stay, a new resolve is refused
            mesh = st._meshes.pop(id(draws[0][0].mesh))
            assert st._lib.tr_mesh_destroy(st._ctx, mesh) == 0
            SU.assert_image(st.present(), o2["present"])
            with pytest.raises(_ffi.TrError, match="destroyed"):
                st.frame_buffer
            clear_screen(st)
            render_mesh_component(st, *draws[0])
            SU.check_present(st, o1["present"])
        finally:
            st.close()


def test_small_triangle_path(oracle, monkeypatch):
    """k_small (one thread per triangle half, taken by triangles of at most 9 scanlines per half in
    draws of 65 536 triangles or more): forced onto every draw here.  Whole frames, counters, bands,
    a frame extended after a resolve, NaN depths and a mesh that mixes small and large triangles."""
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader, ModelRendererState, clear_screen, render_mesh_component
    from tinyrenderer_amd import _ffi

    monkeypatch.setenv("TR_SMALL_DRAW", "1")
    for W, Hh in ((64, 48), (200, 160)):  # the head's 2492 triangles: all small / most small
        draws = _head_scene(W, Hh, with_eye=True)
        o = SU.oracle_render(oracle, W, Hh, draws)
        g = SU.gpu_render(W, Hh, draws, mode=0)
        info = SU.assert_parity(o, g)
        assert info["f64_colour_mismatches"] == 0
        for k in ("samples", "fragments", "scanlines"):
            assert g["counters"][k] == o["counters"][k], k
        top = SU.gpu_render(W, Hh, draws, mode=0, band=(0, 32))
        assert np.array_equal(top["key"][:, :32], o["key"][:, :32]) and (top["key"][:, 32:] == np.uint64(2**64 - 1)).all()
    # a frame extended after an RGBA8-only resolve (the depth pass of the whole frame is redone)
    W, Hh = 160, 120
    draws = _head_scene(W, Hh, with_eye=True)
    o2 = SU.oracle_render(oracle, W, Hh, draws)
    st = ModelRendererState(W, Hh, raster_mode=0)
    try:
        clear_screen(st)
        render_mesh_component(st, *draws[0])
        st.resolve(_ffi.TR_OUT_RGBA8)
        render_mesh_component(st, *draws[1])
        st.resolve(_ffi.TR_OUT_RGBA8)
        SU.check_present(st, o2["present"])
        assert np.array_equal(st.winner_keys(), o2["key"])
        assert st.counters()["samples"] == o2["counters"]["samples"]
    finally:
        st.close()
    # NaN depths from a degenerate small triangle, before and after a triangle that covers the pixel
    N = H.NONE_INDEX
    degenerate = H.Object([[-0.5, 0.25, 0.0], [0.5, 0.25, 0.0], [0.0, 0.25, 0.0]],
                          [[0, 0, 0], [1, 0, 0], [0.5, 1, 0]], np.zeros((0, 3)), [[0, 0, N, 1, 1, N, 2, 2, N]])
    big = H.Object([[-1, -1, 0.5], [1, -1, 0.5], [-1, 1, 0.5]], np.zeros((0, 3)), np.zeros((0, 3)),
                   [[0, N, N, 1, N, N, 2, N, N]])
    d, b = _synthetic_component(degenerate), _synthetic_component(big)
    sh = BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.0, 2.0))
    for order in ((d, b), (b, d)):
        for W, Hh in ((8, 8), (16, 16)):  # at 8x8 both triangles are small
            frame = [(mc, sh) for mc in order]
            SU.assert_parity(SU.oracle_render(oracle, W, Hh, frame), SU.gpu_render(W, Hh, frame, mode=0))
    # random triangles of every size in one mesh (tiny ones, slivers, huge ones, some off screen)
    rng = np.random.default_rng(77)
    for n_tris, W, Hh in ((300, 96, 80), (1500, 64, 64), (60, 24, 16)):
        draws = _random_scene(rng, n_tris, W, Hh, spread=1.1)
        o = SU.oracle_render(oracle, W, Hh, draws)
        g = SU.gpu_render(W, Hh, draws, mode=0)
        SU.assert_parity(o, g)
        assert g["counters"]["samples"] == o["counters"]["samples"]


def test_device_arithmetic_shortcuts_are_exact():
    """tr_selftest_arithmetic: x/255 by fmas and the rounding shortcut against `/` and round() on
    the device, 6e8 operands each."""
    import ctypes as C

    from tinyrenderer_amd import _ffi

    lib = _ffi.load()
    out = (C.c_uint64 * 4)()
    for seed in (1, 0xC0FFEE):
        rc = lib.tr_selftest_arithmetic(0, 300_000_000, seed, out)
        assert rc == 0
        assert out[3] >= 200_000_000
        assert list(out[:3]) == [0, 0, 0], list(out)


def test_context_lifecycle_leaves_no_device_memory_behind(oracle):
    """Contexts of several sizes created, used (every raster mode, extension shaders, a shadow map,
    late plane requests) and destroyed: the device's free memory comes back, results stay right."""
    import ctypes as C

    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import (ModelRendererState, ShadeExt, ShadowMap, clear_screen,
                                                 render_mesh_component)

    def cycle(W, Hh, mode):
        draws = _head_scene(W, Hh, with_eye=True)
        st = ModelRendererState(W, Hh, raster_mode=mode)
        clear_screen(st)
        for d in draws:
            render_mesh_component(st, *d)
        st.resolve(_ffi.TR_OUT_RGBA8 | _ffi.TR_OUT_Z)
        sm = ShadowMap(st)
        clear_screen(st)
        render_mesh_component(st, draws[0][0], draws[0][1], ShadeExt(model=_ffi.TR_SHADE_PHONG, shadow=sm, shadow_dim=0.5))
        st.resolve(_ffi.TR_OUT_RGBA8)
        st.present()
        clear_screen(st)
        for d in draws:
            render_mesh_component(st, *d)
        out = (st.present().copy(), st.winner_keys().copy())
        sm.close()
        st.close()
        return out

    ref = cycle(96, 64, 0)
    o = SU.oracle_render(oracle, 96, 64, _head_scene(96, 64, with_eye=True))
    assert np.array_equal(ref[0], o["present"]) and np.array_equal(ref[1], o["key"])
    # (the HIP runtime the library itself is linked against: no second runtime is initialised for this)
    hip = C.CDLL("libamdhip64.so.7")

    def free_bytes():
        free, total = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    free0 = free_bytes()
    for k in range(12):
        W, Hh = (96, 64) if k % 3 == 0 else ((640, 360) if k % 3 == 1 else (33, 1000))
        got = cycle(W, Hh, k % 3)
        if (W, Hh) == (96, 64):
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    free1 = free_bytes()
    assert free0 - free1 < 64 << 20, f"{(free0 - free1) >> 20} MiB of device memory not returned"


_STALL_CHILD = r"""
import sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import scene_util as SU
from tinyrenderer_amd import _ffi, scenes
from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component
assert b"diag" in _ffi.load().tr_version()
W, Hh = 320, 180
head = SU.component("head")
head.transform.set_rotation((0.025, 0.0, 0.0))
st = ModelRendererState(W, Hh)
t0 = time.perf_counter()
clear_screen(st)
render_mesh_component(st, head, scenes.shader_for(head, W, Hh))
st.resolve(_ffi.TR_OUT_RGBA8)
st.synchronize()
assert time.perf_counter() - t0 < 20.0, "the bounded poll did not give up"
assert st.status_flags() & _ffi.TR_FLAG_PREFIX_STALL, hex(st.status_flags())
st.close()
print("STALL-FLAG-SEEN")
"""


def test_fused_setup_prefix_poll_gives_up_instead_of_hanging(oracle):
    """k_setup<true> polls the slot totals of the workgroups before it; a word that never shows the launch's tag must end
    in TR_FLAG_PREFIX_STALL after a bounded number of polls, not in a hang.  TR_DEBUG_PREFIX_STALL, which makes the pollers
    wait for a tag nobody publishes, exists in the DIAGNOSTIC build only (a production process must not be able to ask
    for incomplete frames through its environment): a child process loads that build.  The release build ignores the
    switch."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TR_DEBUG_PREFIX_STALL="1", TINYRENDERER_HIP_DIAG="1")
    env.pop("TINYRENDERER_HIP_LIB", None)
    res = subprocess.run([sys.executable, "-c", _STALL_CHILD, root], env=env, capture_output=True, text=True, timeout=150)
    assert res.returncode == 0 and "STALL-FLAG-SEEN" in res.stdout, res.stdout + res.stderr
    W, Hh = 320, 180
    draws = _head_scene(W, Hh, with_eye=False)
    os.environ["TR_DEBUG_PREFIX_STALL"] = "1"
    try:
        g = SU.gpu_render(W, Hh, draws)  # (this process holds the release build: the switch is not read)
    finally:
        del os.environ["TR_DEBUG_PREFIX_STALL"]
    o = SU.oracle_render(oracle, W, Hh, draws)
    SU.assert_parity(o, g)


_BOX_CHILD = r"""
import sys, hashlib
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import scene_util as SU
from tinyrenderer_amd import _ffi, scenes, helpers as H_
from tinyrenderer_amd.model_renderer import ModelRendererState, BuiltinShader, clear_screen, render_mesh_component
import test_gpu_parity as T
assert b"diag" in _ffi.load().tr_version()
out = []
def frame(W, Hh, draws, band=None, threads=None):
    st = ModelRendererState(W, Hh)
    try:
        clear_screen(st)
        if band is not None:
            st.set_band(*band)
        for mc, sh in draws:
            render_mesh_component(st, mc, sh)
        st.resolve(_ffi.TR_OUT_RGBA8 | _ffi.TR_OUT_Z | _ffi.TR_OUT_TRI_ID)
        h = hashlib.sha256()
        for a in (st.z_buffer, st.winner_keys(), st.present()):
            h.update(np.ascontiguousarray(a).tobytes())
        c = st.counters()
        assert st.status_flags() == 0, hex(st.status_flags())
        return h.hexdigest() + " %d %d" % (c["samples"], c["fragments"])
    finally:
        st.close()
# the head (chunks whose boxes fit), a band of it, and triangles from slivers to most of a 2048-pixel row (boxes that do
# not fit, scanlines wider than a box may be)
W, Hh = 640, 360
head = T._head_scene(W, Hh, with_eye=True)
out.append(frame(W, Hh, head))
out.append(frame(W, Hh, head, band=(96, 224)))
rng = np.random.default_rng(11)
v = np.zeros((60, 3)); v[:, 0] = rng.uniform(-0.98, 0.98, 60); v[:, 1] = rng.uniform(-0.98, 0.98, 60); v[:, 2] = rng.uniform(-0.9, 0.9, 60)
vt = rng.uniform(0.02, 0.98, (60, 3)); vn = rng.normal(size=(60, 3))
tris = [[a, a, a, b, b, b, c, c, c] for a, b, c in (rng.choice(60, 3, replace=False) for _ in range(24))]
order = np.argsort(v[:, 0])
tris += [[order[k]] * 3 + [order[k + 1]] * 3 + [order[k + 2]] * 3 for k in range(58)]
obj = H_.Object(v.reshape(-1), vt.reshape(-1), vn.reshape(-1), np.array(tris, dtype=np.uint32).reshape(-1))
mixed = [(T._synthetic_component(obj, seed=5), BuiltinShader(np.eye(4), np.eye(4), (0.3, 0.2, 1.0)))]
out.append(frame(2048, 256, mixed))
print("BOX-FRAMES " + " | ".join(out))
"""


def test_depth_pass_lds_boxes_against_the_per_sample_form(oracle):
    """The depth pass reduces a chunk's samples per pixel in LDS boxes before it touches the plane.  The
    diagnostic build can switch the boxes off (TR_RASTER_BOX=0: every sample loads the plane and issues its own atomic, the
    form of rounds 1 and 2): both forms, with workgroups of 128 and of 256 threads (boxes of 512 and 1 024 pixels), must
    leave the same depth plane, winner keys, image and counters -- for a scene whose boxes fit, for a band of it, and for
    one with boxes that do not fit and scanlines wider than a box may be.  (Every other test checks the boxed form, the
    release build's only one, against the oracle.)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = {}
    for box in ("0", "1"):
        for threads in ("128", "256"):
            env = dict(os.environ, TR_RASTER_BOX=box, TR_RASTER_THREADS=threads, TINYRENDERER_HIP_DIAG="1")
            env.pop("TINYRENDERER_HIP_LIB", None)
            res = subprocess.run([sys.executable, "-c", _BOX_CHILD, root], env=env, capture_output=True, text=True, timeout=200)
            lines = [l for l in res.stdout.splitlines() if l.startswith("BOX-FRAMES ")]
            assert res.returncode == 0 and lines, res.stdout + res.stderr
            seen[(box, threads)] = lines[-1]
    assert len(set(seen.values())) == 1, seen


def test_boxed_shade_launch_follows_what_moves(oracle):
    """The cheap resolve launches only the blocks of the frame's screen box plus the box the previous launch painted:
    one context, frames whose content moves, shrinks, vanishes and comes back must each equal the oracle's image
    (nothing stale outside the new box, nothing missing inside it)."""
    from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

    W, Hh = 640, 360
    head = _head_scene(W, Hh, with_eye=False)
    eye_mc = SU.component("eye_inner")
    eye_left = [(eye_mc, scenes.shader_for(eye_mc, W, Hh, eye=(0.9, 0.1, 1.0)))]
    eye_right = [(eye_mc, scenes.shader_for(eye_mc, W, Hh, eye=(-0.9, -0.2, 1.0)))]
    st = ModelRendererState(W, Hh)
    try:
        for n, draws in enumerate((head, eye_left, eye_right, [], eye_left, head, [])):
            clear_screen(st)
            for d in draws:
                render_mesh_component(st, *d)
            got = st.present()
            if draws:
                want = SU.oracle_render(oracle, W, Hh, draws)["present"]
            else:
                want = np.zeros((Hh, W, 4), dtype=np.uint8)
                want[..., 3] = 255
            SU.assert_image(got, want, f"frame {n}")
            assert st.status_flags() == 0
    finally:
        st.close()


def test_band_contexts_drop_clusters_exactly(oracle):
    """Band contexts of a 159 488-triangle mesh drop whole clusters of 64 triangles (k_cull_clusters; the scan and
    k_build_shade consult the same bytes, the dropped clusters' flags are cleared).  One context, frame after frame:
    random bands (narrow, wide, one tile row, the whole frame), the big mesh alone and behind a small draw that puts
    its triangles at a frame position that is no multiple of 64 or 32, the camera moved between frames -- the band's
    rows must equal the oracle's, in keys, depth and colour."""
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import MeshComponent, ModelRendererState, clear_screen, render_mesh_component

    W, Hh = 960, 544
    head = SU.component("head")
    fine = MeshComponent(scenes.subdivide(head.mesh, 3), head.texture, head.normal_map)
    eye_mc = SU.component("eye_inner")
    rng = np.random.default_rng(11)
    st = ModelRendererState(W, Hh, raster_mode=0)
    try:
        for n in range(7):
            fine.transform.set_rotation((0.025 + 0.4 * n, 0.1 * n, 0.0))
            eye = (0.5 * np.sin(n), 0.1 * n - 0.3, 2.0 - 0.15 * n)
            draws = [(fine, scenes.shader_for(fine, W, Hh, eye=eye))]
            if n % 2:
                draws = [(eye_mc, scenes.shader_for(eye_mc, W, Hh, eye=eye))] + draws
            lo = int(rng.integers(0, Hh // 16)) * 16
            hi = min(Hh, lo + 16 * int(rng.integers(1, 12)))
            if n == 5:
                lo, hi = 0, Hh
            o = SU.oracle_render(oracle, W, Hh, draws)
            clear_screen(st)
            st.set_band(lo, hi)
            for d in draws:
                render_mesh_component(st, *d)
            st.resolve(_ffi.TR_OUT_RGBA8 | _ffi.TR_OUT_Z | _ffi.TR_OUT_KEY | _ffi.TR_OUT_RGBA_F64)
            assert np.array_equal(st.winner_keys()[:, lo:hi], o["key"][:, lo:hi]), (n, lo, hi)
            assert np.array_equal(st.z_buffer[..., 0][:, lo:hi], o["z"][:, lo:hi], equal_nan=True), (n, lo, hi)
            assert np.array_equal(st.frame_buffer[:, lo:hi], o["fb"][:, lo:hi]), (n, lo, hi)
            assert np.array_equal(st.present()[Hh - hi:Hh - lo], o["present"][Hh - hi:Hh - lo]), (n, lo, hi)
            assert st.status_flags() == 0
    finally:
        st.close()


def test_tiled_planes_frame_with_every_raster_kernel(oracle):
    """A frame whose FIRST draw is a mesh of small triangles keeps its depth / key planes in tiles of 4 x 2 pixels
    (plane_index): the draws after it -- an ordinary mesh through k_raster, a degenerate triangle whose NaN depth goes
    through the side table -- address the same tiled planes, and the cheap and the full resolve read them back."""
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader, MeshComponent, ModelRendererState, clear_screen, render_mesh_component

    W, Hh = 960, 544
    head = SU.component("head")
    fine = MeshComponent(scenes.subdivide(head.mesh, 3), head.texture, head.normal_map)
    fine.transform.set_rotation((0.3, 0.1, 0.0))
    eye_mc = SU.component("eye_inner")
    N = H.NONE_INDEX
    degenerate = H.Object([[-0.5, 0.25, 0.0], [0.5, 0.25, 0.0], [0.0, 0.25, 0.0]],
                          [[0, 0, 0], [1, 0, 0], [0.5, 1, 0]], np.zeros((0, 3)), [[0, 0, N, 1, 1, N, 2, 2, N]])
    deg = _synthetic_component(degenerate)
    draws = [(fine, scenes.shader_for(fine, W, Hh)), (eye_mc, scenes.shader_for(eye_mc, W, Hh, eye=(0.4, 0.2, 1.1))),
             (deg, BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.0, 2.0)))]
    o = SU.oracle_render(oracle, W, Hh, draws)
    g = SU.gpu_render(W, Hh, draws, mode=0)
    info = SU.assert_parity(o, g)
    assert info["f64_colour_mismatches"] == 0
    st = ModelRendererState(W, Hh, raster_mode=0)  # the cheap resolve, twice (the planes clean themselves in between)
    try:
        for _ in range(2):
            clear_screen(st)
            for d in draws:
                render_mesh_component(st, *d)
            SU.check_present(st, o["present"])
    finally:
        st.close()
