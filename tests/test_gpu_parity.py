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


@pytest.mark.parametrize("W,H", [(64, 64), (128, 96), (200, 200), (853, 480)])
def test_head_small(oracle, W, H):
    draws = _head_scene(W, H)
    o = SU.oracle_render(oracle, W, H, draws)
    g = SU.gpu_render(W, H, draws)
    info = SU.assert_parity(o, g)
    assert g["counters"]["samples"] == o["counters"]["samples"]
    assert g["counters"]["fragments"] == o["counters"]["fragments"]
    assert g["counters"]["scanlines"] == o["counters"]["scanlines"]
    assert info["f64_colour_mismatches"] == 0


def test_config1_head_800(oracle):
    """BASELINE.json configs[0]: african_head 800x800."""
    draws = _head_scene(800, 800)
    o = SU.oracle_render(oracle, 800, 800, draws)
    g = SU.gpu_render(800, 800, draws)
    info = SU.assert_parity(o, g)
    assert o["counters"]["samples"] == 1087161 and info["covered"] == 211671
    assert g["counters"]["covered_pixels"] == 211671
    assert info["f64_colour_mismatches"] == 0


def test_config2_diablo_1024(oracle):
    """BASELINE.json configs[1] (without the shadow pass, which has no reference)."""
    mc = SU.component("diablo")
    mc.transform.set_rotation((0.025, 0.0, 0.0))
    draws = [(mc, scenes.shader_for(mc, 1024, 1024))]
    o = SU.oracle_render(oracle, 1024, 1024, draws)
    g = SU.gpu_render(1024, 1024, draws)
    info = SU.assert_parity(o, g)
    assert o["counters"]["samples"] == 1294206 and info["covered"] == 173184
    assert info["f64_colour_mismatches"] == 0


def test_two_draws_head_and_eye(oracle):
    """The reference's own frame: head then eye_inner parented to it (main.rs:272-273), 853x480."""
    draws = _head_scene(853, 480, rot=0.05, with_eye=True)
    o = SU.oracle_render(oracle, 853, 480, draws)
    g = SU.gpu_render(853, 480, draws)
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
        draws = [(mc, sh) for mc in order]
        o = SU.oracle_render(oracle, 16, 16, draws)
        g = SU.gpu_render(16, 16, draws)
        SU.assert_parity(o, g)
        assert np.isnan(o["z"][0, 0]) == (order[0] is d)


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
    g = SU.gpu_render(120, 90, [(mc, sh)])
    SU.assert_parity(o, g)
    assert o["counters"]["culled"] == 1
    assert g["counters"]["samples"] == o["counters"]["samples"]
    assert g["counters"]["fragments"] == o["counters"]["fragments"] < o["counters"]["samples"]


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
