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
