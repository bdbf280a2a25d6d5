"""EXTENSION shaders (Gouraud / Phong switch, specular map, shadow map: BASELINE.json north_star and
configs[1] name them; the reference has none of them).  They are defined by
include/tinyrenderer_hip.h and restated on the CPU in oracle/ (ora_render_mesh_component_ext): what
is checked here is GPU against that CPU statement, NOT parity with the reference ("parity
unpinned" for these shaders by construction).  Visibility (depth, triangle id, winner key) is the
reference's and stays bit-exact; colour is within 1 LSB (pow() differs between libm and the device
library in the last bits)."""
import numpy as np
import pytest

import oracle_ffi
import scene_util as SU
from tinyrenderer_amd import _ffi, scenes
from tinyrenderer_amd.model_renderer import (HostShadow, ModelRendererState, ShadeExt, ShadowMap, clear_screen,
                                             render_mesh_component, shadow_from_view)

W = H = 192
LW = LH = 160
LIGHT_EYE = (1.2, 0.9, 1.6)


def _scene(name):
    mc = SU.component(name)
    mc.parent = None
    mc.transform.set_rotation((0.0, 0.0, 0.0))
    return mc, scenes.shader_for(mc, W, H), scenes.shader_for(mc, LW, LH, eye=LIGHT_EYE)


def _oracle_light_z(mc, light_sh):
    return SU.oracle_render(oracle_ffi, LW, LH, [(mc, light_sh)])["z"]


def test_oracle_ext_off_is_the_reference():
    mc, sh, _ = _scene("head")
    ref = SU.oracle_render(oracle_ffi, 96, 96, [(mc, scenes.shader_for(mc, 96, 96))])
    ext = SU.oracle_render(oracle_ffi, 96, 96, [(mc, scenes.shader_for(mc, 96, 96), ShadeExt())])
    assert (ref["fb"].view(np.uint64) == ext["fb"].view(np.uint64)).all()
    assert (ref["key"] == ext["key"]).all()


def test_oracle_ext_models_change_colour_only():
    mc, sh, light_sh = _scene("diablo")
    spec = scenes.load_specular("diablo")
    ref = SU.oracle_render(oracle_ffi, W, H, [(mc, sh)])
    gouraud = SU.oracle_render(oracle_ffi, W, H, [(mc, sh, ShadeExt(model=_ffi.TR_SHADE_GOURAUD))])
    phong = SU.oracle_render(oracle_ffi, W, H, [(mc, sh, ShadeExt(model=_ffi.TR_SHADE_PHONG, specular_map=spec))])
    for o in (gouraud, phong):
        assert (o["key"] == ref["key"]).all() and (o["z"].view(np.uint64) == ref["z"].view(np.uint64)).all()
    covered = np.isfinite(ref["z"])
    # the specular term only adds light; Gouraud differs from the normal-mapped shading somewhere
    assert (phong["fb"][covered] >= ref["fb"][covered] - 1e-9).all()
    assert (phong["fb"][covered] > ref["fb"][covered] + 1.0).any()
    assert (np.abs(gouraud["fb"] - ref["fb"])[covered] > 1.0).any()


def test_oracle_shadow_map_darkens_some_pixels_and_only_lit_terms():
    mc, sh, light_sh = _scene("diablo")
    zl = _oracle_light_z(mc, light_sh)
    m = shadow_from_view(sh, W, H, light_sh, LW, LH)
    ref = SU.oracle_render(oracle_ffi, W, H, [(mc, sh)])
    ext = ShadeExt(shadow=HostShadow(zl), shadow_from_view=m, shadow_bias=0.01, shadow_dim=0.3)
    sha = SU.oracle_render(oracle_ffi, W, H, [(mc, sh, ext)])
    covered = np.isfinite(ref["z"])
    darker = (sha["fb"][..., :3].sum(axis=2) < ref["fb"][..., :3].sum(axis=2) - 1.0) & covered
    frac = darker.sum() / covered.sum()
    assert 0.02 < frac < 0.9, frac  # some of the figure is in its own shadow, not all of it
    assert (sha["fb"][covered][:, :3] >= sh.ambient_light - 1e-12).all()  # the ambient term is never dimmed
    # a light at the camera casts no visible shadow (up to the bias): same frame as without a map
    zc = SU.oracle_render(oracle_ffi, W, H, [(mc, sh)])["z"]
    same = ShadeExt(shadow=HostShadow(zc), shadow_from_view=np.eye(4), shadow_bias=1e-6, shadow_dim=0.3)
    unshadowed = SU.oracle_render(oracle_ffi, W, H, [(mc, sh, same)])
    assert np.abs(unshadowed["fb"] - ref["fb"]).max() < 1e-9


def _gpu_with_shadow(draw_specs, light_draws):
    """light_draws: [(mc, light_shader)] rendered at LWxLH into the shadow map; draw_specs: list of
    (mc, shader, ShadeExt-kwargs or None).  Returns (gpu planes, oracle planes)."""
    light = ModelRendererState(LW, LH)
    clear_screen(light)
    for mc, lsh in light_draws:
        render_mesh_component(light, mc, lsh)
    light.resolve(_ffi.TR_OUT_RGBA8 | _ffi.TR_OUT_Z)
    sm = ShadowMap(light, keep_host_copy=True)
    light.close()  # the map outlives the context it was taken from
    zl = SU.oracle_render(oracle_ffi, LW, LH, light_draws)["z"]
    assert (zl.view(np.uint64) == sm.z_reference().view(np.uint64)).all()
    draws = []
    for mc, sh, kw in draw_specs:
        if kw is None:
            draws.append((mc, sh))
        else:
            kw = dict(kw)
            if kw.pop("use_shadow", False):
                kw["shadow"] = sm
            draws.append((mc, sh, ShadeExt(**kw)))
    g = SU.gpu_render(W, H, draws)
    o = SU.oracle_render(oracle_ffi, W, H, draws)
    sm.close()
    return g, o


@pytest.mark.gpu
@pytest.mark.parametrize("model", [_ffi.TR_SHADE_GOURAUD, _ffi.TR_SHADE_PHONG])
def test_gpu_ext_models_match_cpu_statement(model):
    mc, sh, light_sh = _scene("diablo")
    spec = scenes.load_specular("diablo") if model == _ffi.TR_SHADE_PHONG else None
    draws = [(mc, sh, ShadeExt(model=model, specular_map=spec, specular_strength=0.6))]
    o = SU.oracle_render(oracle_ffi, W, H, draws)
    for mode in (_ffi.TR_RASTER_GLOBAL, _ffi.TR_RASTER_TILE, _ffi.TR_RASTER_ROWS):  # the shade pass is shared
        g = SU.gpu_render(W, H, draws, mode=mode)
        r = SU.assert_parity(o, g, color_lsb=1)
        assert r["covered"] > 3000
        if model == _ffi.TR_SHADE_GOURAUD:  # no pow(): the same operations in the same order
            assert r["f64_colour_mismatches"] == 0


@pytest.mark.gpu
def test_gpu_diablo_phong_specular_shadow_map_config1():
    """BASELINE.json configs[1] in small: diablo3_pose, normal + specular maps, shadow-map pass."""
    mc, sh, light_sh = _scene("diablo")
    m = shadow_from_view(sh, W, H, light_sh, LW, LH)
    kw = dict(model=_ffi.TR_SHADE_PHONG, specular_map=scenes.load_specular("diablo"), use_shadow=True,
              shadow_from_view=m, shadow_bias=0.01, shadow_dim=0.3)
    g, o = _gpu_with_shadow([(mc, sh, kw)], [(mc, light_sh)])
    r = SU.assert_parity(o, g, color_lsb=1)
    ref = SU.oracle_render(oracle_ffi, W, H, [(mc, sh)])
    assert (np.abs(o["present"].astype(int) - ref["present"].astype(int)).max(axis=2) > 8).sum() > 200
    assert r["covered"] > 3000


@pytest.mark.gpu
def test_gpu_config1_full_size_1024_phong_specular_shadow():
    """BASELINE.json configs[1] at FULL size: diablo3_pose 1024x1024, normal + specular maps, 1024x1024 shadow map, light
    pass + camera pass.  EXTENSION shaders: the yardstick is the build's own CPU statement (oracle/), directly and through
    the digests it left in tests/golden/hashes.json (make_golden.py --ext).  Visibility and the light's depth plane bit
    for bit, colour within 1 LSB (pow() differs between libm and the device library in the last bits)."""
    import json
    import os

    import big_cases as B

    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hashes.json")))["c2x_diablo_1024_phong_spec_shadow"]
    Wf = Hf = M = 1024
    mc, sh, light_sh, m = B.config1_ext_scene(Wf, Hf, M)
    light = ModelRendererState(M, M)
    try:
        clear_screen(light)
        render_mesh_component(light, mc, light_sh)  # (not resolved: the snapshot resolves for depth only)
        sm = ShadowMap(light, keep_host_copy=True)
    finally:
        light.close()
    try:
        assert B.sha(np.ascontiguousarray(sm.z_reference().T).view(np.uint64)) == want["light_z"]
        g = SU.gpu_render(Wf, Hf, [(mc, sh, B.config1_ext(sm, m))])
    finally:
        sm.close()
    assert g["flags"] == 0
    assert B.sha(np.ascontiguousarray(g["key"].T)) == want["key"] and int((g["key"] != B.NONE_KEY).sum()) == want["covered"]
    assert B.sha(np.ascontiguousarray(g["z"].T).view(np.uint64)) == want["z"]
    for k in ("samples", "fragments", "scanlines"):
        assert g["counters"][k] == want["counters"][k]
    o, _ = B.config1_ext_oracle(oracle_ffi, Wf, Hf, M)
    assert B.sha(o["present"]) == want["rgba8"]  # (the CPU statement still says what it said when the digest was made)
    r = SU.assert_parity(o, g, color_lsb=1)
    ref = SU.oracle_render(oracle_ffi, Wf, Hf, [(mc, sh)])
    changed = (np.abs(o["present"].astype(int) - ref["present"].astype(int)).max(axis=2) > 8).sum()
    assert changed > 5000, "specular term and shadow must change the frame visibly"
    print("configs[1] 1024x1024 ext:", r, "pixels changed by the extensions:", int(changed))


@pytest.mark.gpu
def test_gpu_mixed_frame_reference_draw_stays_exact():
    """A frame with one extension draw and one reference draw: the latter's pixels carry the
    reference's f64 colours bit for bit."""
    head, eye = SU.component("head"), SU.component("eye_inner")
    head.parent = None
    eye.parent = None
    head.transform.set_rotation((0.0, 0.0, 0.0))
    sh_h, sh_e = scenes.shader_for(head, W, H), scenes.shader_for(eye, W, H)
    light_h = scenes.shader_for(head, LW, LH, eye=LIGHT_EYE)
    m = shadow_from_view(sh_h, W, H, light_h, LW, LH)
    kw = dict(model=_ffi.TR_SHADE_PHONG, specular_map=scenes.load_specular("head"), use_shadow=True,
              shadow_from_view=m, shadow_bias=0.01, shadow_dim=0.4)
    g, o = _gpu_with_shadow([(head, sh_h, kw), (eye, sh_e, None)], [(head, light_h)])
    SU.assert_parity(o, g, color_lsb=1)
    ref = SU.oracle_render(oracle_ffi, W, H, [(head, sh_h), (eye, sh_e)])
    n_head = head.mesh.triangles.shape[0]
    eye_px = g["seq"] >= n_head
    assert eye_px.sum() > 50
    assert (g["fb"][eye_px].view(np.uint64) == ref["fb"][eye_px].view(np.uint64)).all()


@pytest.mark.gpu
def test_gpu_ext_argument_checks():
    mc, sh, _ = _scene("head")
    st = ModelRendererState(64, 64)
    try:
        clear_screen(st)
        with pytest.raises(_ffi.TrError):
            render_mesh_component(st, mc, sh, ShadeExt(model=7))
        # ext with nothing switched on is the reference draw
        render_mesh_component(st, mc, sh, ShadeExt())
        a = st.present().copy()
        clear_screen(st)
        render_mesh_component(st, mc, sh)
        assert (st.present() == a).all()
    finally:
        st.close()


@pytest.mark.gpu
def test_gpu_shadow_map_update_and_depth_only_snapshot():
    """A map taken from a light frame nobody resolved (resolved for depth only: the image of that frame
    is still right when asked for afterwards), then refreshed in place from a second light
    (tr_shadow_map_update: no allocation, no host wait) while the camera context keeps rendering."""
    mc, sh, light_a = _scene("diablo")
    light_b = scenes.shader_for(mc, LW, LH, eye=(-1.4, 0.6, 1.5))
    light = ModelRendererState(LW, LH)
    cam = ModelRendererState(W, H)
    spec = scenes.load_specular("diablo")

    def light_pass(lsh):
        clear_screen(light)
        render_mesh_component(light, mc, lsh)  # (no resolve: the snapshot does what it needs)

    def camera_pass(sm, lsh):
        ext = ShadeExt(model=_ffi.TR_SHADE_PHONG, specular_map=spec, specular_strength=0.6, shadow=sm,
                       shadow_from_view=shadow_from_view(sh, W, H, lsh, LW, LH), shadow_bias=0.01, shadow_dim=0.3)
        clear_screen(cam)
        render_mesh_component(cam, mc, sh, ext)
        cam.resolve(_ffi.TR_OUT_RGBA8)
        return ext

    light_pass(light_a)
    sm = ShadowMap(light, keep_host_copy=True)
    o_light = SU.oracle_render(oracle_ffi, LW, LH, [(mc, light_a)])
    assert (o_light["z"].view(np.uint64) == sm.z_reference().view(np.uint64)).all()
    SU.check_present(light, o_light["present"])  # the colour the snapshot skipped
    for lsh in (light_a, light_b, light_a, light_b):
        light_pass(lsh)
        sm.update(light)
        ext = camera_pass(sm, lsh)
        got = cam.present()
        o = SU.oracle_render(oracle_ffi, W, H, [(mc, sh, ShadeExt(model=ext.model, specular_map=spec, specular_strength=0.6,
                                                                     shadow=HostShadow(sm.z_reference()),
                                                                     shadow_from_view=ext.shadow_from_view, shadow_bias=0.01,
                                                                     shadow_dim=0.3))])
        assert np.abs(o["present"].astype(int) - got.astype(int)).max() <= 1
    # frames in flight: several light / camera passes queued without a host wait, only the last one read
    for k in range(6):
        lsh = light_b if k % 2 else light_a
        light_pass(lsh)
        sm.update(light)
        ext = camera_pass(sm, lsh)
    got = cam.present()
    assert np.abs(o["present"].astype(int) - got.astype(int)).max() <= 1  # (the last pass used light_b, like `o`)
    cam.close()
    light.close()
    sm.close()  # (after the contexts that drew with it)
