"""BASELINE.json configs[2], [3] and [4] at FULL size: scene builders and row-major digests shared by
tests/golden/make_golden.py (oracle side, run in the build container), tests/test_fullsize.py (GPU side)
and bench.py's output check.  Hashes are SHA-256 of row-major (H, W) planes: z as f64 bits, winner key
as u64, the presented RGBA8 image (row 0 = top, helpers/mod.rs:204-222)."""
import hashlib

import numpy as np

import scene_util as SU
from tinyrenderer_amd import scenes

ORBIT_VIEWS = (0, 517, 2048, 3333)  # of 4096 (configs[4]); SURVEY.md section 6 counts view 0
NONE_KEY = np.uint64(0xFFFFFFFFFFFFFFFF)


def head_4k():
    """configs[2]: african_head at 3840x2160, the frame bench.py times."""
    W, H = 3840, 2160
    head = SU.component("head")
    head.transform.set_rotation((0.025, 0.0, 0.0))  # frame time = 1, src/main.rs:92-94
    return W, H, [(head, scenes.shader_for(head, W, H))]


_fine = {}


def head_subdivided(levels, W, H):
    """configs[3]: african_head with every triangle split by its edge midpoints `levels` times
    (6 levels = 10 207 232 triangles at 7680x4320)."""
    from tinyrenderer_amd.model_renderer import MeshComponent

    head = SU.component("head")
    if levels not in _fine:
        _fine[levels] = MeshComponent(scenes.subdivide(head.mesh, levels), head.texture, head.normal_map)
    fine = _fine[levels]
    fine.transform.set_rotation((0.025, 0.0, 0.0))
    return W, H, [(fine, scenes.shader_for(fine, W, H))]


def orbit_view(k, n_views=4096, W=1920, H=1080):
    """configs[4]: view k of the camera orbit of diablo3_pose at 1080p."""
    mc = SU.component("diablo")
    mc.transform.set_rotation((0.0, 0.0, 0.0))
    return W, H, [(mc, scenes.shader_for(mc, W, H, eye=scenes.orbit_eye(k, n_views)))]


def diablo_1080p_frame1():
    """The frame SURVEY.md section 6 counted independently (1 412 616 samples, 192 579 covered pixels):
    diablo3_pose at 1920x1080, camera (0,0,2), rotation 0.025."""
    W, H = 1920, 1080
    mc = SU.component("diablo")
    mc.transform.set_rotation((0.025, 0.0, 0.0))
    return W, H, [(mc, scenes.shader_for(mc, W, H))]


def config1_ext_scene(W=1024, H=1024, M=1024):
    """configs[1]: diablo3_pose with normal + specular maps and a shadow-map pass at 1024x1024.  Phong, the specular
    map and the shadow pass are EXTENSION shaders (the reference has one shader pair and loads no *_spec.png):
    defined by include/tinyrenderer_hip.h, restated on the CPU in oracle/ -- parity unpinned by construction.
    Returns (mesh component, camera shader, light shader, camera-viewport -> light-viewport matrix)."""
    from tinyrenderer_amd.model_renderer import shadow_from_view

    mc = SU.component("diablo")
    mc.parent = None
    mc.transform.set_rotation((0.0, 0.0, 0.0))
    light_eye = (1.2, 0.9, 1.6)
    sh = scenes.shader_for(mc, W, H)
    sh.camera_pos = light_eye  # the shader's light direction (main.rs:249-251 uses the camera's position)
    light_sh = scenes.shader_for(mc, M, M, eye=light_eye)
    return mc, sh, light_sh, shadow_from_view(sh, W, H, light_sh, M, M)


def config1_ext(shadow, m):
    """The ShadeExt of configs[1] over a shadow source (a HostShadow for the CPU statement, a ShadowMap on the GPU)."""
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import ShadeExt

    return ShadeExt(model=_ffi.TR_SHADE_PHONG, specular_map=scenes.load_specular("diablo"), specular_strength=0.6,
                    shadow=shadow, shadow_from_view=m, shadow_bias=0.01, shadow_dim=0.3)


def config1_ext_oracle(oracle, W=1024, H=1024, M=1024):
    """The CPU statement's frame of configs[1]: (planes of the camera pass, z plane of the light pass)."""
    from tinyrenderer_amd.model_renderer import HostShadow

    mc, sh, light_sh, m = config1_ext_scene(W, H, M)
    zl = SU.oracle_render(oracle, M, M, [(mc, light_sh)])["z"]
    return SU.oracle_render(oracle, W, H, [(mc, sh, config1_ext(HostShadow(zl), m))]), zl


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a)).hexdigest()


def digest_row_major(z_hw, key_hw, present_hw4, counters):
    """z (H, W) f64, key (H, W) u64, presented image (H, W, 4) u8, counters dict."""
    key_hw = np.ascontiguousarray(key_hw)
    return {"z": sha(np.ascontiguousarray(z_hw).view(np.uint64)), "key": sha(key_hw), "rgba8": sha(present_hw4),
            "covered": int((key_hw != NONE_KEY).sum()),
            "counters": {k: int(counters[k]) for k in ("samples", "fragments", "scanlines")}}


def oracle_digest(oracle, W, H, draws):
    o = SU.oracle_render(oracle, W, H, draws)  # planes in the reference's (W, H) layout
    return digest_row_major(o["z"].T, o["key"].T, o["present"], o["counters"])


def gpu_planes(W, H, draws, band=None, device=0, state=None):
    """One frame through the HIP path, planes read back row-major: (z, key, presented image, counters)."""
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

    st = state or ModelRendererState(W, H, device=device)
    try:
        clear_screen(st)
        st.set_band(*(band if band is not None else (0, H)))  # (a reused state keeps its last band otherwise)
        for d in draws:
            render_mesh_component(st, d[0], d[1])
        # the digests are of the reference's colour arithmetic; the library's default (fused) must stay within 1 LSB of it
        st.set_colour_exact(True)
        st.resolve(_ffi.TR_OUT_RGBA8 | _ffi.TR_OUT_Z | _ffi.TR_OUT_KEY)
        assert st.status_flags() == 0
        out = st.z_row_major(), st.winner_keys(_ffi.TR_LAYOUT_ROW_MAJOR), st.present(), st.counters()
        if not SU.exact_by_default():
            st.set_colour_exact(False)
            st.resolve(_ffi.TR_OUT_RGBA8 | _ffi.TR_OUT_Z | _ffi.TR_OUT_KEY)  # (shaded again from the same keys)
            d = np.abs(st.present().astype(np.int16) - out[2].astype(np.int16))
            assert d.max() <= 1, f"fused colour arithmetic differs from the exact one by {d.max()} LSB"
        return out
    finally:
        if state is None:
            st.close()


def gpu_digest(W, H, draws, **kw):
    z, key, present, counters = gpu_planes(W, H, draws, **kw)
    return digest_row_major(z, key, present, counters)
