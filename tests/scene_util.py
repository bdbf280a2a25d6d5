"""Shared helpers for parity tests: render one scene through the oracle and through the GPU
library from the same inputs, and compare every plane."""
import numpy as np

from tinyrenderer_amd import _ffi, scenes
from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

_components = {}


def component(name):
    """Cached MeshComponent for 'head', 'eye_inner' or 'diablo'."""
    if name not in _components:
        triple = {"head": scenes.AFRICAN_HEAD, "eye_inner": scenes.AFRICAN_HEAD_EYE_INNER,
                  "diablo": scenes.DIABLO}[name]
        _components[name] = scenes.load_component(triple)
    return _components[name]


def pack_key(w):
    """Oracle winner arrays -> the library's u64 key (tri_seq<<36 | half<<35 | i<<18 | j)."""
    none = w["tri"] < 0
    key = ((w["seq"].astype(np.uint64) << np.uint64(36)) | (w["half"].astype(np.uint64) << np.uint64(35))
           | (w["i"].astype(np.uint64) << np.uint64(18)) | w["j"].astype(np.uint64))
    key[none] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return key


def oracle_render(oracle, W, H, draws):
    """draws: list of (MeshComponent, BuiltinShader).  Returns dict of planes in (W,H,..) layout."""
    st = oracle.OracleState(W, H)
    st.clear_screen()
    imgs = {}
    tri_base = []
    base = 0
    for draw in draws:
        mc, sh = draw[0], draw[1]
        ext = draw[2] if len(draw) > 2 else None  # (mc, shader, ShadeExt): extension shaders
        spec = ext.specular_map if ext is not None else None
        for img in (mc.texture, mc.normal_map, spec):
            if img is not None and id(img) not in imgs:
                # the oracle flips by itself (model_renderer/mod.rs:40-41): hand it the un-flipped image
                imgs[id(img)] = oracle.OracleImage(img[::-1], flip=True)
        rc = st.render_mesh_component(mc.mesh, sh, imgs.get(id(mc.texture)), imgs.get(id(mc.normal_map)),
                                      ext=ext, specular_map=imgs.get(id(spec)))
        assert rc == 0, "oracle: the reference would have panicked on this scene"
        tri_base.append(base)
        base += mc.mesh.triangles.shape[0]
    w = st.winners()
    tb = np.array(tri_base + [0], dtype=np.int64)
    w["seq"] = np.where(w["tri"] >= 0, tb[np.clip(w["draw"], 0, len(tri_base) - 1)] + w["tri"], -1)
    return {"z": st.z_buffer()[..., 0], "fb": st.frame_buffer(), "key": pack_key(w), "seq": w["seq"],
            "present": st.present(), "counters": st.counters()}


def gpu_render(W, H, draws, device=0, mode=None, band=None):
    """mode: None (library default = global raster), _ffi.TR_RASTER_GLOBAL, _ffi.TR_RASTER_TILE or _ffi.TR_RASTER_ROWS."""
    st = ModelRendererState(W, H, device=device, raster_mode=mode)
    try:
        clear_screen(st)
        if band is not None:
            st.set_band(*band)
        for draw in draws:
            render_mesh_component(st, draw[0], draw[1], draw[2] if len(draw) > 2 else None)
        st.resolve(_ffi.TR_OUT_RGBA8 | _ffi.TR_OUT_Z | _ffi.TR_OUT_TRI_ID | _ffi.TR_OUT_RGBA_F64)
        out = {"z": st.z_buffer[..., 0], "fb": st.frame_buffer, "key": st.winner_keys(),
               "seq": st.triangle_ids().astype(np.int64), "present": st.present(),
               "counters": st.counters(), "flags": st.status_flags()}
        out["seq"][out["seq"] == 0xFFFFFFFF] = -1
        # the same frame once more the way a frame loop renders it: RGBA8 only, the library's fused colour arithmetic
        # (tr_set_colour_exact(0): +-1 LSB; the resolve above asked for the f64 colour plane and so shaded exactly)
        st.set_colour_exact(False)
        clear_screen(st)
        if band is not None:
            st.set_band(*band)
        for draw in draws:
            render_mesh_component(st, draw[0], draw[1], draw[2] if len(draw) > 2 else None)
        st.resolve(_ffi.TR_OUT_RGBA8)
        out["present_fused"] = st.present()
        out["flags"] |= st.status_flags()
        return out
    finally:
        st.close()


def exact_by_default():
    """TR_SHADE_EXACT=1 in the environment: contexts start with the reference's colour arithmetic."""
    import os

    return os.environ.get("TR_SHADE_EXACT", "0") not in ("", "0")


def assert_image(got, want, what=""):
    """An image rendered with the library's DEFAULT colour arithmetic (fused: within 1 LSB of the oracle's; equal
    when TR_SHADE_EXACT=1 makes the reference's arithmetic the default)."""
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    lim = 0 if exact_by_default() else 1
    assert d.max() <= lim, f"{what}: {int((d.max(axis=-1) > lim).sum())} pixels differ by up to {d.max()} LSB, first {np.argwhere(d > lim)[:5].tolist()}"


def check_present(st, want, what=""):
    """The presented image of a resolved state against the oracle's `want`: within 1 LSB with the library's default
    (fused) colour arithmetic, equal with the reference's (tr_set_colour_exact: the frame is shaded again)."""
    got = st.present()
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"{what}: presented colour differs by {d.max()} LSB at {np.argwhere(d > 1)[:5].tolist()}"
    if not exact_by_default():
        st.set_colour_exact(True)
        try:
            st.resolve(_ffi.TR_OUT_RGBA8)  # (the mode holds from the next resolve on: shade the frame again)
            exact = st.present()
        finally:
            st.set_colour_exact(False)
        assert np.array_equal(exact, want), f"{what}: exact colour arithmetic differs from the oracle's image"
    else:
        assert np.array_equal(got, want), f"{what}: presented image"


def check_present_digest(st, want_sha, what=""):
    """The same against a SHA-256 of the oracle's presented image (full-size frames): the exact arithmetic must hash to
    it, the default (fused) one must stay within 1 LSB of the exact image."""
    import hashlib

    got = st.present()
    if not exact_by_default():
        st.set_colour_exact(True)
        try:
            st.resolve(_ffi.TR_OUT_RGBA8)  # (the mode holds from the next resolve on: shade the frame again)
            exact = st.present()
        finally:
            st.set_colour_exact(False)
        d = np.abs(got.astype(np.int16) - exact.astype(np.int16))
        assert d.max() <= 1, f"{what}: fused colour differs from the exact one by {d.max()} LSB"
    else:
        exact = got
    assert hashlib.sha256(np.ascontiguousarray(exact)).hexdigest() == want_sha, f"{what}: presented image differs from the oracle's digest"


def assert_parity(o, g, color_lsb=1):
    """Bit-exact depth / winner / triangle id; colour within `color_lsb` in the presented u8 domain."""
    assert g["flags"] == 0, f"status flags {g['flags']:#x}"
    zo, zg = o["z"], g["z"]
    assert zo.shape == zg.shape
    same = (zo.view(np.uint64) == zg.view(np.uint64)) | (np.isnan(zo) & np.isnan(zg))
    assert same.all(), f"{(~same).sum()} depth mismatches, first at {np.argwhere(~same)[:5].tolist()}"
    assert (o["seq"] == g["seq"]).all(), f"{(o['seq'] != g['seq']).sum()} triangle-id mismatches"
    assert (o["key"] == g["key"]).all(), f"{(o['key'] != g['key']).sum()} winner-key mismatches"
    d = np.abs(o["present"].astype(np.int16) - g["present"].astype(np.int16))
    assert d.max() <= color_lsb, f"presented colour differs by {d.max()} LSB"
    fused_lsb, fused_px = 0, 0
    if "present_fused" in g:  # the default (fused) colour arithmetic: the same bar, +-1 LSB
        df = np.abs(o["present"].astype(np.int16) - g["present_fused"].astype(np.int16))
        assert df.max() <= 1, f"fused colour arithmetic differs by {df.max()} LSB at {np.argwhere(df > 1)[:5].tolist()}"
        fused_lsb, fused_px = int(df.max()), int((df.max(axis=-1) > 0).sum())
    fo, fg = o["fb"], g["fb"]
    exact = (fo.view(np.uint64) == fg.view(np.uint64)) | (np.isnan(fo) & np.isnan(fg))
    return {"max_lsb": int(d.max()), "fused_max_lsb": fused_lsb, "fused_pixels_off_by_one": fused_px, "f64_colour_mismatches": int((~exact).sum()),
            "covered": int(np.isfinite(zo).sum())}
