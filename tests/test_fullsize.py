"""BASELINE.json configs[2], [3], [4] at FULL size on the GPU, against SHA-256 digests the oracle produced in the
build container (tests/golden/make_golden.py --big; hashes.json c3_* / c4_* / c5_*): depth, winner key and the
presented RGBA8 bit for bit, counters equal.  configs[2] is also compared with the oracle directly (it takes the
CPU 0.2 s), and is rendered the way bench.py times it: frames in flight on four contexts, no host wait."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import big_cases as B
import scene_util as SU
from tinyrenderer_amd import _ffi, sharding

pytestmark = pytest.mark.gpu

HASHES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hashes.json")))


def test_config3_head_4k_against_oracle_and_digest(oracle):
    W, H, draws = B.head_4k()
    o = SU.oracle_render(oracle, W, H, draws)
    g = SU.gpu_render(W, H, draws)
    info = SU.assert_parity(o, g)
    assert info["f64_colour_mismatches"] == 0 and info["covered"] == 1541278
    for k in ("samples", "fragments", "scanlines"):
        assert g["counters"][k] == o["counters"][k] == HASHES["c3_head_4k"]["counters"][k]
    assert B.digest_row_major(g["z"].T, g["key"].T, g["present"], g["counters"]) == HASHES["c3_head_4k"]


@pytest.mark.parametrize("mask", [_ffi.TR_OUT_RGBA8, _ffi.TR_OUT_RGBA8 | _ffi.TR_OUT_Z | _ffi.TR_OUT_KEY],
                         ids=["rgba8_only", "with_planes"])
def test_config3_four_contexts_in_flight(mask):
    """bench.py's timed mode: frames issued round-robin on four live contexts (four HIP streams; k_raster then
    runs its 128-thread instantiation) with no synchronisation in between; afterwards EVERY context must hold the
    frame.  With RGBA8 alone the planes clean themselves (the path the benchmark times); with z / key planes the
    keys of the frames that ran concurrently are compared too."""
    from tinyrenderer_amd.model_renderer import ModelRendererState, make_draw_params

    W, H, draws = B.head_4k()
    want = HASHES["c3_head_4k"]
    lib = _ffi.load()
    states = [ModelRendererState(W, H) for _ in range(4)]
    try:
        prepared = [[(s._mesh_handle(mc.mesh), make_draw_params(s, sh, mc.texture, mc.normal_map)) for mc, sh in draws]
                    for s in states]
        for rnd in range(3):  # 12 frames, 3 per context, none waited for
            for s, prep in zip(states, prepared):
                assert lib.tr_clear(s._ctx) == 0
                for mesh_h, params in prep:
                    assert lib.tr_draw(s._ctx, mesh_h, C.byref(params)) == 0
                assert lib.tr_resolve(s._ctx, mask) == 0
        for s in states:
            s.synchronize()
        for k, s in enumerate(states):
            assert s.status_flags() == 0
            c = s.counters()
            for name in ("samples", "fragments", "scanlines"):
                assert c[name] == want["counters"][name], (k, name)
            assert c["covered_pixels"] == want["covered"], k
            SU.check_present_digest(s, want["rgba8"], f"context {k}")
            if mask != _ffi.TR_OUT_RGBA8:
                assert B.sha(s.winner_keys(_ffi.TR_LAYOUT_ROW_MAJOR)) == want["key"], k
                assert B.sha(s.z_row_major().view(np.uint64)) == want["z"], k
    finally:
        for s in states:
            s.close()


def test_config5_orbit_views_1080p():
    """configs[4]: views of the 4096-view orbit of diablo3_pose at 1920x1080, one context reused for all of them
    (what a rank of the batched-view job does), plus the frame SURVEY.md section 6 counted independently."""
    from tinyrenderer_amd.model_renderer import ModelRendererState

    st = ModelRendererState(1920, 1080)
    try:
        for k in B.ORBIT_VIEWS:
            W, H, draws = B.orbit_view(k)
            z, key, present, counters = B.gpu_planes(W, H, draws, state=st)
            assert B.digest_row_major(z, key, present, counters) == HASHES[f"c5_orbit_1080p_view{k}"], k
        W, H, draws = B.diablo_1080p_frame1()
        got = B.digest_row_major(*B.gpu_planes(W, H, draws, state=st))
        assert got == HASHES["c5_diablo_1080p_frame1"]
        assert got["counters"]["samples"] == 1412616 and got["covered"] == 192579
    finally:
        st.close()


def test_config4_subdivided_10m_triangles_8k_whole_and_8_bands():
    """configs[3] at full size: african_head subdivided 6x (10 207 232 triangles) at 7680x4320, rendered whole and
    as the 8 tile-row bands of an 8-GPU node (tr_set_band; here one after the other on one GPU), stitched."""
    from tinyrenderer_amd.model_renderer import ModelRendererState

    W, H, draws = B.head_subdivided(6, 7680, 4320)
    assert draws[0][0].mesh.triangles.shape[0] == 10207232
    want = HASHES["c4_head_sub6_8k"]
    st = ModelRendererState(W, H)
    try:
        z, key, present, counters = B.gpu_planes(W, H, draws, state=st)
        assert B.digest_row_major(z, key, present, counters) == want
        del z, key, present
        zs = np.full((H, W), -np.inf)
        keys = np.full((H, W), B.NONE_KEY, dtype=np.uint64)
        img = np.zeros((H, W, 4), dtype=np.uint8)
        total = {"samples": 0, "fragments": 0, "scanlines": 0}
        for r in range(8):
            y_lo, y_hi = sharding.band_rows(H, r, 8)
            z, key, present, counters = B.gpu_planes(W, H, draws, band=(y_lo, y_hi), state=st)
            outside = np.ones(H, dtype=bool)
            outside[y_lo:y_hi] = False
            assert (key[outside] == B.NONE_KEY).all(), f"band {r} wrote outside its rows"
            zs[y_lo:y_hi], keys[y_lo:y_hi] = z[y_lo:y_hi], key[y_lo:y_hi]
            img[H - y_hi: H - y_lo] = present[H - y_hi: H - y_lo]  # presented rows are flipped
            total["fragments"] += counters["fragments"]
        assert B.sha(zs.view(np.uint64)) == want["z"] and B.sha(keys) == want["key"] and B.sha(img) == want["rgba8"]
        assert total["fragments"] == want["counters"]["fragments"]  # every in-band sample counted exactly once
    finally:
        st.close()
