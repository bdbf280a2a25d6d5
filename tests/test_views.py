"""tr_render_views (batched views of one mesh, include/tinyrenderer_hip.h) against the oracle run once per view:
the reference renders one view per iteration of its loop (src/main.rs:57-116,272-273), so view v of a batch must
equal clear_screen + render_mesh_component under camera v -- depth and winner key bit for bit (tri_seq counted
over the batch: view v's triangle k is v * n_triangles + k), colour bit for bit too."""
import json
import os

import numpy as np
import pytest

import big_cases as B
import scene_util as SU
from tinyrenderer_amd import _ffi, scenes
from tinyrenderer_amd.model_renderer import BuiltinShader, MeshComponent, ModelRendererState, render_views

pytestmark = pytest.mark.gpu

HASHES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hashes.json")))
MASK = _ffi.TR_OUT_RGBA8 | _ffi.TR_OUT_Z | _ffi.TR_OUT_KEY | _ffi.TR_OUT_RGBA_F64 | _ffi.TR_OUT_TRI_ID


def _local_key(key, view, n_tris):
    """Winner keys of slice `view` with tri_seq counted from the view's own first triangle."""
    out = key.copy()
    has = out != B.NONE_KEY
    out[has] -= np.uint64(view * n_tris) << np.uint64(36)
    return out


def _check_views(oracle, st, mc, shaders, W, H, mask=MASK):
    n_tris = mc.mesh.triangles.shape[0]
    render_views(st, mc, shaders, mask)
    assert st.status_flags() == 0
    total = {"samples": 0, "fragments": 0, "scanlines": 0, "covered": 0}
    for v, sh in enumerate(shaders):
        o = SU.oracle_render(oracle, W, H, [(mc, sh)])
        st.select_view(v)
        z = st.z_buffer[..., 0]
        same = (o["z"].view(np.uint64) == z.view(np.uint64)) | (np.isnan(o["z"]) & np.isnan(z))
        assert same.all(), f"view {v}: {(~same).sum()} depth mismatches"
        assert np.array_equal(_local_key(st.winner_keys(), v, n_tris), o["key"]), f"view {v}: winner keys"
        tri = st.triangle_ids().astype(np.int64)
        tri[tri == 0xFFFFFFFF] = -1
        assert np.array_equal(np.where(tri >= 0, tri - v * n_tris, -1), o["seq"]), f"view {v}: triangle ids"
        fb = st.frame_buffer
        assert ((fb.view(np.uint64) == o["fb"].view(np.uint64)) | (np.isnan(fb) & np.isnan(o["fb"]))).all(), f"view {v}: f64 colour"
        SU.check_present(st, o["present"], f"view {v}")
        for k in ("samples", "fragments", "scanlines"):
            total[k] += o["counters"][k]
        total["covered"] += int((o["key"] != B.NONE_KEY).sum())
    c = st.counters()
    got = (c["samples"], c["fragments"], c["scanlines"], c["covered_pixels"])
    assert got == (total["samples"], total["fragments"], total["scanlines"], total["covered"])
    assert c["triangles"] == n_tris * len(shaders)


def _orbit_shaders(mc, W, H, ks, n=4096):
    return [scenes.shader_for(mc, W, H, eye=scenes.orbit_eye(k, n)) for k in ks]


def test_orbit_batch_matches_the_oracle_view_by_view(oracle):
    W, H = 320, 180
    mc = SU.component("diablo")
    mc.transform.set_rotation((0.0, 0.0, 0.0))
    st = ModelRendererState(W, H, max_views=8)
    try:
        _check_views(oracle, st, mc, _orbit_shaders(mc, W, H, (0, 517, 1024, 2048, 3333, 4000, 77, 3000)), W, H)
        # a smaller batch on the same context: nothing of the eight views before may show through
        _check_views(oracle, st, mc, _orbit_shaders(mc, W, H, (2500, 100, 3900)), W, H)
        with pytest.raises(_ffi.TrError):
            st.select_view(3)
        # the frame loop's path: RGBA8 only (planes clean themselves), twice in a row, then planes asked for late
        shaders = _orbit_shaders(mc, W, H, (600, 1700, 2800, 3900, 5))
        for _ in range(2):
            render_views(st, mc, shaders, _ffi.TR_OUT_RGBA8)
            for v, sh in enumerate(shaders):
                st.select_view(v)
                SU.check_present(st, SU.oracle_render(oracle, W, H, [(mc, sh)])["present"], f"view {v}")
        st.select_view(4)
        o = SU.oracle_render(oracle, W, H, [(mc, shaders[4])])
        assert np.array_equal(_local_key(st.winner_keys(), 4, mc.mesh.triangles.shape[0]), o["key"])
        SU.check_present(st, o["present"])
    finally:
        st.close()


def test_head_batch_odd_size_and_small_triangle_path(oracle, monkeypatch):
    """An odd frame height (the last row of shade blocks of every view is half empty), a frame narrower than a
    shade block, and k_small forced onto the batch (TR_SMALL_DRAW=1)."""
    head = SU.component("head")
    head.transform.set_rotation((0.025, 0.0, 0.0))
    for W, H, small in ((97, 61, False), (200, 160, True), (24, 33, True)):
        monkeypatch.setenv("TR_MID_TRIS", "0" if (W, H) == (200, 160) else "1")  # (with and without k_mid)
        if small:
            monkeypatch.setenv("TR_SMALL_DRAW", "1")
        st = ModelRendererState(W, H, max_views=4)
        try:
            eyes = [(0.0, 0.0, 2.0), (1.0, 0.5, 2.0), (-1.5, -0.3, 1.5), (0.2, 1.0, 3.0)]
            _check_views(oracle, st, head, [scenes.shader_for(head, W, H, eye=e) for e in eyes], W, H)
        finally:
            st.close()
            monkeypatch.delenv("TR_SMALL_DRAW", raising=False)


def test_nan_depth_in_every_view_of_a_batch(oracle):
    """A horizontal degenerate triangle stores NaN at pixel (0,0) (SURVEY.md 8(c) KAT 7): in a batch that is pixel
    (0,0) of every slice, each with its own entry in the NaN side table."""
    from tinyrenderer_amd import helpers as Hh

    N = Hh.NONE_INDEX
    rng = np.random.default_rng(3)
    obj = Hh.Object([[-0.5, 0.25, 0.0], [0.5, 0.25, 0.0], [0.0, 0.25, 0.0], [-0.8, -0.8, 0.3], [0.8, -0.7, 0.1], [0.1, 0.9, -0.2]],
                    [[0, 0, 0], [1, 0, 0], [0.5, 1, 0]], np.zeros((0, 3)),
                    [[0, 0, N, 1, 1, N, 2, 2, N], [3, 0, N, 4, 1, N, 5, 2, N]])
    mc = MeshComponent(obj, rng.integers(0, 256, size=(32, 48, 4), dtype=np.uint8), rng.integers(0, 256, size=(16, 16, 4), dtype=np.uint8))
    W, H = 50, 37
    shaders = [BuiltinShader(np.eye(4), np.eye(4), (0.0, 0.1 * v, 2.0)) for v in range(3)]
    st = ModelRendererState(W, H, max_views=3)
    try:
        _check_views(oracle, st, mc, shaders, W, H)
        for v in range(3):
            st.select_view(v)
            assert np.isnan(st.z_buffer[0, 0, 0])
    finally:
        st.close()


def test_config5_orbit_views_1080p_as_one_batch():
    """configs[4] at full size: the four views of tests/test_fullsize.py rendered as ONE batch; digests equal the
    oracle's per-view digests (keys rebased to the view's own triangle count)."""
    ks = B.ORBIT_VIEWS
    W, H, draws = B.orbit_view(ks[0])
    mc = draws[0][0]
    n_tris = mc.mesh.triangles.shape[0]
    st = ModelRendererState(W, H, max_views=len(ks))
    try:
        render_views(st, mc, _orbit_shaders(mc, W, H, ks), _ffi.TR_OUT_RGBA8 | _ffi.TR_OUT_Z | _ffi.TR_OUT_KEY)
        assert st.status_flags() == 0
        tot = st.counters()
        for v, k in enumerate(ks):
            want = HASHES[f"c5_orbit_1080p_view{k}"]
            st.select_view(v)
            key = _local_key(st.winner_keys(_ffi.TR_LAYOUT_ROW_MAJOR), v, n_tris)
            assert B.sha(key) == want["key"] and int((key != B.NONE_KEY).sum()) == want["covered"], k
            assert B.sha(st.z_row_major().view(np.uint64)) == want["z"], k
            SU.check_present_digest(st, want["rgba8"], f"view {k}")
        for name in ("samples", "fragments", "scanlines"):
            assert tot[name] == sum(HASHES[f"c5_orbit_1080p_view{k}"]["counters"][name] for k in ks)
    finally:
        st.close()


def test_views_context_refusals():
    mc = SU.component("head")
    st = ModelRendererState(64, 48, max_views=2)
    try:
        from tinyrenderer_amd.model_renderer import clear_screen, render_mesh_component

        clear_screen(st)
        with pytest.raises(_ffi.TrError, match="tr_render_views"):
            render_mesh_component(st, mc, scenes.shader_for(mc, 64, 48))
        with pytest.raises(_ffi.TrError):
            render_views(st, mc, [scenes.shader_for(mc, 64, 48)] * 3)  # more views than slices
        with pytest.raises(_ffi.TrError):
            st.set_band(0, 16)
        # an empty frame: every slice is background
        st.resolve(_ffi.TR_OUT_RGBA8)
        for v in range(2):
            st.select_view(v)
            assert (st.present() == np.array([0, 0, 0, 255], dtype=np.uint8)).all()
    finally:
        st.close()
    plain = ModelRendererState(64, 48)
    try:
        with pytest.raises(_ffi.TrError, match="tr_create_views"):
            render_views(plain, mc, [scenes.shader_for(mc, 64, 48)])
    finally:
        plain.close()


def test_cheap_resolve_of_batches_of_changing_size_and_place(oracle):
    """The RGBA8-only resolve of a batch launches the union of the views' screen boxes (plus what earlier launches
    painted) in the slices of THIS batch only: a full batch, a short one somewhere else, a full one in a third place --
    every slice the last batch owns must show its view and nothing of the first batch's."""
    W, H = 320, 200
    mc = SU.component("eye_inner")
    mc.parent = None
    st = ModelRendererState(W, H, max_views=6)
    try:
        def shaders(n, dx, dist):
            return [scenes.shader_for(mc, W, H, eye=(dx + 0.05 * k, 0.1 * k - 0.2, dist)) for k in range(n)]

        for sh in (shaders(6, -0.9, 1.0), shaders(2, 0.9, 1.4), shaders(6, 0.3, 2.5)):
            render_views(st, mc, sh, _ffi.TR_OUT_RGBA8)
            for v, s_ in enumerate(sh):
                st.select_view(v)
                want = SU.oracle_render(oracle, W, H, [(mc, s_)])["present"]
                SU.check_present(st, want, f"view {v}")
    finally:
        st.close()


def test_more_block_rows_times_views_than_a_grid_dimension_holds(oracle):
    """The views shade launch has (block rows per view) x (views) workgroups rows: 551 x 120 = 66 120 > 65 535, the limit
    of a grid dimension -- the views are the grid's z.  A narrow, tall frame; the first, a middle and the last view
    against the oracle."""
    W, H, n = 32, 1102, 120
    mc = SU.component("head")
    mc.transform.set_rotation((0.025, 0.0, 0.0))
    shaders = [scenes.shader_for(mc, W, H, eye=(0.01 * (k - 60), 0.0, 2.0)) for k in range(n)]
    st = ModelRendererState(W, H, max_views=n)
    try:
        assert ((H + 1) // 2) * n > 65535
        for mask in (_ffi.TR_OUT_RGBA8, MASK):
            render_views(st, mc, shaders, mask)
            assert st.status_flags() == 0
            for v in (0, 59, n - 1):
                st.select_view(v)
                o = SU.oracle_render(oracle, W, H, [(mc, shaders[v])])
                assert (o["key"] != B.NONE_KEY).any()
                SU.check_present(st, o["present"], f"view {v}")
                if mask == MASK:
                    assert np.array_equal(_local_key(st.winner_keys(), v, mc.mesh.triangles.shape[0]), o["key"]), v
    finally:
        st.close()


def test_reference_loop_head_and_parented_eye_as_one_batch(oracle):
    """The reference's own scene (src/main.rs:57-116,272-273): every iteration turns the head a little further
    (`0.025 * time`) and draws it and then the eye whose parent it is (mod.rs:67-84).  Six iterations as ONE batch of
    two meshes (tr_render_views_multi), view by view against the oracle's two-draw frame: depth, winner keys (tri_seq
    continues across the two meshes), triangle ids, f64 colour, presented image, counters."""
    from tinyrenderer_amd.model_renderer import render_views_multi

    W, H = 320, 180
    head, eye = SU.component("head"), SU.component("eye_inner")
    eye.parent = head
    n_head, n_eye = head.mesh.triangles.shape[0], eye.mesh.triangles.shape[0]
    T = n_head + n_eye
    times = (1, 2, 7, 40, 63, 100)
    shaders = []
    for t in times:
        head.transform.set_rotation((0.025 * t, 0.0, 0.0))
        shaders.append([scenes.shader_for(head, W, H), scenes.shader_for(eye, W, H)])  # (matrices are captured now)
    st = ModelRendererState(W, H, max_views=8)
    try:
        for mask in (MASK, _ffi.TR_OUT_RGBA8):
            render_views_multi(st, [head, eye], shaders, mask)
            assert st.status_flags() == 0
            total = {"samples": 0, "fragments": 0, "scanlines": 0, "covered": 0}
            eye_won = 0
            for v, (sh_head, sh_eye) in enumerate(shaders):
                o = SU.oracle_render(oracle, W, H, [(head, sh_head), (eye, sh_eye)])
                st.select_view(v)
                SU.check_present(st, o["present"], f"view {v}")
                if mask == MASK:
                    st.resolve(MASK)  # (check_present left an RGBA8-only resolve behind)
                    z = st.z_buffer[..., 0]
                    assert ((o["z"].view(np.uint64) == z.view(np.uint64)) | (np.isnan(o["z"]) & np.isnan(z))).all(), v
                    assert np.array_equal(_local_key(st.winner_keys(), v, T), o["key"]), f"view {v}: winner keys"
                    tri = st.triangle_ids().astype(np.int64)
                    tri[tri == 0xFFFFFFFF] = -1
                    assert np.array_equal(np.where(tri >= 0, tri - v * T, -1), o["seq"]), f"view {v}: triangle ids"
                    fb = st.frame_buffer
                    assert ((fb.view(np.uint64) == o["fb"].view(np.uint64)) | (np.isnan(fb) & np.isnan(o["fb"]))).all(), v
                    eye_won += int((o["seq"] >= n_head).sum())
                for k in ("samples", "fragments", "scanlines"):
                    total[k] += o["counters"][k]
                total["covered"] += int((o["key"] != B.NONE_KEY).sum())
            c = st.counters()
            assert (c["samples"], c["fragments"], c["scanlines"], c["covered_pixels"]) == \
                (total["samples"], total["fragments"], total["scanlines"], total["covered"])
            assert c["triangles"] == T * len(shaders)
            if mask == MASK:
                assert eye_won > 0, "no eye pixel won in any view: the second mesh was not exercised"
        # a one-mesh batch on the same context afterwards: nothing of the two-mesh batch may show through
        head.transform.set_rotation((0.3, 0.0, 0.0))
        one = [scenes.shader_for(head, W, H)]
        render_views(st, head, one, _ffi.TR_OUT_RGBA8)
        st.select_view(0)
        SU.check_present(st, SU.oracle_render(oracle, W, H, [(head, one[0])])["present"], "one mesh after two")
        with pytest.raises(_ffi.TrError):
            render_views_multi(st, [head] * 5, [[one[0]] * 5])  # more than four meshes
    finally:
        eye.parent = None
        head.transform.set_rotation((0.025, 0.0, 0.0))
        st.close()
