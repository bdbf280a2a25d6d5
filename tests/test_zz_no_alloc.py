"""tr_read_rgba8_static (the one entry point that allocates nothing, include/tinyrenderer_hip.h) against
tr_read_rgba8, through the C ABI: same bytes, two calls in a row with the second on a smaller frame (the static
staging array keeps the first call's image), no device memory taken by the call, and its refusals."""
import ctypes as C

import numpy as np
import pytest

import scene_util as SU
from tinyrenderer_amd import _ffi, scenes
from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

pytestmark = pytest.mark.gpu


def _frame(W, H):
    head = SU.component("head")
    head.transform.set_rotation((0.025, 0.0, 0.0))
    st = ModelRendererState(W, H)
    clear_screen(st)
    render_mesh_component(st, head, scenes.shader_for(head, W, H))
    return st


def _free_bytes(hip):
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_static_read_matches_the_allocating_read():
    lib = _ffi.load()
    hip = C.CDLL("libamdhip64.so.7")  # (the runtime the library is linked against)
    # the largest frame the static array holds, then a smaller one, then a non-multiple-of-anything one
    for W, H in ((3840, 2160), (640, 360), (333, 517)):
        st = _frame(W, H)
        try:
            got = np.full((H, W, 4), 0xA5, dtype=np.uint8)
            # not resolved yet: the call resolves nothing and says so
            assert lib.tr_read_rgba8_static(st._ctx, got.ctypes.data, 1) == -1  # TR_ERR_INVALID
            assert b"tr_resolve" in lib.tr_last_error(st._ctx)
            st.resolve(_ffi.TR_OUT_RGBA8)
            st.synchronize()
            # (the first launch of a kernel and the first copy to pageable memory make the HIP runtime set up
            # pools of its own, 2 MB here: not the library's doing, and done once per process)
            assert lib.tr_read_rgba8_static(st._ctx, got.ctypes.data, 1) == 0
            for flip in (1, 0):
                got[:] = 0xA5
                free0 = _free_bytes(hip)
                assert lib.tr_read_rgba8_static(st._ctx, got.ctypes.data, flip) == 0
                assert _free_bytes(hip) == free0, "tr_read_rgba8_static took device memory"
                want = np.empty((H, W, 4), dtype=np.uint8)
                assert lib.tr_read_rgba8(st._ctx, want.ctypes.data, flip) == 0  # (allocates its flip buffer on first use)
                assert np.array_equal(got, want), (W, H, flip)
            assert (got[..., 3] == 255).all() and got[..., :3].any()
        finally:
            st.close()


def test_static_read_refuses_frames_beyond_its_array():
    lib = _ffi.load()
    st = _frame(3856, 2160)
    try:
        st.resolve(_ffi.TR_OUT_RGBA8)
        out = np.empty((2160, 3856, 4), dtype=np.uint8)
        assert lib.tr_read_rgba8_static(st._ctx, out.ctypes.data, 1) == -6  # TR_ERR_SIZE
        assert b"3840 x 2160" in lib.tr_last_error(st._ctx)
        assert lib.tr_read_rgba8(st._ctx, out.ctypes.data, 1) == 0
    finally:
        st.close()
