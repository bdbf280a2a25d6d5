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


def test_static_read_from_two_contexts_at_once_is_refused_not_raced():
    """All contexts of a GPU share the staging array: two host threads, one context each (different frames), calling at the
    same time.  A call either delivers exactly its own context's image or is refused with TR_ERR_BUSY and succeeds when
    made again; never a mixture of the two frames."""
    import threading

    lib = _ffi.load()
    W, H = 1920, 1080
    head = SU.component("head")
    states, wants = [], []
    for rot in (0.025, 1.3):
        head.transform.set_rotation((rot, 0.0, 0.0))
        st = ModelRendererState(W, H)
        clear_screen(st)
        render_mesh_component(st, head, scenes.shader_for(head, W, H))
        st.resolve(_ffi.TR_OUT_RGBA8)
        st.synchronize()
        want = np.empty((H, W, 4), dtype=np.uint8)
        assert lib.tr_read_rgba8(st._ctx, want.ctypes.data, 1) == 0
        states.append(st)
        wants.append(want)
    assert not np.array_equal(wants[0], wants[1])
    results = [{"ok": 0, "busy": 0, "bad": 0, "other": []} for _ in states]
    start = threading.Barrier(2)

    def worker(k):
        out = np.empty((H, W, 4), dtype=np.uint8)
        import time

        start.wait()
        # (until this thread has had five reads of its own, a few hundred tries at most: a refused caller steps back for a
        # moment, or the interpreter lock hands it its turns exactly while the other thread is inside the call)
        for _ in range(400):
            if results[k]["ok"] >= 5:
                break
            out[:] = 0
            rc = lib.tr_read_rgba8_static(states[k]._ctx, out.ctypes.data, 1)  # (ctypes releases the GIL during the call)
            if rc == 0:
                results[k]["ok" if np.array_equal(out, wants[k]) else "bad"] += 1
            elif rc == -7:  # TR_ERR_BUSY
                results[k]["busy"] += 1
                time.sleep(0.0007 * (1 + k))
            else:
                results[k]["other"].append(rc)

    try:
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for k, r in enumerate(results):
            assert r["bad"] == 0 and not r["other"], (k, r)
            assert r["ok"] > 0, (k, r)
        # a refused call simply succeeds when made again
        out = np.empty((H, W, 4), dtype=np.uint8)
        assert lib.tr_read_rgba8_static(states[0]._ctx, out.ctypes.data, 1) == 0 and np.array_equal(out, wants[0])
        print("static read, two contexts:", results)
    finally:
        for st in states:
            st.close()
