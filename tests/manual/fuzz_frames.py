#!/usr/bin/env python3
"""Fuzz of the frame state machine (tr_clear / tr_draw / tr_resolve / tr_read_* on ONE context over many frames): random
sequences of clears, draws of random small meshes at random places, resolves with random plane masks, late requests for
planes a resolve did not ask for, draws that extend a resolved frame, repeated resolves, switches between the exact and the
fused colour arithmetic -- after every resolve what is read must be the oracle's frame of the draws made since the last
clear.  This is what the self-cleaning planes, the boxed shade launch and the host-side plane bookkeeping have
to survive.  python tests/manual/fuzz_frames.py [n_sequences=40] [seed0=8000] [mode=0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    import oracle_ffi
    import scene_util as SU
    import test_gpu_parity as T
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import BuiltinShader, ModelRendererState, clear_screen, render_mesh_component

    import threading
    from concurrent.futures import ThreadPoolExecutor

    lock = threading.Lock()
    totals = {"bad": 0, "steps": 0, "done": 0}

    def run_one(k):
        bad = 0
        steps = 0
        rng = np.random.default_rng(seed0 + k)
        W = int(rng.integers(16, 400))
        H = int(rng.integers(16, 300))
        pool = [T._random_scene(rng, int(rng.choice([1, 5, 30, 120, 400])), W, H, spread=float(rng.choice([0.5, 0.9, 1.3])))[0][0] for _ in range(3)]

        def shader():
            # a random place on the screen, so that what is painted moves from frame to frame
            s = float(rng.uniform(0.3, 1.0))
            mv = np.array([[s, 0, 0, rng.uniform(-0.6, 0.6)], [0, s, 0, rng.uniform(-0.6, 0.6)], [0, 0, 1, 0], [0, 0, 0, 1.0]])
            return BuiltinShader(np.eye(4), mv, (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), 2.0))

        st = ModelRendererState(W, H, raster_mode=mode)
        draws = []
        log = []
        exact = False
        band = [0, H]
        with_bands = os.environ.get("FUZZ_BANDS", "0") != "0"
        # FUZZ_PEER=1: a second context on the device comes and goes between the operations -- the library sizes its raster
        # workgroups and the slot sequence's chunks by whether the context is alone, also in the middle of a frame
        with_peer = os.environ.get("FUZZ_PEER", "0") != "0"
        peer = None

        def check(what, mask_read):
            try:
                o = SU.oracle_render(oracle_ffi, W, H, draws)
            except AssertionError:
                return True  # (the reference would have panicked on a texture index: nothing to compare)
            y_lo, y_hi = band
            if (y_lo, y_hi) != (0, H):
                # a band context: its rows are the whole frame's, the others read back as cleared
                out = np.ones(H, dtype=bool)
                out[y_lo:y_hi] = False
                o = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in o.items()}
                o["z"][:, out] = -np.inf
                o["key"][:, out] = np.uint64(0xFFFFFFFFFFFFFFFF)
                o["seq"][:, out] = -1
                o["fb"][:, out] = np.array([0.0, 0.0, 0.0, 255.0])
                o["present"][out[::-1]] = np.array([0, 0, 0, 255], dtype=np.uint8)
            if mask_read & _ffi.TR_OUT_Z:
                z = st.z_buffer[..., 0]
                same = (z.view(np.uint64) == o["z"].view(np.uint64)) | (np.isnan(z) & np.isnan(o["z"]))
                assert same.all(), f"{what}: {(~same).sum()} depth mismatches"
            if mask_read & _ffi.TR_OUT_KEY:
                assert np.array_equal(st.winner_keys(), o["key"]), f"{what}: winner keys"
            if mask_read & _ffi.TR_OUT_TRI_ID:
                tri = st.triangle_ids().astype(np.int64)
                tri[tri == 0xFFFFFFFF] = -1
                assert np.array_equal(tri, o["seq"]), f"{what}: triangle ids"
            if mask_read & _ffi.TR_OUT_RGBA_F64:
                fb = st.frame_buffer
                assert ((fb.view(np.uint64) == o["fb"].view(np.uint64)) | (np.isnan(fb) & np.isnan(o["fb"]))).all(), f"{what}: f64 colour"
            if mask_read & _ffi.TR_OUT_RGBA8:
                d = np.abs(st.present().astype(np.int16) - o["present"].astype(np.int16))
                if d.max() > (0 if exact else 1):
                    w = np.argwhere(d.max(axis=-1) > 1)
                    got = st.present()
                    raise AssertionError(f"{what}: presented colour off by {d.max()} LSB on {len(w)} pixels, rows {w[:, 0].min()}..{w[:, 0].max()} "
                                         f"columns {w[:, 1].min()}..{w[:, 1].max()}; first: got {got[tuple(w[0])].tolist()} want {o['present'][tuple(w[0])].tolist()}")
            assert st.status_flags() == 0, f"{what}: flags {st.status_flags():#x}"
            return True

        try:
            clear_screen(st)
            for step in range(int(rng.integers(6, 20))):
                op = rng.choice(["clear", "draw", "draw", "resolve", "resolve", "late", "exact"] + (["peer"] if with_peer else []))
                if op == "clear":
                    clear_screen(st)
                    draws = []
                    log.append("clear")
                    if with_bands and rng.random() < 0.5:
                        rows = (H + 15) // 16
                        a_ = int(rng.integers(0, rows))
                        b_ = int(rng.integers(a_ + 1, rows + 1))
                        band = [a_ * 16, min(b_ * 16, H)] if rng.random() < 0.8 else [0, H]
                        st.set_band(band[0], band[1])
                        log.append(f"band {band}")
                elif op == "draw" and len(draws) < 4:
                    mc = pool[int(rng.integers(0, 3))]
                    sh = shader()
                    render_mesh_component(st, mc, sh)
                    draws.append((mc, sh))
                    log.append("draw")
                elif op == "resolve":
                    mask = _ffi.TR_OUT_RGBA8
                    for bit in (_ffi.TR_OUT_Z, _ffi.TR_OUT_KEY, _ffi.TR_OUT_TRI_ID, _ffi.TR_OUT_RGBA_F64):
                        if rng.random() < 0.3:
                            mask |= bit
                    if mask & _ffi.TR_OUT_RGBA_F64:
                        pass  # (such a resolve always shades exactly)
                    st.resolve(mask)
                    log.append(f"resolve {mask:#x}")
                    was = exact
                    exact = exact or bool(mask & _ffi.TR_OUT_RGBA_F64)
                    check(" > ".join(log[-8:]), mask)
                    exact = was
                    steps += 1
                elif op == "late" and log and log[-1].startswith("resolve"):
                    bit = int(rng.choice([_ffi.TR_OUT_Z, _ffi.TR_OUT_KEY, _ffi.TR_OUT_TRI_ID]))
                    log.append(f"late {bit:#x}")
                    check(" > ".join(log[-8:]), bit)
                    steps += 1
                elif op == "peer":
                    if peer is None:
                        peer = ModelRendererState(32, 32)
                    else:
                        peer.close()
                        peer = None
                    log.append("peer" + ("+" if peer is not None else "-"))
                elif op == "exact":
                    exact = not exact
                    st.set_colour_exact(exact)
                    log.append(f"exact={int(exact)}")
        except AssertionError as e:
            bad += 1
            print(f"FAIL seed {seed0 + k} mode {mode} {W}x{H}: {str(e)[:300]}\n     full sequence: " + " > ".join(log), flush=True)
        finally:
            st.close()
            if peer is not None:
                peer.close()
        with lock:
            totals["bad"] += bad
            totals["steps"] += steps
            totals["done"] += 1
            if totals["done"] % 10 == 0:
                print(f"{totals['done']} sequences done ({totals['steps']} checks), {totals['bad']} failures", flush=True)
    # FUZZ_THREADS=k: k host threads, each with a context of its own, at the same time (the library's per-device state:
    # the count of live contexts, the static staging array, the loaders' thread-local error text)
    n_threads = int(os.environ.get("FUZZ_THREADS", "1"))
    if n_threads > 1:
        with ThreadPoolExecutor(n_threads) as ex:
            list(ex.map(run_one, range(n)))
    else:
        for k in range(n):
            run_one(k)
    bad, steps = totals["bad"], totals["steps"]
    print(f"fuzz_frames: {n} sequences, {steps} checks, mode {mode}, {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
