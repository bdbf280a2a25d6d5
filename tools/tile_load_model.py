#!/usr/bin/env python3
"""Where the samples of a frame land, and what that means for a tiled (LDS) raster: a CPU model.

Enumerates the reference's sample lattice (model_renderer/mod.rs:360-452, segment_3d/mod.rs:19-115) of the
african_head frame with numpy (not bit-exact: this is a load model, not a checker), counts samples per pixel and
prints, per candidate tile shape, how the samples spread over the tiles: number of non-empty tiles, mean / p99 /
max samples per tile, and the time a launch would take if a tile were one workgroup's serial job
(max tile / rate) against a perfectly balanced one (total / (rate x slots)).

    python tools/tile_load_model.py [W H]
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def viewport_triangles(W, H):
    import scene_util as SU
    from tinyrenderer_amd import scenes

    head = SU.component("head")
    head.transform.set_rotation((0.025, 0.0, 0.0))
    sh = scenes.shader_for(head, W, H)
    pmv = np.asarray(sh.pmv(), dtype=np.float64).reshape(4, 4)
    obj = head.mesh
    v = np.concatenate([obj.vertices, np.ones((obj.vertices.shape[0], 1))], axis=1)
    # helpers stores matrices row-major as nested lists or column-major? try both and keep the one with w > 0
    for M in (pmv, pmv.T):
        g = v @ M.T
        if (g[:, 3] > 0).all():
            break
    ndc = g[:, :3] / g[:, 3:4]
    vp = np.empty_like(ndc)
    vp[:, 0] = (ndc[:, 0] + 1.0) * (W - 1) / 2.0
    vp[:, 1] = (ndc[:, 1] + 1.0) * (H - 1) / 2.0
    vp[:, 2] = ndc[:, 2]
    tri = obj.triangles.astype(np.int64)[:, [0, 3, 6]]
    return vp[tri]  # (T, 3, 3)


def enumerate_counts(P, W, H):
    cnt = np.zeros((H, W), dtype=np.int64)
    total = 0
    per_tri = np.zeros(P.shape[0], dtype=np.int64)
    for k in range(P.shape[0]):
        p = P[k]
        o = np.argsort(-p[:, 1], kind="stable")
        s0, s1, s2 = p[o[0]], p[o[1]], p[o[2]]
        den = s2[1] - s0[1]
        if den == 0:
            continue
        t = (s1[1] - s0[1]) / den
        sp = s0 + t * (s2 - s0)
        for apex in (s0, s2):
            m = max(abs(s1[0] - apex[0]), abs(s1[1] - apex[1]), abs(sp[0] - apex[0]), abs(sp[1] - apex[1]))
            res = max(2, int(math.ceil(m)))
            ti = np.arange(res + 1) * (1.0 / res)
            A = apex[None, :] + ti[:, None] * (s1 - apex)[None, :]
            B = apex[None, :] + ti[:, None] * (sp - apex)[None, :]
            D = B - A
            hres = np.maximum(2, np.ceil(np.maximum(np.abs(D[:, 0]), np.abs(D[:, 1]))).astype(np.int64) + 1)
            n = hres + 1
            idx = np.repeat(np.arange(res + 1), n)
            j = np.arange(n.sum()) - np.repeat(np.cumsum(n) - n, n)
            s = j * (1.0 / hres[idx])
            x = np.floor(A[idx, 0] + s * D[idx, 0] + 0.5).astype(np.int64)
            y = np.floor(A[idx, 1] + s * D[idx, 1] + 0.5).astype(np.int64)
            ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
            np.add.at(cnt, (y[ok], x[ok]), 1)
            total += int(n.sum())
            per_tri[k] += int(n.sum())
    return cnt, total, per_tri


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 2 else 3840
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
    P = viewport_triangles(W, H)
    cnt, total, per_tri = enumerate_counts(P, W, H)
    print(f"{W}x{H}: {P.shape[0]} triangles, {total} samples, {int((cnt > 0).sum())} covered pixels, "
          f"max {cnt.max()} samples on one pixel, heaviest triangle {per_tri.max()} samples "
          f"(mean {per_tri.mean():.0f})")
    # a workgroup of 256 lanes at the global raster's no-memory rate: 6.57 M samples in 40.5 us over ~1024 resident
    # workgroups = 0.16 samples/ns per ... -> expressed per workgroup below
    rate_wg = 6.57e6 / 40.5e-6 / 1024.0  # samples/s one 256-lane workgroup sustains when the GPU is full of them
    print(f"(one 256-lane workgroup ~ {rate_wg / 1e9:.3f} G samples/s)")
    print(f"{'tile':>9} {'tiles':>7} {'mean':>8} {'p99':>8} {'max':>8} {'max/mean':>8} {'t(max tile) us':>14} {'balanced us':>12}")
    for tw, th in ((16, 16), (32, 8), (32, 16), (32, 32), (64, 8), (64, 16), (64, 32), (64, 64), (128, 64)):
        Hp, Wp = (H + th - 1) // th * th, (W + tw - 1) // tw * tw
        c = np.zeros((Hp, Wp), dtype=np.int64)
        c[:H, :W] = cnt
        t = c.reshape(Hp // th, th, Wp // tw, tw).sum(axis=(1, 3)).ravel()
        t = t[t > 0]
        bal = total / (rate_wg * 1024.0) * 1e6
        print(f"{tw:>4}x{th:<4} {t.size:>7} {t.mean():>8.0f} {np.percentile(t, 99):>8.0f} {t.max():>8} "
              f"{t.max() / t.mean():>8.1f} {t.max() / rate_wg * 1e6:>14.1f} {bal:>12.1f}")


if __name__ == "__main__":
    main()
