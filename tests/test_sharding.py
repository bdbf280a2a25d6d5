"""N>1 path on CPU: world_size-2 gloo run of the view / band partition and the delivery gather.
Frames come from the oracle here (there is no GPU); what is under test is the sharding logic."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tinyrenderer_amd import sharding


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_band_rows_are_whole_tile_rows():
    for h in (16, 17, 480, 1080, 2160, 4320):
        for world in (1, 2, 4, 8):
            spans = [sharding.band_rows(h, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == h
            for i, (lo, hi) in enumerate(spans):
                assert (lo % 16 == 0 or lo == h) and (hi % 16 == 0 or hi == h)
                if i + 1 < world:
                    assert hi == spans[i + 1][0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _render_view(k, n_views, W, H):
    import oracle_ffi
    import scene_util as SU
    from tinyrenderer_amd import scenes

    mc = SU.component("eye_inner")
    mc.parent = None  # (the component cache is shared with other tests of the session)
    draws = [(mc, scenes.shader_for(mc, W, H, eye=scenes.orbit_eye(k, n_views, radius=0.6)))]
    o = SU.oracle_render(oracle_ffi, W, H, draws)
    return np.ascontiguousarray(o["present"]).view(np.int32)[..., 0]  # (H, W) packed RGBA8


def _worker(rank, world, port, n_views, W, H, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(n_views, rank, world)
        local = torch.from_numpy(np.stack([_render_view(k, n_views, W, H) for k in range(lo, hi)]))
        frames = sharding.gather_frames(local, n_views, dst=0)
        # band sharding of view 0: every rank renders the frame and keeps only its band
        y_lo, y_hi = sharding.band_rows(H, rank, world)
        band = torch.from_numpy(_render_view(0, n_views, W, H)[y_lo:y_hi].copy())
        full = sharding.gather_bands(band, H, dst=0)
        # bands of unequal height (what balanced_bands gives), one of them empty when there are more than two ranks
        cuts = [0, 16] + [H] * (world - 1)
        spans = list(zip(cuts[:-1], cuts[1:]))
        a_lo, a_hi = spans[rank]
        uneven = sharding.gather_bands(torch.from_numpy(_render_view(0, n_views, W, H)[a_lo:a_hi].copy()), H, dst=0, spans=spans)
        if rank == 0:
            np.savez(out_path, frames=frames.numpy(), full=full.numpy(), uneven=uneven.numpy())
        else:
            assert frames is None and full is None and uneven is None
    finally:
        dist.destroy_process_group()


def test_world2_gloo_views_and_bands(tmp_path):
    n_views, W, H = 5, 64, 40
    out_path = str(tmp_path / "out.npz")
    mp.start_processes(_worker, args=(2, _free_port(), n_views, W, H, out_path), nprocs=2, join=True,
                       start_method="spawn")
    got = np.load(out_path)
    want = np.stack([_render_view(k, n_views, W, H) for k in range(n_views)])
    assert np.array_equal(got["frames"], want)
    assert np.array_equal(got["full"], want[0])
    assert np.array_equal(got["uneven"], want[0])
    assert (want != np.int32(-16777216)).any()  # something other than the clear colour was drawn


def test_balanced_bands_cover_the_frame_and_follow_the_weights():
    """sharding.balanced_bands: contiguous bands of whole tile rows, none empty while rows remain,
    weights about equal."""
    import random

    from tinyrenderer_amd import sharding as S

    rnd = random.Random(5)
    for world in (1, 2, 3, 8):
        for n in (1, 3, 8, 17, 270):
            w = [rnd.random() * 10 for _ in range(n)]
            height = n * 16 - 5
            b = S.balanced_bands(w, world, height=height)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == height
            assert all(x[1] == y[0] for x, y in zip(b[:-1], b[1:])) and all(x[0] <= x[1] for x in b)
            if n >= world:
                assert all(x[0] < x[1] for x in b)
    w = [1] * 10 + [10] * 10 + [1] * 10  # the middle third carries 10/12 of the weight
    bands = S.balanced_bands(w, 4)
    per = [sum(w[lo // 16: hi // 16]) for lo, hi in bands]
    assert max(per) <= 1.35 * sum(w) / 4, per
    assert S.balanced_bands([1.0] * 8, 4) == [S.band_rows(128, r, 4) for r in range(4)]
