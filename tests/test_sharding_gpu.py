"""The N>1 DATA path with real GPU frames, two ranks on ONE device (gloo; RCCL refuses two ranks on a device): every
rank renders its tile-row band (tr_set_band) / its block of views (tr_render_views), hands the context's device plane
to torch without a copy (sharding.device_frames over tr_device_ptr_rgba8, streams ordered by to_torch / to_renderer),
and the single delivery gather stitches them on rank 0, where they must equal the whole frame rendered by one
context and the oracle's frames."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(W, H):
    import scene_util as SU
    from tinyrenderer_amd import scenes

    head = SU.component("head")
    head.transform.set_rotation((0.025, 0.0, 0.0))
    return head, scenes.shader_for(head, W, H)


def _worker(rank, world, port, W, H, n_views, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import scene_util as SU
        from tinyrenderer_amd import _ffi, scenes, sharding
        from tinyrenderer_amd.model_renderer import (ModelRendererState, clear_screen, render_mesh_component,
                                                     render_views)

        # ---- bands of one frame, two frames in a row with the bands re-cut in between
        head, sh = _scene(W, H)
        st = ModelRendererState(W, H, device=0)
        plane = sharding.device_frames(st)[0]  # (H, W) int32, the context's own memory
        fulls = []
        bands = [sharding.band_rows(H, r, world) for r in range(world)]
        for frame in range(2):
            y_lo, y_hi = bands[rank]
            clear_screen(st)
            st.set_band(y_lo, y_hi)
            render_mesh_component(st, head, sh)
            st.resolve(_ffi.TR_OUT_RGBA8)
            sharding.to_torch(st)
            full = sharding.gather_bands(plane[y_lo:y_hi], H, dst=0, spans=bands)
            sharding.to_renderer(st)
            fulls.append(None if full is None else full.cpu().numpy())
            bands = sharding.rebalance(bands, [1.0 + r for r in range(world)], H)  # (any new cut will do here)
        st.close()
        # ---- a block of views per rank, rendered as one batch, gathered in view order
        mc = SU.component("diablo")
        mc.transform.set_rotation((0.0, 0.0, 0.0))
        lo, hi = sharding.shard_range(n_views, rank, world)
        vs = ModelRendererState(W, H, device=0, max_views=max(2, hi - lo))
        shaders = [scenes.shader_for(mc, W, H, eye=scenes.orbit_eye(k, n_views)) for k in range(lo, hi)]
        render_views(vs, mc, shaders, _ffi.TR_OUT_RGBA8)
        sharding.to_torch(vs)
        frames = sharding.gather_frames(sharding.device_frames(vs, hi - lo), n_views, dst=0)
        sharding.to_renderer(vs)
        vs.close()
        if rank == 0:
            np.savez(out_path, full0=fulls[0], full1=fulls[1], frames=frames.cpu().numpy())
        else:
            assert frames is None and fulls == [None, None]
    finally:
        dist.destroy_process_group()


def test_world2_real_gpu_bands_and_views_stitch_to_the_whole_frame(oracle, tmp_path):
    import scene_util as SU
    from tinyrenderer_amd import scenes

    W, H, n_views = 320, 208, 5
    out_path = str(tmp_path / "out.npz")
    mp.start_processes(_worker, args=(2, _free_port(), W, H, n_views, out_path), nprocs=2, join=True, start_method="spawn")
    got = np.load(out_path)
    head, sh = _scene(W, H)
    whole = SU.gpu_render(W, H, [(head, sh)])["present"]  # one context, no band
    o = SU.oracle_render(oracle, W, H, [(head, sh)])["present"]
    assert np.array_equal(whole, o)
    for name in ("full0", "full1"):  # the device plane is not flipped: row 0 = bottom
        SU.assert_image(got[name].view(np.uint8).reshape(H, W, 4)[::-1], whole, name)
    mc = SU.component("diablo")
    mc.transform.set_rotation((0.0, 0.0, 0.0))
    for k in range(n_views):
        want = SU.oracle_render(oracle, W, H, [(mc, scenes.shader_for(mc, W, H, eye=scenes.orbit_eye(k, n_views)))])["present"]
        SU.assert_image(got["frames"][k].view(np.uint8).reshape(H, W, 4)[::-1], want, f"view {k}")


def _rccl_worker(rank, port, W, H, out_path):
    """world_size 1 over the REAL backend of the multi-GPU job ("nccl" = RCCL): the collectives take the library's own
    hipMalloc memory (sharding.device_frames: no _stage, no .cpu()) exactly as the driver's 8-GPU run will."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import time

        import scene_util as SU
        from tinyrenderer_amd import _ffi, scenes, sharding
        from tinyrenderer_amd.model_renderer import (ModelRendererState, clear_screen, render_mesh_component,
                                                     render_views)

        assert dist.get_backend() == "nccl"
        head, sh = _scene(W, H)
        st = ModelRendererState(W, H, device=0)
        plane = sharding.device_frames(st)  # (1, H, W) int32: the context's device memory itself
        assert plane.is_cuda and plane.data_ptr() != 0
        res = {}
        # ---- frame A through all_gather and gather, then frame B (another rotation) over the same plane right away
        clear_screen(st)
        render_mesh_component(st, head, sh)
        st.resolve(_ffi.TR_OUT_RGBA8)
        sharding.to_torch(st)
        t0 = time.perf_counter()
        got_ag = [torch.empty_like(plane)]
        dist.all_gather(got_ag, plane)
        got_g = [torch.empty_like(plane)]
        dist.gather(plane, got_g, dst=0)
        red = plane.clone()
        dist.all_reduce(red, op=dist.ReduceOp.MAX)  # (one more collective kind on the same memory)
        torch.cuda.synchronize()
        res["gather_ms"] = (time.perf_counter() - t0) * 1e3
        sharding.to_renderer(st)  # the renderer may not overwrite the plane before the collectives have read it
        want_a = st.present()
        head.transform.set_rotation((0.9, 0.0, 0.0))
        clear_screen(st)
        render_mesh_component(st, head, scenes.shader_for(head, W, H))
        st.resolve(_ffi.TR_OUT_RGBA8)
        sharding.to_torch(st)
        got_b = [torch.empty_like(plane)]
        dist.all_gather(got_b, plane)
        torch.cuda.synchronize()
        sharding.to_renderer(st)
        want_b = st.present()
        head.transform.set_rotation((0.025, 0.0, 0.0))
        # ---- the library's own gathers (world 1: identity) and a batch of views
        full = sharding.gather_bands(plane[0], H, dst=0)
        frames = sharding.gather_frames(plane, 1, dst=0)
        mc = SU.component("diablo")
        mc.transform.set_rotation((0.0, 0.0, 0.0))
        vs = ModelRendererState(W, H, device=0, max_views=3)
        shaders = [scenes.shader_for(mc, W, H, eye=scenes.orbit_eye(k, 3)) for k in range(3)]
        render_views(vs, mc, shaders, _ffi.TR_OUT_RGBA8)
        vplane = sharding.device_frames(vs, 3)
        sharding.to_torch(vs)
        got_v = [torch.empty_like(vplane)]
        dist.gather(vplane, got_v, dst=0)
        torch.cuda.synchronize()
        sharding.to_renderer(vs)
        views = []
        for v in range(3):
            vs.select_view(v)
            views.append(vs.present())
        np.savez(out_path, ag=got_ag[0].cpu().numpy(), g=got_g[0].cpu().numpy(), red=red.cpu().numpy(), b=got_b[0].cpu().numpy(),
                 want_a=want_a, want_b=want_b, full=full.cpu().numpy(), frames=frames.cpu().numpy(),
                 gv=got_v[0].cpu().numpy(), views=np.stack(views), gather_ms=res["gather_ms"])
        vs.close()
        st.close()
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_take_the_librarys_device_memory(tmp_path):
    """RCCL ("nccl", world_size 1 on the one GPU of the box) accepts the context's hipMalloc'ed plane as a collective's
    input without a copy: what all_gather / gather / all_reduce deliver equals tr_read_rgba8, for a frame and for the
    frame rendered over the same memory straight after it (to_torch / to_renderer order the two streams)."""
    W, H = 640, 368
    out_path = str(tmp_path / "rccl.npz")
    mp.start_processes(_rccl_worker, args=(_free_port(), W, H, out_path), nprocs=1, join=True, start_method="spawn")
    got = np.load(out_path)

    def img(plane):  # packed (H, W) int32, row 0 = bottom -> the presented image
        return plane.view(np.uint8).reshape(H, W, 4)[::-1]

    assert not np.array_equal(got["want_a"], got["want_b"])
    for name in ("ag", "g", "red"):
        assert np.array_equal(img(got[name][0]), got["want_a"]), name
    assert np.array_equal(img(got["b"][0]), got["want_b"])
    # (the library's own gathers are the identity at world 1: by then the plane holds frame B)
    assert np.array_equal(img(got["full"]), got["want_b"]) and np.array_equal(img(got["frames"][0]), got["want_b"])
    for v in range(3):
        assert np.array_equal(img(got["gv"][v]), got["views"][v]), v
    print("RCCL world-1 collectives on the library's memory ok; gather_ms", float(got["gather_ms"]))
