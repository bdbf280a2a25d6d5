#!/usr/bin/env python3
"""Headline benchmark: frames/s (+ Mfragments/s) of the reference's african_head scene at
3840x2160 through the HIP path, with the shade pass priced against the HBM roofline and the
C++ oracle (a restatement of the reference, NOT the Rust binary) timed on the host beside it.

    python bench.py --gpus N --steps K --warmup W

A step is a BATCH of --frames-per-step frames (100 by default, so that the driver's 20 steps time
0.2 s, not 2 ms); a frame is tr_clear + tr_draw + tr_resolve (the window the reference prints as
"Frametime (nosync)", src/main.rs:58-60,295).  Mesh and textures are resident in HBM before
the timed region.  After the timed region every context's presented image and counters are checked
against the oracle's digest of this frame (tests/golden/hashes.json, c3_head_4k): "verified".  N > 1 (launched by torch.distributed.run) shards independent frames over
ranks with no data-path collective: weak scaling, value = total frames / max-over-ranks time.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_scene(W, H):
    import scene_util as SU
    from tinyrenderer_amd import scenes

    head = SU.component("head")
    head.transform.set_rotation((0.025, 0.0, 0.0))  # frame time = 1, src/main.rs:92-94
    return [(head, scenes.shader_for(head, W, H))]


def cpu_baseline(W, H, draws, budget_s=12.0):
    """Oracle ('port' of the reference, single thread) on the same frame; bounded sample."""
    import oracle_ffi as O

    st = O.OracleState(W, H)
    imgs = {}
    for mc, _ in draws:
        for img in (mc.texture, mc.normal_map):
            if id(img) not in imgs:
                imgs[id(img)] = O.OracleImage(img[::-1], flip=True)
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_all < budget_s and len(times) < 40):
        t0 = time.perf_counter()
        st.clear_screen()
        for mc, sh in draws:
            rc = st.render_mesh_component(mc.mesh, sh, imgs[id(mc.texture)], imgs[id(mc.normal_map)])
            assert rc == 0
        times.append(time.perf_counter() - t0)
    med = float(np.median(times[1:]))
    cpu_model = ""
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": 1.0 / med, "unit": "frames/s", "cores": 1, "kind": "port", "cpu_model": cpu_model, "host_cpus": os.cpu_count(),
            "sample": f"{len(times) - 1} frames of the same {W}x{H} african_head frame (median), "
                      "C++ restatement of the reference (oracle/), single thread, clear + draw, no present",
            "ms_per_frame": med * 1e3, "fragments_per_frame": st.counters()["fragments"]}


def secondary_bands(torch, dist, rank, world, device, frames=8, warm=12):
    """BASELINE.json configs[3]: african_head subdivided 6x (10 207 232 triangles) at 7680x4320, ONE frame cut into
    tile-row bands, one per rank (tr_set_band), re-cut every frame from the ranks' last band times (one all-gather
    of a float each, sharding.rebalance); the RGBA8 bands go to rank 0 straight from the contexts' device memory
    (sharding.device_frames -> dist.gather: RCCL over xGMI).  frames/s = 1 / (slowest rank's render + the gather)."""
    import big_cases as B
    from tinyrenderer_amd import _ffi, sharding
    from tinyrenderer_amd.model_renderer import ModelRendererState, make_draw_params

    levels = int(os.environ.get("TR_BENCH_BANDS_LEVELS", "6"))
    W, H, draws = B.head_subdivided(levels, 7680, 4320)
    st = ModelRendererState(W, H, device=device)
    plane = sharding.device_frames(st)[0]
    lib, cx = _ffi.load(), st._ctx
    mc, sh = draws[0][0], draws[0][1]
    params = make_draw_params(st, sh, mc.texture, mc.normal_map)
    mesh = st._mesh_handle(mc.mesh)
    on_gpu = world == 1 or dist.get_backend() == "nccl"
    bands = [sharding.band_rows(H, r, world) for r in range(world)]
    per_frame, renders, gathers, full = [], [], [], None
    last_band, last_spans = bands[rank], list(bands)
    for it in range(warm + frames):
        y_lo, y_hi = bands[rank]
        last_band, last_spans = bands[rank], list(bands)  # (the cut this frame is rendered and gathered with; re-cut below)
        if world > 1:
            dist.barrier()
        st.synchronize()
        t0 = time.perf_counter()
        # (the C ABI directly: matrices and handles were made once above, as a compiled host would hold them)
        rcs = (lib.tr_clear(cx), lib.tr_set_band(cx, y_lo, y_hi), lib.tr_draw(cx, mesh, C.byref(params)),
               lib.tr_resolve(cx, _ffi.TR_OUT_RGBA8))
        st.synchronize()
        t_render = (time.perf_counter() - t0) * 1e3
        assert rcs == (0, 0, 0, 0), rcs
        sharding.to_torch(st)
        t1 = time.perf_counter()
        full = sharding.gather_bands(plane[y_lo:y_hi], H, dst=0, spans=bands)
        torch.cuda.synchronize()
        t_gather = (time.perf_counter() - t1) * 1e3  # this rank's share of the one delivery gather (send, or all receives on rank 0)
        sharding.to_renderer(st)
        if world > 1:
            dist.barrier()
        t_frame = (time.perf_counter() - t0) * 1e3
        times, gtimes = [t_render], [t_gather]
        if world > 1:
            t = torch.tensor([t_render, t_gather], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
            out = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            times = [float(x[0].item()) for x in out]
            gtimes = [float(x[1].item()) for x in out]
        if it >= warm:
            per_frame.append(t_frame)
            renders.append(times)
            gathers.append(gtimes)
        if world > 1:
            bands = sharding.rebalance(bands, times, H)
    rec = {"workload": f"african_head subdivided {levels}x ({draws[0][0].mesh.triangles.shape[0]} triangles) {W}x{H}, "
                       f"{world} tile-row band(s), RGBA8 gather to rank 0 (BASELINE.json configs[3])",
           "frames_per_s": 1e3 / float(np.median(per_frame)), "ms_per_frame": float(np.median(per_frame)),
           "render_ms_per_rank_last_frame": [round(x, 3) for x in renders[-1]], "band_rows_last_frame": [hi - lo for lo, hi in bands],
           "gather_ms_per_rank_median": [round(float(np.median([g[r] for g in gathers])), 3) for r in range(world)],
           "gather_bytes_per_rank_last_frame": [(hi - lo) * W * 4 for lo, hi in bands], "backend": dist.get_backend() if world > 1 else "none (one rank)"}
    if levels == 6:
        # the timed frames use the library's default (fused, +-1 LSB) colour arithmetic: one more frame over the same bands
        # with the reference's arithmetic must hash to the oracle's digest, and the timed frame stay within 1 LSB of it
        import hashlib

        fused = None if full is None else full.cpu().numpy().view(np.uint8).reshape(H, W, 4)[::-1].astype(np.int16)  # (the plane is not flipped)
        st.set_colour_exact(True)
        y_lo, y_hi = last_band
        rcs = (lib.tr_clear(cx), lib.tr_set_band(cx, y_lo, y_hi), lib.tr_draw(cx, mesh, C.byref(params)), lib.tr_resolve(cx, _ffi.TR_OUT_RGBA8))
        assert rcs == (0, 0, 0, 0), rcs
        sharding.to_torch(st)
        exact = sharding.gather_bands(plane[y_lo:y_hi], H, dst=0, spans=last_spans)
        torch.cuda.synchronize()
        sharding.to_renderer(st)
        if rank == 0:
            want = json.load(open(os.path.join(ROOT, "tests", "golden", "hashes.json")))["c4_head_sub6_8k"]["rgba8"]
            img = exact.cpu().numpy().view(np.uint8).reshape(H, W, 4)[::-1]
            lsb = int(np.abs(fused - img.astype(np.int16)).max())
            rec["max_colour_lsb_vs_exact"] = lsb
            rec["verified"] = hashlib.sha256(np.ascontiguousarray(img)).hexdigest() == want and lsb <= 1
    st.close()
    return rec


def secondary_orbit(torch, dist, rank, world, device, n_total=4096):
    """BASELINE.json configs[4]: the 4096-view camera orbit of diablo3_pose at 1920x1080, a contiguous block of
    views per rank (sharding.shard_range), rendered in batches through tr_render_views; views/s of the whole job
    with the frames left resident in each rank's HBM, and with every batch delivered to rank 0 (dist.gather from
    the contexts' device memory)."""
    import scene_util as SU
    from tinyrenderer_amd import _ffi, scenes, sharding
    from tinyrenderer_amd.model_renderer import ModelRendererState, make_draw_params

    W, H = 1920, 1080
    batch = int(os.environ.get("TR_BENCH_ORBIT_BATCH", "32"))  # (measured on one GPU, 4 contexts: 8 -> 31.0 k, 16 -> 33.2 k, 24 -> 34.4 k, 32 -> 34.9 k views/s)
    n_ctx = int(os.environ.get("TR_BENCH_ORBIT_CONTEXTS", "4"))
    n_total = int(os.environ.get("TR_BENCH_ORBIT_VIEWS", str(n_total)))
    mc = SU.component("diablo")
    mc.transform.set_rotation((0.0, 0.0, 0.0))
    lo, hi = sharding.shard_range(n_total, rank, world)
    n_local = hi - lo
    lib = _ffi.load()
    states = [ModelRendererState(W, H, device=device, max_views=batch) for _ in range(n_ctx)]
    shaders = [scenes.shader_for(mc, W, H, eye=scenes.orbit_eye(k, n_total)) for k in range(lo, hi)]  # host, once
    params, meshes = [], []
    for s_ in states:
        arr = (_ffi.DrawParams * max(1, n_local))()
        for k, sh in enumerate(shaders):
            arr[k] = make_draw_params(s_, sh, mc.texture, mc.normal_map)
        params.append(arr)
        meshes.append(s_._mesh_handle(mc.mesh))
    stride = C.sizeof(_ffi.DrawParams)
    max_local = max(sharding.shard_range(n_total, r, world)[1] - sharding.shard_range(n_total, r, world)[0] for r in range(world))
    planes = [sharding.device_frames(s_, batch) for s_ in states]
    on_gpu = world == 1 or dist.get_backend() == "nccl"
    recv = None
    if rank == 0 and world > 1:
        recv = [torch.empty((batch, H, W), dtype=torch.int32, device="cuda" if on_gpu else "cpu") for _ in range(world)]

    gather_s = [0.0, 0]

    def run(deliver):
        done, i = 0, 0
        while done < max_local:  # (every rank makes the same number of gather calls)
            v = max(0, min(batch, n_local - done))
            s_ = states[i % n_ctx]
            if v:
                p = C.cast(C.byref(params[i % n_ctx], done * stride), C.POINTER(_ffi.DrawParams))
                rc = lib.tr_render_views(s_._ctx, meshes[i % n_ctx], p, v, _ffi.TR_OUT_RGBA8)
                if rc != 0:
                    _ffi.check(s_._ctx, rc, "tr_render_views")
            if deliver and world > 1:
                sharding.to_torch(s_)
                tg = time.perf_counter()
                src = planes[i % n_ctx] if on_gpu else planes[i % n_ctx].cpu()
                dist.gather(src, recv, dst=0)
                gather_s[0] += time.perf_counter() - tg  # (host time of the call: the collective itself is asynchronous on the stream)
                gather_s[1] += 1
                sharding.to_renderer(s_)
            done += batch
            i += 1
        for s_ in states:
            s_.synchronize()
        torch.cuda.synchronize()

    def timed(deliver):
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        run(deliver)
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t0

    run(False)
    t_res = min(timed(False) for _ in range(2))
    t_del = min(timed(True) for _ in range(2)) if world > 1 else t_res
    frags = states[0].counters()["fragments"]
    rec = {"workload": f"diablo3_pose {W}x{H}, {n_total}-view orbit, {world} rank(s) x {n_local} views, batches of {batch} views "
                       f"(tr_render_views) on {n_ctx} contexts (BASELINE.json configs[4])",
           "views_per_s_resident": n_total / t_res, "views_per_s_delivered_to_rank0": n_total / t_del,
           "fragments_last_batch_context0": frags}
    rec["gather_bytes_per_batch_per_rank"] = batch * W * H * 4
    rec["gather_call_ms_per_batch_this_rank"] = (gather_s[0] / gather_s[1] * 1e3) if gather_s[1] else None
    if world > 1:
        t = torch.tensor([rec["gather_call_ms_per_batch_this_rank"] or 0.0], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rec["gather_call_ms_per_batch_per_rank"] = [round(float(x.item()), 3) for x in allt]
        rec["backend"] = dist.get_backend()
    for s_ in states:
        s_.close()
    if rank == 0 and world == 1 and os.environ.get("TR_BENCH_NO_CPU_BASELINE") != "1":
        rec["cpu_baseline"] = cpu_baseline_orbit(n_total)
    return rec


def _orbit_cpu_worker(args):
    """One process of the view-parallel CPU figure: the oracle on a block of views of the orbit (CPU only)."""
    ks, n_total, W, H = args
    import oracle_ffi as O
    import scene_util as SU
    from tinyrenderer_amd import scenes

    mc = SU.component("diablo")
    mc.transform.set_rotation((0.0, 0.0, 0.0))
    st = O.OracleState(W, H)
    tex, nm = O.OracleImage(mc.texture[::-1], flip=True), O.OracleImage(mc.normal_map[::-1], flip=True)
    shaders = [scenes.shader_for(mc, W, H, eye=scenes.orbit_eye(k, n_total)) for k in ks]
    t0 = time.perf_counter()
    for sh in shaders:
        st.clear_screen()
        assert st.render_mesh_component(mc.mesh, sh, tex, nm) == 0
    return time.perf_counter() - t0, len(ks)


def cpu_baseline_orbit(n_total, W=1920, H=1080, views_per_proc=6):
    """the N-process view-parallel CPU figure for configs[4] -- the oracle ('port' of the reference: the
    Rust binary cannot be built here) on N processes, each rendering its own views of the orbit (the loop of
    src/main.rs:57-116 with the view as the loop variable).  A bounded sample: N x views_per_proc views spread over the orbit."""
    import multiprocessing as mp

    n_proc = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)))  # (the box's CPU share: 16)
    n_views = n_proc * views_per_proc
    ks = [int(i * n_total / n_views) for i in range(n_views)]
    blocks = [(ks[p::n_proc], n_total, W, H) for p in range(n_proc)]
    ctx = mp.get_context("spawn")  # (never fork a process that has initialised the GPU)
    with ctx.Pool(n_proc) as pool:
        res = pool.map(_orbit_cpu_worker, blocks)
    slowest = max(t for t, _ in res)
    one = float(np.mean([t / n for t, n in res]))
    cpu_model = ""
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": n_views / slowest, "unit": "views/s", "cores": n_proc, "kind": "port",
            "sample": f"{n_views} views spread over the {n_total}-view orbit at {W}x{H}, {views_per_proc} per process, {n_proc} processes "
                      "(C++ restatement of the reference, oracle/, one thread each; clear + draw per view, no present); "
                      "views / slowest process's render time",
            "ms_per_view_one_core": one * 1e3, "cpu_model": cpu_model, "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=100,
                    help="frames per step per rank: 20 steps = 2 000 frames = 0.2 s, the ramp of four frames in flight is then < 1 %%")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the configs[3] (bands) and configs[4] (orbit) records beside the headline line")
    ap.add_argument("--profile-frames", type=int, default=30)
    ap.add_argument("--contexts", type=int, default=int(os.environ.get("TR_BENCH_CONTEXTS", "4")),
                    help="render contexts (HIP streams) used round-robin: frames of successive steps overlap")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal on a one-GPU box only: TR_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # TR_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).
    if os.environ.get("TR_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("TR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import ModelRendererState, make_draw_params

    lib = _ffi.load()
    W, H = args.width, args.height
    draws = build_scene(W, H)
    states = [ModelRendererState(W, H, device=local_rank) for _ in range(max(1, args.contexts))]
    st = states[0]
    ctx = st._ctx
    prepared_all = [[(s._mesh_handle(mc.mesh), make_draw_params(s, sh, mc.texture, mc.normal_map)) for mc, sh in draws]
                    for s in states]
    turn = [0]

    def frame(which=None):
        k = turn[0] % len(states) if which is None else which
        turn[0] += 1
        cx = states[k]._ctx
        lib.tr_clear(cx)
        for mesh_h, params in prepared_all[k]:
            rc = lib.tr_draw(cx, mesh_h, C.byref(params))
            if rc != 0:
                _ffi.check(cx, rc, "tr_draw")
        rc = lib.tr_resolve(cx, _ffi.TR_OUT_RGBA8)
        if rc != 0:
            _ffi.check(cx, rc, "tr_resolve")

    def sync():
        for s in states:
            lib.tr_synchronize(s._ctx)
        torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    fps_step = max(1, args.frames_per_step)
    for _ in range(args.warmup * fps_step):
        frame()
    sync()
    counters = st.counters()
    flags = st.status_flags()
    no_verify = os.environ.get("TR_BENCH_NO_VERIFY") == "1"  # (timing experiments with kernels that draw wrong pixels: tools/ab_bench.sh)
    assert flags == 0 or no_verify, f"status flags {flags:#x}"

    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps * fps_step):
        frame()
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_frames = args.steps * fps_step * world
    fps = total_frames / elapsed
    frags = counters["fragments"]

    # ---- the frames that were timed, checked: every context's presented image, fragment / sample / scanline
    # counters and covered pixels against the oracle's digest of this frame (made in the build container by
    # tests/golden/make_golden.py --big; the oracle itself is timed below as cpu_baseline)
    verified = None
    max_lsb = 0
    colour_exact_env = os.environ.get("TR_SHADE_EXACT", "0") not in ("", "0")
    if (W, H) == (3840, 2160) and not no_verify:
        import hashlib

        want = json.load(open(os.path.join(ROOT, "tests", "golden", "hashes.json")))["c3_head_4k"]
        verified = True
        for k, s_ in enumerate(states):
            c_ = s_.counters()
            # the timed frame's image (fused colour arithmetic unless TR_SHADE_EXACT=1: within 1 LSB), then the same
            # frame shaded operation for operation as the reference: that image must hash to the oracle's digest
            timed_img = s_.present().astype(np.int16)
            s_.set_colour_exact(True)
            s_.resolve(_ffi.TR_OUT_RGBA8)  # (the mode holds from the next resolve on)
            exact_img = s_.present()
            s_.set_colour_exact(colour_exact_env)
            lsb = int(np.abs(timed_img - exact_img.astype(np.int16)).max())
            max_lsb = max(max_lsb, lsb)
            ok = (s_.status_flags() == 0 and c_["covered_pixels"] == want["covered"]
                  and all(c_[n] == want["counters"][n] for n in ("samples", "fragments", "scanlines"))
                  and hashlib.sha256(np.ascontiguousarray(exact_img)).hexdigest() == want["rgba8"] and lsb <= 1)
            if not ok:
                verified = False
                print(f"bench.py: rank {rank} context {k}: the timed frame does NOT match the oracle's digest", file=sys.stderr)

    # ---- one context alone: frames back to back on one stream, and the latency of a frame the host waits for
    n_single = max(50, min(500, args.steps * fps_step // 4))
    for _ in range(20):
        frame(0)
    sync()
    t1 = time.perf_counter()
    for _ in range(n_single):
        frame(0)
    sync()
    single_fps = n_single / (time.perf_counter() - t1)
    lat = []
    for _ in range(60):
        t1 = time.perf_counter()
        frame(0)
        lib.tr_synchronize(ctx)
        lat.append(time.perf_counter() - t1)
    latency_us = float(np.median(lat)) * 1e6

    # ---- per-stage GPU time with hipEvents on the library's own stream (separate loop:
    # event-bracketing every launch serialises the host, so it is not part of `value`)
    stage_names = ["clear", "transform", "setup_scan", "depth_pass", "winner_pass", "shade"]
    lib.tr_profile_enable(ctx, 1)
    ms = (C.c_double * 6)()
    cnt = (C.c_uint64 * 6)()
    lib.tr_profile_read(ctx, ms, cnt, 1)
    for _ in range(args.profile_frames):
        frame(0)
    sync()
    lib.tr_profile_read(ctx, ms, cnt, 1)
    lib.tr_profile_enable(ctx, 0)
    stages_us = {n: (ms[i] / max(1, cnt[i])) * 1e3 for i, n in enumerate(stage_names)}

    # ---- algorithmic bytes of the shade pass (SURVEY.md 8(d) / DESIGN.md):
    # W*H*(8 B winner-key read + 4 B RGBA8 write) + visible_tris*320 B + sum(min(texture bytes, shaded_px*16 B))
    grid = (C.c_uint32 * 4)()
    lib.tr_get_shade_launch(ctx, grid)  # (of the timed kind of frame: before the plane requests below resolve the whole frame)
    tri = st.triangle_ids(_ffi.TR_LAYOUT_ROW_MAJOR)
    shaded_px = int((tri != 0xFFFFFFFF).sum())
    vis_tris = int(np.unique(tri[tri != 0xFFFFFFFF]).size)
    tex_bytes = 0
    seen = set()
    for mc, _ in draws:
        for img in (mc.texture, mc.normal_map):
            if id(img) not in seen:
                seen.add(id(img))
                tex_bytes += min(img.shape[0] * img.shape[1] * 4, shaded_px * 16)
    shade_bytes = W * H * 12 + vis_tris * 320 + tex_bytes
    traffic, pmc = None, {}
    try:  # HBM bytes per k_shade launch from the committed PMC passes (same workload only)
        import glob

        pmc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_pmc_4k.json")))[-1]))  # the latest round's
        if pmc["workload"]["W"] == W and pmc["workload"]["H"] == H:
            traffic = pmc["k_shade_hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    shade_s = stages_us["shade"] * 1e-6
    achieved = shade_bytes / shade_s / 1e9 if shade_s > 0 else 0.0
    # the bytes of the blocks the boxed launch actually covers (it never reads or writes W*H*12 B), and the pass priced
    # against what bounds it besides bytes: VALU issue, every instruction at the f64 rate of tools/valu_rates.hip
    launched_px = int(grid[0]) * int(grid[1]) * max(1, int(grid[2])) * int(grid[3])
    launched_bytes = min(launched_px, W * H) * 12 + vis_tris * 320 + tex_bytes
    roofline_valu = None
    try:
        insts = pmc.get("k_shade_valu_insts_per_launch") if traffic is not None else None
        if insts and shade_s > 0:
            floor_us = insts * 5.0 / 1024.0 / 2400.0  # wave-instructions x 5 cycles / 1024 SIMDs / 2.4 GHz
            roofline_valu = {"bound": "valu-issue", "kernel": pmc.get("k_shade_kernel"), "valu_wave_insts_per_launch": insts,
                             "cycles_per_inst": 5.0, "simds": 1024, "clock_ghz": 2.4, "floor_us": floor_us,
                             "frac": floor_us / stages_us["shade"],
                             "note": "every VALU wave-instruction priced at the f64 multiply-add issue rate (5 cycles per wave64 at full occupancy, tools/valu_rates.hip): an upper estimate of the issue floor"}
    except NameError:
        pass

    # the raster pair beside it: the depth pass is the longest kernel of the frame and neither HBM- nor MFMA-bound -- its live
    # stage times against the two floors its counters give (VALU issue at 4 cycles per wave-instruction; memory-side atomic
    # service at the ~20 G requests/s of MI355X_MICROARCH.md), counters from the committed single-context PMC passes
    raster_passes = None
    if traffic is not None and "k_depth_valu_insts_per_launch" in pmc:
        raster_passes = {"note": "live hipEvent times of this run (the 128-thread form when four contexts are alive); counters per launch from "
                                 "profiles/ (rocprofv3 --pmc, one context, the 256-thread form); floors: VALU wave-instructions x 4 cycles / 1024 SIMDs / "
                                 "2.4 GHz, atomic requests / 20 G/s"}
        for short, stage in (("k_depth", "depth_pass"), ("k_winner", "winner_pass")):
            insts, reqs = pmc.get(short + "_valu_insts_per_launch"), pmc.get(short + "_atomic_requests_per_launch")
            e = {"kernel": pmc.get(short + "_kernel"), "avg_launch_us": stages_us[stage], "valu_wave_insts_per_launch": insts,
                 "atomic_requests_per_launch": reqs, "hbm_bytes_per_launch": pmc.get(short + "_hbm_bytes_per_launch")}
            if insts:
                e["valu_floor_us"] = insts * 4.0 / 1024.0 / 2400.0
            if reqs:
                e["atomic_floor_us"] = reqs / 20e3
            if stages_us[stage] > 0:
                e["frac_of_larger_floor"] = max(e.get("valu_floor_us", 0.0), e.get("atomic_floor_us", 0.0)) / stages_us[stage]
            raster_passes[stage] = e

    out = {
        "metric": "frames/sec at 4K (african_head 3840x2160, clear+draw+resolve)",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "verified": verified,
        "colour_arithmetic": "exact (0 LSB)" if colour_exact_env else "fused (+-1 LSB of the reference; depth / winner keys exact)",
        "max_colour_lsb_vs_exact": max_lsb,
        "vs_baseline": None, "dtype": "f64", "data": "reference assets (african_head.obj + diffuse/nm_tangent PNG), fixed frame time=1",
        "config": {"workload": f"african_head {W}x{H} single frame, 1 draw (BASELINE.json configs[2])",
                   "frames_per_rank_per_step": fps_step, "contexts_per_gpu": len(states), "parallelism": f"views x{world}" if world > 1 else "single GPU"},
        "mfragments_per_s": fps * frags / 1e6, "single_context_fps": single_fps, "latency_us": latency_us,
        "fragments_per_frame": frags, "samples_per_frame": counters["samples"], "shaded_px_per_frame": shaded_px,
        "stages_us": stages_us,
        "roofline": {"bound": "hbm", "kernel": "k_shade" if colour_exact_env else "k_shade_fused", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes": shade_bytes, "avg_launch_us": stages_us["shade"],
                     "launched_bytes": launched_bytes, "launched_workgroups": [int(grid[0]), int(grid[1]), int(grid[2])],
                     "launched_gbs": launched_bytes / shade_s / 1e9 if shade_s > 0 else 0.0},
        "roofline_valu": roofline_valu,
        "raster_passes": raster_passes,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(W, H, draws)
    for s_ in states:
        s_.close()
    if world > 1:
        v = torch.tensor([0 if verified is False else 1], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        if verified is not None:
            verified = bool(v.item())
            out["verified"] = verified
    # ---- the other two multi-GPU configurations of BASELINE.json, beside the headline (never part of `value`)
    if not args.no_secondary:
        # A failure here must not cost the headline line, and neither may a hang (these records run collectives this
        # container cannot rehearse on more than one GPU): past the limit rank 0 prints the line without them and
        # every rank leaves.
        import threading

        out["secondary"] = {}
        limit = float(os.environ.get("TR_BENCH_SECONDARY_LIMIT_S", "240"))

        def give_up():
            # (a hang in the secondary records may be a hung collective or GPU: the headline line is still printed, but the
            # process must not report success -- exit status 4, or 3 when the headline itself was not verified)
            if rank == 0:
                out["secondary"].setdefault("error", f"not finished after {limit:.0f} s")
                print(json.dumps(out), flush=True)
            os._exit(3 if verified is False else 4)

        watchdog = threading.Timer(limit + (0.0 if rank == 0 else 10.0), give_up)
        watchdog.daemon = True
        watchdog.start()
        for name, fn in (("bands_8k_10M_triangles", secondary_bands), ("orbit_4096_views_1080p", secondary_orbit)):
            try:
                out["secondary"][name] = fn(torch, dist, rank, world, local_rank)
            except Exception as e:
                out["secondary"][name] = {"error": f"{type(e).__name__}: {e}"}
        watchdog.cancel()
    # a secondary record whose frame did not match its digest, or that raised, fails the run too (the line is printed first)
    sec = out.get("secondary", {})
    sec_bad = [n for n, r in sec.items() if isinstance(r, dict) and (r.get("verified") is False or "error" in r)]
    if "error" in sec:
        sec_bad.append("watchdog")
    if sec_bad:
        out["secondary_failed"] = sec_bad
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if verified is False:
        raise SystemExit(3)
    if sec_bad:
        raise SystemExit(5)


if __name__ == "__main__":
    main()
