"""Multi-GPU partitioning of the render path: one process per GPU, no data-path collective.

The path shards two ways; both are exchange-free until one final gather:
  * views: independent camera views, a contiguous block per rank (config 5);
  * bands: horizontal bands of 16-pixel tile rows of ONE frame (config 4) -- exact because the
    per-pixel result is a pure min over the samples that land on that pixel.
The only collective is the delivery gather of finished RGBA8 frames / bands to rank 0
(RCCL over xGMI on GPUs: backend "nccl"; "gloo" on CPU for tests).
"""
import torch
import torch.distributed as dist

TILE = 16


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n items for `rank`; blocks differ by at most one item."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def band_rows(height, rank, world, tile=TILE):
    """Band [y_lo, y_hi) of pixel rows for `rank`: whole tile rows, contiguous, covering the frame."""
    n_tile_rows = (height + tile - 1) // tile
    lo, hi = shard_range(n_tile_rows, rank, world)
    return min(lo * tile, height), min(hi * tile, height)


def balanced_bands(tile_row_weights, world, tile=TILE, height=None):
    """Bands of whole tile rows with about equal WEIGHT each ("balanced by binned-sample
    count, not by area"): `tile_row_weights[k]` is the cost of tile row k, e.g. the covered pixels or
    fragments the previous frame had there plus a constant for the per-row overhead.  Returns a list
    of `world` bands [(y_lo, y_hi)], contiguous and covering the frame; every rank computes the same
    list from the same weights."""
    w = [float(x) for x in tile_row_weights]
    n = len(w)
    height = n * tile if height is None else height
    prefix = [0.0]
    for x in w:
        prefix.append(prefix[-1] + x)
    cuts = [0]
    for r in range(1, world):
        target = prefix[-1] * r / world
        if n >= world:  # every rank gets at least one tile row
            lo_k, hi_k = cuts[-1] + 1, n - (world - r)
        else:
            lo_k, hi_k = cuts[-1], n
        cuts.append(min(range(lo_k, hi_k + 1), key=lambda k: (abs(prefix[k] - target), k)))
    cuts.append(n)
    return [(min(a * tile, height), min(b * tile, height)) for a, b in zip(cuts[:-1], cuts[1:])]


def rebalance(bands, times_ms, height, tile=TILE):
    """Next frame's bands from this frame's: `times_ms[r]` is how long rank r's band took (what every rank learns
    from one all-gather of a float).  The cost per tile row is taken constant inside a band after removing a fixed
    share, `balanced_bands` equalises it, and every cut moves half of the way (damping)."""
    world = len(bands)
    n_rows = (height + tile - 1) // tile
    fixed = 0.5 * min(times_ms)  # rough share of a band's time that does not depend on its rows
    row_w = [0.0] * n_rows
    for (y_lo, y_hi), t in zip(bands, times_ms):
        k0, k1 = y_lo // tile, (y_hi + tile - 1) // tile
        for k in range(k0, k1):
            row_w[k] = max(t - fixed, 1e-3) / max(1, k1 - k0)
    proposed = balanced_bands(row_w, world, tile=tile, height=height)
    cuts = [0] + [((a[1] + b[1]) // 2) // tile * tile for a, b in zip(bands[:-1], proposed[:-1])] + [height]
    for i in range(1, len(cuts)):  # (keep every band non-empty while rows remain)
        cuts[i] = max(cuts[i], min(cuts[i - 1] + tile, height))
    cuts[-1] = height
    return list(zip(cuts[:-1], cuts[1:]))


# ---- the renderer's planes as torch tensors, without a copy ---------------------------------------------------
class _DevicePlane:
    """`__cuda_array_interface__` over the context's RGBA8 plane (tr_device_ptr_rgba8): torch wraps the memory
    in place.  The state must outlive every tensor made from it."""

    def __init__(self, state, n_views):
        import ctypes as C

        from . import _ffi

        p = C.c_void_p()
        if n_views > 1:
            state.select_view(0)  # (the slices are contiguous in view order from view 0's address)
        _ffi.check(state._ctx, state._lib.tr_device_ptr_rgba8(state._ctx, C.byref(p)), "tr_device_ptr_rgba8")
        self._keep = state
        self.__cuda_array_interface__ = {"shape": (n_views, state.frame_height, state.frame_width), "typestr": "<i4",
                                         "data": (int(p.value), False), "version": 2, "strides": None}


def device_frames(state, n_views=1, device=None):
    """(n_views, H, W) int32 tensor (packed RGBA8, row 0 = bottom: the plane is not flipped) that IS the context's
    device memory.  Order the streams with `to_torch` / `to_renderer` around its use."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return torch.as_tensor(_DevicePlane(state, n_views), device=dev)


def render_stream(state):
    """The context's HIP stream (tr_get_stream) as a torch stream."""
    import ctypes as C

    from . import _ffi

    p = C.c_void_p()
    _ffi.check(state._ctx, state._lib.tr_get_stream(state._ctx, C.byref(p)), "tr_get_stream")
    return torch.cuda.ExternalStream(int(p.value))


def to_torch(state):
    """torch's current stream waits for everything the renderer has queued (before a collective reads its planes)."""
    torch.cuda.current_stream().wait_stream(render_stream(state))


def to_renderer(state):
    """The renderer's stream waits for torch's current stream (before the next frame overwrites the planes)."""
    render_stream(state).wait_stream(torch.cuda.current_stream())


def _stage(t):
    """gloo moves host memory: device tensors (rehearsals of several ranks on one GPU) go through the host."""
    if dist.is_initialized() and dist.get_backend() == "gloo" and t.is_cuda:
        return t.cpu()
    return t


def gather_frames(local, n_total, dst=0):
    """Gather per-rank blocks of finished frames (n_local, H, W) uint8/int32 onto `dst` in view
    order.  Returns the (n_total, H, W) tensor on dst, None elsewhere."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return local
    local = _stage(local)
    max_n = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world))
    pad = torch.zeros((max_n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(bufs[r][: hi - lo])
    return torch.cat(parts, dim=0)


def gather_bands(local_band, height, dst=0, tile=TILE, spans=None):
    """Gather per-rank bands (rows [y_lo, y_hi) of one frame, shape (rows, W[, C])) onto `dst`
    and stack them into the (height, W[, C]) frame.  `spans`: the bands of all ranks if they are
    not the equal ones of band_rows (balanced_bands).
    Point-to-point: every rank sends exactly its rows and `dst` receives each band straight into its place in the
    frame -- balanced bands differ in height several times over (8 bands of the 10^7-triangle head: 224 to 1 200
    rows), and a `dist.gather` of bands padded to the tallest moved 2.2x the frame and copied it once more."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return local_band
    local_band = _stage(local_band).contiguous()
    spans = [band_rows(height, r, world, tile) for r in range(world)] if spans is None else list(spans)
    lo, hi = spans[rank]
    assert local_band.shape[0] == hi - lo, (tuple(local_band.shape), spans[rank])
    if rank != dst:
        if hi > lo:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_band, dst)]):
                req.wait()
        return None
    full = torch.empty((height,) + tuple(local_band.shape[1:]), dtype=local_band.dtype, device=local_band.device)
    full[lo:hi] = local_band
    ops = [dist.P2POp(dist.irecv, full[a:b], r) for r, (a, b) in enumerate(spans) if r != dst and b > a]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return full
