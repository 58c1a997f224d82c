"""Tile-sharded frames over several ranks (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The render path has no data-path collective: every rank renders the interleaved tiles it owns (rt_render with
tile_rank / tile_world, compact output).  The only exchange is the per-frame gather of the HDR tiles on rank 0 — the
step that precedes tonemapping / file output in the reference (src/DefaultRenderer.cpp:78).  Disjoint tiles => a
point-to-point gather (7 peers -> rank 0, one xGMI link each), not an all-reduce.
"""
import ctypes as C
import os

import numpy as np

from . import renderer


class TileGather:
    """Pre-computes the tile ownership of a frame and gathers / de-interleaves compact tile buffers on rank 0."""

    def __init__(self, nx, ny, tile_size, rank, world, device, torch, host_staging=False):
        """host_staging: the process group cannot take device tensors (gloo): the tiles rendered into HBM are copied to host
        memory with a synchronous copy, gathered there, and the frame is assembled on the host (a rehearsal path for boxes
        where several ranks share one GPU; production uses the nccl backend = RCCL over xGMI on device tensors)."""
        self.nx, self.ny, self.tile, self.rank, self.world = nx, ny, tile_size, rank, world
        self.torch = torch
        self.host_staging = bool(host_staging)
        L = renderer.library()
        self.counts = [int(L.rt_owned_pixel_count(nx, ny, tile_size, r, world)) for r in range(world)]
        self.max_owned = max(self.counts)
        # Two compact tile buffers: frame i+1 is rendered into one while the gather of frame i still reads the other
        # (rt_render runs on the library's own stream; the gather is enqueued on torch's / RCCL's).  acquire() makes the
        # re-use of a buffer wait for the gather that last read it.
        self.buffers = [torch.zeros((self.max_owned, 3), dtype=torch.float32, device=device) for _ in range(2)]
        self.events = [None, None]
        self.cur = 0
        self.tile_hdr = self.buffers[0]
        gdev = torch.device("cpu") if self.host_staging else device
        self.gathered = torch.zeros((world, self.max_owned, 3), dtype=torch.float32, device=gdev) if rank == 0 else None
        self.frame = torch.zeros((ny * nx, 3), dtype=torch.float32, device=gdev) if rank == 0 else None
        self.orders = None
        if rank == 0:
            self.orders = [torch.from_numpy(renderer.tile_pixel_order(nx, ny, tile_size, r, world).astype(np.int64)).to(gdev)
                           for r in range(world)]

    def acquire(self):
        """The buffer the next frame is rendered into (self.tile_hdr), safe to overwrite."""
        self.cur ^= 1
        self.tile_hdr = self.buffers[self.cur]
        if self.events[self.cur] is not None:
            self.events[self.cur].synchronize()          # the gather that read this buffer two frames ago has finished
        return self.tile_hdr

    def gather(self, dist=None):
        """tile_hdr (this rank's compact pixels) -> rank 0's full frame [ny*nx, 3].  Returns the frame on rank 0.  The work
        is enqueued, not waited for (torch.cuda.synchronize() / the next acquire() of the same buffer do that)."""
        try:
            return self._gather(dist)
        finally:
            if self.tile_hdr.is_cuda:
                ev = self.torch.cuda.Event()
                ev.record()
                self.events[self.cur] = ev

    def _gather(self, dist=None):
        src = self.tile_hdr.cpu() if self.host_staging else self.tile_hdr        # .cpu(): synchronous device-to-host copy
        # a process group of one rank still goes through the collective, so that the one-GPU rehearsal runs the calls a node runs
        if self.world > 1 or (dist is not None and dist.is_available() and dist.is_initialized()):
            dist.gather(src, list(self.gathered.unbind(0)) if self.rank == 0 else None, dst=0)
            if self.rank == 0:
                for r in range(self.world):
                    self.frame.index_copy_(0, self.orders[r], self.gathered[r, :self.counts[r]])
        elif self.rank == 0:
            self.frame.index_copy_(0, self.orders[0], src[:self.counts[0]])
        return self.frame


def broadcast_scene(ctx, dist, rank, src, device, torch, host_staging=False, soup=None):
    """One build, many GPUs: rank `src` has uploaded (built) the scene in `ctx`; every other rank receives the header and the flat
    arenas over the process group and adopts them (rt_scene_export / rt_scene_import) instead of building its own copy.  With the nccl
    backend the arenas travel GPU to GPU (RCCL broadcast over xGMI); host_staging=True is for groups that only take CPU tensors (gloo).
    The default remains one build per rank (DESIGN: every rank has its own PCIe link and the builds run in parallel); this is for hosts
    where the description cannot be read on every rank, or where the host link is the bottleneck."""
    n_arenas = renderer.abi.RT_SCENE_ARENAS
    tdev = torch.device("cpu") if host_staging else device
    hdr_bytes = int(renderer.library().rt_scene_header_bytes())
    meta = torch.zeros(hdr_bytes + 8 * n_arenas, dtype=torch.uint8)
    if rank == src:
        hdr, sizes = ctx.export_scene_header()
        meta[:hdr_bytes] = torch.from_numpy(hdr)
        meta[hdr_bytes:] = torch.from_numpy(np.array(sizes, np.uint64).view(np.uint8))
    meta_t = meta.to(tdev)
    dist.broadcast(meta_t, src=src)
    meta = meta_t.cpu()                                           # synchronous
    sizes = [int(x) for x in meta[hdr_bytes:].numpy().view(np.uint64)]
    bufs = []
    for i, nbytes in enumerate(sizes):
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=tdev)
        if rank == src and nbytes:
            ctx.export_scene_arena(i, buf.data_ptr())             # device -> device (or -> host when staging); returns when the copy is done
        if nbytes:
            if not host_staging:
                torch.cuda.synchronize(device)
            dist.broadcast(buf, src=src)
        bufs.append(buf)
    if rank != src:
        if not host_staging:
            torch.cuda.synchronize(device)
        ctx.import_scene(meta[:hdr_bytes].numpy(), [b.data_ptr() if n else 0 for b, n in zip(bufs, sizes)], soup=soup)
    return sizes


# ---- the same step through the C-ABI of librt_mi355_mgpu.so (include/rt_mi355_mgpu.h): RCCL called from C++, no torch.distributed ----
MGPU_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "librt_mi355_mgpu.so")
RT_GATHER_ID_BYTES = 128
_mgpu = None


def mgpu_library():
    """Loads librt_mi355_mgpu.so (needs librt_mi355.so and RCCL; built by `make -C ray-tracer_amd/csrc`).  No fallback."""
    global _mgpu
    if _mgpu is not None:
        return _mgpu
    renderer.library()                                     # the render library first: the gather library binds its symbols
    if not os.path.exists(MGPU_LIB_PATH):
        raise renderer.RtError(renderer.abi.RT_ERR_INVALID, "%s is missing: build it with `make -C ray-tracer_amd/csrc`" % MGPU_LIB_PATH)
    L = C.CDLL(MGPU_LIB_PATH)
    P = C.POINTER
    L.rt_gather_create.argtypes = [P(C.c_int), C.c_int, P(C.c_void_p)]
    L.rt_gather_unique_id.argtypes = [C.c_void_p]
    L.rt_gather_create_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, P(C.c_void_p)]
    L.rt_gather_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, P(C.c_void_p), C.c_void_p]
    L.rt_scene_broadcast.argtypes = [C.c_void_p, P(C.c_void_p)]
    L.rt_gather_buffers.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, P(C.c_void_p), P(C.c_void_p)]
    L.rt_gather_copy_frame.argtypes = [C.c_void_p, C.c_void_p]
    L.rt_gather_world.argtypes = [C.c_void_p]
    L.rt_gather_destroy.argtypes = [C.c_void_p]
    L.rt_gather_destroy.restype = None
    L.rt_gather_last_error.argtypes = [C.c_void_p]
    L.rt_gather_last_error.restype = C.c_char_p
    L.rt_gather_deinterleave_host.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_void_p), C.c_void_p]
    _mgpu = L
    return L


def deinterleave_host(nx, ny, tile_size, tiles):
    """rt_gather_deinterleave_host: the ranks' compact [owned, 3] float32 arrays -> the [ny, nx, 3] frame (host memory, no GPU)."""
    L = mgpu_library()
    tiles = [np.ascontiguousarray(t, np.float32) for t in tiles]
    ptrs = (C.c_void_p * len(tiles))(*[t.ctypes.data if t.size else None for t in tiles])
    frame = np.zeros((ny, nx, 3), np.float32)
    rc = L.rt_gather_deinterleave_host(nx, ny, tile_size, len(tiles), ptrs, frame.ctypes.data)
    if rc != 0:
        raise renderer.RtError(rc, "rt_gather_deinterleave_host")
    return frame


class RcclGather:
    """rt_gather of the C-ABI: one process that drives `devices` (rank r on devices[r]), or — with `unique_id` — one rank of a
    one-process-per-GPU job.  tiles / frame arguments are device pointers (ints)."""

    def __init__(self, devices=None, unique_id=None, rank=0, world=1, device=0):
        self.L = mgpu_library()
        self.h = C.c_void_p()
        if unique_id is None:
            devs = (C.c_int * len(devices))(*devices)
            rc = self.L.rt_gather_create(devs, len(devices), C.byref(self.h))
            self.n_local = len(devices)
        else:
            buf = (C.c_ubyte * RT_GATHER_ID_BYTES).from_buffer_copy(bytes(unique_id))
            rc = self.L.rt_gather_create_rank(buf, rank, world, device, C.byref(self.h))
            self.n_local = 1
        if rc != 0:
            raise renderer.RtError(rc, self.L.rt_gather_last_error(None).decode())

    @staticmethod
    def unique_id():
        L = mgpu_library()
        buf = (C.c_ubyte * RT_GATHER_ID_BYTES)()
        rc = L.rt_gather_unique_id(buf)
        if rc != 0:
            raise renderer.RtError(rc, L.rt_gather_last_error(None).decode())
        return bytes(buf)

    def _check(self, rc):
        if rc != 0:
            raise renderer.RtError(rc, self.L.rt_gather_last_error(self.h).decode())

    def world(self):
        return int(self.L.rt_gather_world(self.h))

    def gather_tiles(self, nx, ny, tile_size, tile_ptrs, frame_ptr):
        assert len(tile_ptrs) == self.n_local
        ptrs = (C.c_void_p * self.n_local)(*[int(p) if p else None for p in tile_ptrs])
        self._check(self.L.rt_gather_tiles(self.h, nx, ny, tile_size, ptrs, int(frame_ptr) if frame_ptr else None))

    def buffers(self, nx, ny, tile_size):
        """Library-owned device buffers for this frame geometry: ([tile buffer address per local rank], frame address or None)."""
        tiles = (C.c_void_p * self.n_local)()
        frame = C.c_void_p()
        self._check(self.L.rt_gather_buffers(self.h, nx, ny, tile_size, tiles, C.byref(frame)))
        self._frame_shape = (ny, nx, 3)
        return [int(t or 0) for t in tiles], frame.value

    def copy_frame(self):
        """The frame of the last buffers() geometry as a host array [ny, nx, 3]."""
        out = np.zeros(self._frame_shape, np.float32)
        self._check(self.L.rt_gather_copy_frame(self.h, out.ctypes.data))
        return out

    def broadcast_scene(self, contexts):
        """contexts: the renderer.Context of every local rank; rank 0's holds the scene, the others receive it."""
        assert len(contexts) == self.n_local
        ptrs = (C.c_void_p * self.n_local)(*[c.h.value for c in contexts])
        self._check(self.L.rt_scene_broadcast(self.h, ptrs))

    def close(self):
        if self.h:
            self.L.rt_gather_destroy(self.h)
            self.h = C.c_void_p()
