"""Host-side mirror of the reference's plugin interfaces for the accelerated path, on top of the C-ABI.

    Mi355Renderer.RenderSceneIntoPPM(scene)   <->  RenderStrategy::RenderSceneIntoPPM(Scene*)      (src/RenderStrategy.h:16,
                                                   DefaultRenderer, src/DefaultRenderer.cpp:35-162)
    Mi355Accelerator.Intersect(rays)          <->  AccelerationStructure::Intersect(Ray&, ...)      (src/AccelerationStructure.h:21)

Everything numerical happens inside librt_mi355.so (hand-written HIP for gfx950).  There is no CPU fallback: if the
library or a GPU is missing, construction raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librt_mi355.so")
_lib = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rt_mi355 error %d: %s" % (code, msg))
        self.code = code


def library():
    """Loads librt_mi355.so (built in-tree by `make -C ray-tracer_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(abi.RT_ERR_INVALID, "%s is missing: build it with `make -C ray-tracer_amd/csrc` (no fallback path exists)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.rt_abi_version.restype = C.c_uint32
    L.rt_ctx_create.argtypes = [C.c_int, P(C.c_void_p)]
    L.rt_ctx_destroy.argtypes = [C.c_void_p]
    L.rt_ctx_destroy.restype = None
    L.rt_last_error.argtypes = [C.c_void_p]
    L.rt_last_error.restype = C.c_char_p
    L.rt_scene_upload.argtypes = [C.c_void_p, P(abi.SceneDesc)]
    L.rt_scene_bvh_info.argtypes = [C.c_void_p, P(abi.BvhInfo)]
    L.rt_render.argtypes = [C.c_void_p, P(abi.CameraDesc), P(abi.RenderOpts), C.c_void_p, C.c_void_p, C.c_void_p, P(abi.Stats)]
    L.rt_intersect_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, P(abi.Stats)]
    L.rt_intersect_batch_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.rt_noalloc_scratch_bytes.restype = C.c_size_t
    L.rt_owned_pixel_count.argtypes = [C.c_int] * 5
    L.rt_owned_pixel_count.restype = C.c_int64
    L.rt_tile_pixel_order.argtypes = [C.c_int] * 5 + [C.c_void_p]
    L.rt_perlin_tables.argtypes = [C.c_int, C.c_void_p]
    L.rt_resolve_8bit.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.rt_host_bvh_build.argtypes = [P(abi.SceneDesc), P(C.c_void_p)]
    L.rt_host_bvh_info.argtypes = [C.c_void_p, P(abi.BvhInfo)]
    L.rt_host_bvh_leaves.argtypes = [C.c_void_p] * 6
    L.rt_host_bvh_cones.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_host_bvh_cones.restype = C.c_int64
    L.rt_cone_allows_prune.argtypes = [C.c_uint32, C.c_void_p]
    L.rt_host_bvh_free.argtypes = [C.c_void_p]
    L.rt_host_bvh_free.restype = None
    L.rt_abi_sizeof.argtypes = [C.c_int]
    L.rt_image_encode_bound.argtypes = [C.c_int, C.c_int, C.c_int]
    L.rt_image_encode_bound.restype = C.c_size_t
    L.rt_image_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, P(C.c_size_t), P(C.c_double)]
    L.rt_tonemap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, P(abi.TonemapOpts), C.c_void_p, C.c_void_p, P(C.c_double)]
    L.rt_ply_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, P(abi.PlyLayout), C.c_void_p, C.c_void_p, P(C.c_int64), P(C.c_double)]
    L.rt_scene_header_bytes.restype = C.c_size_t
    L.rt_scene_export.argtypes = [C.c_void_p, C.c_void_p, P(C.c_uint64)]
    L.rt_scene_export_arena.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.rt_scene_import.argtypes = [C.c_void_p, C.c_void_p, P(C.c_void_p)]
    L.rt_ply_decode_to.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, P(abi.PlyLayout), C.c_void_p, C.c_void_p, C.c_int32, P(C.c_int64), P(C.c_double)]
    if L.rt_abi_version() != abi.RT_ABI_VERSION:
        raise RtError(abi.RT_ERR_ABI, "librt_mi355.so ABI %d != python mirror %d" % (L.rt_abi_version(), abi.RT_ABI_VERSION))
    _lib = L
    return L


def perlin_tables(n):
    out = np.zeros((max(n, 0), 16), np.int32)
    rc = library().rt_perlin_tables(n, out.ctypes.data)
    if rc != 0:
        raise RtError(rc, "rt_perlin_tables")
    return [out[i].copy() for i in range(n)]


def tile_pixel_order(nx, ny, tile_size, rank, world):
    n = library().rt_owned_pixel_count(nx, ny, tile_size, rank, world)
    out = np.zeros(n, np.int32)
    rc = library().rt_tile_pixel_order(nx, ny, tile_size, rank, world, out.ctypes.data)
    if rc != 0:
        raise RtError(rc, "rt_tile_pixel_order")
    return out


def resolve_8bit(hdr):
    hdr = np.ascontiguousarray(hdr, np.float32)
    out = np.zeros(hdr.shape, np.uint8)
    library().rt_resolve_8bit(hdr.ctypes.data, hdr.size // 3, out.ctypes.data)
    return out


def host_bvh(soup, gpu_device=None):
    """BVH view: dict with info, reference leaf ranges / primitive order / boxes and device leaf runs.  Built on the host
    (no GPU needed) or, with gpu_device, by the GPU builder."""
    L = library()
    d = soup.desc()
    h = C.c_void_p()
    rc = L.rt_host_bvh_build(C.byref(d), C.byref(h)) if gpu_device is None else L.rt_host_bvh_build_gpu(C.byref(d), int(gpu_device), C.byref(h))
    if rc != 0:
        raise RtError(rc, L.rt_last_error(None).decode())
    try:
        info = abi.BvhInfo()
        L.rt_host_bvh_info(h, C.byref(info))
        se = np.zeros((info.n_leaves, 2), np.int64)
        n_prims = len(soup.prim_shape) if soup.prim_runs is None else int(sum(int(r["count"]) for r in soup.prim_runs))
        order = np.zeros(n_prims, np.int32)
        boxes = np.zeros((info.n_leaves, 6), np.float32)
        dsc = np.zeros((info.n_leaves, 2), np.int64)
        dshape = np.zeros(info.n_leaf_prims, np.int32)
        L.rt_host_bvh_leaves(h, se.ctypes.data, order.ctypes.data, boxes.ctypes.data, dsc.ctypes.data, dshape.ctypes.data)
        nc = L.rt_host_bvh_cones(h, None, None)
        cw = np.zeros(max(nc, 0), np.uint32)
        cr = np.zeros((max(nc, 0), 2), np.int64)
        if nc > 0:
            L.rt_host_bvh_cones(h, cw.ctypes.data, cr.ctypes.data)
        return {"info": info, "leaf_start_end": se, "prim_order": order, "leaf_boxes": boxes, "dev_leaf_start_count": dsc, "dev_prim_shape": dshape,
                "cone_words": cw, "cone_ranges": cr}
    finally:
        L.rt_host_bvh_free(h)


class Context:
    """One rt_ctx: a HIP device, an uploaded scene and the wavefront buffers."""

    def __init__(self, device=0):
        self.L = library()
        self.h = C.c_void_p()
        rc = self.L.rt_ctx_create(device, C.byref(self.h))
        if rc != 0:
            raise RtError(rc, self.L.rt_last_error(None).decode())
        self.device = device
        self.soup = None

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, self.L.rt_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, soup):
        d = soup.desc()
        self._check(self.L.rt_scene_upload(self.h, C.byref(d)))
        self.soup = soup
        return self

    # ---- one build, many GPUs (rt_scene_export / rt_scene_import) --------------------------------------------------
    def export_scene_header(self):
        """(header bytes, [byte size of each arena]) of the uploaded scene."""
        hdr = np.zeros(self.L.rt_scene_header_bytes(), np.uint8)
        sizes = (C.c_uint64 * abi.RT_SCENE_ARENAS)()
        self._check(self.L.rt_scene_export(self.h, hdr.ctypes.data, sizes))
        return hdr, [int(x) for x in sizes]

    def export_scene_arena(self, index, dst_ptr):
        """Copies arena `index` to dst_ptr (device or host memory of at least that arena's size)."""
        self._check(self.L.rt_scene_export_arena(self.h, int(index), dst_ptr))

    def import_scene(self, header, arena_ptrs, soup=None):
        """Adopts a scene exported by another context / rank: header as exported, arena_ptrs = device or host addresses of the
        received bytes (0 for an empty arena).  `soup` (cameras etc.) is kept for render(cam=index)."""
        hdr = np.ascontiguousarray(header, np.uint8)
        ptrs = (C.c_void_p * abi.RT_SCENE_ARENAS)(*[C.c_void_p(int(p) or None) for p in arena_ptrs])
        self._check(self.L.rt_scene_import(self.h, hdr.ctypes.data, ptrs))
        self.soup = soup
        return self

    def bvh_info(self):
        info = abi.BvhInfo()
        self._check(self.L.rt_scene_bvh_info(self.h, C.byref(info)))
        return info

    def make_opts(self, seed=0, tile_size=0, tile_rank=0, tile_world=1, output_on_device=False, compact_output=False,
                  max_batch_samples=0, timing=0, num_streams=0, level_counts=0, shadow_stream=0):
        o = abi.RenderOpts()
        o.abi_version = abi.RT_ABI_VERSION
        o.seed, o.tile_size, o.tile_rank, o.tile_world = seed, tile_size, tile_rank, tile_world
        o.output_on_device, o.compact_output = int(output_on_device), int(compact_output)
        o.max_batch_samples, o.timing, o.num_streams = max_batch_samples, int(timing), int(num_streams)
        o.level_counts = int(level_counts)      # 1: the host reads every level's length (rt_render_opts.level_counts)
        o.shadow_stream = int(shadow_stream)    # 1: shadow rays and lighting on a second stream beside the closest-hit chain
        return o

    def render(self, cam=0, want_ids=True, **kw):
        """Renders to host numpy arrays.  `cam` is a camera index of the uploaded soup or an abi.CameraDesc."""
        cam_over = {k: kw.pop(k) for k in list(kw) if k in ("num_samples", "nx", "ny", "plane", "focus_distance", "aperture_size")}
        cd = cam if isinstance(cam, abi.CameraDesc) else self.soup.camera_desc(cam, **cam_over)
        opts = self.make_opts(**kw)
        if opts.compact_output:
            n = self.L.rt_owned_pixel_count(cd.nx, cd.ny, opts.tile_size, opts.tile_rank, opts.tile_world)
            shape3, shape1 = (n, 3), (n,)
        else:
            shape3, shape1 = (cd.ny, cd.nx, 3), (cd.ny, cd.nx)
        hdr = np.zeros(shape3, np.float32)
        pid = np.zeros(shape1, np.int32) if want_ids else None
        pt = np.zeros(shape1, np.float32) if want_ids else None
        st = abi.Stats()
        self._check(self.L.rt_render(self.h, C.byref(cd), C.byref(opts), hdr.ctypes.data, pid.ctypes.data if want_ids else None,
                                     pt.ctypes.data if want_ids else None, C.byref(st)))
        return hdr, pid, pt, st

    def render_device(self, cd, opts, hdr_ptr, pid_ptr=None, pt_ptr=None, stats=None):
        """Renders into caller-provided DEVICE buffers (opts.output_on_device must be 1)."""
        st = stats if stats is not None else abi.Stats()
        self._check(self.L.rt_render(self.h, C.byref(cd), C.byref(opts), hdr_ptr, pid_ptr, pt_ptr, C.byref(st)))
        return st

    def intersect(self, rays, stats=False):
        rays = np.ascontiguousarray(rays, abi.ray_dtype)
        hits = np.zeros(len(rays), abi.hit_dtype)
        st = abi.Stats()
        self._check(self.L.rt_intersect_batch(self.h, rays.ctypes.data, len(rays), hits.ctypes.data, 0, C.byref(st) if stats else None))
        return (hits, st) if stats else hits

    def intersect_device(self, rays_ptr, n, hits_ptr, noalloc=False, stream=None):
        """Closest-hit queries on DEVICE buffers (rt_ray[n] in, rt_hit[n] out).  noalloc selects the entry point that
        allocates nothing (fixed scratch inside the library); `stream` (a hipStream_t as int) only applies to it."""
        if noalloc:
            self._check(self.L.rt_intersect_batch_noalloc(self.h, rays_ptr, n, hits_ptr, stream))
        else:
            self._check(self.L.rt_intersect_batch(self.h, rays_ptr, n, hits_ptr, 1, None))

    def encode_image(self, hdr, fmt=abi.RT_IMG_PPM_P3, nx=None, ny=None, timing=False):
        """8-bit resolve + image encoding on the device (ObtainColorFromUnclampedVector, Image::SaveImageAsPPM).  `hdr` is a
        float32 numpy frame [ny, nx, 3] or a device pointer (int) with nx, ny given.  Returns the encoded bytes."""
        on_device = isinstance(hdr, int)
        if not on_device:
            hdr = np.ascontiguousarray(hdr, np.float32)
            ny, nx = hdr.shape[0], hdr.shape[1]
        cap = self.L.rt_image_encode_bound(nx, ny, fmt)
        out = np.empty(max(cap, 1), np.uint8)
        n, ms = C.c_size_t(0), C.c_double(0)
        self._check(self.L.rt_image_encode(self.h, hdr if on_device else hdr.ctypes.data, nx, ny, int(on_device), fmt, out.ctypes.data, cap,
                                           C.byref(n), C.byref(ms) if timing else None))
        return (out[:n.value], ms.value) if timing else out[:n.value]

    def decode_ply(self, vertex_bytes, face_bytes, layout, timing=False):
        """Body of a binary_little_endian PLY -> (positions float32 [nv, 3], triangles int32 [nt, 3]) on the device
        (rt_ply_decode: SceneParser.cpp:983-1023).  vertex_bytes / face_bytes: bytes-like, starting at the first vertex /
        face record; layout: abi.PlyLayout filled from the header (scene.ply_layout)."""
        vb = np.frombuffer(vertex_bytes, np.uint8)
        fb = np.frombuffer(face_bytes, np.uint8)
        fsize = len(vb) if layout.format == abi.RT_PLY_ASCII else len(fb)       # an ascii body is passed whole as vertex_bytes
        pos = np.empty((int(layout.n_vertices), 3), np.float32)
        tris = np.empty((2 * int(layout.n_faces), 3), np.int32)
        n = C.c_int64(0)
        ms = C.c_double(0)
        self._check(self.L.rt_ply_decode(self.h, vb.ctypes.data if len(vb) else None, fb.ctypes.data if len(fb) else None, fsize, C.byref(layout),
                                         pos.ctypes.data, tris.ctypes.data, C.byref(n), C.byref(ms) if timing else None))
        tris = tris[:n.value]
        return (pos, tris, ms.value) if timing else (pos, tris)

    def decode_ply_to(self, vertex_bytes, face_bytes, layout, d_pos_ptr, d_tris_ptr, index_offset=0):
        """rt_ply_decode_to: the same decoder with its output left in HBM (positions -> DEVICE float[nv*3] at d_pos_ptr,
        triangles -> DEVICE int32[<= 2*nf*3] at d_tris_ptr with index_offset added).  Returns the number of triangles."""
        vb = np.frombuffer(vertex_bytes, np.uint8)
        fb = np.frombuffer(face_bytes, np.uint8)
        fsize = len(vb) if layout.format == abi.RT_PLY_ASCII else len(fb)
        n = C.c_int64(0)
        self._check(self.L.rt_ply_decode_to(self.h, vb.ctypes.data if len(vb) else None, fb.ctypes.data if len(fb) else None, fsize, C.byref(layout),
                                            d_pos_ptr, d_tris_ptr, int(index_offset), C.byref(n), None))
        return int(n.value)

    def tonemap(self, hdr, key, burn_percent, saturation, gamma=2.2, input_8bit=True, nx=None, ny=None, timing=False):
        """Tonemapper::Tonemap (DefaultTonemapStrategy) on the device.  Returns (rgb float32 [ny, nx, 3], (average luminance,
        luminance white)[, ms])."""
        on_device = isinstance(hdr, int)
        if not on_device:
            hdr = np.ascontiguousarray(hdr, np.float32)
            ny, nx = hdr.shape[0], hdr.shape[1]
        o = abi.TonemapOpts(abi.RT_ABI_VERSION, key, burn_percent, saturation, gamma, int(input_8bit), int(on_device), 0)
        out = np.zeros((ny, nx, 3), np.float32)
        params = np.zeros(2, np.float32)
        ms = C.c_double(0)
        self._check(self.L.rt_tonemap(self.h, hdr if on_device else hdr.ctypes.data, nx, ny, C.byref(o), out.ctypes.data, params.ctypes.data,
                                      C.byref(ms) if timing else None))
        return (out, params, ms.value) if timing else (out, params)


class Mi355Accelerator:
    """AccelerationStructure mirror (src/AccelerationStructure.h:12-26): built from a scene's primitives, answers closest-
    hit queries — batched, because one ray per call cannot feed a GPU.  "No hit" is shape == -1 (reference: mat == nullptr)."""

    def __init__(self, scene, device=0, context=None):
        self.ctx = context or Context(device)
        if context is None or context.soup is not scene:
            self.ctx.upload(scene)

    def Intersect(self, rays):
        return self.ctx.intersect(rays)


class Mi355Renderer:
    """RenderStrategy mirror (src/RenderStrategy.h:12-26).  RenderSceneIntoPPM renders every camera of the scene and
    writes `camera.image_name` as a P3 text PPM exactly like Image::SaveImageAsPPM (src/Image.cpp:54-77); the float HDR
    frames stay available in `self.frames` (the reference only ever keeps the 8-bit image, Q20)."""

    def __init__(self, device=0, seed=0, out_dir=".", frames_in_flight=1):
        """frames_in_flight = 2: scenes with several cameras are rendered by two contexts that share one built scene (the second adopts
        the first one's arenas, rt_scene_export / rt_scene_import), each driven by a host thread of its own: one context's launch tails
        overlap the other's launches (config 2: 7.1 instead of 8.1 ms per frame).  The files are the same either way."""
        self.ctx = Context(device)
        self.device = device
        self.seed = seed
        self.out_dir = out_dir
        self.frames_in_flight = max(1, int(frames_in_flight))
        self.frames = {}
        self.stats = {}

    def _render_camera(self, ctx, scene, i):
        cam = scene.cameras[i]
        hdr, pid, pt, st = ctx.render(i, seed=self.seed, timing=1)
        self.frames[cam["image_name"]] = hdr
        self.stats[cam["image_name"]] = st
        tm = getattr(scene, "tonemap", None)
        if tm is None:                                                            # Image::SaveImage (Image.cpp:38-44)
            data = ctx.encode_image(hdr, abi.RT_IMG_PPM_P3)
        else:                                                                     # SaveImageAsEXR: tone map the 8-bit image, HALF EXR
            mapped, _ = ctx.tonemap(hdr, tm[0], tm[1], tm[2], tm[3], input_8bit=True)
            data = ctx.encode_image(mapped, abi.RT_IMG_EXR_HALF)
        with open(os.path.join(self.out_dir, cam["image_name"]), "wb") as f:
            f.write(data.tobytes())

    def RenderSceneIntoPPM(self, scene):
        self.ctx.upload(scene)               # RetrieveRenderingParamsFromScene: builds the accelerator
        n_ctx = min(self.frames_in_flight, len(scene.cameras))
        if n_ctx <= 1:
            for i in range(len(scene.cameras)):
                self._render_camera(self.ctx, scene, i)
            return
        import threading
        hdr, sizes = self.ctx.export_scene_header()
        arenas = [np.zeros(max(nb, 1), np.uint8) for nb in sizes]                 # host copies: the only device allocator here is the library's own
        for ai, nb in enumerate(sizes):
            if nb:
                self.ctx.export_scene_arena(ai, arenas[ai].ctypes.data)
        ctxs = [self.ctx] + [Context(self.device).import_scene(hdr, [a.ctypes.data if nb else 0 for a, nb in zip(arenas, sizes)], soup=scene)
                             for _ in range(n_ctx - 1)]
        del arenas
        todo = list(range(len(scene.cameras)))
        lock = threading.Lock()
        errors = []

        def worker(ctx):
            while True:
                with lock:
                    if not todo or errors:
                        return
                    i = todo.pop(0)
                try:
                    self._render_camera(ctx, scene, i)
                except Exception as e:                                            # noqa: BLE001 — re-raised on the caller's thread
                    errors.append(e)
                    return
        threads = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for c in ctxs[1:]:
            c.close()
        if errors:
            raise errors[0]

    def close(self):
        self.ctx.close()


def write_ppm(path, rgb8):
    """Host writer for an 8-bit numpy image (Context.encode_image is the device path).  Image::SaveImageAsPPM (src/Image.cpp:54-77): "P3\\n<w> <h>\\n255\\n", then per image row every channel as "%d "
    and a newline."""
    h, w, _ = rgb8.shape
    with open(path, "w") as f:
        f.write("P3\n%d %d\n255\n" % (w, h))
        for y in range(h):
            f.write("".join("%d " % v for v in rgb8[y].reshape(-1)))
            f.write("\n")
