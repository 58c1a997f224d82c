"""GPU tests of the two drop-in boundaries at BASELINE sizes: rt_intersect_batch (AccelerationStructure::Intersect) on
random rays, the full config-2 frame (1M triangles) against the oracle, tile sharding, determinism, error paths."""
import importlib

import numpy as np
import pytest

from parity_util import compare_frames

pytestmark = pytest.mark.gpu
abi = importlib.import_module("ray-tracer_amd").abi
synth = importlib.import_module("ray-tracer_amd.synth")


def random_rays(n, seed, center, radius):
    rng = np.random.RandomState(seed)
    r = np.zeros(n, abi.ray_dtype)
    o = rng.randn(n, 3)
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * radius * (1.0 + rng.rand(n, 1)) + center
    tgt = center + (rng.rand(n, 3) - 0.5) * radius
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # edge cases the reference tests implicitly: axis-aligned directions (d.x == 0 => never a hit, Q2), zero components
    d[:64, 0] = 0.0
    d[64:128, 1] = 0.0
    d[128:192] = [0, 0, -1]
    d[:192] /= np.maximum(np.linalg.norm(d[:192], axis=1, keepdims=True), 1e-30)
    r["origin"], r["direction"], r["time"] = o.astype(np.float32), d.astype(np.float32), rng.rand(n).astype(np.float32)
    return r


@pytest.mark.parametrize("name,center,radius", [("scienceTree", (2, 2, 0), 6.0), ("cornellbox_boxes_dynamic", (0, 0, 0), 12.0),
                                                ("spheres", (0, 0, 0), 8.0), ("bunny", (0, 0.1, 0), 0.3)])
def test_intersect_batch(rt, orc, gpu_ctx, scene_path, name, center, radius):
    soup = rt.SceneSoup.load(scene_path(name))
    acc = rt.Mi355Accelerator(soup, context=gpu_ctx)
    rays = random_rays(20000, 5, np.array(center, np.float64), radius)
    g = acc.Intersect(rays)
    c = orc.OracleScene(soup).intersect(rays)
    assert np.array_equal(g["shape"], c["shape"])
    hit = c["shape"] >= 0
    assert hit.sum() > 500
    assert np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))
    assert np.array_equal(g["ip"][hit].view(np.uint32), c["ip"][hit].view(np.uint32))
    assert np.array_equal(g["n"][hit].view(np.uint32), c["n"][hit].view(np.uint32))
    sph = hit & ((c["shape"] & abi.RT_PRIM_SPHERE) != 0)
    assert np.array_equal(g["uv"][hit & ~sph].view(np.uint32), c["uv"][hit & ~sph].view(np.uint32))
    np.testing.assert_allclose(g["uv"][sph], c["uv"][sph], atol=2e-6)          # acosf / atan2f: last-ulp differences allowed
    assert (g["shape"][:64] == -1).all()                                          # d.x == 0


def test_empty_batch_and_empty_scene(rt, gpu_ctx):
    b = rt.SceneBuilder()
    b.add_material()
    b.add_camera((0, 0, 5), (0, 0, -1), (0, 1, 0), (-1, 1, -1, 1), 1, 16, 8)
    soup = b.finish()
    soup.background = np.array([3, 4, 5], np.float32)
    gpu_ctx.upload(soup)
    assert len(gpu_ctx.intersect(np.zeros(0, abi.ray_dtype))) == 0
    h = gpu_ctx.intersect(random_rays(256, 1, np.zeros(3), 3.0))
    assert (h["shape"] == -1).all()
    hdr, pid, pt, st = gpu_ctx.render(0)
    assert (pid == -1).all() and np.allclose(hdr, [3, 4, 5])


@pytest.fixture(scope="module")
def dragon(rt, orc, gpu_ctx):
    soup = synth.dragon_standin_scene(1_000_000, 800, 480)
    osc = orc.OracleScene(soup)
    c = osc.render(0, nthreads=16, dynamic=True)
    osc.close()
    return soup, c


def test_config2_full_frame(rt, gpu_ctx, dragon):
    """BASELINE config 2 at full size: 1M-triangle mesh + mirror floor, 800x480, depth 6 (9.3M rays)."""
    soup, (c_hdr, c_id, c_t, c_st) = dragon
    gpu_ctx.upload(soup)
    g_hdr, g_id, g_t, g_st = gpu_ctx.render(0, timing=1)
    m = compare_frames(g_hdr, c_hdr, g_id, c_id, g_t, c_t)
    assert list(g_st.rays) == list(c_st.rays)
    print("config2", m, list(g_st.rays), "kernels %.2f ms" % g_st.ms_kernels)
    # determinism: a second frame is bit-identical (queue order is not, the result is)
    h2, i2, t2, _ = gpu_ctx.render(0)
    assert np.array_equal(h2.view(np.uint32), g_hdr.view(np.uint32)) and np.array_equal(i2, g_id)
    # small batches (many wavefront batches per frame) give the same frame
    h3, i3, _, _ = gpu_ctx.render(0, max_batch_samples=50000)
    assert np.array_equal(h3.view(np.uint32), g_hdr.view(np.uint32)) and np.array_equal(i3, g_id)


def test_config2_tile_sharding(rt, gpu_ctx, dragon):
    """The frames of N tile ranks tile the full frame exactly (what 8 GPUs + the RCCL gather produce)."""
    soup, (c_hdr, c_id, c_t, c_st) = dragon
    gpu_ctx.upload(soup)
    full, fid, _, fst = gpu_ctx.render(0)
    for world in (2, 8):
        acc = np.zeros_like(full)
        ids = np.full_like(fid, -7)
        rays = np.zeros(4, np.int64)
        for rank in range(world):
            h, i, _, st = gpu_ctx.render(0, tile_size=32, tile_rank=rank, tile_world=world, compact_output=True)
            order = rt.renderer.tile_pixel_order(800, 480, 32, rank, world)
            acc.reshape(-1, 3)[order] = h
            ids.reshape(-1)[order] = i
            rays += np.array(list(st.rays), np.int64)
            # non-compact output of the same rank: owned pixels filled, the rest zero / -1
            if rank == 1:
                h2, i2, _, _ = gpu_ctx.render(0, tile_size=32, tile_rank=rank, tile_world=world)
                mask = np.zeros(800 * 480, bool)
                mask[order] = True
                assert np.array_equal(h2.reshape(-1, 3)[order].view(np.uint32), h.view(np.uint32))
                assert (h2.reshape(-1, 3)[~mask] == 0).all() and (i2.reshape(-1)[~mask] == -1).all()
        assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))
        assert np.array_equal(ids, fid)
        assert rays.tolist() == list(fst.rays)


def test_error_paths(rt, gpu_ctx, scene_path):
    ctx = rt.Context(0)
    with pytest.raises(rt.RtError) as e:
        ctx.intersect(np.zeros(4, abi.ray_dtype))
    assert e.value.code == abi.RT_ERR_NO_SCENE
    soup = rt.SceneSoup.load(scene_path("simple"))
    ctx.upload(soup)
    with pytest.raises(rt.RtError) as e:
        ctx.render(0, tile_size=16, tile_rank=5, tile_world=2)
    assert e.value.code == abi.RT_ERR_INVALID
    opts = ctx.make_opts()
    opts.abi_version = 1
    import ctypes as C
    cd = soup.camera_desc(0)
    hdr = np.zeros((cd.ny, cd.nx, 3), np.float32)
    assert ctx.L.rt_render(ctx.h, C.byref(cd), C.byref(opts), hdr.ctypes.data, None, None, None) == abi.RT_ERR_ABI
    ctx.close()


def test_renderer_writes_ppm(rt, scene_path, tmp_path):
    soup = rt.SceneSoup.load(scene_path("simple"))
    soup.cameras[0]["nx"], soup.cameras[0]["ny"] = 40, 30
    r = rt.Mi355Renderer(0, out_dir=str(tmp_path))
    r.RenderSceneIntoPPM(soup)
    name = soup.cameras[0]["image_name"]
    toks = open(tmp_path / name).read().split()
    assert toks[:4] == ["P3", "40", "30", "255"] and len(toks) == 4 + 40 * 30 * 3
    img = np.array(toks[4:], np.int64).reshape(30, 40, 3)
    assert np.array_equal(img, rt.renderer.resolve_8bit(r.frames[name]))
    r.close()


def test_renderer_with_tonemap_writes_exr(rt, orc, scene_path, tmp_path):
    """A scene with <Tonemap> goes Image::SaveImageAsEXR's way: 8-bit image -> tone mapper -> HALF EXR under the image name
    (Image.cpp:38-52).  The file is the oracle's EXR of the tone-mapped frame; the tone-mapped values themselves are checked
    against the oracle's tone mapper (test_gpu_image.py states the tolerance)."""
    soup = rt.SceneSoup.load(scene_path("cornellbox"))
    soup.cameras[0]["nx"], soup.cameras[0]["ny"] = 64, 48
    soup.tonemap = (0.18, 1.0, 1.0, 2.2)
    r = rt.Mi355Renderer(0, out_dir=str(tmp_path))
    r.RenderSceneIntoPPM(soup)
    name = soup.cameras[0]["image_name"]
    data = open(tmp_path / name, "rb").read()
    mapped, params = r.ctx.tonemap(r.frames[name], 0.18, 1.0, 1.0, input_8bit=True)
    assert data == orc.exr_half(mapped)
    want, wparams = orc.tonemap(orc.resolve_8bit(r.frames[name]).astype(np.float32), 0.18, 1.0, 1.0)
    ok = ~np.isnan(want)
    assert params[1] == wparams[1] and np.allclose(mapped[ok], want[ok], rtol=1e-3, atol=1e-6)
    r.close()


def test_frames_after_the_first_make_no_host_round_trip(rt, orc, gpu_ctx, scene_path):
    """The first frame of a scene reads every level's child count on the host; later frames size queues and launches from the
    previous frame's counts, keep the level lengths on the device (rt_stats.level_syncs == 0) and give the same frame bit for
    bit.  A frame that outgrows the prediction (here: the camera turns from the sky to the mirrors) is rendered a second time
    with host counts (frame_passes == 2) and is still exact."""
    soup = rt.SceneSoup.load(scene_path("cornellbox_recursive"))
    gpu_ctx.upload(soup)
    h1, i1, t1, s1 = gpu_ctx.render(0, nx=200, ny=200, timing=1)
    assert s1.level_syncs > 0 and s1.frame_passes == 1
    h2, i2, t2, s2 = gpu_ctx.render(0, nx=200, ny=200, timing=1)
    assert s2.level_syncs == 0 and s2.frame_passes == 1
    assert np.array_equal(h1.view(np.uint32), h2.view(np.uint32)) and np.array_equal(i1, i2) and np.array_equal(t1.view(np.uint32), t2.view(np.uint32))
    assert list(s1.rays) == list(s2.rays) and s1.trace_rays == s2.trace_rays
    # several batches per frame on two streams, still without a round trip
    h3, i3, _, s3 = gpu_ctx.render(0, nx=200, ny=200, max_batch_samples=6000, num_streams=2)
    assert s3.level_syncs == 0 and s3.frame_passes == 1 and np.array_equal(h3.view(np.uint32), h1.view(np.uint32))
    assert list(s3.rays) == list(s1.rays)                  # rays per kind (refraction rays included) from the level counters of every batch
    # a camera that sees nothing: almost no secondary rays ...
    cd = soup.camera_desc(0, nx=200, ny=200)
    away = type(cd).from_buffer_copy(cd)
    away.gaze[0], away.gaze[1], away.gaze[2] = 0.0, 0.0, 1.0
    h4, i4, _, s4 = gpu_ctx.render(away)
    assert (i4 == -1).all() and s4.frame_passes == 1
    # ... then back to the box: every level outgrows the prediction -> second pass, same frame as before
    h5, i5, t5, s5 = gpu_ctx.render(0, nx=200, ny=200)
    assert s5.frame_passes == 2 and s5.level_syncs > 0
    assert np.array_equal(h5.view(np.uint32), h1.view(np.uint32)) and np.array_equal(i5, i1)
    assert list(s5.rays) == list(s1.rays)
    h6, _, _, s6 = gpu_ctx.render(0, nx=200, ny=200)
    assert s6.level_syncs == 0 and s6.frame_passes == 1 and np.array_equal(h6.view(np.uint32), h1.view(np.uint32))
    c_hdr, c_id, c_t, c_st = orc.OracleScene(soup).render(0, nthreads=8, nx=200, ny=200)
    compare_frames(h6, c_hdr, i1, c_id, t1, c_t)
    assert list(s6.rays) == list(c_st.rays) and s6.rays[rt.abi.RT_RAY_REFRACT] > 0
    assert list(s6.rays) == list(c_st.rays)


def test_shadow_rays_on_a_stream_of_their_own(rt, gpu_ctx, scene_path):
    """rt_render_opts.shadow_stream = 1: in frames whose level lengths stay on the device the shadow rays of a level and the lighting
    that needs them run on a second stream beside the closest-hit chain (14 trace launches instead of 8 fused ones for depth 6).  Frames,
    hit ids, ray counts: bit for bit those of the fused launches — one batch, several batches on two streams, area lights + glossy
    (random numbers keyed per ray), and a frame whose prediction fails (second pass, host-counted, fused)."""
    for name, n in (("cornellbox_recursive", 200), ("cornellbox_brushed_metal", 96), ("scienceTree", 160)):
        soup = rt.SceneSoup.load(scene_path(name))
        for c in soup.cameras:
            c["num_samples"] = min(int(c.get("num_samples", 1)), 4)
        gpu_ctx.upload(soup)
        h0, i0, t0, s0 = gpu_ctx.render(0, nx=n, ny=n, timing=1)                    # first frame: host-counted
        h1, i1, t1, s1 = gpu_ctx.render(0, nx=n, ny=n, timing=1)                    # predicted, fused
        h2, i2, t2, s2 = gpu_ctx.render(0, nx=n, ny=n, timing=1, shadow_stream=1)   # predicted, side stream
        assert s1.level_syncs == 0 and s2.level_syncs == 0 and s2.frame_passes == 1
        assert s2.trace_launches > s1.trace_launches
        for h, i, t, st in ((h1, i1, t1, s1), (h2, i2, t2, s2)):
            assert np.array_equal(h.view(np.uint32), h0.view(np.uint32)) and np.array_equal(i, i0) and np.array_equal(t.view(np.uint32), t0.view(np.uint32))
            assert list(st.rays) == list(s0.rays)
        h3, _, _, s3 = gpu_ctx.render(0, nx=n, ny=n, max_batch_samples=5000, num_streams=2, shadow_stream=1)
        assert s3.level_syncs == 0 and np.array_equal(h3.view(np.uint32), h0.view(np.uint32)) and list(s3.rays) == list(s0.rays)
        h4, _, _, s4 = gpu_ctx.render(0, nx=n, ny=n, max_batch_samples=3000, num_streams=1, shadow_stream=1)      # several batches, one after the other, on one lane
        assert np.array_equal(h4.view(np.uint32), h0.view(np.uint32)) and list(s4.rays) == list(s0.rays)
    # a failed prediction with the side stream on: the second pass repairs it
    soup = rt.SceneSoup.load(scene_path("cornellbox_recursive"))
    gpu_ctx.upload(soup)
    h1 = gpu_ctx.render(0, nx=200, ny=200)[0]
    cd = soup.camera_desc(0, nx=200, ny=200)
    away = type(cd).from_buffer_copy(cd)
    away.gaze[0], away.gaze[1], away.gaze[2] = 0.0, 0.0, 1.0
    gpu_ctx.render(away, shadow_stream=1)
    h5, _, _, s5 = gpu_ctx.render(0, nx=200, ny=200, shadow_stream=1)
    assert s5.frame_passes == 2 and np.array_equal(h5.view(np.uint32), h1.view(np.uint32))
    h6, _, _, s6 = gpu_ctx.render(0, nx=200, ny=200, shadow_stream=1)
    assert s6.frame_passes == 1 and s6.level_syncs == 0 and np.array_equal(h6.view(np.uint32), h1.view(np.uint32))
