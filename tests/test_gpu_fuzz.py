"""Seeded random scenes against the oracle: triangle soups with slivers and duplicates, spheres under non-uniform transforms,
instances with motion blur, every material type and light type, both scene-size regimes (host-built and device-built BVH).
Hit records must equal the oracle's bit for bit; frames within the tolerance of tests/parity_util.py."""
import importlib

import numpy as np
import pytest

from parity_util import compare_frames

pytestmark = pytest.mark.gpu
rtp = importlib.import_module("ray-tracer_amd")
abi = rtp.abi
scene_mod = importlib.import_module("ray-tracer_amd.scene")


def random_scene(seed, n_tris, depth, fancy=False):
    """fancy: textures (Perlin bump, image replace_kd / bump), an area light, glossy mirror, depth of field -- the stochastic
    path (both sides draw from the same Philox stream)."""
    rng = np.random.RandomState(seed)
    b = rtp.SceneBuilder()
    s = b.soup
    s.max_depth = depth
    s.background = rng.rand(3).astype(np.float32) * 30
    s.ambient = (rng.rand(3) * 40).astype(np.float32)
    if fancy: b.add_camera((0, 0, 30), (0, 0, -1), (0, 1, 0), (-10, 10, -10, 10), 10, 64, 64, focus_distance=30.0, aperture_size=0.4)
    else: b.add_camera((0, 0, 30), (0, 0, -1), (0, 1, 0), (-10, 10, -10, 10), 10, 64, 64)
    tex_c = tex_n = tex_b = -1
    if fancy:
        b.add_light(abi.RT_LIGHT_AREA, (0, 14, 6), (3e4,) * 3, (0, -1, 0), size=3.0)
        img = (rng.rand(16, 32, 3) * 255).astype(np.uint8)
        tex_c = b.add_texture(abi.RT_TEX_IMAGE, abi.RT_DECAL_REPLACE_KD, normalizer=255, image=img)
        tex_b = b.add_texture(abi.RT_TEX_IMAGE, abi.RT_DECAL_BUMP_NORMAL, bump_factor=2.0, normalizer=255, image=img)
        tex_n = b.add_texture(abi.RT_TEX_PERLIN, abi.RT_DECAL_BUMP_NORMAL, bump_factor=0.3, noise_scale=2.0, noise_conversion=abi.RT_NOISE_ABSVAL,
                              perm=rng.permutation(16))
    b.add_light(abi.RT_LIGHT_POINT, tuple(rng.randn(3) * 5 + (0, 15, 10)), (6e4,) * 3)
    b.add_light(abi.RT_LIGHT_DIRECTIONAL, intensity=(40, 35, 30), direction=tuple(rng.randn(3) - (0, 2, 0)))
    b.add_light(abi.RT_LIGHT_SPOT, (8, 12, 12), (9e4,) * 3, (-0.5, -0.8, -0.6), coverage=50.0, falloff=30.0)
    mats = [b.add_material(abi.RT_MAT_DEFAULT, (1, 1, 1), tuple(rng.rand(3)), tuple(rng.rand(3)), phong=int(rng.randint(1, 60))),
            b.add_material(abi.RT_MAT_MIRROR, (0, 0, 0), tuple(rng.rand(3) * 0.5), (0, 0, 0), tuple(0.3 + rng.rand(3) * 0.6), phong=1, roughness=0.2 if fancy else 0.0),
            b.add_material(abi.RT_MAT_DIELECTRIC, (0, 0, 0), (0.1, 0.1, 0.1), (0.2, 0.2, 0.2), ac=tuple(rng.rand(3) * 0.05), refraction_index=1.3 + rng.rand() * 0.4, phong=5),
            b.add_material(abi.RT_MAT_CONDUCTOR, (0, 0, 0), (0.1, 0.1, 0.1), (0.3, 0.3, 0.3), tuple(0.5 + rng.rand(3) * 0.5), refraction_index=0.4, absorption_index=2.8, phong=8),
            b.add_material(abi.RT_MAT_DEFAULT, (1, 1, 1), tuple(rng.rand(3)), tuple(rng.rand(3)), phong=20, brdf_kind=abi.RT_BRDF_TORRANCE_SPARROW, brdf_flag=True,
                           refraction_index=1.8, brdf_exponent=30.0)]
    # triangle soup: random, sliver, axis-aligned and duplicated triangles in a few meshes with random transforms
    n_mesh = 4
    handles = []
    for m in range(n_mesh):
        k = n_tris // n_mesh
        c = rng.randn(k, 1, 3) * 6
        tri = c + rng.randn(k, 3, 3) * rng.choice([0.05, 0.5, 2.0], size=(k, 1, 1))
        tri[: k // 10, 2] = tri[: k // 10, 1] + (tri[: k // 10, 1] - tri[: k // 10, 0]) * 1e-3       # slivers
        tri[k // 10: k // 5, :, rng.randint(3)] = np.round(tri[k // 10: k // 5, :1, 0])                # axis-aligned (flat boxes)
        tri[-3:] = tri[:3]                                                                             # exact duplicates
        pos = tri.reshape(-1, 3).astype(np.float32)
        faces = np.arange(3 * k).reshape(k, 3)
        x = scene_mod.mat_mul(scene_mod.mat_translate(tuple(rng.randn(3) * 2)), scene_mod.mat_rotate(float(rng.rand() * 360), tuple(rng.randn(3))))
        if m == 1:
            x = scene_mod.mat_mul(x, scene_mod.mat_scale((1.5, 0.6, 1.0)))
        handles.append(b.add_mesh(pos, faces, uv=rng.rand(3 * k, 2).astype(np.float32), xform=x, material=mats[m % len(mats)], smooth=(m == 2),
                                  motion_blur=((0, 0, 0) if m != 3 else (0.5, 0, 0)), tex_color=(tex_c if m == 0 else -1),
                                  tex_normal=(tex_n if m == 0 else (tex_b if m == 2 else -1))))
    b.add_instance(handles[0], scene_mod.mat_translate((3.0, -2.0, 1.0)), material=mats[1], motion_blur=(0, 1.0, 0))
    for i in range(5):
        x = scene_mod.mat_mul(scene_mod.mat_translate(tuple(rng.randn(3) * 5)), scene_mod.mat_rotate(float(rng.rand() * 180), (1, 1, 0)))
        b.add_sphere(tuple(rng.randn(3)), float(0.5 + rng.rand() * 2), xform=x, material=mats[(i + 2) % len(mats)],
                     tex_color=(tex_c if i == 0 else -1), tex_normal=(tex_b if i == 1 else (tex_n if i == 2 else -1)))
    return b.finish("fuzz%d" % seed)


def rays_for(soup, n, seed):
    rng = np.random.RandomState(seed)
    v = np.asarray(soup.vert_pos, np.float64)
    r = np.zeros(n, abi.ray_dtype)
    o = rng.randn(n, 3) * 12
    tgt = v[rng.randint(len(v), size=n)] * 1.0 + rng.randn(n, 3) * rng.choice([0.0, 0.01, 1.0], size=(n, 1))
    d = tgt - o
    d[: n // 20, rng.randint(3)] = 0.0                                   # zero direction components
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20)
    r["origin"], r["direction"], r["time"] = o.astype(np.float32), d.astype(np.float32), rng.rand(n).astype(np.float32)
    return r


@pytest.mark.parametrize("seed,n_tris,depth,fancy", [(1, 400, 3, False), (2, 1200, 2, False), (3, 6000, 2, False), (4, 9000, 4, False),
                                                     (5, 800, 2, True), (6, 7000, 3, True)])
def test_random_scene_matches_oracle(orc, gpu_ctx, seed, n_tris, depth, fancy):
    soup = random_scene(seed, n_tris, depth, fancy)
    gpu_ctx.upload(soup)
    osc = orc.OracleScene(soup)
    rays = rays_for(soup, 20000, seed)
    g, c = gpu_ctx.intersect(rays), osc.intersect(rays)
    assert np.array_equal(g["shape"], c["shape"])
    assert np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))
    hit = c["shape"] >= 0
    assert hit.sum() > 2000
    assert np.array_equal(g["ip"][hit].view(np.uint32), c["ip"][hit].view(np.uint32))
    tri = hit & ((c["shape"] & abi.RT_PRIM_SPHERE) == 0)
    # normals of degenerate (sliver) triangles can be NaN on both sides: compare bit patterns
    assert np.array_equal(g["n"][tri].view(np.uint32), c["n"][tri].view(np.uint32))
    ghdr, gid, gt, gst = gpu_ctx.render(0, num_samples=4)
    chdr, cid, ct, cst = osc.render(0, nthreads=8, num_samples=4)
    compare_frames(ghdr, chdr, gid, cid, gt, ct)
    assert list(gst.rays) == list(cst.rays)


@pytest.mark.parametrize("host_paths", [False, True])
def test_noise_hit_of_a_sliver_is_not_pruned(orc, gpu_ctx, host_paths, monkeypatch):
    """Found by tools/fuzz_sweep.py (seed 12081): a 5 x 0.005 sliver whose Cramer determinants are rounding noise.  The reference
    accepts a "hit" on it 6.4 units outside the triangle's box and closer than the ray's real hit; a traversal that prunes boxes by
    distance never tests it.  The normal cone of a subtree that holds such a triangle never allows pruning (rt_cone.h: its
    conditioning factor s is below the threshold for every ray) — on both builder paths."""
    seed = 12081
    if host_paths:
        for k in ("RT_HOST_BUILD", "RT_HOST_FLATTEN", "RT_HOST_INGEST"):
            monkeypatch.setenv(k, "1")
    soup = random_scene(seed, 8000, 2, False)
    gpu_ctx.upload(soup)
    rays = rays_for(soup, 8000, seed)
    g, c = gpu_ctx.intersect(rays), orc.OracleScene(soup).intersect(rays)
    assert c["shape"][727] == 4023                      # the sliver
    assert np.array_equal(g["shape"], c["shape"]) and np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))
