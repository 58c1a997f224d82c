"""Synthetic inputs for the configurations whose assets are missing from the reference checkout.

`dragon_standin_scene` concretises BASELINE.json config 2 ("otherdragon ... ~1M tris"): the reference's
src/scenes/other_dragon.xml needs scenes/ply/dragon_remeshed.ply, which is listed in .MISSING_LARGE_BLOBS.  Camera,
lights, materials, recursion depth and the mirror floor are the ones of other_dragon.xml (:1-69); the mesh is a seeded
procedural closed surface (a displaced (2,3) torus-knot tube) with exactly `n_tris` triangles whose bounding box is
about 300 units and is centred near (0, 100, -20), the stand-in the benchmark configuration uses for the missing dragon mesh.
"""
import numpy as np

from . import abi
from .scene import SceneBuilder, mat_identity, mat_mul, mat_rotate, mat_translate

F32 = np.float32


def torus_knot_mesh(n_tris=1_000_000, seed=7, extent=300.0, center=(0.0, 100.0, -20.0)):
    """Closed tube around a (2,3) torus knot, `n_tris` triangles (n_tris = 2*U*V, V the divisor nearest sqrt(n_tris/4))."""
    v_cnt = int(round((n_tris / 4) ** 0.5))
    while v_cnt > 3 and n_tris % (2 * v_cnt):
        v_cnt -= 1
    u_cnt = n_tris // (2 * v_cnt)
    assert 2 * u_cnt * v_cnt == n_tris and v_cnt >= 3, "n_tris must factor as 2*U*V"
    rng = np.random.RandomState(seed)
    u = np.linspace(0.0, 2 * np.pi, u_cnt, endpoint=False)
    v = np.linspace(0.0, 2 * np.pi, v_cnt, endpoint=False)
    p, q = 2.0, 3.0
    r = 2.0 + np.cos(q * u)
    c = np.stack([r * np.cos(p * u), r * np.sin(p * u), 0.9 * np.sin(q * u)], 1)
    # frame by finite differences of the closed curve
    t = np.roll(c, -1, 0) - np.roll(c, 1, 0)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    n1 = np.cross(t, up)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(t, n1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    # smooth multi-octave displacement (seeded phases)
    disp = np.zeros_like(uu)
    for k in range(1, 7):
        a, b = rng.randint(1, 24), rng.randint(1, 12)
        disp += (0.5 ** k) * np.sin(a * uu + rng.uniform(0, 2 * np.pi)) * np.cos(b * vv + rng.uniform(0, 2 * np.pi))
    rad = 0.55 * (1.0 + 0.45 * disp)
    pos = c[:, None, :] + rad[..., None] * (np.cos(vv)[..., None] * n1[:, None, :] + np.sin(vv)[..., None] * n2[:, None, :])
    pos = pos.reshape(-1, 3)
    lo, hi = pos.min(0), pos.max(0)
    pos = (pos - (lo + hi) / 2) * (extent / (hi - lo).max()) + np.asarray(center)
    iu = np.arange(u_cnt)[:, None]
    iv = np.arange(v_cnt)[None, :]
    a = (iu * v_cnt + iv).reshape(-1)
    b = (((iu + 1) % u_cnt) * v_cnt + iv).reshape(-1)
    d = (iu * v_cnt + (iv + 1) % v_cnt).reshape(-1)
    e = (((iu + 1) % u_cnt) * v_cnt + (iv + 1) % v_cnt).reshape(-1)
    faces = np.concatenate([np.stack([a, b, e], 1), np.stack([a, e, d], 1)], 1).reshape(-1, 3)
    return pos.astype(F32), faces.astype(np.int64)


def dragon_camera(nx=800, ny=480, spp=1):
    """The camera of dragon_standin_scene at another resolution / sample count, as a camera dict (SceneSoup.camera_desc(0, **it))."""
    b = SceneBuilder()
    b.add_lookat_camera((500, -100, 100), (0, 100, -60), (0, 0, 1), 32.0, 1.0, nx, ny, num_samples=spp, image_name="other_dragon.ppm")
    return dict(b.soup.cameras[0])


def dragon_standin_scene(n_tris=1_000_000, nx=800, ny=480, seed=7):
    b = SceneBuilder()
    s = b.soup
    s.max_depth = 6
    s.background = np.zeros(3, F32)
    s.ambient = np.array([25, 25, 25], F32)
    b.add_lookat_camera((500, -100, 100), (0, 100, -60), (0, 0, 1), 32.0, 1.0, nx, ny, image_name="other_dragon.ppm")
    b.add_light(abi.RT_LIGHT_POINT, (400, 100, 200), (5e7, 5e7, 5e7))
    b.add_light(abi.RT_LIGHT_POINT, (600, -200, 200), (5e7, 5e7, 5e7))
    m_dragon = b.add_material(abi.RT_MAT_CONDUCTOR, (0, 0, 0), (0, 0, 0), (1, 0.86, 0.57), (1, 0.86, 0.57), (0, 0, 0), 0.370, 2.820, 5)
    m_floor = b.add_material(abi.RT_MAT_MIRROR, (0, 0, 0), (0.5, 0.5, 0.5), (0, 0, 0), (0.5, 0.5, 0.5), phong=1)
    pos, faces = torus_knot_mesh(n_tris, seed)
    b.add_mesh(pos, faces, material=m_dragon)                       # <Mesh id="1"> (flat shading)
    floor = np.array([[-1000, -1000, -100], [1000, -1000, -100], [1000, 1000, -100], [-1000, 1000, -100]], F32)
    b.add_mesh(floor, [[0, 1, 2], [2, 3, 0]], material=m_floor)    # <Mesh id="2">
    return b.finish("dragon_standin_%d" % n_tris)


def instanced_scene(n_instances=50, n_tris=1_000_000, nx=3840, ny=2160, spp=256, seed=7, perlin_table=None, checker=None, compact=False):
    """BASELINE.json config 5: a grid of MeshInstances of the stand-in mesh with distinct rotations,
    half of them with a Perlin bump_normal texture, on a floor with a replace_kd image texture.  compact: the instances and
    the primitive list go to the library as rt_instance / rt_prim_run records (52 M triangles are then never written out on
    the host)."""
    b = SceneBuilder(compact=compact)
    s = b.soup
    s.max_depth = 6
    s.ambient = np.array([25, 25, 25], F32)
    cols = 10
    rows = (n_instances + cols - 1) // cols
    span = 380.0
    cx, cy = (cols - 1) * span / 2, (rows - 1) * span / 2
    b.add_lookat_camera((cx + 2600, cy - 2200, 1500), (cx, cy, -40), (0, 0, 1), 40.0, 1.0, nx, ny, num_samples=spp, image_name="instanced.ppm")
    b.add_light(abi.RT_LIGHT_POINT, (cx + 1500, cy - 800, 1800), (6e8, 6e8, 6e8))
    b.add_light(abi.RT_LIGHT_POINT, (cx - 1200, cy + 900, 1600), (4e8, 4e8, 4e8))
    m_metal = b.add_material(abi.RT_MAT_CONDUCTOR, (0, 0, 0), (0, 0, 0), (1, 0.86, 0.57), (1, 0.86, 0.57), (0, 0, 0), 0.370, 2.820, 5)
    m_diff = b.add_material(abi.RT_MAT_DEFAULT, (1, 1, 1), (0.5, 0.4, 0.3), (0.4, 0.4, 0.4), phong=20)
    m_floor = b.add_material(abi.RT_MAT_DEFAULT, (1, 1, 1), (0.5, 0.5, 0.5), (0, 0, 0), phong=1)
    t_bump = b.add_texture(abi.RT_TEX_PERLIN, abi.RT_DECAL_BUMP_NORMAL, bump_factor=0.08, noise_scale=10.0,
                           noise_conversion=abi.RT_NOISE_ABSVAL, perm=perlin_table)
    if checker is None:
        yy, xx = np.mgrid[0:256, 0:256]
        checker = np.where((((xx // 32) + (yy // 32)) % 2)[..., None] == 0, np.uint8(230), np.uint8(40)).repeat(3, axis=2).astype(np.uint8)
    t_floor = b.add_texture(abi.RT_TEX_IMAGE, abi.RT_DECAL_REPLACE_KD, normalizer=255, image=checker)
    pos, faces = torus_knot_mesh(n_tris, seed, center=(0.0, 0.0, 60.0))
    base_plain = b.add_mesh(pos, faces, material=m_metal, xform=mat_translate((-10000.0, -10000.0, 60.0)))
    base_bump = b.add_mesh(pos, faces, material=m_diff, xform=mat_translate((-10000.0, -11000.0, 60.0)), tex_normal=t_bump)
    for i in range(n_instances):
        gx, gy = i % cols, i // cols
        m = mat_mul(mat_translate((gx * span, gy * span, 0.0)), mat_rotate(17.0 * i, (0, 0, 1)))
        b.add_instance(base_bump if i % 2 else base_plain, m, reset_transform=True)
    ext = max(cx, cy) + 3000.0
    fpos = np.array([[cx - ext, cy - ext, -100], [cx + ext, cy - ext, -100], [cx + ext, cy + ext, -100], [cx - ext, cy + ext, -100]], F32)
    fuv = np.array([[0, 0], [40, 0], [40, 40], [0, 40]], F32)
    b.add_mesh(fpos, [[0, 1, 2], [2, 3, 0]], uv=fuv, material=m_floor, tex_color=t_floor)
    return b.finish("instanced_%dx%d" % (n_instances, n_tris))
