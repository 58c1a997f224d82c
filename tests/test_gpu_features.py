"""GPU parity on the stochastic / textured parts of the path (same counter-based RNG on both sides => per-pixel
comparison, not merely statistical)."""
import importlib

import numpy as np
import pytest

import synthetic_scenes as syn
from parity_util import compare_frames

pytestmark = pytest.mark.gpu
abi = importlib.import_module("ray-tracer_amd").abi
synth = importlib.import_module("ray-tracer_amd.synth")


def run(rt, orc, ctx, soup, seed=3, **cam):
    ctx.upload(soup)
    g_hdr, g_id, g_t, g_st = ctx.render(0, seed=seed, timing=1, **cam)
    osc = orc.OracleScene(soup)
    c_hdr, c_id, c_t, c_st = osc.render(0, seed=seed, nthreads=16, dynamic=True, **cam)
    osc.close()
    m = compare_frames(g_hdr, c_hdr, g_id, c_id, g_t, c_t)
    assert list(g_st.rays) == list(c_st.rays), (list(g_st.rays), list(c_st.rays))
    return m


# BASELINE configs 3 and 4 (reduced spp / resolution so the oracle finishes in seconds) + the other sampled scenes
@pytest.mark.parametrize("name,spp,res", [
    ("cornellbox_brushed_metal", 16, 160),     # glossy mirrors (Q15)
    ("spheres_dof", 16, 160),                  # thin lens (Q25)
    ("cornellbox_boxes_dynamic", 16, 150),     # motion blur, mesh instances (Q27, Q29)
    ("cornellbox_area", 9, 160),               # area light (Q16)
    ("metal_glass_plates", 4, 160),            # dielectric + conductor plates
    ("cornellbox_brushed_metal", 10, 96),      # non-square sample count (Q19)
])
def test_sampled_scenes(rt, orc, gpu_ctx, scene_path, name, spp, res):
    soup = rt.SceneSoup.load(scene_path(name))
    c = soup.cameras[0]
    ny = max(8, int(res * c["ny"] / c["nx"]))
    print(name, run(rt, orc, gpu_ctx, soup, num_samples=spp, nx=res, ny=ny))


@pytest.mark.parametrize("name", ["sphere_perlin", "sphere_perlin_bump", "sphere_perlin_scale", "cube_perlin", "cube_perlin_bump"])
def test_perlin_scenes(rt, orc, gpu_ctx, scene_path, name):
    soup = rt.SceneSoup.load(scene_path(name))
    print(name, run(rt, orc, gpu_ctx, soup, nx=200, ny=200))


# The reference's own textured scenes (image textures from its textures/ directory, converted by tools/convert_scenes.py):
# replace_kd / blend_kd / bump_normal / replace_normal decals on meshes and spheres, nearest and bilinear lookup, a background
# texture, composed transformations, a 50k-triangle ASCII PLY mesh.
@pytest.mark.parametrize("name,spp,res", [
    ("cube_wall", 1, 200), ("cube_wall_normal", 1, 200), ("cube_waves", 1, 200), ("cube_cushion", 1, 200),
    ("sphere_nearest_bilinear", 1, 200), ("sphere_nobump_bump", 1, 200), ("sphere_normal", 4, 160), ("ellipsoids_texture", 1, 200),
    ("galactica_static", 1, 320), ("simple_transform", 4, 160), ("cube_point_hdr", 1, 200), ("tap_0200", 4, 200),
])
def test_reference_textured_scenes(rt, orc, gpu_ctx, scene_path, name, spp, res):
    soup = rt.SceneSoup.load(scene_path(name))
    c = soup.cameras[0]
    ny = max(8, int(res * c["ny"] / c["nx"]))
    print(name, run(rt, orc, gpu_ctx, soup, num_samples=spp, nx=res, ny=ny))


@pytest.mark.parametrize("tri,sph,tk,sk,bg", [
    (abi.RT_DECAL_REPLACE_KD, abi.RT_DECAL_BLEND_KD, "image", "image", False),
    (abi.RT_DECAL_BUMP_NORMAL, abi.RT_DECAL_BUMP_NORMAL, "image", "image", True),
    (abi.RT_DECAL_REPLACE_NORMAL, abi.RT_DECAL_REPLACE_NORMAL, "image", "image", False),
    (abi.RT_DECAL_REPLACE_ALL, abi.RT_DECAL_REPLACE_KD, "image", "perlin", False),
    (abi.RT_DECAL_BLEND_KD, abi.RT_DECAL_REPLACE_ALL, "perlin", "image", True),
    (abi.RT_DECAL_BUMP_NORMAL, abi.RT_DECAL_BUMP_NORMAL, "perlin", "perlin", False),
])
def test_texture_decals(rt, orc, gpu_ctx, tri, sph, tk, sk, bg):
    soup = syn.textured_scene(tri, sph, tk, sk, background_texture=bg)
    print((tri, sph, tk, sk, bg), run(rt, orc, gpu_ctx, soup))


@pytest.mark.parametrize("kind,flag", [(0, False), (1, False), (1, True), (2, False), (3, False), (3, True), (4, False), (4, True)])
def test_lights_and_brdfs(rt, orc, gpu_ctx, kind, flag):
    soup = syn.lights_brdf_scene(kind, flag)
    print((kind, flag), run(rt, orc, gpu_ctx, soup))


@pytest.mark.parametrize("name", ["brdf_blinnphong_original", "brdf_blinnphong_modified", "brdf_blinnphong_modified_normalized",
                                  "brdf_phong_original", "brdf_phong_modified", "brdf_phong_modified_normalized", "brdf_torrancesparrow"])
def test_brdf_scenes(rt, orc, gpu_ctx, scene_path, name):
    soup = rt.SceneSoup.load(scene_path(name))
    print(name, run(rt, orc, gpu_ctx, soup, nx=200, ny=200))


def test_config5_reduced(rt, orc, gpu_ctx):
    """BASELINE config 5 at a size the oracle finishes in seconds: rotated MeshInstances of one mesh (per-instance transforms,
    Q29 blur boxes), half of them with a Perlin bump_normal texture, a replace_kd image floor, two lights, depth 6, 4 spp."""
    soup = synth.instanced_scene(8, 4000, 160, 90, 4, perlin_table=rt.renderer.perlin_tables(1)[0])
    m = run(rt, orc, gpu_ctx, soup)
    print("config5_reduced", m)


def test_more_objects_than_the_traversal_caches(rt, orc, gpu_ctx):
    """The traversal keeps the first 24 object records in LDS (rt_kernels.hip, OBJ_CACHE); 30 rotated instances + 2 base meshes +
    the floor = 33 objects, so a third of the hits enter objects / return hit points through the object table in memory."""
    soup = synth.instanced_scene(30, 1500, 160, 90, 2, perlin_table=rt.renderer.perlin_tables(1)[0])
    assert len(soup.objects) > 24
    m = run(rt, orc, gpu_ctx, soup)
    print("33 objects", m)


def test_config4_with_dielectric(rt, orc, gpu_ctx, scene_path):
    """BASELINE config 4 with its dielectric: cornellbox_boxes_dynamic (motion-blurred instance, mirror + Phong) PLUS the
    dielectric sphere of config 1, so that refraction queues diverge under motion blur.  Reduced spp / resolution."""
    soup = rt.SceneSoup.load(scene_path("cornellbox_boxes_dynamic"))
    soup._contig()
    glass = np.zeros((), abi.material_dtype)
    glass["type"] = abi.RT_MAT_DIELECTRIC
    glass["ka"], glass["kd"], glass["ks"] = (0, 0, 0), (0, 0, 0), (0, 0, 0)
    glass["ac"] = (0.01, 0.01, 0.01)
    glass["refraction_index"], glass["brdf_exponent"] = 1.55, 1
    soup.materials = np.append(soup.materials, glass)
    ob = np.zeros((), abi.object_dtype)
    ob["xform"] = np.eye(4, dtype=np.float32).reshape(16)
    ob["flags"], ob["material"], ob["tex_color"], ob["tex_normal"] = abi.RT_OBJ_HAS_XFORM, len(soup.materials) - 1, -1, -1
    soup.objects = np.append(soup.objects, ob)
    lo, hi = soup.vert_pos.min(0), soup.vert_pos.max(0)
    c = (lo + hi) / 2
    rad = float((hi - lo).min()) * 0.12
    soup.sph_center_radius = np.vstack([soup.sph_center_radius.reshape(-1, 4), [c[0] + rad, lo[1] + 1.5 * rad, c[2] + 2 * rad, rad]]).astype(np.float32)
    soup.sph_object = np.append(soup.sph_object, len(soup.objects) - 1).astype(np.int32)
    soup.prim_shape = np.append(soup.prim_shape, (len(soup.sph_object) - 1) | abi.RT_PRIM_SPHERE).astype(np.int32)
    cam = soup.cameras[0]
    ny = max(8, int(150 * cam["ny"] / cam["nx"]))
    m = run(rt, orc, gpu_ctx, soup, num_samples=16, nx=150, ny=ny)
    print("config4_with_dielectric", m)
