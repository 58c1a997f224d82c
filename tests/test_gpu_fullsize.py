"""BASELINE.json's configurations at their FULL sizes, GPU against the CPU oracle on the same inputs and the same counter-based
RNG: tolerance of tests/parity_util.py (PSNR >= 60 dB and |gpu - cpu| <= 1e-3 max(1, |cpu|) on the float frame), identical
primary (shape, t) for every pixel, identical ray counts per kind.  The oracle needs about a second for each of configs 1-4
on the GPU box's host cores; config 5's geometry runs at the largest size it finishes within a minute (the full 50 M-triangle,
256-spp frame is 15.6 G rays: hours of CPU).  Config 2a (the 1M-triangle mesh) is in test_gpu_boundary.py."""
import importlib
import time

import numpy as np
import pytest

from parity_util import compare_frames

pytestmark = pytest.mark.gpu
abi = importlib.import_module("ray-tracer_amd").abi
synth = importlib.import_module("ray-tracer_amd.synth")


def run_full(rt, orc, ctx, soup, label, **cam):
    ctx.upload(soup)
    g_hdr, g_id, g_t, g_st = ctx.render(0, seed=3, timing=1, **cam)
    osc = orc.OracleScene(soup)
    t0 = time.time()
    c_hdr, c_id, c_t, c_st = osc.render(0, seed=3, nthreads=16, dynamic=True, **cam)
    cpu_s = time.time() - t0
    osc.close()
    m = compare_frames(g_hdr, c_hdr, g_id, c_id, g_t, c_t)
    assert list(g_st.rays) == list(c_st.rays), (list(g_st.rays), list(c_st.rays))
    print("%s: %d rays, GPU kernels %.1f ms, oracle %.1f s, %s" % (label, sum(g_st.rays), g_st.ms_kernels, cpu_s, m), flush=True)
    return m


def test_config1_cornellbox_recursive_800x800(rt, orc, gpu_ctx, scene_path):
    run_full(rt, orc, gpu_ctx, rt.SceneSoup.load(scene_path("cornellbox_recursive")), "config 1")


def test_config2b_sciencetree_1440x720(rt, orc, gpu_ctx, scene_path):
    run_full(rt, orc, gpu_ctx, rt.SceneSoup.load(scene_path("scienceTree")), "config 2b")


@pytest.mark.parametrize("name", ["cornellbox_brushed_metal", "spheres_dof"])
def test_config3_100spp_800x800(rt, orc, gpu_ctx, scene_path, name):
    soup = rt.SceneSoup.load(scene_path(name))
    assert (soup.cameras[0]["nx"], soup.cameras[0]["ny"]) == (800, 800)
    run_full(rt, orc, gpu_ctx, soup, "config 3 " + name, num_samples=100)


def config4_scene(rt, scene_path, with_dielectric):
    soup = rt.SceneSoup.load(scene_path("cornellbox_boxes_dynamic"))
    if not with_dielectric:
        return soup
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
    return soup


@pytest.mark.parametrize("with_dielectric", [False, True])
def test_config4_motion_blur_64spp_750x600(rt, orc, gpu_ctx, scene_path, with_dielectric):
    """cornellbox_boxes_dynamic as shipped (motion-blurred instance, mirror + Phong, depth 4), and with the dielectric sphere of
    config 1 added so that refraction queues diverge under motion blur (the configuration text asks for both)."""
    soup = config4_scene(rt, scene_path, with_dielectric)
    assert (soup.cameras[0]["nx"], soup.cameras[0]["ny"]) == (750, 600)
    run_full(rt, orc, gpu_ctx, soup, "config 4" + (" + dielectric" if with_dielectric else ""), num_samples=64)


def test_config5_geometry_largest_the_oracle_finishes(rt, orc, gpu_ctx):
    """Config 5's scene family — rotated MeshInstances of one mesh, half with Perlin bump, image-textured floor, 2 lights, depth 6 —
    at 20 instances x 200 000 triangles (4 M triangles) and 1280x720 at 4 spp."""
    soup = synth.instanced_scene(20, 200_000, 1280, 720, 4, perlin_table=rt.renderer.perlin_tables(1)[0])
    run_full(rt, orc, gpu_ctx, soup, "config 5 geometry (20 x 200k)")
