"""GPU parity: librt_mi355.so (through the C-ABI) against the CPU oracle on the same inputs."""
import numpy as np
import pytest

from parity_util import compare_frames

pytestmark = pytest.mark.gpu

ONE_SPP = ["cornellbox", "cornellbox_recursive", "simple", "spheres", "bunny", "scienceTree", "cube_point", "cube_directional"]


@pytest.mark.parametrize("name", ONE_SPP)
def test_one_spp_frame(rt, orc, gpu_ctx, scene_path, name):
    soup = rt.SceneSoup.load(scene_path(name))
    gpu_ctx.upload(soup)
    g_hdr, g_id, g_t, g_st = gpu_ctx.render(0, timing=2)
    osc = orc.OracleScene(soup)
    c_hdr, c_id, c_t, c_st = osc.render(0, nthreads=16, dynamic=True)
    m = compare_frames(g_hdr, c_hdr, g_id, c_id, g_t, c_t)
    assert list(g_st.rays) == list(c_st.rays), (list(g_st.rays), list(c_st.rays))
    print(name, m, g_st.as_dict())
