"""librt_mi355_mgpu.so (the multi-GPU step as a C-ABI, include/rt_mi355_mgpu.h) without a GPU: it loads, exports every symbol its
header declares, fails loudly without a device, and its host-side de-interleave puts every rank's compact tile pixels where
rt_tile_pixel_order says — the join that replaces the reference's thread join (src/DefaultRenderer.cpp:73-75)."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mg = importlib.import_module("ray-tracer_amd.multigpu")


def test_header_symbols_exported(rt):
    text = open(os.path.join(ROOT, "include", "rt_mi355_mgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))
    assert len(names) == 11, names
    L = mg.mgpu_library()
    for n in names:
        assert hasattr(L, n), "librt_mi355_mgpu.so does not export %s" % n
    assert mg.RT_GATHER_ID_BYTES == int(re.search(r"#define RT_GATHER_ID_BYTES (\d+)", open(os.path.join(ROOT, "include", "rt_mi355_mgpu.h")).read()).group(1))


@pytest.mark.parametrize("nx,ny,tile,world", [(70, 45, 16, 3), (64, 64, 32, 8), (33, 7, 8, 2), (40, 30, 0, 1), (17, 19, 4, 5)])
def test_deinterleave_host_matches_the_tile_order(rt, nx, ny, tile, world):
    rng = np.random.default_rng(nx * 1000 + ny)
    frame = rng.random((ny * nx, 3), dtype=np.float32)
    tiles, seen = [], np.zeros(ny * nx, np.int32)
    for r in range(world):
        order = rt.renderer.tile_pixel_order(nx, ny, tile, r, world)
        seen[order] += 1
        tiles.append(frame[order])                      # what rank r's compact render output would hold
    assert (seen == 1).all()
    got = mg.deinterleave_host(nx, ny, tile, tiles)
    assert np.array_equal(got.reshape(-1, 3).view(np.uint32), frame.view(np.uint32))


def test_deinterleave_host_rejects_bad_arguments(rt):
    L = mg.mgpu_library()
    out = np.zeros((4, 4, 3), np.float32)
    ptrs = (C.c_void_p * 2)(None, None)
    assert L.rt_gather_deinterleave_host(4, 4, 2, 2, ptrs, out.ctypes.data) == rt.abi.RT_ERR_INVALID      # NULL tile buffer of a rank that owns pixels
    assert L.rt_gather_deinterleave_host(0, 4, 2, 2, ptrs, out.ctypes.data) == rt.abi.RT_ERR_INVALID
    assert L.rt_gather_deinterleave_host(4, 4, 2, 1, None, out.ctypes.data) == rt.abi.RT_ERR_INVALID


def test_no_device_is_a_loud_error(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError) as e:
        mg.RcclGather(devices=[0])
    assert e.value.code == rt.abi.RT_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    with pytest.raises(rt.RtError) as e:
        mg.RcclGather(unique_id=bytes(128), rank=0, world=1, device=0)
    assert e.value.code == rt.abi.RT_ERR_NO_DEVICE
