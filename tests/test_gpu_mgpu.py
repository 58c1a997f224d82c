"""The multi-GPU step on the one GPU this pool offers.

1. librt_mi355_mgpu.so through its C-ABI with a one-rank communicator (both ways to create one): the tile buffer rt_render wrote in
   compact form is de-interleaved into the frame by rt_gather_tiles, bit-equal to the ordinary full-frame render; rt_scene_broadcast
   exports every arena and sends it through ncclBroadcast.  More than one rank needs more than one GPU: unmeasured on hardware.
2. The torch.distributed path of multigpu.py / bench.py on the nccl backend (= RCCL) with WORLD_SIZE=1, on device tensors:
   TileGather(host_staging=False) for three frames through both tile buffers and broadcast_scene(host_staging=False)."""
import importlib
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_gather_with_a_one_rank_communicator(rt):
    import torch
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    synth = importlib.import_module("ray-tracer_amd.synth")
    dev = torch.device("cuda", 0)
    nx, ny, tile = 320, 192, 32
    soup = synth.dragon_standin_scene(60000, nx, ny)
    ctx = rt.Context(0)
    ctx.upload(soup)
    full, _, _, fst = ctx.render(0)
    cd = soup.camera_desc(0)
    g = mg.RcclGather(devices=[0])
    g2 = mg.RcclGather(unique_id=mg.RcclGather.unique_id(), rank=0, world=1, device=0)
    try:
        assert g.world() == 1 and g2.world() == 1
        for gather, t in ((g, tile), (g2, 16), (g, 0), (g, tile)):          # tile 0: the whole frame is one tile; then back to a cached geometry
            opts = ctx.make_opts(seed=0, tile_size=t, tile_rank=0, tile_world=1, output_on_device=True, compact_output=True)
            tiles = torch.zeros((ny * nx, 3), dtype=torch.float32, device=dev)
            frame = torch.full((ny * nx, 3), -1.0, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            st = ctx.render_device(cd, opts, tiles.data_ptr())
            gather.gather_tiles(nx, ny, t, [tiles.data_ptr()], frame.data_ptr())
            got = frame.cpu().numpy().reshape(ny, nx, 3)
            assert np.array_equal(got.view(np.uint32), full.view(np.uint32)), "tile size %d" % t
            assert list(st.rays) == list(fst.rays)
            if t == tile:                                                    # the compact buffer really is in tile order, not row-major
                assert not np.array_equal(tiles.cpu().numpy().reshape(ny, nx, 3).view(np.uint32), full.view(np.uint32))
        # the same with buffers the library owns (what a host without a device allocator uses)
        tp, fp = g.buffers(nx, ny, tile)
        assert len(tp) == 1 and tp[0] and fp
        opts = ctx.make_opts(seed=0, tile_size=tile, tile_rank=0, tile_world=1, output_on_device=True, compact_output=True)
        ctx.render_device(cd, opts, tp[0])
        g.gather_tiles(nx, ny, tile, tp, fp)
        assert np.array_equal(g.copy_frame().view(np.uint32), full.view(np.uint32))
        # errors come back as codes with a message, nothing throws across the boundary
        with pytest.raises(rt.RtError) as e:
            g.gather_tiles(nx, ny, tile, [tiles.data_ptr()], 0)
        assert "frame buffer" in str(e.value)
        with pytest.raises(rt.RtError):
            g.gather_tiles(0, ny, tile, [tiles.data_ptr()], frame.data_ptr())
        # rank 0's scene through export + ncclBroadcast (no other rank to adopt it here); the context still renders the same frame
        g.broadcast_scene([ctx])
        again, _, _, _ = ctx.render(0)
        assert np.array_equal(again.view(np.uint32), full.view(np.uint32))
        empty = rt.Context(0)
        with pytest.raises(rt.RtError) as e:
            g.broadcast_scene([empty])                                       # nothing uploaded on rank 0
        assert e.value.code == rt.abi.RT_ERR_NO_SCENE
        empty.close()
    finally:
        g.close()
        g2.close()
        ctx.close()


def test_c_abi_gather_refuses_bad_communicators(rt):
    """Bad device lists are refused with a code and a message before RCCL is touched."""
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    for devices in ([], [99], [0, 0], [-1]):
        with pytest.raises(rt.RtError) as e:
            mg.RcclGather(devices=devices)
        assert e.value.code == rt.abi.RT_ERR_INVALID, devices
    with pytest.raises(rt.RtError) as e:
        mg.RcclGather(unique_id=mg.RcclGather.unique_id(), rank=1, world=1, device=0)       # rank outside the world
    assert e.value.code == rt.abi.RT_ERR_INVALID
    with pytest.raises(rt.RtError) as e:
        mg.RcclGather(unique_id=mg.RcclGather.unique_id(), rank=0, world=1, device=99)
    assert e.value.code == rt.abi.RT_ERR_INVALID


@pytest.mark.parametrize("name,tile", [("cornellbox_recursive", 32), ("scienceTree", 8)])
def test_cpp_host_renders_through_the_gather_library(rt, scene_path, tmp_path, name, tile):
    """host/rt_render --ranks 1: the C++ host (rt_mi355::Mi355NodeRenderer) drives rt_gather_create (ncclCommInitAll), rt_scene_broadcast,
    one render thread per rank and rt_gather_tiles, and writes the same bytes as its single-context path."""
    exe = os.path.join(ROOT, "host", "rt_render")
    assert os.path.exists(exe), "host/rt_render is not built (make -C host)"
    soup = rt.SceneSoup.load(scene_path(name))
    for c in soup.cameras:
        c["nx"], c["ny"] = 100, 72                     # not a multiple of the tile size: ragged tiles at the right edge
        c["num_samples"] = min(int(c.get("num_samples", 1)), 4)
    soup.save_raw(str(tmp_path / "s.rtscene"))
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(), b.mkdir()
    r1 = subprocess.run([exe, str(tmp_path / "s.rtscene"), str(a)], capture_output=True, text=True, timeout=120)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    r2 = subprocess.run([exe, str(tmp_path / "s.rtscene"), str(b), "--ranks", "1", "--tile", str(tile)], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert "on 1 rank(s)" in r2.stdout
    files = sorted(os.listdir(a))
    assert files and files == sorted(os.listdir(b))
    for f in files:
        assert open(a / f, "rb").read() == open(b / f, "rb").read(), f


NCCL_WORKER = textwrap.dedent('''
    import importlib, os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    print("NCCL_INIT_OK", flush=True)
    rt = importlib.import_module("ray-tracer_amd")
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    synth = importlib.import_module("ray-tracer_amd.synth")
    nx, ny, tile = 320, 192, 32
    soup = synth.dragon_standin_scene(60000, nx, ny)
    ctx = rt.Context(0)
    ctx.upload(soup)
    full, fid, ft, fst = ctx.render(0)
    sizes = mg.broadcast_scene(ctx, dist, rank, 0, dev, torch, host_staging=False, soup=soup)      # export -> device buffers -> dist.broadcast
    assert len(sizes) == rt.abi.RT_SCENE_ARENAS and sum(sizes) > 10_000_000, sizes
    print("BROADCAST_DEVICE_OK", sum(sizes), flush=True)
    cd = soup.camera_desc(0)
    tg = mg.TileGather(nx, ny, tile, rank, world, dev, torch, host_staging=False)
    opts = ctx.make_opts(seed=0, tile_size=tile, tile_rank=rank, tile_world=world, output_on_device=True, compact_output=True)
    seen = set()
    for it in range(3):                                    # three frames through the two alternating tile buffers
        buf = tg.acquire()
        seen.add(buf.data_ptr())
        buf.zero_()
        torch.cuda.synchronize()
        ctx.render_device(cd, opts, buf.data_ptr())
        frame = tg.gather(dist)                            # dist.gather on device tensors (RCCL), then index_copy_ on the device
        torch.cuda.synchronize()
        got = frame.cpu().numpy().reshape(ny, nx, 3)
        assert np.array_equal(got.view(np.uint32), full.view(np.uint32)), "frame %%d differs from the single-context frame" %% it
    assert len(seen) == 2 and frame.is_cuda
    dist.barrier()
    torch.cuda.synchronize()
    print("NCCL_PATH_OK", flush=True)
    ctx.close()
    dist.destroy_process_group()
''')


def test_torch_nccl_path_on_device_tensors_world_size_one(tmp_path):
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", OMP_NUM_THREADS="4",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-u", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "NCCL_PATH_OK" in out.stdout and "BROADCAST_DEVICE_OK" in out.stdout
