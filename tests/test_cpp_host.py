"""The C++ host side (include/rt_mi355.hpp, host/): the reference's main / RenderStrategy / AccelerationStructure shape in the
reference's language over the C-ABI.  CPU: it builds and fails loudly without a GPU.  GPU: the files it writes and the hits
it returns are byte-identical to the Python mirror's (same library underneath, different host language)."""
import importlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "rt_render")
abi = importlib.import_module("ray-tracer_amd").abi


def _run(args, **kw):
    return subprocess.run([BIN] + args, capture_output=True, text=True, timeout=300, **kw)


def test_cpp_host_builds_and_has_no_fallback(rt, scene_path, tmp_path):
    import torch
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    assert _run([]).returncode == 2                                   # usage
    bad = tmp_path / "bad.rtscene"
    bad.write_bytes(b"not a scene")
    out = _run([str(bad)])
    assert out.returncode == 1 and "not a scene file" in out.stderr       # SceneParser's place: bad input is reported, not rendered
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    soup = rt.SceneSoup.load(scene_path("simple"))
    soup.save_raw(str(tmp_path / "s.rtscene"))
    out = _run([str(tmp_path / "s.rtscene"), str(tmp_path)])
    assert out.returncode == 1 and "no CPU fallback" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornellbox_recursive", "scienceTree", "cornellbox_boxes_dynamic"])
def test_cpp_host_equals_python_mirror(rt, scene_path, tmp_path, name):
    soup = rt.SceneSoup.load(scene_path(name))
    for c in soup.cameras:
        c["nx"], c["ny"] = 96, 72
        c["num_samples"] = min(int(c.get("num_samples", 1)), 4)
    soup.save_raw(str(tmp_path / "s.rtscene"))
    rng = np.random.RandomState(4)
    rays = np.zeros(5000, abi.ray_dtype)
    pos = np.asarray(soup.cameras[0]["position"], np.float64)
    v = np.asarray(soup.vert_pos, np.float64)
    tgt = v[rng.randint(len(v), size=len(rays))] + rng.randn(len(rays), 3) * 0.3
    d = tgt - pos
    rays["origin"], rays["direction"] = pos.astype(np.float32), (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["time"] = rng.rand(len(rays)).astype(np.float32)
    rays.tofile(str(tmp_path / "rays.bin"))
    cdir, pdir = tmp_path / "cpp", tmp_path / "py"
    cdir.mkdir(), pdir.mkdir()
    extra = ["--shared-scene"] if name == "scienceTree" else []      # the accelerator adopts the renderer's scene instead of building it again
    out = _run([str(tmp_path / "s.rtscene"), str(cdir), "--rays", str(tmp_path / "rays.bin"), str(tmp_path / "hits.bin")] + extra)
    assert out.returncode == 0, out.stderr
    r = rt.Mi355Renderer(0, out_dir=str(pdir))
    r.RenderSceneIntoPPM(soup)
    for c in soup.cameras:
        assert open(cdir / c["image_name"], "rb").read() == open(pdir / c["image_name"], "rb").read()
    hits = np.fromfile(str(tmp_path / "hits.bin"), abi.hit_dtype)
    want = r.ctx.intersect(rays)
    assert hits.tobytes() == want.tobytes() and (want["shape"] >= 0).sum() > 100
    r.close()


@pytest.mark.gpu
def test_cpp_host_tonemap_exr(rt, scene_path, tmp_path):
    soup = rt.SceneSoup.load(scene_path("cornellbox"))
    soup.cameras[0]["nx"], soup.cameras[0]["ny"] = 64, 48
    soup.tonemap = (0.18, 1.0, 1.0, 2.2)
    soup.save_raw(str(tmp_path / "s.rtscene"))
    cdir, pdir = tmp_path / "cpp", tmp_path / "py"
    cdir.mkdir(), pdir.mkdir()
    assert _run([str(tmp_path / "s.rtscene"), str(cdir)]).returncode == 0
    r = rt.Mi355Renderer(0, out_dir=str(pdir))
    r.RenderSceneIntoPPM(soup)
    name = soup.cameras[0]["image_name"]
    data = open(cdir / name, "rb").read()
    assert data[:4] == bytes([0x76, 0x2f, 0x31, 0x01]) and data == open(pdir / name, "rb").read()
    r.close()


@pytest.mark.gpu
def test_cameras_in_flight_write_the_same_files(rt, scene_path, tmp_path):
    """A scene with several cameras rendered with two frames in flight (two contexts that share one built scene, a host thread each) —
    Python mirror and C++ host — writes byte for byte the files the one-context loop writes."""
    soup = rt.SceneSoup.load(scene_path("cornellbox_recursive"))
    base = dict(soup.cameras[0])
    base["nx"], base["ny"], base["num_samples"] = 120, 90, 1
    soup.cameras = []
    for k in range(5):                                       # five views: the camera slides sideways
        c = dict(base)
        c["position"] = [float(base["position"][0]) + 0.6 * (k - 2), float(base["position"][1]), float(base["position"][2])]
        c["image_name"] = "view_%d.ppm" % k
        soup.cameras.append(c)
    soup.save_raw(str(tmp_path / "s.rtscene"))
    dirs = {}
    for name in ("py1", "py2", "cpp1", "cpp2"):
        dirs[name] = tmp_path / name
        dirs[name].mkdir()
    for name, k in (("py1", 1), ("py2", 2)):
        r = rt.Mi355Renderer(0, out_dir=str(dirs[name]), frames_in_flight=k)
        r.RenderSceneIntoPPM(soup)
        assert len(r.frames) == 5
        r.close()
    for name, k in (("cpp1", 1), ("cpp2", 2)):
        out = _run([str(tmp_path / "s.rtscene"), str(dirs[name]), "--in-flight", str(k)])
        assert out.returncode == 0, out.stderr
    ref = {c["image_name"]: open(dirs["py1"] / c["image_name"], "rb").read() for c in soup.cameras}
    assert len(set(ref.values())) == 5                       # the views differ
    for name in ("py2", "cpp1", "cpp2"):
        for f, data in ref.items():
            assert open(dirs[name] / f, "rb").read() == data, (name, f)
