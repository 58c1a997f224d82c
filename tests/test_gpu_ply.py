"""rt_ply_decode / rt_ply_decode_to (PLY body -> positions + triangles on the device) against the ORACLE's reader (oracle/rt_oracle.cpp
orc_ply_read, restating happly + SceneParser.cpp:983-1023) on the same files: bit-exact positions, identical triangle lists, for
binary little-endian and ascii bodies.  The oracle's reader itself is pinned in tests/test_ply_reader.py."""
import importlib
import os
import time

import numpy as np
import pytest

import ply_util

pytestmark = pytest.mark.gpu
scene = importlib.import_module("ray-tracer_amd.scene")


@pytest.mark.parametrize("name", sorted(ply_util.CASES))
def test_decoder_equals_oracle_reader(rt, orc, gpu_ctx, tmp_path, name):
    kw = dict(ply_util.CASES[name])
    pos, faces = ply_util.random_mesh(9, 3000, 20000, kw.pop("sizes"))
    path = os.path.join(tmp_path, name + ".ply")
    ply_util.write_binary_ply(path, pos, faces, **kw)
    used = []

    def decoder(vb, fb, layout):
        used.append(1)
        return gpu_ctx.decode_ply(vb, fb, layout)

    g_pos, g_tris = scene._read_ply(path, decoder)
    c_pos, c_tris = orc.ply_read(open(path, "rb").read())
    assert used, "the file did not take the device path"
    assert g_pos.dtype == np.float32 and np.array_equal(g_pos.view(np.uint32), c_pos.view(np.uint32))
    assert np.array_equal(g_tris, c_tris)


@pytest.mark.parametrize("name", sorted(ply_util.ASCII_CASES))
def test_ascii_decoder_equals_oracle_reader(rt, orc, gpu_ctx, tmp_path, name):
    path = os.path.join(tmp_path, name + ".ply")
    ply_util.make_ascii_case(path, name, nv=3000, nf=20000)
    used = []

    def decoder(vb, fb, layout):
        used.append(layout.format)
        return gpu_ctx.decode_ply(vb, fb, layout)

    g_pos, g_tris = scene._read_ply(path, decoder)
    c_pos, c_tris = orc.ply_read(open(path, "rb").read())
    assert used == [rt.abi.RT_PLY_ASCII], "the file did not take the device path"
    assert g_pos.dtype == np.float32 and np.array_equal(g_pos.view(np.uint32), c_pos.view(np.uint32))        # -0, subnormals, halfway cases
    assert np.array_equal(g_tris, c_tris)


def test_ascii_every_tricky_spelling(rt, orc, gpu_ctx, tmp_path):
    """Each spelling in every coordinate position and both property types; the device converts what it can prove and the host the rest."""
    toks = ply_util.TRICKY_NUMBERS
    rows = [[toks[i], toks[(i + 7) % len(toks)], toks[(i + 13) % len(toks)]] for i in range(len(toks))]
    for pos_type in ("f4", "f8"):
        path = os.path.join(tmp_path, "t_%s.ply" % pos_type)
        ply_util.write_ascii_ply(path, rows, [[0, 1, 2], [2, 1, 0, 3]], pos_type=pos_type)
        g_pos, g_tris = scene._read_ply(path, gpu_ctx.decode_ply)
        c_pos, c_tris = orc.ply_read(open(path, "rb").read())
        bad = np.nonzero(g_pos.view(np.uint32) != c_pos.view(np.uint32))
        assert len(bad[0]) == 0, [(rows[i][k], float(g_pos[i, k]), float(c_pos[i, k])) for i, k in zip(*bad)][:5]
        assert np.array_equal(g_tris, c_tris)


def _random_decimals(rng, n):
    """Decimal spellings of every shape an exporter or a person produces: 1-25 significant digits, with and without a point, exponents
    over the whole float range and beyond, signs, leading zeros."""
    out = []
    for _ in range(n):
        nd = int(rng.choice([1, 2, 3, 6, 7, 8, 9, 10, 12, 15, 17, 19, 20, 25]))
        digits = "".join(str(d) for d in rng.randint(0, 10, nd))
        point = int(rng.randint(0, nd + 1))
        body = (digits[:point] or ("0" if rng.rand() < 0.7 else "")) + ("." + digits[point:] if point < nd or rng.rand() < 0.2 else "")
        if body in ("", "."):
            body = "0"
        kind = rng.randint(4)
        if kind == 0:
            text = body
        elif kind == 1:
            text = body + "e%d" % rng.randint(-12, 13)
        elif kind == 2:
            text = body + "E%+d" % rng.randint(-50, 45)
        else:
            text = body + "e%d" % rng.choice([-46, -45, -44, -39, -38, -37, 37, 38])
        out.append(("-" if rng.rand() < 0.4 else ("+" if rng.rand() < 0.05 else "")) + text)
    return out


@pytest.mark.parametrize("pos_type", ["f4", "f8"])
def test_ascii_random_decimals(rt, orc, gpu_ctx, tmp_path, pos_type):
    """60 000 random spellings per property type: the device's conversion (with its host hand-over) against the oracle's reader."""
    rng = np.random.RandomState(77)
    toks = _random_decimals(rng, 60000)
    # values that overflow a float make the stream extraction fail; the reference's reader then leaves its element unchanged in a way that
    # depends on the previous token: keep to spellings the C library accepts for the property's type
    keep = [t for t in toks if np.isfinite(np.float32(float(t))) or pos_type == "f8"]
    keep = [t for t in keep if np.isfinite(float(t))]
    keep = keep[: len(keep) // 3 * 3]
    rows = [keep[i:i + 3] for i in range(0, len(keep), 3)]
    path = os.path.join(tmp_path, "r.ply")
    ply_util.write_ascii_ply(path, rows, [[0, 1, 2]], pos_type=pos_type)
    g_pos, _ = scene._read_ply(path, gpu_ctx.decode_ply)
    c_pos, _ = orc.ply_read(open(path, "rb").read())
    bad = np.nonzero(g_pos.view(np.uint32) != c_pos.view(np.uint32))
    assert len(bad[0]) == 0, [(rows[i][k], float(g_pos[i, k]), float(c_pos[i, k])) for i, k in zip(*bad)][:8]
    anchor = np.array([[np.float32(float(t)) for t in row] for row in rows[:2000]], np.float32)      # Python's conversion, where it is a single rounding
    same = anchor.view(np.uint32) == c_pos[:2000].view(np.uint32)
    assert same.mean() > 0.999


def test_ascii_every_coordinate_needs_the_host_conversion(rt, orc, gpu_ctx, tmp_path):
    """A mesh written with 20 significant digits (setprecision(20)): the device can convert none of its 75 000 coordinates itself
    (more than 19 digits), so every one of them goes through the hand-over list — which has room for all of a file's coordinates —,
    is converted by the host the way the reference's reader does, and is patched in with one upload."""
    rng = np.random.RandomState(5)
    nv = 25_000
    vals = (rng.rand(nv, 3) * 200 - 100) * (10.0 ** rng.randint(-3, 4, size=(nv, 1)))
    rows = [["%.20g" % v for v in row] for row in vals]
    assert all(len(t.replace("-", "").replace(".", "").lstrip("0")) >= 20 for t in rows[0])
    faces = [[i, i + 1, i + 2] for i in range(0, nv - 2, 3)]
    path = os.path.join(tmp_path, "p20.ply")
    ply_util.write_ascii_ply(path, rows, faces)
    g_pos, g_tris = scene._read_ply(path, gpu_ctx.decode_ply)
    c_pos, c_tris = orc.ply_read(open(path, "rb").read())
    assert np.array_equal(g_pos.view(np.uint32), c_pos.view(np.uint32)) and np.array_equal(g_tris, c_tris)


def test_loader_falls_back_to_the_host_reader(rt, orc, gpu_ctx, tmp_path):
    """A body the device decoder refuses must still load when the general reader (and happly) can read it: here the decoder is
    made to refuse, and the loader's result equals the oracle reader's."""
    path = os.path.join(tmp_path, "f.ply")
    ply_util.make_ascii_case(path, "blender_tris")
    calls = []

    def refusing(vb, fb, layout):
        calls.append(1)
        raise rt.RtError(rt.abi.RT_ERR_TOO_LARGE, "refused for the test")

    g_pos, g_tris = scene._read_ply(path, refusing)
    c_pos, c_tris = orc.ply_read(open(path, "rb").read())
    assert calls and np.array_equal(g_pos.astype(np.float32).view(np.uint32), c_pos.view(np.uint32)) and np.array_equal(g_tris, c_tris)


def test_ascii_million_triangles(rt, orc, gpu_ctx, tmp_path):
    """The config-2 stand-in mesh as an ASCII PLY in Blender's spelling (the format of every PLY the reference ships): 500 k vertices,
    1 M faces, 45 MB of text."""
    synth = importlib.import_module("ray-tracer_amd.synth")
    pos, faces = synth.torus_knot_mesh(1_000_000, 7)
    path = os.path.join(tmp_path, "knot_ascii.ply")
    with open(path, "w") as fh:
        fh.write("ply\nformat ascii 1.0\ncomment Created by tests\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(pos), len(faces)))
        np.savetxt(fh, np.asarray(pos, np.float64), fmt="%.6f")
        np.savetxt(fh, np.concatenate([np.full((len(faces), 1), 3), np.asarray(faces)], axis=1), fmt="%d")
    ms = []

    def decoder(vb, fb, layout):
        p, t, m = gpu_ctx.decode_ply(vb, fb, layout, timing=True)
        ms.append(m)
        return p, t

    t0 = time.time()
    g_pos, g_tris = scene._read_ply(path, decoder)
    t1 = time.time()
    c_pos, c_tris = orc.ply_read(open(path, "rb").read())
    t2 = time.time()
    assert np.array_equal(g_pos.view(np.uint32), c_pos.view(np.uint32)) and np.array_equal(g_tris, c_tris) and len(g_tris) == len(faces)
    print("1M-triangle ASCII PLY (%.0f MB): device decode %.2f ms of kernels, %.0f ms incl. file read and copies; the oracle's reader (one host core) %.0f ms"
          % (os.path.getsize(path) / 1e6, ms[0], 1e3 * (t1 - t0), 1e3 * (t2 - t1)))


def test_ascii_malformed(rt, gpu_ctx, tmp_path):
    path = os.path.join(tmp_path, "m.ply")
    ply_util.write_ascii_ply(path, ply_util.ascii_positions(1, 10, "blender"), [[0, 1, 2]] * 4)
    with open(path, "rb") as fh:
        fmt, elements = scene._ply_header(fh)
        body = fh.read()
    L = scene.ply_layout(elements, ascii=True)
    pos, tris = gpu_ctx.decode_ply(body, b"", L)
    assert pos.shape == (10, 3) and len(tris) == 4
    with pytest.raises(rt.RtError):
        gpu_ctx.decode_ply(body[:-12], b"", L)                              # a face line is missing
    with pytest.raises(rt.RtError):
        gpu_ctx.decode_ply(body.replace(b"0 1 2", b"0 x 2", 1), b"", L)     # not an index
    lines = body.split(b"\n")
    lines[3] = b"1.0 2.0"                                                   # a vertex line with two tokens
    with pytest.raises(rt.RtError):
        gpu_ctx.decode_ply(b"\n".join(lines), b"", L)
    pos2, tris2 = gpu_ctx.decode_ply(body, b"", L)                          # and the context still works afterwards
    assert np.array_equal(pos, pos2) and np.array_equal(tris, tris2)


def test_empty_and_truncated(rt, gpu_ctx, tmp_path):
    path = os.path.join(tmp_path, "e.ply")
    ply_util.write_binary_ply(path, np.zeros((0, 3)), [])
    pos, tris = scene._read_ply(path, gpu_ctx.decode_ply)
    assert pos.shape == (0, 3) and tris.shape == (0, 3)
    p, f = ply_util.random_mesh(2, 100, 500, (3, 4))
    ply_util.write_binary_ply(path, p, f)
    with open(path, "rb") as fh:
        fmt, elements = scene._ply_header(fh)
        body = fh.read()
    L = scene.ply_layout(elements)
    vbytes = int(L.n_vertices) * int(L.vertex_stride)
    with pytest.raises(rt.RtError):
        gpu_ctx.decode_ply(body[:vbytes], body[vbytes:-40], L)          # the last faces are cut off
    L.count_size = 3
    with pytest.raises(rt.RtError):
        gpu_ctx.decode_ply(body[:vbytes], body[vbytes:], L)


def test_million_triangles(rt, gpu_ctx, tmp_path):
    """The config-2 stand-in mesh as a binary PLY (the stand-in for the missing dragon_remeshed.ply)."""
    synth = importlib.import_module("ray-tracer_amd.synth")
    pos, faces = synth.torus_knot_mesh(1_000_000, 7)
    path = os.path.join(tmp_path, "knot.ply")
    vdt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
    fdt = np.dtype([("k", "u1"), ("i", "<i4", 3)])
    v = np.zeros(len(pos), vdt)
    v["x"], v["y"], v["z"] = pos[:, 0], pos[:, 1], pos[:, 2]
    fa = np.zeros(len(faces), fdt)
    fa["k"], fa["i"] = 3, faces
    with open(path, "wb") as fh:
        fh.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                  "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(pos), len(faces))).encode())
        fh.write(v.tobytes())
        fh.write(fa.tobytes())
    ms = []

    def decoder(vb, fb, layout):
        p, t, m = gpu_ctx.decode_ply(vb, fb, layout, timing=True)
        ms.append(m)
        return p, t

    t0 = time.time()
    g_pos, g_tris = scene._read_ply(path, decoder)
    t1 = time.time()
    c_pos, c_tris = scene._read_ply(path)
    t2 = time.time()
    assert np.array_equal(g_pos.view(np.uint32), np.asarray(pos, np.float32).view(np.uint32)) and np.array_equal(g_tris, np.asarray(faces, np.int64))
    assert np.array_equal(c_pos, g_pos) and np.array_equal(c_tris, g_tris)
    print("1M-triangle PLY: device decode %.3f ms of kernels, %.1f ms incl. file read and copies; numpy reader %.1f ms" % (ms[0], 1e3 * (t1 - t0), 1e3 * (t2 - t1)))


# ---- decoded geometry stays in HBM: PLY -> rt_ply_decode_to -> rt_scene_upload(geometry_on_device) -> frame ----------------------
def _ply_scene(rt, pos, tris, compact, smooth):
    """A mesh + two instances of it + a floor + a sphere, lit by two point lights (the shape of the reference's PLY scenes)."""
    b = rt.SceneBuilder(compact=compact)
    b.soup.max_depth, b.soup.background = 3, np.array([0.05, 0.06, 0.1], np.float32)
    b.soup.ambient = np.array([20, 20, 20], np.float32)
    m0 = b.add_material(ka=(1, 1, 1), kd=(0.7, 0.5, 0.3), ks=(0.4, 0.4, 0.4), phong=20)
    m1 = b.add_material(type=rt.abi.RT_MAT_MIRROR, ka=(1, 1, 1), kd=(0.2, 0.2, 0.2), ks=(0.2, 0.2, 0.2), km=(0.6, 0.6, 0.6), phong=10)
    b.add_light(rt.abi.RT_LIGHT_POINT, (30, 60, 80), (40000, 40000, 40000))
    b.add_light(rt.abi.RT_LIGHT_POINT, (-70, 20, 50), (15000, 12000, 9000))
    sc = rt.scene
    mesh = b.add_mesh(pos, tris, xform=sc.mat_mul(sc.mat_translate((0, -100 * 0.12, -20)), sc.mat_scale((0.12, 0.12, 0.12))), material=m0, smooth=smooth)
    b.add_instance(mesh, xform=sc.mat_translate((35, 0, 0)), material=m1)
    b.add_instance(mesh, xform=sc.mat_mul(sc.mat_translate((-35, 5, 0)), sc.mat_rotate(40, (0, 1, 0))), reset_transform=False)
    b.add_triangle((-200, -25, -200), (200, -25, 200), (200, -25, -200), material=m0)
    b.add_triangle((-200, -25, -200), (-200, -25, 200), (200, -25, 200), material=m0)
    b.add_sphere((0, 25, -30), 8.0, material=m1)
    b.add_lookat_camera((0, 15, 90), (0, 0, -20), (0, 1, 0), 40.0, 1.0, 240, 160)
    return b.finish("ply_device")


@pytest.mark.parametrize("kind,smooth", [("ascii", True), ("binary", False)])
def test_ply_stays_on_device_until_the_frame(rt, orc, gpu_ctx, tmp_path, kind, smooth):
    torch = pytest.importorskip("torch")
    dg = importlib.import_module("ray-tracer_amd.device_geometry")
    import parity_util
    path = os.path.join(tmp_path, "mesh.ply")
    synth = importlib.import_module("ray-tracer_amd.synth")
    pos, faces = synth.torus_knot_mesh(24000, 3)
    if kind == "ascii":
        ply_util.write_ascii_ply(path, [["%.6f" % v for v in row] for row in pos], [list(f) for f in faces])
    else:
        quads = [list(f) for f in faces[:4000]] + [[f[0], f[1], f[2], f[1]] for f in faces[4000:6000]]     # triangles and quads
        ply_util.write_binary_ply(path, pos, quads, vertex_extra=(("nx", "f4"),))
    # the checker: the oracle's reader -> a written-out scene -> the oracle's renderer
    o_pos, o_tris = orc.ply_read(open(path, "rb").read())
    soup = _ply_scene(rt, o_pos, o_tris, compact=True, smooth=smooth)
    want, want_pid, _, _ = orc.OracleScene(soup).render(0, nthreads=8, dynamic=True)           # (the oracle writes the compact forms out first)
    written_out = _ply_scene(rt, o_pos, o_tris, compact=False, smooth=smooth)                  # the same scene as the reference's parser holds it:
    ref_order, _, _, _ = orc.OracleScene(written_out).render(0, nthreads=8, dynamic=True)      # clones numbered where they are declared
    assert np.array_equal(ref_order, want)
    # the product: the file's bytes -> device decoder -> arena in HBM -> device ingest (instances and runs expanded there)
    arena = dg.DeviceGeometry(len(o_pos) + 6, 2 * max(len(faces), 6000) + 2)
    v0, t0, n = arena.add_ply(gpu_ctx, path, obj=0)
    assert (v0, t0, n) == (0, 0, len(o_tris))
    assert np.array_equal(arena.pos[:len(o_pos)].cpu().numpy().view(np.uint32), o_pos.view(np.uint32))
    assert np.array_equal(arena.tri_vert[:n].cpu().numpy(), o_tris)
    floor = soup.vert_pos[len(o_pos):]
    for k in range(2):                                                          # the two standalone triangles: objects 3 and 4
        arena.add_host(floor[3 * k:3 * k + 3], None, [[0, 1, 2]], int(soup.tri_object[len(o_tris) + k]))
    assert arena.n_vertices == len(soup.vert_pos) and arena.n_triangles == len(soup.tri_vert)
    arena.attach(soup)
    gpu_ctx.upload(soup)                                                        # > 4096 primitives: ingest and build run on the device
    got, got_pid, _, _ = gpu_ctx.render(0)
    del arena
    assert np.array_equal(got_pid, want_pid)
    parity_util.compare_frames(got, want)
