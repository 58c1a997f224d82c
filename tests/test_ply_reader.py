"""PLY readers on the host against hand-built files: the reference reads its PLYs through happly and turns the faces into
triangles in SceneParser.cpp:983-1023 (quad -> (0,1,2), (2,3,0); other sizes dropped).  Two readers are checked here: the
oracle's restatement of that path (oracle/rt_oracle.cpp, orc_ply_read: the checker of the GPU decoder in tests/test_gpu_ply.py)
against the definition of each written file, and the loader's numpy reader against the oracle's.  The reference's own PLY
files are ASCII, three floats per vertex in Blender's %.6f spelling and uchar/int index lists; the ASCII cases below use the same shape
(its files themselves are not part of this repository)."""
import importlib
import os

import numpy as np
import pytest

import ply_util

scene = importlib.import_module("ray-tracer_amd.scene")
abi = importlib.import_module("ray-tracer_amd").abi


@pytest.mark.parametrize("name", sorted(ply_util.CASES))
def test_binary_reader(tmp_path, name):
    kw = dict(ply_util.CASES[name])
    pos, faces = ply_util.random_mesh(5, 50, 200, kw.pop("sizes"))
    path = os.path.join(tmp_path, name + ".ply")
    ply_util.write_binary_ply(path, pos, faces, **kw)
    got_pos, got_tris = scene._read_ply(path)
    want = pos.astype(np.float32) if kw.get("pos_type", "f4") == "f8" else pos.astype(np.float32)
    assert np.array_equal(got_pos, want)
    assert np.array_equal(got_tris, ply_util.expected_triangles(faces))


def test_big_endian_reader(tmp_path):
    """happly reads binary_big_endian too; the host reader does, the device decoder is little-endian only (the loader then keeps the host path)."""
    pos, faces = ply_util.random_mesh(6, 40, 120, (3, 4))
    path = os.path.join(tmp_path, "be.ply")
    ply_util.write_binary_ply(path, pos, faces, big_endian=True, index_type="u2", face_post=(("q", "f4"),))
    called = []
    got_pos, got_tris = scene._read_ply(path, lambda *a: called.append(1))
    assert not called
    assert np.array_equal(got_pos, pos.astype(np.float32)) and np.array_equal(got_tris, ply_util.expected_triangles(faces))


def test_ascii_reader(tmp_path):
    path = os.path.join(tmp_path, "a.ply")
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nelement face 3\n"
                "property list uchar int vertex_index\nend_header\n0 0 0\n0 0 1\n1 0 1\n1 0 0\n4 0 1 2 3\n3 0 1 2\n5 0 1 2 3 0\n")
    pos, tris = scene._read_ply(path)
    assert pos.shape == (4, 3) and pos.dtype == np.float32
    assert tris.tolist() == [[0, 1, 2], [2, 3, 0], [0, 1, 2]]


def test_layout_from_header(tmp_path):
    kw = dict(ply_util.CASES["face_scalars_around_list"])
    pos, faces = ply_util.random_mesh(1, 10, 7, kw.pop("sizes"))
    path = os.path.join(tmp_path, "l.ply")
    ply_util.write_binary_ply(path, pos, faces, vertex_extra=(("nx", "f4"),), **kw)
    with open(path, "rb") as f:
        fmt, elements = scene._ply_header(f)
    L = scene.ply_layout(elements)
    assert fmt == "binary_little_endian"
    assert (L.abi_version, L.pos_type, L.n_vertices, L.vertex_stride, list(L.pos_offset)) == (abi.RT_ABI_VERSION, 0, 10, 16, [0, 4, 8])
    assert (L.n_faces, L.face_pre_bytes, L.count_size, L.index_size, L.face_post_bytes) == (7, 1, 1, 4, 6)
    # an element the decoder does not know sends the file to the general reader
    assert scene.ply_layout(elements + [{"name": "edge", "count": 0, "props": []}]) is None


# ---- the oracle's reader (restating happly + the scene parser's PLY branch) ---------------------------------------------------
@pytest.mark.parametrize("name", sorted(ply_util.CASES))
def test_oracle_reader_binary(orc, tmp_path, name):
    kw = dict(ply_util.CASES[name])
    pos, faces = ply_util.random_mesh(5, 50, 200, kw.pop("sizes"))
    path = os.path.join(tmp_path, name + ".ply")
    ply_util.write_binary_ply(path, pos, faces, **kw)
    o_pos, o_tris = orc.ply_read(open(path, "rb").read())
    assert np.array_equal(o_pos, pos.astype(np.float32)) and np.array_equal(o_tris, ply_util.expected_triangles(faces))
    n_pos, n_tris = scene._read_ply(path)
    assert np.array_equal(n_pos.view(np.uint32), o_pos.view(np.uint32)) and np.array_equal(n_tris, o_tris)


@pytest.mark.parametrize("name", sorted(ply_util.ASCII_CASES))
def test_oracle_reader_ascii(orc, tmp_path, name):
    path = os.path.join(tmp_path, name + ".ply")
    pos_text, faces = ply_util.make_ascii_case(path, name)
    o_pos, o_tris = orc.ply_read(open(path, "rb").read())
    assert np.array_equal(o_tris, ply_util.expected_triangles(faces))
    if ply_util.ASCII_CASES[name]["style"] == "g9" and ply_util.ASCII_CASES[name].get("pos_type", "f4") == "f4":
        # nine significant digits are never halfway between two floats in double precision, so Python's conversion + one cast is an
        # anchor that goes through neither reader
        assert np.array_equal(o_pos, np.array([[np.float32(float(t)) for t in row] for row in pos_text], np.float32))
    n_pos, n_tris = scene._read_ply(path)
    assert np.array_equal(n_pos.view(np.uint32), o_pos.view(np.uint32)) and np.array_equal(n_tris, o_tris)      # -0 and subnormals included


def test_oracle_reader_number_spellings(orc, tmp_path):
    """Known answers of the text -> float conversion (the value nearest the decimal, ties to even; one rounding, not two)."""
    known = {"0.1": 0x3dcccccd, "-0": 0x80000000, "16777217": 0x4b800000, "16777219": 0x4b800002, "1e-40": 0x000116c2, "1.4e-45": 0x00000001,
             "7e-46": 0x00000000, "3.4028235e38": 0x7f7fffff, "1.000000059604644775390625": 0x3f800000, "1.0000000596046447753906250000001": 0x3f800001,
             "1.00000005960464477539062499999": 0x3f800000, "1.00000011920928955078125": 0x3f800001, "8388608.5": 0x4b000000, "8388609.5": 0x4b000002,
             ".5": 0x3f000000, "5.": 0x40a00000, "+2.5": 0x40200000, "-00.5e01": 0xc0a00000, "1e23": 0x65a96816}
    toks = sorted(known)
    path = os.path.join(tmp_path, "k.ply")
    ply_util.write_ascii_ply(path, [[t, "0", "1"] for t in toks], [[0, 1, 2]])
    o_pos, _ = orc.ply_read(open(path, "rb").read())
    n_pos, _ = scene._read_ply(path)
    for i, t in enumerate(toks):
        assert int(o_pos[i, 0].view(np.uint32)) == known[t], (t, hex(int(o_pos[i, 0].view(np.uint32))))
        assert int(n_pos[i, 0].view(np.uint32)) == known[t], t


def test_oracle_reader_rejects(orc, tmp_path):
    path = os.path.join(tmp_path, "t.ply")
    ply_util.write_ascii_ply(path, ply_util.ascii_positions(1, 10, "blender"), [[0, 1, 2]] * 4)
    data = open(path, "rb").read()
    with pytest.raises(ValueError):
        orc.ply_read(data[:-12])                                        # the last face line is missing
    with pytest.raises(ValueError):
        orc.ply_read(data.replace(b"element vertex 10", b"element vertex 11"))


def test_ascii_layout_counts_tokens(tmp_path):
    path = os.path.join(tmp_path, "l.ply")
    ply_util.make_ascii_case(path, "extras_before_xyz")
    with open(path, "rb") as f:
        fmt, elements = scene._ply_header(f)
    L = scene.ply_layout(elements, ascii=True)
    assert fmt == "ascii" and L.format == abi.RT_PLY_ASCII and (L.vertex_stride, list(L.pos_offset)) == (4, [1, 2, 3])
    ply_util.make_ascii_case(path, "face_scalars")
    with open(path, "rb") as f:
        fmt, elements = scene._ply_header(f)
    L = scene.ply_layout(elements, ascii=True)
    assert (L.face_pre_bytes, L.face_post_bytes, L.vertex_stride) == (1, 1, 3)


def test_reference_ply_files_pin_the_oracle_reader(scene_path):
    """The four PLY files the reference ships (src/scenes/ply/) went through the oracle's reader in the build container
    (tools/make_ply_golden.py); tests/golden/ref_ply_digests.json holds counts and SHA-256 digests of what it decoded.  The scene
    fixtures scenes/tap_0200.npz and scenes/cube_point.npz were produced from the same files by the Python loader's own PLY reader
    (another implementation, which numbers a mesh's vertices in the order the faces first use them): every PLY mesh is one object
    of the fixture, and the corner positions of its triangles, in face order, must be bit for bit what the oracle's reader decoded
    (SceneParser.cpp:983-1023: positions rounded to float, quads split into (0,1,2), (2,3,0))."""
    import hashlib
    import json
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_ply_digests.json")))
    for scene, files in (("tap_0200", ["tap_tap", "tap_water_200"]), ("cube_point", ["plane", "cube"])):
        d = np.load(scene_path(scene), allow_pickle=False)
        pos, tri, obj = np.asarray(d["vert_pos"], np.float32), np.asarray(d["tri_vert"], np.int64), np.asarray(d["tri_object"])
        digests = {}
        for o in np.unique(obj):
            rows = tri[obj == o]
            digests[hashlib.sha256(np.ascontiguousarray(pos[rows]).tobytes()).hexdigest()] = len(rows)
        for f in files:
            g = gold[f]
            assert digests.get(g["sha256_triangle_corners_f32"]) == g["n_triangles"], (scene, f)
