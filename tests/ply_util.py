"""Writers of small PLY files for the reader / decoder tests (test data generators, not product code)."""
import numpy as np

_T = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}


def write_binary_ply(path, pos, faces, pos_type="f4", count_type="u1", index_type="i4", vertex_extra=(), face_pre=(), face_post=(), xyz_first=True,
                     list_name="vertex_indices", big_endian=False):
    """pos [nv,3]; faces: list of index lists (any sizes).  vertex_extra / face_pre / face_post: sequences of (name, numpy type code)
    scalar properties filled with a deterministic pattern.  xyz_first=False puts the extra vertex properties before x, y, z."""
    E = ">" if big_endian else "<"
    pos = np.asarray(pos, np.float64)
    vprops = [("x", pos_type), ("y", pos_type), ("z", pos_type)]
    vprops = (vprops + list(vertex_extra)) if xyz_first else (list(vertex_extra) + vprops)
    head = ["ply", "format binary_%s_endian 1.0" % ("big" if big_endian else "little"), "comment written by tests/ply_util.py", "element vertex %d" % len(pos)]
    head += ["property %s %s" % (_T[t], n) for n, t in vprops]
    head += ["element face %d" % len(faces)]
    head += ["property %s %s" % (_T[t], n) for n, t in face_pre]
    head += ["property list %s %s %s" % (_T[count_type], _T[index_type], list_name)]
    head += ["property %s %s" % (_T[t], n) for n, t in face_post]
    head += ["end_header"]
    vdt = np.dtype([(n, E + t) for n, t in vprops])
    v = np.zeros(len(pos), vdt)
    for k, n in enumerate("xyz"):
        v[n] = pos[:, k]
    for j, (n, t) in enumerate(vertex_extra):
        v[n] = (np.arange(len(pos)) * (j + 3) % 101).astype(t)
    body = [v.tobytes()]
    for i, f in enumerate(faces):
        for j, (n, t) in enumerate(face_pre):
            body.append(np.array([(i * 7 + j) % 97], E + t).tobytes())
        body.append(np.array([len(f)], E + count_type).tobytes())
        body.append(np.asarray(f, E + index_type).tobytes())
        for j, (n, t) in enumerate(face_post):
            body.append(np.array([(i * 5 + j) % 89], E + t).tobytes())
    with open(path, "wb") as fh:
        fh.write(("\n".join(head) + "\n").encode("ascii"))
        fh.write(b"".join(body))


def expected_triangles(faces):
    """SceneParser.cpp:993-1023: three indices -> one triangle, four -> (0,1,2), (2,3,0), anything else dropped."""
    out = []
    for f in faces:
        if len(f) == 3:
            out.append(tuple(f))
        elif len(f) == 4:
            out.append((f[0], f[1], f[2]))
            out.append((f[2], f[3], f[0]))
    return np.array(out, np.int64).reshape(-1, 3)


def random_mesh(seed, nv, nf, sizes=(3,)):
    rng = np.random.RandomState(seed)
    pos = (rng.rand(nv, 3) * 200 - 100)
    faces = [rng.randint(nv, size=int(rng.choice(sizes))).tolist() for _ in range(nf)]
    return pos, faces


CASES = {
    "tris_uchar_int": dict(sizes=(3,)),
    "quads": dict(sizes=(4,)),
    "mixed_3_4_5": dict(sizes=(3, 4, 5)),
    "pentagons_only": dict(sizes=(5,)),
    "double_coords": dict(sizes=(3,), pos_type="f8"),
    "ushort_indices_uint_count": dict(sizes=(3, 4), count_type="u4", index_type="u2"),
    "vertex_normals_colours": dict(sizes=(3,), vertex_extra=(("nx", "f4"), ("ny", "f4"), ("nz", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1"))),
    "extras_before_xyz": dict(sizes=(4,), vertex_extra=(("quality", "f4"), ("flags", "u1")), xyz_first=False),
    "face_scalars_around_list": dict(sizes=(3, 4), face_pre=(("flags", "u1"),), face_post=(("quality", "f4"), ("mat", "i2"))),
    "vertex_index_name": dict(sizes=(3,), list_name="vertex_index"),
}
