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


# ---- ascii files --------------------------------------------------------------------------------------------------------------
# Number spellings an exporter can produce.  The expected value of every one is what the C library's stream extraction gives
# (the oracle's reader calls it, as the reference's happly does); the device decoder converts most of them itself and hands
# the rest to the host.
TRICKY_NUMBERS = [
    "0", "-0", "0.0", "-0.000", "1", "-1", "+2.5", ".5", "5.", "-.25", "0.1", "0.2", "0.3", "-0.163917", "0.335194", "123456.789",
    "16777216", "16777217", "16777219", "33554435", "4294967296", "0.000001", "1e-6", "1E+6", "1e10", "1e11", "9.999999e9",
    "3.4028235e38", "3.4028234663852886e38", "1.17549435e-38", "1.1754942e-38", "1e-40", "1.4e-45", "7e-46", "1e-50",
    "0.30000001192092896", "0.300000011920928955078125", "1.00000011920928955078125", "1.000000059604644775390625",
    "1.0000000596046447753906250000001", "1.00000005960464477539062499999", "8388608.5", "8388609.5", "0.1e-5", "12345678901234567890",
    "1234567890123456789012345678901234567890e-20", "0.000000000000000000000000000000000000011754943508222875", "2.2250738585072014e-308",
    "1e22", "1e23", "123456789012345678e-5", "9007199254740993", "9007199254740992", "3.14159274101257324", "2.71828174591064453125",
    "100000000000000000000000", "5e-1", "5e+0", "00012.5", "-00.5e01",
]


def write_ascii_ply(path, pos_text, faces, pos_type="f4", vertex_extra=(), face_pre=(), face_post=(), xyz_first=True, list_name="vertex_indices",
                    count_type="u1", index_type="i4", crlf=False, pad_spaces=False, trailing_newline=True, comments=True):
    """pos_text: [nv][3] strings written verbatim.  The extra scalar properties get small integers."""
    nl = "\r\n" if crlf else "\n"
    vprops = [("x", pos_type), ("y", pos_type), ("z", pos_type)]
    vprops = (vprops + list(vertex_extra)) if xyz_first else (list(vertex_extra) + vprops)
    head = ["ply", "format ascii 1.0"] + (["comment written by tests/ply_util.py"] if comments else []) + ["element vertex %d" % len(pos_text)]
    head += ["property %s %s" % (_T[t], n) for n, t in vprops]
    head += ["element face %d" % len(faces)]
    head += ["property %s %s" % (_T[t], n) for n, t in face_pre]
    head += ["property list %s %s %s" % (_T[count_type], _T[index_type], list_name)]
    head += ["property %s %s" % (_T[t], n) for n, t in face_post]
    head += ["end_header"]
    sep = "  " if pad_spaces else " "
    lines = []
    for i, p in enumerate(pos_text):
        extra = [str((i * (j + 3)) % 101) for j in range(len(vertex_extra))]
        toks = (list(p) + extra) if xyz_first else (extra + list(p))
        lines.append((" " if pad_spaces else "") + sep.join(toks) + (" " if pad_spaces else ""))
    for i, f in enumerate(faces):
        toks = [str((i * 7 + j) % 97) for j in range(len(face_pre))] + [str(len(f))] + [str(int(v)) for v in f] + [str((i * 5 + j) % 89) for j in range(len(face_post))]
        lines.append(sep.join(toks))
    text = nl.join(head) + nl + nl.join(lines) + (nl if trailing_newline and lines else "")
    with open(path, "wb") as fh:
        fh.write(text.encode("ascii"))


def ascii_positions(seed, nv, style):
    """[nv][3] coordinate strings.  style: 'blender' (%.6f, the spelling of the reference's own Blender-exported meshes), 'g9'
    (%.9g, round-trip exact), 'sci' (%e), 'mixed' (all of them plus the tricky spellings)."""
    rng = np.random.RandomState(seed)
    vals = (rng.rand(nv, 3) * 200 - 100) * (10.0 ** rng.randint(-6, 7, size=(nv, 1)))
    fm = {"blender": ["%.6f"], "g9": ["%.9g"], "sci": ["%e"], "mixed": ["%.6f", "%.9g", "%e", "%.17g", "%.3f", "%d", "%.12e", "%.25f"]}[style]
    out = []
    for i in range(nv):
        row = []
        for k in range(3):
            f = fm[(i * 3 + k) % len(fm)]
            row.append(f % (int(vals[i, k]) if f == "%d" else vals[i, k]))
        out.append(row)
    if style == "mixed":
        for j, t in enumerate(TRICKY_NUMBERS):
            out[(j * 7) % nv][j % 3] = t
    return out


ASCII_CASES = {
    "blender_tris": dict(style="blender", sizes=(3,)),
    "g9_quads": dict(style="g9", sizes=(4,)),
    "sci_mixed_faces": dict(style="sci", sizes=(3, 4, 5)),
    "tricky_numbers": dict(style="mixed", sizes=(3, 4)),
    "tricky_numbers_double": dict(style="mixed", sizes=(3,), pos_type="f8"),
    "crlf_lines": dict(style="blender", sizes=(3, 4), crlf=True),
    "padded_spaces": dict(style="g9", sizes=(3, 4), pad_spaces=True),
    "no_final_newline": dict(style="blender", sizes=(4,), trailing_newline=False),
    "vertex_extras": dict(style="blender", sizes=(3,), vertex_extra=(("nx", "f4"), ("red", "u1"))),
    "extras_before_xyz": dict(style="g9", sizes=(3, 4), vertex_extra=(("q", "f4"),), xyz_first=False),
    "face_scalars": dict(style="blender", sizes=(3, 4, 6), face_pre=(("flags", "u1"),), face_post=(("mat", "i2"),)),
    "vertex_index_name": dict(style="blender", sizes=(3,), list_name="vertex_index"),
}


def make_ascii_case(path, name, seed=11, nv=500, nf=1500):
    kw = dict(ASCII_CASES[name])
    rng = np.random.RandomState(seed + 1)
    sizes = kw.pop("sizes")
    pos_text = ascii_positions(seed, nv, kw.pop("style"))
    faces = [rng.randint(nv, size=int(rng.choice(sizes))).tolist() for _ in range(nf)]
    write_ascii_ply(path, pos_text, faces, **kw)
    return pos_text, faces
