"""Host-side scene description for the MI355X render path.

`SceneSoup` is the numpy form of `rt_scene_desc` (include/rt_mi355.h).  `load_xml` restates what the
reference's SceneParser (src/SceneParser.cpp:40-1304) + Mesh/Triangle/Sphere constructors put into
`Scene::primitives`, including the ordering quirks that decide the BVH topology (Q5: every <Mesh> triangle is
pushed twice, <MeshInstance> triangles once; brdf / material / texture / transform lookups by push order or by
single-character id).  Scene *parsing* is outside the accelerated path; it only has to produce the same
inputs.  Matrix composition follows glm's operation order in float32 (src/acmath.h:958-1014), but cos/sin come
from numpy rather than glibc: last-ulp differences in rotation matrices are possible ("parity unpinned").
"""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import abi

F32 = np.float32


# --------------------------------------------------------------------------------------------
# float32 matrix helpers in glm memory order: m[col, row]
# --------------------------------------------------------------------------------------------
def mat_identity():
    return np.eye(4, dtype=F32)


def mat_mul(a, b):
    """glm operator*(mat4, mat4): Result[c] = ((A0*b[c][0] + A1*b[c][1]) + A2*b[c][2]) + A3*b[c][3]."""
    r = np.empty((4, 4), dtype=F32)
    for c in range(4):
        r[c] = ((a[0] * b[c, 0] + a[1] * b[c, 1]) + a[2] * b[c, 2]) + a[3] * b[c, 3]
    return r


def mat_translate(v):
    """glm::translate(mat4(1), v), src/acmath.h:977-981."""
    m = mat_identity()
    v = np.asarray(v, dtype=F32)
    m[3] = ((m[0] * v[0] + m[1] * v[1]) + m[2] * v[2]) + m[3]
    return m


def mat_scale(v):
    """glm::scale(mat4(1), v), src/acmath.h:966-970."""
    m = mat_identity()
    v = np.asarray(v, dtype=F32)
    r = mat_identity()
    r[0] = m[0] * v[0]
    r[1] = m[1] * v[1]
    r[2] = m[2] * v[2]
    return r


def mat_rotate(angle_deg, axis):
    """glm::rotate(mat4(1), angle * (3.14f / 180.0f), axis), src/acmath.h:972-975 (Q24)."""
    a = F32(F32(angle_deg) * F32(F32(3.14) / F32(180.0)))
    c = F32(np.cos(a))
    s = F32(np.sin(a))
    v = np.asarray(axis, dtype=F32)
    inv = F32(1.0) / F32(np.sqrt(F32(F32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])))
    ax = (v * inv).astype(F32)
    temp = (F32(F32(1) - c) * ax).astype(F32)
    rot = np.zeros((4, 4), dtype=F32)
    rot[0, 0] = c + temp[0] * ax[0]
    rot[0, 1] = temp[0] * ax[1] + s * ax[2]
    rot[0, 2] = temp[0] * ax[2] - s * ax[1]
    rot[1, 0] = temp[1] * ax[0] - s * ax[2]
    rot[1, 1] = c + temp[1] * ax[1]
    rot[1, 2] = temp[1] * ax[2] + s * ax[0]
    rot[2, 0] = temp[2] * ax[0] + s * ax[1]
    rot[2, 1] = temp[2] * ax[1] - s * ax[0]
    rot[2, 2] = c + temp[2] * ax[2]
    m = mat_identity()
    r = mat_identity()
    r[0] = (m[0] * rot[0, 0] + m[1] * rot[0, 1]) + m[2] * rot[0, 2]
    r[1] = (m[0] * rot[1, 0] + m[1] * rot[1, 1]) + m[2] * rot[1, 2]
    r[2] = (m[0] * rot[2, 0] + m[1] * rot[2, 1]) + m[2] * rot[2, 2]
    r[3] = m[3]
    return r


# --------------------------------------------------------------------------------------------
class SceneSoup:
    """Numpy mirror of rt_scene_desc plus the scene's cameras."""

    def __init__(self):
        self.max_depth = 0                      # Scene.cpp:37
        self.isect_eps = F32(0.0001)            # Scene.cpp:39 (never read from XML, Q-cfg)
        self.shadow_eps = F32(0.005)            # Scene.cpp:38
        self.background = np.zeros(3, F32)
        self.ambient = np.zeros(3, F32)
        self.bg_texture = -1
        self.vert_pos = np.zeros((0, 3), F32)
        self.vert_uv = np.zeros((0, 2), F32)
        self.tri_vert = np.zeros((0, 3), np.int32)
        self.tri_object = np.zeros((0,), np.int32)
        self.sph_center_radius = np.zeros((0, 4), F32)
        self.sph_object = np.zeros((0,), np.int32)
        self.objects = np.zeros((0,), abi.object_dtype)
        self.prim_shape = np.zeros((0,), np.int32)
        self.materials = np.zeros((0,), abi.material_dtype)
        self.lights = np.zeros((0,), abi.light_dtype)
        self.textures = np.zeros((0,), abi.texture_dtype)
        self.texels = np.zeros((0,), np.uint8)
        # compact forms (rt_instance / rt_prim_run): a soup either carries them (the library expands them, on the device for large
        # scenes) or is fully expanded (instances empty, prim_runs None)
        self.instances = np.zeros((0,), abi.instance_dtype)
        self.prim_runs = None
        self.device_geometry = None             # (vert_pos ptr, vert_uv ptr, tri_vert ptr, tri_object ptr, n_vertices, n_triangles, keepalive)
        self.cameras = []                       # list of dicts (see camera_desc)
        self.tonemap = None                     # (key, burn percent, saturation, gamma) of Scene::tmo, or None
        self.name = ""

    # ---- C view -------------------------------------------------------------------------
    def _contig(self):
        self.vert_pos = np.ascontiguousarray(self.vert_pos, F32).reshape(-1, 3)
        self.vert_uv = np.ascontiguousarray(self.vert_uv, F32).reshape(-1, 2)
        self.tri_vert = np.ascontiguousarray(self.tri_vert, np.int32).reshape(-1, 3)
        self.tri_object = np.ascontiguousarray(self.tri_object, np.int32)
        self.sph_center_radius = np.ascontiguousarray(self.sph_center_radius, F32).reshape(-1, 4)
        self.sph_object = np.ascontiguousarray(self.sph_object, np.int32)
        self.objects = np.ascontiguousarray(self.objects, abi.object_dtype)
        self.prim_shape = np.ascontiguousarray(self.prim_shape, np.int32)
        self.materials = np.ascontiguousarray(self.materials, abi.material_dtype)
        self.lights = np.ascontiguousarray(self.lights, abi.light_dtype)
        self.textures = np.ascontiguousarray(self.textures, abi.texture_dtype)
        self.texels = np.ascontiguousarray(self.texels, np.uint8)

    def desc(self):
        """Returns an abi.SceneDesc whose pointers reference this object's arrays (keep `self` alive)."""
        self._contig()
        assert len(self.vert_uv) == len(self.vert_pos)
        d = abi.SceneDesc()
        d.abi_version = abi.RT_ABI_VERSION
        d.max_depth = int(self.max_depth)
        d.isect_eps = float(self.isect_eps)
        d.shadow_eps = float(self.shadow_eps)
        for i in range(3):
            d.background[i] = float(self.background[i])
            d.ambient[i] = float(self.ambient[i])
        d.bg_texture = int(self.bg_texture)
        d.n_vertices = len(self.vert_pos)
        d.vert_pos = self.vert_pos.ctypes.data
        d.vert_uv = self.vert_uv.ctypes.data
        d.n_triangles = len(self.tri_vert)
        d.tri_vert = self.tri_vert.ctypes.data
        d.tri_object = self.tri_object.ctypes.data
        d.n_spheres = len(self.sph_center_radius)
        d.sph_center_radius = self.sph_center_radius.ctypes.data
        d.sph_object = self.sph_object.ctypes.data
        d.n_objects = len(self.objects)
        d.objects = self.objects.ctypes.data
        d.n_prims = len(self.prim_shape)
        d.prim_shape = self.prim_shape.ctypes.data
        d.n_materials = len(self.materials)
        d.n_lights = len(self.lights)
        d.n_textures = len(self.textures)
        d.materials = self.materials.ctypes.data
        d.lights = self.lights.ctypes.data
        d.textures = self.textures.ctypes.data
        d.texels = self.texels.ctypes.data if len(self.texels) else None
        d.texels_bytes = len(self.texels)
        self.instances = np.ascontiguousarray(self.instances, abi.instance_dtype)
        d.n_instances = len(self.instances)
        d.instances = self.instances.ctypes.data if len(self.instances) else None
        if self.prim_runs is not None:
            self.prim_runs = np.ascontiguousarray(self.prim_runs, abi.prim_run_dtype)
            d.n_prim_runs = len(self.prim_runs)
            d.prim_runs = self.prim_runs.ctypes.data
            d.n_prims, d.prim_shape = 0, None
        if self.device_geometry is not None:
            g = self.device_geometry
            d.vert_pos, d.vert_uv, d.tri_vert, d.tri_object, d.n_vertices, d.n_triangles = g[0], g[1], g[2], g[3], g[4], g[5]
            d.geometry_on_device = 1
        return d

    def expanded(self):
        """The same scene with <MeshInstance> clones and primitive runs written out (what the reference's Scene holds after
        parsing: Mesh::Clone, src/Mesh.cpp:126-180) — the form the CPU oracle takes."""
        if len(self.instances) == 0 and self.prim_runs is None:
            return self
        assert self.device_geometry is None
        import copy
        e = copy.copy(self)
        tv, to = [np.asarray(self.tri_vert, np.int32).reshape(-1, 3)], [np.asarray(self.tri_object, np.int32)]
        for ins in self.instances:
            a, n = int(ins["first_triangle"]), int(ins["n_triangles"])
            tv.append(tv[0][a:a + n])
            to.append(np.full(n, int(ins["object"]), np.int32))
        e.tri_vert, e.tri_object = np.concatenate(tv), np.concatenate(to)
        if self.prim_runs is not None:
            e.prim_shape = np.concatenate([(np.arange(int(r["first"]), int(r["first"]) + int(r["count"]), dtype=np.int64) | int(r["flags"])).astype(np.int32)
                                           for r in self.prim_runs]) if len(self.prim_runs) else np.zeros(0, np.int32)
        e.instances, e.prim_runs = np.zeros((0,), abi.instance_dtype), None
        e._contig()
        return e

    def camera_desc(self, index=0, **override):
        c = dict(self.cameras[index])
        c.update(override)
        d = abi.CameraDesc()
        for i in range(3):
            d.position[i] = float(c["position"][i])
            d.gaze[i] = float(c["gaze"][i])
            d.up[i] = float(c["up"][i])
        d.left, d.right, d.bottom, d.top = (float(x) for x in c["plane"])
        d.distance = float(c["distance"])
        d.nx, d.ny = int(c["nx"]), int(c["ny"])
        d.num_samples = int(c.get("num_samples", 1))
        d.focus_distance = float(c.get("focus_distance", 0.0))
        d.aperture_size = float(c.get("aperture_size", 0.0))
        return d

    # ---- persistence (fixtures under scenes/) ----------------------------------------------
    _ARRAYS = ("background", "ambient", "vert_pos", "vert_uv", "tri_vert", "tri_object", "sph_center_radius",
               "sph_object", "objects", "prim_shape", "materials", "lights", "textures", "texels")

    def save(self, path):
        self._contig()
        cams = np.zeros((len(self.cameras), 20), np.float64)
        for i, c in enumerate(self.cameras):
            cams[i, 0:3] = c["position"]
            cams[i, 3:6] = c["gaze"]
            cams[i, 6:9] = c["up"]
            cams[i, 9:13] = c["plane"]
            cams[i, 13] = c["distance"]
            cams[i, 14:16] = (c["nx"], c["ny"])
            cams[i, 16] = c.get("num_samples", 1)
            cams[i, 17] = c.get("focus_distance", 0.0)
            cams[i, 18] = c.get("aperture_size", 0.0)
        names = np.array([c.get("image_name", "") for c in self.cameras])
        extra = {"tonemap": np.asarray(self.tonemap, np.float64)} if self.tonemap is not None else {}
        np.savez_compressed(path, params=np.array([self.max_depth, self.bg_texture], np.int64), **extra,
                            eps=np.array([self.isect_eps, self.shadow_eps], F32), cameras=cams, camera_names=names,
                            **{k: getattr(self, k) for k in self._ARRAYS})

    def save_raw(self, path):
        """The flat scene file the C++ host reads (host/scene_file.h): header, the twelve description arrays, cameras."""
        import ctypes
        import struct
        self._contig()
        tm = self.tonemap
        with open(path, "wb") as f:
            f.write(b"RTSCENE1")
            f.write(struct.pack("<i2f3f3fii4f", int(self.max_depth), float(self.isect_eps), float(self.shadow_eps),
                                *[float(x) for x in self.background], *[float(x) for x in self.ambient], int(self.bg_texture),
                                1 if tm is not None else 0, *([float(x) for x in tm] if tm is not None else [0.0] * 4)))
            for arr, n in ((self.vert_pos, len(self.vert_pos)), (self.vert_uv, len(self.vert_uv)), (self.tri_vert, len(self.tri_vert)),
                           (self.tri_object, len(self.tri_object)), (self.sph_center_radius, len(self.sph_center_radius)),
                           (self.sph_object, len(self.sph_object)), (self.objects, len(self.objects)), (self.prim_shape, len(self.prim_shape)),
                           (self.materials, len(self.materials)), (self.lights, len(self.lights)), (self.textures, len(self.textures)),
                           (self.texels, len(self.texels))):
                f.write(struct.pack("<Q", n))
                f.write(arr.tobytes())
            f.write(struct.pack("<Q", len(self.cameras)))
            for i, c in enumerate(self.cameras):
                name = c.get("image_name", "out.ppm").encode()
                f.write(ctypes.string_at(ctypes.byref(self.camera_desc(i)), ctypes.sizeof(abi.CameraDesc)))
                f.write(struct.pack("<I", len(name)))
                f.write(name)

    @staticmethod
    def load(path):
        z = np.load(path, allow_pickle=False)
        s = SceneSoup()
        s.max_depth, s.bg_texture = int(z["params"][0]), int(z["params"][1])
        s.isect_eps, s.shadow_eps = F32(z["eps"][0]), F32(z["eps"][1])
        for k in SceneSoup._ARRAYS:
            setattr(s, k, z[k])
        s.objects = s.objects.astype(abi.object_dtype)
        s.materials = s.materials.astype(abi.material_dtype)
        s.lights = s.lights.astype(abi.light_dtype)
        s.textures = s.textures.astype(abi.texture_dtype)
        for row, nm in zip(z["cameras"], z["camera_names"]):
            s.cameras.append({"position": row[0:3].astype(F32), "gaze": row[3:6].astype(F32), "up": row[6:9].astype(F32),
                              "plane": row[9:13].astype(F32), "distance": F32(row[13]), "nx": int(row[14]), "ny": int(row[15]),
                              "num_samples": int(row[16]), "focus_distance": F32(row[17]), "aperture_size": F32(row[18]),
                              "image_name": str(nm)})
        if "tonemap" in z.files:
            s.tonemap = tuple(float(x) for x in z["tonemap"])
        s.name = os.path.splitext(os.path.basename(path))[0]
        s._contig()
        return s


# --------------------------------------------------------------------------------------------
# Incremental builder used by the XML loader and by the synthetic scenes
# --------------------------------------------------------------------------------------------
class SceneBuilder:
    def __init__(self, compact=False):
        """compact: <MeshInstance> clones and the primitive list are handed to the library as rt_instance / rt_prim_run records
        (expanded there, on the device for large scenes) instead of being written out here."""
        self.compact = bool(compact)
        self._runs, self._inst, self._virtual = [], [], []
        self.soup = SceneSoup()
        self._vp, self._vuv = [], []
        self._tv, self._to = [], []
        self._sph, self._so = [], []
        self._obj, self._prims = [], []
        self._mats, self._lights, self._tex, self._texels = [], [], [], []
        self._nv = 0
        self._nt = 0

    # materials / lights / textures -----------------------------------------------------------
    def add_material(self, type=abi.RT_MAT_DEFAULT, ka=(0, 0, 0), kd=(0, 0, 0), ks=(0, 0, 0), km=(0, 0, 0), ac=(0, 0, 0),
                     refraction_index=1.0, absorption_index=1.0, phong=0, roughness=0.0, degamma=False,
                     brdf_kind=abi.RT_BRDF_BLINNPHONG_ORIGINAL, brdf_flag=False, brdf_exponent=None):
        m = np.zeros((), abi.material_dtype)
        m["type"] = type
        m["brdf_kind"] = brdf_kind
        m["brdf_flag"] = int(brdf_flag)
        m["degamma"] = int(degamma)
        m["brdf_exponent"] = F32(phong if brdf_exponent is None else brdf_exponent)
        m["refraction_index"], m["absorption_index"], m["roughness"] = refraction_index, absorption_index, roughness
        m["ka"], m["kd"], m["ks"], m["km"], m["ac"] = ka, kd, ks, km, ac
        self._mats.append(m)
        return len(self._mats) - 1

    def add_light(self, type, position=(0, 0, 0), intensity=(0, 0, 0), direction=(0, 0, 0), coverage=0.0, falloff=0.0, size=0.0):
        l = np.zeros((), abi.light_dtype)
        l["type"] = type
        l["position"], l["intensity"], l["direction"] = position, intensity, direction
        l["coverage_deg"], l["falloff_deg"], l["size"] = coverage, falloff, size
        self._lights.append(l)
        return len(self._lights) - 1

    def add_texture(self, kind, decal, bump_factor=1.0, noise_scale=1.0, noise_conversion=abi.RT_NOISE_LINEAR, normalizer=255,
                    image=None, bilinear=False, perm=None):
        t = np.zeros((), abi.texture_dtype)
        t["kind"], t["decal"], t["bump_factor"], t["noise_scale"] = kind, decal, bump_factor, noise_scale
        t["noise_conversion"], t["normalizer"], t["bilinear"] = noise_conversion, normalizer, int(bilinear)
        if kind == abi.RT_TEX_IMAGE:
            img = np.ascontiguousarray(image)
            assert img.ndim == 3 and img.shape[2] == 3
            t["height"], t["width"] = img.shape[0], img.shape[1]
            t["texel_is_float"] = int(img.dtype != np.uint8)
            raw = img.astype(F32 if img.dtype != np.uint8 else np.uint8).tobytes()
            off = sum(len(b) for b in self._texels)
            pad = (-off) % 16
            self._texels.append(b"\0" * pad)
            t["texel_offset"] = off + pad
            self._texels.append(raw)
        else:
            t["perm"] = np.arange(16, dtype=np.int32) if perm is None else np.asarray(perm, np.int32)
        self._tex.append(t)
        return len(self._tex) - 1

    # geometry --------------------------------------------------------------------------------
    def add_object(self, xform=None, flags=abi.RT_OBJ_HAS_XFORM, material=0, tex_color=-1, tex_normal=-1, motion_blur=(0, 0, 0)):
        o = np.zeros((), abi.object_dtype)
        o["xform"] = (mat_identity() if xform is None else np.asarray(xform, F32)).reshape(16)
        o["motion_blur"] = motion_blur
        o["flags"], o["material"], o["tex_color"], o["tex_normal"] = flags, material, tex_color, tex_normal
        self._obj.append(o)
        return len(self._obj) - 1

    def add_vertices(self, pos, uv=None):
        pos = np.asarray(pos, F32).reshape(-1, 3)
        uv = np.zeros((len(pos), 2), F32) if uv is None else np.asarray(uv, F32).reshape(-1, 2)
        base = self._nv
        self._vp.append(pos)
        self._vuv.append(uv)
        self._nv += len(pos)
        return base

    def add_triangles(self, vert_idx, obj):
        vert_idx = np.asarray(vert_idx, np.int32).reshape(-1, 3)
        base = self._nt
        self._tv.append(vert_idx)
        self._to.append(np.full(len(vert_idx), obj, np.int32))
        self._nt += len(vert_idx)
        return base

    def push_prims(self, tri_base, count, blurbox=False):
        if self.compact:
            self._runs.append((tri_base, count, abi.RT_PRIM_BLURBOX if blurbox else 0))
            return
        ids = np.arange(tri_base, tri_base + count, dtype=np.int32)
        if blurbox:
            ids = ids | np.int32(abi.RT_PRIM_BLURBOX)
        self._prims.append(ids)

    def add_sphere(self, center, radius, xform=None, material=0, tex_color=-1, tex_normal=-1):
        """Sphere ctor + Primitive push (SceneParser.cpp:787-790)."""
        obj = self.add_object(xform, abi.RT_OBJ_HAS_XFORM, material, tex_color, tex_normal)
        self._sph.append(np.array([center[0], center[1], center[2], radius], F32))
        self._so.append(obj)
        if self.compact:
            self._runs.append((len(self._sph) - 1, 1, abi.RT_PRIM_SPHERE))
        else:
            self._prims.append(np.array([(len(self._sph) - 1) | abi.RT_PRIM_SPHERE], np.int32))
        return obj

    def add_triangle(self, p0, p1, p2, uv=None, material=0, tex_color=-1, tex_normal=-1):
        """Standalone <Triangle>: objTransform == nullptr (SceneParser.cpp:859-861)."""
        obj = self.add_object(None, 0, material, tex_color, tex_normal)
        vb = self.add_vertices([p0, p1, p2], uv)
        tb = self.add_triangles([[vb, vb + 1, vb + 2]], obj)
        self.push_prims(tb, 1)
        return obj

    def add_mesh(self, pos, faces, uv=None, xform=None, material=0, smooth=False, motion_blur=(0, 0, 0), tex_color=-1, tex_normal=-1):
        """<Mesh>: Mesh ctor pushes one Primitive per triangle (Mesh.cpp:57-58) and Mesh::SetMotionBlur, run for
        every mesh, pushes them all again with the (possibly motion-extended) bbox (Mesh.cpp:182-192) — Q5.
        `pos`/`uv` are the mesh's own Vertex objects (already de-duplicated by index, Q28).  Returns a handle usable
        with add_instance."""
        flags = abi.RT_OBJ_HAS_XFORM | abi.RT_OBJ_COMPOSITE | (abi.RT_OBJ_SMOOTH if smooth else 0)
        obj = self.add_object(xform, flags, material, tex_color, tex_normal, motion_blur)
        vb = self.add_vertices(pos, uv)
        faces = np.asarray(faces, np.int64).reshape(-1, 3) + vb
        tb = self.add_triangles(faces, obj)
        self.push_prims(tb, len(faces), blurbox=False)
        self.push_prims(tb, len(faces), blurbox=True)
        return {"obj": obj, "faces": faces.astype(np.int32), "xform": self._obj[obj]["xform"].reshape(4, 4).copy(),
                "material": material, "tex_color": tex_color, "tex_normal": tex_normal, "tri_base": tb}

    def add_instance(self, base, xform=None, reset_transform=False, material=None, motion_blur=(0, 0, 0)):
        """<MeshInstance> (SceneParser.cpp:1075-1129, Mesh.cpp:126-180): clones share the base Vertex objects, the
        instance matrix is composed after the (kept or reset) base matrix, shading mode is not copied (Q27: flat),
        triangles are pushed once (by SetMotionBlur)."""
        base_m = mat_identity() if reset_transform else base["xform"]
        inst = mat_identity() if xform is None else np.asarray(xform, F32)
        m = mat_mul(inst, base_m)                                  # Shape::SetTransformation, Shape.cpp:29-39
        mat = base["material"] if material is None else material
        obj = self.add_object(m, abi.RT_OBJ_HAS_XFORM | abi.RT_OBJ_COMPOSITE, mat, base["tex_color"], base["tex_normal"], motion_blur)
        if self.compact:
            # the library clones the triangles (rt_instance): virtual triangle indices follow the explicit ones in instance order
            assert "tri_base" in base, "instance of an instance: give the original mesh"
            self._inst.append((base["tri_base"], len(base["faces"]), obj))
            self._virtual.append(len(base["faces"]))
            tb = ("virtual", len(self._inst) - 1)
        else:
            tb = self.add_triangles(base["faces"], obj)
        self.push_prims(tb, len(base["faces"]), blurbox=True)
        return {"obj": obj, "faces": base["faces"], "xform": m, "material": mat, "tex_color": base["tex_color"],
                "tex_normal": base["tex_normal"]}

    def add_camera(self, position, gaze, up, plane, distance, nx, ny, num_samples=1, focus_distance=0.0, aperture_size=0.0, image_name="out.ppm"):
        self.soup.cameras.append({"position": np.asarray(position, F32), "gaze": np.asarray(gaze, F32), "up": np.asarray(up, F32),
                                  "plane": np.asarray(plane, F32), "distance": F32(distance), "nx": int(nx), "ny": int(ny),
                                  "num_samples": int(num_samples), "focus_distance": F32(focus_distance),
                                  "aperture_size": F32(aperture_size), "image_name": image_name})

    def add_lookat_camera(self, position, gaze_point, up, fov_y, distance, nx, ny, **kw):
        """Camera::DeriveBoundaries (src/Camera.cpp:161-172): PI is the double macro 3.14159265 (src/Shape.h:9)."""
        position = np.asarray(position, F32)
        gaze = (np.asarray(gaze_point, F32) - position).astype(F32)
        gaze = (gaze / F32(np.sqrt(F32(F32(gaze[0] * gaze[0] + gaze[1] * gaze[1]) + gaze[2] * gaze[2])))).astype(F32)
        half = F32(F32(fov_y) / F32(2))
        t = F32(float(F32(distance)) * math.tan(float(half) * (3.14159265 / float(F32(180.0)))))
        a = F32(F32(nx) / F32(ny))
        r = F32(a * t)
        self.add_camera(position, gaze, up, (-r, r, -t, t), distance, nx, ny, **kw)

    def finish(self, name=""):
        s = self.soup
        cat = lambda lst, shape, dt: (np.concatenate(lst).astype(dt) if lst else np.zeros(shape, dt))
        s.vert_pos = cat(self._vp, (0, 3), F32)
        s.vert_uv = cat(self._vuv, (0, 2), F32)
        s.tri_vert = cat(self._tv, (0, 3), np.int32)
        s.tri_object = cat(self._to, (0,), np.int32)
        s.sph_center_radius = np.array(self._sph, F32).reshape(-1, 4)
        s.sph_object = np.array(self._so, np.int32)
        s.objects = np.array(self._obj, abi.object_dtype) if self._obj else np.zeros(0, abi.object_dtype)
        s.prim_shape = cat(self._prims, (0,), np.int32)
        if self.compact:
            n_explicit = len(s.tri_vert)
            vbase = np.concatenate([[0], np.cumsum(self._virtual)]).astype(np.int64) + n_explicit
            runs = [((int(vbase[f[1]]) if isinstance(f, tuple) else int(f)), int(c), int(fl)) for f, c, fl in self._runs]
            s.prim_runs = np.array([(a, c, fl, 0) for a, c, fl in runs], abi.prim_run_dtype) if runs else np.zeros(0, abi.prim_run_dtype)
            s.instances = np.array([(a, n, o, 0) for a, n, o in self._inst], abi.instance_dtype) if self._inst else np.zeros(0, abi.instance_dtype)
        s.materials = np.array(self._mats, abi.material_dtype) if self._mats else np.zeros(0, abi.material_dtype)
        s.lights = np.array(self._lights, abi.light_dtype) if self._lights else np.zeros(0, abi.light_dtype)
        s.textures = np.array(self._tex, abi.texture_dtype) if self._tex else np.zeros(0, abi.texture_dtype)
        s.texels = np.frombuffer(b"".join(self._texels), np.uint8).copy() if self._texels else np.zeros(0, np.uint8)
        s.name = name
        s._contig()
        return s


# --------------------------------------------------------------------------------------------
# XML loader (reference scene format)
# --------------------------------------------------------------------------------------------
def _floats(text):
    return [float(x) for x in text.split()]


_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
              "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def _ply_header(f):
    """Reads the text header of an open PLY file; returns (format, [element dicts in file order])."""
    header = []
    while True:
        raw = f.readline()
        if not raw:
            raise ValueError("PLY header without end_header")
        line = raw.decode("ascii", "replace").strip()
        header.append(line)
        if line == "end_header":
            break
    fmt = [h.split()[1] for h in header if h.startswith("format")][0]
    elements, cur = [], None
    for h in header:
        p = h.split()
        if p and p[0] == "element":
            cur = {"name": p[1], "count": int(p[2]), "props": []}
            elements.append(cur)
        elif p and p[0] == "property" and cur is not None:
            cur["props"].append(p[1:])
    return fmt, elements


def ply_layout(elements, ascii=False):
    """abi.PlyLayout of a PLY body for rt_ply_decode, or None when the file needs the general reader (elements other
    than vertex + face, list properties in the vertex element, more than one list in the face element, exotic types).
    ascii: the body is text; offsets and sizes then count tokens instead of bytes."""
    if [e["name"] for e in elements] != ["vertex", "face"]:
        return None
    ve, fe = elements
    L = abi.PlyLayout()
    L.abi_version = abi.RT_ABI_VERSION
    L.format = abi.RT_PLY_ASCII if ascii else abi.RT_PLY_BINARY_LE
    size = (lambda t: 1) if ascii else (lambda t: np.dtype(_PLY_TYPES[t]).itemsize)
    off, where = 0, {}
    for p in ve["props"]:
        if p[0] == "list" or p[0] not in _PLY_TYPES:
            return None
        where[p[-1]] = (off, _PLY_TYPES[p[0]])
        off += size(p[0])
    if not all(k in where for k in "xyz") or len({where[k][1] for k in "xyz"}) != 1 or where["x"][1] not in ("f4", "f8"):
        return None
    L.pos_type = 1 if where["x"][1] == "f8" else 0
    L.n_vertices, L.vertex_stride = ve["count"], off
    for k, name in enumerate("xyz"):
        L.pos_offset[k] = where[name][0]
    pre = post = 0
    lst = None
    for p in fe["props"]:
        if p[0] == "list":
            if lst is not None or p[-1] not in ("vertex_indices", "vertex_index"):
                return None
            lst = p
        elif p[0] not in _PLY_TYPES:
            return None
        elif lst is None:
            pre += size(p[0])
        else:
            post += size(p[0])
    if lst is None or lst[1] not in _PLY_TYPES or lst[2] not in _PLY_TYPES or _PLY_TYPES[lst[1]][0] == "f" or _PLY_TYPES[lst[2]][0] == "f":
        return None
    L.count_size, L.index_size = np.dtype(_PLY_TYPES[lst[1]]).itemsize, np.dtype(_PLY_TYPES[lst[2]]).itemsize
    if L.count_size == 8 or L.index_size == 8:
        return None
    L.n_faces, L.face_pre_bytes, L.face_post_bytes = fe["count"], pre, post
    return L


def _round_text_to_f32(vals64, token_of):
    """float32 values of decimal tokens as a direct text -> float conversion gives them (a float property is extracted as a float
    by happly, src/happly.h:2075-2095).  vals64: the tokens as correctly rounded doubles.  Rounding the double again is the same
    except where the double lies exactly halfway between two floats or in the subnormal float range; those few are decided exactly."""
    out = vals64.astype(F32)
    bits = vals64.view(np.uint64)
    mag = np.abs(vals64)
    suspects = np.nonzero((((bits & np.uint64(0x1fffffff)) == np.uint64(0x10000000)) | ((mag < 1.1754943508222875e-38) & (mag > 0))))
    if len(suspects[0]):
        from fractions import Fraction
        for i, k in zip(*suspects):
            tok = token_of(int(i), int(k))
            exact = Fraction(tok.decode() if isinstance(tok, bytes) else tok)
            c = np.float32(vals64[i, k])
            cands = [c, np.nextafter(c, np.float32(np.inf)), np.nextafter(c, np.float32(-np.inf))]
            best = min(cands, key=lambda f: (abs(Fraction(float(f)) - exact), int(np.float32(f).view(np.uint32)) & 1))
            out[i, k] = best
    return out


def _read_ply(path, decoder=None):
    """PLY reader (ascii / binary_little_endian; vertex x,y,z + face vertex_indices lists) standing in for happly
    (src/SceneParser.cpp:983-985).  Quads become (0,1,2),(2,3,0) as in SceneParser.cpp:995-1011, faces of any other size are
    dropped.  `decoder` (Context.decode_ply) decodes a binary body on the GPU; without it numpy does."""
    with open(path, "rb") as f:
        fmt, elements = _ply_header(f)
        tmap = _PLY_TYPES
        body_at = f.tell()
        # The device decoder takes the layouts real files have; a body it refuses (a face line it cannot parse, ...) goes to the
        # general reader below, which reads whatever happly reads.
        try:
            if fmt == "binary_little_endian" and decoder is not None:
                L = ply_layout(elements)
                if L is not None:
                    body = f.read()
                    vbytes = int(L.n_vertices) * int(L.vertex_stride)
                    pos, tris = decoder(memoryview(body)[:vbytes], memoryview(body)[vbytes:], L)
                    return pos, tris.astype(np.int64)
            if fmt == "ascii" and decoder is not None:
                L = ply_layout(elements, ascii=True)
                if L is not None:
                    pos, tris = decoder(memoryview(f.read()), b"", L)
                    return pos, tris.astype(np.int64)
        except Exception as e:
            if type(e).__name__ != "RtError":
                raise
        f.seek(body_at)
        verts, faces = None, []
        quads_tris = None
        if fmt == "ascii":
            tokens = f.read().split()
            pos = 0
            for el in elements:
                if el["name"] == "vertex":
                    n = len(el["props"])
                    names = [p[-1] for p in el["props"]]
                    arr = np.array(tokens[pos:pos + n * el["count"]], dtype=np.float64).reshape(el["count"], n)
                    cols = [names.index("x"), names.index("y"), names.index("z")]
                    verts = arr[:, cols]
                    if all(tmap.get(el["props"][c][0]) == "f4" for c in cols):
                        verts = _round_text_to_f32(verts, lambda i, k, pos=pos: tokens[pos + i * n + cols[k]])
                    pos += n * el["count"]
                elif el["name"] == "face":
                    for _ in range(el["count"]):
                        for p in el["props"]:
                            if p[0] != "list":
                                pos += 1
                                continue
                            k = int(tokens[pos])
                            if p[-1] in ("vertex_indices", "vertex_index"):
                                faces.append([int(x) for x in tokens[pos + 1:pos + 1 + k]])
                            pos += 1 + k
                else:
                    for _ in range(el["count"]):
                        for p in el["props"]:
                            if p[0] == "list":
                                k = int(tokens[pos]); pos += 1 + k
                            else:
                                pos += 1
        else:
            assert fmt in ("binary_little_endian", "binary_big_endian"), fmt
            E = "<" if fmt == "binary_little_endian" else ">"           # happly reads both byte orders
            for el in elements:
                if el["name"] == "vertex":
                    dt = np.dtype([(p[-1], E + tmap[p[0]]) for p in el["props"]])
                    arr = np.frombuffer(f.read(dt.itemsize * el["count"]), dt)
                    verts = np.stack([arr["x"], arr["y"], arr["z"]], 1).astype(np.float64)
                elif el["name"] == "face":
                    lists = [p for p in el["props"] if p[0] == "list"]
                    p = lists[0]
                    k_list = el["props"].index(p)
                    pre = sum(np.dtype(tmap[q[0]]).itemsize for q in el["props"][:k_list])
                    post = sum(np.dtype(tmap[q[0]]).itemsize for q in el["props"][k_list + 1:]) if len(lists) == 1 else None
                    cdt, idt = np.dtype(E + tmap[p[1]]), np.dtype(E + tmap[p[2]])
                    buf = f.read()
                    # fast path: all faces have the same arity (and no second list)
                    if post is not None and el["count"] > 0 and len(buf) >= pre + cdt.itemsize:
                        k0 = int(np.frombuffer(buf, cdt, 1, pre)[0])
                        rdt = np.dtype([("a", "u1", pre), ("k", cdt), ("i", idt, k0), ("b", "u1", post)])
                        if len(buf) >= rdt.itemsize * el["count"]:
                            arr = np.frombuffer(buf, rdt, el["count"], 0)
                            if np.all(arr["k"] == k0):
                                idx = arr["i"].astype(np.int64).reshape(el["count"], k0)
                                if k0 == 3:
                                    quads_tris = idx
                                elif k0 == 4:
                                    quads_tris = np.stack([idx[:, [0, 1, 2]], idx[:, [2, 3, 0]]], 1).reshape(-1, 3)
                                else:
                                    quads_tris = np.zeros((0, 3), np.int64)
                                continue
                    assert post is not None, "face element with several list properties"
                    pos = 0
                    for _ in range(el["count"]):
                        pos += pre
                        k = int(np.frombuffer(buf, cdt, 1, pos)[0]); pos += cdt.itemsize
                        faces.append(np.frombuffer(buf, idt, k, pos).astype(np.int64).tolist()); pos += k * idt.itemsize + post
                else:
                    for _ in range(el["count"]):
                        for p in el["props"]:
                            if p[0] == "list":
                                k = int(np.frombuffer(f.read(np.dtype(tmap[p[1]]).itemsize), E + tmap[p[1]])[0])
                                f.read(k * np.dtype(tmap[p[2]]).itemsize)
                            else:
                                f.read(np.dtype(tmap[p[0]]).itemsize)
        if quads_tris is not None:
            return verts.astype(F32), quads_tris
        tris = []
        for v in faces:
            if len(v) == 4:
                tris.append((v[0], v[1], v[2]))
                tris.append((v[2], v[3], v[0]))
            elif len(v) == 3:
                tris.append(tuple(v))
        return verts.astype(F32), np.array(tris, np.int64).reshape(-1, 3)


def _dedup_mesh(indices):
    """Mesh.cpp:24-53: Vertex objects are created per distinct scene-vertex pointer, in first-use order."""
    flat = np.asarray(indices, np.int64).reshape(-1)
    uniq, first = np.unique(flat, return_index=True)
    order = np.argsort(first, kind="stable")
    uniq = uniq[order]
    remap = np.empty(flat.max() + 1 if len(flat) else 0, np.int64)
    remap[uniq] = np.arange(len(uniq))
    return uniq, remap[flat].reshape(-1, 3)


def load_xml(path, scene_dir=None, texture_loader=None, perlin_tables=None, ply_decoder=None):
    """Parses a reference scene file.  `scene_dir` is the directory PLY paths are relative to ("scenes/" in the
    reference, SceneParser.cpp:977).  `texture_loader(path)` must return an HxWx3 uint8 (or float32 for .exr) array;
    default uses Pillow.  `perlin_tables(n)` returns n shuffled 16-entry tables (rt_perlin_tables of the library)."""
    root = ET.parse(path).getroot()
    scene_dir = scene_dir or os.path.dirname(os.path.abspath(path))
    b = SceneBuilder()
    s = b.soup

    def txt(el, tag):
        e = el.find(tag)
        return None if e is None or e.text is None else e.text

    v = txt(root, "MaxRecursionDepth")
    if v is not None:
        s.max_depth = int(v)
    v = txt(root, "BackgroundColor")
    if v is not None:
        s.background = np.array(_floats(v)[:3], F32)
    v = txt(root, "ShadowRayEpsilon")
    if v is not None:
        s.shadow_eps = F32(float(v))
    # "intersectionTestEpsilon" is looked up with a lower-case i (SceneParser.cpp:77): scenes never match -> 1e-4.
    v = txt(root, "intersectionTestEpsilon")
    if v is not None:
        s.isect_eps = F32(float(v))

    for cam in root.find("Cameras").findall("Camera"):
        pos = _floats(txt(cam, "Position"))
        up = _floats(txt(cam, "Up"))
        ns = int(txt(cam, "NumSamples") or 1)
        fd = float(txt(cam, "FocusDistance") or 0.0)
        ap = float(txt(cam, "ApertureSize") or 0.0)
        dist = float(txt(cam, "NearDistance"))
        nx, ny = (int(x) for x in txt(cam, "ImageResolution").split())
        name = "".join(txt(cam, "ImageName").split())
        kw = dict(num_samples=ns, focus_distance=fd, aperture_size=ap, image_name=name)
        tm = cam.find("Tonemap")
        if tm is not None:                       # SceneParser.cpp:162-181: one scene-wide Tonemapper, the last camera's wins
            key, burn = (float(x) for x in txt(tm, "TMOOptions").split()[:2])
            s.tonemap = (key, burn, float(txt(tm, "Saturation")), float(txt(tm, "Gamma")))
        if cam.get("type") == "lookAt":
            b.add_lookat_camera(pos, _floats(txt(cam, "GazePoint")), up, float(txt(cam, "FovY")), dist, nx, ny, **kw)
        else:
            b.add_camera(pos, _floats(txt(cam, "Gaze"))[:3], up[:3], _floats(txt(cam, "NearPlane"))[:4], dist, nx, ny, **kw)

    # BRDFs: pushed category by category, looked up by push position (SceneParser.cpp:188-278, :380)
    brdfs = []
    be = root.find("BRDFs")
    if be is not None:
        def tobool(x):
            return str(x).lower() in ("true", "1")
        for tag, kind, flagname in (("ModifiedBlinnPhong", abi.RT_BRDF_BLINNPHONG_MODIFIED, "normalized"),
                                    ("OriginalPhong", abi.RT_BRDF_PHONG_ORIGINAL, None),
                                    ("OriginalBlinnPhong", abi.RT_BRDF_BLINNPHONG_ORIGINAL, None),
                                    ("ModifiedPhong", abi.RT_BRDF_PHONG_MODIFIED, "normalized"),
                                    ("TorranceSparrow", abi.RT_BRDF_TORRANCE_SPARROW, "kdfresnel")):
            for e in be.findall(tag):
                ex = txt(e, "Exponent")
                brdfs.append((kind, tobool(e.get(flagname, "false")) if flagname else False, float(ex) if ex is not None else 1.0))

    def vec3(el, tag, default=(0.0, 0.0, 0.0)):
        t = txt(el, tag)
        return tuple(_floats(t)[:3]) if t is not None else default

    for m in root.find("Materials").findall("Material"):
        typ = {"mirror": abi.RT_MAT_MIRROR, "dielectric": abi.RT_MAT_DIELECTRIC, "conductor": abi.RT_MAT_CONDUCTOR}.get(m.get("type"), abi.RT_MAT_DEFAULT)
        degamma = str(m.get("degamma", "false")).lower() in ("true", "1")
        ka, kd, ks = (np.array(vec3(m, t), F32) for t in ("AmbientReflectance", "DiffuseReflectance", "SpecularReflectance"))
        phong_t = txt(m, "PhongExponent")
        phong = int(float(phong_t)) if phong_t is not None else 0           # QueryIntText
        if degamma:                                                         # SceneParser.cpp:386-399 (powf)
            ka, kd, ks = (np.power(x, F32(2.2)).astype(F32) for x in (ka, kd, ks))
        brdf_id = int(m.get("BRDF", -1))
        if brdf_id != -1:
            kind, flag, ex = brdfs[brdf_id - 1]
        else:
            kind, flag, ex = abi.RT_BRDF_BLINNPHONG_ORIGINAL, False, float(phong)
        b.add_material(typ, ka, kd, ks, vec3(m, "MirrorReflectance"), vec3(m, "AbsorptionCoefficient"),
                       float(txt(m, "RefractionIndex") or 1.0), float(txt(m, "AbsorptionIndex") or 1.0), phong,
                       float(txt(m, "Roughness") or 0.0), degamma, kind, flag, ex)

    # textures (SceneParser.cpp:440-595)
    te = root.find("Textures")
    tex_index = []          # scene->textures push order -> builder texture index
    if te is not None:
        images = []
        ie = te.find("Images")
        if ie is not None:
            images = [e.text.strip() for e in ie.findall("Image") if e.text]
        maps = te.findall("TextureMap")
        n_perlin = sum(1 for e in maps if e.get("type") == "perlin")
        tables = perlin_tables(n_perlin) if (perlin_tables and n_perlin) else [None] * n_perlin
        pi = 0
        for e in maps:
            decal = {"bump_normal": abi.RT_DECAL_BUMP_NORMAL, "replace_kd": abi.RT_DECAL_REPLACE_KD,
                     "replace_normal": abi.RT_DECAL_REPLACE_NORMAL, "replace_background": abi.RT_DECAL_REPLACE_BACKGROUND,
                     "replace_all": abi.RT_DECAL_REPLACE_ALL, "blend_kd": abi.RT_DECAL_BLEND_KD}[txt(e, "DecalMode").strip()]
            bump = float(txt(e, "BumpFactor") or 1.0)
            if e.get("type") == "perlin":
                conv = abi.RT_NOISE_ABSVAL if (txt(e, "NoiseConversion") or "").strip() == "absval" else abi.RT_NOISE_LINEAR
                tex_index.append(b.add_texture(abi.RT_TEX_PERLIN, decal, bump, float(txt(e, "NoiseScale") or 1.0), conv, perm=tables[pi]))
                pi += 1
            elif e.get("type") == "image":
                ipath = images[int(txt(e, "ImageId")) - 1]
                full = os.path.join(os.path.dirname(scene_dir.rstrip("/")), ipath) if not os.path.isabs(ipath) else ipath
                if texture_loader is None:
                    from PIL import Image
                    img = np.asarray(Image.open(full).convert("RGB"), np.uint8)
                else:
                    img = texture_loader(full)
                # Interpolation: the parser's `itype == ...` is a comparison, so it is always NEAREST (Q23)
                tex_index.append(b.add_texture(abi.RT_TEX_IMAGE, decal, bump, normalizer=int(txt(e, "Normalizer") or 255), image=img))
            else:
                continue
            if decal == abi.RT_DECAL_REPLACE_BACKGROUND:
                s.bg_texture = tex_index[-1]

    def resolve_textures(el):
        """<Textures> lists single-character ids (SceneParser.cpp:755-785): normal changers vs colour changers."""
        tc = tn = -1
        t = txt(el, "Textures")
        if t is None:
            return tc, tn
        for ch in t:
            if ch in " \t\n":
                continue
            ti = tex_index[ord(ch) - 49]
            decal = int(b._tex[ti]["decal"])
            if decal in (abi.RT_DECAL_BUMP_NORMAL, abi.RT_DECAL_REPLACE_NORMAL):
                tn = ti
            elif decal != abi.RT_DECAL_REPLACE_BACKGROUND:
                tc = ti
        return tc, tn

    # transformations (SceneParser.cpp:597-650); ids are single characters (Q24)
    xf = {"s": {}, "r": {}, "t": {}}
    tr = root.find("Transformations")
    if tr is not None:
        for e in tr.findall("Scaling"):
            xf["s"].setdefault(int(e.get("id")), mat_scale(_floats(e.text)[:3]))
        for e in tr.findall("Rotation"):
            f = _floats(e.text)
            xf["r"].setdefault(int(e.get("id")), mat_rotate(f[0], f[1:4]))
        for e in tr.findall("Translation"):
            xf["t"].setdefault(int(e.get("id")), mat_translate(_floats(e.text)[:3]))

    def compose(el):
        """ComputeTransformMatrix (SceneParser.cpp:1249-1304): M := T_k * M for each listed transform, left to right."""
        m = mat_identity()
        t = txt(el, "Transformations")
        if t is None:
            return m
        i = 0
        while i < len(t):
            if t[i] in "rts" and i + 1 < len(t):
                tid = ord(t[i + 1]) - 48
                if tid in xf[t[i]]:
                    m = mat_mul(xf[t[i]][tid], m)
            i += 1
        return m

    vd = txt(root, "VertexData")
    vertices = np.array(_floats(vd), np.float64).reshape(-1, 3).astype(F32) if vd else np.zeros((0, 3), F32)
    tcd = txt(root, "TexCoordData")
    if tcd is not None:
        tc = np.array(_floats(tcd), np.float64).reshape(-1, 2).astype(F32)
        coords = np.zeros((max(len(vertices), len(tc)), 2), F32)
        coords[:len(tc)] = tc
        single_coord = False
    else:
        coords = np.zeros((len(vertices) + 1, 2), F32)
        single_coord = len(vertices) == 0           # vertexCoords.size() == 1 only when there is no VertexData
    objs = root.find("Objects")
    mat_of = lambda el: int(txt(el, "Material")) - 1

    for e in objs.findall("Sphere"):
        tcx, tnx = resolve_textures(e)
        b.add_sphere(vertices[int(txt(e, "Center")) - 1], float(txt(e, "Radius")), compose(e), mat_of(e), tcx, tnx)

    for e in objs.findall("Triangle"):
        i0, i1, i2 = (int(x) - 1 for x in txt(e, "Indices").split())
        tcx, tnx = resolve_textures(e)
        b.add_triangle(vertices[i0], vertices[i1], vertices[i2], coords[[i0, i1, i2]], mat_of(e), tcx, tnx)

    meshes = {}
    for e in objs.findall("Mesh"):
        fe = e.find("Faces")
        voff = int(fe.get("vertexOffset", 0))
        toff = int(fe.get("textureOffset", 0))
        mb = tuple(_floats(txt(e, "MotionBlur"))[:3]) if txt(e, "MotionBlur") else (0.0, 0.0, 0.0)
        smooth = e.get("shadingMode") == "smooth"
        if fe.get("plyFile"):
            pos, faces = _read_ply(os.path.join(scene_dir, fe.get("plyFile")), ply_decoder)
            used, faces = _dedup_mesh(faces)
            mpos, muv = pos[used], np.zeros((len(used), 2), F32)
        else:
            idx = np.array([int(x) for x in fe.text.split()], np.int64).reshape(-1, 3) - 1 + voff
            used, faces = _dedup_mesh(idx)
            mpos = vertices[used]
            muv = np.zeros((len(used), 2), F32) if single_coord else coords[used - voff + toff]
        tcx, tnx = resolve_textures(e)
        h = b.add_mesh(mpos, faces, muv, compose(e), mat_of(e), smooth, mb, tcx, tnx)
        meshes.setdefault(int(e.get("id")), h)

    for e in objs.findall("MeshInstance"):
        base = meshes[int(e.get("baseMeshId"))]
        reset = str(e.get("resetTransform", "false")).lower() in ("true", "1")
        mb = tuple(_floats(txt(e, "MotionBlur"))[:3]) if txt(e, "MotionBlur") else (0.0, 0.0, 0.0)
        mat = mat_of(e) if txt(e, "Material") is not None else None
        b.add_instance(base, compose(e), reset, mat, mb)

    le = root.find("Lights")
    v = txt(le, "AmbientLight")
    if v is not None:
        s.ambient = np.array(_floats(v)[:3], F32)
    for e in le.findall("PointLight"):
        b.add_light(abi.RT_LIGHT_POINT, vec3(e, "Position"), vec3(e, "Intensity"))
    for e in le.findall("DirectionalLight"):
        b.add_light(abi.RT_LIGHT_DIRECTIONAL, (0, 0, 0), vec3(e, "Radiance"), vec3(e, "Direction"))
    for e in le.findall("SpotLight"):
        b.add_light(abi.RT_LIGHT_SPOT, vec3(e, "Position"), vec3(e, "Intensity"), vec3(e, "Direction"),
                    float(txt(e, "CoverageAngle")), float(txt(e, "FalloffAngle")))
    for e in le.findall("AreaLight"):
        b.add_light(abi.RT_LIGHT_AREA, vec3(e, "Position"), vec3(e, "Radiance"), vec3(e, "Normal"), size=float(txt(e, "Size")))
    return b.finish(os.path.splitext(os.path.basename(path))[0])
