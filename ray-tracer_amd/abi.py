"""ctypes / numpy mirror of include/rt_mi355.h (the C-ABI of librt_mi355.so).

Layout only: no behaviour.  Every struct here must match the C header byte for byte; tests/test_abi.py
checks the sizes against the library's own sizeof() table.
"""
import ctypes as C

import numpy as np

RT_ABI_VERSION = 6

RT_OK, RT_ERR_INVALID, RT_ERR_ABI, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_NO_SCENE, RT_ERR_TOO_LARGE = 0, -1, -2, -3, -4, -5, -6

RT_MAT_DEFAULT, RT_MAT_MIRROR, RT_MAT_DIELECTRIC, RT_MAT_CONDUCTOR = 0, 1, 2, 3
(RT_BRDF_BLINNPHONG_ORIGINAL, RT_BRDF_BLINNPHONG_MODIFIED, RT_BRDF_PHONG_ORIGINAL, RT_BRDF_PHONG_MODIFIED,
 RT_BRDF_TORRANCE_SPARROW) = range(5)
RT_LIGHT_POINT, RT_LIGHT_DIRECTIONAL, RT_LIGHT_SPOT, RT_LIGHT_AREA = 0, 1, 2, 3
RT_TEX_IMAGE, RT_TEX_PERLIN = 0, 1
(RT_DECAL_REPLACE_KD, RT_DECAL_BLEND_KD, RT_DECAL_REPLACE_ALL, RT_DECAL_REPLACE_NORMAL, RT_DECAL_BUMP_NORMAL,
 RT_DECAL_REPLACE_BACKGROUND) = range(6)
RT_NOISE_LINEAR, RT_NOISE_ABSVAL = 0, 1
RT_OBJ_SMOOTH, RT_OBJ_HAS_XFORM, RT_OBJ_COMPOSITE = 1, 2, 4
RT_PRIM_SPHERE, RT_PRIM_BLURBOX, RT_PRIM_INDEX_MASK = 0x40000000, 0x20000000, 0x1FFFFFFF
RT_RAY_PRIMARY, RT_RAY_SHADOW, RT_RAY_REFLECT, RT_RAY_REFRACT, RT_RAY_KINDS = 0, 1, 2, 3, 4

material_dtype = np.dtype([
    ("type", "<i4"), ("brdf_kind", "<i4"), ("brdf_flag", "<i4"), ("degamma", "<i4"),
    ("brdf_exponent", "<f4"), ("refraction_index", "<f4"), ("absorption_index", "<f4"), ("roughness", "<f4"),
    ("ka", "<f4", 3), ("kd", "<f4", 3), ("ks", "<f4", 3), ("km", "<f4", 3), ("ac", "<f4", 3), ("_pad", "<f4")])
assert material_dtype.itemsize == 96

light_dtype = np.dtype([
    ("type", "<i4"), ("position", "<f4", 3), ("intensity", "<f4", 3), ("direction", "<f4", 3),
    ("coverage_deg", "<f4"), ("falloff_deg", "<f4"), ("size", "<f4"), ("_pad", "<f4", 3)])
assert light_dtype.itemsize == 64

texture_dtype = np.dtype([
    ("kind", "<i4"), ("decal", "<i4"), ("bump_factor", "<f4"), ("noise_scale", "<f4"), ("noise_conversion", "<i4"),
    ("normalizer", "<i4"), ("width", "<i4"), ("height", "<i4"), ("bilinear", "<i4"), ("texel_is_float", "<i4"),
    ("texel_offset", "<i8"), ("perm", "<i4", 16), ("_pad", "<i4", 2)])
assert texture_dtype.itemsize == 120

object_dtype = np.dtype([
    ("xform", "<f4", 16), ("motion_blur", "<f4", 3), ("flags", "<i4"), ("material", "<i4"),
    ("tex_color", "<i4"), ("tex_normal", "<i4"), ("_pad", "<i4")])
assert object_dtype.itemsize == 96

ray_dtype = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("time", "<f4")])
assert ray_dtype.itemsize == 28
hit_dtype = np.dtype([("t", "<f4"), ("shape", "<i4"), ("ip", "<f4", 3), ("n", "<f4", 3), ("uv", "<f4", 2)])
assert hit_dtype.itemsize == 40


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("max_depth", C.c_int32), ("isect_eps", C.c_float), ("shadow_eps", C.c_float),
        ("background", C.c_float * 3), ("ambient", C.c_float * 3), ("bg_texture", C.c_int32), ("_pad0", C.c_int32),
        ("n_vertices", C.c_int64), ("vert_pos", C.c_void_p), ("vert_uv", C.c_void_p),
        ("n_triangles", C.c_int64), ("tri_vert", C.c_void_p), ("tri_object", C.c_void_p),
        ("n_spheres", C.c_int64), ("sph_center_radius", C.c_void_p), ("sph_object", C.c_void_p),
        ("n_objects", C.c_int64), ("objects", C.c_void_p),
        ("n_prims", C.c_int64), ("prim_shape", C.c_void_p),
        ("n_materials", C.c_int32), ("n_lights", C.c_int32), ("n_textures", C.c_int32), ("_pad1", C.c_int32),
        ("materials", C.c_void_p), ("lights", C.c_void_p), ("textures", C.c_void_p),
        ("texels", C.c_void_p), ("texels_bytes", C.c_int64),
        ("n_instances", C.c_int64), ("instances", C.c_void_p), ("n_prim_runs", C.c_int64), ("prim_runs", C.c_void_p),
        ("geometry_on_device", C.c_int32), ("_pad2", C.c_int32)]


instance_dtype = np.dtype([("first_triangle", "<i8"), ("n_triangles", "<i8"), ("object", "<i4"), ("_pad", "<i4")])
assert instance_dtype.itemsize == 24
prim_run_dtype = np.dtype([("first", "<i8"), ("count", "<i8"), ("flags", "<i4"), ("_pad", "<i4")])
assert prim_run_dtype.itemsize == 24


class CameraDesc(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3), ("gaze", C.c_float * 3), ("up", C.c_float * 3),
        ("left", C.c_float), ("right", C.c_float), ("bottom", C.c_float), ("top", C.c_float), ("distance", C.c_float),
        ("nx", C.c_int32), ("ny", C.c_int32), ("num_samples", C.c_int32),
        ("focus_distance", C.c_float), ("aperture_size", C.c_float)]


class RenderOpts(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("seed", C.c_uint32), ("tile_size", C.c_int32), ("tile_rank", C.c_int32),
        ("tile_world", C.c_int32), ("output_on_device", C.c_int32), ("compact_output", C.c_int32),
        ("max_batch_samples", C.c_int32), ("timing", C.c_int32), ("num_streams", C.c_int32), ("level_counts", C.c_int32),
        ("shadow_stream", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64 * RT_RAY_KINDS), ("node_visits", C.c_uint64), ("box_tests", C.c_uint64),
        ("tri_tests", C.c_uint64), ("sphere_tests", C.c_uint64), ("ms_total", C.c_double), ("ms_kernels", C.c_double),
        ("ms_trace", C.c_double), ("trace_launches", C.c_uint64), ("trace_rays", C.c_uint64), ("ms_trace_busy", C.c_double),
        ("level_syncs", C.c_uint64), ("frame_passes", C.c_uint64)]

    def total_rays(self):
        return int(sum(self.rays))

    def as_dict(self):
        return {"rays": [int(x) for x in self.rays], "node_visits": int(self.node_visits), "box_tests": int(self.box_tests),
                "tri_tests": int(self.tri_tests), "sphere_tests": int(self.sphere_tests), "ms_total": self.ms_total,
                "ms_kernels": self.ms_kernels, "ms_trace": self.ms_trace, "trace_launches": int(self.trace_launches),
                "trace_rays": int(self.trace_rays), "ms_trace_busy": self.ms_trace_busy, "level_syncs": int(self.level_syncs),
                "frame_passes": int(self.frame_passes)}


RT_IMG_RGB8, RT_IMG_PPM_P3, RT_IMG_PPM_P6, RT_IMG_EXR_HALF = 0, 1, 2, 3


class TonemapOpts(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("key", C.c_float), ("burn_percent", C.c_float), ("saturation", C.c_float),
                ("gamma", C.c_float), ("input_8bit", C.c_int32), ("in_on_device", C.c_int32), ("out_on_device", C.c_int32)]


class PlyLayout(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("pos_type", C.c_int32), ("n_vertices", C.c_int64), ("vertex_stride", C.c_int32),
                ("pos_offset", C.c_int32 * 3), ("n_faces", C.c_int64), ("face_pre_bytes", C.c_int32), ("count_size", C.c_int32),
                ("index_size", C.c_int32), ("face_post_bytes", C.c_int32), ("format", C.c_int32), ("_pad", C.c_int32)]


RT_PLY_BINARY_LE, RT_PLY_ASCII = 0, 1
RT_SCENE_ARENAS = 12


class BvhInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_leaves", C.c_int64), ("max_depth", C.c_int32), ("n_wide_nodes", C.c_int32),
                ("n_leaf_prims", C.c_int64)]
