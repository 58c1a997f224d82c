"""Scene geometry that is produced in HBM and never visits the host: PLY bodies are decoded by the device decoder straight
into one vertex / triangle arena (rt_ply_decode_to), inline geometry is copied next to them, and the arena is handed to
rt_scene_upload as device-resident geometry (rt_scene_desc.geometry_on_device) together with primitive runs and mesh instances,
which the library expands on the device.  What the reference does on the host for the same input: happly + the PLY branch of the
scene parser (src/SceneParser.cpp:983-1023), Mesh / Triangle construction (src/Mesh.cpp:14-60) and Mesh::Clone (src/Mesh.cpp:126-180).

torch is used for the device allocations only."""
import numpy as np
import torch

from . import abi
from . import scene as _scene


class DeviceGeometry:
    def __init__(self, max_vertices, max_triangles, device="cuda:0"):
        self.device = torch.device(device)
        self.pos = torch.zeros((max(1, max_vertices), 3), dtype=torch.float32, device=self.device)
        self.uv = torch.zeros((max(1, max_vertices), 2), dtype=torch.float32, device=self.device)
        self.tri_vert = torch.zeros((max(1, max_triangles), 3), dtype=torch.int32, device=self.device)
        self.tri_object = torch.zeros((max(1, max_triangles),), dtype=torch.int32, device=self.device)
        self.max_vertices, self.max_triangles = int(max_vertices), int(max_triangles)
        self.n_vertices = 0
        self.n_triangles = 0

    def add_ply(self, ctx, path, obj):
        """Decodes a PLY file (binary little endian or ascii) into the arena; its triangles belong to object `obj`.
        Returns (first vertex, first triangle, number of triangles)."""
        with open(path, "rb") as f:
            fmt, elements = _scene._ply_header(f)
            body = f.read()
        L = _scene.ply_layout(elements, ascii=(fmt == "ascii")) if fmt in ("ascii", "binary_little_endian") else None
        if L is None:
            pos, tris = _scene._read_ply(path)                   # a layout the device decoder does not take: general host reader
            return self.add_host(pos, None, tris, obj)
        nv, nf = int(L.n_vertices), int(L.n_faces)
        if self.n_vertices + nv > self.max_vertices or self.n_triangles + 2 * nf > self.max_triangles:
            raise ValueError("device geometry arena too small for %s (a face can give two triangles)" % path)
        if L.format == abi.RT_PLY_ASCII:
            vb, fb = memoryview(body), b""
        else:
            vbytes = nv * int(L.vertex_stride)
            vb, fb = memoryview(body)[:vbytes], memoryview(body)[vbytes:]
        v0, t0 = self.n_vertices, self.n_triangles
        torch.cuda.synchronize(self.device)                      # the decoder runs on the context's own stream
        try:
            n = ctx.decode_ply_to(vb, fb, L, self.pos[v0:].data_ptr(), self.tri_vert[t0:].data_ptr(), index_offset=v0)
        except Exception as e:                                   # a body the device decoder refuses: the general host reader
            if type(e).__name__ != "RtError":
                raise
            pos, tris = _scene._read_ply(path)
            return self.add_host(pos, None, tris, obj)
        self.tri_object[t0:t0 + n] = int(obj)
        self.n_vertices, self.n_triangles = v0 + nv, t0 + n
        return v0, t0, n

    def add_host(self, pos, uv, tris, obj):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        tris = np.ascontiguousarray(tris, np.int64).reshape(-1, 3)
        v0, t0 = self.n_vertices, self.n_triangles
        if v0 + len(pos) > self.max_vertices or t0 + len(tris) > self.max_triangles:
            raise ValueError("device geometry arena too small")
        self.pos[v0:v0 + len(pos)] = torch.from_numpy(pos).to(self.device)
        if uv is not None:
            self.uv[v0:v0 + len(pos)] = torch.from_numpy(np.ascontiguousarray(uv, np.float32).reshape(-1, 2)).to(self.device)
        self.tri_vert[t0:t0 + len(tris)] = torch.from_numpy((tris + v0).astype(np.int32)).to(self.device)
        self.tri_object[t0:t0 + len(tris)] = int(obj)
        self.n_vertices, self.n_triangles = v0 + len(pos), t0 + len(tris)
        return v0, t0, len(tris)

    def attach(self, soup):
        """Makes `soup` (objects, spheres, materials, lights, instances, primitive runs filled in; no host triangles) describe
        this arena's geometry.  The arena must stay alive until rt_scene_upload has returned."""
        torch.cuda.synchronize(self.device)
        soup.vert_pos, soup.vert_uv = np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32)
        soup.tri_vert, soup.tri_object = np.zeros((0, 3), np.int32), np.zeros((0,), np.int32)
        soup.device_geometry = (self.pos.data_ptr(), self.uv.data_ptr(), self.tri_vert.data_ptr(), self.tri_object.data_ptr(),
                                self.n_vertices, self.n_triangles, self)
        return soup
