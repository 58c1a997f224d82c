// rt_ingest.hip — scene ingest on the device: what the reference computes in its shape constructors,
// for scenes large enough to be built on the GPU.  The raw description arrays (positions, indices, object ids, primitive
// list) are uploaded once; everything derived from them is produced here and stays in HBM:
//   Triangle::Triangle            src/Triangle.cpp:48-76   edges, face normal                         k_tri_setup
//   Mesh smooth normals           src/Mesh.cpp:88-95, src/Triangle.cpp:78-105 (Q28)                   k_corner_keys + sort + k_vertex_normals
//   Primitive world boxes         src/acmath.h:1034-1050, src/Shape.cpp:55-65 (Q5, Q29)              k_prim_box
// Results are bit-identical to the host loops of rt_build.hip (same inline functions, -ffp-contract=off); the only order
// dependent step is the normal sum of a vertex, which the reference accumulates in triangle order: corners are sorted by
// vertex with a STABLE radix sort, so every vertex sums its face normals in exactly that order.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "rt_build.h"
#include "rt_bvh_split.h"
#include "rt_math.h"

namespace {

#define IN_TRY(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); rc = RT_ERR_HIP; goto done; } \
    } while (0)

constexpr unsigned NO_VERTEX = 0xffffffffu;

__global__ __launch_bounds__(256) void k_tri_setup(const float *vpos, const int *tri_vert, const int *tri_obj, int nT, DevPrim *rec) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nT) return;
    const int i0 = tri_vert[3 * t], i1 = tri_vert[3 * t + 1], i2 = tri_vert[3 * t + 2];
    const f3 a = ld3(vpos + 3ll * i0), b = ld3(vpos + 3ll * i1), c = ld3(vpos + 3ll * i2);
    const f3 ab = a - b, ac = a - c;                                   // p0p1, p0p2 (Triangle.cpp:63-64)
    const f3 fn = normalize3(cross3(-ab, -ac));                        // Triangle.cpp:66
    DevPrim r;
    r.p0[0] = a.x; r.p0[1] = a.y; r.p0[2] = a.z;
    r.e1[0] = ab.x; r.e1[1] = ab.y; r.e1[2] = ab.z;
    r.e2[0] = ac.x; r.e2[1] = ac.y; r.e2[2] = ac.z;
    r.nrm[0] = fn.x; r.nrm[1] = fn.y; r.nrm[2] = fn.z;
    r.object = tri_obj[t]; r.shape = t; r.kind = DEV_PRIM_TRIANGLE; r.ill = tri_ill_conditioned(ab, ac) ? 1 : 0;
    rec[t] = r;
}

// One key per triangle corner: the vertex it touches when the triangle's object is smooth, NO_VERTEX (sorts last) otherwise.
__global__ __launch_bounds__(256) void k_corner_keys(const int *tri_vert, const int *tri_obj, const DevObject *objects, int n_corners, unsigned *key, unsigned *val) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_corners) return;
    const bool smooth = (objects[tri_obj[i / 3]].flags & DEV_OBJ_SMOOTH) != 0;
    key[i] = smooth ? (unsigned)tri_vert[i] : NO_VERTEX;
    val[i] = (unsigned)i;
}

__device__ __forceinline__ int lower_bound_u(const unsigned *a, int n, unsigned x) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

// A vertex adds the face normals of its corners in corner order (= triangle order inside its mesh) and is normalised once
// (Mesh.cpp:88-95).  A vertex touched by SMOOTH triangles of two different objects would be normalised between the two
// meshes in the reference; that case is reported through *shared and left to the host path.
__global__ __launch_bounds__(256) void k_vertex_normals(const unsigned *skey, const unsigned *sval, int n_corners, const DevPrim *rec, const int *tri_obj,
                                                       const float *vuv, int nV, DevVertex *verts, int *shared) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nV) return;
    DevVertex dv;
    dv.n[0] = dv.n[1] = dv.n[2] = 0.0f;
    dv.uv[0] = vuv ? vuv[2 * v] : 0.0f;
    dv.uv[1] = vuv ? vuv[2 * v + 1] : 0.0f;
    dv._pad[0] = dv._pad[1] = dv._pad[2] = 0.0f;
    if (n_corners > 0) {
        int i = lower_bound_u(skey, n_corners, (unsigned)v);
        int ob = -1;
        bool any = false;
        for (; i < n_corners && skey[i] == (unsigned)v; ++i) {
            const int t = (int)(sval[i] / 3u);
            dv.n[0] += rec[t].nrm[0]; dv.n[1] += rec[t].nrm[1]; dv.n[2] += rec[t].nrm[2];
            const int o = tri_obj[t];
            if (any && o != ob) *shared = 1;
            ob = o; any = true;
        }
        if (any) { const f3 n = normalize3(ld3(dv.n)); dv.n[0] = n.x; dv.n[1] = n.y; dv.n[2] = n.z; }
    }
    verts[v] = dv;
}

// Transform::operator()(BoundingVolume3f) — the 8 corners through M, merged (acmath.h:1034-1050); same order as the host's.
__device__ __forceinline__ Aabb transform_box_dev(const float *M, const Aabb &b) {
    const f3 c0 = mat_mul_xyz(M, b.lo.x, b.lo.y, b.lo.z, 1.0f);
    Aabb r = box_of(c0, c0);
    r = merge(r, mat_mul_xyz(M, b.hi.x, b.lo.y, b.lo.z, 1.0f));
    r = merge(r, mat_mul_xyz(M, b.lo.x, b.hi.y, b.lo.z, 1.0f));
    r = merge(r, mat_mul_xyz(M, b.lo.x, b.lo.y, b.hi.z, 1.0f));
    r = merge(r, mat_mul_xyz(M, b.lo.x, b.hi.y, b.hi.z, 1.0f));
    r = merge(r, mat_mul_xyz(M, b.hi.x, b.hi.y, b.lo.z, 1.0f));
    r = merge(r, mat_mul_xyz(M, b.hi.x, b.lo.y, b.hi.z, 1.0f));
    r = merge(r, mat_mul_xyz(M, b.hi.x, b.hi.y, b.hi.z, 1.0f));
    return r;
}

// World box of every reference primitive (Q5: mesh triangles appear twice, the second time with the motion-blur extension).
__global__ __launch_bounds__(256) void k_prim_box(const int *prim_shape, int nP, int nT, const float *vpos, const int *tri_vert, const DevPrim *rec,
                                                 const DevObject *objects, float *pbox, int *pshape) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nP) return;
    const int ref = prim_shape[i];
    const int idx = ref & RT_PRIM_INDEX_MASK;
    const bool sph = (ref & RT_PRIM_SPHERE) != 0;
    const int si = sph ? nT + idx : idx;
    Aabb sb;
    if (sph) {                                                         // Sphere.cpp:9-21
        const f3 c = ld3(rec[si].p0);
        const float rad = rec[si].e1[0];
        sb = box_of(c + mk3(-rad, -rad, -rad), c + mk3(rad, rad, rad));
    } else {                                                           // Triangle.cpp:68-69
        const f3 a = ld3(vpos + 3ll * tri_vert[3 * idx]), b = ld3(vpos + 3ll * tri_vert[3 * idx + 1]), c = ld3(vpos + 3ll * tri_vert[3 * idx + 2]);
        sb = box_of(max3(a, max3(b, c)), min3(a, min3(b, c)));
    }
    const DevObject &ob = objects[rec[si].object];
    Aabb wb = (ob.flags & DEV_OBJ_HAS_XFORM) ? transform_box_dev(ob.M, sb) : sb;
    if ((ref & RT_PRIM_BLURBOX) && (ob.flags & DEV_OBJ_MOTION)) {      // Shape::AdaptWorldBoundingBoxForMotionBlur
        float moved[16];
        mat_translate_mul(ld3(ob.mblur), ob.M, moved);
        wb = merge(wb, transform_box_dev(moved, sb));
    }
    float *o = pbox + 6ll * i;
    o[0] = wb.lo.x; o[1] = wb.lo.y; o[2] = wb.lo.z; o[3] = wb.hi.x; o[4] = wb.hi.y; o[5] = wb.hi.z;
    pshape[i] = si;
}

// Mesh::Clone (src/Mesh.cpp:126-180) for every rt_instance: the clone's triangles take the vertex indices of the base mesh's
// triangles and the instance's object.  inst_start[i] = first virtual triangle of instance i (inst_start[n_inst] = end).
__global__ __launch_bounds__(256) void k_expand_instances(const rt_instance *inst, const long long *inst_start, int n_inst, int nT_explicit, int nT_total,
                                                         int *tri_vert, int *tri_obj) {
    const int t = nT_explicit + blockIdx.x * 256 + threadIdx.x;
    if (t >= nT_total) return;
    int lo = 0, hi = n_inst - 1;                                       // last instance whose start is <= t
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (inst_start[mid] <= t) lo = mid; else hi = mid - 1; }
    const rt_instance in = inst[lo];
    const long long src = in.first_triangle + (t - inst_start[lo]);
    tri_vert[3ll * t] = tri_vert[3 * src]; tri_vert[3ll * t + 1] = tri_vert[3 * src + 1]; tri_vert[3ll * t + 2] = tri_vert[3 * src + 2];
    tri_obj[t] = in.object;
}
// Scene::primitives written out from rt_prim_run records; run_start[i] = position of the first entry of run i.
__global__ __launch_bounds__(256) void k_expand_runs(const rt_prim_run *runs, const long long *run_start, int n_runs, int nP, int *prim_shape) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= nP) return;
    int lo = 0, hi = n_runs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (run_start[mid] <= p) lo = mid; else hi = mid - 1; }
    prim_shape[p] = (int)(runs[lo].first + (p - run_start[lo])) | runs[lo].flags;
}
// Range checks of geometry that never visited the host: every vertex index inside [0, nV), every object inside [0, n_obj).
__global__ __launch_bounds__(256) void k_check_triangles(const int *tri_vert, const int *tri_obj, int nT, int nV, int n_obj, int *bad) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nT) return;
    const int a = tri_vert[3ll * t], b = tri_vert[3ll * t + 1], c = tri_vert[3ll * t + 2], o = tri_obj[t];
    if (a < 0 || a >= nV || b < 0 || b >= nV || c < 0 || c >= nV) atomicOr(bad, 1);
    if (o < 0 || o >= n_obj) atomicOr(bad, 2);
}

template <typename T>
hipError_t ialloc(T **p, size_t n) { return hipMalloc((void **)p, (n ? n : 1) * sizeof(T)); }

double in_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

void RtGpuIngest::release() {
    hipFree(pbox); hipFree(shape_rec); hipFree(pshape); hipFree(verts); hipFree(tri_vert);
    pbox = nullptr; shape_rec = nullptr; pshape = nullptr; verts = nullptr; tri_vert = nullptr;
}

// Returns RT_OK, a negative status, or 1 when the scene needs the host path (a vertex shared by two smooth objects).
// `d` has been validated by the caller (index ranges).  sphere_rec: the nS sphere records (host).
int rt_ingest_gpu(const rt_scene_desc *d, int64_t n_tris_total, int64_t n_prims_total, const std::vector<DevObject> &objects, const std::vector<DevPrim> &sphere_rec,
                  RtGpuIngest &out, std::string &err) {
    out = RtGpuIngest();
    int rc = RT_OK;
    const bool tm = getenv("RT_BUILD_TIMING") != nullptr;
    double t_mark = in_now_ms();
    auto lap = [&](const char *what) { if (tm) { hipDeviceSynchronize(); const double t = in_now_ms(); fprintf(stderr, "[rt gpu ingest] %-23s %8.2f ms\n", what, t - t_mark); t_mark = t; } };
    const int nTe = (int)d->n_triangles;                               // explicit triangles of the description
    const int nT = (int)n_tris_total, nS = (int)d->n_spheres, nP = (int)n_prims_total, nV = (int)d->n_vertices;
    const int n_corners = 3 * nT;
    if (n_tris_total > 0x7fffffff / 3) { err = "too many triangles for the device ingest"; return RT_ERR_TOO_LARGE; }
    const bool on_dev = d->geometry_on_device != 0;
    float *d_vpos = nullptr, *d_vuv = nullptr;
    int *d_tri_obj = nullptr, *d_prim_shape = nullptr, *d_shared = nullptr;
    DevObject *d_objects = nullptr;
    rt_instance *d_inst = nullptr; rt_prim_run *d_runs = nullptr; long long *d_starts = nullptr;
    unsigned *d_key[2] = {nullptr, nullptr}, *d_val[2] = {nullptr, nullptr};
    void *d_tmp = nullptr;
    hipStream_t st = nullptr;
    const dim3 blk(256);
    auto grid = [](int n) { return dim3((unsigned)((n + 255) / 256)); };
    {
        IN_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        IN_TRY(ialloc(&out.tri_vert, 3 * (size_t)nT)); IN_TRY(ialloc(&d_tri_obj, (size_t)nT));
        IN_TRY(ialloc(&d_prim_shape, (size_t)nP)); IN_TRY(ialloc(&d_objects, objects.size())); IN_TRY(ialloc(&d_shared, 2));
        IN_TRY(ialloc(&out.shape_rec, (size_t)nT + (size_t)nS)); IN_TRY(ialloc(&out.verts, (size_t)nV));
        IN_TRY(ialloc(&out.pbox, 6 * (size_t)nP)); IN_TRY(ialloc(&out.pshape, (size_t)nP));
        if (on_dev) {                                                  // vertices are borrowed from the caller; triangles are copied device to device
            d_vpos = const_cast<float *>(d->vert_pos); d_vuv = const_cast<float *>(d->vert_uv);
            if (nTe) IN_TRY(hipMemcpyAsync(out.tri_vert, d->tri_vert, 3 * (size_t)nTe * sizeof(int), hipMemcpyDeviceToDevice, st));
            if (nTe) IN_TRY(hipMemcpyAsync(d_tri_obj, d->tri_object, (size_t)nTe * sizeof(int), hipMemcpyDeviceToDevice, st));
        } else {
            IN_TRY(ialloc(&d_vpos, 3 * (size_t)nV));
            if (d->vert_uv) IN_TRY(ialloc(&d_vuv, 2 * (size_t)nV));
            if (nV) IN_TRY(hipMemcpyAsync(d_vpos, d->vert_pos, 3 * (size_t)nV * sizeof(float), hipMemcpyHostToDevice, st));
            if (nV && d->vert_uv) IN_TRY(hipMemcpyAsync(d_vuv, d->vert_uv, 2 * (size_t)nV * sizeof(float), hipMemcpyHostToDevice, st));
            if (nTe) IN_TRY(hipMemcpyAsync(out.tri_vert, d->tri_vert, 3 * (size_t)nTe * sizeof(int), hipMemcpyHostToDevice, st));
            if (nTe) IN_TRY(hipMemcpyAsync(d_tri_obj, d->tri_object, (size_t)nTe * sizeof(int), hipMemcpyHostToDevice, st));
        }
        if (!objects.empty()) IN_TRY(hipMemcpyAsync(d_objects, objects.data(), objects.size() * sizeof(DevObject), hipMemcpyHostToDevice, st));
        if (nS) IN_TRY(hipMemcpyAsync(out.shape_rec + nT, sphere_rec.data(), (size_t)nS * sizeof(DevPrim), hipMemcpyHostToDevice, st));
        IN_TRY(hipMemsetAsync(d_shared, 0, 2 * sizeof(int), st));
        // the small tables of the compact forms, with the start of every record after expansion
        std::vector<long long> starts;
        if (d->n_instances > 0) {
            long long at = nTe;
            for (int64_t i = 0; i < d->n_instances; ++i) { starts.push_back(at); at += d->instances[i].n_triangles; }
            starts.push_back(at);
            IN_TRY(ialloc(&d_inst, (size_t)d->n_instances)); IN_TRY(ialloc(&d_starts, starts.size() + (size_t)d->n_prim_runs + 2));
            IN_TRY(hipMemcpyAsync(d_inst, d->instances, (size_t)d->n_instances * sizeof(rt_instance), hipMemcpyHostToDevice, st));
            IN_TRY(hipMemcpyAsync(d_starts, starts.data(), starts.size() * sizeof(long long), hipMemcpyHostToDevice, st));
            IN_TRY(hipStreamSynchronize(st));                          // `starts` is re-used below
            hipLaunchKernelGGL(k_expand_instances, grid(nT - nTe), blk, 0, st, d_inst, d_starts, (int)d->n_instances, nTe, nT, out.tri_vert, d_tri_obj);
        }
        if (d->prim_runs) {
            std::vector<long long> rs;
            long long at = 0;
            for (int64_t i = 0; i < d->n_prim_runs; ++i) { rs.push_back(at); at += d->prim_runs[i].count; }
            rs.push_back(at);
            long long *d_rs = nullptr;
            IN_TRY(ialloc(&d_runs, (size_t)d->n_prim_runs));
            if (!d_starts) IN_TRY(ialloc(&d_starts, (size_t)d->n_prim_runs + 2));
            d_rs = d_starts + (d->n_instances > 0 ? (size_t)d->n_instances + 1 : 0);
            if (d->n_prim_runs) IN_TRY(hipMemcpyAsync(d_runs, d->prim_runs, (size_t)d->n_prim_runs * sizeof(rt_prim_run), hipMemcpyHostToDevice, st));
            IN_TRY(hipMemcpyAsync(d_rs, rs.data(), rs.size() * sizeof(long long), hipMemcpyHostToDevice, st));
            IN_TRY(hipStreamSynchronize(st));
            if (nP) hipLaunchKernelGGL(k_expand_runs, grid(nP), blk, 0, st, d_runs, d_rs, (int)d->n_prim_runs, nP, d_prim_shape);
        } else if (nP) IN_TRY(hipMemcpyAsync(d_prim_shape, d->prim_shape, (size_t)nP * sizeof(int), hipMemcpyHostToDevice, st));
        if (on_dev && nTe) {
            // geometry that never visited the host is range-checked here, BEFORE any kernel dereferences its indices
            int bad = 0;
            hipLaunchKernelGGL(k_check_triangles, grid(nTe), blk, 0, st, out.tri_vert, d_tri_obj, nTe, nV, (int)objects.size(), d_shared + 1);
            IN_TRY(hipMemcpyAsync(&bad, d_shared + 1, sizeof(int), hipMemcpyDeviceToHost, st));
            IN_TRY(hipStreamSynchronize(st));
            if (bad) { err = (bad & 1) ? "triangle references a missing vertex" : "triangle references a missing object"; rc = RT_ERR_INVALID; goto done; }
        }
        lap("upload description");
        if (nT) hipLaunchKernelGGL(k_tri_setup, grid(nT), blk, 0, st, d_vpos, out.tri_vert, d_tri_obj, nT, out.shape_rec);
        bool any_smooth = false;
        for (const DevObject &o : objects) any_smooth = any_smooth || (o.flags & DEV_OBJ_SMOOTH);
        int sorted_corners = 0;
        if (any_smooth && n_corners > 0) {
            IN_TRY(ialloc(&d_key[0], (size_t)n_corners)); IN_TRY(ialloc(&d_key[1], (size_t)n_corners));
            IN_TRY(ialloc(&d_val[0], (size_t)n_corners)); IN_TRY(ialloc(&d_val[1], (size_t)n_corners));
            hipLaunchKernelGGL(k_corner_keys, grid(n_corners), blk, 0, st, out.tri_vert, d_tri_obj, d_objects, n_corners, d_key[0], d_val[0]);
            size_t tmp_bytes = 0;
            IN_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key[0], d_key[1], d_val[0], d_val[1], n_corners, 0, 32, st));
            IN_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
            IN_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_key[0], d_key[1], d_val[0], d_val[1], n_corners, 0, 32, st));
            sorted_corners = n_corners;
        }
        if (nV) hipLaunchKernelGGL(k_vertex_normals, grid(nV), blk, 0, st, d_key[1], d_val[1], sorted_corners, out.shape_rec, d_tri_obj, d_vuv, nV, out.verts, d_shared);
        if (nP) hipLaunchKernelGGL(k_prim_box, grid(nP), blk, 0, st, d_prim_shape, nP, nT, d_vpos, out.tri_vert, out.shape_rec, d_objects, out.pbox, out.pshape);
        IN_TRY(hipGetLastError());
        int flags[2] = {0, 0};
        IN_TRY(hipMemcpyAsync(flags, d_shared, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        IN_TRY(hipStreamSynchronize(st));
        lap("records, normals, boxes");
        if (flags[1]) { err = (flags[1] & 1) ? "triangle references a missing vertex" : "triangle references a missing object"; rc = RT_ERR_INVALID; }
        else if (flags[0]) rc = 1;
    }
done:
    if (!on_dev) { hipFree(d_vpos); hipFree(d_vuv); }
    hipFree(d_tri_obj); hipFree(d_prim_shape); hipFree(d_objects); hipFree(d_shared); hipFree(d_inst); hipFree(d_runs); hipFree(d_starts);
    hipFree(d_key[0]); hipFree(d_key[1]); hipFree(d_val[0]); hipFree(d_val[1]); hipFree(d_tmp);
    if (st) hipStreamDestroy(st);
    if (rc != RT_OK) out.release();
    return rc;
}
