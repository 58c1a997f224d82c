// rt_build.hip — host side of rt_scene_upload: what the reference computes in its Shape / Mesh / Primitive
// constructors and in BVHTree::BuildTree, re-done as flat arrays and emitted in the HBM layout of
// rt_device.h.  Host code only (no kernels); compiled by hipcc with -ffp-contract=off.
//
//   shape setup      src/Triangle.cpp:48-76, src/Sphere.cpp:9-21, src/Mesh.cpp:88-95 (smooth normals, Q28)
//   world boxes      src/acmath.h:1034-1050, src/Shape.cpp:55-65 (motion-blur extension, Q29), src/Primitive.h:21-26 (Q5)
//   BVH build        src/BVHTree.cpp:39-153: 4-bin SAH on centroids, std::partition (same libstdc++ call, so the
//                    re-ordering inside each side is the reference's), leaves keep index ranges
//   flattening       new: preorder internal nodes with both child boxes, leaves numbered in DFS order (Q4 tie-break)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>

#include "rt_build.h"
#include "rt_bvh_split.h"
#include "rt_cone.h"
#include "rt_math.h"

namespace {

// Transform::operator()(BoundingVolume3f) — the 8 corners through M, merged (acmath.h:1034-1050)
Aabb transform_box(const float *M, const Aabb &b) {
    f3 c0 = mat_mul_xyz(M, b.lo.x, b.lo.y, b.lo.z, 1.0f);
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

struct BuildNode {
    Aabb box;
    int32_t left, right;     // build-node indices, -1 for a leaf
    int32_t start, end;
    int32_t depth;
};

struct Task {
    int32_t node, start, end, depth;
};

constexpr int kBins = RT_BVH_BINS;

// DevObject::cone_k (rt_cone.h): sigma_min^2 / sigma_max of the linear part of the object matrix, from the eigenvalues of
// A^T A (cyclic Jacobi in double).  0 (= never prune by distance inside this object) for a matrix that is not affine.
float cone_k_of_matrix(const float *M) {
    if (M[3] != 0.0f || M[7] != 0.0f || M[11] != 0.0f || M[15] != 1.0f) return 0.0f;
    double S[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double v = 0;
            for (int r = 0; r < 3; ++r) v += (double)M[i * 4 + r] * (double)M[j * 4 + r];      // columns i and j of A
            S[i][j] = v;
        }
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = std::fabs(S[0][1]) + std::fabs(S[0][2]) + std::fabs(S[1][2]);
        if (!(off > 1e-300)) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (S[p][q] == 0.0) continue;
                const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; ++k) { const double a = S[k][p], b = S[k][q]; S[k][p] = c * a - sn * b; S[k][q] = sn * a + c * b; }
                for (int k = 0; k < 3; ++k) { const double a = S[p][k], b = S[q][k]; S[p][k] = c * a - sn * b; S[q][k] = sn * a + c * b; }
            }
    }
    const double lmax = std::max(S[0][0], std::max(S[1][1], S[2][2])), lmin = std::min(S[0][0], std::min(S[1][1], S[2][2]));
    if (!(lmax > 0.0) || !(lmin > 0.0) || !std::isfinite(lmax)) return 0.0f;
    const double k = lmin / std::sqrt(lmax) * (1.0 - 1e-4);
    return std::isfinite(k) ? (float)k : 0.0f;
}

}  // namespace

// The compact forms of a scene description (rt_instance, rt_prim_run, device-resident geometry) written out on the host: what
// Mesh::Clone (src/Mesh.cpp:126-180) and the pushes into Scene::primitives (src/Mesh.cpp:57-58, 182-192) produce.  Used by
// the host build path (small scenes, A/B runs); the device path expands them in rt_ingest.hip.
struct HostExpanded {
    std::vector<float> vert_pos, vert_uv;
    std::vector<int32_t> tri_vert, tri_object, prim_shape;
};
static thread_local bool g_force_host_ingest = false;      // set while a scene the device ingest handed back is built by the host loops
static bool scene_is_compact(const rt_scene_desc *d) { return d->n_instances > 0 || d->prim_runs != nullptr || d->geometry_on_device != 0; }
// totals after expansion; false + err on an inconsistent record
static bool compact_totals(const rt_scene_desc *d, int64_t &nT_total, int64_t &nP_total, std::string &err) {
    nT_total = d->n_triangles;
    if (d->n_instances < 0 || d->n_prim_runs < 0 || (d->n_instances && !d->instances)) { err = "bad instance / primitive-run table"; return false; }
    for (int64_t i = 0; i < d->n_instances; ++i) {
        const rt_instance &in = d->instances[i];
        if (in.first_triangle < 0 || in.n_triangles < 0 || in.first_triangle + in.n_triangles > d->n_triangles) { err = "instance references triangles outside the explicit ones"; return false; }
        if (in.object < 0 || in.object >= d->n_objects) { err = "instance references a missing object"; return false; }
        nT_total += in.n_triangles;
        if (nT_total > RT_PRIM_INDEX_MASK) { err = "too many triangles (limit 2^29-1)"; return false; }
    }
    nP_total = d->n_prims;
    if (d->prim_runs) {
        nP_total = 0;
        for (int64_t i = 0; i < d->n_prim_runs; ++i) {
            const rt_prim_run &r = d->prim_runs[i];
            const int64_t limit = (r.flags & RT_PRIM_SPHERE) ? d->n_spheres : nT_total;
            if (r.first < 0 || r.count < 0 || r.first + r.count > limit || (r.flags & ~(RT_PRIM_SPHERE | RT_PRIM_BLURBOX))) { err = "primitive run references a missing shape"; return false; }
            nP_total += r.count;
            if (nP_total > RT_PRIM_INDEX_MASK) { err = "too many primitives (limit 2^29-1)"; return false; }
        }
    }
    return true;
}
static int expand_on_host(const rt_scene_desc *d, HostExpanded &x, rt_scene_desc &dd, std::string &err) {
    int64_t nT_total = 0, nP_total = 0;
    if (!compact_totals(d, nT_total, nP_total, err)) return RT_ERR_INVALID;
    dd = *d;
    x.tri_vert.resize(3 * (size_t)nT_total); x.tri_object.resize((size_t)nT_total);
    if (d->geometry_on_device) {
        x.vert_pos.resize(3 * (size_t)d->n_vertices);
        if (d->n_vertices && hipMemcpy(x.vert_pos.data(), d->vert_pos, x.vert_pos.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { err = "copy of device-resident vertices failed"; return RT_ERR_HIP; }
        if (d->vert_uv) {
            x.vert_uv.resize(2 * (size_t)d->n_vertices);
            if (d->n_vertices && hipMemcpy(x.vert_uv.data(), d->vert_uv, x.vert_uv.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { err = "copy of device-resident uv failed"; return RT_ERR_HIP; }
        }
        if (d->n_triangles && (hipMemcpy(x.tri_vert.data(), d->tri_vert, 3 * (size_t)d->n_triangles * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
                               hipMemcpy(x.tri_object.data(), d->tri_object, (size_t)d->n_triangles * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)) { err = "copy of device-resident triangles failed"; return RT_ERR_HIP; }
        dd.vert_pos = x.vert_pos.data(); dd.vert_uv = d->vert_uv ? x.vert_uv.data() : nullptr;
    } else {
        if (d->n_triangles) { std::memcpy(x.tri_vert.data(), d->tri_vert, 3 * (size_t)d->n_triangles * sizeof(int32_t)); std::memcpy(x.tri_object.data(), d->tri_object, (size_t)d->n_triangles * sizeof(int32_t)); }
    }
    int64_t t = d->n_triangles;
    for (int64_t i = 0; i < d->n_instances; ++i) {
        const rt_instance &in = d->instances[i];
        for (int64_t k = 0; k < in.n_triangles; ++k, ++t) {
            for (int q = 0; q < 3; ++q) x.tri_vert[3 * (size_t)t + q] = x.tri_vert[3 * (size_t)(in.first_triangle + k) + q];
            x.tri_object[(size_t)t] = in.object;
        }
    }
    dd.n_triangles = nT_total; dd.tri_vert = x.tri_vert.data(); dd.tri_object = x.tri_object.data();
    if (d->prim_runs) {
        x.prim_shape.resize((size_t)nP_total);
        size_t p = 0;
        for (int64_t i = 0; i < d->n_prim_runs; ++i)
            for (int64_t k = 0; k < d->prim_runs[i].count; ++k) x.prim_shape[p++] = (int32_t)(d->prim_runs[i].first + k) | d->prim_runs[i].flags;
        dd.n_prims = nP_total; dd.prim_shape = x.prim_shape.data();
    }
    dd.n_instances = 0; dd.instances = nullptr; dd.n_prim_runs = 0; dd.prim_runs = nullptr; dd.geometry_on_device = 0;
    return RT_OK;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int rt_build_host_scene(const rt_scene_desc *d, HostScene &hs, std::string &err, int gpu_build, RtGpuScene *device_out) {
    const bool tm = getenv("RT_BUILD_TIMING") != nullptr;
    double t_mark = now_s();
    auto lap = [&](const char *what) { if (tm) { double t = now_s(); fprintf(stderr, "[rt build] %-28s %8.1f ms\n", what, (t - t_mark) * 1e3); t_mark = t; } };
    if (!d) { err = "scene description is NULL"; return RT_ERR_INVALID; }
    if (d->abi_version != RT_ABI_VERSION) { err = "rt_scene_desc.abi_version mismatch"; return RT_ERR_ABI; }
    if (d->n_vertices < 0 || d->n_triangles < 0 || d->n_spheres < 0 || d->n_objects < 0 || d->n_prims < 0 || d->n_materials < 0 ||
        d->n_lights < 0 || d->n_textures < 0) { err = "negative count in scene description"; return RT_ERR_INVALID; }
    if (d->n_prims > RT_PRIM_INDEX_MASK || d->n_triangles > RT_PRIM_INDEX_MASK) { err = "too many primitives (limit 2^29-1)"; return RT_ERR_TOO_LARGE; }
    if ((d->n_triangles && (!d->tri_vert || !d->tri_object || !d->vert_pos)) || (d->n_spheres && (!d->sph_center_radius || !d->sph_object)) ||
        (d->n_objects && !d->objects) || (d->n_prims && !d->prim_shape && !d->prim_runs) || (d->n_materials && !d->materials) ||
        (d->n_lights && !d->lights) || (d->n_textures && !d->textures)) { err = "NULL array with non-zero count"; return RT_ERR_INVALID; }
    // compact forms: instances / primitive runs / device-resident geometry.  The device path below expands them in HBM; every
    // other path gets them written out here first.
    const bool compact = scene_is_compact(d);
    int64_t nT_total = d->n_triangles, nP_total = d->n_prims;
    if (compact && !compact_totals(d, nT_total, nP_total, err)) return RT_ERR_INVALID;
    const bool device_path = nP_total > 0 && device_out && (gpu_build == 2 || (gpu_build == 1 && nP_total >= RT_GPU_BUILD_MIN_PRIMS && !getenv("RT_HOST_BUILD"))) &&
                             !getenv("RT_HOST_FLATTEN") && !getenv("RT_HOST_INGEST") && !g_force_host_ingest;
    if (compact && !device_path) {
        HostExpanded x;
        rt_scene_desc dd;
        const int rc = expand_on_host(d, x, dd, err);
        if (rc != RT_OK) return rc;
        return rt_build_host_scene(&dd, hs, err, gpu_build, device_out);
    }

    // ---- tables copied as they are ---------------------------------------------------------------
    hs.materials.assign(d->materials, d->materials + d->n_materials);
    hs.lights.assign(d->lights, d->lights + d->n_lights);
    hs.textures.assign(d->textures, d->textures + d->n_textures);
    hs.texels.clear();
    if (d->texels_bytes > 0 && d->texels) hs.texels.assign(d->texels, d->texels + d->texels_bytes);
    for (const rt_texture &t : hs.textures) {
        if (t.kind == RT_TEX_IMAGE) {
            int64_t need = (int64_t)t.width * t.height * 3 * (t.texel_is_float ? 4 : 1);
            if (t.width <= 0 || t.height <= 0 || t.texel_offset < 0 || t.texel_offset + need > (int64_t)hs.texels.size()) { err = "image texture outside the texel arena"; return RT_ERR_INVALID; }
        } else if (t.kind == RT_TEX_PERLIN) {
            for (int i = 0; i < 16; ++i) if (t.perm[i] < 0 || t.perm[i] > 15) { err = "perlin permutation entry out of range"; return RT_ERR_INVALID; }
        } else { err = "unknown texture kind"; return RT_ERR_INVALID; }
    }
    if (d->bg_texture >= d->n_textures) { err = "bg_texture out of range"; return RT_ERR_INVALID; }

    // ---- objects ---------------------------------------------------------------------------------
    hs.objects.resize(d->n_objects);
    for (int64_t i = 0; i < d->n_objects; ++i) {
        const rt_object &o = d->objects[i];
        DevObject &q = hs.objects[i];
        if (o.material < 0 || o.material >= d->n_materials || o.tex_color >= d->n_textures || o.tex_normal >= d->n_textures) { err = "object references a missing material/texture"; return RT_ERR_INVALID; }
        std::memcpy(q.M, o.xform, sizeof(q.M));
        float Mt[16];
        mat_transpose(q.M, Mt);                      // Transform::UpdateTransform, acmath.h:998-1002
        mat_inverse(Mt, q.inv);
        q.mblur[0] = o.motion_blur[0]; q.mblur[1] = o.motion_blur[1]; q.mblur[2] = o.motion_blur[2];
        bool moving = o.motion_blur[0] != 0 || o.motion_blur[1] != 0 || o.motion_blur[2] != 0;   // Shape.h:111
        q.flags = (o.flags & RT_OBJ_SMOOTH ? DEV_OBJ_SMOOTH : 0) | (o.flags & RT_OBJ_HAS_XFORM ? DEV_OBJ_HAS_XFORM : 0) |
                  (o.flags & RT_OBJ_COMPOSITE ? DEV_OBJ_COMPOSITE : 0) | (moving ? DEV_OBJ_MOTION : 0) | (mat_is_identity(Mt) ? DEV_OBJ_IDENTITY : 0);
        {   // bit for bit the unit matrix, not moving: the object-space ray of every such object is the same
            static const float unit[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
            if ((o.flags & RT_OBJ_HAS_XFORM) && !moving && std::memcmp(q.M, unit, sizeof(unit)) == 0) q.flags |= DEV_OBJ_UNIT;
        }
        q.material = o.material; q.tex_color = o.tex_color; q.tex_normal = o.tex_normal;
        q.cone_k = (o.flags & RT_OBJ_HAS_XFORM) ? cone_k_of_matrix(q.M) : 1.0f;
    }

    // ---- large scenes: shape setup, build and flattening all on the device (rt_ingest.hip, rt_bvh_gpu.hip) -----------
    const bool want_gpu = nP_total > 0 && (gpu_build == 2 || (gpu_build == 1 && nP_total >= RT_GPU_BUILD_MIN_PRIMS && !getenv("RT_HOST_BUILD")));
    bool device_done = false;
    int maxDepth = 0;
    if (device_path) {
        const int64_t nT = nT_total, nS = d->n_spheres, nP = nP_total;
        for (int64_t t = 0; t < d->n_triangles && !d->geometry_on_device; ++t) {      // the range checks of the host loops below (device-resident
            const int32_t *iv = d->tri_vert + 3 * t;                                      // geometry is checked by rt_ingest.hip)
            const int32_t ob = d->tri_object[t];
            if (ob < 0 || ob >= d->n_objects) { err = "triangle references a missing object"; return RT_ERR_INVALID; }
            for (int k = 0; k < 3; ++k) if (iv[k] < 0 || iv[k] >= d->n_vertices) { err = "triangle references a missing vertex"; return RT_ERR_INVALID; }
        }
        std::vector<DevPrim> sphereRec((size_t)nS);
        for (int64_t s = 0; s < nS; ++s) {                                   // Sphere.cpp:9-21
            const int32_t ob = d->sph_object[s];
            if (ob < 0 || ob >= d->n_objects) { err = "sphere references a missing object"; return RT_ERR_INVALID; }
            DevPrim &r = sphereRec[(size_t)s];
            std::memset(&r, 0, sizeof(r));
            r.p0[0] = d->sph_center_radius[4 * s]; r.p0[1] = d->sph_center_radius[4 * s + 1]; r.p0[2] = d->sph_center_radius[4 * s + 2];
            r.e1[0] = d->sph_center_radius[4 * s + 3];
            r.object = ob; r.shape = (int32_t)s | RT_PRIM_SPHERE; r.kind = DEV_PRIM_SPHERE;
        }
        for (int64_t i = 0; i < nP && !d->prim_runs; ++i) {               // (runs were checked by compact_totals)
            const int32_t ref = d->prim_shape[i];
            if ((ref & RT_PRIM_INDEX_MASK) >= ((ref & RT_PRIM_SPHERE) ? nS : nT)) { err = "primitive references a missing shape"; return RT_ERR_INVALID; }
        }
        lap("tables, range checks");
        RtGpuIngest ing;
        const int irc = rt_ingest_gpu(d, nT_total, nP_total, hs.objects, sphereRec, ing, err);
        if (irc < 0) return irc;
        if (irc == 1 && compact) {        // a vertex shared by two smooth objects: the host loops, on the written-out scene
            HostExpanded x;
            rt_scene_desc dd;
            const int rc = expand_on_host(d, x, dd, err);
            if (rc != RT_OK) return rc;
            g_force_host_ingest = true;
            const int rc2 = rt_build_host_scene(&dd, hs, err, gpu_build, device_out);
            g_force_host_ingest = false;
            return rc2;
        }
        if (irc == RT_OK) {
            const bool ok = rt_bvh_gpu_build_flatten(ing.pbox, nP, ing.shape_rec, nT + nS, ing.pshape, hs.objects, *device_out, err, /*inputs_on_device=*/true);
            if (!ok) { ing.release(); return RT_ERR_HIP; }
            device_out->verts = ing.verts; device_out->tri_vert = ing.tri_vert; device_out->n_verts = d->n_vertices; device_out->n_tris = nT;
            ing.verts = nullptr; ing.tri_vert = nullptr;
            ing.release();
            maxDepth = device_out->max_depth;
            device_done = true;
            hs.verts.clear(); hs.tri_vert.clear();
            lap("device ingest + build");
        }
    }

    std::vector<DevPrim> shapeRec;
    std::vector<int32_t> pshape, order;
    std::vector<BuildNode> bn;
    const int64_t nP = d->n_prims;
    if (!device_done) {
    // ---- triangles: edges, face normal, object box; smooth vertex normals --------------------------
    const int64_t nT = d->n_triangles, nS = d->n_spheres;
    shapeRec.assign((size_t)(nT + nS), DevPrim());
    std::vector<Aabb> shapeBox(nT + nS);
    hs.verts.assign(d->n_vertices, DevVertex());
    for (int64_t v = 0; v < d->n_vertices; ++v) {
        DevVertex &dv = hs.verts[v];
        dv.n[0] = dv.n[1] = dv.n[2] = 0.0f;
        dv.uv[0] = d->vert_uv ? d->vert_uv[2 * v] : 0.0f;
        dv.uv[1] = d->vert_uv ? d->vert_uv[2 * v + 1] : 0.0f;
        dv._pad[0] = dv._pad[1] = dv._pad[2] = 0.0f;
    }
    hs.tri_vert.assign(d->tri_vert, d->tri_vert + 3 * nT);
    for (int64_t t = 0; t < nT; ++t) {
        const int32_t *iv = d->tri_vert + 3 * t;
        int32_t ob = d->tri_object[t];
        if (ob < 0 || ob >= d->n_objects) { err = "triangle references a missing object"; return RT_ERR_INVALID; }
        for (int k = 0; k < 3; ++k) if (iv[k] < 0 || iv[k] >= d->n_vertices) { err = "triangle references a missing vertex"; return RT_ERR_INVALID; }
        f3 a = ld3(d->vert_pos + 3 * iv[0]), b = ld3(d->vert_pos + 3 * iv[1]), c = ld3(d->vert_pos + 3 * iv[2]);
        f3 ab = a - b, ac = a - c;                                   // p0p1, p0p2 (Triangle.cpp:63-64)
        f3 fn = normalize3(cross3(-ab, -ac));                        // Triangle.cpp:66
        DevPrim &r = shapeRec[t];
        r.p0[0] = a.x; r.p0[1] = a.y; r.p0[2] = a.z;
        r.e1[0] = ab.x; r.e1[1] = ab.y; r.e1[2] = ab.z;
        r.e2[0] = ac.x; r.e2[1] = ac.y; r.e2[2] = ac.z;
        r.nrm[0] = fn.x; r.nrm[1] = fn.y; r.nrm[2] = fn.z;
        r.object = ob; r.shape = (int32_t)t; r.kind = DEV_PRIM_TRIANGLE; r.ill = tri_ill_conditioned(ab, ac) ? 1 : 0;
        shapeBox[t] = box_of(max3(a, max3(b, c)), min3(a, min3(b, c)));   // Triangle.cpp:68-69
    }
    {   // Mesh.cpp:88-95 + Triangle.cpp:78-105: accumulate face normals per smooth object in triangle order, then
        // normalise every touched vertex once.
        std::vector<std::vector<int64_t>> lists(d->n_objects);
        for (int64_t t = 0; t < nT; ++t)
            if (hs.objects[d->tri_object[t]].flags & DEV_OBJ_SMOOTH) lists[d->tri_object[t]].push_back(t);
        std::vector<uint8_t> pending(d->n_vertices, 0);
        for (const auto &list : lists) {
            for (int64_t t : list)
                for (int k = 0; k < 3; ++k) {
                    DevVertex &dv = hs.verts[d->tri_vert[3 * t + k]];
                    dv.n[0] += shapeRec[t].nrm[0]; dv.n[1] += shapeRec[t].nrm[1]; dv.n[2] += shapeRec[t].nrm[2];
                    pending[d->tri_vert[3 * t + k]] = 1;
                }
            for (int64_t t : list)
                for (int k = 0; k < 3; ++k) {
                    int32_t vi = d->tri_vert[3 * t + k];
                    if (!pending[vi]) continue;
                    f3 n = normalize3(ld3(hs.verts[vi].n));
                    hs.verts[vi].n[0] = n.x; hs.verts[vi].n[1] = n.y; hs.verts[vi].n[2] = n.z;
                    pending[vi] = 0;
                }
        }
    }
    for (int64_t s = 0; s < nS; ++s) {                                   // Sphere.cpp:9-21
        int32_t ob = d->sph_object[s];
        if (ob < 0 || ob >= d->n_objects) { err = "sphere references a missing object"; return RT_ERR_INVALID; }
        f3 c = ld3(d->sph_center_radius + 4 * s);
        float rad = d->sph_center_radius[4 * s + 3];
        DevPrim &r = shapeRec[nT + s];
        std::memset(&r, 0, sizeof(r));
        r.p0[0] = c.x; r.p0[1] = c.y; r.p0[2] = c.z; r.e1[0] = rad;
        r.object = ob; r.shape = (int32_t)s | RT_PRIM_SPHERE; r.kind = DEV_PRIM_SPHERE;
        shapeBox[nT + s] = box_of(c + mk3(-rad, -rad, -rad), c + mk3(rad, rad, rad));
    }

    lap("tables, triangles, normals");
    // ---- reference primitives and their world boxes (Q5, Q29) -------------------------------------
    std::vector<Aabb> pbox(nP);
    pshape.assign((size_t)nP, 0);              // index into shapeRec
    for (int64_t i = 0; i < nP; ++i) {
        int32_t ref = d->prim_shape[i];
        int64_t idx = ref & RT_PRIM_INDEX_MASK;
        bool sph = (ref & RT_PRIM_SPHERE) != 0;
        if (idx >= (sph ? nS : nT)) { err = "primitive references a missing shape"; return RT_ERR_INVALID; }
        int64_t si = sph ? nT + idx : idx;
        const DevObject &ob = hs.objects[shapeRec[si].object];
        Aabb wb = (ob.flags & DEV_OBJ_HAS_XFORM) ? transform_box(ob.M, shapeBox[si]) : shapeBox[si];
        if ((ref & RT_PRIM_BLURBOX) && (ob.flags & DEV_OBJ_MOTION)) {   // Shape::AdaptWorldBoundingBoxForMotionBlur
            float moved[16];
            mat_translate_mul(ld3(ob.mblur), ob.M, moved);
            wb = merge(wb, transform_box(moved, shapeBox[si]));
        }
        pbox[i] = wb;
        pshape[i] = (int32_t)si;
    }

    lap("primitive boxes");
    // ---- BVH build (BVHTree.cpp:39-153), explicit work stack, nodes created in preorder ------------
    order.assign((size_t)nP, 0);
    for (int64_t i = 0; i < nP; ++i) order[i] = (int32_t)i;
    bool built = false;
    if (want_gpu && device_out && !getenv("RT_HOST_FLATTEN")) {
        // build AND flatten on the device: the traversal arrays never exist on the host
        static_assert(sizeof(Aabb) == 6 * sizeof(float), "Aabb must be six packed floats (lo, hi)");
        if (!rt_bvh_gpu_build_flatten(reinterpret_cast<const float *>(pbox.data()), nP, shapeRec.data(), (int64_t)shapeRec.size(), pshape.data(), hs.objects, *device_out, err)) return RT_ERR_HIP;
        maxDepth = device_out->max_depth;
        built = device_done = true;
    }
    if (!built && nP > 0 && (gpu_build == 2 || (gpu_build == 1 && nP >= RT_GPU_BUILD_MIN_PRIMS && !getenv("RT_HOST_BUILD")))) {
        static_assert(sizeof(Aabb) == 6 * sizeof(float), "Aabb must be six packed floats (lo, hi)");
        std::vector<RtBuildNode> gn;
        if (!rt_bvh_gpu_build(reinterpret_cast<const float *>(pbox.data()), nP, order, gn, maxDepth, err)) return RT_ERR_HIP;
        bn.resize(gn.size());
        for (size_t i = 0; i < gn.size(); ++i) {
            bn[i].box.lo = ld3(gn[i].lo); bn[i].box.hi = ld3(gn[i].hi);
            bn[i].left = gn[i].left; bn[i].right = gn[i].right; bn[i].start = gn[i].start; bn[i].end = gn[i].end; bn[i].depth = gn[i].depth;
        }
        built = true;
    }
    if (nP > 0 && !built) {
        bn.reserve((size_t)nP);
        std::vector<Task> work;
        bn.push_back(BuildNode());
        work.push_back(Task{0, 0, (int32_t)nP, 1});
        while (!work.empty()) {
            Task tk = work.back();
            work.pop_back();
            maxDepth = std::max(maxDepth, tk.depth);
            const int count = tk.end - tk.start;
            bn[tk.node].start = tk.start; bn[tk.node].end = tk.end; bn[tk.node].left = bn[tk.node].right = -1; bn[tk.node].depth = tk.depth;
            if (count <= 1) { bn[tk.node].box = pbox[order[tk.start]]; continue; }
            Aabb all = empty_box(), cen = empty_box();
            for (int i = tk.start; i < tk.end; ++i) {
                all = merge(all, pbox[order[i]]);
                cen = merge(cen, centroid(pbox[order[i]]));
            }
            int ax;
            float clo, chi;
            if (!bvh_split_axis(cen, ax, clo, chi)) { bn[tk.node].box = all; continue; }
            int binCount[kBins] = {0, 0, 0, 0};
            Aabb binBox[kBins] = {empty_box(), empty_box(), empty_box(), empty_box()};
            for (int i = tk.start; i < tk.end; ++i) {
                int bin = bvh_bin_raw(pbox[order[i]], ax, clo, chi);
                if (bin == kBins) --bin;
                binCount[bin]++;
                binBox[bin] = merge(binBox[bin], pbox[order[i]]);
            }
            int left_count = 0;
            const int split = bvh_choose_split(binCount, binBox, all, count, left_count);
            if (split < 0) { bn[tk.node].box = all; continue; }
            int32_t *mid = std::partition(&order[tk.start], &order[tk.end - 1] + 1, [&](int32_t p) {
                return bvh_bin_raw(pbox[p], ax, clo, chi) <= split;
            });
            const int32_t m = (int32_t)(mid - &order[0]);
            const int32_t l = (int32_t)bn.size();
            bn.push_back(BuildNode());
            bn.push_back(BuildNode());
            bn[tk.node].left = l; bn[tk.node].right = l + 1;
            work.push_back(Task{l + 1, m, tk.end, tk.depth + 1});     // right is built after the whole left subtree
            work.push_back(Task{l, tk.start, m, tk.depth + 1});
        }
        for (int64_t i = (int64_t)bn.size() - 1; i >= 0; --i)          // BuildInternal: box = Merge(left, right)
            if (bn[i].left >= 0) bn[i].box = merge(bn[bn[i].left].box, bn[bn[i].right].box);
    }

    }

    lap("reference BVH build");
    // ---- flatten ------------------------------------------------------------------------------------
    hs.nodes.clear(); hs.chain.clear(); hs.node_chain.clear(); hs.leaves.clear(); hs.prims.clear(); hs.ref_leaf_range.clear(); hs.leaf_boxes.clear();
    hs.ref_prim_order.clear(); hs.cone_words.clear(); hs.cone_ranges.clear();
    if (!device_done) {
        hs.ref_prim_order.resize(nP);
        for (int64_t i = 0; i < nP; ++i) hs.ref_prim_order[i] = shapeRec[pshape[order[i]]].shape;
    }
    std::memset(&hs.params, 0, sizeof(hs.params));
    hs.params.root = RT_NO_ROOT;
    auto store3 = [](float *dst, f3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; };
    if (device_done) {
        hs.ref_prim_order.clear();
        hs.params.root = device_out->root;
        for (int k = 0; k < 3; ++k) { hs.params.root_min[k] = device_out->root_lo[k]; hs.params.root_max[k] = device_out->root_hi[k]; }
        hs.params.stack_bound = device_out->stack_bound;
    }
    if (!bn.empty()) {
        // DFS numbering of leaves (left to right); primitive records in that order
        int wideDepth = 0;
        std::vector<int32_t> ref(bn.size(), 0);
        // normal cone of every reference node's subtree (rt_cone.h), bottom-up: children are created after their parent
        std::vector<NCone> cone(bn.size());
        for (int64_t n = (int64_t)bn.size() - 1; n >= 0; --n) {
            if (bn[n].left >= 0) { cone[n] = cone_merge(cone[bn[n].left], cone[bn[n].right]); continue; }
            NCone c = cone_empty();
            for (int i = bn[n].start; i < bn[n].end; ++i) {
                const DevPrim &rec = shapeRec[pshape[order[i]]];
                if (rec.kind != DEV_PRIM_TRIANGLE) continue;
                const DevObject &ob = hs.objects[rec.object];
                c = cone_merge(c, cone_of_triangle(ld3(rec.e1), ld3(rec.e2), (ob.flags & DEV_OBJ_HAS_XFORM) ? ob.inv : nullptr, ob.cone_k));
            }
            cone[n] = c;
        }
        std::vector<int32_t> stack;
        stack.push_back(0);
        while (!stack.empty()) {
            int32_t n = stack.back();
            stack.pop_back();
            if (bn[n].left < 0) {
                DevLeaf lf;
                lf.start = (int32_t)hs.prims.size();
                for (int i = bn[n].start; i < bn[n].end; ++i) {
                    const DevPrim &rec = shapeRec[pshape[order[i]]];
                    bool dup = false;      // a second reference to the same shape inside one leaf can never win: same t, and the leaf rule needs t < best - 0.001
                    for (size_t k = lf.start; k < hs.prims.size() && !dup; ++k) dup = hs.prims[k].shape == rec.shape;
                    if (!dup) hs.prims.push_back(rec);
                }
                lf.count = (int32_t)hs.prims.size() - lf.start;
                // leaf reference: ~(count << 26 | start) so that the traversal needs no DevLeaf fetch; leaves that do not
                // fit (more than 15 primitives, start >= 2^26) keep count = 0 and carry the leaf index instead
                if (lf.count >= 1 && lf.count <= DEV_LEAF_MAX_PACKED && lf.start < (1 << DEV_LEAF_START_BITS))
                    ref[n] = ~(int32_t)(((uint32_t)lf.count << DEV_LEAF_START_BITS) | (uint32_t)lf.start);
                else {
                    if (hs.leaves.size() >= (size_t)(1 << DEV_LEAF_START_BITS)) { err = "too many BVH leaves for the packed leaf reference"; return RT_ERR_TOO_LARGE; }
                    ref[n] = ~(int32_t)hs.leaves.size();
                }
                hs.leaves.push_back(lf);
                hs.ref_leaf_range.push_back(bn[n].start); hs.ref_leaf_range.push_back(bn[n].end);
                const Aabb &b = bn[n].box;
                const float bb[6] = {b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z};
                hs.leaf_boxes.insert(hs.leaf_boxes.end(), bb, bb + 6);
            } else {
                stack.push_back(bn[n].right);
                stack.push_back(bn[n].left);
            }
        }
        // one chain record per reference node
        hs.chain.resize(bn.size());
        for (size_t n = 0; n < bn.size(); ++n) {
            DevChain &c = hs.chain[n];
            store3(c.lo, bn[n].box.lo); store3(c.hi, bn[n].box.hi);
            c._pad = 0;
            if (n == 0) c.parent = -1;
            if (bn[n].left >= 0) { hs.chain[bn[n].left].parent = (int32_t)n; hs.chain[bn[n].right].parent = (int32_t)n; }
        }
        // 4-wide collapse: the children of a wide node are reference nodes; the internal child with the largest surface
        // area is opened until four children are reached.  Wide nodes are numbered in preorder, siblings adjacent.
        if (bn[0].left >= 0) {
            struct WTask { int32_t bnode, windex, depth; };
            std::vector<WTask> wwork;
            hs.nodes.push_back(DevNode4());
            wwork.push_back(WTask{0, 0, 1});
            while (!wwork.empty()) {
                const WTask tk = wwork.back();
                wwork.pop_back();
                wideDepth = std::max(wideDepth, tk.depth);
                int32_t kids[4];
                int nk = 2;
                kids[0] = bn[tk.bnode].left; kids[1] = bn[tk.bnode].right;
                while (nk < 4) {
                    int pick = -1;
                    float area = -1.0f;
                    for (int i = 0; i < nk; ++i)
                        if (bn[kids[i]].left >= 0) { const float a = surface_area(bn[kids[i]].box); if (a > area) { area = a; pick = i; } }
                    if (pick < 0) break;
                    const int32_t open = kids[pick];
                    for (int i = nk; i > pick + 1; --i) kids[i] = kids[i - 1];      // keep left-to-right order
                    kids[pick] = bn[open].left; kids[pick + 1] = bn[open].right;
                    ++nk;
                }
                int32_t wref[4];
                for (int i = nk - 1; i >= 0; --i) {        // allocate indices so that child 0's subtree follows first (preorder)
                    if (bn[kids[i]].left < 0) wref[i] = ref[kids[i]];
                    else { wref[i] = 0; }
                }
                for (int i = 0; i < nk; ++i)
                    if (bn[kids[i]].left >= 0) { wref[i] = (int32_t)hs.nodes.size(); hs.nodes.push_back(DevNode4()); }
                for (int i = nk - 1; i >= 0; --i)
                    if (bn[kids[i]].left >= 0) wwork.push_back(WTask{kids[i], wref[i], tk.depth + 1});
                DevNode4 &w = hs.nodes[tk.windex];
                if (hs.node_chain.size() < 4 * hs.nodes.size()) hs.node_chain.resize(4 * hs.nodes.size(), -1);
                for (int i = 0; i < 4; ++i) {
                    if (i < nk) {
                        const Aabb &b = bn[kids[i]].box;
                        w.minx[i] = b.lo.x; w.miny[i] = b.lo.y; w.minz[i] = b.lo.z; w.maxx[i] = b.hi.x; w.maxy[i] = b.hi.y; w.maxz[i] = b.hi.z;
                        w.ref[i] = wref[i]; w.cone[i] = cone_pack(cone[kids[i]]);
                        hs.node_chain[4 * (size_t)tk.windex + i] = kids[i];
                        hs.cone_words.push_back(w.cone[i]); hs.cone_ranges.push_back(bn[kids[i]].start); hs.cone_ranges.push_back(bn[kids[i]].end);
                    } else {
                        w.minx[i] = w.miny[i] = w.minz[i] = 0.0f; w.maxx[i] = w.maxy[i] = w.maxz[i] = 0.0f;
                        w.ref[i] = DEV_REF_EMPTY; w.cone[i] = RT_CONE_WORD_NEVER;
                        hs.node_chain[4 * (size_t)tk.windex + i] = -1;
                    }
                }
            }
            ref[0] = 0;
        }
        hs.params.stack_bound = 3 * wideDepth + 2;     // a wide node pushes at most three children
        hs.params.root = ref[0];
        store3(hs.params.root_min, bn[0].box.lo); store3(hs.params.root_max, bn[0].box.hi);
    }
    hs.params.max_depth = d->max_depth; hs.params.isect_eps = d->isect_eps; hs.params.shadow_eps = d->shadow_eps;
    { volatile float one = 1.0f; hs.params.isect_eps_p1 = one + d->isect_eps; }
    for (int k = 0; k < 3; ++k) { hs.params.background[k] = d->background[k]; hs.params.ambient[k] = d->ambient[k]; }
    hs.params.bg_texture = d->bg_texture;
    hs.params.n_lights = d->n_lights; hs.params.n_materials = d->n_materials; hs.params.n_textures = d->n_textures;
    hs.params.bvh_depth = maxDepth;
    hs.params.flt_max = 3.402823466e+38f;
    hs.params.interleave_shift = 0;             // measured on config 2 (1 spp): runs of 1 / 4 / 8 / 16 neighbours: 9.20 / 9.23 / 9.35 / 9.41 ms, none: 9.73
    hs.params.interleave_max_share = 1024;      // launches above ~16 rays per lane balance through the queue; interleaving them costs 4 % (coherence)
    hs.params.tri_min_lanes = RT_TRI_MIN_LANES;
    hs.params.donate = 1;
    hs.params.tl_slot = -1;
    hs.params.tail_window_waves = 1;
    hs.params.refill_min_idle = RT_REFILL_MIN_IDLE;
    hs.params.donate_min_idle = RT_DONATE_MIN_IDLE;
    hs.params.has_normal_tex = 0; hs.params.has_rng = 0;
    for (const DevObject &o : hs.objects) if (o.tex_normal >= 0) hs.params.has_normal_tex = 1;
    for (const rt_light &l : hs.lights) if (l.type == RT_LIGHT_AREA) hs.params.has_rng = 1;
    for (const rt_material &m : hs.materials) if (m.roughness > 0) hs.params.has_rng = 1;
    hs.params.has_motion = 0;
    hs.params.n_objects = (int32_t)hs.objects.size();
    hs.params.unit_object = -1;
    for (size_t i = 0; i < hs.objects.size() && i < 24; ++i) if (hs.objects[i].flags & DEV_OBJ_UNIT) { hs.params.unit_object = (int32_t)i; break; }
    for (const DevObject &o : hs.objects) if (o.flags & DEV_OBJ_MOTION) hs.params.has_motion = 1;
    float scale = 1.0f;
    if (!bn.empty() || device_done)
        for (int k = 0; k < 3; ++k) scale = std::max(scale, std::max(std::fabs(hs.params.root_min[k]), std::fabs(hs.params.root_max[k])));
    hs.scene_scale = scale;
    hs.params.prune_rel = 2e-4f;
    hs.params.prune_err = (4.0f * d->isect_eps + 2e-4f) * scale;   // eps*edge slack of the Cramer test + float error
    hs.info.n_nodes = (int64_t)bn.size(); hs.info.n_leaves = (int64_t)hs.leaves.size(); hs.info.max_depth = maxDepth; hs.info.n_wide_nodes = (int32_t)hs.nodes.size();
    hs.info.n_leaf_prims = (int64_t)hs.prims.size();
    if (device_done) {
        hs.info.n_nodes = device_out->n_nodes; hs.info.n_leaves = device_out->n_leaves; hs.info.n_wide_nodes = (int32_t)device_out->n_wide;
        hs.info.n_leaf_prims = device_out->n_leaf_prims;
    }
    lap("flatten (leaves, wide nodes)");
    return RT_OK;
}
