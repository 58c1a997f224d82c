// rt_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the render hot path.
//
//   k_raygen       Camera::GenerateRay / GenerateRayForPixelSample, JitteredPixelSampler, MultiSampledRayGenerator
//                  (src/Camera.cpp:76-147, src/PixelSampler.cpp:27-38, src/MultiSampledRayGenerator.cpp:25-33)
//   k_trace*       BVHTree::Intersect and everything under it (src/BVHTree.cpp:155-206, src/acmath.h:770-820,
//                  src/Triangle.cpp:115-302, src/Sphere.cpp:31-131, src/Shape.cpp:67-82)
//   k_shade_pre    normal tweak + shadow-ray generation (src/Intersection.cpp:25-29, src/LightContributionCalculator.cpp:114-145,
//                  src/Light.cpp:89-129)
//   k_shade_post   lights, BRDFs, Fresnel, child-ray generation with wave64 ballot compaction
//                  (src/LightContributionCalculator.cpp:46-124, src/Light.cpp:37-87, src/brdf.cpp, src/Material.cpp,
//                  src/RecursiveComputation.cpp:8-228)
//   k_backward     the "+=" side of RecursiveComputation::Perform{Reflection,Refraction}/AddDielectricConsumption
//   k_resolve      RenderMultiSampled's mean (src/DefaultRenderer.cpp:106-138)
//
// No MFMA anywhere: the path is pointer-chasing + scalar-per-lane float math, bound by L2/HBM latency and VALU issue.
// Wave64 throughout; the traversal stack lives in LDS (one 4-byte column per lane, bank == lane, conflict-free).
#include <float.h>

#include <algorithm>
#include <vector>

#include "rt_cone.h"
#include "rt_math.h"
#include "rt_wavefront.h"

#define TRACE_BLOCK 256
#define LDS_STACK 20
#define TRACE_RESIDENT_PER_CU 4      // blocks of the traversal kernel a CU holds (128 VGPRs, 39 KB LDS each of the 160 KB)

// compute units of the device the calling thread's context runs on (hipDeviceProp_t::multiProcessorCount; 256 on MI355X): sizes the
// persistent grid of the traversal and its stack-overflow area.  Every entry point of rt_api.hip sets it from its context, so that
// contexts on devices of different size (or on different host threads) never share the value.
static thread_local int g_num_cus = 256;
void rt_set_num_cus(int n) { if (n > 0) g_num_cus = n; }
int rt_trace_grid_threads() { return g_num_cus * TRACE_RESIDENT_PER_CU * TRACE_BLOCK; }
int rt_trace_lds_stack_depth() { return LDS_STACK; }

// ====================================================================================================
// Counter-based RNG: Philox4x32-10, key = (seed, pixel), counter = (sample, path, slot, 0).
// ====================================================================================================
enum { SLOT_PIXEL = 0, SLOT_TIME = 1, SLOT_GLOSSY = 2, SLOT_LIGHT0 = 3 };

__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// the four 24-bit uniforms of one slot, as doubles in [0,1)
__device__ __forceinline__ void rng_slot(unsigned seed, unsigned pixel, unsigned sample, unsigned path, unsigned slot, double u[4]) {
    unsigned w[4];
    philox4x32(sample, path, slot, 0u, seed, pixel, w);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = (double)(w[i] >> 8) * (1.0 / 16777216.0);
}
// std::uniform_real_distribution<double>(a,b): u*(b-a)+a
__device__ __forceinline__ double uni(double u, double a, double b) { return u * (b - a) + a; }

// ====================================================================================================
// Traversal
// ====================================================================================================
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) float lds_float;
// rt_math.h's mat * (x, y, z, w) and transpose(mat) * (x, y, z, w) for a matrix in LDS (same association)
__device__ __forceinline__ f3 mat_mul_xyz(const lds_float *m, float x, float y, float z, float w) {
    f3 r;
    r.x = (m[0] * x + m[4] * y) + (m[8] * z + m[12] * w);
    r.y = (m[1] * x + m[5] * y) + (m[9] * z + m[13] * w);
    r.z = (m[2] * x + m[6] * y) + (m[10] * z + m[14] * w);
    return r;
}
__device__ __forceinline__ f3 matT_mul_xyz(const lds_float *m, float x, float y, float z, float w) {
    f3 r;
    r.x = (m[0] * x + m[1] * y) + (m[2] * z + m[3] * w);
    r.y = (m[4] * x + m[5] * y) + (m[6] * z + m[7] * w);
    r.z = (m[8] * x + m[9] * y) + (m[10] * z + m[11] * w);
    return r;
}

struct Cand {
    float t;       // x-only world parameter (Q2)
    f3 ip;         // world intersection point
    float beta, gamma;
    int prim;      // DevPrim record index, -1 = none
};

struct ObjXf {     // per-lane cache of "the ray in the space of object `id`" (Q8: identical for all triangles of a mesh)
    int id;
    int flags;     // DEV_OBJ_* of the object, DEV_OBJ_IDENTITY re-evaluated for motion-blurred objects
    f3 o, d;       // object-space ray
};

// Matrices in effect for `ob` at ray time: M (object->world) and inv = inverse(transpose(M)).
__device__ __forceinline__ void object_matrices(const DevObject &ob, float time, float *M, float *inv, int &flags) {
    flags = ob.flags;
    if (ob.flags & DEV_OBJ_MOTION) {                                  // Shape.cpp:71-79, Triangle.cpp:168-175
        f3 mv = ld3(ob.mblur) * time;
        mat_translate_mul(mv, ob.M, M);
        float Mt[16];
        mat_transpose(M, Mt);
        mat_inverse(Mt, inv);
        flags = (flags & ~DEV_OBJ_IDENTITY) | (mat_is_identity(Mt) ? DEV_OBJ_IDENTITY : 0);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) { M[i] = ob.M[i]; inv[i] = ob.inv[i]; }
    }
}

// moving object: the matrices are rebuilt for the ray's time (kept out of line: inverse + transposes)
__device__ __noinline__ void enter_object_motion(const DevScene &sc, int obj, f3 o, f3 d, float time, ObjXf *out) {
    const DevObject &ob = sc.objects[obj];
    float M[16], inv[16];
    int flags;
    object_matrices(ob, time, M, inv, flags);
    out->id = obj; out->flags = flags;
    out->o = matT_mul_xyz(inv, o.x, o.y, o.z, 1.0f);
    out->d = normalize3(matT_mul_xyz(inv, d.x, d.y, d.z, 0.0f));
}
// The traversal keeps the records of the first OBJ_CACHE objects in LDS (OBJ_CACHE_DWORDS each: M, inv, flags): entering an
// object and taking a hit point back to world space then cost an LDS gather instead of a dependent trip to L2 in the middle of
// the primitive path (config 2: nearly every primitive-path iteration has a lane that enters an object).
#define OBJ_CACHE 24
#define OBJ_CACHE_DWORDS 33
template <bool MOTION>
__device__ __forceinline__ void enter_object(const DevScene &sc, const lds_float *objc, int obj, f3 o, f3 d, float time, ObjXf &x) {
    if (obj < OBJ_CACHE) {
        const lds_float *c = objc + obj * OBJ_CACHE_DWORDS;
        const int flags = __float_as_int(c[32]);
        if (!(MOTION && (flags & DEV_OBJ_HAS_XFORM) && (flags & DEV_OBJ_MOTION))) {
            x.id = obj;
            if (!(flags & DEV_OBJ_HAS_XFORM)) { x.flags = flags | DEV_OBJ_IDENTITY; x.o = o; x.d = d; return; }
            x.flags = flags;
            x.o = matT_mul_xyz(c + 16, o.x, o.y, o.z, 1.0f);
            x.d = normalize3(matT_mul_xyz(c + 16, d.x, d.y, d.z, 0.0f));
            return;
        }
    }
    const DevObject &ob = sc.objects[obj];
    const int flags = ob.flags;
    if (MOTION && (flags & DEV_OBJ_HAS_XFORM) && (flags & DEV_OBJ_MOTION)) {
        ObjXf tmp;                                                    // the only ObjXf that lives in memory
        enter_object_motion(sc, obj, o, d, time, &tmp);
        x.id = tmp.id; x.flags = tmp.flags; x.o = tmp.o; x.d = tmp.d;
        return;
    }
    x.id = obj;
    if (!(flags & DEV_OBJ_HAS_XFORM)) {                               // standalone <Triangle>: objTransform == nullptr (Shape.cpp:69)
        x.flags = flags | DEV_OBJ_IDENTITY;
        x.o = o; x.d = d;
        return;
    }
    // Transform::operator()(Ray, obj2world=false): transpose(inv) * (o,1), transpose(inv) * (d,0), renormalise (acmath.h:1053-1069)
    x.flags = flags;
    x.o = matT_mul_xyz(ob.inv, o.x, o.y, o.z, 1.0f);
    x.d = normalize3(matT_mul_xyz(ob.inv, d.x, d.y, d.z, 0.0f));
}

// object -> world for a hit point of object `obj` (glm mat*vec4(p,1): (m0*x + m1*y) + (m2*z + m3*1)); the matrix is read
// from the object table (static objects) or rebuilt for the ray's time (moving objects) — only hit candidates of
// transformed objects come here.
// (the path through the object table loads whole columns: with the same scalar loads as the LDS path the compiler merges the two
// into FLAT loads, whose wait also waits for the node record that is in flight)
template <bool MOTION>
__device__ __forceinline__ f3 object_point_to_world_far(const DevObject *objects, int obj, float time, f3 p) {
    const DevObject &ob = objects[obj];
    const float4 *m4 = reinterpret_cast<const float4 *>(ob.M);
    const float4 c0 = m4[0], c1 = m4[1], c2 = m4[2], c3 = m4[3];
    const float Ms[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
    if (!MOTION || !(ob.flags & DEV_OBJ_MOTION)) return mat_mul_xyz(Ms, p.x, p.y, p.z, 1.0f);
    float M[16];
    mat_translate_mul(ld3(ob.mblur) * time, Ms, M);
    return mat_mul_xyz(M, p.x, p.y, p.z, 1.0f);
}
template <bool MOTION>
__device__ __forceinline__ f3 object_point_to_world(const DevScene &sc, const lds_float *objc, int obj, float time, f3 p) {
    if (obj < OBJ_CACHE) {
        const lds_float *c = objc + obj * OBJ_CACHE_DWORDS;
        if (!MOTION || !(__float_as_int(c[32]) & DEV_OBJ_MOTION)) return mat_mul_xyz(c, p.x, p.y, p.z, 1.0f);
    }
    return object_point_to_world_far<MOTION>(sc.objects, obj, time, p);
}

__device__ __forceinline__ float det3(float ax, float ay, float az, float bx, float by, float bz, float cx, float cy, float cz) {
    // r1=(ax,ay,az) r2=(bx,by,bz) r3=(cx,cy,cz); Triangle.cpp:197 association
    return ax * (by * cz - bz * cy) + bx * (az * cy - ay * cz) + cx * (ay * bz - az * by);
}

// Triangle::Intersect up to the world-space t (Triangle.cpp:115-135, 186-250; Q2, Q6, Q8)
template <bool MOTION>
__device__ __forceinline__ bool hit_triangle(const DevScene &sc, const lds_float *objc, const float4 q0, const float4 q1, const float q2x, const ObjXf &x, float wox, float wdx,
                                             float time, float eps, float err, Cand &c) {
    const f3 p0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2x);
    const f3 d = x.d;
    const f3 pr = mk3(p0.x - x.o.x, p0.y - x.o.y, p0.z - x.o.z);
    const float detM = det3(e1.x, e2.x, d.x, e1.y, e2.y, d.y, e1.z, e2.z, d.z);
    if (detM == 0.0f) return false;
    const float inv = 1 / detM;
    const float beta = det3(pr.x, e2.x, d.x, pr.y, e2.y, d.y, pr.z, e2.z, d.z) * inv;
    if (beta < -eps || beta > err) return false;
    const float gamma = det3(e1.x, pr.x, d.x, e1.y, pr.y, d.y, e1.z, pr.z, d.z) * inv;
    if (gamma < -eps || gamma > err || beta + gamma > err) return false;
    const float t = det3(e1.x, e2.x, pr.x, e1.y, e2.y, pr.y, e1.z, e2.z, pr.z) * inv;
    if (!(t >= -eps)) return false;
    f3 ip = x.o + d * t;
    if (!(x.flags & DEV_OBJ_IDENTITY)) ip = object_point_to_world<MOTION>(sc, objc, x.id, time, ip);      // Triangle.cpp:296-298
    c.t = (ip.x - wox) / wdx;                                                  // Ray::operator()(Vector3f), acmath.h:742
    c.ip = ip; c.beta = beta; c.gamma = gamma;
    return true;
}

// Sphere::CalculateTValueForIntersection (Sphere.cpp:61-91): object-space t or "no hit"
__device__ __forceinline__ bool sphere_t(f3 center, float radius, f3 o, f3 d, float &t) {
    f3 oc = o - center;
    float b = dot3(oc, d);
    float dd = dot3(d, d);
    float disc = b * b - dd * (dot3(oc, oc) - radius * radius);
    if (disc < 0) return false;
    disc = sqrtf(disc);
    t = -1 * b - disc;
    if (t < 0 || ((-b + disc) < t && -b + disc > 0)) t = -1 * b + disc;
    t /= dd;
    return t > 0;
}
template <bool MOTION>
__device__ __forceinline__ bool hit_sphere(const DevScene &sc, const lds_float *objc, const float4 q0, const ObjXf &x, float wox, float wdx, float time, Cand &c) {
    float t;
    if (!sphere_t(mk3(q0.x, q0.y, q0.z), q0.w, x.o, x.d, t)) return false;
    f3 ip = object_point_to_world<MOTION>(sc, objc, x.id, time, x.o + x.d * t);             // Sphere.cpp:128 (no identity shortcut, Q9)
    c.t = (ip.x - wox) / wdx;
    c.ip = ip; c.beta = 0; c.gamma = 0;
    return true;
}

// BoundingVolume3::Intersect (acmath.h:770-820, Q3), exact form: true divisions, the reference's ternaries.
__device__ __noinline__ bool slab_exact(f3 bmin, f3 bmax, f3 o, f3 d) {
    float tn = 0, tf = FLT_MAX;
    float a = (bmin.x - o.x) / d.x, b = (bmax.x - o.x) / d.x;
    if (a > b) { float s = a; a = b; b = s; }
    tn = sel_max(tn, a); tf = sel_min(tf, b);
    a = (bmin.y - o.y) / d.y; b = (bmax.y - o.y) / d.y;
    if (a > b) { float s = a; a = b; b = s; }
    tn = sel_max(tn, a); tf = sel_min(tf, b);
    a = (bmin.z - o.z) / d.z; b = (bmax.z - o.z) / d.z;
    if (a > b) { float s = a; a = b; b = s; }
    tn = sel_max(tn, a); tf = sel_min(tf, b);
    return !(tf < tn);
}

// Fast form of the same test for one child box of a wide node: multiplies by r = 1/d (computed once per ray with an IEEE
// division).  Every quotient is within 2 ulp of the reference's, so the hit/miss decision is the reference's unless tn and
// tf are within ~1e-6 relative of each other; then (and for any NaN) the result is "undecided".  Also returns the entry
// distance (child ordering only) and a conservative lower bound `lb` on the x-only t of anything hit inside the box.
//   returns 0 = miss, 1 = hit, 2 = undecided
//   amb = 1e-6 normally; +inf for a ray with a zero / denormal direction component (every test undecided, marg = +inf too)
__device__ __forceinline__ int slab_fast(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, f3 o, f3 r, f3 marg,
                                         float amb, float &tn_out, float &lb_out) {
    const float ax = (bminx - o.x) * r.x, bx = (bmaxx - o.x) * r.x;
    const float ay = (bminy - o.y) * r.y, by = (bmaxy - o.y) * r.y;
    const float az = (bminz - o.z) * r.z, bz = (bmaxz - o.z) * r.z;
    const float nx = fminf(ax, bx), fx = fmaxf(ax, bx);
    const float ny = fminf(ay, by), fy = fmaxf(ay, by);
    const float nz = fminf(az, bz), fz = fmaxf(az, bz);
    const float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
    const float tf = fminf(fminf(fx, fy), fminf(fz, FLT_MAX));
    // (the caller compares lb with the bound scaled by 1 + 3e-6: the relative slack of the bound, paid once per node step instead of per child)
    const float lb = fmaxf(fmaxf(nx - marg.x, ny - marg.y), nz - marg.z) - marg.x;
    tn_out = tn; lb_out = lb;
    const float diff = tf - tn;
    const float thr = amb * (fabsf(tf) + fabsf(tn));
    if (!(fabsf(diff) > thr)) return 2;
    return diff < 0.0f ? 0 : 1;
}

// Second look at a LEAF box whose fast test was undecided, still without memory traffic.  The reference's test fails iff some
// far_b < near_a (or far_b < 0).  A pair from ONE axis can never fail: near and far are the min and max of the same two
// quotients, in the reference's arithmetic as in this one — which is why every hit on a flat box (near_y == far_y, the
// config-2 floor) looks undecided to the tn/tf form although nothing about it is close.  So compare across axes only, each
// pair with the usual 1e-6 margin: a decided failure of any pair is a miss; all pairs decided passes are a hit of this box
// and, because boxes nest pair by pair, of every ancestor's; anything else stays undecided (a true graze) and takes the walk.
__device__ __forceinline__ int slab_cross(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, f3 o, f3 r, float amb) {
    const float ax = (bminx - o.x) * r.x, bx = (bmaxx - o.x) * r.x;
    const float ay = (bminy - o.y) * r.y, by = (bmaxy - o.y) * r.y;
    const float az = (bminz - o.z) * r.z, bz = (bmaxz - o.z) * r.z;
    const float nx = fminf(ax, bx), fx = fmaxf(ax, bx);
    const float ny = fminf(ay, by), fy = fmaxf(ay, by);
    const float nz = fminf(az, bz), fz = fmaxf(az, bz);
    // per axis a: near_a against the smaller far of the other two axes; and 0 against the smallest far
    const float n[4] = {nx, ny, nz, 0.0f};
    const float f[4] = {fminf(fy, fz), fminf(fx, fz), fminf(fx, fy), fminf(fminf(fx, fy), fz)};
    bool miss = false, open = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float diff = f[k] - n[k];
        const bool decided = fabsf(diff) > amb * (fabsf(f[k]) + fabsf(n[k]));
        miss = miss || (decided && diff < 0.0f);
        open = open || !decided;
    }
    return miss ? 0 : (open ? 2 : 1);
}

// Would the reference traversal reach the leaf whose reference node is `node`?  It does iff the exact box test passes
// for the leaf's box and for every ancestor's (BVHTree.cpp:160-206).  Only called for leaves whose own fast test was
// undecided (a ray grazing the leaf box within 1e-6 relative — every hit on a flat, axis-aligned leaf box is one — or a ray
// with a zero direction component).  Walks up: an undecided box gets the exact test; the first box the fast test decides
// ends the walk, because a decided hit implies that every box above it passes too (boxes nest and every step of the slab formula is monotonic).
__device__ __noinline__ bool chain_visit(const DevChain *chain, int node, bool leaf, f3 o, f3 d, f3 r, float amb) {
    if (!leaf) {                      // internal child of a zero-component ray: the reference's test of that one box
        const DevChain c = chain[node];
        return slab_exact(ld3(c.lo), ld3(c.hi), o, d);
    }
    const f3 zero = mk3(0, 0, 0);
    while (node >= 0) {
        const DevChain c = chain[node];
        float tn, lb;
        const int cls = slab_fast(c.lo[0], c.lo[1], c.lo[2], c.hi[0], c.hi[1], c.hi[2], o, r, zero, amb, tn, lb);
        if (cls == 0) return false;
        if (cls == 1) return true;
        if (!slab_exact(ld3(c.lo), ld3(c.hi), o, d)) return false;
        node = c.parent;
    }
    return true;
}

// ----------------------------------------------------------------------------------------------------
// Persistent wave64 traversal with per-lane ray refill.
//
// Rays differ wildly in traversal length (config 2: mean 25 wide-node steps + 7 primitive tests, tail > 400).  A wave is a
// persistent worker: it owns a chunk of the ray queue (one atomic per TRACE_CHUNK rays) and whenever enough lanes are idle
// they retire their results and start new rays while the other lanes continue.
//
// Every iteration a lane that holds a ray does a node step (four child boxes) and, when the wave runs its primitive path,
// one primitive test of its pending leaf.  Both records are requested at the top of the iteration, the primitive record
// first, so an iteration costs one memory round trip and the primitive test runs while the node record is in flight —
// provided nothing in between waits for ALL outstanding loads (spill reloads, FLAT loads, a hoisted use of the primitive
// record).  The expensive primitive path only runs when at least tri_min_lanes lanes hold a pending
// leaf (or no lane has a node left); retiring and refilling is batched the same way (refill_min_idle).
//
// Registers: the hit payload that is only needed at the end (ip, beta, gamma of the leaf-local and of the best candidate)
// lives in LDS next to the traversal stack.
//
// IO policy: load(i, o, d, time) -> false when slot i needs no traversal (result already written);
//            store(sc, i, best)   writes the result of ray i.
// ----------------------------------------------------------------------------------------------------
#define CUR_DONE 0x7ffffffe      // lane finished its ray (stack empty), result not yet stored
#define CUR_NONE 0x7ffffffd      // no node left to visit (stack empty); the lane may still hold a pending leaf
#define CUR_WAIT 0x7ffffffc      // lane finished its own traversal and waits for parts it gave away
#define TRACE_CHUNK 64           // rays a wave reserves per atomic
// (thresholds: DevSceneParams::refill_min_idle / tri_min_lanes, defaults RT_REFILL_MIN_IDLE / RT_TRI_MIN_LANES in rt_device.h)
#define LDS_PAYLOAD 15           // dwords per lane: ip.xyz, beta, gamma of the leaf-local candidate, then of the best one, then the
                                 // rarely used lane state below
#define PAY_RAY_IDX 10           // index of the ray in the queue
#define PAY_PENDING 11           // parts given away and not yet merged
#define PAY_DY 12                // direction y, z (the hot path only needs 1/d and d.x)
#define PAY_DZ 13
#define PAY_DLEN 14              // |d| (normal-cone test of a child that is about to be pruned by distance)

struct TraceDbg { unsigned nodes, boxes, tris, sph, v[14]; };

// Timeline builds only (make TUNE=1 TL=1; the extra live values cost the traversal loop registers, so never otherwise): per trace
// launch and wave, when the wave started, when it found the queue empty, when it left the traversal loop (100 MHz ticks).
// RT_TIMELINE=1 prints the percentiles per launch of a frame.
#ifdef RT_TIMELINE_BUILD
#define TL_SLOTS 32
#define TL_WAVES 4096
__device__ unsigned long long g_tl[TL_SLOTS * TL_WAVES * 4];
#define TL_SET(k) do { if (sc.p.tl_slot >= 0 && (threadIdx.x & 63) == 0) g_tl[((size_t)sc.p.tl_slot * TL_WAVES + (blockIdx.x * (TRACE_BLOCK / 64) + (threadIdx.x >> 6))) * 4 + (k)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define TL_SET(k) do { } while (0)
#endif

// Traversal stack: LDS_STACK entries per lane in LDS ([entry][thread], conflict-free), deeper entries in a global column.
// The overflow accesses are volatile so that the compiler keeps two separate paths (ds_* and global_*) instead of one
// FLAT access through a selected pointer, which would tie every push/pop to both memory counters.
// LDS pointers are kept in their own address space: 32-bit offsets instead of 64-bit generic pointers (two VGPRs each, which the
// register allocator spilled and reloaded on the pop path).  The overflow column is addressed as uniform base + thread index.
__device__ __forceinline__ void stack_push(lds_int *lds_col, int *ovf, int ovf_idx, int ovf_stride, int sp, int v) {
    if (sp < LDS_STACK) lds_col[sp * TRACE_BLOCK] = v;
    else *(volatile int *)(ovf + ((size_t)(sp - LDS_STACK) * ovf_stride + ovf_idx)) = v;
}
__device__ __forceinline__ int stack_pop(const lds_int *lds_col, const int *ovf, int ovf_idx, int ovf_stride, int sp) {
    if (sp < LDS_STACK) return lds_col[sp * TRACE_BLOCK];
    return *(const volatile int *)(ovf + ((size_t)(sp - LDS_STACK) * ovf_stride + ovf_idx));
}

template <bool COUNT, bool MOTION, class IO>
__device__ __forceinline__ void trace_persistent(const DevScene &sc, const IO io, int n, int *fetch_counter, lds_int *lds_col, lds_float *pay, const lds_float *objc,
                                                 int *ovf, int ovf_stride, TraceDbg &dbg, unsigned long long *hist) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const __amdgpu_buffer_rsrc_t node_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sc.nodes, 0, sc.p.node_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t prim_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sc.prims, 0, sc.p.prim_bytes, 0x00020000);
    const float eps = sc.p.isect_eps;
    const bool no_scene = sc.p.root == RT_NO_ROOT;

    bool has_ray = false;
    float amb = 1e-6f;
    int sp = 0, cur = CUR_DONE, prim_i = 0, prim_end = 0;
    lds_int *payi = (lds_int *)pay;
    const int ovf_idx = blockIdx.x * TRACE_BLOCK + threadIdx.x;      // this thread's column of the overflow area
    int best_prim = -1, local_prim = -1;
    float best_t = FLT_MAX, local_t = FLT_MAX;
    f3 o = mk3(0, 0, 0), r = mk3(0, 0, 1), marg = mk3(0, 0, 0);
    float dirx = 0;                  // d.x (Q2: every hit parameter divides by it); d.y, d.z live in LDS
    float time = 0;
    ObjXf xf;
    xf.id = -2; xf.flags = 0; xf.o = o; xf.d = o;
    unsigned ray_steps = 0;
    unsigned long long ray_t0 = 0;
    bool exhausted = false;          // wave-uniform: the global queue has no more chunks
    int chunk_next = 0, chunk_end = 0;
    // Rays a wave reserves per atomic: TRACE_CHUNK while the queue is long.  A wave that reserves 64 rays with 16 idle lanes keeps
    // 48 of them waiting for its own lanes; once the queue is within `tail_window` positions of its end those rays are what the
    // launch's last waves are still busy with while the lanes of every other wave idle, so from there on a wave reserves
    // exactly as many rays as it has idle lanes (one atomic per refill round instead of one per 64 rays).
    const int tail_window = sc.p.tail_window_waves * (int)gridDim.x * (TRACE_BLOCK / 64) * TRACE_CHUNK;
    int last_base = 0;               // queue position of the wave's latest reservation (wave-uniform)
    const int chunk = TRACE_CHUNK;
    // Queue positions -> ray indices.  Neighbouring rays are about equally long, so a launch with only a few rays per lane ends
    // when the wave that drew the longest 64 neighbours ends.  Such launches are interleaved: position 64 c + q (chunk c, slot q)
    // is ray (q >> s) * (C << s) + (c << s) + (q & (2^s - 1)) with C chunks in all, so that a wave draws 64 >> s runs of 2^s
    // neighbours from places evenly spread over the queue, every wave gets the same mix, and the idle lanes of a wave take over
    // parts of its long rays.  npos: positions to hand out (a multiple of 64 when interleaved; positions beyond n are empty).
    int il_shift = 6, il_chunks = 0, npos = n;
    {
        const int waves = gridDim.x * (TRACE_BLOCK / 64);
        const int share = (int)(((long long)n + waves - 1) / waves);
        if (sc.p.interleave_shift < 6 && share <= sc.p.interleave_max_share) { il_shift = sc.p.interleave_shift; il_chunks = (n + 63) >> 6; npos = il_chunks << 6; }
    }
    // work donation (see the retire block): a lane may hand the bottom entry of its stack to an idle lane of the wave
    int sbase = 0;                   // the stack is [sbase, sp): entries below sbase were given away
    int plane = lane;                // lane this part of a ray reports to (its own lane: it owns the ray)
    float bound = FLT_MAX;           // pruning bound: min over this lane's best_t and its parent's bound
    bool any_parts = false;          // wave-uniform: some lane works on a donated part

    for (;;) {
        if (COUNT) { ++dbg.v[0]; if (has_ray && cur != CUR_DONE) ++dbg.v[1]; }
#ifdef RT_VALU_PROBE
        // issue-slot probe: RT_VALU_PROBE x 8 vector no-ops per iteration (how much of the loop's time is VALU issue?)
#pragma unroll
        for (int q = 0; q < RT_VALU_PROBE; ++q) asm volatile("v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop");
#endif
        // ---------------- retire finished rays, refill idle lanes (batched) ----------------
        // When the queue is exhausted the wave would run on with a few long rays and mostly idle lanes (config 2: 2-6 rays per
        // lane and launch, so this phase is a large part of every launch).  Then lanes that still have stack entries DONATE
        // the bottom one (the largest pending subtree) to idle lanes: the part is traversed like a ray of its own, prunes
        // with its parent's current bound, and is merged back by the (t, leaf order) rule, which does not depend on the
        // order of visits.  A lane finishes its ray only when all parts it gave away have been merged.
        {
            if (any_parts && has_ray && cur == CUR_DONE && payi[PAY_PENDING * TRACE_BLOCK] > 0) cur = CUR_WAIT;     // retire only after the merges
            const bool working = has_ray && cur != CUR_DONE && cur != CUR_WAIT;
            bool fin = has_ray && cur == CUR_DONE;
            const unsigned long long idle0 = __ballot(!has_ray || fin);
            unsigned long long parts_done = any_parts ? __ballot(fin && plane != lane) : 0ull;
            const int min_idle = (exhausted && chunk_next >= chunk_end) ? sc.p.donate_min_idle : sc.p.refill_min_idle;
            if (__popcll(idle0) >= min_idle || parts_done || !__any(working)) {
                // 1. merge finished parts into the lanes they report to
                while (parts_done) {
                    const int s = __ffsll((long long)parts_done) - 1;
                    parts_done &= parts_done - 1;
                    const int p = __shfl(plane, s), bp = __shfl(best_prim, s);
                    const float bt = __shfl(best_t, s);
                    if (lane == p) {
                        // records are stored in leaf order: a larger record index is a later leaf (parts cover disjoint leaves)
                        if (bp >= 0 && (bt < best_t || (bt == best_t && bp > best_prim))) {
                            best_t = bt; best_prim = bp;
#pragma unroll
                            for (int k = 5; k < 10; ++k) pay[k * TRACE_BLOCK] = pay[k * TRACE_BLOCK + (s - lane)];
                        }
                        const int left = payi[PAY_PENDING * TRACE_BLOCK] - 1;
                        payi[PAY_PENDING * TRACE_BLOCK] = left;
                        if (left == 0 && cur == CUR_WAIT) cur = CUR_DONE;
                    }
                    if (lane == s) { has_ray = false; plane = lane; }
                }
                // 2. retire finished rays
                fin = has_ray && cur == CUR_DONE && plane == lane;
                if (fin) {
                    const int ray_idx = payi[PAY_RAY_IDX * TRACE_BLOCK];
                    if (COUNT && hist) {
                        const int lg = ray_steps ? 32 - __clz(ray_steps) : 0;
                        const int bk = io.kind(ray_idx) * 16 + (lg < 15 ? lg : 15);
                        atomicAdd(&hist[bk], 1ull);
                        atomicAdd(&hist[32 + bk], (unsigned long long)(wall_clock64() - ray_t0));     // 100 MHz ticks
                        atomicAdd(&hist[64 + bk], (unsigned long long)ray_steps);
                        ray_steps = 0;
                    }
                    Cand best;
                    best.t = best_t; best.prim = best_prim; best.ip = mk3(0, 0, 0); best.beta = best.gamma = 0;
                    if (best_prim >= 0) {
                        best.ip = mk3(pay[5 * TRACE_BLOCK], pay[6 * TRACE_BLOCK], pay[7 * TRACE_BLOCK]);
                        best.beta = pay[8 * TRACE_BLOCK]; best.gamma = pay[9 * TRACE_BLOCK];
                    }
                    io.store(sc, ray_idx, best);
                    has_ray = false;
                }
                const unsigned long long idle = __ballot(!has_ray);
                const int nidle = __popcll(idle);
                if (nidle >= sc.p.refill_min_idle && (!exhausted || chunk_next < chunk_end)) {
                    // 3a. new rays from the wave's chunk of the queue
                    if (chunk_next >= chunk_end) {
                        const int want = (npos - last_base <= tail_window) ? nidle : chunk;
                        int base = 0;
                        if (lane == 0) base = atomicAdd(fetch_counter, want);
                        base = __shfl(base, 0);
                        last_base = base;
                        chunk_next = base;
                        chunk_end = base + want < npos ? base + want : npos;
                        if (base + want >= npos) { exhausted = true; TL_SET(1); }
                    }
                    const int avail = chunk_end > chunk_next ? chunk_end - chunk_next : 0;
                    const int rank = __popcll(idle & lt_mask);
                    if (COUNT) { ++dbg.v[2]; if (!has_ray && rank < avail) ++dbg.v[3]; }
                    if (!has_ray && rank < avail) {
                        int mine = chunk_next + rank;
                        if (il_shift < 6) {
                            const int q = mine & 63;
                            mine = (((q >> il_shift) * il_chunks + (mine >> 6)) << il_shift) + (q & ((1 << il_shift) - 1));
                        }
                        f3 d;
                        if (mine < n && io.load(mine, o, d, time)) {
                            payi[PAY_RAY_IDX * TRACE_BLOCK] = mine; payi[PAY_PENDING * TRACE_BLOCK] = 0;
                            pay[PAY_DY * TRACE_BLOCK] = d.y; pay[PAY_DZ * TRACE_BLOCK] = d.z; pay[PAY_DLEN * TRACE_BLOCK] = norm3(d);
                            dirx = d.x;
                            if (COUNT) ray_t0 = wall_clock64();
                            best_t = FLT_MAX; best_prim = -1; sp = sbase = 0; xf.id = -2; prim_i = prim_end = 0;
                            plane = lane; bound = FLT_MAX;
                            // Q2: the hit parameter is (ip.x - o.x)/d.x; with d.x == 0 (or NaN) no hit is ever accepted.
                            bool alive = !no_scene && (d.x != 0.0f);
                            // A NaN anywhere in the ray makes every box test of the reference pass (its comparisons are all
                            // false) and every primitive test fail (t >= -eps is false for NaN): the reference walks the WHOLE
                            // tree and finds nothing.  Same answer at once.
                            alive = alive && (o.x == o.x) && (o.y == o.y) && (o.z == o.z) && (d.x == d.x) && (d.y == d.y) && (d.z == d.z);
                            if (COUNT) ++dbg.boxes;
                            // The reference tests the root box first.  Below a wide root that test is implied: a leaf is visited only
                            // on a decided hit of its own box (which implies every ancestor's exact test, the root's included) or
                            // after the exact chain walk, which ends at the root.  Only a root that is itself a leaf needs it here.
                            if (alive && sc.p.root < 0) alive = slab_exact(ld3(sc.p.root_min), ld3(sc.p.root_max), o, d);
                            if (alive) {
                                r = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                                // a zero / denormal-range direction component (inf and NaN quotients): every fast test undecided
                                amb = (fabsf(d.y) > 1e-30f && fabsf(d.z) > 1e-30f && fabsf(d.x) > 1e-30f) ? 1e-6f : __int_as_float(0x7f800000);
                                const float err = sc.p.prune_err + sc.p.prune_rel * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z));
                                marg = mk3(err * fabsf(r.x) * 1.000001f, err * fabsf(r.y) * 1.000001f, err * fabsf(r.z) * 1.000001f);
                                cur = sc.p.root;
                                has_ray = true;
                                // The object-space ray of the scene's unit-matrix objects (Triangle.cpp:141-161 transforms and renormalises
                                // also then, Q8) is prepared here, once per ray and with many lanes, instead of in the primitive path.
                                if (sc.p.unit_object >= 0) enter_object<MOTION>(sc, objc, sc.p.unit_object, o, d, time, xf);
                            } else {
                                Cand none;
                                none.t = FLT_MAX; none.prim = -1; none.ip = mk3(0, 0, 0); none.beta = none.gamma = 0;
                                io.store(sc, mine, none);
                            }
                        }
                    }
                    chunk_next += nidle < avail ? nidle : avail;
                } else if (nidle >= min_idle && sc.p.donate) {
                    // 3b. nothing left in the queue: split the rays still in flight
                    const bool can_give = has_ray && cur != CUR_DONE && sp > sbase;
                    const unsigned long long givers = __ballot(can_give);
                    const int ngive = __popcll(givers);
                    if (ngive > 0) {
                        // idle lane number i takes entry i / ngive (counted from the bottom) of giver number i % ngive: a wave with
                        // few long rays left fills its idle lanes in one round
                        const int irank = __popcll(idle & lt_mask);
                        int src = lane;
                        if (!has_ray) {
                            unsigned long long m = givers;
                            for (int q = irank % ngive; q > 0; --q) m &= m - 1;
                            src = __ffsll((long long)m) - 1;
                        }
                        const int g_sbase = __shfl(sbase, src), g_sp = __shfl(sp, src);
                        const int slot = g_sbase + irank / ngive;
                        const bool take = !has_ray && slot < g_sp;
                        int entry = 0;
                        if (take) entry = stack_pop(lds_col + (src - lane), ovf, ovf_idx + (src - lane), ovf_stride, slot);
                        if (can_give) {
                            const int grank = __popcll(givers & lt_mask);
                            const int want = (nidle - grank + ngive - 1) / ngive;       // idle lanes assigned to this giver
                            const int given = want < sp - sbase ? want : sp - sbase;
                            sbase += given;
                            payi[PAY_PENDING * TRACE_BLOCK] += given;
                        }
                        const float ox = __shfl(o.x, src), oy = __shfl(o.y, src), oz = __shfl(o.z, src);
                        const float dx = __shfl(dirx, src);
                        const float rx = __shfl(r.x, src), ry = __shfl(r.y, src), rz = __shfl(r.z, src);
                        const float mx = __shfl(marg.x, src), my = __shfl(marg.y, src), mz = __shfl(marg.z, src);
                        const float tm = MOTION ? __shfl(time, src) : 0.0f, am = __shfl(amb, src), bd = __shfl(bound, src);
                        const int xi = __shfl(xf.id, src), xg = __shfl(xf.flags, src);
                        const float xox = __shfl(xf.o.x, src), xoy = __shfl(xf.o.y, src), xoz = __shfl(xf.o.z, src);
                        const float xdx = __shfl(xf.d.x, src), xdy = __shfl(xf.d.y, src), xdz = __shfl(xf.d.z, src);
                        const int en = entry;
                        if (take) {
                            o = mk3(ox, oy, oz); dirx = dx; r = mk3(rx, ry, rz); marg = mk3(mx, my, mz);
                            time = tm; amb = am;
                            payi[PAY_RAY_IDX * TRACE_BLOCK] = payi[PAY_RAY_IDX * TRACE_BLOCK + (src - lane)];
                            pay[PAY_DY * TRACE_BLOCK] = pay[PAY_DY * TRACE_BLOCK + (src - lane)];
                            pay[PAY_DZ * TRACE_BLOCK] = pay[PAY_DZ * TRACE_BLOCK + (src - lane)];
                            pay[PAY_DLEN * TRACE_BLOCK] = pay[PAY_DLEN * TRACE_BLOCK + (src - lane)];
                            payi[PAY_PENDING * TRACE_BLOCK] = 0;
                            best_t = bd; bound = bd; best_prim = -1;      // nothing at or beyond the giver's bound can win
                            sp = sbase = 0; prim_i = prim_end = 0;
                            xf.id = xi; xf.flags = xg; xf.o = mk3(xox, xoy, xoz); xf.d = mk3(xdx, xdy, xdz);      // the giver's object-space ray is this ray's
                            plane = src; cur = en; has_ray = true;
                            if (COUNT) { ray_steps = 0; ++dbg.v[3]; }
                        }
                        any_parts = true;
                    }
                }
            }
        }
        if (!__any(has_ray) && exhausted && chunk_next >= chunk_end) break;      // (a wave between two chunks just runs an empty iteration)

        // pruning bound: own best, and for a donated part whatever its parent has found meanwhile (all lanes take part)
        bound = fminf(bound, best_t);
        if (any_parts) bound = fminf(bound, __shfl(bound, plane));

        // A lane that arrived at a leaf (root, descent or pop) unpacks its primitive run (no memory round trip of its own),
        // keeps it as its PENDING leaf and goes on with the next node from its stack: the primitive path is expensive and
        // runs only when enough lanes of the wave hold a pending leaf, and meanwhile the lane does not idle.  Visiting
        // order does not matter for the result (the reduction over leaves is a minimum over (t, record index)); the lane merely prunes with a slightly stale bound.  A lane
        // that reaches a second leaf while one is pending waits for the primitive path.
        if (has_ray && cur < 0 && !(prim_i < prim_end)) {
            const unsigned u = (unsigned)~cur;
            int start = (int)(u & ((1u << DEV_LEAF_START_BITS) - 1)), count = (int)(u >> DEV_LEAF_START_BITS);
            if (count == 0) { const DevLeaf lf = sc.leaves[start]; start = lf.start; count = lf.count; }
            prim_i = start; prim_end = start + count;
            local_t = FLT_MAX; local_prim = -1;
            if (sp == sbase) cur = CUR_NONE;
            else { --sp; cur = stack_pop(lds_col, ovf, ovf_idx, ovf_stride, sp); }
        }

        // ---------------- per lane and iteration: a node step AND, when the wave runs its primitive path, one primitive test ----------------
        // Both records are requested at the top (the primitive record first), the primitive test runs while the node record is
        // still on its way, then the node step.  A lane in a leaf therefore keeps walking its stack while it works through the
        // leaf's primitives; the order of visits does not matter for the result (same reduction).
        const bool in_leaf = has_ray && prim_i < prim_end;
        const bool at_node = has_ray && cur >= 0 && cur < CUR_WAIT;
        const int n_leaf = __popcll(__ballot(in_leaf)), n_node = __popcll(__ballot(at_node));
        const bool run_tri = n_leaf > 0 && (n_leaf >= sc.p.tri_min_lanes || n_node == 0);      // wave-uniform
        const bool do_tri = in_leaf && run_tri;
        // Primitive record (the same way as the node record below: a lane that tests no primitive asks beyond the range)
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const unsigned poff = do_tri ? (unsigned)prim_i * 64u : 0xffffff00u;
        const v4u pr0 = __builtin_amdgcn_raw_buffer_load_b128(prim_rsrc, (int)poff, 0, 0), pr1 = __builtin_amdgcn_raw_buffer_load_b128(prim_rsrc, (int)poff + 16, 0, 0),
                  pr2 = __builtin_amdgcn_raw_buffer_load_b128(prim_rsrc, (int)poff + 32, 0, 0);      // e2.z, object, kind, shape (DevPrim, rt_device.h)
        const float4 p0 = make_float4(__uint_as_float(pr0.x), __uint_as_float(pr0.y), __uint_as_float(pr0.z), __uint_as_float(pr0.w));
        const float4 p1 = make_float4(__uint_as_float(pr1.x), __uint_as_float(pr1.y), __uint_as_float(pr1.z), __uint_as_float(pr1.w));
        const float p2x = __uint_as_float(pr2.x);
        int p_obj = (int)pr2.y, p_kind = (int)pr2.z;
        // Node record: eight 16-byte pieces through a buffer descriptor.  A lane without a node asks for an offset beyond the
        // descriptor's range: the range check answers 0 without touching memory, so no branch, no merging of "loaded" and "not
        // loaded" registers (the flat form cost 32 register clears per iteration) and 32-bit offsets instead of 64-bit addresses.
        // the cone words (piece 7) only matter to a lane that can prune by distance, i.e. one that has a bound when the loads go out; a bound
        // found by this iteration's primitive test prunes from the next node step on (what loads the vector memory path is lanes x loads)
        const bool can_prune = bound < sc.p.flt_max;
        const unsigned voff = at_node ? (unsigned)cur * 128u : 0xffffff00u, voff7 = (at_node && can_prune) ? voff : 0xffffff00u;
#define NODE_PIECE(k, off) const v4u r##k = __builtin_amdgcn_raw_buffer_load_b128(node_rsrc, (int)(off) + 16 * k, 0, 0); \
                           const float4 q##k = make_float4(__uint_as_float(r##k.x), __uint_as_float(r##k.y), __uint_as_float(r##k.z), __uint_as_float(r##k.w))
        NODE_PIECE(0, voff); NODE_PIECE(1, voff); NODE_PIECE(2, voff); NODE_PIECE(3, voff); NODE_PIECE(4, voff); NODE_PIECE(5, voff); NODE_PIECE(6, voff); NODE_PIECE(7, voff7);
#undef NODE_PIECE
        // keeps the compiler from testing the primitive's kind (and waiting for it) before the node loads above are issued
        asm volatile("" : "+v"(p_kind), "+v"(p_obj) : : "memory");
        if (COUNT) { ++dbg.v[4]; if (at_node) ++dbg.v[5]; if (do_tri) ++dbg.v[7]; if (run_tri) ++dbg.v[6]; }
        if (run_tri) {
            if (do_tri) {
                // ProcessIntersectionForLeafNode (BVHTree.cpp:177-190): fresh result per leaf, array order, accept t>0 && t < local-0.001
                const int obj = p_obj, kind = p_kind;
                // every unit-matrix object has the same object-space ray (and the same matrices for the way back)
                bool same = obj == xf.id;
                if (!same && (xf.flags & DEV_OBJ_UNIT) && obj < OBJ_CACHE) same = (__float_as_int(objc[obj * OBJ_CACHE_DWORDS + 32]) & DEV_OBJ_UNIT) != 0;
                if (!same) enter_object<MOTION>(sc, objc, obj, o, mk3(dirx, pay[PAY_DY * TRACE_BLOCK], pay[PAY_DZ * TRACE_BLOCK]), time, xf);
                Cand c;
                bool ok;
                if (kind == DEV_PRIM_TRIANGLE) {
                    if (COUNT) ++dbg.tris;
                    ok = hit_triangle<MOTION>(sc, objc, p0, p1, p2x, xf, o.x, dirx, time, eps, sc.p.isect_eps_p1, c);
                } else {
                    if (COUNT) ++dbg.sph;
                    ok = hit_sphere<MOTION>(sc, objc, p0, xf, o.x, dirx, time, c);
                }
                if (ok && c.t > 0 && c.t < local_t - 0.001f) {
                    local_t = c.t; local_prim = prim_i;
                    pay[0 * TRACE_BLOCK] = c.ip.x; pay[1 * TRACE_BLOCK] = c.ip.y; pay[2 * TRACE_BLOCK] = c.ip.z;
                    pay[3 * TRACE_BLOCK] = c.beta; pay[4 * TRACE_BLOCK] = c.gamma;
                }
                if (++prim_i == prim_end) {
                    // ProcessIntersectionForInternalNode (BVHTree.cpp:192-206) reduced over all leaves: smaller t wins, ties
                    // go to the leaf that comes later in DFS order (Q4); records are stored in that order and a leaf's records
                    // are contiguous, so any record index of this leaf against the best record's index (another leaf's) decides
                    if (local_prim >= 0 && (local_t < best_t || (local_t == best_t && local_prim > best_prim))) {
                        best_t = local_t; best_prim = local_prim;
#pragma unroll
                        for (int k = 0; k < 5; ++k) pay[(5 + k) * TRACE_BLOCK] = pay[k * TRACE_BLOCK];
                    }
                    prim_i = prim_end = 0;
                }
            }
        }
        bound = fminf(bound, best_t);
        if (at_node) {
            // four child boxes (structure of arrays: q0..q2 = min x,y,z of the four children, q3..q5 = max), q6 = references
            const float mnx[4] = {q0.x, q0.y, q0.z, q0.w}, mny[4] = {q1.x, q1.y, q1.z, q1.w}, mnz[4] = {q2.x, q2.y, q2.z, q2.w};
            const float mxx[4] = {q3.x, q3.y, q3.z, q3.w}, mxy[4] = {q4.x, q4.y, q4.z, q4.w}, mxz[4] = {q5.x, q5.y, q5.z, q5.w};
            int cref[4] = {__float_as_int(q6.x), __float_as_int(q6.y), __float_as_int(q6.z), __float_as_int(q6.w)};
            const unsigned ccone[4] = {(unsigned)__float_as_int(q7.x), (unsigned)__float_as_int(q7.y), (unsigned)__float_as_int(q7.z), (unsigned)__float_as_int(q7.w)};
            float key[4];
            bool need_chain = false;
            int cls[4];
            bool far[4], any_far = false;
            // a child is "far" when its conservative lower bound lies beyond the best hit by more than 3e-6 relative (without a hit
            // the bound is FLT_MAX and the product +inf: never far)
            const float bound_far = bound * 1.000003f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float tn, lb;
                int c = slab_fast(mnx[i], mny[i], mnz[i], mxx[i], mxy[i], mxz[i], o, r, marg, amb, tn, lb);
                if (cref[i] == DEV_REF_EMPTY) c = 0;
                // The child's box begins beyond the best hit (an undecided box never does).  It is skipped only if no triangle
                // below it is so nearly in a plane with the ray that the reference's Cramer solve could put a "hit" anywhere
                // on the ray (rt_cone.h) — decided after this loop, for the lanes that have such a child.
                far[i] = c == 1 && can_prune && lb > bound_far;      // can_prune: the cone words were only requested by lanes that had a bound then
                any_far = any_far || far[i];
                // An undecided LEAF is decided exactly by the chain walk.  An undecided INTERNAL child of an ordinary ray (grazing
                // its box) is simply entered: boxes nest, so that is conservative and cheap.  A ray with a zero direction component
                // has EVERY test undecided and would walk the whole tree that way (measured: 3 s per 4K frame of config 5), so
                // for it the internal child gets the reference's exact test of its own box.
                if (c == 2) { if (cref[i] < 0 || !(amb < 1.0f)) need_chain = true; else c = 1; }
                cls[i] = c;
                key[i] = fminf(tn, sc.p.flt_max);    // tn >= 0, never NaN (maxima with 0); +inf -> FLT_MAX (+inf marks a child that is not taken)
            }
            if (COUNT) { ++dbg.nodes; dbg.boxes += 4; ++ray_steps; }
            if (any_far) {
                const float dy = pay[PAY_DY * TRACE_BLOCK], dz = pay[PAY_DZ * TRACE_BLOCK], dl = pay[PAY_DLEN * TRACE_BLOCK];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool prune = far[i] && (sc.p.cone_off || cone_allows_prune(ccone[i], dirx, dy, dz, dl));
                    if (COUNT && far[i]) { ++dbg.v[10]; if (prune) ++dbg.v[11]; else if ((ccone[i] >> 24) == RT_CONE_NEVER) ++dbg.v[12]; else ++dbg.v[13]; }
                    if (prune) cls[i] = 0;
                }
            }
            if (need_chain && amb < 1.0f) {
                need_chain = false;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (cls[i] == 2) {
                        cls[i] = slab_cross(mnx[i], mny[i], mnz[i], mxx[i], mxy[i], mxz[i], o, r, amb);
                        need_chain = need_chain || cls[i] == 2;
                    }
            }
            if (COUNT) { if (need_chain) ++dbg.v[9]; if (__any(need_chain)) ++dbg.v[8]; }
            if (need_chain) {
                const int4 cc4 = reinterpret_cast<const int4 *>(sc.node_chain)[cur];
                const int cch[4] = {cc4.x, cc4.y, cc4.z, cc4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (cls[i] == 2)
                        cls[i] = chain_visit(sc.chain, cch[i], cref[i] < 0, o, mk3(dirx, pay[PAY_DY * TRACE_BLOCK], pay[PAY_DZ * TRACE_BLOCK]), r, amb) ? 1 : 0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) if (cls[i] == 0) key[i] = __int_as_float(0x7f800000);      // +inf: not taken, sorts last
            // sorting network on (entry distance, reference): nearest first
#define CSWAP(a, b) { const bool sw = key[b] < key[a]; const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b]; \
                      const int ra = sw ? cref[b] : cref[a], rb = sw ? cref[a] : cref[b]; key[a] = ka; key[b] = kb; cref[a] = ra; cref[b] = rb; }
            CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
#undef CSWAP
            const float inf = __int_as_float(0x7f800000);
            if (key[0] < inf) {
                const int npush = (key[1] < inf ? 1 : 0) + (key[2] < inf ? 1 : 0) + (key[3] < inf ? 1 : 0);
                if (sp + 3 <= LDS_STACK) {
                    // branch-free: the taken children go to sp .. sp+npush-1 (farthest lowest), the others to a scratch row
                    const int i1 = npush >= 1 ? sp + npush - 1 : LDS_STACK, i2 = npush >= 2 ? sp + npush - 2 : LDS_STACK, i3 = npush >= 3 ? sp : LDS_STACK;
                    lds_col[i1 * TRACE_BLOCK] = cref[1]; lds_col[i2 * TRACE_BLOCK] = cref[2]; lds_col[i3 * TRACE_BLOCK] = cref[3];
                } else {
                    if (npush >= 3) stack_push(lds_col, ovf, ovf_idx, ovf_stride, sp, cref[3]);
                    if (npush >= 2) stack_push(lds_col, ovf, ovf_idx, ovf_stride, sp + npush - 2, cref[2]);
                    if (npush >= 1) stack_push(lds_col, ovf, ovf_idx, ovf_stride, sp + npush - 1, cref[1]);
                }
                sp += npush;
                cur = cref[0];
            } else if (sp == sbase) cur = CUR_NONE;
            else { --sp; cur = stack_pop(lds_col, ovf, ovf_idx, ovf_stride, sp); }
        }
        if (has_ray && cur == CUR_NONE && !(prim_i < prim_end)) cur = CUR_DONE;
    }
}

struct HitOut {
    float t; int shape, object, prim;
    f3 ip, n; float u, v, theta, phi;
};

// inverse(transpose(M)) of a moving object at ray time (kept out of line; static objects read DevObject::inv directly)
__device__ __noinline__ void motion_inverse(const DevObject &ob, float time, float *inv, int *flags) {
    float M[16];
    object_matrices(ob, time, M, inv, *flags);
}

// What Triangle::Intersect / Sphere::Intersect put into the SurfaceIntersection of the winning primitive.
// ALL_FIELDS = false (render path): uv is only computed for objects that carry a texture, the only reader of it there; a flat,
// untextured triangle then needs none of its three vertex records (108 bytes of gathers per hit).
template <bool ALL_FIELDS>
__device__ __forceinline__ void finalize_hit(const DevScene &sc, f3 o, f3 d, float time, const Cand &best, HitOut &h) {
    h.t = best.t; h.prim = best.prim; h.ip = best.ip; h.theta = h.phi = 0; h.u = h.v = 0; h.n = mk3(0, 0, 0);
    if (best.prim < 0) { h.shape = -1; h.object = -1; h.t = FLT_MAX; return; }
    const DevPrim &P = sc.prims[best.prim];
    const DevObject &ob = sc.objects[P.object];
    h.shape = P.shape; h.object = P.object;
    float inv_m[16];
    const float *inv = ob.inv;
    int flags = ob.flags | ((ob.flags & DEV_OBJ_HAS_XFORM) ? 0 : DEV_OBJ_IDENTITY);
    if ((ob.flags & DEV_OBJ_HAS_XFORM) && (ob.flags & DEV_OBJ_MOTION)) { motion_inverse(ob, time, inv_m, &flags); inv = inv_m; }
    if (P.kind == DEV_PRIM_TRIANGLE) {
        const float eps = sc.p.isect_eps;
        const float e = 1 + eps;
        const float w0 = e - best.beta - best.gamma;                           // Triangle.cpp:252-266
        f3 n = ld3(P.nrm);
        const bool smooth = (ob.flags & DEV_OBJ_SMOOTH) != 0;
        if (ALL_FIELDS || smooth || ob.tex_color >= 0 || ob.tex_normal >= 0) {
            const int *tv = sc.tri_vert + 3 * (size_t)P.shape;
            const DevVertex &a = sc.verts[tv[0]], &b = sc.verts[tv[1]], &c = sc.verts[tv[2]];
            h.u = a.uv[0] * w0 + b.uv[0] * best.beta + c.uv[0] * best.gamma;
            h.v = a.uv[1] * w0 + b.uv[1] * best.beta + c.uv[1] * best.gamma;
            if (smooth) n = normalize3(ld3(a.n) * w0 + ld3(b.n) * best.beta + ld3(c.n) * best.gamma);
        }
        if (!(flags & DEV_OBJ_IDENTITY)) n = normalize3(mat_mul_xyz(inv, n.x, n.y, n.z, 0.0f));   // Triangle.cpp:299-300
        h.n = n;
    } else {
        // Sphere.cpp:31-54, 98-131 re-evaluated for the winner (deterministic: same values as during traversal)
        const f3 center = ld3(P.p0);
        const float radius = P.e1[0];
        f3 so = matT_mul_xyz(inv, o.x, o.y, o.z, 1.0f), sd = normalize3(matT_mul_xyz(inv, d.x, d.y, d.z, 0.0f));
        float t = 0;
        sphere_t(center, radius, so, sd, t);
        f3 p = so + sd * t;
        f3 n = (p - center) / radius;
        float theta = acosf((p.y - center.y) / radius);
        float phi = atan2f((p.z - center.z), (p.x - center.x));
        const double PI = 3.14159265;
        h.u = (float)((-phi + PI) / (2 * PI));
        h.v = (float)(theta / PI);
        h.theta = (float)(h.v * PI);
        h.phi = (float)(PI - (2 * PI) * h.u);
        h.n = normalize3(mat_mul_xyz(inv, n.x, n.y, n.z, 0.0f));
    }
}

template <bool COUNT>
__device__ __forceinline__ void flush_counters(DevCounters *counters, unsigned n, unsigned b, unsigned t, unsigned s) {
    if (!COUNT || !counters) return;
    // wave64 reduction, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_down(n, off); b += __shfl_down(b, off); t += __shfl_down(t, off); s += __shfl_down(s, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counters->node_visits, (unsigned long long)n); atomicAdd(&counters->box_tests, (unsigned long long)b);
        atomicAdd(&counters->tri_tests, (unsigned long long)t); atomicAdd(&counters->sphere_tests, (unsigned long long)s);
    }
}

// Ray source / result sink of the render launches.  One launch serves the closest-hit queue of a level (indices
// [0, n_closest)) AND the compact shadow queue of the level before it (the indices after that); either part may be empty.
//   closest: raw record hit0 = (t, prim record | -1, beta, gamma), hit1 = (ip.xyz, unused), completed by k_shade_pre
//            (finalize_hit needs acosf/atan2f/inverse code that would otherwise set the register allocation here)
//   shadow (Q7): a full closest-hit query, then the distance window test of
//            LightContributionCalculator::IsThereAnObjectBetweenLightAndIntersectionPoint (:130-145)
struct RenderIO {
    enum { DBG_SET = 0 };
    LevelBuf lv;
    int n_closest;                 // closest-hit rays of lv; when closest_dev is set: their upper bound (the queue's capacity)
    const int *closest_dev;        // device counter holding the real number (a level whose length the host never reads), or null
    int shadow_cap;                // capacity of the shadow queue (its length is the device counter sb.count)
    ShadowBuf sb;
    // called once per thread at kernel start: takes the counts from the device; returns the total queue length
    __device__ __forceinline__ int resolve(int, const int *) {
        if (closest_dev) { const int c = *closest_dev; n_closest = c <= n_closest ? c : 0; }      // an outgrown level is not traced (its pass is discarded)
        int ns = 0;
        if (shadow_cap > 0) { ns = *sb.count; ns = ns < shadow_cap ? ns : shadow_cap; }
        return n_closest + ns;
    }
    __device__ __forceinline__ int kind(int i) const { return i < n_closest ? 0 : 1; }
    __device__ __forceinline__ bool load(int i, f3 &o, f3 &d, float &time) const {
        const float4 *po = lv.ray_o, *pd = lv.ray_d;
        if (i >= n_closest) { i -= n_closest; po = sb.o; pd = sb.d; }
        const float4 ro = po[i], rd = pd[i];
        o = mk3(ro.x, ro.y, ro.z); d = mk3(rd.x, rd.y, rd.z); time = ro.w;
        return true;
    }
    __device__ __forceinline__ void store(const DevScene &sc, int i, const Cand &best) const {
        if (i < n_closest) {
            lv.hit0[i] = make_float4(best.t, __int_as_float(best.prim), best.beta, best.gamma);
            lv.hit1[i] = make_float4(best.ip.x, best.ip.y, best.ip.z, 0.0f);
            return;
        }
        i -= n_closest;
        const float4 ref = sb.ipref[i];             // shaded point, result slot
        int occ = 0;
        if (best.prim >= 0) {
            const float dist = norm3(best.ip - mk3(ref.x, ref.y, ref.z));
            const float se = sc.p.shadow_eps;
            occ = (dist > se && dist < sb.d[i].w - se) ? 1 : 0;
        }
        sb.occluded[__float_as_int(ref.w)] = occ;
    }
};
// rt_intersect_batch: AccelerationStructure::Intersect for n independent rays (traversal, then k_finalize_batch)
struct BatchIO {
    enum { DBG_SET = 0 };
    __device__ __forceinline__ int resolve(int n, const int *n_dev) { return n_dev ? n + *n_dev : n; }
    __device__ __forceinline__ int kind(int) const { return 0; }
    const rt_ray *rays;
    rt_hit *hits;
    __device__ __forceinline__ bool load(int i, f3 &o, f3 &d, float &time) const {
        const rt_ray r = rays[i];
        o = ld3(r.origin); d = ld3(r.direction); time = r.time;
        return true;
    }
    __device__ __forceinline__ void store(const DevScene &, int i, const Cand &best) const {
        rt_hit out;                      // raw form: shape = prim record, uv = (beta, gamma); completed by k_finalize_batch
        out.t = best.t; out.shape = best.prim;
        out.ip[0] = best.ip.x; out.ip[1] = best.ip.y; out.ip[2] = best.ip.z;
        out.n[0] = out.n[1] = out.n[2] = 0.0f;
        out.uv[0] = best.beta; out.uv[1] = best.gamma;
        hits[i] = out;
    }
};

template <bool COUNT, bool MOTION, class IO>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void k_trace_generic(DevScene sc, IO io, int n, const int *n_dev, int *fetch_counter, int *ovf, int ovf_stride, DevCounters *counters) {
    __shared__ int s_stack[(LDS_STACK + 1) * TRACE_BLOCK];      // + one scratch row for the branch-free push
    __shared__ float s_pay[LDS_PAYLOAD * TRACE_BLOCK];
    __shared__ float s_obj[OBJ_CACHE * OBJ_CACHE_DWORDS];
    {
        const int nobj = sc.p.n_objects < OBJ_CACHE ? sc.p.n_objects : OBJ_CACHE;
        for (int i = threadIdx.x; i < nobj * OBJ_CACHE_DWORDS; i += TRACE_BLOCK) {
            const int ob = i / OBJ_CACHE_DWORDS, k = i - ob * OBJ_CACHE_DWORDS;
            const DevObject &src = sc.objects[ob];
            s_obj[i] = k < 16 ? src.M[k] : (k < 32 ? src.inv[k - 16] : __int_as_float(src.flags));
        }
        __syncthreads();
    }
    n = io.resolve(n, n_dev);                     // queue lengths that were produced on the device
    TL_SET(0);
    TraceDbg dbg;
    dbg.nodes = dbg.boxes = dbg.tris = dbg.sph = 0;
    for (int k = 0; k < 14; ++k) dbg.v[k] = 0;
    trace_persistent<COUNT, MOTION, IO>(sc, io, n, fetch_counter, (lds_int *)s_stack + threadIdx.x, (lds_float *)s_pay + threadIdx.x, (const lds_float *)s_obj, ovf, ovf_stride, dbg,
                                        (COUNT && counters && counters->want_hist) ? &counters->hist[0] : nullptr);
    TL_SET(3);
    flush_counters<COUNT>(counters, dbg.nodes, dbg.boxes, dbg.tris, dbg.sph);
    if (COUNT && counters) {
        for (int k = 0; k < 14; ++k) {
            unsigned v = dbg.v[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if ((threadIdx.x & 63) == 0) atomicAdd(&counters->dbg[IO::DBG_SET][k], (unsigned long long)v);
        }
    }
}
__global__ __launch_bounds__(256) void k_finalize_batch(DevScene sc, const rt_ray *rays, rt_hit *hits, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rt_ray r = rays[i];
    rt_hit raw = hits[i];
    Cand best;
    best.t = raw.t; best.prim = raw.shape; best.ip = ld3(raw.ip); best.beta = raw.uv[0]; best.gamma = raw.uv[1];
    HitOut h;
    finalize_hit<true>(sc, ld3(r.origin), ld3(r.direction), r.time, best, h);
    rt_hit out;
    out.t = h.t; out.shape = h.shape;
    out.ip[0] = h.ip.x; out.ip[1] = h.ip.y; out.ip[2] = h.ip.z;
    out.n[0] = h.n.x; out.n[1] = h.n.y; out.n[2] = h.n.z;
    out.uv[0] = h.u; out.uv[1] = h.v;
    hits[i] = out;
}

__global__ void k_fill_int(int *p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// a fill kernel instead of hipMemsetAsync: safe to record into a HIP graph
void rt_launch_fill_int(hipStream_t s, int *p, int n, int v) {
    if (n > 0) hipLaunchKernelGGL(k_fill_int, dim3((n + 255) / 256), dim3(256), 0, s, p, n, v);
}
int rt_trace_block_threads() { return TRACE_BLOCK; }
int rt_trace_resident_blocks() { return g_num_cus * TRACE_RESIDENT_PER_CU; }

static int g_trace_block_cap = 0;
void rt_set_trace_block_cap(int cap) { g_trace_block_cap = cap; }
// Persistent grid: one block per 256 rays up to the 1024 blocks that are resident at a time (4 per CU); blocks beyond those
// only start when the queue is already empty.  (Fewer, longer-running blocks for small launches: +27 % time at 200x120.)
static int g_small_rays_per_block = TRACE_BLOCK;
void rt_set_small_rays_per_block(int v) { g_small_rays_per_block = v < 16 ? 16 : (v > TRACE_BLOCK ? TRACE_BLOCK : v); }
static int trace_blocks(int n) {
    int want = (n + g_small_rays_per_block - 1) / g_small_rays_per_block;
    int cap = g_num_cus * TRACE_RESIDENT_PER_CU;
    if (g_trace_block_cap > 0 && g_trace_block_cap < cap) cap = g_trace_block_cap;
    return want < cap ? (want < 1 ? 1 : want) : cap;
}
template <class IO>
static void trace_dispatch(hipStream_t s, const DevScene &sc, IO io, int n, const int *n_dev, int *fetch_counter, int *ovf, DevCounters *counters,
                           int max_blocks = 0, int ovf_stride = 0) {
    int blocks = trace_blocks(n);
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    const dim3 g(blocks), b(TRACE_BLOCK);
    const int stride = ovf_stride > 0 ? ovf_stride : rt_trace_grid_threads();
    const bool cnt = counters != nullptr, mot = sc.p.has_motion != 0;
    if (cnt && mot) hipLaunchKernelGGL((k_trace_generic<true, true, IO>), g, b, 0, s, sc, io, n, n_dev, fetch_counter, ovf, stride, counters);
    else if (cnt) hipLaunchKernelGGL((k_trace_generic<true, false, IO>), g, b, 0, s, sc, io, n, n_dev, fetch_counter, ovf, stride, counters);
    else if (mot) hipLaunchKernelGGL((k_trace_generic<false, true, IO>), g, b, 0, s, sc, io, n, n_dev, fetch_counter, ovf, stride, counters);
    else hipLaunchKernelGGL((k_trace_generic<false, false, IO>), g, b, 0, s, sc, io, n, n_dev, fetch_counter, ovf, stride, counters);
}
// `fetch_counter` must point to a zeroed int that no other launch in flight uses.  n_closest rays of lv (may be 0) and the
// shadow queue sb (its length is read on the device; `shadow_bound` >= that length sizes the launch; 0 = no shadow part).
#ifdef RT_TIMELINE_BUILD
static int g_tl_next = -1;            // -1: timeline off
void rt_tl_begin() {
    void *p = nullptr;
    hipGetSymbolAddress(&p, HIP_SYMBOL(g_tl));
    hipMemset(p, 0, sizeof(unsigned long long) * TL_SLOTS * TL_WAVES * 4);
    g_tl_next = 0;
}
void rt_tl_end() {
    static std::vector<unsigned long long> h((size_t)TL_SLOTS * TL_WAVES * 4);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_tl), h.size() * sizeof(unsigned long long));
    const int n = g_tl_next;
    g_tl_next = -1;
    unsigned long long prev_end = 0;
    for (int i = 0; i < n && i < TL_SLOTS; ++i) {
        std::vector<double> st, ex, en;
        unsigned long long t0 = ~0ull, t3 = 0;
        for (int w = 0; w < TL_WAVES; ++w) { const unsigned long long *r = &h[((size_t)i * TL_WAVES + w) * 4]; if (r[0]) { t0 = std::min(t0, r[0]); t3 = std::max(t3, r[3]); } }
        if (t0 == ~0ull) continue;
        for (int w = 0; w < TL_WAVES; ++w) {
            const unsigned long long *r = &h[((size_t)i * TL_WAVES + w) * 4];
            if (!r[0]) continue;
            st.push_back((r[0] - t0) * 0.01); if (r[1]) ex.push_back((r[1] - t0) * 0.01); en.push_back((r[3] - t0) * 0.01);
        }
        auto pct = [](std::vector<double> &v, double q) { if (v.empty()) return -1.0; std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
        fprintf(stderr, "[tl] launch %d (%zu waves): gap %.1f | start p50 %.1f max %.1f | queue empty first %.1f p50 %.1f | wave exit p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us\n",
                i, st.size(), prev_end ? ((double)t0 - (double)prev_end) * 0.01 : 0.0, pct(st, 0.5), pct(st, 1.0), pct(ex, 0.0), pct(ex, 0.5),
                pct(en, 0.1), pct(en, 0.5), pct(en, 0.9), pct(en, 0.99), pct(en, 1.0));
        prev_end = t3;
    }
}
#else
void rt_tl_begin() {}
void rt_tl_end() {}
#endif
void rt_launch_trace(hipStream_t s, const DevScene &sc_in, LevelBuf lv, int n_closest, const int *closest_dev, ShadowBuf sb, int shadow_bound, int *fetch_counter, int *ovf,
                     DevCounters *counters) {
    if (n_closest <= 0 && shadow_bound <= 0) return;
    DevScene sc = sc_in;
    sc.p.tl_slot = -1;
#ifdef RT_TIMELINE_BUILD
    if (g_tl_next >= 0 && g_tl_next < TL_SLOTS) sc.p.tl_slot = g_tl_next++;
#endif
    RenderIO io; io.lv = lv; io.n_closest = n_closest > 0 ? n_closest : 0; io.closest_dev = n_closest > 0 ? closest_dev : nullptr; io.sb = sb;
    io.shadow_cap = shadow_bound > 0 ? shadow_bound : 0;
    const dim3 g(trace_blocks(io.n_closest + io.shadow_cap)), b(TRACE_BLOCK);
    const int stride = rt_trace_grid_threads();
    const int *n_dev = nullptr;
    const bool cnt = counters != nullptr, mot = sc.p.has_motion != 0;
    if (cnt && mot) hipLaunchKernelGGL((k_trace_generic<true, true, RenderIO>), g, b, 0, s, sc, io, io.n_closest, n_dev, fetch_counter, ovf, stride, counters);
    else if (cnt) hipLaunchKernelGGL((k_trace_generic<true, false, RenderIO>), g, b, 0, s, sc, io, io.n_closest, n_dev, fetch_counter, ovf, stride, counters);
    else if (mot) hipLaunchKernelGGL((k_trace_generic<false, true, RenderIO>), g, b, 0, s, sc, io, io.n_closest, n_dev, fetch_counter, ovf, stride, counters);
    else hipLaunchKernelGGL((k_trace_generic<false, false, RenderIO>), g, b, 0, s, sc, io, io.n_closest, n_dev, fetch_counter, ovf, stride, counters);
}
// max_blocks / ovf_stride (0 = defaults): a caller that brings a smaller stack-overflow area caps the persistent grid at
// max_blocks blocks and lays the overflow columns out with ovf_stride = max_blocks * TRACE_BLOCK threads.
void rt_launch_trace_batch(hipStream_t s, const DevScene &sc, const rt_ray *rays, rt_hit *hits, int n, int *fetch_counter, int *ovf, DevCounters *counters,
                           int max_blocks, int ovf_stride) {
    if (n <= 0) return;
    BatchIO io; io.rays = rays; io.hits = hits;
    trace_dispatch(s, sc, io, n, nullptr, fetch_counter, ovf, counters, max_blocks, ovf_stride);
    hipLaunchKernelGGL(k_finalize_batch, dim3((n + 255) / 256), dim3(256), 0, s, sc, rays, hits, n);
}

// ====================================================================================================
// Ray generation
// ====================================================================================================
__device__ __forceinline__ f3 camera_direction(const DevCamera &c, int row, int col, float ho, float vo) {   // Camera.cpp:103-125
    float hm = c.left + (col + ho) * c.pixel_w;
    float vm = c.top - (row + vo) * c.pixel_h;
    return normalize3(ld3(c.gaze) * c.distance + ld3(c.right) * hm + ld3(c.up) * vm);
}

__global__ __launch_bounds__(256) void k_raygen(DevCamera cam, const int *pixel_list, int first_pixel, int n, LevelBuf lv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int pixel = pixel_list[first_pixel + i / cam.spp];
    const int s = i % cam.spp;
    const int row = pixel / cam.nx, col = pixel % cam.nx;
    f3 o = ld3(cam.pos), d;
    float time = 0.0f;                                                 // Q21
    if (cam.spp > 1) {
        double u[4];
        rng_slot(cam.seed, (unsigned)pixel, (unsigned)s, 1u, SLOT_PIXEL, u);
        float sx = (s % cam.jitter_n) * cam.jitter_w;                 // FindIthPartition, acmath.h:901-907 (Q19)
        float sy = (s / cam.jitter_n) * cam.jitter_w;
        sx += (float)uni(u[0], 0.0f, cam.jitter_w);                   // PixelSampler.cpp:31-35
        sy += (float)uni(u[1], 0.0f, cam.jitter_w);
        d = camera_direction(cam, row, col, sx, sy);
        if (cam.aperture > 0) {                                        // ApplyLensTilt, Camera.cpp:132-147 (Q25)
            float ax = (float)uni(u[2], -cam.aperture / 2.0, cam.aperture / 2.0);
            float ay = (float)uni(u[3], -cam.aperture / 2.0, cam.aperture / 2.0);
            f3 moved = ld3(cam.pos) + ld3(cam.right) * ax + ld3(cam.up) * ay;
            float td = cam.focus / dot3(d, ld3(cam.gaze));
            f3 p = o + d * td;
            d = normalize3(p - moved);
            o = moved;
        }
        double ut[4];
        rng_slot(cam.seed, (unsigned)pixel, (unsigned)s, 1u, SLOT_TIME, ut);
        time = (float)uni(ut[0], 0.0, 1.0);                           // MultiSampledRayGenerator.cpp:30
    } else {
        d = camera_direction(cam, row, col, 0.5f, 0.5f);              // Camera.cpp:76-79
    }
    lv.ray_o[i] = make_float4(o.x, o.y, o.z, time);
    lv.ray_d[i] = make_float4(d.x, d.y, d.z, 1.0f);                   // Material::DefaultMaterial: n = 1
    lv.ray_m[i] = make_int4(pixel, s, 1, -1);
}
void rt_launch_raygen(hipStream_t s, const DevCamera &cam, const int *pixel_list, int first_pixel, int n_samples, LevelBuf lv) {
    if (n_samples <= 0) return;
    hipLaunchKernelGGL(k_raygen, dim3((n_samples + 255) / 256), dim3(256), 0, s, cam, pixel_list, first_pixel, n_samples, lv);
}

// ====================================================================================================
// Textures (Q22, Q23)
// ====================================================================================================
__device__ __forceinline__ float lerpf(float a, float b, float t) { return (1 - t) * a + t * b; }             // acmath.h:696
__device__ __forceinline__ float fade(float t) { return 6 * t * t * t * t * t - 15 * t * t * t * t + 10 * t * t * t; } // acmath.h:703

__constant__ float c_grad[16][3] = {{1, 1, 0}, {-1, 1, 0}, {1, -1, 0}, {-1, -1, 0}, {1, 0, 1}, {-1, 0, 1}, {1, 0, -1}, {-1, 0, -1},
                                    {0, 1, 1}, {0, -1, 1}, {0, 1, -1}, {0, -1, -1}, {1, 1, 0}, {-1, 1, 0}, {0, -1, 1}, {0, -1, -1}};

__device__ __forceinline__ int perm_at(const rt_texture &tx, int i) { int r = i % 16; if (r < 0) r += 16; return tx.perm[r]; }
__device__ __forceinline__ float grad_dot(const rt_texture &tx, int i, int j, int k, float x, float y, float z) {
    const int g = perm_at(tx, i + perm_at(tx, j + perm_at(tx, k)));
    return c_grad[g][0] * x + c_grad[g][1] * y + c_grad[g][2] * z;
}
__device__ float perlin_value(const rt_texture &tx, float u, float v, float w) {                               // PerlinTextureImpl.cpp:74-140
    const float px = u * tx.noise_scale, py = v * tx.noise_scale, pz = w * tx.noise_scale;
    const int fx = (int)floorf(px), fy = (int)floorf(py), fz = (int)floorf(pz);
    const int x0 = fx % 16, y0 = fy % 16, z0 = fz % 16;
    const int x1 = (x0 + 1) % 16, y1 = (y0 + 1) % 16, z1 = (z0 + 1) % 16;
    const float dxf = px - fx, dyf = py - fy, dzf = pz - fz;
    const float dxc = dxf - 1, dyc = dyf - 1, dzc = dzf - 1;
    const float w0 = fade(dxf), w1 = fade(dyf), w2 = fade(dzf);
    const float l0 = lerpf(grad_dot(tx, x0, y0, z0, dxf, dyf, dzf), grad_dot(tx, x1, y0, z0, dxc, dyf, dzf), w0);
    const float l1 = lerpf(grad_dot(tx, x0, y1, z0, dxf, dyc, dzf), grad_dot(tx, x1, y1, z0, dxc, dyc, dzf), w0);
    const float l2 = lerpf(grad_dot(tx, x0, y0, z1, dxf, dyf, dzc), grad_dot(tx, x1, y0, z1, dxc, dyf, dzc), w0);
    const float l3 = lerpf(grad_dot(tx, x0, y1, z1, dxf, dyc, dzc), grad_dot(tx, x1, y1, z1, dxc, dyc, dzc), w0);
    float r = lerpf(lerpf(l0, l1, w1), lerpf(l2, l3, w1), w2);
    if (tx.noise_conversion == RT_NOISE_ABSVAL) r = fabsf(r); else r = (r + 1) / 2.0f;
    return r;
}
__device__ __forceinline__ int wrap_once(int c, int hi) { return c > hi ? 0 : (c < 0 ? hi : c); }              // TextureReader.cpp:89-95
__device__ f3 fetch_texel(const DevScene &sc, const rt_texture &tx, int row, int col) {
    row = wrap_once(row, tx.height - 1); col = wrap_once(col, tx.width - 1);
    const long long idx = 3ll * ((long long)row * tx.width + col);
    if (tx.texel_is_float) { const float *f = reinterpret_cast<const float *>(sc.texels + tx.texel_offset); return mk3(f[idx], f[idx + 1], f[idx + 2]); }
    const uint8_t *b = sc.texels + tx.texel_offset;
    return mk3((float)b[idx], (float)b[idx + 1], (float)b[idx + 2]);
}
__device__ f3 image_rgb(const DevScene &sc, const rt_texture &tx, float u, float v) {                          // TextureValueRetrieveMethod.cpp:20-55
    if (u > 1.0f) u = u - (int)u;
    if (v > 1.0f) v = v - (int)v;
    const float rc = u * (tx.width - 1), rr = v * (tx.height - 1);
    if (!tx.bilinear) return fetch_texel(sc, tx, (int)round((double)rr), (int)round((double)rc));
    const int row = (int)floor((double)rr), col = (int)floor((double)rc);
    const float ro = rr - row, co = rc - col;
    return fetch_texel(sc, tx, row, col) * (1 - ro) * (1 - co) + fetch_texel(sc, tx, row + 1, col) * (ro) * (1 - co) +
           fetch_texel(sc, tx, row, col + 1) * (1 - ro) * (co) + fetch_texel(sc, tx, row + 1, col + 1) * (ro) * (co);
}
__device__ f3 texture_rgb(const DevScene &sc, const rt_texture &tx, float u, float v, float w) {               // Texture.cpp:23-29
    if (tx.kind == RT_TEX_PERLIN) { float r = perlin_value(tx, u, v, w); return mk3(r, r, r); }
    return image_rgb(sc, tx, u, v);
}

struct Surf {          // the SurfaceIntersection fields shading needs
    f3 ip, n; float u, v, theta, phi;
    int shape, object, prim;
};

__device__ f3 diffuse_coefficient(const DevScene &sc, const rt_material &m, int tex_color, const Surf &s) {    // Intersection.cpp:31-42, ColorChangerTexture.cpp
    const f3 kd = ld3(m.kd);
    if (tex_color < 0) return kd;
    const rt_texture &tx = sc.textures[tex_color];
    f3 base;
    if (tx.decal == RT_DECAL_REPLACE_KD || tx.decal == RT_DECAL_BLEND_KD) {
        if (tx.kind == RT_TEX_PERLIN) base = texture_rgb(sc, tx, s.ip.x, s.ip.y, s.ip.z);
        else base = image_rgb(sc, tx, s.u, s.v) / (float)tx.normalizer;
        if (tx.decal == RT_DECAL_REPLACE_KD) return base;
        f3 r = (kd + base) / 2;
        return mk3(clampf(r.x, 0.0f, 1.0f), clampf(r.y, 0.0f, 1.0f), clampf(r.z, 0.0f, 1.0f));
    }
    return texture_rgb(sc, tx, s.u, s.v, 0.0f);
}

__device__ f3 perlin_bump(const rt_texture &tx, const Surf &s) {                                               // PerlinTextureImpl.cpp:45-67
    const float e = 0.001f;
    const float c0 = perlin_value(tx, s.ip.x, s.ip.y, s.ip.z);
    const float cx = perlin_value(tx, s.ip.x + e, s.ip.y, s.ip.z);
    const float cy = perlin_value(tx, s.ip.x, s.ip.y + e, s.ip.z);
    const float cz = perlin_value(tx, s.ip.x, s.ip.y, s.ip.z + e);
    const f3 diff = mk3((cx - c0) / e, (cy - c0) / e, (cz - c0) / e);
    const f3 gp = dot3(diff, s.n) * s.n;
    f3 gn = diff - gp;
    gn = s.n - gn * tx.bump_factor;
    return normalize3(gn);
}
__device__ f3 image_normal_value(const DevScene &sc, const rt_texture &tx, float u, float v) {                 // ImageTextureImpl.cpp:87-94
    const f3 n = image_rgb(sc, tx, u, v) / 255.0f;
    return normalize3(mk3((n.x - 0.5f) * 2.0f, (n.y - 0.5f) * 2.0f, (n.z - 0.5f) * 2.0f));
}
__device__ void image_color_values(const DevScene &sc, const rt_texture &tx, float uu, float vv, float off, float &c0, float &ch, float &cv) { // :157-176
    const float u = uu - floorf(uu), v = vv - floorf(vv);
    const float mul = tx.bump_factor / 255;
    const f3 a = image_rgb(sc, tx, u, v) * mul, h = image_rgb(sc, tx, u + off, v) * mul, w = image_rgb(sc, tx, u, v + off) * mul;
    c0 = (a.x + a.y + a.z) / 3; ch = (h.x + h.y + h.z) / 3; cv = (w.x + w.y + w.z) / 3;
}
// tangent / bitangent of a triangle from inverse(uv edge matrix) * position edge matrix (ImageTextureImpl.cpp:40-49, 131-141)
__device__ void triangle_tangents(const DevScene &sc, const DevPrim &P, bool negate_uv, f3 ea, f3 eb, f3 &tangent, f3 &bitangent) {
    const int *tv = sc.tri_vert + 3 * (size_t)P.shape;
    const DevVertex &a = sc.verts[tv[0]], &b = sc.verts[tv[1]], &c = sc.verts[tv[2]];
    float e1x = -(a.uv[0] - b.uv[0]), e1y = -(a.uv[1] - b.uv[1]), e2x = -(a.uv[0] - c.uv[0]), e2y = -(a.uv[1] - c.uv[1]);
    float m00 = e1x, m01 = e2x, m10 = e1y, m11 = e2y;                  // m[col][row]
    if (negate_uv) { m00 = -m00; m01 = -m01; m10 = -m10; m11 = -m11; }
    const float ood = 1.0f / (+m00 * m11 - m10 * m01);                // glm::inverse(mat2)
    const float i00 = +m11 * ood, i01 = -m01 * ood, i10 = -m10 * ood, i11 = +m00 * ood;
    const f3 rx = mk3(i00 * ea.x + i10 * eb.x, i00 * ea.y + i10 * eb.y, i00 * ea.z + i10 * eb.z);
    const f3 ry = mk3(i01 * ea.x + i11 * eb.x, i01 * ea.y + i11 * eb.y, i01 * ea.z + i11 * eb.z);
    tangent = normalize3(rx); bitangent = normalize3(ry);
}
__device__ void sphere_tbn(const DevPrim &P, const Surf &s, f3 &tangent, f3 &bitangent, f3 &normal) {          // ImageTextureImpl.cpp:203-228
    const double PI = 3.14159265;
    const float radius = P.e1[0];
    const float x = (float)(radius * sin(s.v * PI) * cos(PI - 2 * s.u * PI));
    const float y = (float)(radius * cos(s.v * PI));
    const float z = (float)(radius * sin(s.v * PI) * sin(PI - 2 * s.u * PI));
    const f3 T = mk3((float)(2 * z * PI), .0f, (float)(-2 * x * PI));
    const f3 B = mk3((float)(y * cosf(s.phi) * PI), (float)(-radius * sinf(s.theta) * PI), (float)(y * sinf(s.phi) * PI));
    tangent = normalize3(T); bitangent = normalize3(B); normal = s.n;
    tangent = tangent - normal * dot3(tangent, normal);
    bitangent = bitangent - dot3(bitangent, normal) * normal - dot3(tangent, bitangent) * tangent;
}
// SurfaceIntersection::TweakSurfaceNormal (Intersection.cpp:25-29, NormalChangerTexture.cpp)
__device__ __noinline__ f3 tweak_normal(const DevScene &sc, const Surf &s, int tex_normal) {
    const rt_texture &tx = sc.textures[tex_normal];
    const DevPrim &P = sc.prims[s.prim];
    const bool tri = P.kind == DEV_PRIM_TRIANGLE;
    if (tx.decal == RT_DECAL_REPLACE_NORMAL) {
        if (tx.kind == RT_TEX_PERLIN) return mk3(0, 0, 0);             // PerlinTextureImpl.cpp:16-24
        const f3 tn = image_normal_value(sc, tx, s.u, s.v);
        if (tri) {                                                     // ImageTextureImpl.cpp:36-55
            f3 t, b;
            triangle_tangents(sc, P, true, ld3(P.e1), ld3(P.e2), t, b);
            const f3 n = ld3(P.nrm);
            return mk3(t.x * tn.x + b.x * tn.y + n.x * tn.z, t.y * tn.x + b.y * tn.y + n.y * tn.z, t.z * tn.x + b.z * tn.y + n.z * tn.z);
        }
        f3 t, b, n;
        sphere_tbn(P, s, t, b, n);
        return normalize3(mk3(t.x * tn.x + b.x * tn.y + n.x * tn.z, t.y * tn.x + b.y * tn.y + n.y * tn.z, t.z * tn.x + b.z * tn.y + n.z * tn.z));
    }
    if (tx.decal != RT_DECAL_BUMP_NORMAL) return s.n;
    if (tx.kind == RT_TEX_PERLIN) return perlin_bump(tx, s);
    if (tri) {                                                         // ImageTextureImpl.cpp:96-155
        const DevObject &ob = sc.objects[P.object];
        const f3 e1 = -ld3(P.e1), e2 = -ld3(P.e2);
        const f3 ea = mat_mul_xyz(ob.M, e1.x, e1.y, e1.z, 0.0f), eb = mat_mul_xyz(ob.M, e2.x, e2.y, e2.z, 0.0f);
        f3 t, b;
        triangle_tangents(sc, P, false, ea, eb, t, b);
        const f3 n = cross3(t, b);
        t = t - n * dot3(t, n);
        b = b - dot3(b, n) * n - dot3(t, b) * t;
        float c0, ch, cv;
        image_color_values(sc, tx, s.u, s.v, 0.001f, c0, ch, cv);
        return cross3(t + (ch - c0) * n, b + (cv - c0) * n);
    }
    f3 t, b, n;                                                        // ImageTextureImpl.cpp:190-243
    sphere_tbn(P, s, t, b, n);
    float c0, ch, cv;
    image_color_values(sc, tx, s.u, s.v, 0.002f, c0, ch, cv);
    f3 bumped = n - t * (ch - c0) - b * (cv - c0);
    bumped = bumped * (1.0f / sqrtf(dot3(bumped, bumped)));            // glm::normalize
    if (dot3(bumped, n) < 0) bumped = bumped * -1.0f;
    return bumped;
}

// ====================================================================================================
// Lights and BRDFs (Q16, Q17)
// ====================================================================================================
template <bool RNG>
__device__ __forceinline__ void light_vars(const rt_light &L, int li, f3 ip, unsigned seed, const int4 meta, f3 &to_light, float &dist) {   // Light.cpp:89-129
    if (L.type == RT_LIGHT_DIRECTIONAL) {
        dist = 1e9;
        to_light = normalize3(-ld3(L.direction));
        return;
    }
    f3 pos = ld3(L.position);
    if (RNG && L.type == RT_LIGHT_AREA) {
        double u[4];
        rng_slot(seed, (unsigned)meta.x, (unsigned)meta.y, (unsigned)meta.z, SLOT_LIGHT0 + li, u);
        double p1 = uni(u[0], 0.0f, 1.0f), p2 = uni(u[1], 0.0f, 1.0f);
        p1 -= 0.5; p2 -= 0.5;
        const f3 to_right = ((float)p1 * L.size) * mk3(1.0f, 0.0f, 0.0f);
        const f3 to_up = ((float)p2 * L.size) * mk3(0.0f, 0.0f, 1.0f);
        pos = pos + to_up + to_right;
    }
    to_light = pos - ip;
    dist = norm3(to_light);
    to_light = normalize3(to_light);
}
__device__ f3 light_intensity(const rt_light &L, f3 to_light, float dist) {                                     // Light.cpp:58-87
    const f3 I = ld3(L.intensity);
    if (L.type == RT_LIGHT_DIRECTIONAL) return I;
    if (L.type == RT_LIGHT_SPOT) {
        const float deg2rad = 3.1415926f / 180.0f, rad2deg = 180.0f / 3.1415926f;
        const float cos_cov = cosf(L.coverage_deg * deg2rad / 2);
        const float theta = acosf(-dot3(to_light, ld3(L.direction)));
        const float falloff = (cosf(theta) - cos_cov) / (cosf(L.falloff_deg * deg2rad / 2) - cos_cov);
        if (theta * rad2deg > L.coverage_deg / 2) return mk3(0, 0, 0);
        f3 faded = I / (dist * dist);
        if (theta * rad2deg > L.falloff_deg / 2) faded = faded * falloff;
        return faded;
    }
    if (L.type == RT_LIGHT_AREA) return I * fabsf(dot3(-to_light, ld3(L.direction))) * L.size * L.size / (dist * dist);
    return I / (dist * dist);
}
__device__ float torrance_fr(const rt_material &m, float cosb, float ri) {                                       // brdf.cpp:18-46
    cosb = clampf(cosb, -1, 1);
    if (!(cosb > 0)) cosb = -cosb;
    const float n1 = ri;
    const float c2 = cosb * cosb, nk = n1 * n1 + m.absorption_index * m.absorption_index, n2c = 2 * n1 * cosb;
    const float Rs = (nk - n2c + c2) / (nk + n2c + c2), Rp = (nk * c2 - n2c + 1) / (nk * c2 + n2c + 1);
    return 0.5f * (Rs + Rp);
}
__device__ f3 brdf_eval(const rt_material &m, f3 wi, f3 wo, f3 n, f3 kd, f3 ks, float ri) {                      // brdf.cpp:58-170
    const float pi = 3.14159265f;
    const float e = m.brdf_exponent;
    const f3 zero = mk3(0, 0, 0);
    if (m.brdf_kind == RT_BRDF_BLINNPHONG_ORIGINAL) {
        const float cost = dot3(wi, n);
        if (cost <= 0) return zero;
        return kd * cost + ks * powf(fmaxf(0.0f, dot3(n, normalize3(wi + wo))), e);
    }
    if (m.brdf_kind == RT_BRDF_BLINNPHONG_MODIFIED) {
        const float cost = dot3(wi, n);
        if (cost < 0) return zero;
        if (!((double)acosf(cost) < pi / 2.0)) return zero;
        if (m.brdf_flag) return (kd * (1.0f / pi) + (ks * (e + 8) * powf(fmaxf(0.0f, dot3(n, normalize3(wi + wo))), e) / (8 * pi))) * cost;
        return (kd + ks * powf(fmaxf(0.0f, dot3(n, normalize3(wi + wo))), e)) * cost;
    }
    if (m.brdf_kind == RT_BRDF_PHONG_ORIGINAL) {
        const float cost = dot3(wi, n);
        if (cost < 0.0f) return zero;
        if ((double)acosf(cost) < pi / 2.0) return kd * cost + ks * powf(dot3(wo, normalize3(wi - 2.0f * n * cost)), e);
        return zero;
    }
    if (m.brdf_kind == RT_BRDF_PHONG_MODIFIED) {
        const float cost = dot3(wi, n);
        if (cost < 0.0f) return zero;
        if (!((double)acosf(cost) < pi / 2.0)) return zero;
        if (m.brdf_flag) return (kd * (1.0f / pi) + ks * (e + 2) * powf(dot3(wo, wi - 2.0f * n * cost), e) / (2.0f * pi)) * cost;
        return (kd + ks * powf(dot3(wo, wi - 2.0f * n * cost), e)) * cost;
    }
    const float cost = dot3(n, wi);                                    // Torrance-Sparrow
    if (cost < 0) return zero;
    if ((double)acosf(cost) < pi / 2.0) {
        const f3 wh = normalize3(wi + wo);
        const float cosa = fmaxf(0.0f, dot3(n, wh)), cosb = fmaxf(0.0f, dot3(wo, wh)), cosc = dot3(wo, n);
        const float dterm = powf(cosa, e) * (e + 2) / (2.0f * pi);
        const float g = fminf(1.0f, dot3(n, wh) * 2.0f / dot3(wo, wh) * fminf(dot3(n, wo), dot3(n, wi)));
        if (m.brdf_flag) return (1.0f - torrance_fr(m, cosb, ri)) * kd * (1.0f / pi) * cost + ks * g * torrance_fr(m, cosb, ri) * dterm / (4.0f * cosc);
        return kd * (1.0f / pi) * cost + ks * g * dterm / (4.0f * cosc);
    }
    return zero;
}

// ====================================================================================================
// Shading
// ====================================================================================================
__device__ __forceinline__ Surf load_surf(const LevelBuf &lv, int i, float &t) {
    const float4 h0 = lv.hit0[i], h1 = lv.hit1[i], h2 = lv.hit2[i], h3 = lv.hit3[i];
    Surf s;
    t = h0.x; s.shape = __float_as_int(h0.y); s.object = __float_as_int(h0.z); s.prim = __float_as_int(h0.w);
    s.ip = mk3(h1.x, h1.y, h1.z); s.n = mk3(h1.w, h2.x, h2.y); s.u = h2.z; s.v = h2.w; s.theta = h3.x; s.phi = h3.y;
    return s;
}
__device__ __forceinline__ bool replaces_all(const DevScene &sc, const DevObject &ob) {
    return ob.tex_color >= 0 && sc.textures[ob.tex_color].decal == RT_DECAL_REPLACE_ALL;
}

// Per hit, right after the closest-hit trace of level `depth`: complete the hit record, apply the normal texture, emit one
// shadow ray per light into the compact shadow queue, and spawn the reflection / refraction children into level depth+1
// (RecursiveComputation.cpp) together with the coefficients the backward sweep folds them with.  Nothing here needs the
// shadow results, so the children of this level and the shadow rays of this level are traced by ONE launch.
// NTEX: the scene has normal textures; RNG: it has area lights or glossy materials (the only consumers of random numbers
// here).  The plain instantiation carries neither code path (and not their registers: 2x the occupancy).
template <bool NTEX, bool RNG>
__global__ __launch_bounds__(256) void k_shade_pre(DevScene sc, DevCamera cam, LevelBuf lv, int n, const int *n_dev, ShadowBuf sb, LevelBuf next, int depth,
                                                   int *next_count, int *refr_count) {
    // Queue space is reserved once per 256-thread block (two atomics per block: same-address atomics from every wave cost
    // ~6 ns each, chip-wide serial).  Inactive threads of the last block take part in the barriers with lit = false, no children.
    __shared__ int s_lit[4], s_child[4], s_refl[4], s_base[2];
    // the level's length lives on the device; n is its capacity.  A counter above the capacity means that the level was outgrown (blocks whose
    // children did not fit wrote none): the host discards such a pass, so nothing of the level is processed instead of its holes
    if (n_dev) { const int c = *n_dev; n = c <= n ? c : 0; }
    if ((int)(blockIdx.x * blockDim.x) >= n) return;               // whole block beyond the queue (block-uniform: before any barrier)
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i0 < n;
    const int i = active ? i0 : n - 1;          // inactive threads recompute the last ray and write nothing
    const int nl = sc.p.n_lights;
    const int4 meta = lv.ray_m[i];
    const float4 ray_o = lv.ray_o[i], ray_d = lv.ray_d[i];
    const f3 o = mk3(ray_o.x, ray_o.y, ray_o.z), d = mk3(ray_d.x, ray_d.y, ray_d.z);
    const float time = ray_o.w;
    Surf s;
    {   // complete the raw record k_trace left (what Triangle::Intersect / Sphere::Intersect return for the winner)
        const float4 r0 = lv.hit0[i], r1 = lv.hit1[i];
        Cand best;
        best.t = r0.x; best.prim = __float_as_int(r0.y); best.beta = r0.z; best.gamma = r0.w; best.ip = mk3(r1.x, r1.y, r1.z);
        HitOut h;
        finalize_hit<false>(sc, o, d, time, best, h);
        s.ip = h.ip; s.n = h.n; s.u = h.u; s.v = h.v; s.theta = h.theta; s.phi = h.phi; s.shape = h.shape; s.object = h.object; s.prim = h.prim;
        if (active) {
            lv.hit0[i] = make_float4(h.t, __int_as_float(h.shape), __int_as_float(h.object), __int_as_float(h.prim));
            lv.hit3[i] = make_float4(h.theta, h.phi, 0.0f, 0.0f);
        }
    }
    bool lit = s.object >= 0, recurse = false;
    const bool internal = s.object >= 0 && meta.w >= 0 && meta.w == s.object;      // IsInternalReflection (Q26)
    int mat_index = 0;
    if (lit) {
        const DevObject &ob = sc.objects[s.object];
        mat_index = ob.material;
        if (replaces_all(sc, ob)) lit = false;                         // LightContributionCalculator.cpp:54-55
        else {
            if (NTEX && ob.tex_normal >= 0) s.n = tweak_normal(sc, s, ob.tex_normal);
            recurse = true;
            if (internal) lit = false;
        }
    }
    if (active) {
        lv.hit1[i] = make_float4(s.ip.x, s.ip.y, s.ip.z, s.n.x);
        lv.hit2[i] = make_float4(s.n.y, s.n.z, s.u, s.v);
    } else { lit = false; recurse = false; }

    // ---- children (RecursiveComputation.cpp:25-30) ----
    f3 mrc = mk3(0, 0, 0), absorb = mk3(1, 1, 1);
    float refl_coef = 0, refr_coef = 0;
    int flags = s.object >= 0 ? NODE_HIT : 0;
    bool has_refr = false, has_refl = false;
    f3 refl_o = mk3(0, 0, 0), refl_d = mk3(0, 0, 0), refr_o = mk3(0, 0, 0), refr_d = mk3(0, 0, 0);
    float refl_n = 1, refr_n = 1;
    int refl_c = -1, refr_c = -1;
    if (recurse && depth < sc.p.max_depth) {
        const rt_material &m = sc.materials[mat_index];
        if (m.type != RT_MAT_DEFAULT) {
            f3 N = normalize3(s.n);
            const float se = sc.p.shadow_eps;
            float fraction = 0;
            if (m.type == RT_MAT_DIELECTRIC || m.type == RT_MAT_CONDUCTOR) {
                // ComputeFresnel (:203-228)
                float n_out = ray_d.w, n_in = m.refraction_index;
                if (internal) n_in = 1;
                float corr = clampf(dot3(d, N), -1, 1);
                if (corr > 0) {
                    if (m.type == RT_MAT_CONDUCTOR) { float sw = n_out; n_out = n_in; n_in = sw; }
                    N = -N;
                } else corr = -corr;
                const float sin1 = sqrtf(1 - corr * corr);
                const float sin2 = (n_out * sin1) / n_in;
                const float cos2 = sqrtf(1 - sin2 * sin2);
                if (m.type == RT_MAT_DIELECTRIC) {             // Material.cpp:13-24
                    const float n2c1 = n_in * corr, n1c2 = n_out * cos2;
                    const float rpll = (n2c1 - n1c2) / (n2c1 + n1c2);
                    const float n1c1 = n_out * corr, n2c2 = n_in * cos2;
                    const float rperp = (n1c1 - n2c2) / (n1c1 + n2c2);
                    fraction = 0.5f * (rpll * rpll + rperp * rperp);
                } else {                                       // Material.cpp:26-38 (Q13)
                    const float n1 = m.refraction_index;
                    const float c2 = corr * corr, nk = n1 * n1 + m.absorption_index * m.absorption_index, n2c = 2 * n1 * corr;
                    const float Rs = (nk - n2c + c2) / (nk + n2c + c2), Rp = (nk * c2 - n2c + 1) / (nk * c2 + n2c + 1);
                    fraction = 0.5f * (Rs + Rp);
                }
                // ComputeRefractionParameters (:181-201), PerformRefraction (:154-179) (Q14)
                const float param = n_out / n_in;
                const float cost = 1 - corr * corr;
                const float det = 1 - param * param * cost;
                if (det >= 0) {
                    has_refr = true;
                    refr_d = normalize3((d + N * corr) * (param)-N * sqrtf(det));
                    refr_o = s.ip + -N * se;
                    if (!internal) { refr_n = m.refraction_index; refr_c = s.object; }
                    else { refr_n = 1.0f; refr_c = -1; }
                } else if (internal) fraction = 1;
                refr_coef = 1 - fraction;
            }
            // reflection ray (:62-122)
            const f3 to_viewer2 = normalize3(o - s.ip);
            f3 vrd = normalize3(N * 2.0f * dot3(to_viewer2, N) - to_viewer2);
            if (RNG && m.type != RT_MAT_DIELECTRIC && m.roughness > 0) {   // glossy tilt (Q15); the dielectric path does not call it
                const int mi = fabsf(vrd.x) < fabsf(vrd.y) ? (fabsf(vrd.x) < fabsf(vrd.z) ? 0 : 2) : (fabsf(vrd.y) < fabsf(vrd.z) ? 1 : 2);
                f3 rp = vrd;
                if (mi == 0) rp.x = 1.0f; else if (mi == 1) rp.y = 1.0f; else rp.z = 1.0f;
                rp = normalize3(rp);
                const f3 u = normalize3(cross3(vrd, rp));
                const f3 v = normalize3(cross3(vrd, u));
                double uu[4];
                rng_slot(cam.seed, (unsigned)meta.x, (unsigned)meta.y, (unsigned)meta.z, SLOT_GLOSSY, uu);
                const f3 r1 = (float)uni(uu[0], -0.5, 0.5) * u;
                const f3 r2 = (float)uni(uu[1], -0.5, 0.5) * v;
                vrd = normalize3(vrd + m.roughness * (r1 + r2));
            }
            has_refl = true;
            refl_o = s.ip + N * se;
            refl_d = vrd;
            refl_n = ray_d.w; refl_c = meta.w;                 // inherits currMat / currShape (Q26)
            if (m.type == RT_MAT_MIRROR) { mrc = ld3(m.km); refl_coef = m.degamma ? 2.2f : 1.0f; flags |= NODE_REFL_MRC; }
            else if (m.type == RT_MAT_CONDUCTOR) { mrc = ld3(m.km); refl_coef = fraction; flags |= NODE_REFL_MRC; }
            else {
                refl_coef = fraction; flags |= NODE_REFL_ALWAYS;
                if (internal) {                                // GetDielectricPowerAbsorptionParameter (:143-152, Q12)
                    const f3 ac = ld3(m.ac);
                    const float distance = (s.ip.x - o.x) / d.x;
                    const float ecoeff = expf(-(0.01f * distance));
                    absorb = ecoeff * mk3(expf(-ac.x * distance), expf(-ac.y * distance), expf(-ac.z * distance));
                    flags |= NODE_ABSORB;
                }
            }
        }
    }
    // ---- block-wide reservation in the shadow queue and in the next level's queue ----
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long m_lit = __ballot(lit), m1 = __ballot(has_refr), m2 = __ballot(has_refl);
    if (lane == 0) { s_lit[wave] = __popcll(m_lit); s_child[wave] = __popcll(m1); s_refl[wave] = __popcll(m2); }
    __syncthreads();
    const int block_lit = s_lit[0] + s_lit[1] + s_lit[2] + s_lit[3];
    const int block_refr = s_child[0] + s_child[1] + s_child[2] + s_child[3];
    if (threadIdx.x == 0) {
        const int block_child = block_refr + s_refl[0] + s_refl[1] + s_refl[2] + s_refl[3];
        s_base[0] = (block_lit && nl) ? atomicAdd(sb.count, block_lit * nl) : 0;
        int cb = block_child ? atomicAdd(next_count, block_child) : 0;
        // The next level's queue was sized from an estimate: a block whose children do not fit writes none.  The counter keeps
        // counting, so the host sees count > capacity afterwards and renders the batch again with exact sizes.
        if (cb + block_child > next.capacity) cb = -1;
        s_base[1] = cb;
        // how many of the children are refraction rays (the host splits the level's length into rays per kind with it)
        if (block_refr && cb >= 0 && refr_count) atomicAdd(refr_count, block_refr);
    }
    __syncthreads();
    if (s_base[1] < 0) { has_refr = false; has_refl = false; }
    // the block's refraction children sit together, then its reflection children (rays of one kind are the coherent ones)
    int lit_rank = __popcll(m_lit & below), slot_refr = s_base[1] + __popcll(m1 & below), slot_refl = s_base[1] + block_refr + __popcll(m2 & below);
    for (int w = 0; w < wave; ++w) { lit_rank += s_lit[w]; slot_refr += s_child[w]; slot_refl += s_refl[w]; }

    // shadow rays: COMPACT queue; the block's rays towards one light sit together (also the coherent order); the result
    // is written to the ray's slot (hit i, light l) by the trace kernel.
    for (int l = 0; l < nl && active; ++l) {
        const int k = i * nl + l;
        if (!lit) { sb.occluded[k] = 0; continue; }
        const int q = s_base[0] + l * block_lit + lit_rank;
        f3 to_light; float dist;
        light_vars<RNG>(sc.lights[l], l, s.ip, cam.seed, meta, to_light, dist);
        const f3 so = s.ip + s.n * sc.p.shadow_eps;                    // LightContributionCalculator.cpp:133
        sb.ldir[k] = make_float4(to_light.x, to_light.y, to_light.z, dist);
        sb.o[q] = make_float4(so.x, so.y, so.z, time);
        sb.d[q] = make_float4(to_light.x, to_light.y, to_light.z, dist);
        sb.ipref[q] = make_float4(s.ip.x, s.ip.y, s.ip.z, __int_as_float(k));
    }
    if (!active) return;
    int ci_refr = -1, ci_refl = -1;
    if (has_refr) {
        ci_refr = slot_refr;
        next.ray_o[ci_refr] = make_float4(refr_o.x, refr_o.y, refr_o.z, time);
        next.ray_d[ci_refr] = make_float4(refr_d.x, refr_d.y, refr_d.z, refr_n);
        next.ray_m[ci_refr] = make_int4(meta.x, meta.y, meta.z * 2, refr_c);
    }
    if (has_refl) {
        ci_refl = slot_refl;
        next.ray_o[ci_refl] = make_float4(refl_o.x, refl_o.y, refl_o.z, time);
        next.ray_d[ci_refl] = make_float4(refl_d.x, refl_d.y, refl_d.z, refl_n);
        next.ray_m[ci_refl] = make_int4(meta.x, meta.y, meta.z * 2 + 1, refl_c);
    }
    lv.node_a[i] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(flags));      // k_shade_post adds the local colour
    lv.node_b[i] = make_float4(mrc.x, mrc.y, mrc.z, refl_coef);
    lv.node_c[i] = make_float4(absorb.x, absorb.y, absorb.z, refr_coef);
    lv.child[i] = make_int2(ci_refr, ci_refl);
}
// Rays per kind of a level (one "ray" = one closest-hit query), only when the caller asked for rt_stats: the shadow rays and
// the children this level spawned, from the child table (block reduction, one atomic per block and kind).
__global__ __launch_bounds__(256) void k_count_kinds(LevelBuf lv, int n, const int *n_dev, const int *shadow_count, unsigned long long *kind_counts) {
    __shared__ unsigned s_r[4], s_t[4];
    if (n_dev) { const int c = *n_dev; n = c <= n ? c : 0; }
    unsigned r = 0, t = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int2 ch = lv.child[i];
        t += ch.x >= 0 ? 1u : 0u; r += ch.y >= 0 ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) { r += __shfl_down(r, off); t += __shfl_down(t, off); }
    if ((threadIdx.x & 63) == 0) { s_t[threadIdx.x >> 6] = t; s_r[threadIdx.x >> 6] = r; }
    __syncthreads();
    if (threadIdx.x == 0) {
        t = s_t[0] + s_t[1] + s_t[2] + s_t[3]; r = s_r[0] + s_r[1] + s_r[2] + s_r[3];
        if (r) atomicAdd(&kind_counts[RT_RAY_REFLECT], (unsigned long long)r);
        if (t) atomicAdd(&kind_counts[RT_RAY_REFRACT], (unsigned long long)t);
        if (blockIdx.x == 0 && shadow_count) atomicAdd(&kind_counts[RT_RAY_SHADOW], (unsigned long long)*shadow_count);
    }
}
void rt_launch_shade_pre(hipStream_t s, const DevScene &sc, const DevCamera &cam, LevelBuf lv, int n, const int *n_dev, ShadowBuf sb, LevelBuf next, int depth,
                         int *next_count, int *refr_count) {
    if (n <= 0) return;
    const dim3 g((n + 255) / 256), b(256);
    const bool ntex = sc.p.has_normal_tex != 0, rng = sc.p.has_rng != 0;
    if (ntex && rng) hipLaunchKernelGGL((k_shade_pre<true, true>), g, b, 0, s, sc, cam, lv, n, n_dev, sb, next, depth, next_count, refr_count);
    else if (ntex) hipLaunchKernelGGL((k_shade_pre<true, false>), g, b, 0, s, sc, cam, lv, n, n_dev, sb, next, depth, next_count, refr_count);
    else if (rng) hipLaunchKernelGGL((k_shade_pre<false, true>), g, b, 0, s, sc, cam, lv, n, n_dev, sb, next, depth, next_count, refr_count);
    else hipLaunchKernelGGL((k_shade_pre<false, false>), g, b, 0, s, sc, cam, lv, n, n_dev, sb, next, depth, next_count, refr_count);
}
// rays per kind of a level after its surface pass (statistics only; a launch of its own so that the host can queue it behind
// the copy of the child count)
void rt_launch_count_kinds(hipStream_t s, LevelBuf lv, int n, const int *n_dev, ShadowBuf sb, unsigned long long *kind_counts) {
    if (n <= 0 || !kind_counts) return;
    hipLaunchKernelGGL(k_count_kinds, dim3(std::min((n + 255) / 256, 512)), dim3(256), 0, s, lv, n, n_dev, sb.count, kind_counts);
}

// After the shadow rays of the level are traced: the local colour of every hit (ProcessLights, LightContributionCalculator.cpp:98-124),
// the background for misses, the texture for replace_all decals.
__global__ __launch_bounds__(256) void k_shade_post(DevScene sc, DevCamera cam, LevelBuf lv, int n, const int *n_dev, ShadowBuf sb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) { const int c = *n_dev; n = c <= n ? c : 0; }
    if (i >= n) return;
    float t;
    const Surf s = load_surf(lv, i, t);
    f3 L = mk3(0, 0, 0);
    if (s.object < 0) {                                            // miss: outColor = background (Q10)
        if (sc.p.bg_texture >= 0) L = texture_rgb(sc, sc.textures[sc.p.bg_texture], 0.0f, 0.0f, 0.0f);
        else L = ld3(sc.p.background);
    } else {
        const DevObject &ob = sc.objects[s.object];
        const rt_material &m = sc.materials[ob.material];
        if (replaces_all(sc, ob)) {
            L = texture_rgb(sc, sc.textures[ob.tex_color], s.u, s.v, 0.0f);
        } else {
            const int4 meta = lv.ray_m[i];
            const bool internal = meta.w >= 0 && meta.w == s.object;
            if (!internal) {                                       // ProcessLights :98-124
                const float4 ro = lv.ray_o[i];
                const f3 to_viewer = normalize3(mk3(ro.x, ro.y, ro.z) - s.ip);             // LightContributionCalculator.cpp:75
                f3 all = mk3(0, 0, 0);
                all = all + ld3(sc.p.ambient) * ld3(m.ka);
                const int nl = sc.p.n_lights;
                for (int l = 0; l < nl; ++l) {
                    const int k = i * nl + l;
                    f3 c = mk3(0, 0, 0);
                    if (!sb.occluded[k]) {
                        const float4 sd = sb.ldir[k];
                        const f3 to_light = mk3(sd.x, sd.y, sd.z);
                        const f3 kd = diffuse_coefficient(sc, m, ob.tex_color, s);
                        c = brdf_eval(m, to_light, to_viewer, s.n, kd, ld3(m.ks), m.refraction_index) * light_intensity(sc.lights[l], to_light, sd.w) * 1.0f;
                    }
                    all = all + c;
                }
                L = L + all;
            }
        }
    }
    float4 a = lv.node_a[i];
    a.x = L.x; a.y = L.y; a.z = L.z;
    lv.node_a[i] = a;
}
void rt_launch_shade_post(hipStream_t s, const DevScene &sc, const DevCamera &cam, LevelBuf lv, int n, const int *n_dev, ShadowBuf sb) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_shade_post, dim3((n + 255) / 256), dim3(256), 0, s, sc, cam, lv, n, n_dev, sb);
}

// Backward sweep: colour(ray) = local + children, in the association order of RecursiveComputation.cpp:62-69,130-141,172-173.
__global__ __launch_bounds__(256) void k_backward(LevelBuf lv, int n, const int *n_dev, LevelBuf next, int has_next) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) { const int c = *n_dev; n = c <= n ? c : 0; }
    if (i >= n) return;
    const float4 a = lv.node_a[i];
    const int flags = __float_as_int(a.w);
    f3 C = mk3(a.x, a.y, a.z);
    const int2 ch = lv.child[i];
    if (has_next && (ch.x >= 0 || ch.y >= 0)) {
        const float4 b = lv.node_b[i], c = lv.node_c[i];
        if (ch.x >= 0) {                                               // refraction: only if the child hit something (Q11)
            const float4 cc = next.color[ch.x];
            if (cc.w != 0.0f) C = C + mk3(cc.x, cc.y, cc.z) * c.w;
        }
        if (ch.y >= 0) {
            const float4 cc = next.color[ch.y];
            const f3 col = mk3(cc.x, cc.y, cc.z);
            if (flags & NODE_REFL_ALWAYS) C = C + col * b.w;
            else if (cc.w != 0.0f) C = C + col * mk3(b.x, b.y, b.z) * b.w;
        }
        if (flags & NODE_ABSORB) C = C * mk3(c.x, c.y, c.z);
    }
    lv.color[i] = make_float4(C.x, C.y, C.z, (flags & NODE_HIT) ? 1.0f : 0.0f);
}
void rt_launch_backward(hipStream_t s, LevelBuf lv, int n, const int *n_dev, LevelBuf next, int has_next) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_backward, dim3((n + 255) / 256), dim3(256), 0, s, lv, n, n_dev, next, has_next);
}

// RenderMultiSampled's mean (DefaultRenderer.cpp:111-122) / RenderWithOneSample; also the primary hit id of sample 0.
__global__ __launch_bounds__(256) void k_resolve(DevCamera cam, const int *pixel_list, int first_pixel, int n_pixels, LevelBuf lv0, int compact,
                                                 float *hdr, int *prim_id, float *prim_t) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    const int spp = cam.spp;
    f3 sum = mk3(0, 0, 0);
    if (spp > 1) {
        for (int s = 0; s < spp; ++s) { const float4 c = lv0.color[p * spp + s]; sum = sum + mk3(c.x, c.y, c.z); }
        sum = mk3(sum.x / spp, sum.y / spp, sum.z / spp);
    } else {
        const float4 c = lv0.color[p];
        sum = mk3(c.x, c.y, c.z);
    }
    const size_t dst = compact ? (size_t)(first_pixel + p) : (size_t)pixel_list[first_pixel + p];
    hdr[3 * dst + 0] = sum.x; hdr[3 * dst + 1] = sum.y; hdr[3 * dst + 2] = sum.z;
    if (prim_id || prim_t) {
        const float4 h0 = lv0.hit0[p * spp];
        if (prim_id) prim_id[dst] = __float_as_int(h0.y);
        if (prim_t) prim_t[dst] = h0.x;
    }
}
void rt_launch_resolve(hipStream_t s, const DevCamera &cam, const int *pixel_list, int first_pixel, int n_pixels, LevelBuf lv0, int compact,
                       float *hdr, int *prim_id, float *prim_t) {
    if (n_pixels <= 0) return;
    hipLaunchKernelGGL(k_resolve, dim3((n_pixels + 255) / 256), dim3(256), 0, s, cam, pixel_list, first_pixel, n_pixels, lv0, compact, hdr, prim_id, prim_t);
}
