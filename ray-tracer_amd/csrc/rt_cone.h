// rt_cone.h — normal cones: when may the traversal skip a subtree because its box begins beyond the ray's best hit?
//
// The reference visits EVERY leaf whose box the ray hits (BVHTree.cpp:160-206, no pruning by distance) and its triangle
// test (Triangle.cpp:186-250) is a Cramer solve in float: t = det(e1, e2, p0 - o) / det(e1, e2, d).  When the ray lies
// almost in the triangle's plane, or the triangle is a sliver, the divisor is rounding noise and the reference accepts
// "hits" whose point lies anywhere on the ray — far outside the triangle's box, possibly in front of the ray's real hit.
// A traversal that prunes by distance never tests such a triangle and would return a different primitive.
//
// How wrong the solve can be is governed by one number, the conditioning of (ray, triangle):
//     rho = |(e1 x e2) . d| / (|e1| |e2| |d|)  =  sin(angle between the edges) * |cos(angle between normal and ray)|.
// The computed hit point of an ACCEPTED hit lies within c * u * D / rho of the triangle (u = 2^-24, D = distances involved,
// c a small constant); measured over 10^8 (ray, triangle) pairs the excursion beyond the box is <= 6e-5 / rho of the
// pruning margin the traversal already keeps (DevSceneParams::prune_err), so for rho >= RT_CONE_RHO_MIN = 1e-2 it stays
// 100x inside that margin and pruning by distance is safe.  For rho below that a subtree must not be pruned.
//
// Per subtree the builder therefore keeps a cone of (unoriented) triangle normals in WORLD space and the smallest
// edge-angle sine s of its triangles; a child of a wide node carries both packed into 32 bits:
//     bits  0..7, 8..15, 16..23   cone axis: three integer components in [-127, 127], stored + 127 (largest component = +-127)
//     bits 24..31                 k: the subtree may be pruned by distance iff |axis_int . d| >= k * (220 / 255) * |d|
// k is rounded up from sin(half-angle + asin(rho_min / s) + what the integer axis loses) * |axis_int|; k = 255 never allows it
// (|axis_int| < 220: wide cones, slivers), k = 0 always does (a subtree of spheres only: their test is a well-conditioned quadratic).
// The test is evaluated for every far child of every node step, so the word is made to decode in three byte conversions.
// An object transform M = [A | t] maps object-space normals by inverse(transpose(A)); the object-space conditioning is at least
// |m_w . d_w| / |d_w| with m_w = (sigma_min(A)^2 / sigma_max(A)) * inverse(transpose(A)) * (s * n_obj), which also covers the
// growth of the object-space error on its way back to world space (DevObject::cone_k holds the factor).
#ifndef RT_CONE_H_
#define RT_CONE_H_

#include "rt_math.h"

#define RT_CONE_RHO_MIN 1e-2f
#define RT_CONE_FULL_ALPHA 1.5f        // half-angles from here on never allow pruning (alpha + asin(rho/s) must stay below pi/2)
#define RT_CONE_NEVER 255u
#define RT_CONE_WORD_NEVER ((RT_CONE_NEVER << 24) | 0x7f7f7fu)
#define RT_CONE_WORD_ALWAYS 0u
#define RT_CONE_K_SCALE (220.0f / 255.0f)

struct NCone {
    f3 a;            // unit axis (sign irrelevant: normals are lines)
    float alpha;     // half-angle in radians; < 0: empty set; >= RT_CONE_FULL_ALPHA: everything
    float smin;      // smallest conditioning factor s_w of the triangles inside
};

RT_HD NCone cone_empty() { NCone c; c.a = mk3(0, 0, 1); c.alpha = -1.0f; c.smin = 1.0f; return c; }
RT_HD NCone cone_full() { NCone c; c.a = mk3(0, 0, 1); c.alpha = 4.0f; c.smin = 0.0f; return c; }
RT_HD bool cone_is_full(const NCone &c) { return !(c.alpha < RT_CONE_FULL_ALPHA) || !(c.smin > 0.0f); }

// angle between two unit vectors, accurate for small angles
RT_HD float cone_angle(f3 a, f3 b) {
    const float h = 0.5f * norm3(a - b);
    return 2.0f * asinf(h < 1.0f ? h : 1.0f);
}

// Cone of one triangle (edges e1 = p0 - p1, e2 = p0 - p2 in object space).  inv = inverse(transpose(M)) of its object or
// nullptr (no transform), k = DevObject::cone_k.
RT_HD NCone cone_of_triangle(f3 e1, f3 e2, const float *inv, float k) {
    const f3 n = cross3(e1, e2);
    const float ln = norm3(n), le = norm3(e1) * norm3(e2);
    if (!(ln > 0.0f) || !(le > 0.0f) || !(ln < 3.0e38f) || !(le < 3.0e38f)) return cone_full();
    float s = ln / le;
    f3 a = n / ln;
    if (inv) {
        const f3 v = mat_mul_xyz(inv, a.x, a.y, a.z, 0.0f);
        const float lv = norm3(v);
        if (!(lv > 0.0f) || !(lv < 3.0e38f)) return cone_full();
        a = v / lv;
        s = s * k * lv;
    }
    if (!(s > 0.0f)) return cone_full();
    NCone c;
    c.a = a; c.alpha = 0.0f; c.smin = s < 1.0f ? s : 1.0f;
    return c;
}

// Smallest cone (up to the padding) that contains both; axes are compared as lines.
RT_HD NCone cone_merge(const NCone &A, const NCone &B0) {
    if (A.alpha < 0.0f) return B0;
    if (B0.alpha < 0.0f) return A;
    if (cone_is_full(A) || cone_is_full(B0)) return cone_full();
    NCone B = B0;
    if (dot3(A.a, B.a) < 0.0f) B.a = -B.a;
    const float smin = A.smin < B.smin ? A.smin : B.smin;
    const float th = cone_angle(A.a, B.a);
    NCone r;
    r.smin = smin;
    if (th + B.alpha <= A.alpha) { r.a = A.a; r.alpha = A.alpha; return r; }
    if (th + A.alpha <= B.alpha) { r.a = B.a; r.alpha = B.alpha; return r; }
    const float alpha = 0.5f * (A.alpha + B.alpha + th) + 2e-5f;
    if (!(alpha < RT_CONE_FULL_ALPHA)) return cone_full();
    if (th < 1e-5f) { r.a = A.a; r.alpha = (A.alpha > B.alpha ? A.alpha : B.alpha) + th + 2e-5f; return r; }
    const float phi = alpha - A.alpha;                       // rotate A's axis by phi towards B's (0 <= phi <= th)
    const f3 ax = A.a * sinf(th - phi) + B.a * sinf(phi);    // slerp, unnormalised
    const float l = norm3(ax);
    if (!(l > 1e-6f)) return cone_full();
    r.a = ax / l; r.alpha = alpha;
    return r;
}

// ---- packing ---------------------------------------------------------------------------------------------------------
// axis as the decoder reconstructs it (integer components, not normalised)
RT_HD f3 cone_axis_decode(unsigned w) {
    return mk3((float)(w & 255u) - 127.0f, (float)((w >> 8) & 255u) - 127.0f, (float)((w >> 16) & 255u) - 127.0f);
}
// May a subtree with cone word w be pruned by distance for a ray of direction d (|d| = dlen)?
RT_HD bool cone_allows_prune(unsigned w, float dx, float dy, float dz, float dlen) {
    const f3 a = cone_axis_decode(w);
    const float dot = fmaf(a.x, dx, fmaf(a.y, dy, a.z * dz));
    return fabsf(dot) >= (float)(w >> 24) * (RT_CONE_K_SCALE * dlen);
}
RT_HD unsigned cone_pack(const NCone &c) {
    if (c.alpha < 0.0f) return RT_CONE_WORD_ALWAYS;
    if (cone_is_full(c)) return RT_CONE_WORD_NEVER;
    const float q = RT_CONE_RHO_MIN / c.smin;
    if (!(q < 1.0f)) return RT_CONE_WORD_NEVER;
    const float ax = fabsf(c.a.x), ay = fabsf(c.a.y), az = fabsf(c.a.z);
    const float m = ax > ay ? (ax > az ? ax : az) : (ay > az ? ay : az);
    if (!(m > 0.0f) || !(m < 3.0e38f)) return RT_CONE_WORD_NEVER;
    float comp[3] = {c.a.x / m * 127.0f, c.a.y / m * 127.0f, c.a.z / m * 127.0f};
    unsigned uv = 0;
    for (int i = 0; i < 3; ++i) {
        float r = floorf(comp[i] + 0.5f);
        r = r < -127.0f ? -127.0f : (r > 127.0f ? 127.0f : r);
        if (!(r == r)) return RT_CONE_WORD_NEVER;
        uv |= (unsigned)(int)(r + 127.0f) << (8 * i);
    }
    const f3 ad = cone_axis_decode(uv);
    const float lad = norm3(ad);
    if (!(lad > 0.0f)) return RT_CONE_WORD_NEVER;
    f3 an = ad / lad;
    if (dot3(an, c.a) < 0.0f) an = -an;
    const float delta = cone_angle(an, c.a);                  // what the integer axis loses
    const float beta = c.alpha + asinf(q) + delta + 1e-4f;
    if (!(beta < 1.55f)) return RT_CONE_WORD_NEVER;
    // the decoder's dot product carries three roundings of terms up to 220 |d| (about 4e-5 |d|): the 1e-3 covers them
    const float thr = sinf(beta) * lad * 1.00001f + 1e-3f;
    const float fk = ceilf(thr / RT_CONE_K_SCALE);
    if (!(fk < 255.0f)) return RT_CONE_WORD_NEVER;
    return uv | ((unsigned)fk << 24);
}

#endif
