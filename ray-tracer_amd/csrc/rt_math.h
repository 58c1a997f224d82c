// rt_math.h — float3 / 4x4 helpers for host flattening and gfx950 kernels.
//
// Bit-exactness contract: every expression keeps the operation order of the reference's scalar C++
// (src/acmath.h) and of the vendored glm it calls, the translation unit is compiled with
// -ffp-contract=off, '/' and sqrtf are IEEE correctly rounded (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), and min/max are written as the reference's ternaries
// so NaN propagation is identical (Q3).  Geometry therefore agrees with the CPU bit for bit; only
// libm-style functions (powf, expf, acosf, atan2f, sinf, cosf) may differ in the last ulp.
#ifndef RT_MATH_H_
#define RT_MATH_H_

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define RT_HD __host__ __device__ __forceinline__

struct f3 {
    float x, y, z;
};

RT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
RT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_HD f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }   // acmath.h:671: scalar*v == v*scalar
RT_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_HD float norm3(f3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }   // "SqLength" (Q1)
RT_HD f3 normalize3(f3 v) { float l = norm3(v); return v / l; }                // acmath.h:591
RT_HD float sel_max(float a, float b) { return a < b ? b : a; }                 // SetMax, acmath.h:687
RT_HD float sel_min(float a, float b) { return a > b ? b : a; }                 // SetMin, acmath.h:690
RT_HD f3 max3(f3 a, f3 b) { return mk3(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z); }
RT_HD f3 min3(f3 a, f3 b) { return mk3(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z); }
RT_HD float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
RT_HD float comp3(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

// ---- 4x4, glm order m[col*4+row] -------------------------------------------------------------
// mat * (x,y,z,w), vendored glm type_mat4x4.inl:526-537: (m0*x + m1*y) + (m2*z + m3*w), xyz rows only.
// A triangle whose edges p0p1, p0p2 are (nearly) parallel: the three determinants of Triangle::Intersect (Triangle.cpp:197-240) are
// then differences of products that cancel to rounding noise, and the reference accepts "hits" whose point lies anywhere on the
// ray — metres outside the triangle's box (found by the fuzz sweep: a 5 x 0.005 sliver, hit point 6.4 units outside).  The
// traversal may not prune such a primitive's box by distance.  Criterion: sin(angle between the edges) < 1e-2, or no area at all.
RT_HD bool tri_ill_conditioned(f3 e1, f3 e2) {
    const f3 n = cross3(e1, e2);
    const float nn = dot3(n, n), ee = dot3(e1, e1) * dot3(e2, e2);
    return !(nn >= 1e-4f * ee) || !(ee > 0.0f);
}

RT_HD f3 mat_mul_xyz(const float *m, float x, float y, float z, float w) {
    f3 r;
    r.x = (m[0] * x + m[4] * y) + (m[8] * z + m[12] * w);
    r.y = (m[1] * x + m[5] * y) + (m[9] * z + m[13] * w);
    r.z = (m[2] * x + m[6] * y) + (m[10] * z + m[14] * w);
    return r;
}
// transpose(m) * (x,y,z,w): element [c][k] of the transpose is m[k*4+c].
RT_HD f3 matT_mul_xyz(const float *m, float x, float y, float z, float w) {
    f3 r;
    r.x = (m[0] * x + m[1] * y) + (m[2] * z + m[3] * w);
    r.y = (m[4] * x + m[5] * y) + (m[6] * z + m[7] * w);
    r.z = (m[8] * x + m[9] * y) + (m[10] * z + m[11] * w);
    return r;
}
RT_HD void mat_transpose(const float *a, float *r) {
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 4; ++k) r[c * 4 + k] = a[k * 4 + c];
}
RT_HD bool mat_is_identity(const float *a) {
    bool ok = true;
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 4; ++k) ok = ok && (a[c * 4 + k] == ((c == k) ? 1.0f : 0.0f));
    return ok;
}
// glm::inverse(mat4), vendored func_matrix.inl:296-354 (cofactor form, same association).
RT_HD void mat_inverse(const float *m, float *o) {
#define M_(c, r) m[(c) * 4 + (r)]
    float c00 = M_(2, 2) * M_(3, 3) - M_(3, 2) * M_(2, 3);
    float c02 = M_(1, 2) * M_(3, 3) - M_(3, 2) * M_(1, 3);
    float c03 = M_(1, 2) * M_(2, 3) - M_(2, 2) * M_(1, 3);
    float c04 = M_(2, 1) * M_(3, 3) - M_(3, 1) * M_(2, 3);
    float c06 = M_(1, 1) * M_(3, 3) - M_(3, 1) * M_(1, 3);
    float c07 = M_(1, 1) * M_(2, 3) - M_(2, 1) * M_(1, 3);
    float c08 = M_(2, 1) * M_(3, 2) - M_(3, 1) * M_(2, 2);
    float c10 = M_(1, 1) * M_(3, 2) - M_(3, 1) * M_(1, 2);
    float c11 = M_(1, 1) * M_(2, 2) - M_(2, 1) * M_(1, 2);
    float c12 = M_(2, 0) * M_(3, 3) - M_(3, 0) * M_(2, 3);
    float c14 = M_(1, 0) * M_(3, 3) - M_(3, 0) * M_(1, 3);
    float c15 = M_(1, 0) * M_(2, 3) - M_(2, 0) * M_(1, 3);
    float c16 = M_(2, 0) * M_(3, 2) - M_(3, 0) * M_(2, 2);
    float c18 = M_(1, 0) * M_(3, 2) - M_(3, 0) * M_(1, 2);
    float c19 = M_(1, 0) * M_(2, 2) - M_(2, 0) * M_(1, 2);
    float c20 = M_(2, 0) * M_(3, 1) - M_(3, 0) * M_(2, 1);
    float c22 = M_(1, 0) * M_(3, 1) - M_(3, 0) * M_(1, 1);
    float c23 = M_(1, 0) * M_(2, 1) - M_(2, 0) * M_(1, 1);
    const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    const float f3_[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {M_(1, 0), M_(0, 0), M_(0, 0), M_(0, 0)}, v1[4] = {M_(1, 1), M_(0, 1), M_(0, 1), M_(0, 1)};
    const float v2[4] = {M_(1, 2), M_(0, 2), M_(0, 2), M_(0, 2)}, v3[4] = {M_(1, 3), M_(0, 3), M_(0, 3), M_(0, 3)};
    for (int k = 0; k < 4; ++k) {
        float sa = (k & 1) ? -1.0f : 1.0f, sb = (k & 1) ? 1.0f : -1.0f;
        o[0 + k] = ((v1[k] * f0[k] - v2[k] * f1[k]) + v3[k] * f2[k]) * sa;
        o[4 + k] = ((v0[k] * f0[k] - v2[k] * f3_[k]) + v3[k] * f4[k]) * sb;
        o[8 + k] = ((v0[k] * f1[k] - v1[k] * f3_[k]) + v3[k] * f5[k]) * sa;
        o[12 + k] = ((v0[k] * f2[k] - v1[k] * f4[k]) + v2[k] * f5[k]) * sb;
    }
    float d0 = M_(0, 0) * o[0], d1 = M_(0, 1) * o[4], d2 = M_(0, 2) * o[8], d3 = M_(0, 3) * o[12];
    float det = (d0 + d1) + (d2 + d3);
    float ood = 1.0f / det;
    for (int i = 0; i < 16; ++i) o[i] = o[i] * ood;
#undef M_
}
// (T * M) where T = glm::translate(mat4(1), v): the matrix a motion-blurred shape uses at ray time
// (Shape.cpp:73-76, Triangle.cpp:170-173, acmath.h:1004-1008, glm mat*mat type_mat4x4.inl:595-613).
RT_HD void mat_translate_mul(f3 v, const float *M, float *r) {
    float t3[4];
    {   // fourth column of glm::translate(I, v): m0*vx + m1*vy + m2*vz + m3
        t3[0] = ((1.0f * v.x + 0.0f * v.y) + 0.0f * v.z) + 0.0f;
        t3[1] = ((0.0f * v.x + 1.0f * v.y) + 0.0f * v.z) + 0.0f;
        t3[2] = ((0.0f * v.x + 0.0f * v.y) + 1.0f * v.z) + 0.0f;
        t3[3] = ((0.0f * v.x + 0.0f * v.y) + 0.0f * v.z) + 1.0f;
    }
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 4; ++k) {
            float a0 = (k == 0) ? 1.0f : 0.0f, a1 = (k == 1) ? 1.0f : 0.0f, a2 = (k == 2) ? 1.0f : 0.0f;
            r[c * 4 + k] = ((a0 * M[c * 4 + 0] + a1 * M[c * 4 + 1]) + a2 * M[c * 4 + 2]) + t3[k] * M[c * 4 + 3];
        }
}

#endif
