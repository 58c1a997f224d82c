// PLY body decoding on the device (the step before the scene upload: PLY file -> position and index arrays).
//
// Replaces what SceneParser.cpp:983-1023 does with happly's vector<array<double,3>> / vector<vector<size_t>>: vertex
// positions become float (Vector3f(a[0], a[1], a[2]): a double property is rounded to float, a float property is taken as
// it is), every face of three indices becomes one triangle, every face of four becomes (0,1,2), (2,3,0), faces of any
// other size are dropped.  The header (text) is parsed by the caller, who describes the binary_little_endian body in
// rt_ply_layout.
//
// Faces are variable-length records, so the byte offset of a face depends on all faces before it.  Files in practice have
// one face size: the records are then read at i * record_size and a flag is raised if any count differs from the first
// one; only then the host walks the counts once (a dependent chain, nothing to parallelise), uploads the offsets and the
// triangles are placed by an exclusive scan of the per-face triangle counts.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "rt_wavefront.h"

namespace {

__device__ __forceinline__ unsigned load_le(const uint8_t *p, int size) {
    unsigned v = p[0];
    if (size >= 2) v |= (unsigned)p[1] << 8;
    if (size == 4) v |= (unsigned)p[2] << 16 | (unsigned)p[3] << 24;
    return v;
}
__device__ __forceinline__ float load_coord(const uint8_t *p, int type) {
    if (type == 0) return __uint_as_float(load_le(p, 4));
    const unsigned long long lo = load_le(p, 4), hi = load_le(p + 4, 4);
    return (float)__longlong_as_double((long long)(lo | hi << 32));      // round to nearest even, as the C++ conversion
}

__global__ __launch_bounds__(256) void k_ply_vertices(const uint8_t *vb, long long nv, int stride, int ox, int oy, int oz, int type, float *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const uint8_t *p = vb + i * (long long)stride;
    out[3 * i + 0] = load_coord(p + ox, type);
    out[3 * i + 1] = load_coord(p + oy, type);
    out[3 * i + 2] = load_coord(p + oz, type);
}

__device__ __forceinline__ void write_face(const uint8_t *idx, int isize, unsigned count, int *out) {
    if (count == 3) {
        out[0] = (int)load_le(idx, isize); out[1] = (int)load_le(idx + isize, isize); out[2] = (int)load_le(idx + 2 * isize, isize);
    } else if (count == 4) {                                                     // SceneParser.cpp:995-1011
        const int a = (int)load_le(idx, isize), b = (int)load_le(idx + isize, isize), c = (int)load_le(idx + 2 * isize, isize), d = (int)load_le(idx + 3 * isize, isize);
        out[0] = a; out[1] = b; out[2] = c; out[3] = c; out[4] = d; out[5] = a;
    }
}

// every face has `k0` indices (verified here: *mismatch is set otherwise and the output is discarded by the caller)
__global__ __launch_bounds__(256) void k_ply_faces_uniform(const uint8_t *fb, long long nf, int pre, int csize, int isize, int post, unsigned k0, int *out, int *mismatch) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const long long rec = (long long)pre + csize + (long long)k0 * isize + post;
    const uint8_t *p = fb + i * rec + pre;
    const unsigned count = load_le(p, csize);
    if (count != k0) { *mismatch = 1; return; }
    const int per = k0 == 4 ? 2 : 1;
    write_face(p + csize, isize, count, out + 3 * per * i);
}

__global__ __launch_bounds__(256) void k_ply_face_tris(const uint8_t *fb, const long long *offs, long long nf, int pre, int csize, int *ntri) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nf) return;
    if (i == nf) { ntri[i] = 0; return; }                                        // the scan's last element = total
    const unsigned count = load_le(fb + offs[i] + pre, csize);
    ntri[i] = count == 3 ? 1 : (count == 4 ? 2 : 0);
}
__global__ __launch_bounds__(256) void k_ply_faces_mixed(const uint8_t *fb, const long long *offs, const int *tri_off, long long nf, int pre, int csize, int isize, int *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const uint8_t *p = fb + offs[i] + pre;
    write_face(p + csize, isize, load_le(p, csize), out + 3 * (long long)tri_off[i]);
}

// ---- ascii bodies (happly parseASCII, src/happly.h:2075-2095: one element per line, tokens split at ' ') -------------------
__global__ __launch_bounds__(256) void k_ascii_newlines(const uint8_t *text, long long n, int *flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    flag[i] = (i < n && text[i] == '\n') ? 1 : 0;                     // flag[n] = 0: scan[n] = number of newlines
}
// line_start[k] = first byte of line k (line 0 starts at 0; line k > 0 after the k-th newline)
__global__ __launch_bounds__(256) void k_ascii_line_starts(const uint8_t *text, long long n, const int *scan, long long n_lines, long long *line_start) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0) line_start[0] = 0;
    if (text[i] == '\n') { const long long k = (long long)scan[i] + 1; if (k <= n_lines) line_start[k] = i + 1; }
}
// Token scanner of one line: tokens are maximal runs of bytes other than ' ' (tokenSplit, src/happly.h:1390-1413); '\r' and
// '\n' at a token's end are trimmed (trimSpaces, :1379-1388).
struct Tok { const uint8_t *p; int len; };
__device__ __forceinline__ bool next_token(const uint8_t *&cur, const uint8_t *end, Tok &t) {
    for (;;) {
        while (cur < end && *cur == ' ') ++cur;
        if (cur >= end) return false;
        const uint8_t *b = cur;
        while (cur < end && *cur != ' ') ++cur;
        int len = (int)(cur - b);
        while (len > 0 && (b[len - 1] == '\r' || b[len - 1] == '\n' || b[len - 1] == ' ')) --len;
        if (len > 0) { t.p = b; t.len = len; return true; }
    }
}
// Non-negative decimal integer; false for anything else (the caller reports the file as unsupported).
__device__ __forceinline__ bool parse_uint(const Tok &t, unsigned &v) {
    unsigned long long a = 0;
    if (t.len <= 0 || t.len > 10) return false;
    for (int i = 0; i < t.len; ++i) { const unsigned c = t.p[i] - '0'; if (c > 9) return false; a = a * 10 + c; }
    if (a > 0x7fffffffull) return false;
    v = (unsigned)a;
    return true;
}
// Decimal token -> float, the value operator>> of the property's type gives (strtof for a float property; strtod, then the
// conversion to float, for a double property).  Returns false when the device cannot PROVE its result is that value; the host
// then converts the token with the stream extraction itself.
//   * sign, digits, optional '.', digits, optional exponent; at most 19 significant digits (an exact 64-bit integer M)
//   * float property, M < 2^24 and |E| <= 10: M and 10^|E| are exact floats, one correctly rounded operation
//   * otherwise M < 2^53 and |E| <= 22: the same in double (correctly rounded double, Clinger); for a float property the
//     second rounding to float is the same as a direct one unless the double is exactly halfway between two floats
__device__ bool parse_decimal(const Tok &t, int is_double, float &out) {
    const uint8_t *p = t.p, *e = t.p + t.len;
    bool neg = false;
    if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    unsigned long long M = 0;
    int digits = 0, exp10 = 0, any = 0;
    bool seen_dot = false;
    for (; p < e; ++p) {
        const unsigned c = *p - '0';
        if (c <= 9) {
            any = 1;
            if (M == 0 && c == 0) { if (seen_dot) --exp10; continue; }      // leading zeros
            if (digits >= 19) return false;
            M = M * 10 + c; ++digits;
            if (seen_dot) --exp10;
        } else if (*p == '.' && !seen_dot) seen_dot = true;
        else break;
    }
    if (!any) return false;
    if (p < e) {
        if (*p != 'e' && *p != 'E') return false;
        ++p;
        bool eneg = false;
        if (p < e && (*p == '-' || *p == '+')) { eneg = *p == '-'; ++p; }
        if (p >= e) return false;
        int ev = 0;
        for (; p < e; ++p) { const unsigned c = *p - '0'; if (c > 9 || ev > 9999) return false; ev = ev * 10 + (int)c; }
        exp10 += eneg ? -ev : ev;
    }
    if (M == 0) { out = neg ? -0.0f : 0.0f; return true; }
    const int ae = exp10 < 0 ? -exp10 : exp10;
    if (!is_double && M < (1ull << 24) && ae <= 10) {
        const float p10[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
        const float m = (float)M;
        const float v = exp10 < 0 ? m / p10[ae] : m * p10[ae];
        out = neg ? -v : v;
        return true;
    }
    if (M >= (1ull << 53) || ae > 22) return false;
    const double q10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const double m = (double)M;
    const double dv = exp10 < 0 ? m / q10[ae] : m * q10[ae];
    if (!(dv >= 1.2e-38 && dv <= 3.4e38)) return false;              // outside the normal float range: left to the host
    if (!is_double && ((unsigned long long)__double_as_longlong(dv) & 0x1fffffffull) == 0x10000000ull) return false;   // exactly between two floats
    const float v = (float)dv;
    out = neg ? -v : v;
    return true;
}
// One vertex line: x, y, z are the tokens number tx, ty, tz.  A token the device cannot convert is appended to the host's list
// (hard[0] = count, then (vertex, coordinate, byte offset of the token) triples); a line with too few tokens raises *bad.
__global__ __launch_bounds__(256) void k_ascii_vertices(const uint8_t *text, const long long *line_start, long long text_len, long long nv, int tx, int ty, int tz,
                                                       int is_double, float *out, int *hard, int hard_cap, int *bad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const uint8_t *cur = text + line_start[i], *end = text + (line_start[i + 1] > 0 ? line_start[i + 1] : text_len);
    const int last = tx > ty ? (tx > tz ? tx : tz) : (ty > tz ? ty : tz);
    int found = 0;
    for (int k = 0; k <= last; ++k) {
        Tok t;
        if (!next_token(cur, end, t)) { atomicOr(bad, 1); break; }
        const int c = k == tx ? 0 : (k == ty ? 1 : (k == tz ? 2 : -1));
        if (c < 0) continue;
        float v = 0.0f;
        if (!parse_decimal(t, is_double, v)) {
            const int slot = atomicAdd(hard, 1);
            if (slot < hard_cap) { hard[1 + 3 * slot] = (int)i; hard[2 + 3 * slot] = c; hard[3 + 3 * slot] = (int)(t.p - text); }
        }
        out[3 * i + c] = v;
        ++found;
    }
    if (found != 3) atomicOr(bad, 1);
}
// Faces: `pre` scalar tokens, the list count, the indices; a face of three indices is one triangle, of four two (SceneParser.cpp:
// 995-1011), any other none.  Pass 1 counts, pass 2 (after the scan) writes.
__device__ __forceinline__ bool face_header(const uint8_t *&cur, const uint8_t *end, int pre, unsigned &count) {
    Tok t;
    for (int k = 0; k < pre; ++k) if (!next_token(cur, end, t)) return false;
    if (!next_token(cur, end, t)) return false;
    return parse_uint(t, count);
}
__global__ __launch_bounds__(256) void k_ascii_face_tris(const uint8_t *text, const long long *line_start, long long text_len, long long first_line, long long nf, int pre,
                                                        int *ntri, int *bad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nf) return;
    if (i == nf) { ntri[i] = 0; return; }
    const long long ln = first_line + i;
    const uint8_t *cur = text + line_start[ln], *end = text + (line_start[ln + 1] > 0 ? line_start[ln + 1] : text_len);
    unsigned count = 0;
    if (!face_header(cur, end, pre, count)) { atomicOr(bad, 2); ntri[i] = 0; return; }
    ntri[i] = count == 3 ? 1 : (count == 4 ? 2 : 0);
}
__global__ __launch_bounds__(256) void k_ascii_faces(const uint8_t *text, const long long *line_start, long long text_len, long long first_line, long long nf, int pre,
                                                    const int *tri_off, int *out, int *bad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const long long ln = first_line + i;
    const uint8_t *cur = text + line_start[ln], *end = text + (line_start[ln + 1] > 0 ? line_start[ln + 1] : text_len);
    unsigned count = 0;
    if (!face_header(cur, end, pre, count) || (count != 3 && count != 4)) return;
    unsigned idx[4] = {0, 0, 0, 0};
    for (unsigned k = 0; k < count; ++k) {
        Tok t;
        if (!next_token(cur, end, t) || !parse_uint(t, idx[k])) { atomicOr(bad, 2); return; }
    }
    int *o = out + 3ll * tri_off[i];
    o[0] = (int)idx[0]; o[1] = (int)idx[1]; o[2] = (int)idx[2];
    if (count == 4) { o[3] = (int)idx[2]; o[4] = (int)idx[3]; o[5] = (int)idx[0]; }
}
__global__ __launch_bounds__(256) void k_offset_copy(const int *src, int *dst, long long n, int offset) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + offset;
}

}  // namespace

hipError_t rt_ply_ascii_scan_bytes(int64_t n_text, size_t *bytes) {
    *bytes = 0;
    return hipcub::DeviceScan::ExclusiveSum(nullptr, *bytes, (const int *)nullptr, (int *)nullptr, (int)(n_text + 1), (hipStream_t) nullptr);
}
// flag / scan: n_text + 1 ints; line_start: n_lines + 2 entries, zeroed by the caller (0 = "runs to the end of the text")
hipError_t rt_ply_ascii_lines(hipStream_t s, const uint8_t *text, int64_t n_text, int *flag, int *scan, void *scan_tmp, size_t scan_bytes, int64_t n_lines, long long *line_start) {
    const dim3 g((unsigned)((n_text + 1 + 255) / 256)), b(256);
    hipLaunchKernelGGL(k_ascii_newlines, g, b, 0, s, text, (long long)n_text, flag);
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, flag, scan, (int)(n_text + 1), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_ascii_line_starts, g, b, 0, s, text, (long long)n_text, scan, (long long)n_lines, line_start);
    return hipGetLastError();
}
hipError_t rt_ply_ascii_vertices(hipStream_t s, const uint8_t *text, const long long *line_start, int64_t n_text, int64_t nv, const int tok[3], int is_double, float *out,
                                 int *hard, int hard_cap, int *bad) {
    if (nv <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ascii_vertices, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s, text, line_start, (long long)n_text, (long long)nv, tok[0], tok[1], tok[2],
                       is_double, out, hard, hard_cap, bad);
    return hipGetLastError();
}
hipError_t rt_ply_ascii_faces_count(hipStream_t s, const uint8_t *text, const long long *line_start, int64_t n_text, int64_t first_line, int64_t nf, int pre, int *ntri,
                                    int *tri_off, void *scan_tmp, size_t scan_bytes, int *bad) {
    hipLaunchKernelGGL(k_ascii_face_tris, dim3((unsigned)((nf + 1 + 255) / 256)), dim3(256), 0, s, text, line_start, (long long)n_text, (long long)first_line, (long long)nf, pre, ntri, bad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, ntri, tri_off, (int)(nf + 1), s);
}
hipError_t rt_ply_ascii_faces_write(hipStream_t s, const uint8_t *text, const long long *line_start, int64_t n_text, int64_t first_line, int64_t nf, int pre,
                                    const int *tri_off, int *out, int *bad) {
    if (nf <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ascii_faces, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, text, line_start, (long long)n_text, (long long)first_line, (long long)nf, pre, tri_off, out, bad);
    return hipGetLastError();
}
hipError_t rt_ply_offset_copy(hipStream_t s, const int *src, int *dst, int64_t n, int offset) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_offset_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, (long long)n, offset);
    return hipGetLastError();
}

hipError_t rt_ply_vertices(hipStream_t s, const uint8_t *vb, int64_t nv, int stride, const int off[3], int type, float *out) {
    if (nv <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ply_vertices, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s, vb, (long long)nv, stride, off[0], off[1], off[2], type, out);
    return hipGetLastError();
}
hipError_t rt_ply_faces_uniform(hipStream_t s, const uint8_t *fb, int64_t nf, int pre, int csize, int isize, int post, unsigned k0, int *out, int *mismatch) {
    if (nf <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ply_faces_uniform, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, fb, (long long)nf, pre, csize, isize, post, k0, out, mismatch);
    return hipGetLastError();
}
hipError_t rt_ply_scan_bytes(int64_t nf, size_t *bytes) {
    *bytes = 0;
    return hipcub::DeviceScan::ExclusiveSum(nullptr, *bytes, (const int *)nullptr, (int *)nullptr, (int)(nf + 1), (hipStream_t) nullptr);
}
// ntri / tri_off: nf + 1 ints each; tri_off[nf] = number of triangles
hipError_t rt_ply_faces_mixed_count(hipStream_t s, const uint8_t *fb, const long long *offs, int64_t nf, int pre, int csize, int *ntri, int *tri_off, void *scan_tmp,
                                    size_t scan_bytes) {
    hipLaunchKernelGGL(k_ply_face_tris, dim3((unsigned)((nf + 1 + 255) / 256)), dim3(256), 0, s, fb, offs, (long long)nf, pre, csize, ntri);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, ntri, tri_off, (int)(nf + 1), s);
}
hipError_t rt_ply_faces_mixed_write(hipStream_t s, const uint8_t *fb, const long long *offs, const int *tri_off, int64_t nf, int pre, int csize, int isize, int *out) {
    if (nf <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ply_faces_mixed, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, fb, offs, tri_off, (long long)nf, pre, csize, isize, out);
    return hipGetLastError();
}

// Tokens the device could not convert exactly were converted on the host: out[3 * vertex + coordinate] = value for every listed
// triple (vertex, coordinate, byte offset) of `hard` (the list the vertex pass wrote, still in device memory).
__global__ void k_ply_patch(const int *hard, const float *value, int n, float *out) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < n) out[3 * (size_t)hard[1 + 3 * h] + hard[2 + 3 * h]] = value[h];
}
hipError_t rt_ply_patch(hipStream_t s, const int *hard, const float *value, int n, float *out) {
    if (n > 0) hipLaunchKernelGGL(k_ply_patch, dim3((n + 255) / 256), dim3(256), 0, s, hard, value, n, out);
    return hipGetLastError();
}
