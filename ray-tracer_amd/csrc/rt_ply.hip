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

}  // namespace

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
