// rt_image.hip — what the reference does with a finished frame, on the device:
//   * ObtainColorFromUnclampedVector (reference src/DefaultRenderer.cpp:18-27): float radiance -> 8-bit Color
//   * Image::SaveImageAsPPM (src/Image.cpp:54-77): the P3 text, byte for byte what the fprintf loop produces
//   * DefaultTonemapStrategy (src/DefaultTonemapStrategy.cpp:15-114): photographic tone mapping of the frame
// All of it is streaming, HBM-bound byte/float work: one pass per step, coalesced reads, text staged through LDS so
// that the variable-length output leaves the block as aligned dwords.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>

#include "rt_wavefront.h"

namespace {

constexpr int IMG_BLOCK = 256;
constexpr int P3_MAX_PIXEL = 13;            // "255 255 255 " + '\n'

// ">255 -> 255", then the float -> unsigned char conversion (truncation).  Negative and NaN inputs are undefined
// behaviour in the reference; 0 here (the same choice as the host helper rt_resolve_8bit).
__device__ __forceinline__ unsigned resolve8(float v) { return v > 255.0f ? 255u : (v > 0.0f ? (unsigned)v : 0u); }

__global__ __launch_bounds__(IMG_BLOCK) void k_resolve8_vec(const float4 *hdr, int64_t n4, uint32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x;
    if (i >= n4) return;
    const float4 v = hdr[i];
    out[i] = resolve8(v.x) | (resolve8(v.y) << 8) | (resolve8(v.z) << 16) | (resolve8(v.w) << 24);
}

__global__ __launch_bounds__(IMG_BLOCK) void k_resolve8(const float *hdr, int64_t first, int64_t n, uint8_t *out) {
    const int64_t i = first + (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x;
    if (i < n) out[i] = (uint8_t)resolve8(hdr[i]);
}

// ---- P3 text ------------------------------------------------------------------------------------------
// fprintf("%d ", channel) for the three channels of every pixel, fprintf("\n") after every row.
__device__ __forceinline__ int p3_len(unsigned v) { return 2 + (v >= 10u) + (v >= 100u); }

__device__ __forceinline__ uint8_t *p3_put(uint8_t *p, unsigned v) {
    const unsigned h = v / 100u, t = (v / 10u) % 10u, o = v % 10u;
    if (v >= 100u) *p++ = (uint8_t)('0' + h);
    if (v >= 10u) *p++ = (uint8_t)('0' + t);
    *p++ = (uint8_t)('0' + o);
    *p++ = ' ';
    return p;
}

__device__ __forceinline__ int p3_pixel(const uint8_t *rgb, unsigned i, unsigned npx, unsigned nx, unsigned v[3], bool &eol) {
    eol = false;
    if (i >= npx) return 0;
    v[0] = rgb[3ull * i]; v[1] = rgb[3ull * i + 1]; v[2] = rgb[3ull * i + 2];
    eol = (i % nx) == nx - 1;
    return p3_len(v[0]) + p3_len(v[1]) + p3_len(v[2]) + (eol ? 1 : 0);
}

__global__ __launch_bounds__(IMG_BLOCK) void k_p3_len(const uint8_t *rgb, unsigned npx, unsigned nx, unsigned long long *block_len) {
    using Reduce = hipcub::BlockReduce<int, IMG_BLOCK>;
    __shared__ typename Reduce::TempStorage tmp;
    unsigned v[3]; bool eol;
    const int len = p3_pixel(rgb, blockIdx.x * IMG_BLOCK + threadIdx.x, npx, nx, v, eol);
    const int total = Reduce(tmp).Sum(len);
    if (threadIdx.x == 0) block_len[blockIdx.x] = (unsigned long long)total;
}

__global__ __launch_bounds__(IMG_BLOCK) void k_p3_write(const uint8_t *rgb, unsigned npx, unsigned nx, const unsigned long long *block_off, uint8_t *out) {
    using Scan = hipcub::BlockScan<int, IMG_BLOCK>;
    __shared__ typename Scan::TempStorage tmp;
    __shared__ uint32_t s_words[(IMG_BLOCK * P3_MAX_PIXEL + 8) / 4 + 1];
    uint8_t *s = reinterpret_cast<uint8_t *>(s_words);
    unsigned v[3] = {0, 0, 0}; bool eol;
    const int len = p3_pixel(rgb, blockIdx.x * IMG_BLOCK + threadIdx.x, npx, nx, v, eol);
    int off, total;
    Scan(tmp).ExclusiveSum(len, off, total);
    uint8_t *dst = out + block_off[blockIdx.x];
    const int a = (int)(reinterpret_cast<uintptr_t>(dst) & 3u);       // stage at the same offset inside a dword as in memory
    if (len) {
        uint8_t *p = s + a + off;
        p = p3_put(p, v[0]); p = p3_put(p, v[1]); p = p3_put(p, v[2]);
        if (eol) *p = '\n';
    }
    __syncthreads();
    uint32_t *gw = reinterpret_cast<uint32_t *>(dst - a);
    const int end = a + total, nwords = (end + 3) >> 2;
    for (int w = threadIdx.x; w < nwords; w += IMG_BLOCK) {
        const int b0 = 4 * w, b1 = b0 + 4;
        if (b0 >= a && b1 <= end) gw[w] = s_words[w];
        else                                                          // first / last partial dword: neighbours own the other bytes
            for (int b = (b0 > a ? b0 : a); b < (b1 < end ? b1 : end); ++b) reinterpret_cast<uint8_t *>(gw)[b] = s[b];
    }
}

// ---- EXR (half, uncompressed scan lines) ---------------------------------------------------------------------
// tinyexr's float_to_half_full (reference src/tinyexr.h:7239-7274), bit for bit: truncation of the mantissa plus "round up
// when the first dropped bit is set" (ties go away from zero, unlike the hardware's round-to-nearest-even conversion).
__device__ __forceinline__ unsigned half_bits_tinyexr(float v) {
    const unsigned f = __float_as_uint(v);
    const unsigned sign = (f >> 16) & 0x8000u, exponent = (f >> 23) & 0xffu, mantissa = f & 0x7fffffu;
    unsigned o = 0;
    if (exponent == 0) o = 0;                                          // signed zero / denormal
    else if (exponent == 255) o = (31u << 10) | (mantissa ? 0x200u : 0u);
    else {
        const int newexp = (int)exponent - 127 + 15;
        if (newexp >= 31) o = 31u << 10;                               // overflow: infinity
        else if (newexp <= 0) {
            if ((14 - newexp) <= 24) {
                const unsigned mant = mantissa | 0x800000u;
                o = mant >> (14 - newexp);
                if ((mant >> (13 - newexp)) & 1u) o++;
            }
        } else {
            o = ((unsigned)newexp << 10) | (mantissa >> 13);
            if (mantissa & 0x1000u) o++;
        }
    }
    return (o & 0x7fffu) | sign;
}

// One scan-line block per image row (SaveEXRImageToMemory, tinyexr.h:12318-12449, compression NONE): int y, int data size,
// then the row's B, G and R values as halfs, channel after channel (Tonemapper::SaveEXR names the channels B, G, R and
// hands them over in that order, Tonemapper.cpp:57-74).  Rows are 8 + 6*nx bytes: every half is 2-byte aligned in `body`.
__global__ __launch_bounds__(IMG_BLOCK) void k_exr_rows(const float *rgb, int nx, int ny, uint16_t *body) {
    const int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x;
    if (i >= (int64_t)nx * ny) return;
    const int y = (int)(i / nx), x = (int)(i - (int64_t)y * nx);
    uint16_t *row = body + (size_t)y * (4 + 3 * (size_t)nx);
    if (x == 0) {
        const unsigned len = 6u * (unsigned)nx;
        row[0] = (uint16_t)((unsigned)y & 0xffffu); row[1] = (uint16_t)((unsigned)y >> 16);
        row[2] = (uint16_t)(len & 0xffffu); row[3] = (uint16_t)(len >> 16);
    }
    const float *p = rgb + 3 * i;
    row[4 + x] = (uint16_t)half_bits_tinyexr(p[2]);
    row[4 + nx + x] = (uint16_t)half_bits_tinyexr(p[1]);
    row[4 + 2 * (size_t)nx + x] = (uint16_t)half_bits_tinyexr(p[0]);
}

// ---- tone mapping ---------------------------------------------------------------------------------------
// ComputeLuminance (DefaultTonemapStrategy.h:43-46); channels are 8-bit Colors converted to float in the reference.
__device__ __forceinline__ float tm_lum(float r, float g, float b) { return 0.27f * r + 0.67f * g + 0.06f * b; }

template <bool FROM8>
__device__ __forceinline__ void tm_load(const void *src, int64_t i, float &r, float &g, float &b) {
    if (FROM8) { const uint8_t *p = static_cast<const uint8_t *>(src) + 3 * i; r = (float)p[0]; g = (float)p[1]; b = (float)p[2]; }
    else { const float *p = static_cast<const float *>(src) + 3 * i; r = p[0]; g = p[1]; b = p[2]; }
}

// Order-preserving map float -> unsigned (the order a radix sort of floats uses: -NaN < -inf < ... < -0 < +0 < ... < +inf < NaN).
__device__ __forceinline__ unsigned tm_key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float tm_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// RecordPixelLuminance (:64-71): the luminance of every pixel and log(luminance + 0.00001f) in double (the unqualified
// `log` there is ::log(double)); partial sums and the largest luminance per block, in a fixed order.
template <bool FROM8>
__global__ __launch_bounds__(IMG_BLOCK) void k_tm_lum(const void *src, int64_t npx, float *lum, double *partial, unsigned *partial_max) {
    using Reduce = hipcub::BlockReduce<double, IMG_BLOCK>;
    using ReduceU = hipcub::BlockReduce<unsigned, IMG_BLOCK>;
    __shared__ typename Reduce::TempStorage tmp;
    __shared__ typename ReduceU::TempStorage tmpu;
    const int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x;
    double lg = 0.0;
    unsigned k = 0;
    if (i < npx) {
        float r, g, b;
        tm_load<FROM8>(src, i, r, g, b);
        const float l = tm_lum(r, g, b);
        lum[i] = l;
        lg = log((double)(l + 0.00001f));
        k = tm_key(l);
    }
    const double sum = Reduce(tmp).Sum(lg);
    const unsigned mx = ReduceU(tmpu).Reduce(k, hipcub::Max());
    if (threadIdx.x == 0) { partial[blockIdx.x] = sum; partial_max[blockIdx.x] = mx; }
}

// HeapifyLuminanceVector (:73-77) sorts all luminances, ComputeLuminanceWhite (:104-107) then reads ONE of them and the
// largest.  Here: a three-pass radix SELECT of that one element (11 + 11 + 10 bits of the order-preserving key; per pass a
// histogram of the candidates that still match the prefix, then the bin holding the wanted rank), 4 bytes per pixel and
// pass instead of the ~64 bytes per pixel of a full radix sort.  sel: [0] key prefix found so far, [1] rank inside it.
constexpr int TM_BINS = 2048;
template <int PASS>
__global__ __launch_bounds__(IMG_BLOCK) void k_tm_hist(const float *lum, int64_t npx, const unsigned *sel, unsigned *hist) {
    __shared__ unsigned h[TM_BINS];
    for (int b = threadIdx.x; b < TM_BINS; b += IMG_BLOCK) h[b] = 0;
    __syncthreads();
    const unsigned prefix = sel[0];
    for (int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x; i < npx; i += (int64_t)gridDim.x * IMG_BLOCK) {
        const unsigned k = tm_key(lum[i]);
        if (PASS == 0) atomicAdd(&h[k >> 21], 1u);
        else if (PASS == 1) { if ((k >> 21) == (prefix >> 21)) atomicAdd(&h[(k >> 10) & 2047u], 1u); }
        else { if ((k >> 10) == (prefix >> 10)) atomicAdd(&h[k & 1023u], 1u); }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < TM_BINS; b += IMG_BLOCK) if (h[b]) atomicAdd(&hist[b], h[b]);
}

__global__ void k_tm_init(unsigned *work, unsigned rank) {          // select state {prefix 0, rank} and a cleared histogram
    for (int i = threadIdx.x; i < RT_TM_WORK_WORDS; i += blockDim.x) work[i] = i == 1 ? rank : 0u;
}

// The bin that holds rank sel[1]; clears the histogram for the next pass.
template <int PASS>
__global__ __launch_bounds__(IMG_BLOCK) void k_tm_pick(unsigned *hist, unsigned *sel) {
    using Scan = hipcub::BlockScan<unsigned, IMG_BLOCK>;
    __shared__ typename Scan::TempStorage tmp;
    constexpr int PER = TM_BINS / IMG_BLOCK;
    unsigned v[PER], sum = 0;
    for (int j = 0; j < PER; ++j) { v[j] = hist[threadIdx.x * PER + j]; sum += v[j]; hist[threadIdx.x * PER + j] = 0; }
    unsigned before;
    Scan(tmp).ExclusiveSum(sum, before);
    const unsigned rank = sel[1];
    __syncthreads();                                                  // every thread has read the rank before one rewrites it
    if (rank >= before && rank < before + sum) {
        unsigned acc = before;
        for (int j = 0; j < PER; ++j) {
            if (rank < acc + v[j]) {
                const unsigned bin = threadIdx.x * PER + j;
                sel[0] = PASS == 0 ? bin << 21 : (PASS == 1 ? (sel[0] | (bin << 10)) : (sel[0] | bin));
                sel[1] = rank - acc;
                break;
            }
            acc += v[j];
        }
    }
}

// SetupGeneralAttributes (:37-46) + ComputeLuminanceWhite (:104-107).  params: [0] average luminance, [1] luminance white.
__global__ __launch_bounds__(IMG_BLOCK) void k_tm_params(const double *partial, const unsigned *partial_max, int n_partial, const unsigned *sel, int64_t npx, float *params) {
    using Reduce = hipcub::BlockReduce<double, IMG_BLOCK>;
    using ReduceU = hipcub::BlockReduce<unsigned, IMG_BLOCK>;
    __shared__ typename Reduce::TempStorage tmp;
    __shared__ typename ReduceU::TempStorage tmpu;
    double acc = 0.0;
    unsigned mx = 0;
    for (int i = threadIdx.x; i < n_partial; i += IMG_BLOCK) { acc += partial[i]; mx = partial_max[i] > mx ? partial_max[i] : mx; }
    const double sum = Reduce(tmp).Sum(acc);
    mx = ReduceU(tmpu).Reduce(mx, hipcub::Max());
    if (threadIdx.x != 0) return;
    float s = (float)sum;
    s = s / (float)npx;                                               // luminanceSum /= GetImageSize()
    params[0] = (float)exp((double)s);
    params[1] = tm_unkey(sel[0]) / tm_unkey(mx);                      // mLuminances[rank] / mLuminances.back()
}

// ComputeTonemappedPixelColor (:79-102).
template <bool FROM8>
__global__ __launch_bounds__(IMG_BLOCK) void k_tm_apply(const void *src, int64_t npx, const float *params, float key, float saturation, float *out) {
    const int64_t i = (int64_t)blockIdx.x * IMG_BLOCK + threadIdx.x;
    if (i >= npx) return;
    float r, g, b;
    tm_load<FROM8>(src, i, r, g, b);
    const float avg = params[0], lw = params[1];
    const float l = tm_lum(r, g, b);
    const float reg = l * key / avg;
    const float res = (reg * (1 + (reg / (lw * lw)))) / (1 + reg);
    float *o = out + 3 * i;
    o[0] = powf(r / l, saturation) * res;
    o[1] = powf(g / l, saturation) * res;
    o[2] = powf(b / l, saturation) * res;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + IMG_BLOCK - 1) / IMG_BLOCK); }

}  // namespace

hipError_t rt_img_resolve8(hipStream_t s, const float *hdr, int64_t n_values, uint8_t *out) {
    if (n_values <= 0) return hipSuccess;
    int64_t done = 0;
    if ((reinterpret_cast<uintptr_t>(hdr) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 3u) == 0 && n_values >= 4) {
        const int64_t n4 = n_values / 4;
        hipLaunchKernelGGL(k_resolve8_vec, dim3(blocks_for(n4)), dim3(IMG_BLOCK), 0, s, reinterpret_cast<const float4 *>(hdr), n4, reinterpret_cast<uint32_t *>(out));
        done = 4 * n4;
    }
    if (done < n_values) hipLaunchKernelGGL(k_resolve8, dim3(blocks_for(n_values - done)), dim3(IMG_BLOCK), 0, s, hdr, done, n_values, out);
    return hipGetLastError();
}

int64_t rt_img_p3_blocks(int64_t npx) { return (npx + IMG_BLOCK - 1) / IMG_BLOCK; }

hipError_t rt_img_p3_scan_bytes(int64_t npx, size_t *bytes) {
    *bytes = 0;
    return hipcub::DeviceScan::ExclusiveSum(nullptr, *bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int)(rt_img_p3_blocks(npx) + 1));
}

// block_len / block_off hold blocks+1 entries; block_off[blocks] is the length of the whole body.
hipError_t rt_img_p3_measure(hipStream_t s, const uint8_t *rgb, int64_t npx, int nx, unsigned long long *block_len, unsigned long long *block_off,
                             void *scan_tmp, size_t scan_bytes) {
    const int64_t nb = rt_img_p3_blocks(npx);
    hipError_t e = hipMemsetAsync(block_len + nb, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_p3_len, dim3((unsigned)nb), dim3(IMG_BLOCK), 0, s, rgb, (unsigned)npx, (unsigned)nx, block_len);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, block_len, block_off, (int)(nb + 1), s);
}

hipError_t rt_img_p3_write(hipStream_t s, const uint8_t *rgb, int64_t npx, int nx, const unsigned long long *block_off, uint8_t *out_body) {
    hipLaunchKernelGGL(k_p3_write, dim3((unsigned)rt_img_p3_blocks(npx)), dim3(IMG_BLOCK), 0, s, rgb, (unsigned)npx, (unsigned)nx, block_off, out_body);
    return hipGetLastError();
}

hipError_t rt_img_exr_rows(hipStream_t s, const float *rgb, int nx, int ny, uint8_t *body) {
    hipLaunchKernelGGL(k_exr_rows, dim3(blocks_for((int64_t)nx * ny)), dim3(IMG_BLOCK), 0, s, rgb, nx, ny, reinterpret_cast<uint16_t *>(body));
    return hipGetLastError();
}

// The rank ComputeLuminanceWhite reads (:106): round(size * (100.f - percentage) / 100.f) in the reference's float
// arithmetic; outside [0, n) the reference indexes out of bounds, clamped here.
static int64_t tm_rank(int64_t npx, float burn_percent) {
    const float pos = (float)npx * (100.f - burn_percent) / 100.f;
    double ri = round((double)pos);
    if (!(ri >= 0.0)) ri = 0.0;
    int64_t idx = (int64_t)ri;
    return idx > npx - 1 ? npx - 1 : idx;
}

// work: TM_WORK_WORDS unsigned words (select state + histogram); partial / partial_max: one entry per 256 pixels.
hipError_t rt_img_tonemap(hipStream_t s, const void *src, int from8, int64_t npx, float key, float burn_percent, float saturation,
                          float *lum, double *partial, unsigned *partial_max, unsigned *work, float *params, float *out) {
    const unsigned nb = blocks_for(npx), nh = nb < 2048u ? nb : 2048u;
    unsigned *sel = work, *hist = work + 4;
    hipLaunchKernelGGL(k_tm_init, dim3(1), dim3(IMG_BLOCK), 0, s, work, (unsigned)tm_rank(npx, burn_percent));
    if (from8) hipLaunchKernelGGL(k_tm_lum<true>, dim3(nb), dim3(IMG_BLOCK), 0, s, src, npx, lum, partial, partial_max);
    else hipLaunchKernelGGL(k_tm_lum<false>, dim3(nb), dim3(IMG_BLOCK), 0, s, src, npx, lum, partial, partial_max);
    hipLaunchKernelGGL(k_tm_hist<0>, dim3(nh), dim3(IMG_BLOCK), 0, s, lum, npx, sel, hist);
    hipLaunchKernelGGL(k_tm_pick<0>, dim3(1), dim3(IMG_BLOCK), 0, s, hist, sel);
    hipLaunchKernelGGL(k_tm_hist<1>, dim3(nh), dim3(IMG_BLOCK), 0, s, lum, npx, sel, hist);
    hipLaunchKernelGGL(k_tm_pick<1>, dim3(1), dim3(IMG_BLOCK), 0, s, hist, sel);
    hipLaunchKernelGGL(k_tm_hist<2>, dim3(nh), dim3(IMG_BLOCK), 0, s, lum, npx, sel, hist);
    hipLaunchKernelGGL(k_tm_pick<2>, dim3(1), dim3(IMG_BLOCK), 0, s, hist, sel);
    hipLaunchKernelGGL(k_tm_params, dim3(1), dim3(IMG_BLOCK), 0, s, partial, partial_max, (int)nb, sel, npx, params);
    if (from8) hipLaunchKernelGGL(k_tm_apply<true>, dim3(nb), dim3(IMG_BLOCK), 0, s, src, npx, params, key, saturation, out);
    else hipLaunchKernelGGL(k_tm_apply<false>, dim3(nb), dim3(IMG_BLOCK), 0, s, src, npx, params, key, saturation, out);
    return hipGetLastError();
}
