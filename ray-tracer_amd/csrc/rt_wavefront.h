// rt_wavefront.h — per-bounce ray queues and the launch interface between rt_api.hip and rt_kernels.hip.
//
// The reference renders a pixel with one recursive C++ call (LightContributionCalculator::CalculateLight ->
// RecursiveComputation -> CalculateLight ...).  Here recursion depth k becomes "level k": a structure-of-arrays
// ray queue, traced and shaded by whole-queue kernels.  A forward sweep (levels 0..maxDepth) traces and shades
// and appends children to the next level; a backward sweep (maxDepth..0) folds child colours into their
// parents in exactly the reference's post-order association, so the result does not depend on queue order.
#ifndef RT_WAVEFRONT_H_
#define RT_WAVEFRONT_H_

#include <hip/hip_runtime.h>

#include "rt_device.h"

// One level of the wavefront.  All arrays have `capacity` entries; 16-byte elements so that lane i of a
// wave reads/writes base + 16*i (fully coalesced 1 KiB per wave instruction).
struct LevelBuf {
    // ray queue
    float4 *ray_o;      // origin.xyz, time
    float4 *ray_d;      // direction.xyz, refraction index of the medium the ray travels in (Ray::currMat)
    int4 *ray_m;        // pixel index, sample index, path code, container object of the medium (Ray::currShape) or -1
    // closest hit (written by trace, normal possibly re-written by shade_pre after texture tweaks)
    float4 *hit0;       // t, shape (int bits), object (int bits), prim record (int bits)
    float4 *hit1;       // ip.xyz, n.x
    float4 *hit2;       // n.y, n.z, uv.x, uv.y
    float4 *hit3;       // sphere theta, phi (SurfaceIntersection::lip), unused, unused
    // shading node (written by shade_post, consumed by the backward sweep)
    float4 *node_a;     // local light colour rgb, flags (int bits)
    float4 *node_b;     // mirror reflectance rgb, reflection coefficient
    float4 *node_c;     // absorption factor rgb, refraction coefficient
    int2 *child;        // index of the refraction / reflection child in the next level, or -1
    float4 *color;      // resolved colour rgb, hit flag
    int capacity;
    int count;
};

enum {
    NODE_HIT = 1,            // ray hit something (what CalculateLight returns)
    NODE_REFL_ALWAYS = 2,    // dielectric reflection: added even when the child misses (Q11)
    NODE_REFL_MRC = 4,       // reflection colour is multiplied by MRC before the coefficient (mirror, conductor)
    NODE_ABSORB = 8          // Beer absorption multiplies the accumulated colour (Q12)
};

// Shadow rays of one level.  Results and light vectors are indexed by SLOT = hit * n_lights + light; the rays themselves
// sit in a compact queue (only lit hits cast them) whose length is counted on the device.
struct ShadowBuf {
    float4 *o;          // queue: origin.xyz, time
    float4 *d;          // queue: direction.xyz, distance to the light
    float4 *ipref;      // queue: the shaded point ip.xyz, slot (int bits)
    int *count;         // queue length (device counter, zeroed per batch)
    float4 *ldir;       // slot: direction to the light, distance (read by shade_post)
    int *occluded;      // slot: 1 = an object lies between point and light (Q7)
    int capacity;
};

struct DevCamera {
    float pos[3], gaze[3], up[3], right[3];
    float left, top, distance, pixel_w, pixel_h;
    int nx, ny, spp;
    int jitter_n;           // (int)sqrt(spp)  (Q19)
    float jitter_w;         // 1.0f / jitter_n
    float aperture, focus;
    unsigned seed;
};

struct DevCounters {        // accumulated with atomics when counting is enabled
    unsigned long long node_visits, box_tests, tri_tests, sphere_tests;
    // lane-occupancy diagnostics of the persistent traversal (lane-summed; printed when RT_TRACE_DEBUG is set):
    // 0 outer iterations x64, 1 lanes holding a ray, 2 refill rounds x64, 3 lanes refilled, 4 node steps x64, 5 lanes stepping,
    // 6 leaf primitive iterations x64, 7 lanes testing a primitive, 8 exact-slab fallback rounds x64, 9 lanes needing it,
    // 10 children whose box begins beyond the best hit, 11 of those skipped (the normal cone allows it), 12 entered because the
    // cone word is "never", 13 entered because the ray lies inside the cone's band
    unsigned long long dbg[2][14];   // [closest | shadow]
    unsigned long long want_hist;     // set by the host when RT_TRACE_DEBUG asks for the per-ray histogram (three atomics per ray)
    unsigned long long hist[96];      // rays by floor(log2(node steps))+1, [closest 16 | shadow 16]; then ticks (100 MHz) and steps per bucket
};

void rt_launch_raygen(hipStream_t s, const DevCamera &cam, const int *pixel_list, int first_pixel, int n_samples, LevelBuf lv);
// fetch_counter: a zeroed int in device memory, private to the launch (the persistent waves draw ray indices from it).
// One launch traces lv's closest-hit rays and the shadow queue sb (length on the device, at most shadow_bound; 0 = none).
// Queue lengths: every launcher takes (n, n_dev).  n_dev == nullptr: the queue holds n entries.  Otherwise the real length is
// the device counter *n_dev (a level whose size the host never reads back) and n is its upper bound = the capacity the
// queue was allocated with; the grid is sized for n and the kernels process *n_dev entries — none when *n_dev > n: such a level was
// outgrown, has holes where blocks could not write their children, and belongs to a pass the host renders again.
void rt_tl_begin();              // tuning builds: per-launch wave timeline (RT_TIMELINE=1); no-ops otherwise
void rt_tl_end();
void rt_launch_trace(hipStream_t s, const DevScene &sc, LevelBuf lv, int n_closest, const int *closest_dev, ShadowBuf sb, int shadow_bound, int *fetch_counter,
                     int *overflow_stack, DevCounters *counters);
void rt_launch_trace_batch(hipStream_t s, const DevScene &sc, const rt_ray *rays, rt_hit *hits, int n, int *fetch_counter, int *overflow_stack, DevCounters *counters,
                           int max_blocks = 0, int ovf_stride = 0);
void rt_launch_fill_int(hipStream_t s, int *p, int n, int v);   // p[0..n) = v, as a kernel
int rt_trace_block_threads();    // threads per block of the trace kernels
int rt_trace_resident_blocks();  // blocks of the trace kernels resident on the device at a time (the grid cap)
// next.capacity bounds the children written: a block whose children do not fit writes none, *next_count still counts them
void rt_launch_shade_pre(hipStream_t s, const DevScene &sc, const DevCamera &cam, LevelBuf lv, int n, const int *n_dev, ShadowBuf sb, LevelBuf next, int depth,
                         int *next_count, int *refr_count);
void rt_launch_count_kinds(hipStream_t s, LevelBuf lv, int n, const int *n_dev, ShadowBuf sb, unsigned long long *ray_kind_counts);
void rt_launch_shade_post(hipStream_t s, const DevScene &sc, const DevCamera &cam, LevelBuf lv, int n, const int *n_dev, ShadowBuf sb);
void rt_launch_backward(hipStream_t s, LevelBuf lv, int n, const int *n_dev, LevelBuf next, int has_next);
void rt_launch_resolve(hipStream_t s, const DevCamera &cam, const int *pixel_list, int first_pixel, int n_pixels, LevelBuf lv0, int compact,
                       float *hdr, int *prim_id, float *prim_t);
void rt_set_small_rays_per_block(int v);   // experiment hook: rays per block that size the grid of a launch smaller than the resident grid
void rt_set_trace_block_cap(int cap);   // experiment hook: upper bound of the persistent grid (0 = default)
void rt_set_num_cus(int n);      // compute units of the device (sizes the persistent grid)
int rt_trace_grid_threads();     // most persistent threads a trace launch has (stride of the stack-overflow columns)
int rt_trace_lds_stack_depth();  // stack entries kept in LDS per lane

// rt_image.hip: frame post-processing (8-bit resolve, P3 text, tone mapping); all pointers are device pointers.
hipError_t rt_img_resolve8(hipStream_t s, const float *hdr, int64_t n_values, uint8_t *out);
int64_t rt_img_p3_blocks(int64_t npx);
hipError_t rt_img_p3_scan_bytes(int64_t npx, size_t *bytes);
hipError_t rt_img_p3_measure(hipStream_t s, const uint8_t *rgb, int64_t npx, int nx, unsigned long long *block_len, unsigned long long *block_off,
                             void *scan_tmp, size_t scan_bytes);
hipError_t rt_img_p3_write(hipStream_t s, const uint8_t *rgb, int64_t npx, int nx, const unsigned long long *block_off, uint8_t *out_body);
hipError_t rt_img_exr_rows(hipStream_t s, const float *rgb, int nx, int ny, uint8_t *body);   // body: ny * (8 + 6 nx) bytes, 2-byte aligned

// ---- PLY body decoding (rt_ply.hip) ----
hipError_t rt_ply_vertices(hipStream_t s, const uint8_t *vb, int64_t nv, int stride, const int off[3], int type, float *out);
hipError_t rt_ply_faces_uniform(hipStream_t s, const uint8_t *fb, int64_t nf, int pre, int csize, int isize, int post, unsigned k0, int *out, int *mismatch);
hipError_t rt_ply_scan_bytes(int64_t nf, size_t *bytes);
hipError_t rt_ply_faces_mixed_count(hipStream_t s, const uint8_t *fb, const long long *offs, int64_t nf, int pre, int csize, int *ntri, int *tri_off, void *scan_tmp,
                                    size_t scan_bytes);
hipError_t rt_ply_faces_mixed_write(hipStream_t s, const uint8_t *fb, const long long *offs, const int *tri_off, int64_t nf, int pre, int csize, int isize, int *out);
// ascii bodies: line table, then per-line token scanners
hipError_t rt_ply_ascii_scan_bytes(int64_t n_text, size_t *bytes);
hipError_t rt_ply_ascii_lines(hipStream_t s, const uint8_t *text, int64_t n_text, int *flag, int *scan, void *scan_tmp, size_t scan_bytes, int64_t n_lines, long long *line_start);
hipError_t rt_ply_ascii_vertices(hipStream_t s, const uint8_t *text, const long long *line_start, int64_t n_text, int64_t nv, const int tok[3], int is_double, float *out,
                                 int *hard, int hard_cap, int *bad);
hipError_t rt_ply_ascii_faces_count(hipStream_t s, const uint8_t *text, const long long *line_start, int64_t n_text, int64_t first_line, int64_t nf, int pre, int *ntri,
                                    int *tri_off, void *scan_tmp, size_t scan_bytes, int *bad);
hipError_t rt_ply_ascii_faces_write(hipStream_t s, const uint8_t *text, const long long *line_start, int64_t n_text, int64_t first_line, int64_t nf, int pre,
                                    const int *tri_off, int *out, int *bad);
hipError_t rt_ply_patch(hipStream_t s, const int *hard, const float *value, int n, float *out);   // out[3 * vertex + coordinate] = value, per listed token
hipError_t rt_ply_offset_copy(hipStream_t s, const int *src, int *dst, int64_t n, int offset);
#define RT_TM_WORK_WORDS (4 + 2048)   // select state + histogram of rt_img_tonemap
hipError_t rt_img_tonemap(hipStream_t s, const void *src, int from8, int64_t npx, float key, float burn_percent, float saturation,
                          float *lum, double *partial, unsigned *partial_max, unsigned *work, float *params, float *out);

#endif
