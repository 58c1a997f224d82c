/* rt_mi355_mgpu.h — the multi-GPU step of the render path as a C-ABI (librt_mi355_mgpu.so; links librt_mi355.so and RCCL).
 *
 * What it replaces: the reference's only parallel construct, the eight row-band threads of one frame and their join
 * (reference src/DefaultRenderer.cpp:65-75: std::thread per band, join, then Image::SaveImage).  Here a frame is cut into
 * interleaved square tiles (rt_render_opts.tile_rank / tile_world of rt_mi355.h), every GPU renders its tiles into a compact
 * RGB32F buffer in HBM, and this library brings the tiles together on rank 0 over xGMI:
 *
 *   rt_scene_broadcast   once per scene: the twelve arenas of the scene rank 0 has built (rt_scene_export) travel to every other
 *                        rank with ncclBroadcast and become that rank's scene (rt_scene_import) — the replicated BVH.
 *   rt_gather_tiles      once per frame: grouped ncclSend / ncclRecv of every rank's compact tile buffer to rank 0 (point to
 *                        point: one xGMI link per peer, no ring), then one kernel on rank 0 that puts every pixel at its place
 *                        in the frame (the order is rt_tile_pixel_order's).
 *
 * Two ways to use it:
 *   one process, N GPUs   rt_gather_create(device ids, N)            (ncclCommInitAll; what host/rt_render --ranks N does)
 *   one process per GPU   rt_gather_unique_id on rank 0, hand the 128 bytes to every rank by any means, then
 *                         rt_gather_create_rank(id, rank, world, device) on every rank.
 * Plain pointers and sizes, negative status codes of rt_mi355.h, nothing throws.  No CPU fallback: RT_ERR_NO_DEVICE without a GPU
 * (rt_gather_deinterleave_host is the only call that needs none).
 */
#ifndef RT_MI355_MGPU_H_
#define RT_MI355_MGPU_H_

#include "rt_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_gather rt_gather;

#define RT_GATHER_ID_BYTES 128   /* sizeof(ncclUniqueId) */

/* One process that drives n_ranks GPUs: rank r lives on hip_device_ids[r] (distinct devices), rank 0 receives.
 * Replaces the creation of the worker threads of DefaultRenderer::RenderCamera (src/DefaultRenderer.cpp:65-71). */
int rt_gather_create(const int *hip_device_ids, int n_ranks, rt_gather **out);

/* One process per GPU: rank 0 calls rt_gather_unique_id and distributes the RT_GATHER_ID_BYTES bytes; then every rank calls
 * rt_gather_create_rank (a collective call: it returns when all `world` ranks have joined). */
int rt_gather_unique_id(void *id_bytes);
int rt_gather_create_rank(const void *id_bytes, int rank, int world, int hip_device_id, rt_gather **out);

/* The frame of one camera from the ranks' tiles — the join of the row-band threads (src/DefaultRenderer.cpp:73-75).
 *   d_tiles[i]  compact tile buffer of the i-th LOCAL rank (device memory of that rank's GPU): what rt_render wrote with
 *               tile_rank = that rank, tile_world = world, compact_output = 1, output_on_device = 1; 3 floats per owned pixel.
 *               One process with N GPUs passes N pointers (rank order), one process per GPU passes one.
 *   d_frame     on rank 0 only (NULL elsewhere): ny * nx * 3 floats in rank 0's device memory, row-major.
 * The tile buffers must be complete when the call is made (rt_render has returned: it waits for its stream).
 * Blocking: returns when rank 0's frame is complete (other ranks: when their send has been handed over). */
int rt_gather_tiles(rt_gather *g, int nx, int ny, int tile_size, const float *const *d_tiles, float *d_frame);

/* Buffers for hosts that have no device allocator of their own (a plain C++ host without the HIP runtime): for a frame of nx * ny
 * pixels cut into tile_size tiles, d_tiles[i] = a compact tile buffer on the i-th local rank's GPU (the hdr_rgb argument of that
 * rank's rt_render) and, on the process that drives rank 0, *d_frame = the frame rt_gather_tiles assembles (d_frame may be NULL
 * elsewhere).  Owned by g: valid until a call with a larger geometry or rt_gather_destroy.  rt_gather_copy_frame copies the frame of
 * the last rt_gather_buffers geometry to host memory (ny * nx * 3 floats), the input of rt_image_encode / rt_tonemap — the frame the
 * reference hands to Image::SaveImage (src/DefaultRenderer.cpp:78). */
int rt_gather_buffers(rt_gather *g, int nx, int ny, int tile_size, float **d_tiles, float **d_frame);
int rt_gather_copy_frame(rt_gather *g, float *frame_rgb);

/* The scene rank 0's context holds becomes the scene of every other rank's context (rt_scene_export -> ncclBroadcast of the
 * RT_SCENE_ARENAS arenas -> rt_scene_import).  ctxs[i] = context of the i-th local rank, created on that rank's device.
 * Replaces nothing in the reference (its threads share one Scene object); it is what "replicate the BVH on every GPU" needs. */
int rt_scene_broadcast(rt_gather *g, rt_ctx *const *ctxs);

int rt_gather_world(const rt_gather *g);          /* number of ranks */
void rt_gather_destroy(rt_gather *g);
const char *rt_gather_last_error(const rt_gather *g);   /* g == NULL: the error of the last failed create call of this thread */

/* Host-only view of the de-interleave (no GPU, no RCCL): frame[pixel] = tiles[r][i] for the i-th pixel of
 * rt_tile_pixel_order(nx, ny, tile_size, r, world); tiles[r] holds 3 floats per owned pixel.  For tests and for hosts that
 * gathered the tiles by other means. */
int rt_gather_deinterleave_host(int nx, int ny, int tile_size, int world, const float *const *tiles, float *frame_rgb);

#ifdef __cplusplus
}
#endif
#endif
