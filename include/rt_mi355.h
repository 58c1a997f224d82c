/*
 * rt_mi355.h — C-ABI of the MI355X-native render hot path (librt_mi355.so).
 *
 * This is the drop-in boundary for the ONE accelerated path of fuchsteufelswild/Ray-Tracer:
 *   boundary #1  RenderStrategy::RenderSceneIntoPPM(Scene*)            (reference src/RenderStrategy.h:16,
 *                implemented by DefaultRenderer, src/DefaultRenderer.cpp:35-162)
 *   boundary #2  AccelerationStructure::Intersect(Ray&, SurfaceIntersection&, float)
 *                                                                       (reference src/AccelerationStructure.h:21,
 *                implemented by BVHTree, src/BVHTree.cpp:155-206)
 *
 * Conventions: plain C, fixed-width PODs, caller owns every host buffer it passes in, the library
 * owns device memory.  Every entry point returns 0 on success or a negative rt_status; nothing throws
 * across the ABI.  One context per host thread (not thread-safe).  All matrices are 16 floats in glm
 * memory order (column-major: m[col*4 + row]) because that is what the reference's Transform stores
 * (src/acmath.h:983-1070).
 *
 * The same input structs are consumed by the CPU oracle (oracle/rt_oracle.cpp), which is test
 * infrastructure only and shares nothing else with this library.
 */
#ifndef RT_MI355_H_
#define RT_MI355_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 6u
#define RT_MAX_BATCH_RAYS 0x7ff00000u   /* most rays one rt_intersect_batch / rt_intersect_batch_noalloc call takes (the queue positions of the
                                           persistent waves stay below 2^31 with every wave one chunk past the end) */

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument / inconsistent scene description          */
    RT_ERR_ABI = -2,         /* abi_version mismatch                                    */
    RT_ERR_NO_DEVICE = -3,   /* no usable HIP device                                    */
    RT_ERR_HIP = -4,         /* a HIP runtime call failed (see rt_last_error)           */
    RT_ERR_NO_SCENE = -5,    /* render / intersect called before rt_scene_upload        */
    RT_ERR_TOO_LARGE = -6    /* request exceeds a documented limit                      */
} rt_status;

/* ---- materials: reference src/Material.h:11-91, src/brdf.h:9-74, parser src/SceneParser.cpp:280-438 ---- */
enum { RT_MAT_DEFAULT = 0, RT_MAT_MIRROR = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_CONDUCTOR = 3 };
enum { RT_BRDF_BLINNPHONG_ORIGINAL = 0, RT_BRDF_BLINNPHONG_MODIFIED = 1, RT_BRDF_PHONG_ORIGINAL = 2,
       RT_BRDF_PHONG_MODIFIED = 3, RT_BRDF_TORRANCE_SPARROW = 4 };

typedef struct rt_material {
    int32_t type;             /* RT_MAT_*                                                   */
    int32_t brdf_kind;        /* RT_BRDF_*                                                  */
    int32_t brdf_flag;        /* "normalized" (modified BRDFs) or "kdfresnel" (Torrance)    */
    int32_t degamma;          /* Material::degamma (only scales the mirror weight, Q11)     */
    float brdf_exponent;      /* BRDFBase::specularPhong                                    */
    float refraction_index;   /* Material::rIndex                                           */
    float absorption_index;   /* Conductor::aIndex == BRDFBase::ai                          */
    float roughness;          /* Material::roughness (glossy reflection, Q15)               */
    float ka[3];              /* ARC */
    float kd[3];              /* DRC */
    float ks[3];              /* SRC */
    float km[3];              /* MRC */
    float ac[3];              /* Dielectric::AC */
    float _pad;
} rt_material;                /* 96 bytes */

/* ---- lights: reference src/Light.h:14-99, src/Light.cpp:37-129 ---- */
enum { RT_LIGHT_POINT = 0, RT_LIGHT_DIRECTIONAL = 1, RT_LIGHT_SPOT = 2, RT_LIGHT_AREA = 3 };

typedef struct rt_light {
    int32_t type;             /* RT_LIGHT_*                                   */
    float position[3];
    float intensity[3];       /* intensity or radiance                        */
    float direction[3];       /* direction (directional, spot) / normal (area) */
    float coverage_deg;       /* spot */
    float falloff_deg;        /* spot */
    float size;               /* area */
    float _pad[3];
} rt_light;                   /* 64 bytes */

/* ---- textures: reference src/Texture.h, src/PerlinTextureImpl.cpp, src/ImageTextureImpl.cpp, parser src/SceneParser.cpp:440-595 ---- */
enum { RT_TEX_IMAGE = 0, RT_TEX_PERLIN = 1 };
enum { RT_DECAL_REPLACE_KD = 0, RT_DECAL_BLEND_KD = 1, RT_DECAL_REPLACE_ALL = 2, RT_DECAL_REPLACE_NORMAL = 3,
       RT_DECAL_BUMP_NORMAL = 4, RT_DECAL_REPLACE_BACKGROUND = 5 };
enum { RT_NOISE_LINEAR = 0, RT_NOISE_ABSVAL = 1 };

typedef struct rt_texture {
    int32_t kind;             /* RT_TEX_*                                                        */
    int32_t decal;            /* RT_DECAL_*                                                      */
    float bump_factor;
    float noise_scale;        /* perlin                                                          */
    int32_t noise_conversion; /* RT_NOISE_*                                                      */
    int32_t normalizer;       /* image: ImageTextureImpl::mNormalizer                            */
    int32_t width, height;    /* image                                                           */
    int32_t bilinear;         /* image: always 0 when produced from the reference's parser (Q23) */
    int32_t texel_is_float;   /* 0: 8-bit RGB (stb_image path), 1: float RGB (EXR path)          */
    int64_t texel_offset;     /* byte offset of this image inside rt_scene_desc.texels           */
    int32_t perm[16];         /* perlin: shuffled index table (PerlinTextureImpl::mIndices)      */
    int32_t _pad[2];
} rt_texture;                 /* 120 bytes */

/* ---- objects: one per reference Shape that owns a Transform (Sphere, standalone Triangle, Mesh,
 *      MeshInstance).  It is also the "container shape" identity used by Q26. ---- */
enum { RT_OBJ_SMOOTH = 1,        /* Shape::ShadingMode::SMOOTH (mesh triangles)                              */
       RT_OBJ_HAS_XFORM = 2,     /* objTransform != nullptr (standalone <Triangle> passes nullptr)           */
       RT_OBJ_COMPOSITE = 4 };   /* triangles owned by a Mesh (IsOwnedByComposite, src/Shape.h:116)          */

typedef struct rt_object {
    float xform[16];          /* Transform::transposeMatrix == object->world matrix M (glm order)  */
    float motion_blur[3];     /* Shape::motionBlur                                                   */
    int32_t flags;            /* RT_OBJ_*                                                            */
    int32_t material;         /* index into materials                                                */
    int32_t tex_color;        /* ColorChangerTexture index or -1                                     */
    int32_t tex_normal;       /* NormalChangerTexture index or -1                                    */
    int32_t _pad;
} rt_object;                  /* 96 bytes */

#define RT_PRIM_SPHERE 0x40000000   /* bit set in prim_shape[] when the primitive's shape is a sphere     */
#define RT_PRIM_BLURBOX 0x20000000  /* Primitive::bbox was taken after the motion-blur extension (Q5,Q29) */
#define RT_PRIM_INDEX_MASK 0x1FFFFFFF

/* <MeshInstance>: Mesh::Clone (reference src/Mesh.cpp:126-180, parser src/SceneParser.cpp:1075-1129).  The clone's triangles
 * share the base mesh's vertices; the library creates them ("virtual" triangles): instance i owns the triangle indices
 * n_triangles + (sum of n_triangles of the instances before it) + k, k < n_triangles, each with the vertex indices of base
 * triangle first_triangle + k and this instance's object.  On the GPU build path the expansion runs on the device. */
typedef struct rt_instance {
    int64_t first_triangle;   /* first triangle of the base mesh (an explicit triangle, < rt_scene_desc.n_triangles) */
    int64_t n_triangles;      /* number of triangles of the base mesh                                                */
    int32_t object;           /* the clone's object: its own transform (instance matrix composed after the kept or reset
                                 base matrix, Shape::SetTransformation src/Shape.cpp:29-39), material, motion blur   */
    int32_t _pad;
} rt_instance;                /* 24 bytes */

/* A run of consecutive entries of Scene::primitives: shapes first, first + 1, ... first + count - 1 (triangle indices,
 * explicit or virtual, or sphere indices when flags has RT_PRIM_SPHERE), every entry with the same flags.  A <Mesh> is two
 * runs over its triangles (constructor, then SetMotionBlur with RT_PRIM_BLURBOX: Q5), a <MeshInstance> one run. */
typedef struct rt_prim_run {
    int64_t first;
    int64_t count;
    int32_t flags;            /* RT_PRIM_SPHERE and / or RT_PRIM_BLURBOX */
    int32_t _pad;
} rt_prim_run;                /* 24 bytes */

typedef struct rt_scene_desc {
    uint32_t abi_version;     /* RT_ABI_VERSION */
    int32_t max_depth;        /* Scene::maxRecursionDepth   */
    float isect_eps;          /* Scene::intTestEps (Q-cfg)  */
    float shadow_eps;         /* Scene::shadowRayEps        */
    float background[3];      /* Scene::backgroundColor     */
    float ambient[3];         /* Scene::ambientLight        */
    int32_t bg_texture;       /* texture index or -1        */
    int32_t _pad0;

    int64_t n_vertices;       /* reference Vertex objects (per mesh, de-duplicated by index, Q28) */
    const float *vert_pos;    /* [n_vertices][3] object-space positions                          */
    const float *vert_uv;     /* [n_vertices][2]                                                 */

    int64_t n_triangles;
    const int32_t *tri_vert;  /* [n_triangles][3] vertex indices */
    const int32_t *tri_object;/* [n_triangles] object index      */

    int64_t n_spheres;
    const float *sph_center_radius; /* [n_spheres][4] centre xyz (object space), radius */
    const int32_t *sph_object;      /* [n_spheres] object index                         */

    int64_t n_objects;
    const rt_object *objects;

    int64_t n_prims;          /* Scene::primitives, in the reference's push order (Q5)            */
    const int32_t *prim_shape;/* [n_prims] triangle or sphere index | RT_PRIM_* flags             */

    int32_t n_materials;
    int32_t n_lights;
    int32_t n_textures;
    int32_t _pad1;
    const rt_material *materials;
    const rt_light *lights;
    const rt_texture *textures;
    const uint8_t *texels;    /* image arena addressed by rt_texture.texel_offset */
    int64_t texels_bytes;

    /* compact forms (all optional; zero / NULL = not used) */
    int64_t n_instances;      /* <MeshInstance> clones for the library to expand (see rt_instance)                    */
    const rt_instance *instances;
    int64_t n_prim_runs;      /* when prim_runs != NULL it replaces prim_shape / n_prims (see rt_prim_run)             */
    const rt_prim_run *prim_runs;
    int32_t geometry_on_device; /* 1: vert_pos, vert_uv, tri_vert and tri_object are DEVICE pointers (for example filled by
                                 rt_ply_decode_to): the geometry never visits the host                               */
    int32_t _pad2;
} rt_scene_desc;

/* ---- camera: reference src/Camera.h:20-85 (constructor arguments, before SetupCameraCoordinateAxes) ---- */
typedef struct rt_camera_desc {
    float position[3];
    float gaze[3];
    float up[3];
    float left, right, bottom, top, distance; /* Camera::ImagePlane */
    int32_t nx, ny;
    int32_t num_samples;      /* Camera::nSamples */
    float focus_distance;
    float aperture_size;
} rt_camera_desc;

typedef struct rt_render_opts {
    uint32_t abi_version;
    uint32_t seed;            /* key of the counter-based RNG (Philox4x32-10) that replaces Random.h   */
    int32_t tile_size;        /* interleaved square tiles; <=0 means one tile = whole frame            */
    int32_t tile_rank;        /* this context renders tiles with (tile_id % tile_world) == tile_rank   */
    int32_t tile_world;       /* <=1: render every tile                                                */
    int32_t output_on_device; /* 1: hdr_rgb / prim_id / prim_t are DEVICE pointers (HBM-resident)      */
    int32_t compact_output;   /* 1: outputs hold only owned pixels, in rt_tile_pixel_order() order      */
    int32_t max_batch_samples;/* samples traced per wavefront batch; <=0 selects the default (8M)      */
    int32_t timing;           /* 1: bracket kernels with HIP events and fill rt_stats ms fields        */
    int32_t num_streams;      /* sub-batches in flight on separate HIP streams; <=0 selects the default (1), max 8 */
    int32_t level_counts;     /* 0: after a scene's first frame the lengths of the ray queues stay on the device (no host round
                                 trip between ray generation and resolve; rt_stats.frame_passes = 2 when a prediction failed);
                                 1: the host reads every level's length (one round trip per level, exact queue sizes)       */
    int32_t shadow_stream;    /* 1: in frames whose queue lengths stay on the device, the shadow rays of a level and the lighting that
                                 needs them run on a second HIP stream beside the closest-hit chain of the deeper levels (14 trace
                                 launches per frame instead of 8 fused ones; measured: -1.9 % at 800x480, +5 % at 1600x960);
                                 0: one fused launch per level on one stream.  Results do not depend on it. */
} rt_render_opts;

enum { RT_RAY_PRIMARY = 0, RT_RAY_SHADOW = 1, RT_RAY_REFLECT = 2, RT_RAY_REFRACT = 3, RT_RAY_KINDS = 4 };

typedef struct rt_stats {
    uint64_t rays[RT_RAY_KINDS];  /* closest-hit queries per kind (one "ray" = one AccelerationStructure::Intersect) */
    uint64_t node_visits;         /* internal nodes whose child boxes were tested   */
    uint64_t box_tests;           /* child/root boxes slab-tested                   */
    uint64_t tri_tests;           /* triangle intersection tests                    */
    uint64_t sphere_tests;        /* sphere intersection tests                      */
    double ms_total;              /* wall time of the whole rt_render call          */
    double ms_kernels;            /* HIP-event time first kernel -> last kernel     */
    double ms_trace;              /* summed HIP-event durations of the trace-kernel launches (launches on
                                     different streams overlap, so this can exceed ms_kernels)                  */
    uint64_t trace_launches;      /* number of trace-kernel launches                */
    uint64_t trace_rays;          /* rays processed by those launches               */
    double ms_trace_busy;         /* time during which at least one trace launch was running (union of intervals) */
    uint64_t level_syncs;         /* host round trips for a level's ray count between ray generation and resolve: 0 when the
                                     frame was rendered from predicted queue sizes (every frame of a scene after its first) */
    uint64_t frame_passes;        /* 1; 2 when a level outgrew its predicted queue and the frame was rendered again          */
} rt_stats;

typedef struct rt_ray {
    float origin[3];
    float direction[3];
    float time;
} rt_ray;                         /* 28 bytes */

typedef struct rt_hit {
    float t;                      /* x-only world parameter (Q2); FLT_MAX when there is no hit       */
    int32_t shape;                /* triangle index, or sphere index | RT_PRIM_SPHERE, or -1 (no hit) */
    float ip[3];                  /* world intersection point                                         */
    float n[3];                   /* world (shading) normal, before any texture tweak                 */
    float uv[2];
} rt_hit;                         /* 40 bytes */

typedef struct rt_bvh_info {
    int64_t n_nodes;              /* internal + leaf nodes of the reference-topology tree */
    int64_t n_leaves;
    int32_t max_depth;
    int32_t n_wide_nodes;         /* 4-wide traversal nodes the reference tree was collapsed into (device layout) */
    int64_t n_leaf_prims;         /* leaf primitive references after duplicate removal    */
} rt_bvh_info;

typedef struct rt_ctx rt_ctx;

/* Creates a context bound to HIP device `hip_device_id`.  Fails with RT_ERR_NO_DEVICE when no GPU is
 * present: there is no CPU fallback.  (Reference: `new DefaultRenderer()`, src/main.cpp:20.) */
int rt_ctx_create(int hip_device_id, rt_ctx **out);

/* Destroys the context and frees its device memory.  (Reference: `delete renderer`, src/main.cpp:26,
 * DefaultRenderer::~DefaultRenderer src/DefaultRenderer.cpp:29-33.) */
void rt_ctx_destroy(rt_ctx *ctx);

/* Last error message of the context (or of rt_ctx_create when ctx == NULL). */
const char *rt_last_error(const rt_ctx *ctx);

/* Builds the reference-topology BVH on the host, flattens scene + BVH into the device layout and
 * uploads it.  Replaces DefaultRenderer::RetrieveRenderingParamsFromScene (src/DefaultRenderer.cpp:47-57),
 * AccelerationStructureFactory::CreateAccelerationStructure (src/AccelerationStructureFactory.cpp:8-21)
 * and BVHTree::BVHTree/BuildTree (src/BVHTree.cpp:10-153). */
int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *scene);

/* Describes the BVH of the uploaded scene. */
int rt_scene_bvh_info(const rt_ctx *ctx, rt_bvh_info *out);

/* Renders one camera of the uploaded scene: replaces DefaultRenderer::RenderCamera up to (not including)
 * Image::SaveImage (src/DefaultRenderer.cpp:59-162): ray generation, closest-hit traversal, shadow rays,
 * shading and the reflect/refract recursion as a wavefront pipeline.
 *   hdr_rgb : [ny*nx*3] float, row-major, the per-pixel mean over samples BEFORE the 8-bit clamp (Q20).
 *   prim_id : optional [ny*nx] int32, shape hit by the first sample's primary ray (-1 = miss).
 *   prim_t  : optional [ny*nx] float, its x-only t.
 * With opts->compact_output the arrays hold rt_owned_pixel_count() entries instead. */
int rt_render(rt_ctx *ctx, const rt_camera_desc *camera, const rt_render_opts *opts,
              float *hdr_rgb, int32_t *prim_id, float *prim_t, rt_stats *stats);

/* Batched form of AccelerationStructure::Intersect (src/AccelerationStructure.h:21; BVHTree::Intersect
 * src/BVHTree.cpp:155): n independent closest-hit queries with the scene's intersection epsilon.
 * `rays` / `hits` are host pointers unless on_device != 0. */
int rt_intersect_batch(rt_ctx *ctx, const rt_ray *rays, size_t n, rt_hit *hits, int on_device,
                       rt_stats *stats);

/* rt_intersect_batch for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc): no counterpart in the reference.  Same queries and results as rt_intersect_batch(on_device = 1), but the
 * call allocates nothing: `d_rays` / `d_hits` are DEVICE pointers the caller owns, and the traversal's scratch (the fetch
 * counter and the stack-overflow columns) is a fixed device array inside the library, rt_noalloc_scratch_bytes() bytes
 * (32 MiB + 256 B) per GPU for the life of the process.  hip_stream: a hipStream_t to launch on (the call then returns
 * without waiting), or NULL for the context's stream (the call then returns after the results are complete).
 * Limits: n <= 2^31-1; a scene whose BVH needs more stack-overflow entries per thread than the array holds for one 256-thread
 * block (32768) is rejected with RT_ERR_TOO_LARGE; deeper trees run with proportionally fewer persistent blocks.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two contexts) race. */
int rt_intersect_batch_noalloc(rt_ctx *ctx, const rt_ray *d_rays, size_t n, rt_hit *d_hits, void *hip_stream);
size_t rt_noalloc_scratch_bytes(void);

/* Host-only helpers (no GPU needed). */

/* Number of pixels rendered by (tile_rank, tile_world) and their row-major pixel indices in the order the
 * compact outputs use (tile by tile, row-major inside a tile).  `out_index` may be NULL. */
int64_t rt_owned_pixel_count(int nx, int ny, int tile_size, int tile_rank, int tile_world);
int rt_tile_pixel_order(int nx, int ny, int tile_size, int tile_rank, int tile_world, int32_t *out_index);

/* The Perlin permutation tables the reference would build: std::shuffle(0..15) driven by one
 * default-constructed std::default_random_engine shared by successive Perlin textures
 * (src/PerlinTextureImpl.cpp:26-33, src/Scene.h:51, src/SceneParser.cpp:551).  Fills n_tables*16 ints. */
int rt_perlin_tables(int n_tables, int32_t *out);

/* ObtainColorFromUnclampedVector (src/DefaultRenderer.cpp:18-27): float RGB -> 8-bit, n = pixel count. */
int rt_resolve_8bit(const float *hdr_rgb, int64_t n_pixels, uint8_t *out_rgb);

/* ---- what the reference does with the finished frame (8-bit image, PPM / EXR file, tone mapper), on the device ---------------------- */

enum { RT_IMG_RGB8 = 0,      /* the reference's Image contents: ny*nx*3 unsigned chars                            */
       RT_IMG_PPM_P3 = 1,    /* the file Image::SaveImageAsPPM writes, byte for byte                                */
       RT_IMG_PPM_P6 = 2,    /* binary PPM of the same 8-bit image (not in the reference; 4-5x smaller)             */
       RT_IMG_EXR_HALF = 3 };/* the file Tonemapper::SaveEXR writes through tinyexr (src/Tonemapper.cpp:34-99,
                                src/tinyexr.h:12126-12598): channels B, G, R as HALF, uncompressed scan lines; the input
                                frame is the float array rt_tonemap returns, no 8-bit step                             */

/* Upper bound of the bytes rt_image_encode produces for an nx*ny frame in `format` (0 for bad arguments). */
size_t rt_image_encode_bound(int nx, int ny, int format);

/* Replaces ObtainColorFromUnclampedVector (src/DefaultRenderer.cpp:18-27: >255 -> 255, float -> unsigned char) and
 * Image::SaveImageAsPPM (src/Image.cpp:54-77: "P3\n%d %d\n255\n", then "%d " per channel and "\n" per row) for a
 * float frame as rt_render produces it.  hdr_rgb is a device pointer when hdr_on_device != 0.  The encoded bytes go to
 * the HOST buffer out_host (capacity out_capacity); *out_bytes receives their number; *ms_kernels (optional) the HIP
 * event time of the kernels. */
int rt_image_encode(rt_ctx *ctx, const float *hdr_rgb, int nx, int ny, int hdr_on_device, int format, void *out_host,
                    size_t out_capacity, size_t *out_bytes, double *ms_kernels);

typedef struct rt_tonemap_opts {
    uint32_t abi_version;
    float key;                /* <TMOOptions> first number  (TonemapSettings::imageBurnKey)                          */
    float burn_percent;       /* <TMOOptions> second number (TonemapSettings::saturationPercentage)                  */
    float saturation;         /* <Saturation>                                                                        */
    float gamma;              /* <Gamma>: parsed, never used by the reference                                        */
    int32_t input_8bit;       /* 1: clamp the frame to 8 bit first, as the reference does (it tone-maps its Image of
                                 Colors, Q20); 0: tone-map the float radiance                                        */
    int32_t in_on_device;     /* hdr_rgb is a device pointer                                                         */
    int32_t out_on_device;    /* out_rgb is a device pointer                                                         */
} rt_tonemap_opts;            /* 32 bytes */

/* Replaces Tonemapper::Tonemap / DefaultTonemapStrategy::ComputeTonemappedValues (src/Tonemapper.cpp:27-32,
 * src/DefaultTonemapStrategy.cpp:15-114): luminance per pixel, exp(mean log luminance), luminance white from the sorted
 * luminances at the burn percentile, Reinhard curve with burn, saturation exponent.  out_rgb: [ny*nx*3] float, the array
 * the reference hands to SaveEXR.  out_params (optional, host): [0] average luminance, [1] luminance white.
 * The reference's function is undefined behaviour as written (FillLuminanceVectorAndRecordSum returns no value, rows are
 * bounded by the image WIDTH, Image.h:60-63, and burn_percent == 0 indexes one past the end); implemented here is what
 * the code computes when the sum is returned, rows are bounded by the height and the index is clamped. */
int rt_tonemap(rt_ctx *ctx, const float *hdr_rgb, int nx, int ny, const rt_tonemap_opts *opts, float *out_rgb,
               float *out_params, double *ms_kernels);

/* Body of a PLY file as the caller's header parse describes it: binary_little_endian, or ascii (format = 1).  For an ascii
 * body "bytes" reads "tokens": vertex_stride = scalar properties per vertex line, pos_offset = token index of x, y, z,
 * face_pre_bytes / face_post_bytes = scalar properties before / after the list; count_size and index_size are unused. */
enum { RT_PLY_BINARY_LE = 0, RT_PLY_ASCII = 1 };
typedef struct rt_ply_layout {
    uint32_t abi_version;
    int32_t pos_type;         /* type of the x, y, z properties: 0 = float (32 bit), 1 = double                          */
    int64_t n_vertices;
    int32_t vertex_stride;    /* bytes per vertex record (all its properties are scalars)                                */
    int32_t pos_offset[3];    /* byte offsets of x, y, z inside a vertex record                                          */
    int64_t n_faces;
    int32_t face_pre_bytes;   /* bytes of scalar properties before the vertex_indices list in a face record              */
    int32_t count_size;       /* bytes of the list's count field: 1, 2 or 4                                              */
    int32_t index_size;       /* bytes per index: 1, 2 or 4                                                              */
    int32_t face_post_bytes;  /* bytes of scalar properties after the list                                               */
    int32_t format;           /* RT_PLY_BINARY_LE or RT_PLY_ASCII                                                        */
    int32_t _pad;
} rt_ply_layout;              /* 64 bytes */

/* Replaces the PLY branch of the scene parser (src/SceneParser.cpp:983-1023: happly's getVertexPositions() rounded to
 * Vector3f, getFaceIndices() with a face of three indices kept as a triangle, a face of four split into (0,1,2), (2,3,0),
 * any other face dropped) for a binary body; the text header stays with the caller.  vertex_bytes / face_bytes: HOST
 * pointers to the first vertex / face record (face_bytes_size = bytes available from there).  out_pos: HOST [n_vertices*3]
 * float.  out_tris: HOST [up to 2*n_faces*3] int32 vertex indices, *out_n_tris their number.  Indices are not range-checked
 * here (rt_scene_upload does that).  No CPU fallback: RT_ERR_NO_DEVICE without a GPU. */
int rt_ply_decode(rt_ctx *ctx, const void *vertex_bytes, const void *face_bytes, size_t face_bytes_size, const rt_ply_layout *layout,
                  float *out_pos, int32_t *out_tris, int64_t *out_n_tris, double *ms_kernels);
/* For an ascii body (layout->format == RT_PLY_ASCII: what happly's parseASCII does, src/happly.h:2075-2095 — one element per line,
 * tokens split at ' ', values through operator>> of the property's type) pass the whole body text as vertex_bytes and its
 * length as face_bytes_size; face_bytes is ignored.  Decimal numbers are converted on the device where that is provably the
 * correctly rounded result (at most 19 significant digits, a power of ten that is exact in the working type); the few tokens
 * outside that are converted by the host with the very stream extraction the reference's library uses. */

/* The same decoder with its output left in HBM: positions go to the DEVICE array d_out_pos [n_vertices * 3], triangles to the
 * DEVICE array d_out_tris [up to 2 * n_faces * 3] with index_offset added to every vertex index (the mesh's first vertex
 * inside a scene-wide vertex array).  Together with rt_scene_desc.geometry_on_device a mesh goes from the file bytes to the
 * traversal arrays without its geometry visiting the host. */
int rt_ply_decode_to(rt_ctx *ctx, const void *vertex_bytes, const void *face_bytes, size_t face_bytes_size, const rt_ply_layout *layout,
                     float *d_out_pos, int32_t *d_out_tris, int32_t index_offset, int64_t *out_n_tris, double *ms_kernels);

/* ---- one build, many GPUs: the uploaded scene as a header + flat arenas ----------------------------------------------------
 * The reference holds one Scene and one BVHTree that all its worker threads read (DefaultRenderer.cpp:29-33, 65-75); with one
 * process per GPU every rank either builds its own copy (rt_scene_upload on every rank, the default) or one rank builds and the
 * others receive the result: the arenas hold indices only, no pointers, so their bytes are valid on any GPU.
 *   rt_scene_header_bytes()                 size of the opaque header (a few hundred bytes: parameters, counts, tree statistics)
 *   rt_scene_export(ctx, header, bytes)     fills the header and the byte size of each of the RT_SCENE_ARENAS arenas
 *   rt_scene_export_arena(ctx, i, dst)      copies arena i to dst (device memory of any GPU of the process, or host memory)
 *   rt_scene_import(ctx, header, arenas)    makes ctx hold that scene: arenas[i] = bytes[i] bytes received from the exporting
 *                                           rank (device or host memory; the library copies them into arenas of its own)
 * The transport in between is the caller's (ncclBroadcast / torch.distributed.broadcast of byte buffers).  A header of another
 * library build (different ABI version or record sizes) is rejected with RT_ERR_ABI. */
#define RT_SCENE_ARENAS 12
size_t rt_scene_header_bytes(void);
int rt_scene_export(rt_ctx *ctx, void *header, uint64_t arena_bytes[RT_SCENE_ARENAS]);
int rt_scene_export_arena(rt_ctx *ctx, int index, void *dst);
int rt_scene_import(rt_ctx *ctx, const void *header, const void *const arenas[RT_SCENE_ARENAS]);

/* Host-only view of the BVH that rt_scene_upload would build (reference topology, BVHTree.cpp:39-153), so that tests
 * can compare it with an independent build without a GPU.  Leaves are listed in left-to-right DFS order.
 *   leaf_start_end      [2*n_leaves]  range of each leaf in the reference's re-ordered primitive array
 *   prim_order          [n_prims]     shape id (rt_hit.shape convention) of every re-ordered reference primitive
 *   leaf_boxes          [6*n_leaves]  min xyz, max xyz
 *   dev_leaf_start_count[2*n_leaves]  run of each leaf in the device primitive records (duplicates removed)
 *   dev_prim_shape      [n_leaf_prims] shape id of every device primitive record
 * Any output pointer may be NULL. */
typedef struct rt_host_bvh rt_host_bvh;
int rt_host_bvh_build(const rt_scene_desc *scene, rt_host_bvh **out);
/* The same view with the tree built by the GPU builder (what rt_scene_upload uses for scenes of >= 4096 primitives;
 * here for any size).  Replaces BVHTree::BuildTree (reference src/BVHTree.cpp:39-153) with identical topology, primitive
 * order and boxes; needs a HIP device (RT_ERR_NO_DEVICE otherwise). */
int rt_host_bvh_build_gpu(const rt_scene_desc *scene, int hip_device_id, rt_host_bvh **out);
int rt_host_bvh_info(const rt_host_bvh *bvh, rt_bvh_info *out);
int rt_host_bvh_leaves(const rt_host_bvh *bvh, int64_t *leaf_start_end, int32_t *prim_order, float *leaf_boxes,
                       int64_t *dev_leaf_start_count, int32_t *dev_prim_shape);
/* The pruning rule of the traversal, for tests.  A child of a 4-wide traversal node may be skipped because its box begins
 * beyond the ray's best hit only if no triangle below it is ill-conditioned for that ray — the reference does not prune by
 * distance at all (src/BVHTree.cpp:160-206) and its float Cramer solve (src/Triangle.cpp:186-250) puts the "hit" of a
 * triangle that is nearly in a plane with the ray anywhere on the ray.  rt_host_bvh_cones lists, for every child of every wide
 * node, the packed normal cone that decides this and the [start, end) range of the child's subtree in prim_order; it returns
 * their number (either pointer may be NULL).  rt_cone_allows_prune evaluates the rule the kernels apply: 1 = a subtree
 * with this cone word may be pruned by distance for a ray of direction[3]. */
int64_t rt_host_bvh_cones(const rt_host_bvh *bvh, uint32_t *words, int64_t *start_end);
int rt_cone_allows_prune(uint32_t word, const float *direction);
void rt_host_bvh_free(rt_host_bvh *bvh);

/* sizeof() of the ABI structs, in header order (0 rt_material, 1 rt_light, 2 rt_texture, 3 rt_object, 4 rt_scene_desc,
 * 5 rt_camera_desc, 6 rt_render_opts, 7 rt_stats, 8 rt_ray, 9 rt_hit, 10 rt_bvh_info, 11 rt_tonemap_opts, 12 rt_ply_layout,
 * 13 rt_instance, 14 rt_prim_run); -1 for anything else. */
int rt_abi_sizeof(int which);

uint32_t rt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355_H_ */
