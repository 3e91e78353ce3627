/*
 * mirt.h -- C ABI of libmirt.so, the MI355X-native replacement for the hot path of
 * GJ0407790/cuda_ray_tracer (LBVH build + BVH-traversal render).
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference
 * tree).  Plain C: opaque handles, POD structs, pointers and sizes; no C++ or torch types.
 * All functions return 0 on success and a non-zero MirtStatus otherwise; mirt_last_error() returns a
 * thread-local description.  The library never calls exit() (the reference's CUDA_CHECK does,
 * main.cu:14-23); the CLI maps a non-zero status to the reference's message + exit code.
 *
 * Device pointers are ordinary HIP device pointers (hipMalloc, or torch tensor data_ptr()).
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).  A MirtScene is bound to the
 * HIP device it was created on and is not thread-safe; render calls are asynchronous on `stream`
 * unless stated otherwise.  There is no CPU fallback: without a HIP device scene creation fails.
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRT_VERSION 3

typedef enum MirtStatus {
  MIRT_OK = 0,
  MIRT_ERR_IO = 1,          /* "Error opening file..."            parse.cpp:22-25 */
  MIRT_ERR_PARSE = 2,       /* "One of the lines are not valid."  parse.cpp:218-221 */
  MIRT_ERR_ARG = 3,
  MIRT_ERR_HIP = 4,         /* any HIP runtime failure (CUDA_CHECK, main.cu:14-23) */
  MIRT_ERR_NO_DEVICE = 5,
  MIRT_ERR_STATE = 6        /* e.g. render before build; a capacity overflow reported by mirt_get_stats */
} MirtStatus;

/* ---- POD scene structs: same field order and size as the reference's classes -------------------- */
typedef struct MirtVec3 { float x, y, z; } MirtVec3;                       /* vec3.cuh:21-92, 12 B */
typedef struct MirtRGB { float r, g, b; } MirtRGB;                         /* struct.cuh:11-34, 12 B */
typedef struct MirtMaterials {                                             /* object.cuh:17-38, 44 B */
  MirtRGB color, shininess, trans;
  float ior, roughness;
} MirtMaterials;
typedef struct MirtSphere { MirtVec3 c; float r; MirtMaterials mat; } MirtSphere;            /* object.cuh:95-119, 60 B */
typedef struct MirtTriangle { MirtVec3 p0, p1, p2, nor, e1, e2; MirtMaterials mat; } MirtTriangle; /* object.cuh:165-194, 116 B */
typedef struct MirtPlane { float a, b, c, d; MirtVec3 nor, point; MirtMaterials mat; } MirtPlane;  /* object.cuh:124-149, 84 B */
typedef struct MirtSun { MirtVec3 dir; MirtRGB color; } MirtSun;           /* object.cuh:232-248, 24 B */
typedef struct MirtBulb { MirtVec3 point; MirtRGB color; } MirtBulb;       /* object.cuh:250-266, 24 B */
typedef struct MirtPrimRef { uint32_t type; uint32_t id; } MirtPrimRef;    /* object.cuh:72-88, 8 B; 0 sphere, 1 triangle */

/* Scalars of StlConfig/RawConfig (config.hpp:24-73, 75-126) + host arrays in file order.
 * Pointers are borrowed for the duration of the call that takes the descriptor. */
typedef struct MirtSceneDesc {
  int32_t width, height, bounces, aa;
  float dof_focus, dof_lens;
  MirtVec3 forward, right, up, eye;
  float expose;                       /* +inf = exposure off (config.hpp:50) */
  int32_t fisheye, panorama, gi;
  int32_t num_spheres, num_triangles, num_prims, num_planes, num_suns, num_bulbs;
  const MirtSphere* spheres;
  const MirtTriangle* triangles;
  const MirtPrimRef* prim_refs;       /* host_primitive_references, config.hpp:64 */
  const MirtPlane* planes;
  const MirtSun* suns;
  const MirtBulb* bulbs;
} MirtSceneDesc;

typedef struct MirtHostScene MirtHostScene;   /* parsed scene on the host  (StlConfig, config.hpp:24-73) */
typedef struct MirtScene MirtScene;           /* device-resident scene     (RawConfig, config.hpp:75-126) */

const char* mirt_last_error(void);
int mirt_version(void);

/* ---- scene front end -------------------------------------------------------------------------- */
/* parseInput(argv, StlConfig&), parse.hpp:10 / parse.cpp:16-39.  Same grammar (parse.cpp:41-222). */
int mirt_parse_scene_file(const char* path, MirtHostScene** out);
int mirt_parse_scene_text(const char* text, size_t len, MirtHostScene** out);
/* Deterministic synthetic scene of BASELINE config 5; not in the reference. */
int mirt_synthetic_scene(uint64_t seed, int num_spheres, int num_triangles, MirtHostScene** out);
void mirt_host_scene_destroy(MirtHostScene* hs);
int mirt_host_scene_desc(const MirtHostScene* hs, MirtSceneDesc* out);   /* pointers stay owned by hs */
const char* mirt_host_scene_filename(const MirtHostScene* hs);           /* the `png W H name` name, parse.cpp:47-51 */

/* ---- device scene ----------------------------------------------------------------------------- */
/* initRawConfigFromStl + copyConfigDataToDevice, config_utils.cuh:11-17 / config_utils.cu:18-199.
 * Uploads the scene to HIP device `device` in the SoA layout of DESIGN.md. */
int mirt_scene_create(const MirtSceneDesc* desc, int device, MirtScene** out);
/* freeRawConfigDeviceMemory, config_utils.cuh:20 (frees everything; the reference leaks the SoA arrays). */
void mirt_scene_destroy(MirtScene* sc);

/* Mode switches and tuning values of a scene, by name.  Not in the reference (its knobs are compile-time constants).
 *   build (take effect at the next mirt_build_lbvh):
 *     "bounds_as_shipped" 0/1 (default 0): 1 reproduces the shipped reference's tree -- scene bounds never stored
 *                         (parse.cpp:28), every Morton code 0
 *   render:
 *     "traversal"         MIRT_TRAVERSAL_*: 0 the reference's left-first descent (bvh_traversal.cu:149-157); 1 (default)
 *                         near-child-first on the quantised records of a sphere-only scene (see qnodes) -- same pixels, fewer
 *                         visits; the reference's order wherever the exact records are walked (scenes with triangles below
 *                         65536 primitives, qnodes 0, wavefront) and over the wide records; 2 near-child-first everywhere (a
 *                         sample may differ where the reference's own result depends on its visiting order: triangle
 *                         silhouettes, sphere hits within an ulp of their box from a far camera)
 *     "qnodes"            0/1/2 (default 1): quantised node records in the single-kernel path.  1: sphere-only scenes walk 32-byte
 *                         records (two memory requests per node visit instead of four; traversal >= 1); scenes with triangles of
 *                         65536 primitives or more walk the wide records (the boxes of a node's four grandchildren per 64-byte
 *                         record, two levels per step, the reference's order: traversal 1 only; a triangle hit the reference's
 *                         walk may not reach sends its ray over the exact records again).  2: every scene.  0: never.  Same
 *                         pixels in every case; not with wavefront; and only where the records' grid resolves the scene's
 *                         coordinates (on every axis below 64 extents of the scene box: a scene that sits far from the world
 *                         origin compared with its size walks the exact records).  Over the quantised records the sphere hit
 *                         a nearest-hit walk ends with -- shadow rays to point lights included -- is checked against the
 *                         reference's own box test, and a ray whose hit the reference may never test is walked again its way
 *     "shadow_anyhit"     0/1 (default 1): a shadow ray ends at its first occluder; 0: a nearest-hit query like every other ray,
 *                         as diffuseLight does (draw.cu:347-352, 365-370) -- same boolean, more node visits
 *     "skip_unlit"        0/1 (default 1): shadow rays towards lights the shading normal faces away from are not traced (their
 *                         term is 0 either way, draw.cu:353-357); 0: every light's shadow ray is traced (draw.cu:342-374)
 *                         {traversal 0, shadow_anyhit 0, skip_unlit 0, qnodes 0} is the reference's walk, ray for ray
 *     "specialise"        0/1 (default 1): a scene without point lights (and, sphere-only scenes, without transparent materials
 *                         and gi) is rendered by the kernel compiled without those features (same pixels and counters;
 *                         0: the general kernel)
 *     "wavefront"         0/1: the trace/shade kernel pair instead of the single kernel
 *     "slab_log2"         (default 28) a call is rendered in slabs of at most 2^slab_log2 samples: 16 B of workspace per
 *                         sample, i.e. at most 4 GiB per frame in flight, however large the frame
 *     "sched"             0/1/2 (default 2): longest-first hand-out of a call's samples, measured by the first call of a shape and reused
 *                         (the scene is immutable): 2 by sample (every sample in its cost class, expensive classes first, positions
 *                         within a class kept; 4 B per sample), 1 by chunk of 64..256 samples (one-slab calls), 0 frame order
 *     "stack_lds_depth", "refill_k", "init_k", "batch_k", "leaf_k", "reps", "drain_lanes", "chunk_shift", "trace_waves",
 *     "wf_pool", "wf_refill_k": tuning (defaults are the measured optima)
 * Environment variables MIRT_<NAME> override the defaults of the TUNING values once, when the scene is created (the mode
 * switches -- bounds_as_shipped, traversal, wavefront, qnodes, shadow_anyhit, skip_unlit -- only with MIRT_ALLOW_ENV=1); nothing
 * reads the environment during a render. */
#define MIRT_TRAVERSAL_REFERENCE 0
#define MIRT_TRAVERSAL_ORDERED 1
#define MIRT_TRAVERSAL_ORDERED_ALL 2
int mirt_scene_set_option(MirtScene* sc, const char* name, int value);
int mirt_scene_get_option(const MirtScene* sc, const char* name, int* value);

/* build_lbvh_karas(RawConfig&, int morton_bits), lbvh_builder.cuh:14 / lbvh_builder.cu:401-521:
 * scene bounds -> 30-bit Morton codes -> stable radix sort -> Karras hierarchy -> AABB refit -> 64-byte
 * two-child node records, primitive records in sorted order (+ 32-byte quantised node records for sphere-only scenes).  Synchronous (like the reference, lbvh_builder.cu:475).  build_ms (nullable)
 * receives the device time measured with HIP events (the reference prints it, lbvh_builder.cu:489). */
int mirt_build_lbvh(MirtScene* sc, void* stream, float* build_ms);

/* ---- render ----------------------------------------------------------------------------------- */
#define MIRT_RENDER_COUNTERS 1u   /* also count rays / node visits / leaf tests (slower kernel variant) */

/* The frame is cut into horizontal stripes of `stripe_rows` rows; stripe i belongs to part
 * (i % num_parts).  A call renders the stripes of part `part` into a compact buffer (the part's stripes
 * in increasing order, each row-major).  num_parts = 1, part = 0 renders the whole frame row-major. */
typedef struct MirtRenderParams {
  int32_t width, height;     /* frame size (RawConfig::width/height) */
  int32_t spp;               /* the reference's `aa`: 0 = one un-jittered sample, 1 = one jittered, >1 = spp samples */
  int32_t stripe_rows, num_parts, part;
  uint32_t flags;
} MirtRenderParams;

/* number of pixels the call writes */
int64_t mirt_render_num_pixels(const MirtRenderParams* p);

/* render(pixel_t* d_image, w, h, aa, RawConfig*), draw.cuh:10 / draw.cu:215-239.
 * d_rgba8: num_pixels * 4 bytes, RGBA (pixel_t, libpng.h:23-27).  d_rgba_f32 (nullable): num_pixels * 4
 * floats, the linear RGBA sample mean before sRGB/quantisation (for parity checks). */
int mirt_render(MirtScene* sc, const MirtRenderParams* p, void* d_rgba8, void* d_rgba_f32, void* stream);

/* render_kernel_atomic_aa + finalize_kernel, draw.cu:13-92 (in the reference tree but not called by its render()): the
 * accumulate / finalise pair for progressive rendering and for any number of samples per pixel.
 * mirt_render_accumulate adds, for every pixel of the part, the samples [sample_first, sample_first + sample_count) to
 * d_accum_f32 (num_pixels * 4 floats, zeroed by the caller before the first call); sample s of pixel p is seeded
 * curand_init(1234 + p, s, 0) and jittered (draw.cu:74-84) whatever p->spp says.  The reference adds with atomicAdd, i.e. in
 * no particular order; here one call adds one value per pixel -- the sum of its samples in the xor-butterfly order of
 * draw.cu:181-189 -- so results do not depend on timing.  At most 4096 sample indices per call; any number over several calls.
 * mirt_finalize writes the 8-bit image: mean over total_samples, sRGB, clamp * 255 + 0.5 (draw.cu:22-46).
 * A single mirt_render_accumulate of samples [0, spp) followed by mirt_finalize gives the bytes of mirt_render (spp > 1). */
int mirt_render_accumulate(MirtScene* sc, const MirtRenderParams* p, void* d_accum_f32, int sample_first, int sample_count, void* stream);
int mirt_finalize(const MirtRenderParams* p, const void* d_accum_f32, int total_samples, void* d_rgba8, void* stream);

/* Where local pixel `local` of a part's compact buffer lies in the frame (host arithmetic: the mapping mirt_render,
 * mirt_scatter_part and the multi-GPU gather use).  Returns MIRT_ERR_ARG when `local` is not a pixel of the part. */
int mirt_part_pixel_xy(const MirtRenderParams* p, int64_t local, int32_t* x, int32_t* y);
/* Scatter a compact part buffer back into a full row-major frame (device to device). */
int mirt_scatter_part(const MirtRenderParams* p, const void* d_part_rgba8, void* d_frame_rgba8, void* stream);

/* ---- several GPUs in one process --------------------------------------------------------------- */
/* Not in the reference (single GPU, main.cu:25-94).  The scene is uploaded to every listed device and every device builds
 * the identical LBVH; a frame is cut into interleaved stripes of `stripe_rows` rows, device r renders part r (a
 * MirtRenderParams with num_parts = ngpu, part = r), the parts are gathered to the first device with one grouped RCCL
 * send/recv exchange over xGMI and re-interleaved there.  Any partition gives the bytes of the single-GPU frame (samples are
 * seeded by global pixel and sample index, draw.cu:162).  RCCL is loaded at run time, and only for ngpu > 1. */
#define MIRT_MULTI_MAX_GPUS 16
typedef struct MirtMulti MirtMulti;
typedef struct MirtMultiStats {
  int32_t num_gpus;
  float build_ms;                          /* slowest device's LBVH build */
  float render_ms[MIRT_MULTI_MAX_GPUS];    /* device time of each part's render (HIP events on its stream) */
  float gather_ms;                         /* end of device 0's render -> frame re-interleaved on device 0 (includes waiting for the slowest peer) */
  float frame_ms;                          /* host wall clock of the call, host copy included */
} MirtMultiStats;
/* devices: ngpu device indices, or NULL for 0..ngpu-1.  Uploads and builds synchronously, all devices at once (one host thread
 * each).  MIRT_MULTI_GATHER=copy in the environment: peer-to-peer copies instead of RCCL, and a device may be listed more than
 * once -- or, with devices == NULL, ngpu may exceed the GPUs present (part r on GPU r mod their number): parts time-sharing a
 * GPU, a rehearsal of the N > 1 path on a small box, never a scaling measurement. */
int mirt_multi_create(const MirtSceneDesc* desc, int ngpu, const int* devices, MirtMulti** out);
void mirt_multi_destroy(MirtMulti* mm);
int mirt_multi_num_parts(const MirtMulti* mm);
int mirt_multi_set_option(MirtMulti* mm, const char* name, int value);      /* mirt_scene_set_option on every device's scene */
/* Frames in flight: mirt_multi_submit issues one width x height frame at spp samples per pixel on every device and returns at
 * once with a ticket; mirt_multi_wait blocks until that frame is gathered (and copied to host_rgba, nullable, which must stay
 * valid until then).  Up to MIRT_MULTI_MAX_IN_FLIGHT frames may be in flight: consecutive frames overlap on every device (the
 * next frame's waves take the slots the draining frame frees), the gathers run in submission order.  mirt_multi_wait returns
 * MIRT_ERR_STATE when a capacity overflowed on any device (checked whenever no other frame is in flight; MirtStats.overflow_events). */
#define MIRT_MULTI_MAX_IN_FLIGHT 4
int mirt_multi_submit(MirtMulti* mm, int width, int height, int spp, int stripe_rows, uint8_t* host_rgba, uint64_t* ticket);
int mirt_multi_wait(MirtMulti* mm, uint64_t ticket, MirtMultiStats* stats);
/* One frame, synchronously (submit + wait). */
int mirt_render_frame_multi(MirtMulti* mm, int width, int height, int spp, int stripe_rows, uint8_t* host_rgba, MirtMultiStats* stats);
/* nframes frames back to back with `in_flight` of them in flight; host_rgba_last / last_stats (nullable) receive the last frame;
 * ms_per_frame: host wall clock of the call / nframes. */
int mirt_render_frames_multi(MirtMulti* mm, int width, int height, int spp, int stripe_rows, int nframes, int in_flight, uint8_t* host_rgba_last,
                             MirtMultiStats* last_stats, float* ms_per_frame);
/* mirt_get_stats of device `part`'s scene (waits for its frames in flight). */
struct MirtStats;
int mirt_multi_get_stats(MirtMulti* mm, int part, struct MirtStats* out);

typedef struct MirtStats {
  /* filled by a render with MIRT_RENDER_COUNTERS */
  uint64_t samples, rays, shadow_rays, internal_visits, sphere_tests, tri_tests, mat_fetches, max_stack;
  /* capacity overflows of the frames finished since the previous call: a full pending-ray list (refraction / GI children).
   * Always counted; when non-zero mirt_get_stats fills the struct and returns MIRT_ERR_STATE -- the image is not
   * trustworthy.  (The reference's other capacity, its 64-entry traversal stack -- bvh_traversal.cu:8,154-164 prints a
   * warning and drops the subtree -- cannot overflow: the tree is at most 58 levels deep) */
  uint64_t overflow_events;
  /* device time of the last render's trace kernel and of the whole render call, HIP events, ms */
  float trace_kernel_ms, render_ms;
  float build_ms;
  int32_t num_nodes;
  /* mean trace-kernel time over the frames finished since the previous mirt_get_stats call, and their number */
  float trace_kernel_ms_mean;
  int32_t frames_timed;
  /* (version 3) launches of the trace kernel in the last render call (one per slab of at most 2^slab_log2 samples);
   * trace_kernel_ms is the SUM of their durations (one HIP event pair per launch), not a bracket around them */
  int32_t trace_launches;
  /* bytes of one internal-node record of the walk the last render used: 64 (exact boxes, two children per record; also the wide
   * quantised records: four grandchildren per record) or 32 (quantised records of a sphere-only scene) */
  int32_t node_record_bytes;
  /* with MIRT_RENDER_COUNTERS: rays that entered the BVH walk (`rays` counts the reference's hitNearest calls: a shadow ray towards
   * a light the surface faces away from, or one an infinite plane already blocks, is answered without a walk) */
  uint64_t rays_traversed;
} MirtStats;
/* Waits for every frame in flight, then reports (and resets the running mean).  mirt_render / mirt_render_accumulate never
 * report a capacity overflow themselves (they are asynchronous): poll this call after a render, or before using its image.  Up to four frames may be in flight on
 * different streams: a scene keeps four sets of render workspaces and reuses one only when its frame has finished. */
int mirt_get_stats(MirtScene* sc, MirtStats* out);

/* ---- introspection for parity tests ----------------------------------------------------------- */
typedef struct MirtTreeNode {      /* the reference's LBVHNode (lbvh.cuh:6-28), flattened */
  float xmin, xmax, ymin, ymax, zmin, zmax;
  uint32_t left, right, prim_offset, count;
} MirtTreeNode;
/* Copies the tree back in the reference's numbering: internal nodes [0,N-2], leaves [N-1,2N-2].
 * nodes: 2N-1 entries; codes: N sorted Morton codes; refs: N sorted primitive references;
 * bounds: 6 floats (min xyz, max xyz).  Any pointer may be NULL. */
int mirt_get_tree(MirtScene* sc, MirtTreeNode* nodes, uint32_t* codes, MirtPrimRef* refs, float* bounds);

/* Device-side probes of the arithmetic the kernels use (which: 0 logf 1 expf 2 sinf 3 cosf 4 pow(x,1/2.4)
 * 5 RGBtosRGB 6 sqrtf 7 1/x), and of the XORWOW generator. */
int mirt_probe_math(int device, int which, int n, const float* host_in, float* host_out);
/* Stream i of num_streams is seeded exactly as the trace kernel seeds it: spp > 1: pixel i / spp, sample i % spp
 * (curand_init(1234 + pixel, sample, 0), draw.cu:162); spp <= 1: pixel i (curand_init(1234, pixel, 0), draw.cu:105).
 * host_out[i * draws + k] is the k-th raw 32-bit output. */
int mirt_probe_xorwow(int device, int spp, int num_streams, int draws, uint32_t* host_out);

/* ---- output ----------------------------------------------------------------------------------- */
/* Image::save, libpng.cpp:73-107: 8-bit RGBA, non-interlaced PNG (own encoder; zlib only). */
int mirt_write_png(const char* path, const uint8_t* rgba, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
