// Device-resident scene (the reference's RawConfig, config.hpp:75-126) and kernel argument structs.
#ifndef MIRT_SCENE_DEV_H
#define MIRT_SCENE_DEV_H

#include <hip/hip_runtime.h>
#include <string>
#include <vector>

#include "../../include/mirt.h"
#include "device_common.h"
#include "xorwow_tables.h"

namespace mirt {

// child reference inside a packed node / the root reference: names a record of the heap [nodes | primitives]
//   bit 31 = leaf (a primitive record); bit 30 = primitive type (0 sphere, 1 triangle);
//   bits 0..27 = the record's offset in the heap in 16-byte units (node i: 4 i; the primitive of sorted leaf j:
//   prim_base/16 + j + 2 * (triangles among leaves [0, j))), so that the traversal step gets a record's byte offset with
//   one shift.  Primitive records are stored in sorted (Morton) order: the offset of a leaf grows with its sorted index,
//   which is what breaks exact ties in the hit distance the way the reference's left-first walk does.
constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_TRI = 0x40000000u;
constexpr uint32_t REF_OFFMASK = 0x0fffffffu;  // the record's offset in the heap, in 16-byte units: `ref << 4` is its byte offset (the shift drops the flags)
constexpr uint32_t REF_NONE = 0xf0000000u;   // no record (bits 29..28 are set in no real reference; offset bits 0)
// Quantised node records (option "qnodes"): 32 bytes instead of 64, i.e. two 16-byte requests per node visit instead of four
// and half the bytes -- the address units bound the trace kernel on the bundled scenes, memory on the 2 M-primitive one.  Both
// child boxes as 12 x uint16 on the scene-bounds grid (65535 steps per axis, rounded outwards by one more step), then the two
// child references.  The boxes are supersets of the exact ones, so the walk visits a superset of the reference's nodes in the
// same order.  For spheres the closest hit cannot change (shade_common.h, sphere_leaf_box_admits restores the one clause of the
// reference's box test that a larger box weakens).  A triangle hit is accepted only if the reference's walk provably reaches
// that leaf (triangle_leaf_reached, render.hip); otherwise the ray is walked again over the 64-byte exact records.
//   words 0..2 child 0: (xmin | xmax << 16), (ymin | ymax << 16), (zmin | zmax << 16); words 3..5 child 1; 6, 7: references
//   (bit 29 of word 6, REF_QPURE: both subtrees of this node hold spheres only -- NODE_SWAP_PURE of the 64-byte record)
// Node i sits at heap offset qnode_base + 32 i: its reference is qnode_base / 16 + 2 i.
constexpr float QGRID = 65535.0f;
constexpr uint32_t REF_QPURE = 0x20000000u;
// Wide quantised records (scenes with triangles, option "wide"): one 64-byte record per internal node P holding the boxes of P's
// GRANDchildren (a child of P that is a leaf stands for itself) in the reference's visiting order -- up to four boxes of
// 3 words each (words 0..11; an absent slot holds an inverted box, which no ray hits) -- and their four references (words
// 12..15).  A step tests four boxes and descends two levels: half as many dependent memory round trips per ray.  The walk
// visits a superset of the reference's leaves in the reference's order (render.hip), which is all that exactness needs.
// Node i sits at heap offset wnode_base + 64 i.
constexpr uint32_t QBOX_NONE = 0x0000ffffu;   // lo = 65535, hi = 0
constexpr int STACK_TOTAL_WIDE = 88;          // up to three pushes per two levels of a tree at most 58 levels deep
constexpr float QINV_STEPS = 1152921504606846976.0f;   // 2^60: |1 / d| is clamped to this many grid steps per unit of t (quantised_axis)
// node record word 14 (after the two child references): which descent orders the node allows
constexpr uint32_t NODE_SWAP_PURE = 1u;       // both subtrees hold spheres only: near-child-first cannot change the closest hit
constexpr uint32_t NODE_SWAP_ANY = 2u;        // always set (the mask of MIRT_TRAVERSAL_ORDERED_ALL)

struct PlaneDev { float nx, ny, nz, px, py, pz; float mat[11]; float pad; };   // 72 B
// suns: direction; bulbs: position.  n* = normalize(direction) and i* = 1 / n* for suns, computed on the host with the same
// IEEE operations the kernels would use (vec3::normalize, bvh_traversal.cu:106), so shadow rays to suns need no sqrt/div.
struct LightDev { float x, y, z, r, g, b; float nx, ny, nz, ix, iy, iz; };

// Arguments of the trace kernel (passed by value in the kernarg segment: uniform, scalar loads)
struct RenderArgs {
  // frame / camera (RawConfig scalars)
  int width, height, bounces, spp, gi;
  int fisheye, panorama;
  float dof_focus, dof_lens, expose;
  f3 forward, right, up, eye;
  // partition (MirtRenderParams)
  int stripe_rows, num_parts, part;
  long long num_local_pixels;     // pixels of this launch (a slab of the call's part)
  long long num_samples;          // num_local_pixels * samples per pixel of this launch
  long long pixel_base;           // first local pixel of this launch within the call's compact part buffer
  int sample_first;               // index of a pixel's first sample in this launch (0 for mirt_render; mirt_render_accumulate: any)
  int sample_count;               // samples per pixel in this launch (max(spp, 1) for mirt_render)
  int seed_per_pixel;             // 1: curand_init(1234 + pixel, sample, 0) (draw.cu:74,162); 0: curand_init(1234, pixel, 0) (draw.cu:105)
  // geometry
  const float4* nodes;            // record heap: 4 x float4 per internal node, then the primitive records in sorted order:
                                  // sphere (cx,cy,cz,r); triangle 3 x float4: p0.xyz nor.x | nor.yz e1.xy | e1.z e2.xyz
  const uint32_t* unit_prim;      // per 16-byte unit of the primitive region: type << 31 | index in the scene's sphere / triangle array
  const float4* mats;             // 3 x float4 per primitive: spheres first, then triangles
  uint32_t root_ref;
  uint32_t prim_base16;           // offset of the primitive region in the heap, in 16-byte units
  uint32_t swap_mask;             // NODE_SWAP_* bits that allow near-child-first descent (0: the reference's left-first order)
  int reach_check;                // the walk is not the reference's own (order or boxes): sphere hits are vetted before they are shaded (hit_needs_literal_walk)
  float reach_slack;              // 2^-21 x the largest coordinate magnitude of the scene box
  int skip_unlit;                 // 1: shadow rays towards lights the shading normal faces away from are not traced (all colours finite)
  int shadow_anyhit;              // 1: a shadow ray ends at its first occluder; 0: nearest-hit query like every other ray (draw.cu:347-352)
  const float* qparams;           // quantised nodes in use: grid origin xyz, grid step xyz, 2^60 / grid step xyz (else null)
  const float4* tri_boxes;        // 2 x float4 per triangle (scene order): its exact leaf box (xmin, xmax, ymin, ymax) (zmin, zmax, -, -)
  int num_spheres;
  int num_prims;
  const PlaneDev* planes; int num_planes;
  const LightDev* suns; int num_suns;
  const LightDev* bulbs; int num_bulbs;
  // rng
  RngTablesDev rng;
  int needs_rng;
  // outputs / workspace
  float4* samples;                // one RGBA per sample
  uint32_t* stack_spill;          // [STACK_TOTAL - STACK_LDS][grid threads]
  float* pending;                 // [pending_slots][16 words][grid threads] or null
  int pending_slots;
  int refill_k;                   // leave the traversal loop when this many lanes wait to shade
  int lds_depth;                  // traversal-stack entries kept in LDS (<= the compiled STACK_LDS); deeper ones spill
  int drain_lanes;                // a wave with at most this many live lanes and nothing left to fetch stops batching
  int batch_k;                    // start the next rays of ray batches when this many lanes wait for one
  int init_k;                     // start new samples when this many lanes wait for one (or the wave has nothing else to do)
  unsigned long long* counters;   // MirtStats head (8 x u64) or null
  unsigned long long* overflow;   // never null: capacity overflows (traversal stack beyond 64, pending list), must stay 0
  unsigned long long* work_counter; // next unclaimed chunk of the frame (single-kernel path)
  const uint32_t* chunk_order;      // chunk k of the hand-out order is chunk chunk_order[k] of the frame (null: identity)
  int chunk_shift;                  // log2 of the number of consecutive samples a wave takes from the frame per atomic
  uint32_t* chunk_cost;             // per chunk: the largest number of traversal steps one of its samples took
  const uint32_t* sample_order;     // sched = 2: position k of the hand-out is sample sample_order[k] of the launch (null: identity)
  uint32_t* sample_key;             // sched = 2, measuring frame: per sample 255 - its cost class (sorted ascending: expensive first)
};

// What the traversal loop of the single-kernel path touches, passed by value (scalar registers).  Everything else is read
// through a pointer to the RenderArgs in device memory, and only inside the shade phase, so that it does not occupy
// scalar registers across the hot loop.
struct HotArgs {
  const float4* nodes;            // start of the record heap
  uint32_t root_ref;
  uint32_t swap_mask;
  const float* qparams;
  const PlaneDev* planes; int num_planes;
  const LightDev* suns; int num_suns;
  const LightDev* bulbs; int num_bulbs;
  int shadow_anyhit;
  uint32_t* stack_spill;
  int lds_depth, refill_k, batch_k, drain_lanes;
  int reps;                       // traversal steps per pass through the loop header
  int leaf_k;                     // primitive tests are held back until this many lanes of the wave have one pending
  int reach_check;                // RenderArgs::reach_check
};

struct ResolveArgs {
  const float4* samples;          // this launch's samples: [num_local_pixels][count]
  unsigned char* rgba8;           // the call's part buffer (nullable when accumulating)
  float4* rgba_f32;               // nullable
  float4* accum;                  // non-null: add the per-pixel sample sums here instead of writing pixels (mirt_render_accumulate)
  long long num_local_pixels;     // pixels of this launch
  long long pixel_base;           // where they start in rgba8 / rgba_f32 / accum
  int spp;                        // the call's spp (quantiser choice: draw.cu:129-132 for spp <= 1, :9-11,202-205 otherwise)
  int count;                      // samples per pixel in `samples`
};

// device copies of the skip-ahead tables, keyed by spp (spp > 1) or by -(frame pixels) (spp <= 1)
struct RngCache {
  RngTables host;
  long long key = -1;
  uint4* A = nullptr; uint32_t* B = nullptr; uint32_t* K = nullptr; uint32_t* R2 = nullptr;
};

// Mode switches and tuning values of a scene.  The defaults are the measured optima.  MIRT_<NAME> environment variables
// override them ONCE, when the scene is created (tools/ sweeps); mirt_scene_set_option changes them afterwards.  Nothing
// reads the environment during a render.
struct Options {
  int bounds_as_shipped = 0;   // build: 1 = scene bounds never stored, every Morton code 0 -- the tree of the shipped reference (parse.cpp:28)
  int traversal = 1;           // MIRT_TRAVERSAL_*: 0 reference (left first), 1 ordered where pixels cannot change, 2 ordered everywhere
  int wavefront = 0;           // 1: the trace / shade kernel pair instead of the single kernel
  int stack_lds_depth = -1;    // traversal-stack entries kept in LDS (-1: the compiled size); tests force the spill path with it
  int refill_k = 0;            // leave the traversal loop when this many lanes wait to shade; 0 = by kind of kernel (render.hip)
  int batch_k = 8, drain_lanes = 16;
  int leaf_k = 0;              // primitive tests are held back until this many lanes have one pending; 0 = by kind of kernel (8; exact records 4)
  int reps = 0;                // traversal steps per pass through the loop header; 0 = by kind of kernel (4; wide records 5)
  int init_k = 0;              // lanes without a sample are refilled once this many wait (1: at every shade phase); 0 = by kind of kernel (render.hip)
  int chunk_shift = 0;         // 0: by frame size
  int trace_waves = 0;         // 0: fill the device
  int shadow_anyhit = 1;       // 0: shadow rays are nearest-hit queries, as in diffuseLight (draw.cu:347-352, 365-370): the reference's walk, more node visits
  int skip_unlit = 1;          // 0: shadow rays towards lights the shading normal faces away from are traced as well (draw.cu:342-374 traces them all)
  int qnodes = 1;              // quantised node records in the single-kernel path: 0 never; 1 sphere-only scenes (traversal >= 1) and scenes with
                               // triangles of 65536 primitives or more (wide records, traversal = 1); 2 every scene
  int specialise = 1;          // kernels compiled without what the scene does not have: point lights; transparency and gi (SPEC_*, shade_common.h)
  int sched = 2;               // longest-first hand-out measured on the first call of a shape: 2 by sample (stable within a cost class), 1 by chunk (one-slab calls), 0 off
  int slab_log2 = 28;          // a call is rendered in slabs of at most 2^slab_log2 samples (4 GiB of per-sample workspace; 2^26: +1.8 % on config 5)
  int wf_pool = 1 << 21, wf_refill_k = 16;
};

constexpr int MIRT_MAX_FRAMES = 4;
// everything one frame in flight owns
struct RenderCtx {
  float4* samples = nullptr; size_t samples_cap = 0;
  uint32_t* stack_spill = nullptr; size_t spill_cap = 0;
  float* pending = nullptr; size_t pending_cap = 0;
  unsigned long long* counters = nullptr;  // device: [0..7] MirtStats counters, [8] work counter, [9] overflow events, [10] scratch (mirt_get_stats), [11] rays traversed, [12] waves past the end of the work
  hipStream_t stream = nullptr;            // the stream this context's latest frame was issued on
  RenderArgs* args_dev = nullptr;          // this frame's RenderArgs in device memory (a ring of slots, one per slab in flight)
  RenderArgs args_host[4];                 // what each slot holds
  bool args_valid[4] = {false, false, false, false};
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;   // render start / first trace start / last trace end / render end
  std::vector<hipEvent_t> slab_ev;         // start / end of every trace launch of the last call beyond the first (ev1 / ev2 serve a one-slab call)
  int launches = 0;                        // trace launches of the last call
  int node_bytes = 64;                     // record size of the walk the last call used
  bool used = false, counted = false, timed = true;
  // longest-first scheduling: this frame's per-chunk cost, and the hand-out orders computed from it (two buffers used in
  // turn, so that a frame still reading an order never sees it rewritten)
  uint32_t* chunk_cost = nullptr;
  // chunk orders this context produced, used round-robin.  Three of them: the one written at use u is next written at use
  // u + 3, i.e. 12 frames later, and by then the host has waited (on context reuse) for every frame that could read it --
  // so no frame ever needs a device-side wait on another frame's stream.
  static constexpr unsigned ORDER_BUFS = 3;
  uint32_t* order_out[ORDER_BUFS] = {nullptr, nullptr, nullptr};
  size_t chunk_cap = 0;
  unsigned uses = 0;
  unsigned long long frame_id = 0;         // sequence number of the frame that last used this context
  long long order_key = -1;                // frame size (and chunk size) the order in order_out[(order_writes - 1) % ORDER_BUFS] was measured on
  unsigned order_writes = 0;               // orders this context has produced
  unsigned long long order_frame = 0;      // frame_id of the frame that measured it
  hipEvent_t order_ev = nullptr;           // ... and the end of its sort
  float wf_trace_ms = -1.0f;               // >= 0: the wavefront path ran; summed trace-kernel time
};

} // namespace mirt

struct MirtScene {
  int device = 0;
  MirtSceneDesc d{};
  int N = 0, Ns = 0, Nt = 0;
  mirt::Options opt;
  int grid_blocks = 0, wf_trace_blocks = 0;   // persistent-grid sizes for this scene's device (filled on first use)
  // uploaded geometry, file order (inputs of the build)
  float4* spheres = nullptr;            // (cx, cy, cz, r)
  float4* tris = nullptr;               // 3 x float4 per triangle: p0.xyz nor.x | nor.yz e1.xy | e1.z e2.xyz
  float4* mats = nullptr;
  MirtPrimRef* refs_in = nullptr;       // file order
  float4* tri_verts = nullptr;          // 3 x float4 per triangle (p0,p1,p2) -- build only
  mirt::PlaneDev* planes = nullptr;
  mirt::LightDev* suns = nullptr;
  mirt::LightDev* bulbs = nullptr;
  // build products
  uint32_t* codes = nullptr;            // sorted Morton codes [N]
  uint32_t* order = nullptr;            // sorted position -> file-order index [N]
  uint32_t* child_l = nullptr;          // reference numbering [N-1]
  uint32_t* child_r = nullptr;
  int* parent = nullptr;                // [2N-1]
  float* boxes = nullptr;               // [2N-1][6] xmin,xmax,ymin,ymax,zmin,zmax
  float4* nodes = nullptr;              // packed [N-1][4]; start of the record heap [nodes | primitive records, sorted order | pad]
  unsigned char* heap = nullptr;
  uint32_t prim_base = 0;               // byte offset of the primitive region in the heap
  uint32_t* unit_prim = nullptr;        // [Ns + 3 Nt]: per 16-byte unit of the primitive region, type << 31 | index (first unit of a record)
  uint32_t* tris_before = nullptr;      // [N + 1]: triangles among sorted leaves [0, j)
  uint2* range = nullptr;               // [N - 1]: sorted-leaf range (first, last) of every internal node
  uint32_t qnode_base = 0;              // byte offset of the quantised node records in the heap (0: none built)
  uint32_t wnode_base = 0;              // byte offset of the wide quantised records (scenes with triangles; 0: none built)
  uint32_t root_ref_w = mirt::REF_NONE; // root reference into the wide records
  uint32_t root_ref_q = mirt::REF_NONE; // root reference into the quantised records
  float* qparams = nullptr;             // [9] device: grid origin, grid step, 2^60 / grid step
  float4* tri_boxes = nullptr;          // [2 Nt]: exact leaf box of every triangle (scene order), for the quantised walk's triangle check
  uint32_t* build_ws = nullptr; size_t build_ws_words = 0;   // LBVH build workspace (sort buffers, histograms, arrival counters)
  uint32_t* bounds_keys = nullptr;      // [6] ordered-uint min xyz, max xyz
  uint32_t root_ref = mirt::REF_NONE;
  bool built = false;
  float build_ms = 0.0f;
  float coord_max = 0.0f;               // largest |coordinate| of the scene box (set by the build)
  bool grid_ok = false;                 // the grid of the quantised records resolves the scene's coordinates (set by the build): they may be walked
  // render workspaces: MIRT_MAX_FRAMES contexts so that several frames can be in flight on different streams (the next frame's blocks fill the
  // CUs the draining frame frees); a context is reused only after its previous frame has finished
  mirt::RenderCtx ctx[mirt::MIRT_MAX_FRAMES];
  unsigned frame_no = 0;
  unsigned long long frame_seq = 0;
  mirt::RenderCtx* last = nullptr;        // context of the most recent render (mirt_get_stats)
  unsigned long long overflow_events = 0;  // capacity overflows seen since the last mirt_get_stats
  double trace_ms_sum = 0.0; int trace_frames = 0;   // trace-kernel time of the frames finished since the last mirt_get_stats
  // wavefront path workspace (wavefront.hip)
  uint32_t* wf_state = nullptr; size_t wf_state_cap = 0;
  float4* wf_rays = nullptr; size_t wf_rays_cap = 0;
  unsigned long long* wf_ctr = nullptr;
  unsigned long long* wf_ctr_host = nullptr;
  std::vector<hipEvent_t> wf_events;
  int wf_rounds = 0;
  // sched = 2: the frame's samples in order of decreasing cost class, measured once per frame size (shared by the contexts)
  uint32_t* so_order = nullptr; uint32_t* so_keys = nullptr; uint32_t* so_keys2 = nullptr; uint32_t* so_ws = nullptr;
  size_t so_cap = 0, so_slab_cap = 0; long long so_key = -1, so_pending_key = -1, so_total = -1; hipEvent_t so_ev = nullptr; bool so_busy = false;
  // rng tables cache
  mirt::RngCache rng;
  // LBVH build timing
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool colors_finite = true;               // every material colour and light colour is finite (0 * colour == 0)
  bool any_trans = false;                  // some material has transparency != 0
  bool any_rough = false;
};

namespace mirt {
int hip_fail(hipError_t e, const char* what, const char* file, int line);
// lbvh_build.hip
int build_lbvh(MirtScene* sc, hipStream_t stream);
size_t sort_low_byte_ws_words(long long n);
int sort_low_byte(const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, long long n, uint32_t* ws, hipStream_t stream);
int get_tree(MirtScene* sc, MirtTreeNode* nodes, uint32_t* codes, MirtPrimRef* refs, float* bounds);
// render.hip
int render(MirtScene* sc, const MirtRenderParams* p, void* d_rgba8, void* d_rgba_f32, hipStream_t stream);
int scatter_part(const MirtRenderParams* p, const void* d_part, void* d_frame, hipStream_t stream);
int64_t render_num_pixels(const MirtRenderParams* p);
int part_pixel(const MirtRenderParams* p, int64_t local, int32_t* x, int32_t* y);
int trace_ms_of(RenderCtx& cx, float* ms);
int probe_math(int device, int which, int n, const float* in, float* out);
int probe_xorwow(int device, int spp, int nstreams, int draws, uint32_t* out);
int ensure_rng_tables(RngCache* rc, int sample_tables, long long frame_pixels, hipStream_t stream, RngTablesDev* out, bool allow_larger);
int render_accumulate(MirtScene* sc, const MirtRenderParams* p, void* d_accum, int sample_first, int sample_count, hipStream_t stream);
int finalize(const MirtRenderParams* p, const void* d_accum, int total_samples, void* d_rgba8, hipStream_t stream);
void rng_cache_free(RngCache* rc);
// wavefront.hip
int wavefront_trace(MirtScene* sc, RenderCtx& cx, RenderArgs& a, bool count, hipStream_t stream, float* trace_ms);
}
#define MIRT_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return mirt::hip_fail(e_, #call, __FILE__, __LINE__); } while (0)

#endif
