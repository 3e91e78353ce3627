/*
 * oracle/oracle.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * A single-threaded CPU restatement of the hot path of GJ0407790/cuda_ray_tracer
 * (LBVH build: Morton -> stable sort -> Karras -> refit; render: camera ray, BVH stack walk,
 * sphere/triangle/plane intersection, the recursive shading tree, sample reduction, sRGB and
 * quantisation), written from the reference's algorithm with the reference file:line cited at each
 * function.  It exists to check the HIP product; it is NOT part of the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (cuda_ray_tracer_amd/) never includes, links or calls anything in this directory.
 *
 * Pinning: the reference has no tests and cannot be built here (it needs nvcc, cuda_runtime.h,
 * curand_kernel.h, Thrust and png.h, none of which are in the image; writing stand-ins is not allowed).
 * The oracle is therefore pinned only by the RNG-free known answers SURVEY.md section 8c recorded from
 * the reference's own code (tri.txt 256x256 aa 0: SHA-256 / byte sum / pixel values, the N=5 node
 * dump, the traversal statistics of Appendix G); see tests/test_oracle_golden.py.  Everything that
 * consumes random numbers is PARITY UNPINNED against a CUDA run: cuRAND's XORWOW constants are
 * restated from its published algorithm and CUDA's libm is replaced by omath.h.
 *
 * Arithmetic: fp32 with one rounding per source operation (compile with -ffp-contract=off), which is
 * what the reference source says literally; transcendental functions come from omath.h.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <vector>
#include <algorithm>
#include <numeric>

#include "omath.h"

#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {

/* ---- POD mirrors of the reference structs (object.cuh:17-38,95-119,124-149,165-194,232-266) ---- */
typedef struct { float r, g, b; } ORGB;
typedef struct { float x, y, z; } OV3;
typedef struct { ORGB color, shininess, trans; float ior, roughness; } OMat;   /* 44 B */
typedef struct { OV3 c; float r; OMat mat; } OSphere;                           /* 60 B */
typedef struct { OV3 p0, p1, p2, nor, e1, e2; OMat mat; } OTriangle;            /* 116 B */
typedef struct { float a, b, c, d; OV3 nor, point; OMat mat; } OPlane;          /* 84 B */
typedef struct { OV3 dir; ORGB color; } OSun;                                   /* 24 B */
typedef struct { OV3 point; ORGB color; } OBulb;                                /* 24 B */
typedef struct { uint32_t type, id; } OPrimRef;                                 /* 8 B; type 0 sphere, 1 triangle */

typedef struct {
  int32_t width, height, bounces, aa;
  float dof_focus, dof_lens;
  OV3 forward, right, up, eye;
  float expose;
  int32_t fisheye, panorama, gi;
  int32_t num_spheres, num_triangles, num_prims, num_planes, num_suns, num_bulbs;
  const OSphere* spheres;
  const OTriangle* triangles;
  const OPrimRef* prim_refs;
  const OPlane* planes;
  const OSun* suns;
  const OBulb* bulbs;
} OSceneDesc;

/* lbvh.cuh:6-28, flattened (visited counter dropped) */
typedef struct {
  float xmin, xmax, ymin, ymax, zmin, zmax;
  uint32_t left, right, prim_offset, count;
} ONode;

typedef struct {
  uint64_t samples, rays, shadow_rays, node_iters, internal_visits, sphere_tests, tri_tests,
           mat_fetches, max_stack, prim_hits, overflow, qn_retraces, traversals;
} OStats;

typedef struct {
  float t;          /* distance, or -1 when nothing was hit */
  uint32_t kind;    /* 0 none, 1 sphere, 2 triangle, 3 plane */
  uint32_t id;      /* index in its array */
  float nx, ny, nz;
} OHit;

} /* extern "C" */

#define ORC_FLAG_ANYHIT_SHADOW 1u   /* shadow rays stop at the first occluder (same boolean as draw.cu:347-352,365-370) */
#define ORC_FLAG_NORMAL_ZYX    2u   /* evaluate the three standerdD() arguments right-to-left (draw.cu:335-337 is unspecified) */
/* Ordered traversal (the product's default; NOT in the reference, which always descends left first,
 * bvh_traversal.cu:149-157).  Where both children of a node are hit and both subtrees hold spheres only, the child
 * whose box the ray enters first is descended first; exact ties in the hit distance go to the primitive with the
 * smaller sorted index, which is the one the left-first walk meets first.  For spheres the closest hit does not depend
 * on the visiting order (a sphere is only ever hit inside its box), so pixels are those of the reference order and only
 * the visit counters change.  Triangles are different: the reference accepts hits up to 0.001 (barycentric) outside a
 * triangle, i.e. possibly outside its box, and whether such a hit is found depends on the order -- subtrees that hold
 * a triangle are therefore walked in the reference order unless ORDERED_ALL is given. */
/* Shadow rays towards lights the shading normal faces away from are answered without a traversal (the product always
 * does this when every colour is finite): the light's term is colour * light * max(dot, 0) = 0 whether it is occluded
 * or not (draw.cu:353-357, 371-374).  The ray still counts as a ray; it visits no node. */
#define ORC_FLAG_SKIP_UNLIT    16u
/* Quantised node records (the product's single-kernel path, scene_dev.h): child boxes on the 65535-step grid of the scene
 * bounds, rounded outwards, tested with one fused multiply-add per plane; a sphere hit then has to pass the
 * `t_enter < t_exit && t_exit > t_min` clauses of its exact leaf box (bvh_traversal.cu:43), which the larger boxes weaken.
 * Same closest hit as the exact boxes; more node visits.  Only meaningful with ORDERED / ORDERED_ALL.
 * Triangles: the reference accepts hits up to 0.001 (barycentric) outside a triangle, possibly outside its box, and then
 * whether the hit is found depends on which exact boxes culled the way to it.  A triangle hit that is about to be accepted
 * is therefore checked against its exact leaf box (bvh_traversal.cu:11-44): if the box passes its order-independent clauses
 * and the ray enters it before the hit (t_enter < t), every exact ancestor box -- each contains the leaf box, and the slab
 * arithmetic is monotone in the box planes -- passes too with the best distance of its time (t_enter_ancestor <= t_enter <
 * t <= t_max then), so the reference's walk reaches this leaf and finds the same hit.  Otherwise the ray is walked AGAIN from
 * the root over the exact boxes (same order flags): rare (triangle silhouettes), exact. */
#define ORC_FLAG_QNODES        32u
/* Wide walk over the quantised records (the product's records for scenes with triangles, scene_dev.h): a step at internal node P
 * tests the boxes of P's GRANDchildren (a child that is a leaf stands for itself) -- up to four, in the reference's order
 * (left subtree first) -- and descends into the first one hit, pushing the others.  The reference tests the children of R only
 * when R is popped, i.e. against a best distance that may have shrunk meanwhile, and it tests L and R themselves; this walk
 * tests them early and skips L and R (their boxes contain their children's): it visits a superset of the reference's leaves in
 * the same order, which is all the argument under ORC_FLAG_QNODES needs.  Half as many dependent steps per ray.  Needs QNODES;
 * the descent order is the reference's at every node (no near-child-first, not even between sphere-only subtrees). */
#define ORC_FLAG_WIDE          64u
/* A walk over the quantised boxes (ORC_FLAG_QNODES; the exact boxes are walked in the reference's order) vets the sphere hit a
 * nearest-hit query ends with before it is shaded (the product: hit_needs_literal_walk, shade_common.h): a sphere's hit distance can round to just below the
 * entry distance of its own leaf box -- a ray that touches the sphere where the sphere touches its box, from far away -- and
 * then the reference tests that sphere or not depending on what it found before, while another walk may end with it.  If the
 * exact leaf box passes its order-independent clauses and is entered before the hit, the reference provably reaches the leaf
 * (the argument under ORC_FLAG_QNODES); otherwise the ray is walked again by traverse(), literally.  Shadow queries to suns
 * are not vetted (occluded iff anything is hit, in any order); those to point lights -- "occluded" = nearer than the light -- are
 * traced to their nearest hit and vetted the same way (occluded()).  Not set for the wavefront path. */
#define ORC_FLAG_REACH         128u
#define ORC_FLAG_ORDERED       4u
#define ORC_FLAG_ORDERED_ALL   8u   /* near child first at every node (triangle silhouettes may differ from the reference order) */

/* ------------------------------------------------------------------------------------------------ */
/* vec3 / RGB (vec3.cuh:7-107, struct.cuh:11-62)                                                     */
/* ------------------------------------------------------------------------------------------------ */
struct V3 { float x, y, z; };
static inline V3 mk(float a, float b, float c) { V3 v; v.x = a; v.y = b; v.z = c; return v; }
static inline V3 mk(const OV3& o) { return mk(o.x, o.y, o.z); }
static inline V3 operator+(const V3& a, const V3& b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(const V3& a, const V3& b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(const V3& a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(float s, const V3& a) { return mk(a.x * s, a.y * s, a.z * s); }
static inline V3 operator/(const V3& a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
static inline V3 operator-(const V3& a) { return mk(-a.x, -a.y, -a.z); }
static inline float dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(const V3& a, const V3& b)
{
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vec3.cuh:7-18 */
static inline bool fequal(float a, float b, float epsilon = 1e-6f)
{
  float diff = fabsf(a - b);
  float largest = fmaxf(fabsf(a), fabsf(b));
  if (largest < 1e-6f) return diff < epsilon;
  return diff / largest < epsilon;
}
static inline float length(const V3& v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
/* vec3.cuh:72-82 */
static inline V3 normalize(const V3& v)
{
  float mag = length(v);
  if (fequal(mag, 0.0f)) return mk(0.0f, 0.0f, 0.0f);
  float inv_mag = 1.0f / mag;
  return mk(v.x * inv_mag, v.y * inv_mag, v.z * inv_mag);
}

struct C3 { float r, g, b; };                 /* RGB  struct.cuh:11-34 */
struct C4 { float r, g, b, a; };              /* RGBA struct.cuh:37-62 */
static inline C3 c3(float r, float g, float b) { C3 c; c.r = r; c.g = g; c.b = b; return c; }
static inline C3 c3(const ORGB& o) { return c3(o.r, o.g, o.b); }
static inline C4 c4(float r, float g, float b, float a) { C4 c; c.r = r; c.g = g; c.b = b; c.a = a; return c; }
static inline C4 c4zero() { return c4(0.0f, 0.0f, 0.0f, 0.0f); }
static inline bool ceq(const C3& a, const C3& b) { return fequal(a.r, b.r) && fequal(a.g, b.g) && fequal(a.b, b.b); }
static inline C3 operator-(const C3& a, const C3& b) { return c3(a.r - b.r, a.g - b.g, a.b - b.b); }
static inline C3 operator*(const C3& a, const C3& b) { return c3(a.r * b.r, a.g * b.g, a.b * b.b); }
static inline C4 operator+(const C4& a, const C4& b) { return c4(a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a); }
/* struct.cuh:52-55: rgb * RGBA keeps the RGBA's alpha */
static inline C4 operator*(const C3& k, const C4& o) { return c4(k.r * o.r, k.g * o.g, k.b * o.b, o.a); }

/* ------------------------------------------------------------------------------------------------ */
/* cuRAND-compatible XORWOW (curand_init / curand / curand_uniform / curand_normal) */
/* ------------------------------------------------------------------------------------------------ */
struct Rng {
  uint32_t v[5];
  uint32_t d;
  int bm_flag;
  float bm_extra;
  /* lazy skip-ahead: applied on first draw */
  bool pending;
  uint64_t subseq;
};

static inline uint32_t xw_step_v(uint32_t* v)
{
  uint32_t t = v[0] ^ (v[0] >> 2);
  v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
  v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
  return v[4];
}

/* 160x160 GF(2) matrix, column-major: col[i] = image of basis vector i (5 words) */
struct M160 { uint32_t col[160][5]; };

static void m160_apply(const M160& m, const uint32_t* in, uint32_t* out)
{
  uint32_t acc[5] = {0, 0, 0, 0, 0};
  for (int w = 0; w < 5; ++w) {
    uint32_t bits = in[w];
    while (bits) {
      int b = __builtin_ctz(bits);
      bits &= bits - 1;
      const uint32_t* c = m.col[w * 32 + b];
      acc[0] ^= c[0]; acc[1] ^= c[1]; acc[2] ^= c[2]; acc[3] ^= c[3]; acc[4] ^= c[4];
    }
  }
  memcpy(out, acc, sizeof(acc));
}
static void m160_mul(const M160& a, const M160& b, M160& out) /* out = a * b (apply b first) */
{
  M160 tmp;
  for (int i = 0; i < 160; ++i) m160_apply(a, b.col[i], tmp.col[i]);
  out = tmp;
}

struct JumpTables {
  M160 pow2[64];      /* (step^(2^67))^(2^k), k = 0..63 */
  bool ready;
};
static JumpTables g_jump = {{}, false};

static void jump_init()
{
  if (g_jump.ready) return;
  M160 a;
  for (int i = 0; i < 160; ++i) {
    uint32_t v[5] = {0, 0, 0, 0, 0};
    v[i / 32] = 1u << (i % 32);
    xw_step_v(v);
    memcpy(a.col[i], v, sizeof(v));
  }
  for (int k = 0; k < 67; ++k) m160_mul(a, a, a);   /* a = step^(2^67): skipahead_sequence(1) */
  g_jump.pow2[0] = a;
  for (int k = 1; k < 64; ++k) m160_mul(g_jump.pow2[k - 1], g_jump.pow2[k - 1], g_jump.pow2[k]);
  g_jump.ready = true;
}

static void rng_apply_subseq(Rng* s)
{
  uint64_t n = s->subseq;
  for (int k = 0; n; ++k, n >>= 1)
    if (n & 1) m160_apply(g_jump.pow2[k], s->v, s->v);
  s->pending = false;
}

/* curand_init(seed, subsequence, offset, &state) */
static void rng_init(Rng* s, uint64_t seed, uint64_t subseq, uint64_t offset)
{
  uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
  uint32_t s1 = ((uint32_t)(seed >> 32)) ^ 0xf7dcefddu;
  uint32_t t0 = 1099087573u * s0;
  uint32_t t1 = 2591861531u * s1;
  s->d = 6615241u + t1 + t0;
  s->v[0] = 123456789u + t0;
  s->v[1] = 362436069u ^ t0;
  s->v[2] = 521288629u + t1;
  s->v[3] = 88675123u ^ t1;
  s->v[4] = 5783321u + t0;
  s->bm_flag = 0;
  s->bm_extra = 0.0f;
  s->subseq = subseq;
  s->pending = subseq != 0;
  if (offset) { /* not used by the reference (always 0); plain stepping */
    if (s->pending) rng_apply_subseq(s);
    for (uint64_t i = 0; i < offset; ++i) { xw_step_v(s->v); s->d += 362437u; }
  }
}
static inline uint32_t rng_next(Rng* s)
{
  if (s->pending) rng_apply_subseq(s);
  uint32_t x = xw_step_v(s->v);
  s->d += 362437u;
  return x + s->d;
}
/* curand_uniform: (0,1] */
static inline float rng_uniform(Rng* s)
{
  return (float)rng_next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
/* curand_normal: Box-Muller on two draws, second value cached */
static inline float rng_normal(Rng* s)
{
  if (s->bm_flag) { s->bm_flag = 0; return s->bm_extra; }
  uint32_t x = rng_next(s);
  uint32_t y = rng_next(s);
  float u = (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
  float v = (float)y * 1.46291807926e-9f + (1.46291807926e-9f / 2.0f);
  float sq = sqrtf(-2.0f * o_logf(u));
  double sn, cs;
  o_sincos((double)v, &sn, &cs);
  s->bm_extra = sq * (float)cs;
  s->bm_flag = 1;
  return sq * (float)sn;
}
/* helper.cu:82-89 */
static inline float randD(float start, float end, Rng* s) { float u = rng_uniform(s); return start + (end - start) * u; }
static inline float standerdD(float stddev, Rng* s) { return rng_normal(s) * stddev; }

/* ------------------------------------------------------------------------------------------------ */
/* scene                                                                                             */
/* ------------------------------------------------------------------------------------------------ */
struct Scene {
  OSceneDesc d;
  std::vector<OSphere> spheres;
  std::vector<OTriangle> tris;
  std::vector<OPrimRef> refs;       /* sorted in place by the build, like d_primitive_references */
  std::vector<OPlane> planes;
  std::vector<OSun> suns;
  std::vector<OBulb> bulbs;
  std::vector<uint32_t> codes;      /* sorted morton codes */
  std::vector<ONode> nodes;         /* internal [0,N-2], leaves [N-1,2N-2] */
  float smin[3], smax[3];
  bool built;
  /* ordered traversal: which subtrees hold spheres only */
  std::vector<int> rfirst, rlast;   /* sorted-leaf range of every internal node (determine_range) */
  std::vector<uint32_t> tris_before; /* [N+1]: number of triangles among sorted leaves [0, j) */
  bool pure(int first, int last) const { return tris_before[(size_t)last + 1] == tris_before[(size_t)first]; }
};

struct AABB { float xmin, xmax, ymin, ymax, zmin, zmax; };

/* interval.cuh:55-59: AABB(point a, point b) */
static AABB aabb2(const V3& a, const V3& b)
{
  AABB r;
  if (a.x <= b.x) { r.xmin = a.x; r.xmax = b.x; } else { r.xmin = b.x; r.xmax = a.x; }
  if (a.y <= b.y) { r.ymin = a.y; r.ymax = b.y; } else { r.ymin = b.y; r.ymax = a.y; }
  if (a.z <= b.z) { r.zmin = a.z; r.zmax = b.z; } else { r.zmin = b.z; r.zmax = a.z; }
  return r;
}
/* interval.cuh:61-81: AABB(a,b,c) with the +-0.01 pad of thin axes */
static AABB aabb3(const V3& a, const V3& b, const V3& c)
{
  AABB r;
  r.xmin = fminf(fminf(a.x, b.x), c.x); r.xmax = fmaxf(fmaxf(a.x, b.x), c.x);
  r.ymin = fminf(fminf(a.y, b.y), c.y); r.ymax = fmaxf(fmaxf(a.y, b.y), c.y);
  r.zmin = fminf(fminf(a.z, b.z), c.z); r.zmax = fmaxf(fmaxf(a.z, b.z), c.z);
  if (r.xmax - r.xmin < 0.01f) { r.xmin = r.xmin - 0.01f; r.xmax = r.xmax + 0.01f; }
  if (r.ymax - r.ymin < 0.01f) { r.ymin = r.ymin - 0.01f; r.ymax = r.ymax + 0.01f; }
  if (r.zmax - r.zmin < 0.01f) { r.zmin = r.zmin - 0.01f; r.zmax = r.zmax + 0.01f; }
  return r;
}
/* lbvh_builder.cu:33-57 */
static AABB prim_aabb(const Scene& sc, const OPrimRef& ref)
{
  if (ref.type == 0) {
    V3 c = mk(sc.spheres[ref.id].c);
    float r = sc.spheres[ref.id].r;
    V3 rv = mk(r, r, r);
    return aabb2(c - rv, c + rv);
  }
  const OTriangle& t = sc.tris[ref.id];
  return aabb3(mk(t.p0), mk(t.p1), mk(t.p2));
}

/* lbvh_utils.cu:10-30 */
static inline uint32_t expand_bits(uint32_t v)
{
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
static inline uint32_t morton_3d(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
static inline uint32_t quantize_coordinate(float coord, float smin, float range, int bits)
{
  if (range <= 1e-6f) return 0;
  float normalized = (coord - smin) / range;
  normalized = fmaxf(0.0f, fminf(1.0f, normalized));
  return (uint32_t)(normalized * ((1 << bits) - 1));
}

static inline int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }

/* lbvh_builder.cu:76-101 */
static inline int adapted_delta(int a, int b, uint32_t n, const uint32_t* codes)
{
  bool inv_a = (a < 0 || a >= (int)n);
  bool inv_b = (b < 0 || b >= (int)n);
  if (inv_a || inv_b) return -1;
  uint32_t ka = codes[a], kb = codes[b];
  if (ka == kb) return 32 + clz32((uint32_t)a ^ (uint32_t)b);
  return clz32(ka ^ kb);
}

/* lbvh_builder.cu:103-182 */
static void determine_range(const uint32_t* codes, uint32_t n, int i, int* first, int* last)
{
  const int delta_l = adapted_delta(i, i - 1, n, codes);
  const int delta_r = adapted_delta(i, i + 1, n, codes);
  int d, delta_min;
  if (n <= 1) { *first = i; *last = i; return; }
  if (delta_r > delta_l) { d = 1; delta_min = delta_l; } else { d = -1; delta_min = delta_r; }
  uint32_t l_max = 1;
  int nb = (int)((uint32_t)i + l_max * (uint32_t)d);
  int cur = adapted_delta(i, nb, n, codes);
  while (cur > delta_min) {
    l_max <<= 1;
    nb = (int)((uint32_t)i + l_max * (uint32_t)d);
    if (nb < 0 || nb >= (int)n) break;
    cur = adapted_delta(i, nb, n, codes);
  }
  uint32_t l = 0;
  for (uint32_t t = l_max >> 1; t > 0; t >>= 1) {
    int nbb = (int)((uint32_t)i + (l + t) * (uint32_t)d);
    if (nbb >= 0 && nbb < (int)n) {
      int c = adapted_delta(i, nbb, n, codes);
      if (c > delta_min) l += t;
    }
  }
  const int j = (int)((uint32_t)i + l * (uint32_t)d);
  if (i < j) { *first = i; *last = j; } else { *first = j; *last = i; }
}

/* lbvh_builder.cu:186-221 */
static int find_split(const uint32_t* codes, int first, int last, uint32_t n)
{
  if (first == last) return first;
  const int common = adapted_delta(first, last, n, codes);
  int split = first;
  int step = last - first;
  do {
    step = (step + 1) >> 1;
    const int cand = split + step;
    if (cand < last) {
      const int sp = adapted_delta(first, cand, n, codes);
      if (sp > common) split = cand;
    }
  } while (step > 1);
  return split;
}

/* build_lbvh_karas, lbvh_builder.cu:401-521 + lbvh_utils.cu:77-129.
 * bounds_mode 0: scene bounds = union of primitive boxes (what parse.cpp:147-155,193-200 computes but never
 *                stores); 1: bounds as shipped (+inf/-inf, every code 0 -- reference bug #1). */
static int build(Scene& sc, int bounds_mode)
{
  const int N = (int)sc.refs.size();
  sc.nodes.clear();
  sc.codes.assign(N, 0);
  sc.built = true;
  if (N == 0) return 0;

  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (bounds_mode == 0) {
    for (int i = 0; i < N; ++i) {
      AABB b = prim_aabb(sc, sc.refs[i]);
      mn[0] = fminf(mn[0], b.xmin); mx[0] = fmaxf(mx[0], b.xmax);
      mn[1] = fminf(mn[1], b.ymin); mx[1] = fmaxf(mx[1], b.ymax);
      mn[2] = fminf(mn[2], b.zmin); mx[2] = fmaxf(mx[2], b.zmax);
    }
  }
  memcpy(sc.smin, mn, sizeof(mn)); memcpy(sc.smax, mx, sizeof(mx));

  /* generate_morton_codes_kernel, lbvh_utils.cu:32-75 */
  std::vector<uint32_t> codes(N);
  for (int i = 0; i < N; ++i) {
    const OPrimRef& ref = sc.refs[i];
    V3 c;
    if (ref.type == 0) c = mk(sc.spheres[ref.id].c);
    else {
      const OTriangle& t = sc.tris[ref.id];
      c = ((mk(t.p0) + mk(t.p1)) + mk(t.p2)) / 3.0f;
    }
    float rx = mx[0] - mn[0], ry = mx[1] - mn[1], rz = mx[2] - mn[2];
    uint32_t qx = quantize_coordinate(c.x, mn[0], rx, 10);
    uint32_t qy = quantize_coordinate(c.y, mn[1], ry, 10);
    uint32_t qz = quantize_coordinate(c.z, mn[2], rz, 10);
    codes[i] = morton_3d(qx, qy, qz);
  }
  /* thrust::sort_by_key (stable), lbvh_utils.cu:110-115 */
  std::vector<int> order(N);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return codes[a] < codes[b]; });
  std::vector<OPrimRef> refs2(N);
  for (int i = 0; i < N; ++i) { refs2[i] = sc.refs[order[i]]; sc.codes[i] = codes[order[i]]; }
  sc.refs.swap(refs2);

  const int total = 2 * N - 1;
  sc.nodes.assign(total, ONode());
  sc.rfirst.assign(N > 1 ? N - 1 : 0, 0); sc.rlast.assign(N > 1 ? N - 1 : 0, 0);
  sc.tris_before.assign((size_t)N + 1, 0);
  for (int i = 0; i < N; ++i) sc.tris_before[(size_t)i + 1] = sc.tris_before[(size_t)i] + (sc.refs[i].type != 0 ? 1u : 0u);
  std::vector<int> parent(total, -1);
  const uint32_t leaf_base = (uint32_t)(N - 1);
  /* initialize_leaf_nodes_kernel, lbvh_builder.cu:59-71 */
  for (int i = 0; i < N; ++i) { sc.nodes[leaf_base + i].count = 1; sc.nodes[leaf_base + i].prim_offset = (uint32_t)i; }
  /* generate_internal_nodes_karas_kernel, lbvh_builder.cu:224-322 */
  const uint32_t* cd = sc.codes.data();
  for (int i = 0; i < N - 1; ++i) {
    int first, last;
    determine_range(cd, (uint32_t)N, i, &first, &last);
    sc.rfirst[i] = first; sc.rlast[i] = last;
    if (first > last) continue;
    const int split = find_split(cd, first, last, (uint32_t)N);
    if (split < first || split >= last) {
      sc.nodes[i].left = 0xFFFFFFFFu; sc.nodes[i].right = 0xFFFFFFFFu; sc.nodes[i].count = 0;
      continue;
    }
    uint32_t lc, rc;
    int delta_at_split = adapted_delta(split, split + 1, (uint32_t)N, cd);
    if (split == first) lc = leaf_base + (uint32_t)split;
    else {
      int dl = adapted_delta(first, split, (uint32_t)N, cd);
      lc = (dl > delta_at_split) ? (uint32_t)split : leaf_base + (uint32_t)first;
    }
    if (split + 1 == last) rc = leaf_base + (uint32_t)last;
    else {
      int dr = adapted_delta(split + 1, last, (uint32_t)N, cd);
      rc = (dr > delta_at_split) ? (uint32_t)(split + 1) : leaf_base + (uint32_t)last;
    }
    sc.nodes[i].count = 0; sc.nodes[i].left = lc; sc.nodes[i].right = rc;
    if (lc < (uint32_t)total && rc < (uint32_t)total) { parent[lc] = i; parent[rc] = i; }
    if (i == 0) parent[0] = -1;
  }
  /* set_aabb_kernel_adapted, lbvh_builder.cu:324-387 (sequential: the second arriver merges) */
  std::vector<uint32_t> visited(N > 1 ? N - 1 : 0, 0);
  for (int i = 0; i < N; ++i) {
    uint32_t cur = leaf_base + (uint32_t)i;
    AABB b = prim_aabb(sc, sc.refs[sc.nodes[cur].prim_offset]);
    ONode& ln = sc.nodes[cur];
    ln.xmin = b.xmin; ln.xmax = b.xmax; ln.ymin = b.ymin; ln.ymax = b.ymax; ln.zmin = b.zmin; ln.zmax = b.zmax;
    int p = parent[cur];
    while (p != -1 && p < N - 1) {
      uint32_t prev = visited[p]++;
      if (prev == 0) break;
      ONode& pn = sc.nodes[p];
      const ONode& a = sc.nodes[pn.left];
      const ONode& c = sc.nodes[pn.right];
      /* interval.cuh:83-88 */
      pn.xmin = fminf(a.xmin, c.xmin); pn.xmax = fmaxf(a.xmax, c.xmax);
      pn.ymin = fminf(a.ymin, c.ymin); pn.ymax = fmaxf(a.ymax, c.ymax);
      pn.zmin = fminf(a.zmin, c.zmin); pn.zmax = fmaxf(a.zmax, c.zmax);
      cur = (uint32_t)p;
      p = parent[cur];
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* rays and intersection                                                                             */
/* ------------------------------------------------------------------------------------------------ */
struct Ray { V3 eye, dir; int bounce; };
static inline Ray mkray(const V3& eye, const V3& dir, int bounce) /* object.cuh:69: normalises dir */
{
  Ray r; r.eye = eye; r.dir = normalize(dir); r.bounce = bounce; return r;
}

struct Mat { C3 color, shininess, trans; float ior, roughness; };
static inline Mat mat_default() { Mat m; m.color = c3(0, 0, 0); m.shininess = c3(0, 0, 0); m.trans = c3(0, 0, 0); m.ior = 1.458f; m.roughness = 0.0f; return m; }
static inline Mat mat_from(const OMat& o) { Mat m; m.color = c3(o.color); m.shininess = c3(o.shininess); m.trans = c3(o.trans); m.ior = o.ior; m.roughness = o.roughness; return m; }

/* object.cuh:44-56 */
struct Obj {
  bool isHit; float distance; V3 i_point, normal; Mat mat;
  uint32_t kind, id; /* bookkeeping only: 1 sphere, 2 triangle, 3 plane */
  uint32_t leaf;     /* bookkeeping: sorted position of the primitive (its leaf is node N - 1 + leaf) */
  bool literal;      /* bookkeeping: the result of a second, literal walk (ORC_FLAG_QNODES / ORC_FLAG_REACH) */
};
static inline Obj obj_none() { Obj o; o.isHit = false; o.distance = -1.0f; o.i_point = mk(0, 0, 0); o.normal = mk(0, 0, 0); o.mat = mat_default(); o.kind = 0; o.id = 0; o.leaf = 0; o.literal = false; return o; }
static inline Obj obj_hit(float t, const V3& p, const V3& n, const Mat& m, uint32_t kind, uint32_t id)
{ Obj o; o.isHit = true; o.distance = t; o.i_point = p; o.normal = n; o.mat = m; o.kind = kind; o.id = id; o.leaf = 0; o.literal = false; return o; }

struct Ctx {
  const Scene* sc;
  int width, height;      /* frame size used by the camera (struct.cu:17-20) */
  uint32_t flags;
  OStats st;
};

/* struct.cu:64-109 */
static Obj check_sphere(const Ctx& cx, const Ray& ray, uint32_t idx)
{
  const OSphere& s = cx.sc->spheres[idx];
  V3 c = mk(s.c);
  float r = s.r;
  V3 cr0 = c - ray.eye;
  bool inside = (dot(cr0, cr0) < r * r);
  float tc = dot(cr0, ray.dir);
  if (!inside && tc < 0.0f) return obj_none();
  V3 dv = ray.eye + (tc * ray.dir) - c;
  float d2 = dot(dv, dv);
  if (!inside && (r * r) < d2) return obj_none();
  float t_offset = sqrtf((r * r) - d2);
  float t = inside ? (tc + t_offset) : (tc - t_offset);
  V3 p = t * ray.dir + ray.eye;
  V3 nor = inside ? (c - p) : (p - c);
  nor = normalize(nor);
  return obj_hit(t, p, nor, mat_from(s.mat), 1, idx);
}

/* struct.cu:111-163 */
static Obj check_triangle(const Ctx& cx, const Ray& ray, uint32_t idx)
{
  const OTriangle& tr = cx.sc->tris[idx];
  V3 p0 = mk(tr.p0), n = mk(tr.nor);
  float denom = dot(ray.dir, n);
  if (fabsf(denom) < 1e-9f) return obj_none();
  float t = dot(p0 - ray.eye, n) / denom;
  if (t <= 0.001f) return obj_none();
  V3 p = t * ray.dir + ray.eye;
  V3 e1 = mk(tr.e1), e2 = mk(tr.e2);
  float b1 = dot(e1, p - p0);
  float b2 = dot(e2, p - p0);
  float b0 = 1.0f - b1 - b2;
  bool inside = (b0 >= -0.001f) && (b1 >= -0.001f) && (b2 >= -0.001f);
  if (!inside) return obj_none();
  V3 fn = (denom < 0.0f) ? n : -n;
  return obj_hit(t, p, fn, mat_from(tr.mat), 2, idx);
}

/* bvh_traversal.cu:11-44 */
static inline bool hit_aabb(const ONode& b, const V3& o, const V3& inv, float tmin, float tmax)
{
  float tx1 = (b.xmin - o.x) * inv.x, tx2 = (b.xmax - o.x) * inv.x;
  float tnx = fminf(tx1, tx2), tfx = fmaxf(tx1, tx2);
  float ty1 = (b.ymin - o.y) * inv.y, ty2 = (b.ymax - o.y) * inv.y;
  float tny = fminf(ty1, ty2), tfy = fmaxf(ty1, ty2);
  float tz1 = (b.zmin - o.z) * inv.z, tz2 = (b.zmax - o.z) * inv.z;
  float tnz = fminf(tz1, tz2), tfz = fmaxf(tz1, tz2);
  float t_enter = fmaxf(fmaxf(tnx, tny), tnz);
  float t_exit = fminf(fminf(tfx, tfy), tfz);
  return t_enter < t_exit && t_enter < tmax && t_exit > tmin;
}

/* traverse_lbvh, bvh_traversal.cu:92-183 (+ intersect_leaf_primitives :47-89).
 * stop_below: when >= 0, return as soon as the best hit distance is < stop_below (ORC_FLAG_ANYHIT_SHADOW only;
 * pass +inf for "any hit"). */
static Obj traverse(Ctx& cx, const Ray& ray, float initial_t_max, bool early, float stop_below)
{
  const Scene& sc = *cx.sc;
  Obj best = obj_none();
  best.distance = initial_t_max;
  float tmax = initial_t_max;
  if (sc.nodes.empty()) return best;
  V3 inv = mk(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
  const float tmin = 0.0001f;
  uint32_t stack[64];
  int sp = 0;
  uint32_t cur = 0;
  while (true) {
    const ONode& node = sc.nodes[cur];
    cx.st.node_iters++;
    if (node.count > 0) {
      const OPrimRef& ref = sc.refs[node.prim_offset];
      Obj h;
      if (ref.type == 0) { h = check_sphere(cx, ray, ref.id); cx.st.sphere_tests++; }
      else { h = check_triangle(cx, ray, ref.id); cx.st.tri_tests++; }
      if (h.isHit && h.distance > 1e-6f && h.distance < tmax) {
        tmax = h.distance;
        best = h;
        if (early && best.distance < stop_below) return best;
      }
      if (sp == 0) break;
      cur = stack[--sp];
      continue;
    }
    cx.st.internal_visits++;
    uint32_t l = node.left, r = node.right;
    bool hl = hit_aabb(sc.nodes[l], ray.eye, inv, tmin, tmax);
    bool hr = hit_aabb(sc.nodes[r], ray.eye, inv, tmin, tmax);
    if (hl && hr) {
      cur = l;
      if (sp < 64) { stack[sp++] = r; if ((uint64_t)sp > cx.st.max_stack) cx.st.max_stack = (uint64_t)sp; }
      else cx.st.overflow++;   /* the reference prints a warning and drops the subtree, bvh_traversal.cu:154-164 */
    } else if (hl) cur = l;
    else if (hr) cur = r;
    else {
      if (sp == 0) break;
      cur = stack[--sp];
    }
  }
  return best;
}

/* hit_aabb_adapted as above, also returning the entry distance */
static inline bool hit_aabb_t(const ONode& b, const V3& o, const V3& inv, float tmin, float tmax, float* te)
{
  float tx1 = (b.xmin - o.x) * inv.x, tx2 = (b.xmax - o.x) * inv.x;
  float tnx = fminf(tx1, tx2), tfx = fmaxf(tx1, tx2);
  float ty1 = (b.ymin - o.y) * inv.y, ty2 = (b.ymax - o.y) * inv.y;
  float tny = fminf(ty1, ty2), tfy = fmaxf(ty1, ty2);
  float tz1 = (b.zmin - o.z) * inv.z, tz2 = (b.zmax - o.z) * inv.z;
  float tnz = fminf(tz1, tz2), tfz = fmaxf(tz1, tz2);
  float t_enter = fmaxf(fmaxf(tnx, tny), tnz);
  float t_exit = fminf(fminf(tfx, tfy), tfz);
  *te = t_enter;
  return t_enter < t_exit && t_enter < tmax && t_exit > tmin;
}

static inline uint32_t q_lo(float x, float smin, float step) { float q = floorf((x - smin) / step) - 1.0f; return (uint32_t)fminf(fmaxf(q, 0.0f), 65535.0f); }
static inline uint32_t q_hi(float x, float smin, float step) { float q = ceilf((x - smin) / step) + 1.0f; return (uint32_t)fminf(fmaxf(q, 0.0f), 65535.0f); }
/* One axis of a ray in the grid of the quantised boxes, as the product sets it up (shade_common.h quantised_axis): the plane
 * parameter is f * A + C with f = 2^23 + q (exact in float), A = step * inv, C = (smin - o) * inv - 2^23 * A -+ a margin.  The
 * near plane (low coordinate iff A >= 0) uses Cn, the far plane Cf; the margin bounds every rounding on the way, so that the
 * box tested contains the grid box. */
struct QAxis { float A, Cn, Cf; bool low_near; };
static inline QAxis quantised_axis(float gmin, float gstep, float o, float inv)
{
  QAxis a;
  /* A direction component of (nearly) zero: 1 / d is infinite or so large that 2^23 A would overflow.  The reciprocal is clamped
   * to +-2^60 grid steps per unit of t, i.e. the axis is treated as that of a ray which needs 2^-60 of t per grid step: far
   * beyond any distance a scene has, and still a proper slab test -- a box whose slab the origin is not in is culled, as
   * (plane - o) * inf does in hit_aabb_adapted (bvh_traversal.cu:11-44).  (Until round 3 such an axis was ignored, which made
   * every shadow ray of redchair.txt's `sun 0 1 2` test two axes only.) */
  const float lim = 1152921504606846976.0f / gstep;
  const float ic = fminf(fmaxf(inv, -lim), lim);
  const float B = (gmin - o) * ic;
  a.A = gstep * ic;
  const float hA = 1.0625f * fabsf(a.A);
  const float E = fmaf(fabsf(B), 9.5367431640625e-07f, hA);
  const float C0 = fmaf(-8388608.0f, a.A, B);
  a.Cn = C0 - E;
  a.Cf = C0 + E;
  a.low_near = a.A >= 0.0f;
  return a;
}
/* box test on the quantised form of box b */
static inline bool hit_qbox(const ONode& b, const float* smin, const float* step, const QAxis* ax, float tmin, float tmax, float* te)
{
  const float lo[3] = {8388608.0f + (float)q_lo(b.xmin, smin[0], step[0]), 8388608.0f + (float)q_lo(b.ymin, smin[1], step[1]), 8388608.0f + (float)q_lo(b.zmin, smin[2], step[2])};
  const float hi[3] = {8388608.0f + (float)q_hi(b.xmax, smin[0], step[0]), 8388608.0f + (float)q_hi(b.ymax, smin[1], step[1]), 8388608.0f + (float)q_hi(b.zmax, smin[2], step[2])};
  float tn[3], tf[3];
  for (int k = 0; k < 3; ++k) {
    tn[k] = fmaf(ax[k].low_near ? lo[k] : hi[k], ax[k].A, ax[k].Cn);
    tf[k] = fmaf(ax[k].low_near ? hi[k] : lo[k], ax[k].A, ax[k].Cf);
  }
  float t_enter = fmaxf(fmaxf(tn[0], tn[1]), tn[2]);
  float t_exit = fminf(fminf(tf[0], tf[1]), tf[2]);
  *te = t_enter;
  return t_enter < t_exit && t_enter < tmax && t_exit > tmin;
}

/* traverse() with the product's near-child-first descent (see ORC_FLAG_ORDERED above): same closest hit as traverse(),
 * fewer node visits. */
static Obj traverse_ordered(Ctx& cx, const Ray& ray, float initial_t_max, bool early, float stop_below, bool allow_qn = true)
{
  const Scene& sc = *cx.sc;
  const bool order_pure = (cx.flags & (ORC_FLAG_ORDERED | ORC_FLAG_ORDERED_ALL)) != 0, order_all = (cx.flags & ORC_FLAG_ORDERED_ALL) != 0;
  Obj best = obj_none();
  best.distance = initial_t_max;
  float tmax = initial_t_max;
  bool have = false;
  uint32_t best_leaf = 0;
  if (sc.nodes.empty()) return best;
  V3 inv = mk(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
  const float tmin = 0.0001f;
  const uint32_t N = (uint32_t)sc.refs.size();
  uint32_t stack[64];
  int sp = 0;
  uint32_t cur = 0;
  const bool qn = allow_qn && (cx.flags & ORC_FLAG_QNODES) != 0 && N > 1;      /* (a single primitive has no node records) */
  float qstep[3];
  for (int k = 0; k < 3; ++k) { float range = sc.smax[k] - sc.smin[k]; qstep[k] = range > 0.0f ? range / 65535.0f : 1.0f; }
  const QAxis qax[3] = {quantised_axis(sc.smin[0], qstep[0], ray.eye.x, inv.x), quantised_axis(sc.smin[1], qstep[1], ray.eye.y, inv.y),
                        quantised_axis(sc.smin[2], qstep[2], ray.eye.z, inv.z)};
  while (true) {
    const ONode& node = sc.nodes[cur];
    cx.st.node_iters++;
    if (node.count > 0) {
      const uint32_t k = node.prim_offset;
      const OPrimRef& ref = sc.refs[k];
      Obj h;
      if (ref.type == 0) {
        h = check_sphere(cx, ray, ref.id); cx.st.sphere_tests++;
        if (qn && h.isHit) {      /* the two order-independent clauses of the exact leaf box test */
          float te;
          hit_aabb_t(node, ray.eye, inv, tmin, INFINITY, &te);
          float tx1 = (node.xmin - ray.eye.x) * inv.x, tx2 = (node.xmax - ray.eye.x) * inv.x;
          float ty1 = (node.ymin - ray.eye.y) * inv.y, ty2 = (node.ymax - ray.eye.y) * inv.y;
          float tz1 = (node.zmin - ray.eye.z) * inv.z, tz2 = (node.zmax - ray.eye.z) * inv.z;
          float t_exit = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
          if (!(te < t_exit && t_exit > tmin)) h.isHit = false;
        }
      }
      else { h = check_triangle(cx, ray, ref.id); cx.st.tri_tests++; }
      if (h.isHit && h.distance > 1e-6f && (h.distance < tmax || (have && h.distance == tmax && k < best_leaf))) {
        if (qn && ref.type != 0) {      /* would the reference's walk have reached this triangle? (ORC_FLAG_QNODES above) */
          float te;
          const bool box_ok = hit_aabb_t(node, ray.eye, inv, tmin, INFINITY, &te);      /* t_enter < t_exit && t_exit > t_min */
          if (!(box_ok && te < h.distance)) { cx.st.qn_retraces++; Obj lit = traverse(cx, ray, initial_t_max, early, stop_below); lit.literal = true; return lit; }
        }
        tmax = h.distance; best = h; best.leaf = k; best_leaf = k; have = true;
        if (early && best.distance < stop_below) return best;
      }
      if (sp == 0) break;
      cur = stack[--sp];
      continue;
    }
    cx.st.internal_visits++;
    uint32_t l = node.left, r = node.right;
    float tl, tr;
    bool hl = qn ? hit_qbox(sc.nodes[l], sc.smin, qstep, qax, tmin, tmax, &tl) : hit_aabb_t(sc.nodes[l], ray.eye, inv, tmin, tmax, &tl);
    bool hr = qn ? hit_qbox(sc.nodes[r], sc.smin, qstep, qax, tmin, tmax, &tr) : hit_aabb_t(sc.nodes[r], ray.eye, inv, tmin, tmax, &tr);
    if (hl && hr) {
      bool swap = false;
      if (order_pure && tr < tl) {
        auto sub_pure = [&](uint32_t c) { return c >= N - 1 ? sc.refs[c - (N - 1)].type == 0 : sc.pure(sc.rfirst[c], sc.rlast[c]); };
        swap = order_all || (sub_pure(l) && sub_pure(r));
      }
      cur = swap ? r : l;
      if (sp < 64) { stack[sp++] = swap ? l : r; if ((uint64_t)sp > cx.st.max_stack) cx.st.max_stack = (uint64_t)sp; }
      else cx.st.overflow++;
    } else if (hl) cur = l;
    else if (hr) cur = r;
    else {
      if (sp == 0) break;
      cur = stack[--sp];
    }
  }
  return best;
}

/* ORC_FLAG_WIDE (above): the leaf handling is traverse_ordered's with quantised boxes. */
static Obj traverse_wide(Ctx& cx, const Ray& ray, float initial_t_max, bool early, float stop_below)
{
  const Scene& sc = *cx.sc;
  Obj best = obj_none();
  best.distance = initial_t_max;
  float tmax = initial_t_max;
  bool have = false;
  uint32_t best_leaf = 0;
  V3 inv = mk(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
  const float tmin = 0.0001f;
  uint32_t stack[192];
  int sp = 0;
  uint32_t cur = 0;
  float qstep[3];
  for (int k = 0; k < 3; ++k) { float range = sc.smax[k] - sc.smin[k]; qstep[k] = range > 0.0f ? range / 65535.0f : 1.0f; }
  const QAxis qax[3] = {quantised_axis(sc.smin[0], qstep[0], ray.eye.x, inv.x), quantised_axis(sc.smin[1], qstep[1], ray.eye.y, inv.y),
                        quantised_axis(sc.smin[2], qstep[2], ray.eye.z, inv.z)};
  while (true) {
    const ONode& node = sc.nodes[cur];
    cx.st.node_iters++;
    if (node.count > 0) {
      const uint32_t k = node.prim_offset;
      const OPrimRef& ref = sc.refs[k];
      Obj h;
      if (ref.type == 0) {
        h = check_sphere(cx, ray, ref.id); cx.st.sphere_tests++;
        if (h.isHit) {      /* the two order-independent clauses of the exact leaf box test */
          float te;
          if (!hit_aabb_t(node, ray.eye, inv, tmin, INFINITY, &te)) h.isHit = false;
        }
      }
      else { h = check_triangle(cx, ray, ref.id); cx.st.tri_tests++; }
      if (h.isHit && h.distance > 1e-6f && (h.distance < tmax || (have && h.distance == tmax && k < best_leaf))) {
        if (ref.type != 0) {
          float te;
          const bool box_ok = hit_aabb_t(node, ray.eye, inv, tmin, INFINITY, &te);
          if (!(box_ok && te < h.distance)) { cx.st.qn_retraces++; Obj lit = traverse(cx, ray, initial_t_max, early, stop_below); lit.literal = true; return lit; }
        }
        tmax = h.distance; best = h; best.leaf = k; best_leaf = k; have = true;
        if (early && best.distance < stop_below) return best;
      }
      if (sp == 0) break;
      cur = stack[--sp];
      continue;
    }
    cx.st.internal_visits++;
    uint32_t kids[4];
    int nk = 0;
    const uint32_t two[2] = {node.left, node.right};
    for (int s = 0; s < 2; ++s) {
      const ONode& c = sc.nodes[two[s]];
      if (c.count > 0) kids[nk++] = two[s];
      else { kids[nk++] = c.left; kids[nk++] = c.right; }
    }
    uint32_t hit[4];
    int nh = 0;
    for (int i = 0; i < nk; ++i) { float te; if (hit_qbox(sc.nodes[kids[i]], sc.smin, qstep, qax, tmin, tmax, &te)) hit[nh++] = kids[i]; }
    if (nh == 0) {
      if (sp == 0) break;
      cur = stack[--sp];
      continue;
    }
    cur = hit[0];
    for (int i = nh - 1; i >= 1; --i) {
      if (sp < 192) { stack[sp++] = hit[i]; if ((uint64_t)sp > cx.st.max_stack) cx.st.max_stack = (uint64_t)sp; }
      else cx.st.overflow++;
    }
  }
  return best;
}

static inline Obj traverse_any(Ctx& cx, const Ray& ray, float initial_t_max, bool early, float stop_below)
{
  if (!cx.sc->nodes.empty()) cx.st.traversals++;      /* rays that enter the BVH walk (MirtStats.rays_traversed) */
  if ((cx.flags & ORC_FLAG_WIDE) && (cx.flags & ORC_FLAG_QNODES) && cx.sc->refs.size() > 1) return traverse_wide(cx, ray, initial_t_max, early, stop_below);
  /* (QNODES without an ORDERED flag: the quantised boxes in the reference's order -- a scene whose grid is too coarse for near child first) */
  if (cx.flags & (ORC_FLAG_ORDERED | ORC_FLAG_ORDERED_ALL | ORC_FLAG_QNODES)) return traverse_ordered(cx, ray, initial_t_max, early, stop_below);
  return traverse(cx, ray, initial_t_max, early, stop_below);
}

/* checkPlane, draw.cu:581-615 */
static Obj check_plane(const Ctx& cx, const Ray& ray)
{
  float t_sol = INFINITY;
  V3 p_sol = mk(0, 0, 0), nor = mk(0, 0, 0);
  Mat mats = mat_default();
  uint32_t pid = 0;
  const Scene& sc = *cx.sc;
  for (size_t i = 0; i < sc.planes.size(); ++i) {
    const OPlane& pl = sc.planes[i];
    V3 pn = mk(pl.nor);
    float t = dot(mk(pl.point) - ray.eye, pn) / dot(ray.dir, pn);
    if (t <= 1e-6f) continue;
    V3 ip = t * ray.dir + ray.eye;
    if (t < t_sol && t > 0.001f) {
      t_sol = t; p_sol = ip;
      nor = (dot(pn, ray.dir) < 0.0f) ? pn : -pn;
      mats = mat_from(pl.mat);
      pid = (uint32_t)i;
    }
  }
  if (t_sol >= (float)(INT_MAX - 10)) return obj_none();
  return obj_hit(t_sol, p_sol, nor, mats, 3, pid);
}

/* hitNearest, draw.cu:292-318 */
/* vet: ORC_FLAG_REACH applies to this query (every query that is shaded; shadow queries towards point lights) */
static Obj hit_nearest(Ctx& cx, const Ray& ray, bool count_mat = true, int vet = -1)
{
  if (vet < 0) vet = count_mat ? 1 : 0;
  if (ray.bounce == 0) return obj_none();
  cx.st.rays++;
  Obj b = traverse_any(cx, ray, INFINITY, false, -1.0f);
  Obj p = check_plane(cx, ray);
  const uint32_t N = (uint32_t)cx.sc->refs.size();
  /* ORC_FLAG_REACH: where the product reads the sphere's record to shade it, i.e. when the BVH hit is the nearer one */
  if ((cx.flags & ORC_FLAG_REACH) && vet && b.isHit && b.kind == 1 && !b.literal && N > 1 &&
      (!p.isHit || b.distance < p.distance) && (cx.flags & ORC_FLAG_QNODES)) {
    const V3 inv = mk(1.0f / ray.dir.x, 1.0f / ray.dir.y, 1.0f / ray.dir.z);
    float te;
    const bool box_ok = hit_aabb_t(cx.sc->nodes[N - 1 + b.leaf], ray.eye, inv, 0.0001f, INFINITY, &te);
    if (!(box_ok && te < b.distance)) { cx.st.qn_retraces++; b = traverse(cx, ray, INFINITY, false, -1.0f); }
  }
  Obj r;
  if (b.isHit && p.isHit) r = (b.distance < p.distance) ? b : p;
  else if (b.isHit) r = b;
  else if (p.isHit) r = p;
  else r = obj_none();
  if (r.isHit && r.kind != 3) { cx.st.prim_hits++; if (count_mat) cx.st.mat_fetches++; }
  return r;
}

/* Shadow query = hitNearest on a bounce-1 ray followed by the caller's test (draw.cu:347-352, 365-370).
 * limit = +inf for suns (occluded iff anything is hit), |bulbDir| for bulbs (occluded iff distance < limit). */
static bool occluded(Ctx& cx, const Ray& ray, float limit, bool bulb = false)
{
  cx.st.shadow_rays++;
  /* ORC_FLAG_REACH, a walk that is not the reference's own, a point light: "occluded" means nearer than the light, and a hit the
   * reference never tests can be the one that is (from far enough away every sphere near the light is within an ulp of the
   * light's distance) -- the product traces these rays to their nearest hit and vets it like any other */
  const bool vetted_bulb = bulb && (cx.flags & ORC_FLAG_REACH) && cx.sc->refs.size() > 1 && (cx.flags & ORC_FLAG_QNODES);
  if (!(cx.flags & ORC_FLAG_ANYHIT_SHADOW) || vetted_bulb) {
    Obj h = hit_nearest(cx, ray, false, vetted_bulb ? 1 : 0);
    return h.isHit && h.distance < limit;
  }
  /* early-exit form: same boolean, fewer node visits; the plane is asked first */
  cx.st.rays++;
  Obj p = check_plane(cx, ray);
  if (p.isHit && p.distance < limit) return true;
  Obj b = traverse_any(cx, ray, INFINITY, true, limit);
  return b.isHit && b.distance < limit;
}

/* helper.cu:40-45 */
static inline float set_expose(float c, float expose)
{
  if (expose == INFINITY) return c;
  return (float)(1.0 - (double)o_expf(-expose * c));
}
/* draw.cu:627-636 */
static C4 color_sun(float lambert, const C3& oc, const C3& lc, float expose)
{
  float r = oc.r * (lc.r * lambert), g = oc.g * (lc.g * lambert), b = oc.b * (lc.b * lambert);
  return c4(set_expose(r, expose), set_expose(g, expose), set_expose(b, expose), 0.0f);
}
/* draw.cu:649-659 */
static C4 color_bulb(float lambert, const C3& oc, const C3& lc, float t, float expose)
{
  float i = 1.0f / (t * t);
  float r = oc.r * (lc.r * lambert), g = oc.g * (lc.g * lambert), b = oc.b * (lc.b * lambert);
  return c4(set_expose(r, expose) * i, set_expose(g, expose) * i, set_expose(b, expose) * i, 0.0f);
}

static V3 rough_normal(const Ctx& cx, const V3& n, float roughness, Rng* rng)
{
  /* draw.cu:333-338 / 393-398: the evaluation order of the three arguments is unspecified in C++ */
  float a = standerdD(roughness, rng), b = standerdD(roughness, rng), c = standerdD(roughness, rng);
  if (cx.flags & ORC_FLAG_NORMAL_ZYX) return n + mk(c, b, a);
  return n + mk(a, b, c);
}

/* diffuseLight, draw.cu:329-377 */
static C4 diffuse_light(Ctx& cx, const Obj& obj, Rng* rng)
{
  const Scene& sc = *cx.sc;
  C4 color = c4zero();
  V3 normal = obj.normal;
  if (obj.mat.roughness > 0.0f) normal = rough_normal(cx, normal, obj.mat.roughness, rng);
  normal = normalize(normal);
  for (size_t i = 0; i < sc.suns.size(); ++i) {
    V3 ld = mk(sc.suns[i].dir);
    Ray sr = mkray(obj.i_point + obj.normal * 0.001f, ld, 1);
    const bool unlit = (cx.flags & ORC_FLAG_SKIP_UNLIT) && !(dot(normal, normalize(ld)) > 0.0f);
    if (unlit) { cx.st.rays++; cx.st.shadow_rays++; }
    else if (occluded(cx, sr, INFINITY)) continue;
    float lambert = fmaxf(dot(normal, normalize(ld)), 0.0f);
    color = color + color_sun(lambert, obj.mat.color, c3(sc.suns[i].color), sc.d.expose);
  }
  for (size_t i = 0; i < sc.bulbs.size(); ++i) {
    V3 bd = mk(sc.bulbs[i].point) - obj.i_point;
    Ray sr = mkray(obj.i_point + obj.normal * 0.001f, bd, 1);
    const bool unlit = (cx.flags & ORC_FLAG_SKIP_UNLIT) && !(dot(normal, normalize(bd)) > 0.0f);
    if (unlit) { cx.st.rays++; cx.st.shadow_rays++; }
    else if (occluded(cx, sr, length(bd), true)) continue;
    float lambert = fmaxf(dot(normal, normalize(bd)), 0.0f);
    color = color + color_bulb(lambert, obj.mat.color, c3(sc.bulbs[i].color), length(bd), sc.d.expose);
  }
  return color;
}

static C4 refraction_light(Ctx& cx, const Ray& ray, const Obj& obj, Rng* rng);

/* reflectionLight, draw.cu:389-437 */
static C4 reflection_light(Ctx& cx, const Ray& ray, const Obj& obj, Rng* rng)
{
  if (ceq(obj.mat.shininess, c3(0, 0, 0)) || ray.bounce <= 0) return c4zero();
  V3 normal = obj.normal;
  if (obj.mat.roughness > 0.0f) normal = rough_normal(cx, normal, obj.mat.roughness, rng);
  normal = normalize(normal);
  V3 rd = ray.dir - 2.0f * (dot(normal, ray.dir)) * normal;
  Ray second = mkray(obj.i_point + obj.normal * 0.001f, rd, ray.bounce - 1);
  Obj so = hit_nearest(cx, second);
  C3 shine, trans;
  if (ray.bounce == 1) { shine = c3(0, 0, 0); trans = c3(0, 0, 0); }
  else { shine = so.mat.shininess; trans = so.mat.trans; }
  C4 color;
  if (so.isHit) {
    C4 diffuse = diffuse_light(cx, so, rng);
    C4 reflect = reflection_light(cx, second, so, rng);
    C4 refract = refraction_light(cx, ray, obj, rng);   /* draw.cu:424: the ORIGINAL ray and object */
    C3 one = c3(1.0f, 1.0f, 1.0f);
    color = shine * reflect + (one - shine) * trans * refract + (one - shine) * (one - trans) * diffuse;
  } else color = c4(0.0f, 0.0f, 0.0f, 1.0f);
  return color;
}

/* refractionLight, draw.cu:456-527 */
static C4 refraction_light(Ctx& cx, const Ray& ray, const Obj& obj, Rng* rng)
{
  if (ceq(obj.mat.trans, c3(0, 0, 0)) || ray.bounce <= 0) return c4zero();
  V3 refract_dir;
  Ray inside_ray, final_ray;
  float ior = 1.0f / obj.mat.ior;
  V3 dir = ray.dir;
  V3 normal = normalize(obj.normal);
  V3 i_point = obj.i_point;
  int bounce = ray.bounce;
  float dn = dot(normal, dir);
  float k = 1.0f - (ior * ior) * (1.0f - (dn * dn));
  if (k < 0) {
    refract_dir = dir - 2.0f * (dot(normal, dir)) * normal;
    final_ray = mkray(i_point + normal * 0.001f, refract_dir, --bounce);
  } else {
    refract_dir = ior * dir - (ior * (dot(normal, dir)) + sqrtf(k)) * normal;
    inside_ray = mkray(i_point - normal * 0.0001f, refract_dir, bounce);
    Obj other = hit_nearest(cx, inside_ray);   /* no miss check in the reference (draw.cu:482-487) */
    normal = normalize(other.normal);
    ior = other.mat.ior;
    dir = inside_ray.dir;
    i_point = other.i_point;
    float dn2 = dot(normal, dir);
    k = 1.0f - ior * ior * (1.0f - (dn2 * dn2));
    refract_dir = ior * dir - (ior * (dot(normal, dir)) + sqrtf(k)) * normal;
    final_ray = mkray(i_point - normal * 0.0001f, refract_dir, --bounce);
  }
  Obj fo = hit_nearest(cx, final_ray);
  C3 shine, trans;
  if (bounce == 0) { shine = c3(0, 0, 0); trans = c3(0, 0, 0); }
  else { shine = fo.mat.shininess; trans = fo.mat.trans; }
  C4 color;
  if (fo.isHit) {
    C4 diffuse = diffuse_light(cx, fo, rng);
    C4 reflect = reflection_light(cx, final_ray, fo, rng);
    C4 refract = refraction_light(cx, final_ray, fo, rng);
    C3 one = c3(1.0f, 1.0f, 1.0f);
    color = shine * reflect + (one - shine) * trans * refract + (one - shine) * (one - trans) * diffuse;
  } else color = c4(0.0f, 0.0f, 0.0f, 1.0f);
  return color;
}

/* spherePoint, helper.cu:91-101 */
static V3 sphere_point(Rng* rng)
{
  float z = 2.0f * randD(0.0f, 1.0f, rng) - 1.0f;
  float theta = 2.0f * 3.14159265f * randD(0.0f, 1.0f, rng);
  float r = sqrtf(1.0f - z * z);
  float x = r * o_cosf(theta);
  float y = r * o_sinf(theta);
  return mk(x, y, z);
}

/* globalIllumination, draw.cu:540-568 */
static C4 global_illumination(Ctx& cx, const Obj& obj, int gi_bounce, Rng* rng)
{
  if (cx.sc->d.gi == 0 || gi_bounce == 0) return c4zero();
  V3 normal = obj.normal;
  V3 i_point = obj.i_point;
  V3 gi_dir = normalize(normal + sphere_point(rng));
  Ray gi_ray = mkray(i_point + normal * 0.001f, gi_dir, gi_bounce - 1);
  Obj go = hit_nearest(cx, gi_ray);
  C4 color = c4zero();
  if (go.isHit) {
    C4 diffuse = diffuse_light(cx, go, rng);
    C4 reflect = reflection_light(cx, gi_ray, go, rng);
    C4 refract = refraction_light(cx, gi_ray, go, rng);
    C4 gi_color = go.mat.color * global_illumination(cx, go, gi_bounce - 1, rng);
    C3 one = c3(1.0f, 1.0f, 1.0f);
    color = go.mat.shininess * reflect + (one - go.mat.shininess) * go.mat.trans * refract +
            (one - go.mat.shininess) * (one - go.mat.trans) * (diffuse + gi_color);
    color.a = 1.0f;
  }
  return color;
}

/* Ray::Ray(x, y, state, config), struct.cu:16-62 */
static Ray primary_ray(const Ctx& cx, float x, float y, Rng* rng)
{
  const OSceneDesc& d = cx.sc->d;
  const float PI = 3.14159265358979323846f;
  float max_dim = fmaxf((float)cx.width, (float)cx.height);
  float sx = (2.0f * x - (float)cx.width) / max_dim;
  float sy = ((float)cx.height - 2.0f * y) / max_dim;
  V3 fwd = mk(d.forward), right = mk(d.right), up = mk(d.up);
  Ray r;
  r.eye = mk(d.eye);
  V3 dir;
  if (d.fisheye) {
    dir = (sx * right + sy * up) + sqrtf(1.0f - (sx * sx) - (sy * sy)) * fwd;
  } else if (d.panorama) {
    sx = x / (float)cx.width;
    sy = y / (float)cx.height;
    float theta = (sx - 0.5f) * 2.0f * PI;
    float phi = (sy - 0.5f) * PI;
    dir = o_cosf(phi) * (o_cosf(theta) * fwd + o_sinf(theta) * right) - o_sinf(phi) * up;
    dir = normalize(dir);
  } else if (d.dof_focus != 0.0f) {
    float theta = randD(0.0f, 2.0f * PI, rng);
    float rr = randD(0.0f, d.dof_lens, rng);
    float lx = rr * o_cosf(theta);
    float ly = rr * o_sinf(theta);
    r.eye = r.eye + lx * up + ly * right;
    V3 old_dir = fwd + sx * right + sy * up;
    dir = (mk(d.eye) + normalize(old_dir) * d.dof_focus - r.eye) / d.dof_focus;
  } else {
    dir = fwd + sx * right + sy * up;
  }
  r.bounce = d.bounces;
  r.dir = normalize(dir);
  return r;
}

/* shootPrimaryRay, draw.cu:260-285 */
static C4 shoot_primary(Ctx& cx, float x, float y, Rng* rng, OHit* aov)
{
  Ray ray = primary_ray(cx, x, y, rng);
  Obj obj = hit_nearest(cx, ray);
  if (aov) {
    aov->t = obj.isHit ? obj.distance : -1.0f; aov->kind = obj.kind; aov->id = obj.id;
    aov->nx = obj.normal.x; aov->ny = obj.normal.y; aov->nz = obj.normal.z;
  }
  C4 color = c4zero();
  if (obj.isHit) {
    C4 diffuse = diffuse_light(cx, obj, rng);
    C4 reflect = reflection_light(cx, ray, obj, rng);
    C4 refract = refraction_light(cx, ray, obj, rng);
    C4 gi_color = obj.mat.color * global_illumination(cx, obj, cx.sc->d.gi, rng);
    C3 one = c3(1.0f, 1.0f, 1.0f);
    color = obj.mat.shininess * reflect + (one - obj.mat.shininess) * obj.mat.trans * refract +
            (one - obj.mat.shininess) * (one - obj.mat.trans) * (diffuse + gi_color);
    color.a = 1.0f;
  }
  return color;
}

/* RGBtosRGB, helper.cu:12-27 */
static float rgb_to_srgb(float l)
{
  float sol;
  if (l < 0.0031308f) sol = 12.92f * l;
  else sol = 1.055f * o_powf(l, 1 / 2.4f) - 0.055f;
  sol = fminf(1.0f, fmaxf(0.0f, sol));
  return sol;
}
/* draw.cu:9-11 */
static inline uint8_t float_to_uchar_round(float f) { return (uint8_t)(fminf(fmaxf(f, 0.0f), 1.0f) * 255.0f + 0.5f); }
/* draw.cu:129-132: float -> unsigned char conversion (values are in [0,255]; NaN -> 0) */
static inline uint8_t float_to_uchar_trunc(float f)
{
  if (!(f > 0.0f)) return 0;
  if (f >= 255.0f) return 255;
  return (uint8_t)f;
}

static void stats_add(OStats* a, const OStats& b)
{
  a->samples += b.samples; a->rays += b.rays; a->shadow_rays += b.shadow_rays; a->node_iters += b.node_iters;
  a->internal_visits += b.internal_visits; a->sphere_tests += b.sphere_tests; a->tri_tests += b.tri_tests;
  a->mat_fetches += b.mat_fetches; a->prim_hits += b.prim_hits; a->overflow += b.overflow; a->qn_retraces += b.qn_retraces; a->traversals += b.traversals;
  if (b.max_stack > a->max_stack) a->max_stack = b.max_stack;
}

/* One pixel.  spp <= 1: render_kernel, draw.cu:94-133 (seed 1234, subsequence = pixel index).
 * spp > 1: render_kernel_warp_aa, draw.cu:135-213, generalised from its fixed 32 samples to spp samples
 * (SURVEY.md section 0.4): sample s uses curand_init(1234 + pixel, s, 0); the samples are summed in the
 * kernel's xor-butterfly order over the next power of two >= spp (absent samples add 0) and scaled by 1/spp. */
static void render_pixel(Ctx& cx, int px, int py, int spp, float* out_f, uint8_t* out_u8, OHit* aov)
{
  const int pixel = py * cx.width + px;
  C4 rgba;
  if (spp <= 1) {
    Rng rng;
    rng_init(&rng, 1234, (uint64_t)pixel, 0);
    if (spp == 0) {
      rgba = shoot_primary(cx, (float)px, (float)py, &rng, aov);
    } else {
      C4 acc = c4zero();
      for (int i = 0; i < spp; ++i) {
        float nw = (float)px + randD(-0.5f, 0.5f, &rng);
        float nh = (float)py + randD(-0.5f, 0.5f, &rng);
        acc = acc + shoot_primary(cx, nw, nh, &rng, aov);
      }
      float inv = 1.0f / (float)spp;
      rgba = c4(acc.r * inv, acc.g * inv, acc.b * inv, acc.a * inv);
    }
    cx.st.samples += 1;
    if (out_f) { out_f[0] = rgba.r; out_f[1] = rgba.g; out_f[2] = rgba.b; out_f[3] = rgba.a; }
    if (out_u8) {
      out_u8[0] = float_to_uchar_trunc(rgb_to_srgb(rgba.r) * 255);
      out_u8[1] = float_to_uchar_trunc(rgb_to_srgb(rgba.g) * 255);
      out_u8[2] = float_to_uchar_trunc(rgb_to_srgb(rgba.b) * 255);
      out_u8[3] = float_to_uchar_trunc(rgba.a * 255);
    }
    return;
  }
  int P = 1; while (P < spp) P <<= 1;
  std::vector<C4> v((size_t)P, c4zero());
  for (int s = 0; s < spp; ++s) {
    Rng rng;
    rng_init(&rng, 1234ull + (uint64_t)pixel, (uint64_t)s, 0);
    float jx = randD(-0.5f, 0.5f, &rng);
    float jy = randD(-0.5f, 0.5f, &rng);
    float sw = (float)px + jx, sh = (float)py + jy;
    v[(size_t)s] = shoot_primary(cx, sw, sh, &rng, (aov && s == 0) ? aov : NULL);
    cx.st.samples += 1;
  }
  /* draw.cu:181-189: for (mask = P/2; mask > 0; mask /= 2) val += shfl_xor(val, mask) */
  for (int mask = P / 2; mask > 0; mask /= 2) {
    std::vector<C4> nv((size_t)P);
    for (int l = 0; l < P; ++l) nv[(size_t)l] = v[(size_t)l] + v[(size_t)(l ^ mask)];
    v.swap(nv);
  }
  const float inv = 1.0f / (float)spp;
  float ar = v[0].r * inv, ag = v[0].g * inv, ab = v[0].b * inv, aa = v[0].a * inv;
  if (out_f) { out_f[0] = ar; out_f[1] = ag; out_f[2] = ab; out_f[3] = aa; }
  if (out_u8) {
    out_u8[0] = float_to_uchar_round(rgb_to_srgb(ar));
    out_u8[1] = float_to_uchar_round(rgb_to_srgb(ag));
    out_u8[2] = float_to_uchar_round(rgb_to_srgb(ab));
    out_u8[3] = float_to_uchar_round(aa);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* C entry points (ctypes)                                                                           */
/* ------------------------------------------------------------------------------------------------ */
extern "C" {

void* orc_scene_create(const OSceneDesc* d)
{
  jump_init();
  Scene* sc = new Scene();
  sc->d = *d;
  sc->spheres.assign(d->spheres, d->spheres + d->num_spheres);
  sc->tris.assign(d->triangles, d->triangles + d->num_triangles);
  sc->refs.assign(d->prim_refs, d->prim_refs + d->num_prims);
  sc->planes.assign(d->planes, d->planes + d->num_planes);
  sc->suns.assign(d->suns, d->suns + d->num_suns);
  sc->bulbs.assign(d->bulbs, d->bulbs + d->num_bulbs);
  sc->d.spheres = NULL; sc->d.triangles = NULL; sc->d.prim_refs = NULL; sc->d.planes = NULL; sc->d.suns = NULL; sc->d.bulbs = NULL;
  sc->built = false;
  for (int i = 0; i < 3; ++i) { sc->smin[i] = INFINITY; sc->smax[i] = -INFINITY; }
  return sc;
}
void orc_scene_destroy(void* h) { delete (Scene*)h; }

int orc_build_lbvh(void* h, int bounds_mode) { return build(*(Scene*)h, bounds_mode); }
int orc_num_nodes(void* h) { return (int)((Scene*)h)->nodes.size(); }
void orc_get_nodes(void* h, ONode* out) { Scene* s = (Scene*)h; if (!s->nodes.empty()) memcpy(out, s->nodes.data(), s->nodes.size() * sizeof(ONode)); }
void orc_get_codes(void* h, uint32_t* out) { Scene* s = (Scene*)h; if (!s->codes.empty()) memcpy(out, s->codes.data(), s->codes.size() * 4); }
void orc_get_refs(void* h, OPrimRef* out) { Scene* s = (Scene*)h; if (!s->refs.empty()) memcpy(out, s->refs.data(), s->refs.size() * sizeof(OPrimRef)); }
void orc_get_bounds(void* h, float* mn, float* mx) { Scene* s = (Scene*)h; memcpy(mn, s->smin, 12); memcpy(mx, s->smax, 12); }

/* Render the tile [x0,x0+tw) x [y0,y0+th) of a width x height frame at spp samples per pixel.
 * out_f: tw*th*4 floats (linear RGBA mean before sRGB), out_u8: tw*th*4 bytes, aov: tw*th primary hit records
 * (any may be NULL).  nthreads > 1 uses OpenMP over rows (results do not depend on it). */
int orc_render(void* h, int width, int height, int spp, int x0, int y0, int tw, int th,
               float* out_f, uint8_t* out_u8, OHit* aov, OStats* stats, uint32_t flags, int nthreads)
{
  Scene* sc = (Scene*)h;
  if (!sc->built) return 1;
  OStats total; memset(&total, 0, sizeof(total));
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
  #pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
  for (int ty = 0; ty < th; ++ty) {
    Ctx cx; cx.sc = sc; cx.width = width; cx.height = height; cx.flags = flags; memset(&cx.st, 0, sizeof(cx.st));
    for (int tx = 0; tx < tw; ++tx) {
      size_t o = (size_t)ty * tw + tx;
      render_pixel(cx, x0 + tx, y0 + ty, spp, out_f ? out_f + o * 4 : NULL, out_u8 ? out_u8 + o * 4 : NULL, aov ? aov + o : NULL);
    }
#ifdef _OPENMP
    #pragma omp critical
#endif
    stats_add(&total, cx.st);
  }
  (void)nthreads;
  if (stats) *stats = total;
  return 0;
}

/* render_kernel_atomic_aa, draw.cu:49-92, as the product's mirt_render_accumulate defines it: for every pixel of the tile the
 * samples [first, first + count) (curand_init(1234 + pixel, s, 0), jittered) are summed in the xor-butterfly order over the
 * next power of two >= count and the sum is ADDED to accum (tw*th*4 floats).  finalize_kernel (draw.cu:13-47) is
 * orc_finalize. */
int orc_render_accumulate(void* h, int width, int height, int x0, int y0, int tw, int th, int first, int count, float* accum,
                          uint32_t flags, int nthreads)
{
  Scene* sc = (Scene*)h;
  if (!sc->built || count < 1) return 1;
  int P = 1; while (P < count) P <<= 1;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
  #pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
  for (int ty = 0; ty < th; ++ty) {
    Ctx cx; cx.sc = sc; cx.width = width; cx.height = height; cx.flags = flags; memset(&cx.st, 0, sizeof(cx.st));
    for (int tx = 0; tx < tw; ++tx) {
      const int px = x0 + tx, py = y0 + ty, pixel = py * width + px;
      std::vector<C4> v((size_t)P, c4zero());
      for (int s = 0; s < count; ++s) {
        Rng rng;
        rng_init(&rng, 1234ull + (uint64_t)pixel, (uint64_t)(first + s), 0);
        float jx = randD(-0.5f, 0.5f, &rng);
        float jy = randD(-0.5f, 0.5f, &rng);
        v[(size_t)s] = shoot_primary(cx, (float)px + jx, (float)py + jy, &rng, NULL);
      }
      for (int mask = P / 2; mask > 0; mask /= 2) {
        std::vector<C4> nv((size_t)P);
        for (int l = 0; l < P; ++l) nv[(size_t)l] = v[(size_t)l] + v[(size_t)(l ^ mask)];
        v.swap(nv);
      }
      float* a = accum + ((size_t)ty * tw + tx) * 4;
      a[0] = a[0] + v[0].r; a[1] = a[1] + v[0].g; a[2] = a[2] + v[0].b; a[3] = a[3] + v[0].a;
    }
  }
  (void)nthreads;
  return 0;
}
void orc_finalize(const float* accum, int n, int aa, uint8_t* out)
{
  const float inv = 1.0f / (float)aa;
  for (int i = 0; i < n; ++i) {
    out[4 * i + 0] = float_to_uchar_round(rgb_to_srgb(accum[4 * i + 0] * inv));
    out[4 * i + 1] = float_to_uchar_round(rgb_to_srgb(accum[4 * i + 1] * inv));
    out[4 * i + 2] = float_to_uchar_round(rgb_to_srgb(accum[4 * i + 2] * inv));
    out[4 * i + 3] = float_to_uchar_round(accum[4 * i + 3] * inv);
  }
}

/* Strided sub-sample of a frame (every step-th pixel in x and y) for the bench's bounded CPU baseline. */
int orc_render_subsample(void* h, int width, int height, int spp, int step, OStats* stats, uint32_t flags, int nthreads, float* checksum)
{
  Scene* sc = (Scene*)h;
  if (!sc->built) return 1;
  OStats total; memset(&total, 0, sizeof(total));
  double sum = 0.0;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
  #pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+:sum)
#endif
  for (int y = 0; y < height; y += step) {
    Ctx cx; cx.sc = sc; cx.width = width; cx.height = height; cx.flags = flags; memset(&cx.st, 0, sizeof(cx.st));
    for (int x = 0; x < width; x += step) {
      float f[4];
      render_pixel(cx, x, y, spp, f, NULL, NULL);
      sum += (double)f[0] + f[1] + f[2] + f[3];
    }
#ifdef _OPENMP
    #pragma omp critical
#endif
    stats_add(&total, cx.st);
  }
  (void)nthreads;
  if (stats) *stats = total;
  if (checksum) *checksum = (float)sum;
  return 0;
}

/* ---- small probes for unit tests ---- */
void orc_xorwow_seq(uint64_t seed, uint64_t subseq, uint64_t offset, int n, uint32_t* out)
{
  jump_init();
  Rng r; rng_init(&r, seed, subseq, offset);
  for (int i = 0; i < n; ++i) out[i] = rng_next(&r);
}
void orc_xorwow_state(uint64_t seed, uint64_t subseq, uint32_t* out6)
{
  jump_init();
  Rng r; rng_init(&r, seed, subseq, 0);
  if (r.pending) rng_apply_subseq(&r);
  memcpy(out6, r.v, 20); out6[5] = r.d;
}
void orc_xorwow_floats(uint64_t seed, uint64_t subseq, int n_uniform, int n_normal, float* out)
{
  jump_init();
  Rng r; rng_init(&r, seed, subseq, 0);
  for (int i = 0; i < n_uniform; ++i) out[i] = rng_uniform(&r);
  for (int i = 0; i < n_normal; ++i) out[n_uniform + i] = rng_normal(&r);
}
/* the step^(2^67) matrix, column-major 160 x 5 words, for cross-checks */
void orc_jump_matrix(int k, uint32_t* out) { jump_init(); memcpy(out, g_jump.pow2[k].col, sizeof(g_jump.pow2[k].col)); }
void orc_math_probe(int which, int n, const float* in, float* out)
{
  for (int i = 0; i < n; ++i) {
    switch (which) {
      case 0: out[i] = o_logf(in[i]); break;
      case 1: out[i] = o_expf(in[i]); break;
      case 2: out[i] = o_sinf(in[i]); break;
      case 3: out[i] = o_cosf(in[i]); break;
      case 4: out[i] = o_powf(in[i], 1 / 2.4f); break;
      case 5: out[i] = rgb_to_srgb(in[i]); break;
      case 6: out[i] = sqrtf(in[i]); break;
      case 7: out[i] = 1.0f / in[i]; break;
      default: out[i] = in[i];
    }
  }
}
uint32_t orc_morton(float cx, float cy, float cz, const float* mn, const float* mx)
{
  return morton_3d(quantize_coordinate(cx, mn[0], mx[0] - mn[0], 10), quantize_coordinate(cy, mn[1], mx[1] - mn[1], 10),
                   quantize_coordinate(cz, mn[2], mx[2] - mn[2], 10));
}
int orc_sizeof(int which)
{
  switch (which) {
    case 0: return (int)sizeof(OMat); case 1: return (int)sizeof(OSphere); case 2: return (int)sizeof(OTriangle);
    case 3: return (int)sizeof(OPlane); case 4: return (int)sizeof(OSun); case 5: return (int)sizeof(OPrimRef);
    case 6: return (int)sizeof(OSceneDesc); case 7: return (int)sizeof(ONode); case 8: return (int)sizeof(OStats);
    case 9: return (int)sizeof(OHit);
  }
  return -1;
}

} /* extern "C" */
