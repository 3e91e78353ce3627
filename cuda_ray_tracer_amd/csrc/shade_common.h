// Device code shared by the render kernels: the shading state machine (the reference's recursive
// shootPrimaryRay / diffuseLight / reflectionLight / refractionLight / globalIllumination, draw.cu:260-568, as an explicit
// ray-tree walk), the camera (struct.cu:16-62), plane test (draw.cu:581-615) and per-lane state.  Included by
// render.hip (single-kernel path) and wavefront.hip (trace / shade kernel pair).
#ifndef MIRT_SHADE_COMMON_H
#define MIRT_SHADE_COMMON_H

#include "scene_dev.h"

#include <climits>
#include <cmath>

namespace mirt {
namespace {

constexpr int STACK_TOTAL = 64;      // TRAVERSAL_STACK_SIZE, bvh_traversal.cu:8
constexpr int PENDING_WORDS = 16;

constexpr float EPSILON = 0.001f;    // draw.cu:7, struct.cu:8

enum : int { ST_PRIMARY = 0, ST_BATCH, ST_REFR_INSIDE, ST_REFR_FINAL, ST_GI };
enum : int { M_BATCH = 0, M_POP, M_TRACE, M_DONE };
enum : uint32_t { PEND_F = 1u, PEND_G = 2u };

struct Mat { f3 color, shininess, trans; float ior, roughness; };

MIRT_DEV Mat load_mat(const float4* __restrict__ mats, uint32_t idx)
{
  const float4 a = mats[3 * (size_t)idx + 0], b = mats[3 * (size_t)idx + 1], c = mats[3 * (size_t)idx + 2];
  Mat m;
  m.color = mk3(a.x, a.y, a.z); m.shininess = mk3(a.w, b.x, b.y); m.trans = mk3(b.z, b.w, c.x); m.ior = c.y; m.roughness = c.z;
  return m;
}
MIRT_DEV Mat plane_mat(const PlaneDev& p)
{
  Mat m;
  m.color = mk3(p.mat[0], p.mat[1], p.mat[2]); m.shininess = mk3(p.mat[3], p.mat[4], p.mat[5]); m.trans = mk3(p.mat[6], p.mat[7], p.mat[8]);
  m.ior = p.mat[9]; m.roughness = p.mat[10];
  return m;
}

// setExpose, helper.cu:40-45 (the subtraction is in double)
MIRT_DEV float set_expose(float c, float expose)
{
  if (expose == INFINITY) return c;
  return (float)(1.0 - (double)dm_expf(-expose * c));
}

// Ray(eye, dir, bounce) normalises dir, object.cuh:69
struct RayS { f3 o, d; int bounce; };
MIRT_DEV RayS mkray(const f3& o, const f3& d, int bounce) { RayS r; r.o = o; r.d = normalize(d); r.bounce = bounce; return r; }

// Ray::Ray(x, y, state, config), struct.cu:16-62
MIRT_DEV RayS primary_ray(const RenderArgs& a, float x, float y, Xorwow& rng)
{
  const float PI = 3.14159265358979323846f;
  const float max_dim = fmaxf((float)a.width, (float)a.height);
  float sx = (2.0f * x - (float)a.width) / max_dim;
  float sy = ((float)a.height - 2.0f * y) / max_dim;
  RayS r;
  r.o = a.eye;
  f3 dir;
  if (a.fisheye) {
    dir = (sx * a.right + sy * a.up) + sqrtf(1.0f - (sx * sx) - (sy * sy)) * a.forward;
  } else if (a.panorama) {
    sx = x / (float)a.width;
    sy = y / (float)a.height;
    const float theta = (sx - 0.5f) * 2.0f * PI;
    const float phi = (sy - 0.5f) * PI;
    dir = dm_cosf(phi) * (dm_cosf(theta) * a.forward + dm_sinf(theta) * a.right) - dm_sinf(phi) * a.up;
    dir = normalize(dir);
  } else if (a.dof_focus != 0.0f) {
    const float theta = randD(0.0f, 2.0f * PI, rng);
    const float rr = randD(0.0f, a.dof_lens, rng);
    const float lx = rr * dm_cosf(theta);
    const float ly = rr * dm_sinf(theta);
    r.o = r.o + lx * a.up + ly * a.right;
    const f3 old_dir = a.forward + sx * a.right + sy * a.up;
    dir = (a.eye + normalize(old_dir) * a.dof_focus - r.o) / a.dof_focus;
  } else {
    dir = a.forward + sx * a.right + sy * a.up;
  }
  r.bounce = a.bounces;
  r.d = normalize(dir);
  return r;
}

// spherePoint, helper.cu:91-101
MIRT_DEV f3 sphere_point(Xorwow& rng)
{
  const float z = 2.0f * randD(0.0f, 1.0f, rng) - 1.0f;
  const float theta = 2.0f * 3.14159265f * randD(0.0f, 1.0f, rng);
  const float r = sqrtf(1.0f - z * z);
  const float x = r * dm_cosf(theta);
  const float y = r * dm_sinf(theta);
  return mk3(x, y, z);
}

// draw.cu:333-338 / 393-398 (argument evaluation order: left to right)
MIRT_DEV f3 rough_normal(const f3& n, float roughness, Xorwow& rng)
{
  const float a = standerdD(roughness, rng);
  const float b = standerdD(roughness, rng);
  const float c = standerdD(roughness, rng);
  return n + mk3(a, b, c);
}

// hit_aabb_adapted, bvh_traversal.cu:11-44, on both child boxes of a node record (q0..q2: six (min, max) pairs).  Written
// on 2-vectors so that the subtract/multiply of a pair is one packed instruction (v_pk_add_f32 / v_pk_mul_f32); every
// lane value is the same single-rounded (plane - origin) * inv_dir product as in the scalar form.
typedef float v2f __attribute__((ext_vector_type(2)));
MIRT_DEV void box_pair(const float4 q0, const float4 q1, const float4 q2, float o_x, float o_y, float o_z, float i_x, float i_y, float i_z,
                       float tbest, float tmin, bool& hl, bool& hr, float& tel, float& ter)
{
  const v2f ox = {o_x, o_x}, oy = {o_y, o_y}, oz = {o_z, o_z};
  const v2f ix = {i_x, i_x}, iy = {i_y, i_y}, iz = {i_z, i_z};
  const v2f lx = (v2f{q0.x, q0.y} - ox) * ix, ly = (v2f{q0.z, q0.w} - oy) * iy, lz = (v2f{q1.x, q1.y} - oz) * iz;
  const v2f rx = (v2f{q1.z, q1.w} - ox) * ix, ry = (v2f{q2.x, q2.y} - oy) * iy, rz = (v2f{q2.z, q2.w} - oz) * iz;
  float te = fmaxf(fmaxf(fminf(lx.x, lx.y), fminf(ly.x, ly.y)), fminf(lz.x, lz.y));
  float tx = fminf(fminf(fmaxf(lx.x, lx.y), fmaxf(ly.x, ly.y)), fmaxf(lz.x, lz.y));
  hl = te < tx && te < tbest && tx > tmin;
  tel = te;
  te = fmaxf(fmaxf(fminf(rx.x, rx.y), fminf(ry.x, ry.y)), fminf(rz.x, rz.y));
  tx = fminf(fminf(fmaxf(rx.x, rx.y), fmaxf(ry.x, ry.y)), fmaxf(rz.x, rz.y));
  hr = te < tx && te < tbest && tx > tmin;
  ter = te;
}

// The same test on a quantised node record (scene_dev.h): a box plane is grid_origin + q * grid_step, so its ray parameter is
// q * A + B with A = grid_step * inv_dir and B = (grid_origin - origin) * inv_dir, both set up once per ray (start_ray): one
// fused multiply-add per plane.
// * No integer-to-float conversion: a byte permute puts the 16-bit coordinate into the mantissa of 2^23, giving the float
//   f = 2^23 + q exactly, and the offset carries the -2^23 * A (start_ray).
// * No per-axis min / max: whether the low or the high coordinate is the plane the ray meets first depends on the sign of A
//   alone, so the permute's per-lane selector (sel: low half, or high half when A < 0) picks the near plane and
//   sel ^ 0x0202 the far one.  Twelve permutes + twelve fused multiply-adds + two 3-way max / min per node, where the
//   convert-and-compare form needed twelve conversions and twelve min / max more.
// * (The packed form of the multiply-adds, v_pk_fma_f32 on the two boxes' planes of one kind, needs its operands in register
//   pairs: the moves eat the saving and push spills into the loop, measured twice.)
// * The near planes use the offset Cn, the far planes Cf: quantised_axis moves Cn down and Cf up by a bound on every rounding
//   error on the way, so that the box the kernel tests contains the grid box for any ray origin, however far from the scene
//   (origins on an infinite plane are).
MIRT_DEV void box_pair_q(const uint4 w0, const uint32_t w4, const uint32_t w5, const f3& A, const f3& Cn, const f3& Cf, uint32_t sx, uint32_t sy, uint32_t sz,
                         float tbest, float tmin, bool& hl, bool& hr, float& tel, float& ter)
{
  const uint32_t magic = 0x4b000000u;      // 2^23: v_perm_b32 takes bytes 4-7 from its first operand, 0-3 from the second
#define MIRT_QN(w, s) __uint_as_float(__builtin_amdgcn_perm(magic, (w), (s)))
#define MIRT_QF(w, s) __uint_as_float(__builtin_amdgcn_perm(magic, (w), (s) ^ 0x0202u))
  const float lnx = __builtin_fmaf(MIRT_QN(w0.x, sx), A.x, Cn.x), lfx = __builtin_fmaf(MIRT_QF(w0.x, sx), A.x, Cf.x);
  const float lny = __builtin_fmaf(MIRT_QN(w0.y, sy), A.y, Cn.y), lfy = __builtin_fmaf(MIRT_QF(w0.y, sy), A.y, Cf.y);
  const float lnz = __builtin_fmaf(MIRT_QN(w0.z, sz), A.z, Cn.z), lfz = __builtin_fmaf(MIRT_QF(w0.z, sz), A.z, Cf.z);
  const float rnx = __builtin_fmaf(MIRT_QN(w0.w, sx), A.x, Cn.x), rfx = __builtin_fmaf(MIRT_QF(w0.w, sx), A.x, Cf.x);
  const float rny = __builtin_fmaf(MIRT_QN(w4, sy), A.y, Cn.y),   rfy = __builtin_fmaf(MIRT_QF(w4, sy), A.y, Cf.y);
  const float rnz = __builtin_fmaf(MIRT_QN(w5, sz), A.z, Cn.z),   rfz = __builtin_fmaf(MIRT_QF(w5, sz), A.z, Cf.z);
#undef MIRT_QN
#undef MIRT_QF
  float te = fmaxf(fmaxf(lnx, lny), lnz);
  float tx = fminf(fminf(lfx, lfy), lfz);
  hl = te < tx && te < tbest && tx > tmin;
  tel = te;
  te = fmaxf(fmaxf(rnx, rny), rnz);
  tx = fminf(fminf(rfx, rfy), rfz);
  hr = te < tx && te < tbest && tx > tmin;
  ter = te;
}

// one box of a wide record (scene_dev.h): the same arithmetic as box_pair_q on three words
MIRT_DEV bool box_q(const uint32_t wx, const uint32_t wy, const uint32_t wz, const f3& A, const f3& Cn, const f3& Cf, uint32_t sx, uint32_t sy, uint32_t sz,
                    float tbest, float tmin)
{
  const uint32_t magic = 0x4b000000u;
  const float nx = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(magic, wx, sx)), A.x, Cn.x), fx = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(magic, wx, sx ^ 0x0202u)), A.x, Cf.x);
  const float ny = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(magic, wy, sy)), A.y, Cn.y), fy = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(magic, wy, sy ^ 0x0202u)), A.y, Cf.y);
  const float nz = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(magic, wz, sz)), A.z, Cn.z), fz = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_perm(magic, wz, sz ^ 0x0202u)), A.z, Cf.z);
  const float te = fmaxf(fmaxf(nx, ny), nz);
  const float tx = fminf(fminf(fx, fy), fz);
  return te < tx && te < tbest && tx > tmin;
}

// One axis of a ray in the grid of the quantised node records (start_ray): A, the offsets of the near and the far plane for
// f = 2^23 + q, and the permute selector of the near plane's half-word.  C0 = B - 2^23 A is the offset of the plain form
// q * A + B; the margin E = 2^-20 |B| + 1.0625 |A| exceeds every rounding on the way: of A, of B and of the final fused
// multiply-add (2^-21 (|B| + 65535 |A|) covers those with room to spare), of C0 and of C0 -+ E (each at most
// 2^-24 |B| + |A| / 2, half an ulp of a number near 2^23 |A|).  The tested box is the grid box grown by about one more grid
// step.
// A direction component of (nearly) zero -- 1 / d infinite, or so large that 2^23 A would overflow -- has its reciprocal
// clamped to +-`lim` = 2^60 grid steps per unit of t: the axis becomes that of a ray needing 2^-60 of t per grid step, far
// beyond any distance of a scene, and stays a proper slab test: a box whose slab the origin is not in is culled, as
// (plane - o) * inf does in hit_aabb_adapted (bvh_traversal.cu:11-44); all of the accounting above is linear in A.  (Rounds
// 1-2 ignored such an axis, a superset too, but every shadow ray of redchair.txt's `sun 0 1 2` then tested two axes only.)
MIRT_DEV void quantised_axis(float gmin, float gstep, float lim, float o, float inv, float& A, float& Cn, float& Cf, uint32_t& sel)
{
  const float ic = fminf(fmaxf(inv, -lim), lim);
  const float B = (gmin - o) * ic;
  A = gstep * ic;
  const float E = __builtin_fmaf(fabsf(B), 9.5367431640625e-07f, 1.0625f * fabsf(A));
  const float C0 = __builtin_fmaf(-8388608.0f, A, B);
  Cn = C0 - E;
  Cf = C0 + E;
  sel = A >= 0.0f ? 0x07060100u : 0x07060302u;
}

// The reference's leaf box of a sphere (c -+ r, lbvh_builder.cu:33-41) against a ray, as hit_aabb_adapted evaluates it
// (bvh_traversal.cu:11-44): entry and exit parameters.
MIRT_DEV void sphere_leaf_box(const float4 q, const f3& o, const f3& inv, float& te, float& tx)
{
  const float ax = q.x - q.w, bx = q.x + q.w, ay = q.y - q.w, by = q.y + q.w, az = q.z - q.w, bz = q.z + q.w;
  const float xmin = (ax <= bx) ? ax : bx, xmax = (ax <= bx) ? bx : ax;
  const float ymin = (ay <= by) ? ay : by, ymax = (ay <= by) ? by : ay;
  const float zmin = (az <= bz) ? az : bz, zmax = (az <= bz) ? bz : az;
  const float tx1 = (xmin - o.x) * inv.x, tx2 = (xmax - o.x) * inv.x;
  const float ty1 = (ymin - o.y) * inv.y, ty2 = (ymax - o.y) * inv.y;
  const float tz1 = (zmin - o.z) * inv.z, tz2 = (zmax - o.z) * inv.z;
  te = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
  tx = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
}

// A sphere reached through quantised (larger) boxes skipped part of the box test its leaf gets in the reference's walk.  Of the
// order-independent clauses of that test (bvh_traversal.cu:11-44) the one a hit sphere fails in exact arithmetic is
// `t_exit > t_min`: the ray leaves the sphere's box within 1e-4 of its origin (it starts inside an overlapping sphere, just
// under its surface) and the reference does not see the hit.  The far intersection t_far bounds the box's exit from below, so
// t_far comfortably above 1e-4 settles it; otherwise the box test is evaluated exactly as the reference does.  (What rounding
// can do to the other clauses is the business of hit_needs_literal_walk below.)
MIRT_DEV bool sphere_leaf_box_admits(const float4 q, const f3& o, const f3& d, float tc, float t_far)
{
  if (t_far > 0.0001f + 1e-5f * (fabsf(tc) + fabsf(q.w))) return true;
  float te, tx;
  sphere_leaf_box(q, o, mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), te, tx);
  return te < tx && tx > 0.0001f;
}

// The descent order at a node whose children are both hit.  The reference goes left first (bvh_traversal.cu:149-157).  Where the
// node's record allows it (NODE_SWAP_* & swap_mask) the child whose box the ray enters first is taken first instead: fewer
// visits, the same closest hit (see closer_hit).  Returns the two references in visiting order.
MIRT_DEV void order_children(bool hl, bool hr, float tel, float ter, uint32_t node_flags, uint32_t swap_mask, uint32_t& lref, uint32_t& rref)
{
  const bool swp = hl && hr && (node_flags & swap_mask) != 0u && ter < tel;
  const uint32_t l = lref;
  lref = swp ? rref : l;
  rref = swp ? l : rref;
}

// intersect_leaf_primitives' acceptance test (bvh_traversal.cu:66-84): hit, t > 1e-6, closer than the best so far -- and, on an
// exact tie in t, the primitive with the smaller sorted index wins.  The reference's left-first walk meets the primitives
// in sorted order and keeps the first of a tie (strict <); primitive records are stored in sorted order, so comparing the
// record offsets gives the same winner in whatever order the primitives are met.  (REF_NONE has offset bits 0: nothing
// ties with "no hit yet".)
MIRT_DEV bool closer_hit(bool hit, float t, float tbest, uint32_t off16, uint32_t refbest)
{
  return hit && t > 1e-6f && (t < tbest || (t == tbest && off16 < (refbest & REF_OFFMASK)));
}

// checkTriangleIntersectionSoA, struct.cu:111-163.  The early returns of the reference become one boolean: a wave with
// several lanes here never skips the code anyway, and values computed past a failed test (a division by ~0, a barycentric of
// a point behind the ray) are simply not selected.
MIRT_DEV bool triangle_hit(const float4 q0, const float4 q1, const float4 q2, const f3& o, const f3& d, float& t)
{
  const f3 p0 = mk3(q0.x, q0.y, q0.z), nor = mk3(q0.w, q1.x, q1.y);
  const float denom = dot(d, nor);
  t = dot(p0 - o, nor) / denom;
  const f3 ip = t * d + o;
  const f3 e1 = mk3(q1.z, q1.w, q2.x), e2 = mk3(q2.y, q2.z, q2.w);
  const float b1 = dot(e1, ip - p0);
  const float b2 = dot(e2, ip - p0);
  const float b0 = 1.0f - b1 - b2;
  return !(fabsf(denom) < 1e-9f) && !(t <= EPSILON) && (b0 >= -EPSILON) && (b1 >= -EPSILON) && (b2 >= -EPSILON);
}

// checkSphereIntersectionSoA, struct.cu:64-109 (same remark)
MIRT_DEV bool sphere_hit(const float4 q, const f3& o, const f3& d, float& t, float& tc_out, float& t_far)
{
  const f3 c = mk3(q.x, q.y, q.z);
  const float r = q.w;
  const f3 cr0 = c - o;
  const bool inside = (dot(cr0, cr0) < r * r);
  const float tc = dot(cr0, d);
  const f3 dv = o + (tc * d) - c;
  const float d2 = dot(dv, dv);
  const float toff = sqrtf((r * r) - d2);
  t = inside ? (tc + toff) : (tc - toff);
  tc_out = tc; t_far = tc + toff;
  return !(!inside && tc < 0.0f) && !(!inside && (r * r) < d2);
}

struct Counters { uint32_t samples, rays, shadow_rays, internal_visits, sphere_tests, tri_tests, mat_fetches, max_stack, traversed; };

// Everything a lane carries from one loop iteration to the next.
struct Lane {
  // sample being evaluated (g < 0: none)
  int g;              // index of the sample within this launch (a launch has < 2^31 samples: render_impl), -1: none
  Xorwow rng;
  f3 L;
  float alpha;
  uint32_t steps;     // traversal steps this sample has taken so far (scheduling statistic)
  // current shading node H: the ray that produced it and the hit
  f3 Hdir, Hp, Hn, Hcolor;
  int Hbounce;
  float Hior, Hrough;
  bool HtransNZ;
  f3 wt, wD, pn;
  int pc, refr_bounce, gi_n, state;
  // the node's ray batch: shadow rays to every light, then the reflection ray; all leave from bo
  f3 bo, rdir;
  int li;                      // index of the batch ray in flight: < nlights shadow, == nlights reflection
  unsigned long long occl;     // bit i: light i is occluded
  bool batch_pending, has_reflect;
  // ray in flight
  f3 o, d, inv;       // (with quantised nodes `inv` holds A = grid_step / d, qb / qc the near / far offsets of box_pair_q,
  f3 qb, qc;          //  qsx.. the near-plane selectors)
  uint32_t qsx, qsy, qsz;
  int bounce;
  float limit;        // shadow rays: occluded iff something is hit closer than this
  bool shadow;
  float tplane;
  int plane_id;
  // traversal
  bool trav;
  uint32_t cur;
  uint32_t tos;       // top of the traversal stack (entries below it live in LDS / the spill area)
  int sp;
  float tbest;
  uint32_t refbest;
};

// hitNearest's plane half (checkPlane, draw.cu:581-615) and the decision whether the BVH must be walked at all.
template <bool COUNT, bool HAVE_INV = false, bool QN = false, typename Args = RenderArgs>
MIRT_DEV void start_ray(const Args& a, Lane& S, Counters& cn)
{
  const bool shadow = S.shadow;      // (an any-hit shadow ray: batch_next)
  if (COUNT) cn.rays++;
  if (!HAVE_INV) S.inv = mk3(1.0f / S.d.x, 1.0f / S.d.y, 1.0f / S.d.z);
  if (QN) {
    // the ray in the grid of the quantised node records (uniform values: scalar loads)
    typedef const float __attribute__((address_space(4))) * ConstF;
    const ConstF qp = (ConstF)(unsigned long long)a.qparams;
    f3 A;
    quantised_axis(qp[0], qp[3], qp[6], S.o.x, S.inv.x, A.x, S.qb.x, S.qc.x, S.qsx);
    quantised_axis(qp[1], qp[4], qp[7], S.o.y, S.inv.y, A.y, S.qb.y, S.qc.y, S.qsy);
    quantised_axis(qp[2], qp[5], qp[8], S.o.z, S.inv.z, A.z, S.qb.z, S.qc.z, S.qsz);
    S.inv = A;
  }
  float tplane = INFINITY;
  int plane_id = -1;
  // (the planes are the same for every lane and never written by a kernel: read them through the constant address space,
  // i.e. with scalar loads, not through the per-lane address unit)
  typedef const PlaneDev __attribute__((address_space(4))) * ConstPlanes;
  const ConstPlanes planes = (ConstPlanes)(unsigned long long)a.planes;
  for (int i = 0; i < a.num_planes; ++i) {
    const f3 pnor = mk3(planes[i].nx, planes[i].ny, planes[i].nz);
    const float t = dot(mk3(planes[i].px, planes[i].py, planes[i].pz) - S.o, pnor) / dot(S.d, pnor);
    if (t <= 1e-6f) continue;
    if (t < tplane && t > EPSILON) { tplane = t; plane_id = i; }
  }
  if (tplane >= (float)(INT_MAX - 10)) { tplane = INFINITY; plane_id = -1; }
  S.tplane = tplane; S.plane_id = plane_id;
  S.tbest = INFINITY; S.refbest = REF_NONE;
  S.cur = a.root_ref; S.sp = 0;
  // a shadow ray the plane already blocks needs no traversal (same boolean as draw.cu:347-352 / 365-370)
  S.trav = (a.root_ref != REF_NONE) && !(shadow && plane_id >= 0 && tplane < S.limit);
  if (COUNT && S.trav) cn.traversed++;
}

MIRT_DEV void set_ray(Lane& S, const RayS& r) { S.o = r.o; S.d = r.d; S.bounce = r.bounce; }

// The walk of lane S ended with a sphere hit that is about to be shaded: would the reference's walk have found it?  Any order
// other than the reference's, and any boxes larger than its own, rest on one property of its box test (bvh_traversal.cu:11-44):
// a sphere that is hit passes the test of its leaf box and of every box above it, whatever the best distance so far -- the hit
// lies inside them.  In float arithmetic that fails within a few ulp of t: a hit distance that rounds to just below the entry
// distance of the sphere's own box (a ray that touches the sphere where the sphere touches its box, seen from so far away that
// ulp(t) is the size of the gap).  The reference then sees the sphere or not depending on what it found before; a walk in
// another order, or over larger boxes, can end with a hit the reference never tests.  So the hit a nearest-hit walk ends with
// must be one the reference provably reaches: its exact leaf box passes the order-independent clauses and is entered before
// the hit -- every box above contains the leaf box and the slab arithmetic is monotone in the box planes, so t_enter(above) <=
// t_enter(leaf) < t <= the best distance at the time that box was tested.  Otherwise the ray is walked again exactly as the
// reference does it (walk_literally below).  Two stages (hit_needs_literal_walk), at the start of the shade phase that consumes
// the walk:
// hit_at_risk -- the point o + T d lies inside the sphere's box by more than every rounding of the box test: with u = 2^-24,
// per axis  T - t_enter >= ((r - |v|)(1 - 2.01u) - 2.1u(|c| + r + |o|)) / |d| - 2.01u T  for v = o + T d - c evaluated within
// 3u(|o| + T + |c|), hence  r - |v| > 8u(|o| + |c| + r + T)  suffices, and |c| + r is at most the largest coordinate of the
// scene box (`slack` = 2^-21 of it) -- says "provably reached" for all but a few hits per million; for the others the leaf
// box test itself is evaluated.  A shadow ray to a sun needs none of this: it is occluded iff any sphere is hit, in any order
// (the reference culls by distance only once it has a hit).  Towards a point light "occluded" means nearer than the light, and a
// hit the reference never tests can be the one that is: from a point on an infinite plane 10^8 units away every sphere near the
// light is within an ulp of the light's distance (the fuzzer again: seed 72, scene 2868).  Those rays are traced to their
// nearest hit and vetted like the others (batch_next; the loop header leaves their results to the shade phase).
// What stays an assumption is the mirror image: that no walk of the product CULLS a box over a sphere whose computed hit distance
// lies below that box's entry distance and below the best distance of the moment, while the reference -- in its order -- gets
// there first.  Over the quantised boxes that takes a disagreement of more than a grid step plus their 8-ulp margin; over the
// exact boxes, near child first, one ulp.  traversal = 0 assumes nothing; the fuzzer compares the two.
// Found by the mode fuzzer (tools/fuzz_modes.py seed 47, scene 795: one ray in 2.7e8; tests/golden/far_camera_tie.txt).
MIRT_DEV bool hit_at_risk(const float4 q, const f3& o, const f3& d, float t, float slack)
{
  const float vx = __builtin_fmaf(t, d.x, o.x) - q.x, vy = __builtin_fmaf(t, d.y, o.y) - q.y, vz = __builtin_fmaf(t, d.z, o.z) - q.z;
  const float m = fmaxf(fmaxf(fabsf(vx), fabsf(vy)), fabsf(vz));
  const float margin = __builtin_fmaf(((fabsf(o.x) + fabsf(o.y)) + fabsf(o.z)) + t, 4.76837158203125e-07f, slack);
  return !(q.w - m > margin);
}

// The vetting of a lane whose finished walk is about to be shaded (advance): true = walk the ray again.
template <bool QN>
MIRT_DEV bool hit_needs_literal_walk(const RenderArgs& a, const Lane& S)
{
  // (nothing to vet: no sphere hit, or the plane is nearer; the reference's own order over its own boxes; a ray that was already
  // walked again; a shading node without a reflection ray -- nothing was traced for it, advance_core's no_ray)
  if (S.refbest == REF_NONE || (S.refbest & REF_TRI) != 0u || !a.reach_check || (QN && S.qsx == 0u)) return false;
  if ((S.plane_id >= 0 && !(S.tbest < S.tplane)) || (!S.batch_pending && S.state == ST_BATCH && !S.has_reflect)) return false;
  const float4 q = a.nodes[S.refbest & REF_OFFMASK];
  if (!hit_at_risk(q, S.o, S.d, S.tbest, a.reach_slack)) return false;
  float te, tx;
  sphere_leaf_box(q, S.o, mk3(1.0f / S.d.x, 1.0f / S.d.y, 1.0f / S.d.z), te, tx);
  return !(te < tx && tx > 0.0001f && te < S.tbest);
}

// A light the (rough) shading normal faces away from contributes colour * light * max(dot, 0) = 0 whether it is occluded or
// not (draw.cu:353-357, 371-374), so its shadow ray is answered without a traversal: it still counts as a ray, and the light
// is left "not occluded" (its term is then computed as usual and is 0).  Needs every colour of the scene to be finite
// (0 * colour == 0; checked at scene creation) and at most 32 lights (the mask shares a word pair with the occlusion bits).
// Kernel specialisations by scene (template parameter SPEC of the shading functions): what a scene does not have is left out of
// its kernel at compile time -- less code in the divergent shading phase and, above all, fewer live registers.
//   SPEC_NOTRI   no triangles (the scenes with quantised nodes)
//   SPEC_NOBULB  no point lights
//   SPEC_NOPEND  no transparent material and gi 0: no refraction states, no pending-children list
constexpr int SPEC_NOTRI = 1, SPEC_NOBULB = 2, SPEC_NOPEND = 4;

template <typename Args, int SPEC = 0>
MIRT_DEV uint32_t unlit_mask(const Args& a, const f3& pn, const f3& Hp)
{
  uint32_t m = 0u;
  if (a.skip_unlit) {
    for (int li = 0; li < a.num_suns; ++li) {
      const LightDev& lt = a.suns[li];
      if (!(dot(pn, mk3(lt.nx, lt.ny, lt.nz)) > 0.0f)) m |= 1u << li;
    }
    if (!(SPEC & SPEC_NOBULB)) for (int li = 0; li < a.num_bulbs; ++li) {
      const LightDev& lt = a.bulbs[li];
      if (!(dot(pn, normalize(mk3(lt.x, lt.y, lt.z) - Hp)) > 0.0f)) m |= 1u << (a.num_suns + li);
    }
  }
  return m;
}

// The ray of the batch that was in flight has finished: note a shadow result, start the next ray of the batch
// (shadow rays of diffuseLight, draw.cu:342-374, then the reflection ray of reflectionLight, draw.cu:402-404).
// Shadow rays consume no random numbers, so tracing them after the reflection direction was drawn changes nothing.
template <bool COUNT, bool QN = false, typename Args = RenderArgs, int SPEC = 0>
MIRT_DEV void batch_next(const Args& a, Lane& S, Counters& cn)
{
  constexpr bool NOBULB = (SPEC & SPEC_NOBULB) != 0;
  const int nlights = NOBULB ? a.num_suns : a.num_suns + a.num_bulbs;
  if (S.li >= 0 && S.li < nlights) {
    const bool occluded = (S.plane_id >= 0 && S.tplane < S.limit) || (S.refbest != REF_NONE && S.tbest < S.limit);
    if (occluded) S.occl |= 1ull << S.li;
  }
  // the lanes of a wave are at different rays of their batches: each kind only sets the ray up, and all of them share
  // one start_ray (one copy of the plane loop instead of three executed one after the other)
  bool go = true;
  // next light whose shadow ray is traced: the upper half of S.occl marks the lights the shading normal faces away from
  // (unlit_mask, set when the node was entered); their rays count as rays but visit nothing
  {
    const uint32_t unlit = (nlights <= 32) ? (uint32_t)(S.occl >> 32) : 0u;      // (with more lights those bits are occlusion bits)
    const int from = S.li + 1;
    const uint32_t rest = (from < 32) ? (unlit | ((1u << from) - 1u)) : 0xffffffffu;       // lights before `from` are done
    const int nxt = (~rest != 0u) ? (int)__builtin_ctz(~rest) : 32;
    S.li = (unlit != 0u && from < nlights) ? min(nxt, nlights) : from;                     // (from > nlights after the reflection ray: the batch is over)
    if (COUNT && unlit) { const int k = __popc(unlit & ~((from < 32) ? ((1u << from) - 1u) : 0xffffffffu) & ((S.li < 32) ? ((1u << S.li) - 1u) : 0xffffffffu)); cn.rays += k; cn.shadow_rays += k; }
  }
  if (S.li < nlights) {
    // shadow ray, draw.cu:346 / 362-363
    S.o = S.bo;
    S.limit = INFINITY;
    // shadow_anyhit = 0: the ray is traced to its nearest hit like any other, exactly as hitNearest does for diffuseLight
    // (draw.cu:347-352, 365-370); the occlusion test above reads the same boolean off the result either way
    // (towards a point light, in a walk that is not the reference's own, the ray is traced to its nearest hit all the same and
    // that hit is vetted like any other -- hit_needs_literal_walk: "occluded" there means nearer than the light)
    S.shadow = a.shadow_anyhit != 0 && (NOBULB || S.li < a.num_suns || !a.reach_check);
    S.bounce = 1;
    if (COUNT) cn.shadow_rays++;
    if (NOBULB || S.li < a.num_suns) {
      // direction and its reciprocal are per-light constants (host-computed, same arithmetic)
      const LightDev& lt = a.suns[S.li];
      S.d = mk3(lt.nx, lt.ny, lt.nz); S.inv = mk3(lt.ix, lt.iy, lt.iz);
    } else {
      const LightDev& lt = a.bulbs[S.li - a.num_suns];
      const f3 bd = mk3(lt.x, lt.y, lt.z) - S.Hp;
      S.d = mkray(S.bo, bd, 1).d;
      S.inv = mk3(1.0f / S.d.x, 1.0f / S.d.y, 1.0f / S.d.z);
      S.limit = length(bd);
    }
  } else if (S.li == nlights && S.has_reflect) {
    S.o = S.bo; S.d = S.rdir; S.bounce = S.Hbounce - 1;
    S.inv = mk3(1.0f / S.d.x, 1.0f / S.d.y, 1.0f / S.d.z);
    S.limit = INFINITY;
    S.shadow = false;
  } else {
    go = false;
    S.batch_pending = false;
    S.shadow = false;
    S.trav = false;
  }
  if (go) start_ray<COUNT, true, QN>(a, S, cn);
}

// The pending-children LIFO of a lane (refraction / gi children waiting for their turn): one 64-byte entry per slot, a lane's
// slots next to each other -- a push writes one cache line and the pop that follows reads it back from the L2 (with the words
// of an entry strided by the grid size, as in rounds 1-2, each of an entry's eleven words dirtied a line of its own).
MIRT_DEV float4* pending_entry(const RenderArgs& a, const long long gid, const int slot)
{
  return reinterpret_cast<float4*>(a.pending) + ((size_t)gid * (size_t)a.pending_slots + (size_t)slot) * (PENDING_WORDS / 4);
}

// The ray of lane S again, exactly as the reference walks it (traverse_lbvh, bvh_traversal.cu:92-183: the exact 64-byte records
// from node 0 at heap offset 0, left child first, nearest hit), in place and by this lane alone: the shade phase found that its
// hit may be one the reference never tests (hit_at_risk above; a few rays per 10^7).  The lane's traversal
// stack is empty at this point: the walk keeps its own in the lane's column of the spill area.
template <bool COUNT, bool NOTRI>
MIRT_DEV void walk_literally(const RenderArgs& a, Lane& S, Counters& cn, const long long gid, const long long gthreads)
{
  uint32_t* const stack = a.stack_spill;
  const f3 inv = mk3(1.0f / S.d.x, 1.0f / S.d.y, 1.0f / S.d.z);
  const float tmin = 0.0001f;
  float tbest = INFINITY;
  uint32_t refbest = REF_NONE, cur = 0u;
  int sp = 0;
  for (;;) {
    const float4* rec = a.nodes + (cur & REF_OFFMASK);
    bool pop;
    if (cur & REF_LEAF) {
      float t = 0.0f, tc, t_far;
      bool hit;
      if (!NOTRI && (cur & REF_TRI)) { if (COUNT) cn.tri_tests++; hit = triangle_hit(rec[0], rec[1], rec[2], S.o, S.d, t); }
      else { if (COUNT) cn.sphere_tests++; hit = sphere_hit(rec[0], S.o, S.d, t, tc, t_far); }
      if (closer_hit(hit, t, tbest, cur & REF_OFFMASK, refbest)) { tbest = t; refbest = cur; }
      pop = true;
    } else {
      if (COUNT) cn.internal_visits++;
      // hit_aabb_adapted on the left child, then on the right one (one box at a time: this code runs a few times per frame and
      // must not cost the shade phase around it a register); the record: left x, y | left z, right x | right y, z | references
      bool hc[2];
#pragma unroll 1
      for (int c = 0; c < 2; ++c) {
        const float2* b = reinterpret_cast<const float2*>(rec) + 3 * c;
        const float2 bx = b[0], by = b[1], bz = b[2];
        const float tx1 = (bx.x - S.o.x) * inv.x, tx2 = (bx.y - S.o.x) * inv.x;
        const float ty1 = (by.x - S.o.y) * inv.y, ty2 = (by.y - S.o.y) * inv.y;
        const float tz1 = (bz.x - S.o.z) * inv.z, tz2 = (bz.y - S.o.z) * inv.z;
        const float te = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
        const float tx = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
        hc[c] = te < tx && te < tbest && tx > tmin;
      }
      const uint2 ch = *reinterpret_cast<const uint2*>(rec + 3);
      if (hc[0] && hc[1]) {
        stack[(size_t)sp * gthreads + gid] = ch.y;
        ++sp;
        if (COUNT) cn.max_stack = max(cn.max_stack, (uint32_t)sp);
      }
      cur = hc[0] ? ch.x : ch.y;
      pop = !(hc[0] || hc[1]);
    }
    if (pop) {
      if (sp == 0) break;
      --sp;
      cur = stack[(size_t)sp * gthreads + gid];
    }
  }
  S.tbest = tbest; S.refbest = refbest;
}

// Consume the finished trace of lane S and run its shading state machine until it either has the next ray(s) or the
// sample is complete.  Returns M_DONE (sample finished: S.L / S.alpha hold the RGBA), M_BATCH (a new shading node was
// entered: S.bo, S.rdir, S.has_reflect describe its ray batch, S.li = -1) or M_TRACE (a single ray is in S.o/S.d/S.bounce).
// `gid` names this lane's pending-children LIFO (pending_entry).
template <bool COUNT, int SPEC = 0>
MIRT_DEV int advance_core(const RenderArgs& a, Lane& S, Counters& cn, const long long gid, const long long gthreads)
{
  constexpr bool NOTRI = (SPEC & SPEC_NOTRI) != 0, NOBULB = (SPEC & SPEC_NOBULB) != 0, NOPEND = (SPEC & SPEC_NOPEND) != 0;
  (void)gthreads;
  const int nlights = NOBULB ? a.num_suns : a.num_suns + a.num_bulbs;
  int micro = M_TRACE;
  const bool bvh_hit = S.refbest != REF_NONE;
  const bool pl_hit = S.plane_id >= 0;
  const f3 rd0 = S.d, ro0 = S.o;
  bool no_ray = false;     // the node had no reflection ray: nothing was traced, treat as a miss
  if (S.state == ST_BATCH) {
    // diffuseLight's light loops (draw.cu:342-374) with the occlusion bits the batch collected
    f3 Dacc = mk3(0.0f, 0.0f, 0.0f);
    for (int li = 0; li < nlights; ++li) {
      if ((S.occl >> li) & 1ull) continue;
      if (NOBULB || li < a.num_suns) {
        const LightDev& lt = a.suns[li];
        const float lambert = fmaxf(dot(S.pn, mk3(lt.nx, lt.ny, lt.nz)), 0.0f);
        const float r = S.Hcolor.x * (lt.r * lambert), gg = S.Hcolor.y * (lt.g * lambert), b = S.Hcolor.z * (lt.b * lambert);
        Dacc = Dacc + mk3(set_expose(r, a.expose), set_expose(gg, a.expose), set_expose(b, a.expose));
      } else {
        const LightDev& lt = a.bulbs[li - a.num_suns];
        const f3 bd = mk3(lt.x, lt.y, lt.z) - S.Hp;
        const float lambert = fmaxf(dot(S.pn, normalize(bd)), 0.0f);
        const float tl = length(bd);
        const float inv2 = 1.0f / (tl * tl);
        const float r = S.Hcolor.x * (lt.r * lambert), gg = S.Hcolor.y * (lt.g * lambert), b = S.Hcolor.z * (lt.b * lambert);
        Dacc = Dacc + mk3(set_expose(r, a.expose) * inv2, set_expose(gg, a.expose) * inv2, set_expose(b, a.expose) * inv2);
      }
    }
    S.L = S.L + S.wD * Dacc;
    no_ray = !S.has_reflect;
  }
  {
    // hitNearest, draw.cu:292-318: the nearer of BVH hit and plane hit (the plane wins an exact tie)
    const bool use_bvh = !no_ray && bvh_hit && (!pl_hit || S.tbest < S.tplane);
    const bool hit = !no_ray && (bvh_hit || pl_hit);
    f3 Np = mk3(0, 0, 0), Nn = mk3(0, 0, 0);
    Mat nm;
    nm.color = mk3(0, 0, 0); nm.shininess = mk3(0, 0, 0); nm.trans = mk3(0, 0, 0); nm.ior = 1.458f; nm.roughness = 0.0f;
    if (use_bvh) {
      // the hit primitive's record (heap, sorted order) and its index in the scene's arrays (for the material)
      const uint32_t off16 = S.refbest & REF_OFFMASK;
      const float4* rec = a.nodes + off16;
      const uint32_t id = a.unit_prim[off16 - a.prim_base16] & 0x7fffffffu;
      Np = S.tbest * rd0 + ro0;
      if (!NOTRI && (S.refbest & REF_TRI)) {
        const float4 q0 = rec[0], q1 = rec[1];
        const f3 nor = mk3(q0.w, q1.x, q1.y);
        const float denom = dot(rd0, nor);
        Nn = (denom < 0.0f) ? nor : -nor;
        nm = load_mat(a.mats, (uint32_t)a.num_spheres + id);
      } else {
        const float4 s = rec[0];
        const f3 c = mk3(s.x, s.y, s.z);
        const f3 cr0 = c - ro0;
        const bool inside = (dot(cr0, cr0) < s.w * s.w);
        Nn = normalize(inside ? (c - Np) : (Np - c));
        nm = load_mat(a.mats, id);
      }
      if (COUNT) cn.mat_fetches++;
    } else if (hit) {
      const PlaneDev& pl = a.planes[S.plane_id];
      const f3 pnor = mk3(pl.nx, pl.ny, pl.nz);
      Np = S.tplane * rd0 + ro0;
      Nn = (dot(pnor, rd0) < 0.0f) ? pnor : -pnor;
      nm = plane_mat(pl);
    }

    if (!NOPEND && S.state == ST_REFR_INSIDE) {
      // second half of refractionLight, draw.cu:484-493.  No miss check: a miss yields the default ObjectInfo
      // (normal 0, ior 1.458, point 0), which is what Np/Nn/nm hold then.
      const f3 normal = normalize(Nn);
      const float ior = nm.ior;
      const float dn = dot(normal, rd0);
      const float k = 1.0f - ior * ior * (1.0f - (dn * dn));
      const f3 rd = ior * rd0 - (ior * (dot(normal, rd0)) + sqrtf(k)) * normal;
      set_ray(S, mkray(Np - normal * 0.0001f, rd, S.refr_bounce - 1));
      S.state = ST_REFR_FINAL;
      micro = (S.bounce == 0) ? M_POP : M_TRACE;
    } else if (!hit) {
      // primary miss: RGBA(0,0,0,0) (draw.cu:267,284).  A secondary miss contributes nothing to rgb.
      micro = (S.state == ST_PRIMARY) ? M_DONE : M_POP;
    } else {
      // refraction arguments X of the node being entered: the parent's H after a reflection (draw.cu:424), else H itself
      f3 Xdir, Xp, Xn;
      int Xbounce;
      float Xior;
      bool XtransNZ, x_parent = false, has_gi = false;
      if (S.state == ST_PRIMARY) { S.alpha = 1.0f; S.wt = mk3(1.0f, 1.0f, 1.0f); has_gi = true; S.gi_n = NOPEND ? 0 : a.gi; }
      else if (S.state == ST_BATCH) x_parent = true;
      else if (S.state == ST_GI) has_gi = true;     // gi_n was set when the ray was made
      Xdir = S.Hdir; Xbounce = S.Hbounce; Xp = S.Hp; Xn = S.Hn; Xior = S.Hior; XtransNZ = S.HtransNZ;
      S.Hdir = rd0; S.Hbounce = S.bounce; S.Hp = Np; S.Hn = Nn;
      S.Hcolor = nm.color; S.Hior = nm.ior; S.Hrough = nm.roughness; S.HtransNZ = !is_black(nm.trans);
      if (!x_parent) { Xdir = S.Hdir; Xbounce = S.Hbounce; Xp = S.Hp; Xn = S.Hn; Xior = S.Hior; XtransNZ = S.HtransNZ; }
      // weights of this node's terms (draw.cu:277-281, 426-428, 517-519, 561-563)
      const f3 one = mk3(1.0f, 1.0f, 1.0f);
      const f3 Sh = nm.shininess, T = nm.trans;
      const f3 K = (one - Sh) * (one - T);
      // (a full pending list would drop the child: counted, and mirt_get_stats reports it as an error -- the list is sized for
      // the deepest chain the scene's bounces / gi allow, so this does not happen)
      if (NOPEND) { has_gi = false; XtransNZ = false; }      // (gi 0 and no transparent material: nothing is ever pending)
      if (has_gi && a.gi != 0 && S.gi_n != 0 && S.pc >= a.pending_slots) atomicAdd(a.overflow, 1ull);
      if (XtransNZ && Xbounce > 0 && S.pc + ((has_gi && a.gi != 0 && S.gi_n != 0) ? 1 : 0) >= a.pending_slots) atomicAdd(a.overflow, 1ull);
      if (has_gi && a.gi != 0 && S.gi_n != 0 && S.pc < a.pending_slots) {
        const f3 w = (S.wt * K) * nm.color;
        float4* e = pending_entry(a, gid, S.pc);
        e[0] = make_float4(__uint_as_float(PEND_G), S.Hp.x, S.Hp.y, S.Hp.z);
        e[1] = make_float4(S.Hn.x, S.Hn.y, S.Hn.z, __int_as_float(S.gi_n));
        e[2] = make_float4(w.x, w.y, w.z, 0.0f);
        ++S.pc;
      }
      if (XtransNZ && Xbounce > 0 && S.pc < a.pending_slots) {
        const f3 w = S.wt * ((one - Sh) * T);
        float4* e = pending_entry(a, gid, S.pc);
        e[0] = make_float4(__uint_as_float(PEND_F), Xp.x, Xp.y, Xp.z);
        e[1] = make_float4(Xn.x, Xn.y, Xn.z, __int_as_float(Xbounce));
        e[2] = make_float4(w.x, w.y, w.z, Xdir.x);
        e[3] = make_float4(Xdir.y, Xdir.z, Xior, 0.0f);
        ++S.pc;
      }
      S.wD = S.wt * K;
      S.wt = S.wt * Sh;
      const bool reflect_ok = !is_black(Sh) && S.Hbounce > 0;
      // diffuseLight prologue, draw.cu:331-340
      S.pn = S.Hn;
      if (S.Hrough > 0.0f) S.pn = rough_normal(S.Hn, S.Hrough, S.rng);
      S.pn = normalize(S.pn);
      // reflectionLight prologue, draw.cu:389-402 (its draws follow diffuseLight's; the shadow rays in between draw nothing)
      S.has_reflect = false;
      if (reflect_ok) {
        f3 normal = S.Hn;
        if (S.Hrough > 0.0f) normal = rough_normal(S.Hn, S.Hrough, S.rng);
        normal = normalize(normal);
        S.rdir = normalize(S.Hdir - 2.0f * (dot(normal, S.Hdir)) * normal);
        S.has_reflect = (S.Hbounce - 1) != 0;      // a bounce-0 ray never hits (draw.cu:294)
      }
      S.bo = S.Hp + S.Hn * EPSILON;
      S.occl = (unsigned long long)unlit_mask<RenderArgs, SPEC>(a, S.pn, S.Hp) << 32;
      S.li = -1;
      S.batch_pending = true;
      S.state = ST_BATCH;
      micro = M_BATCH;
    }
  }

  // run the micro-states until this lane has a ray or is finished
  while (micro == M_POP) {
    {
      if (NOPEND || S.pc == 0) micro = M_DONE;
      else {
        --S.pc;
        const float4* e = pending_entry(a, gid, S.pc);
        const float4 e0 = e[0], e1 = e[1], e2 = e[2];
        const uint32_t tag = __float_as_uint(e0.x);
        const f3 p = mk3(e0.y, e0.z, e0.w);
        const f3 n = mk3(e1.x, e1.y, e1.z);
        const int ib = __float_as_int(e1.w);
        S.wt = mk3(e2.x, e2.y, e2.z);
        if (tag == PEND_G) {
          // globalIllumination, draw.cu:540-549
          const f3 gi_dir = normalize(n + sphere_point(S.rng));
          set_ray(S, mkray(p + n * EPSILON, gi_dir, ib - 1));
          S.gi_n = ib - 1;
          S.state = ST_GI;
          micro = (S.bounce == 0) ? M_POP : M_TRACE;
        } else {
          // refractionLight, draw.cu:456-480
          const float4 e3 = e[3];
          const f3 dir = mk3(e2.w, e3.x, e3.y);
          const float ior = 1.0f / e3.z;
          const f3 normal = normalize(n);
          const float dn = dot(normal, dir);
          const float k = 1.0f - (ior * ior) * (1.0f - (dn * dn));
          if (k < 0) {
            const f3 rd = dir - 2.0f * (dot(normal, dir)) * normal;
            set_ray(S, mkray(p + normal * EPSILON, rd, ib - 1));
            S.state = ST_REFR_FINAL;
            micro = (S.bounce == 0) ? M_POP : M_TRACE;
          } else {
            const f3 rd = ior * dir - (ior * (dot(normal, dir)) + sqrtf(k)) * normal;
            set_ray(S, mkray(p - normal * 0.0001f, rd, ib));
            S.refr_bounce = ib;
            S.state = ST_REFR_INSIDE;
            micro = M_TRACE;   // ib > 0 is guaranteed by the push condition
          }
        }
      }
    }
  }
  return micro;
}

// Cost class of a sample for the longest-first hand-out (sched = 2): about four classes per doubling of its traversal steps; the
// key sorts ascending, so 255 - class puts the expensive ones first.  0 steps (a bounce-0 frame): last.
MIRT_DEV uint32_t cost_key(uint32_t steps)
{
  if (steps == 0) return 255u;
  const int e = 31 - __clz((int)steps);
  const uint32_t frac = e >= 2 ? (steps >> (e - 2)) & 3u : 0u;
  return 255u - (uint32_t)(4 * e) - frac;
}

// Megakernel form: consume the finished trace, shade, and start the next ray of this lane (or finish the sample).
template <bool COUNT, bool QN = false, int SPEC = 0>
MIRT_DEV void advance(const RenderArgs& a, Lane& S, Counters& cn, const long long gid, const long long gthreads)
{
  const int micro = advance_core<COUNT, SPEC>(a, S, cn, gid, gthreads);
  if (micro == M_DONE) {
    a.samples[S.g] = make_float4(S.L.x, S.L.y, S.L.z, S.alpha);
    // scheduling statistic: the chunk's most expensive sample.  A plain (possibly stale, never too large) load first: most
    // samples are not their chunk's maximum and issue no atomic (one atomic per sample was 0.76 GB of write traffic per frame)
    if (COUNT && a.chunk_cost) {
      uint32_t* cc = &a.chunk_cost[S.g >> a.chunk_shift];
      if (S.steps > *cc) atomicMax(cc, S.steps);
    }
    if (COUNT && a.sample_key) a.sample_key[S.g] = cost_key(S.steps);
    S.g = -1;
    S.trav = false;
  } else if (micro == M_BATCH) {
    batch_next<COUNT, QN, RenderArgs, SPEC>(a, S, cn);
  } else {
    S.shadow = false;
    S.limit = INFINITY;
    start_ray<COUNT, false, QN>(a, S, cn);
  }
}

// Start sample `idx` on this lane: pixel, RNG stream, jitter, primary ray (draw.cu:105-123 / 162-171).
template <bool COUNT, int TABLES = 0>
MIRT_DEV void init_sample_core(const RenderArgs& a, Lane& S, Counters& cn, const long long idx)
{
  // sample of this launch -> local pixel of the part -> frame pixel, in 32-bit arithmetic (a launch has < 2^31 samples, a
  // part < 2^31 pixels: checked by render())
  const uint32_t sppe = (uint32_t)a.sample_count;
  const uint32_t stripe_pixels = (uint32_t)a.stripe_rows * (uint32_t)a.width;
  const uint32_t i32 = (uint32_t)idx;
  const uint32_t lq = i32 / sppe;                          // pixel within the launch
  const uint32_t lp = (uint32_t)a.pixel_base + lq;         // pixel within the part
  const int sidx = a.sample_first + (int)(i32 - lq * sppe);
  const uint32_t ls = lp / stripe_pixels;
  const uint32_t within = lp - ls * stripe_pixels;
  const uint32_t gs = ls * (uint32_t)a.num_parts + (uint32_t)a.part;
  const uint32_t wy = within / (uint32_t)a.width;
  const int py = (int)(gs * (uint32_t)a.stripe_rows + wy);
  const int px = (int)(within - wy * (uint32_t)a.width);
  const uint32_t pixel = (uint32_t)py * (uint32_t)a.width + (uint32_t)px;
  S.g = (int)idx;
  S.steps = 0;
  S.L = mk3(0.0f, 0.0f, 0.0f);
  S.alpha = 0.0f;
  S.pc = 0;
  S.state = ST_PRIMARY;
  S.limit = INFINITY;
  S.shadow = false;
  S.batch_pending = false;
  if (a.needs_rng) xw_init<TABLES>(S.rng, a.rng, pixel, a.seed_per_pixel ? (uint32_t)sidx : 0u);
  float fx = (float)px, fy = (float)py;
  if (a.spp >= 1) {
    const float jx = randD(-0.5f, 0.5f, S.rng);
    const float jy = randD(-0.5f, 0.5f, S.rng);
    fx = (float)px + jx; fy = (float)py + jy;
  }
  set_ray(S, primary_ray(a, fx, fy, S.rng));
  if (COUNT) cn.samples++;
}

template <bool COUNT, int TABLES = 0, bool QN = false>
MIRT_DEV void init_sample(const RenderArgs& a, Lane& S, Counters& cn, const long long idx)
{
  init_sample_core<COUNT, TABLES>(a, S, cn, idx);
  if (S.bounce == 0) {   // hitNearest: a ray with bounce 0 never hits (draw.cu:294)
    a.samples[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (COUNT && a.sample_key) a.sample_key[idx] = 255u;
    S.g = -1;
    S.trav = false;
  } else {
    start_ray<COUNT, false, QN>(a, S, cn);
  }
}

} // namespace
} // namespace mirt
#endif
