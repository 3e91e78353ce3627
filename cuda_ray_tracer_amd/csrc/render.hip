// Render on the device: replaces render() and the render kernels of draw.cu:94-239 together with the device code
// they call (shootPrimaryRay/hitNearest/diffuseLight/reflectionLight/refractionLight/globalIllumination/checkPlane,
// draw.cu:260-659; traverse_lbvh, bvh_traversal.cu:11-183; Ray::Ray and the primitive tests, struct.cu:16-163).
//
// trace_kernel    one lane = one sample.  A persistent grid walks the samples; every lane runs a small state machine
//                 (the reference's mutually recursive shading functions turned into an explicit ray-tree walk) around ONE
//                 shared BVH traversal loop, so that primary, shadow, reflection, refraction and GI rays of different
//                 lanes are traversed together.  Traversal stack: 32 entries per lane in LDS ([entry][lane], conflict
//                 free), spilling to global memory above that.  Nodes are 64-byte two-child records.
// resolve_kernel  per pixel: sum the samples in the reference's xor-butterfly order (draw.cu:181-189), mean, sRGB,
//                 quantise (draw.cu:129-132 for spp <= 1, draw.cu:9-11,202-205 otherwise).
//
// The ray tree is evaluated top-down (every ray carries the product of the mixing weights above it) instead of the
// reference's bottom-up recursion; geometry and random-number consumption are identical, colours agree to rounding.
#include "scene_dev.h"
#include "host_scene.h"

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace mirt {
namespace {

constexpr int RBLOCK = 256;
constexpr int STACK_LDS = 32;
constexpr int STACK_TOTAL = 64;      // TRAVERSAL_STACK_SIZE, bvh_traversal.cu:8
constexpr int PENDING_WORDS = 16;

constexpr float EPSILON = 0.001f;    // draw.cu:7, struct.cu:8

enum : int { ST_PRIMARY = 0, ST_SHADOW, ST_REFLECT, ST_REFR_INSIDE, ST_REFR_FINAL, ST_GI };
enum : int { M_ENTER = 0, M_LIGHT, M_REFLECT, M_POP, M_TRACE, M_DONE };
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

struct Counters { uint32_t samples, rays, shadow_rays, internal_visits, sphere_tests, tri_tests, mat_fetches, max_stack; };

template <bool COUNT>
__global__ void __launch_bounds__(RBLOCK) trace_kernel(const RenderArgs a)
{
  __shared__ uint32_t lds_stack[STACK_LDS * RBLOCK];
  const int tid = threadIdx.x;
  const long long gid = (long long)blockIdx.x * RBLOCK + tid;
  const long long gthreads = (long long)gridDim.x * RBLOCK;
  const int sppe = a.spp > 1 ? a.spp : 1;
  const long long stripe_pixels = (long long)a.stripe_rows * a.width;
  const int nlights = a.num_suns + a.num_bulbs;
  Counters cn = {0, 0, 0, 0, 0, 0, 0, 0};

  // every wave walks the sample range in steps of the grid size; lanes of a wave hold consecutive samples
  const long long wave_first = gid - (tid & 63);
  for (long long g0 = wave_first; g0 < a.num_samples; g0 += gthreads) {
    const long long g = g0 + (tid & 63);
    bool alive = g < a.num_samples;

    // ---- sample -> pixel ------------------------------------------------------------------------
    int px = 0, py = 0, sidx = 0;
    uint32_t pixel = 0;
    if (alive) {
      const long long lp = g / sppe;
      sidx = (int)(g - lp * sppe);
      const long long ls = lp / stripe_pixels;
      const long long within = lp - ls * stripe_pixels;
      const long long gs = ls * a.num_parts + a.part;
      py = (int)(gs * a.stripe_rows + within / a.width);
      px = (int)(within % a.width);
      pixel = (uint32_t)py * (uint32_t)a.width + (uint32_t)px;
    }

    // ---- per-sample state -------------------------------------------------------------------------
    Xorwow rng;
    rng.v0 = rng.v1 = rng.v2 = rng.v3 = rng.v4 = rng.d = 0; rng.bm_flag = 0; rng.bm_extra = 0.0f;
    f3 L = mk3(0.0f, 0.0f, 0.0f);
    float alpha = 0.0f;
    // current shading node H (ray that produced it + hit)
    f3 Hdir = mk3(0, 0, 0), Hp = mk3(0, 0, 0), Hn = mk3(0, 0, 0), Hcolor = mk3(0, 0, 0);
    int Hbounce = 0;
    float Hior = 1.458f, Hrough = 0.0f;
    bool HtransNZ = false, reflect_ok = false;
    // refraction arguments X of the node being entered (the parent's H after a reflection, else H itself)
    f3 Xdir = mk3(0, 0, 0), Xp = mk3(0, 0, 0), Xn = mk3(0, 0, 0);
    int Xbounce = 0;
    float Xior = 1.458f;
    bool XtransNZ = false, x_parent = false;
    bool has_gi = false;
    int gi_n = 0;
    f3 wt = mk3(1.0f, 1.0f, 1.0f), wD = mk3(0, 0, 0), Dacc = mk3(0, 0, 0), pn = mk3(0, 0, 0);
    int li = 0, pc = 0, refr_bounce = 0;
    // the ray to trace next
    RayS ray; ray.o = mk3(0, 0, 0); ray.d = mk3(0, 0, 1); ray.bounce = 0;
    float limit = INFINITY;     // shadow rays: occluded iff something is hit closer than this
    int state = ST_PRIMARY;
    bool want = false;          // this lane has a ray for the trace phase

    if (alive) {
      if (a.needs_rng) xw_init(rng, a.rng, pixel, (uint32_t)sidx);
      float fx = (float)px, fy = (float)py;
      if (a.spp >= 1) {   // draw.cu:120-121 / 165-168
        const float jx = randD(-0.5f, 0.5f, rng);
        const float jy = randD(-0.5f, 0.5f, rng);
        fx = (float)px + jx; fy = (float)py + jy;
      }
      ray = primary_ray(a, fx, fy, rng);
      if (COUNT) cn.samples++;
      if (ray.bounce == 0) alive = false;   // hitNearest: bounce 0 never hits (draw.cu:294)
      else want = true;
    }

    while (__ballot(alive)) {
      // =========================== trace phase (all lanes together) ===========================
      float tbest = INFINITY;
      uint32_t refbest = REF_NONE;     // leaf reference of the best BVH hit
      float tplane = INFINITY;
      int plane_id = -1;
      const bool shadow = (state == ST_SHADOW);
      bool trav = false;
      if (want) {
        if (COUNT) { cn.rays++; if (shadow) cn.shadow_rays++; }
        // checkPlane, draw.cu:581-615
        for (int i = 0; i < a.num_planes; ++i) {
          const PlaneDev& pl = a.planes[i];
          const f3 pnor = mk3(pl.nx, pl.ny, pl.nz);
          const float t = dot(mk3(pl.px, pl.py, pl.pz) - ray.o, pnor) / dot(ray.d, pnor);
          if (t <= 1e-6f) continue;
          if (t < tplane && t > EPSILON) { tplane = t; plane_id = i; }
        }
        if (tplane >= (float)(INT_MAX - 10)) { tplane = INFINITY; plane_id = -1; }
        trav = (a.root_ref != REF_NONE) && !(shadow && plane_id >= 0 && tplane < limit);
      }
      {
        // traverse_lbvh, bvh_traversal.cu:92-183
        const f3 inv = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
        const float tmin = 0.0001f;
        uint32_t cur = a.root_ref;
        int sp = 0;
        while (__ballot(trav)) {
          if (trav) {
            if (cur & REF_LEAF) {
              // intersect_leaf_primitives, bvh_traversal.cu:47-89
              const uint32_t id = cur & REF_IDMASK;
              float t = 0.0f;
              bool hit = false;
              if (cur & REF_TRI) {
                // checkTriangleIntersectionSoA, struct.cu:111-163
                const float4 q0 = a.tris[3 * (size_t)id + 0], q1 = a.tris[3 * (size_t)id + 1], q2 = a.tris[3 * (size_t)id + 2];
                if (COUNT) cn.tri_tests++;
                const f3 p0 = mk3(q0.x, q0.y, q0.z), nor = mk3(q0.w, q1.x, q1.y);
                const float denom = dot(ray.d, nor);
                if (!(fabsf(denom) < 1e-9f)) {
                  t = dot(p0 - ray.o, nor) / denom;
                  if (!(t <= EPSILON)) {
                    const f3 ip = t * ray.d + ray.o;
                    const f3 e1 = mk3(q1.z, q1.w, q2.x), e2 = mk3(q2.y, q2.z, q2.w);
                    const float b1 = dot(e1, ip - p0);
                    const float b2 = dot(e2, ip - p0);
                    const float b0 = 1.0f - b1 - b2;
                    hit = (b0 >= -EPSILON) && (b1 >= -EPSILON) && (b2 >= -EPSILON);
                  }
                }
              } else {
                // checkSphereIntersectionSoA, struct.cu:64-109
                const float4 s = a.spheres[id];
                if (COUNT) cn.sphere_tests++;
                const f3 c = mk3(s.x, s.y, s.z);
                const float r = s.w;
                const f3 cr0 = c - ray.o;
                const bool inside = (dot(cr0, cr0) < r * r);
                const float tc = dot(cr0, ray.d);
                if (!(!inside && tc < 0.0f)) {
                  const f3 dv = ray.o + (tc * ray.d) - c;
                  const float d2 = dot(dv, dv);
                  if (!(!inside && (r * r) < d2)) {
                    const float toff = sqrtf((r * r) - d2);
                    t = inside ? (tc + toff) : (tc - toff);
                    hit = true;
                  }
                }
              }
              if (hit && t > 1e-6f && t < tbest) {
                tbest = t; refbest = cur;
                if (shadow && tbest < limit) trav = false;      // any-hit exit: the caller only asks "closer than limit?"
              }
              if (trav) {
                if (sp == 0) trav = false;
                else {
                  --sp;
                  cur = (sp < STACK_LDS) ? lds_stack[sp * RBLOCK + tid] : a.stack_spill[(size_t)(sp - STACK_LDS) * gthreads + gid];
                }
              }
            } else {
              const float4 n0 = a.nodes[4 * (size_t)cur + 0], n1 = a.nodes[4 * (size_t)cur + 1];
              const float4 n2 = a.nodes[4 * (size_t)cur + 2], n3 = a.nodes[4 * (size_t)cur + 3];
              if (COUNT) cn.internal_visits++;
              // hit_aabb_adapted, bvh_traversal.cu:11-44, on both children
              float tx1 = (n0.x - ray.o.x) * inv.x, tx2 = (n0.w - ray.o.x) * inv.x;
              float ty1 = (n0.y - ray.o.y) * inv.y, ty2 = (n1.x - ray.o.y) * inv.y;
              float tz1 = (n0.z - ray.o.z) * inv.z, tz2 = (n1.y - ray.o.z) * inv.z;
              float te = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
              float tx = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
              const bool hl = te < tx && te < tbest && tx > tmin;
              tx1 = (n1.z - ray.o.x) * inv.x; tx2 = (n2.y - ray.o.x) * inv.x;
              ty1 = (n1.w - ray.o.y) * inv.y; ty2 = (n2.z - ray.o.y) * inv.y;
              tz1 = (n2.x - ray.o.z) * inv.z; tz2 = (n2.w - ray.o.z) * inv.z;
              te = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
              tx = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
              const bool hr = te < tx && te < tbest && tx > tmin;
              const uint32_t lref = __float_as_uint(n3.x), rref = __float_as_uint(n3.y);
              if (hl && hr) {
                cur = lref;
                if (sp < STACK_TOTAL) {
                  if (sp < STACK_LDS) lds_stack[sp * RBLOCK + tid] = rref;
                  else a.stack_spill[(size_t)(sp - STACK_LDS) * gthreads + gid] = rref;
                  ++sp;
                  if (COUNT) cn.max_stack = max(cn.max_stack, (uint32_t)sp);
                }
              } else if (hl) cur = lref;
              else if (hr) cur = rref;
              else {
                if (sp == 0) trav = false;
                else {
                  --sp;
                  cur = (sp < STACK_LDS) ? lds_stack[sp * RBLOCK + tid] : a.stack_spill[(size_t)(sp - STACK_LDS) * gthreads + gid];
                }
              }
            }
          }
        }
      }

      // =========================== shade phase (per lane) ===========================
      if (alive) {
        int micro = M_TRACE;
        // ---- consume the trace result -------------------------------------------------------------
        // hitNearest, draw.cu:292-318: the nearer of BVH hit and plane hit (the plane wins an exact tie)
        const bool bvh_hit = refbest != REF_NONE;
        const bool pl_hit = plane_id >= 0;
        if (state == ST_SHADOW) {
          const bool occluded = (pl_hit && tplane < limit) || (bvh_hit && tbest < limit);
          if (!occluded) {
            // draw.cu:354-355 / 372-373
            if (li < a.num_suns) {
              const LightDev& lt = a.suns[li];
              const float lambert = fmaxf(dot(pn, ray.d), 0.0f);   // ray.d == normalize(light.dir)
              const float r = Hcolor.x * (lt.r * lambert), gg = Hcolor.y * (lt.g * lambert), b = Hcolor.z * (lt.b * lambert);
              Dacc = Dacc + mk3(set_expose(r, a.expose), set_expose(gg, a.expose), set_expose(b, a.expose));
            } else {
              const LightDev& lt = a.bulbs[li - a.num_suns];
              const float lambert = fmaxf(dot(pn, ray.d), 0.0f);   // ray.d == normalize(bulbDir)
              const float tl = limit;                               // == bulbDir.length()
              const float inv2 = 1.0f / (tl * tl);
              const float r = Hcolor.x * (lt.r * lambert), gg = Hcolor.y * (lt.g * lambert), b = Hcolor.z * (lt.b * lambert);
              Dacc = Dacc + mk3(set_expose(r, a.expose) * inv2, set_expose(gg, a.expose) * inv2, set_expose(b, a.expose) * inv2);
            }
          }
          ++li;
          micro = M_LIGHT;
        } else {
          const bool use_bvh = bvh_hit && (!pl_hit || tbest < tplane);
          const bool hit = bvh_hit || pl_hit;
          // resolve the hit: point, normal, material (same expressions as the primitive tests)
          f3 Np = mk3(0, 0, 0), Nn = mk3(0, 0, 0);
          Mat nm;
          nm.color = mk3(0, 0, 0); nm.shininess = mk3(0, 0, 0); nm.trans = mk3(0, 0, 0); nm.ior = 1.458f; nm.roughness = 0.0f;
          if (use_bvh) {
            const uint32_t id = refbest & REF_IDMASK;
            Np = tbest * ray.d + ray.o;
            if (refbest & REF_TRI) {
              const float4 q0 = a.tris[3 * (size_t)id + 0], q1 = a.tris[3 * (size_t)id + 1];
              const f3 nor = mk3(q0.w, q1.x, q1.y);
              const float denom = dot(ray.d, nor);
              Nn = (denom < 0.0f) ? nor : -nor;
              nm = load_mat(a.mats, (uint32_t)a.num_spheres + id);
            } else {
              const float4 s = a.spheres[id];
              const f3 c = mk3(s.x, s.y, s.z);
              const f3 cr0 = c - ray.o;
              const bool inside = (dot(cr0, cr0) < s.w * s.w);
              Nn = normalize(inside ? (c - Np) : (Np - c));
              nm = load_mat(a.mats, id);
            }
            if (COUNT) cn.mat_fetches++;
          } else if (pl_hit) {
            const PlaneDev& pl = a.planes[plane_id];
            const f3 pnor = mk3(pl.nx, pl.ny, pl.nz);
            Np = tplane * ray.d + ray.o;
            Nn = (dot(pnor, ray.d) < 0.0f) ? pnor : -pnor;
            nm = plane_mat(pl);
          }

          if (state == ST_REFR_INSIDE) {
            // second half of refractionLight, draw.cu:484-493.  No miss check: a miss yields the default ObjectInfo
            // (normal 0, ior 1.458, point 0), which the code above has already produced in Np/Nn/nm.
            const f3 normal = normalize(Nn);
            const float ior = nm.ior;
            const f3 dir = ray.d;
            const float dn = dot(normal, dir);
            const float k = 1.0f - ior * ior * (1.0f - (dn * dn));
            const f3 rd = ior * dir - (ior * (dot(normal, dir)) + sqrtf(k)) * normal;
            ray = mkray(Np - normal * 0.0001f, rd, refr_bounce - 1);
            state = ST_REFR_FINAL;
            micro = (ray.bounce == 0) ? M_POP : M_TRACE;
          } else if (!hit) {
            // primary miss: RGBA(0,0,0,0) (draw.cu:267,284).  Secondary miss contributes nothing to rgb.
            micro = (state == ST_PRIMARY) ? M_DONE : M_POP;
          } else {
            if (state == ST_PRIMARY) { alpha = 1.0f; wt = mk3(1.0f, 1.0f, 1.0f); x_parent = false; has_gi = true; gi_n = a.gi; }
            else if (state == ST_REFLECT) {
              // the new node's refraction term uses the ORIGINAL ray and object (draw.cu:424)
              Xdir = Hdir; Xbounce = Hbounce; Xp = Hp; Xn = Hn; Xior = Hior; XtransNZ = HtransNZ; x_parent = true; has_gi = false;
            } else if (state == ST_REFR_FINAL) { x_parent = false; has_gi = false; }
            else { x_parent = false; has_gi = true; }   // ST_GI: gi_n was set when the ray was made
            Hdir = ray.d; Hbounce = ray.bounce; Hp = Np; Hn = Nn;
            Hcolor = nm.color; Hior = nm.ior; Hrough = nm.roughness; HtransNZ = !is_black(nm.trans);
            // ---- M_ENTER: weights of this node's four terms (draw.cu:277-281, 426-428, 517-519, 561-563) ----
            const f3 one = mk3(1.0f, 1.0f, 1.0f);
            const f3 S = nm.shininess, T = nm.trans;
            const f3 K = (one - S) * (one - T);
            if (has_gi && a.gi != 0 && gi_n != 0 && pc < a.pending_slots) {
              const f3 w = (wt * K) * nm.color;
              float* e = a.pending + ((size_t)pc * PENDING_WORDS) * gthreads + gid;
              e[0 * gthreads] = __uint_as_float(PEND_G);
              e[1 * gthreads] = Hp.x; e[2 * gthreads] = Hp.y; e[3 * gthreads] = Hp.z;
              e[4 * gthreads] = Hn.x; e[5 * gthreads] = Hn.y; e[6 * gthreads] = Hn.z;
              e[7 * gthreads] = __int_as_float(gi_n);
              e[8 * gthreads] = w.x; e[9 * gthreads] = w.y; e[10 * gthreads] = w.z;
              ++pc;
            }
            if (!x_parent) { Xdir = Hdir; Xbounce = Hbounce; Xp = Hp; Xn = Hn; Xior = Hior; XtransNZ = HtransNZ; }
            if (XtransNZ && Xbounce > 0 && pc < a.pending_slots) {
              const f3 w = wt * ((one - S) * T);
              float* e = a.pending + ((size_t)pc * PENDING_WORDS) * gthreads + gid;
              e[0 * gthreads] = __uint_as_float(PEND_F);
              e[1 * gthreads] = Xp.x; e[2 * gthreads] = Xp.y; e[3 * gthreads] = Xp.z;
              e[4 * gthreads] = Xn.x; e[5 * gthreads] = Xn.y; e[6 * gthreads] = Xn.z;
              e[7 * gthreads] = __int_as_float(Xbounce);
              e[8 * gthreads] = w.x; e[9 * gthreads] = w.y; e[10 * gthreads] = w.z;
              e[11 * gthreads] = Xdir.x; e[12 * gthreads] = Xdir.y; e[13 * gthreads] = Xdir.z;
              e[14 * gthreads] = Xior;
              ++pc;
            }
            wD = wt * K;
            wt = wt * S;
            reflect_ok = !is_black(S) && Hbounce > 0;
            // diffuseLight prologue, draw.cu:331-340
            pn = Hn;
            if (Hrough > 0.0f) pn = rough_normal(Hn, Hrough, rng);
            pn = normalize(pn);
            Dacc = mk3(0.0f, 0.0f, 0.0f);
            li = 0;
            micro = M_LIGHT;
          }
        }

        // ---- run the micro-states until this lane has a ray or is finished ---------------------------
        while (micro != M_TRACE && micro != M_DONE) {
          if (micro == M_LIGHT) {
            if (li < nlights) {
              // shadow ray, draw.cu:346 / 362-363
              if (li < a.num_suns) {
                const LightDev& lt = a.suns[li];
                ray = mkray(Hp + Hn * EPSILON, mk3(lt.x, lt.y, lt.z), 1);
                limit = INFINITY;
              } else {
                const LightDev& lt = a.bulbs[li - a.num_suns];
                const f3 bd = mk3(lt.x, lt.y, lt.z) - Hp;
                ray = mkray(Hp + Hn * EPSILON, bd, 1);
                limit = length(bd);
              }
              state = ST_SHADOW;
              micro = M_TRACE;
            } else {
              L = L + wD * Dacc;
              micro = M_REFLECT;
            }
          } else if (micro == M_REFLECT) {
            // reflectionLight, draw.cu:389-404
            if (!reflect_ok) micro = M_POP;
            else {
              f3 normal = Hn;
              if (Hrough > 0.0f) normal = rough_normal(Hn, Hrough, rng);
              normal = normalize(normal);
              const f3 rd = Hdir - 2.0f * (dot(normal, Hdir)) * normal;
              ray = mkray(Hp + Hn * EPSILON, rd, Hbounce - 1);
              state = ST_REFLECT;
              micro = (ray.bounce == 0) ? M_POP : M_TRACE;
            }
          } else {   // M_POP
            if (pc == 0) micro = M_DONE;
            else {
              --pc;
              const float* e = a.pending + ((size_t)pc * PENDING_WORDS) * gthreads + gid;
              const uint32_t tag = __float_as_uint(e[0 * gthreads]);
              const f3 p = mk3(e[1 * gthreads], e[2 * gthreads], e[3 * gthreads]);
              const f3 n = mk3(e[4 * gthreads], e[5 * gthreads], e[6 * gthreads]);
              const int ib = __float_as_int(e[7 * gthreads]);
              wt = mk3(e[8 * gthreads], e[9 * gthreads], e[10 * gthreads]);
              if (tag == PEND_G) {
                // globalIllumination, draw.cu:540-549
                const f3 gi_dir = normalize(n + sphere_point(rng));
                ray = mkray(p + n * EPSILON, gi_dir, ib - 1);
                gi_n = ib - 1;
                state = ST_GI;
                micro = (ray.bounce == 0) ? M_POP : M_TRACE;
              } else {
                // refractionLight, draw.cu:456-480
                const f3 dir = mk3(e[11 * gthreads], e[12 * gthreads], e[13 * gthreads]);
                const float ior = 1.0f / e[14 * gthreads];
                const f3 normal = normalize(n);
                const float dn = dot(normal, dir);
                const float k = 1.0f - (ior * ior) * (1.0f - (dn * dn));
                if (k < 0) {
                  const f3 rd = dir - 2.0f * (dot(normal, dir)) * normal;
                  ray = mkray(p + normal * EPSILON, rd, ib - 1);
                  state = ST_REFR_FINAL;
                  micro = (ray.bounce == 0) ? M_POP : M_TRACE;
                } else {
                  const f3 rd = ior * dir - (ior * (dot(normal, dir)) + sqrtf(k)) * normal;
                  ray = mkray(p - normal * 0.0001f, rd, ib);
                  refr_bounce = ib;
                  state = ST_REFR_INSIDE;
                  micro = M_TRACE;   // ib > 0 is guaranteed by the push condition
                }
              }
            }
          }
        }
        if (micro == M_DONE) alive = false;
      }
      want = alive;
    }
    if (g < a.num_samples) a.samples[g] = make_float4(L.x, L.y, L.z, alpha);
  }

  if (COUNT && a.counters) {
    uint32_t v[8] = {cn.samples, cn.rays, cn.shadow_rays, cn.internal_visits, cn.sphere_tests, cn.tri_tests, cn.mat_fetches, cn.max_stack};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      unsigned long long x = v[k];
      if (k == 7) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { unsigned long long y = __shfl_xor(x, off); x = x > y ? x : y; }
        if ((tid & 63) == 0) atomicMax(&a.counters[k], x);
      } else {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if ((tid & 63) == 0) atomicAdd(&a.counters[k], x);
      }
    }
  }
}

// draw.cu:9-11
MIRT_DEV unsigned char to_uchar_round(float f) { return (unsigned char)(fminf(fmaxf(f, 0.0f), 1.0f) * 255.0f + 0.5f); }
// draw.cu:129-132: plain float -> unsigned char conversion
MIRT_DEV unsigned char to_uchar_trunc(float f)
{
  if (!(f > 0.0f)) return 0;
  if (f >= 255.0f) return 255;
  return (unsigned char)f;
}

__global__ void __launch_bounds__(RBLOCK) resolve_kernel(const ResolveArgs a)
{
  const long long lp = (long long)blockIdx.x * RBLOCK + threadIdx.x;
  if (lp >= a.num_local_pixels) return;
  float4 m;
  if (a.spp <= 1) {
    m = a.samples[lp];
  } else {
    // Sum in the order of `for (mask = P/2; mask > 0; mask /= 2) v += shfl_xor(v, mask)` as lane 0 sees it
    // (draw.cu:181-189), P = next power of two >= spp, absent samples = 0: a pairwise tree over the samples in
    // bit-reversed order.
    int P = 1, lg = 0;
    while (P < a.spp) { P <<= 1; ++lg; }
    const float4* s = a.samples + lp * a.spp;
    float4 stk[12];
    int top = 0;
    for (int i = 0; i < P; ++i) {
      const int idx = (int)(__brev((unsigned)i) >> (32 - lg));
      float4 x = (idx < a.spp) ? s[idx] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      int j = i;
      while (j & 1) {
        --top;
        const float4 l = stk[top];
        x = make_float4(l.x + x.x, l.y + x.y, l.z + x.z, l.w + x.w);
        j >>= 1;
      }
      stk[top++] = x;
    }
    const float4 sum = stk[0];
    const float inv = 1.0f / (float)a.spp;
    m = make_float4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv);
  }
  if (a.rgba_f32) a.rgba_f32[lp] = m;
  uchar4 o;
  if (a.spp <= 1) {
    o.x = to_uchar_trunc(rgb_to_srgb(m.x) * 255);
    o.y = to_uchar_trunc(rgb_to_srgb(m.y) * 255);
    o.z = to_uchar_trunc(rgb_to_srgb(m.z) * 255);
    o.w = to_uchar_trunc(m.w * 255);
  } else {
    o.x = to_uchar_round(rgb_to_srgb(m.x));
    o.y = to_uchar_round(rgb_to_srgb(m.y));
    o.z = to_uchar_round(rgb_to_srgb(m.z));
    o.w = to_uchar_round(m.w);
  }
  reinterpret_cast<uchar4*>(a.rgba8)[lp] = o;
}

__global__ void __launch_bounds__(RBLOCK) scatter_kernel(const uchar4* __restrict__ part, uchar4* __restrict__ frame, long long n,
                                                         int width, int height, int stripe_rows, int num_parts, int part_id)
{
  const long long lp = (long long)blockIdx.x * RBLOCK + threadIdx.x;
  if (lp >= n) return;
  const long long stripe_pixels = (long long)stripe_rows * width;
  const long long ls = lp / stripe_pixels, within = lp - ls * stripe_pixels;
  const long long gs = ls * num_parts + part_id;
  const long long y = gs * stripe_rows + within / width, x = within % width;
  frame[y * width + x] = part[lp];
}

__global__ void probe_math_kernel(int which, int n, const float* __restrict__ in, float* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = in[i];
  float y;
  switch (which) {
    case 0: y = dm_logf(x); break;
    case 1: y = dm_expf(x); break;
    case 2: y = dm_sinf(x); break;
    case 3: y = dm_cosf(x); break;
    case 4: y = dm_powf(x, 1 / 2.4f); break;
    case 5: y = rgb_to_srgb(x); break;
    case 6: y = sqrtf(x); break;
    case 7: y = 1.0f / x; break;
    default: y = x;
  }
  out[i] = y;
}

// out[i*draws + k]: k-th raw 32-bit draw of stream i.  mode 0: curand_init(1234 + i/spp.., ...) is exercised through
// xw_init exactly as the trace kernel does: pixel = i / spp, sample = i % spp (spp > 1) or pixel = i (spp <= 1).
__global__ void probe_xorwow_kernel(RngTablesDev t, int spp, int nstreams, int draws, uint32_t* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nstreams) return;
  Xorwow s;
  if (spp > 1) xw_init(s, t, (uint32_t)(i / spp), (uint32_t)(i % spp));
  else xw_init(s, t, (uint32_t)i, 0u);
  for (int k = 0; k < draws; ++k) out[(size_t)i * draws + k] = xw_next(s);
}

int64_t local_pixels(const MirtRenderParams* p)
{
  if (p->width <= 0 || p->height <= 0 || p->stripe_rows <= 0 || p->num_parts <= 0 || p->part < 0 || p->part >= p->num_parts) return -1;
  const int64_t nstripes = ((int64_t)p->height + p->stripe_rows - 1) / p->stripe_rows;
  int64_t rows = 0;
  for (int64_t s = p->part; s < nstripes; s += p->num_parts) {
    const int64_t r0 = s * p->stripe_rows;
    const int64_t r1 = r0 + p->stripe_rows < p->height ? r0 + p->stripe_rows : p->height;
    rows += r1 - r0;
  }
  return rows * p->width;
}

} // namespace

int ensure_rng_tables(RngCache* rc, int spp, long long frame_pixels, hipStream_t stream, RngTablesDev* out)
{
  const long long key = spp > 1 ? (long long)spp : -frame_pixels;
  if (rc->key != key) {
    if (spp > 1) build_sample_tables(spp, rc->host);
    else build_pixel_tables(frame_pixels, 1234, rc->host);
    MIRT_HIP(hipStreamSynchronize(stream));
    rng_cache_free(rc);
    const RngTables& t = rc->host;
    MIRT_HIP(hipMalloc(&rc->A, t.A.size() * 4)); MIRT_HIP(hipMemcpy(rc->A, t.A.data(), t.A.size() * 4, hipMemcpyHostToDevice));
    MIRT_HIP(hipMalloc(&rc->B, t.B.size() * 4)); MIRT_HIP(hipMemcpy(rc->B, t.B.data(), t.B.size() * 4, hipMemcpyHostToDevice));
    MIRT_HIP(hipMalloc(&rc->K, t.K.size() * 4)); MIRT_HIP(hipMemcpy(rc->K, t.K.data(), t.K.size() * 4, hipMemcpyHostToDevice));
    if (!t.R2.empty()) { MIRT_HIP(hipMalloc(&rc->R2, t.R2.size() * 4)); MIRT_HIP(hipMemcpy(rc->R2, t.R2.data(), t.R2.size() * 4, hipMemcpyHostToDevice)); }
    rc->key = key;
  }
  const RngTables& t = rc->host;
  out->A = rc->A; out->B = rc->B; out->K = rc->K; out->R2 = rc->R2;
  out->mode = t.mode; out->chunk_bits = t.chunk_bits; out->nin_words = t.nin_words; out->nchunks = t.nchunks; out->d0 = t.d0;
  return MIRT_OK;
}

void rng_cache_free(RngCache* rc)
{
  hipFree(rc->A); hipFree(rc->B); hipFree(rc->K); hipFree(rc->R2);
  rc->A = nullptr; rc->B = nullptr; rc->K = nullptr; rc->R2 = nullptr; rc->key = -1;
}

static int grid_blocks(int device)
{
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 1024;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trace_kernel<false>, RBLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 4;
  return prop.multiProcessorCount * per_cu;
}

int render(MirtScene* sc, const MirtRenderParams* p, void* d_rgba8, void* d_rgba_f32, hipStream_t stream)
{
  if (!sc->built) { set_error("mirt_render: call mirt_build_lbvh first"); return MIRT_ERR_STATE; }
  const int64_t npix = local_pixels(p);
  if (npix < 0 || p->spp < 0 || !d_rgba8) { set_error("mirt_render: bad parameters"); return MIRT_ERR_ARG; }
  if ((int64_t)p->width * p->height > 0x7fffffffll - 1234) { set_error("mirt_render: frame too large for the 32-bit pixel seed"); return MIRT_ERR_ARG; }
  if (npix == 0) return MIRT_OK;
  const int sppe = p->spp > 1 ? p->spp : 1;
  const long long nsamples = (long long)npix * sppe;
  const bool count = (p->flags & MIRT_RENDER_COUNTERS) != 0;

  static int blocks_cached = 0;
  if (!blocks_cached) blocks_cached = grid_blocks(sc->device);
  long long want_blocks = (nsamples + RBLOCK - 1) / RBLOCK;
  const int blocks = (int)(want_blocks < blocks_cached ? want_blocks : blocks_cached);
  const size_t gthreads = (size_t)blocks * RBLOCK;

  // workspace
  if (sc->samples_cap < (size_t)nsamples) {
    MIRT_HIP(hipStreamSynchronize(stream));
    hipFree(sc->samples); sc->samples = nullptr; sc->samples_cap = 0;
    MIRT_HIP(hipMalloc(&sc->samples, sizeof(float4) * (size_t)nsamples));
    sc->samples_cap = (size_t)nsamples;
  }
  const size_t spill_need = (size_t)(STACK_TOTAL - STACK_LDS) * gthreads;
  if (sc->spill_cap < spill_need) {
    MIRT_HIP(hipStreamSynchronize(stream));
    hipFree(sc->stack_spill); sc->stack_spill = nullptr; sc->spill_cap = 0;
    MIRT_HIP(hipMalloc(&sc->stack_spill, sizeof(uint32_t) * spill_need));
    sc->spill_cap = spill_need;
  }
  const bool need_pending = sc->any_trans || sc->d.gi != 0;
  const int pending_slots = need_pending ? 2 * (sc->d.bounces + (sc->d.gi > 0 ? sc->d.gi : 0) + 2) : 0;
  const size_t pending_need = (size_t)pending_slots * PENDING_WORDS * gthreads;
  if (sc->pending_cap < pending_need) {
    MIRT_HIP(hipStreamSynchronize(stream));
    hipFree(sc->pending); sc->pending = nullptr; sc->pending_cap = 0;
    MIRT_HIP(hipMalloc(&sc->pending, sizeof(float) * pending_need));
    sc->pending_cap = pending_need;
  }

  RenderArgs a;
  memset(&a, 0, sizeof(a));
  a.width = p->width; a.height = p->height; a.bounces = sc->d.bounces; a.spp = p->spp; a.gi = sc->d.gi;
  a.fisheye = sc->d.fisheye; a.panorama = sc->d.panorama;
  a.dof_focus = sc->d.dof_focus; a.dof_lens = sc->d.dof_lens; a.expose = sc->d.expose;
  a.forward.x = sc->d.forward.x; a.forward.y = sc->d.forward.y; a.forward.z = sc->d.forward.z;
  a.right.x = sc->d.right.x; a.right.y = sc->d.right.y; a.right.z = sc->d.right.z;
  a.up.x = sc->d.up.x; a.up.y = sc->d.up.y; a.up.z = sc->d.up.z;
  a.eye.x = sc->d.eye.x; a.eye.y = sc->d.eye.y; a.eye.z = sc->d.eye.z;
  a.stripe_rows = p->stripe_rows; a.num_parts = p->num_parts; a.part = p->part;
  a.num_local_pixels = npix; a.num_samples = nsamples;
  a.nodes = sc->nodes; a.spheres = sc->spheres; a.tris = sc->tris; a.mats = sc->mats;
  a.root_ref = sc->root_ref; a.num_spheres = sc->Ns; a.num_prims = sc->N;
  a.planes = sc->planes; a.num_planes = sc->d.num_planes;
  a.suns = sc->suns; a.num_suns = sc->d.num_suns;
  a.bulbs = sc->bulbs; a.num_bulbs = sc->d.num_bulbs;
  // random numbers are consumed only by jitter (spp >= 1), depth of field, rough normals and GI
  a.needs_rng = (p->spp >= 1) || (sc->d.dof_focus != 0.0f && !sc->d.fisheye && !sc->d.panorama) || sc->any_rough || sc->d.gi != 0;
  if (a.needs_rng) {
    int rc = ensure_rng_tables(&sc->rng, p->spp, (long long)p->width * p->height, stream, &a.rng);
    if (rc != MIRT_OK) return rc;
  }
  a.samples = sc->samples;
  a.stack_spill = sc->stack_spill;
  a.pending = sc->pending; a.pending_slots = pending_slots;
  a.counters = count ? sc->counters : nullptr;

  MIRT_HIP(hipEventRecord(sc->ev0, stream));
  if (count) MIRT_HIP(hipMemsetAsync(sc->counters, 0, 8 * sizeof(unsigned long long), stream));
  MIRT_HIP(hipEventRecord(sc->ev1, stream));
  if (count) hipLaunchKernelGGL(trace_kernel<true>, dim3(blocks), dim3(RBLOCK), 0, stream, a);
  else hipLaunchKernelGGL(trace_kernel<false>, dim3(blocks), dim3(RBLOCK), 0, stream, a);
  MIRT_HIP(hipGetLastError());
  MIRT_HIP(hipEventRecord(sc->ev2, stream));

  ResolveArgs ra;
  ra.samples = sc->samples; ra.rgba8 = (unsigned char*)d_rgba8; ra.rgba_f32 = (float4*)d_rgba_f32;
  ra.num_local_pixels = npix; ra.spp = p->spp;
  hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((npix + RBLOCK - 1) / RBLOCK)), dim3(RBLOCK), 0, stream, ra);
  MIRT_HIP(hipGetLastError());
  MIRT_HIP(hipEventRecord(sc->ev3, stream));
  sc->last_stream = stream; sc->have_render = true; sc->last_counted = count;
  return MIRT_OK;
}

int scatter_part(const MirtRenderParams* p, const void* d_part, void* d_frame, hipStream_t stream)
{
  const int64_t n = local_pixels(p);
  if (n < 0 || !d_part || !d_frame) { set_error("mirt_scatter_part: bad parameters"); return MIRT_ERR_ARG; }
  if (n == 0) return MIRT_OK;
  hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((n + RBLOCK - 1) / RBLOCK)), dim3(RBLOCK), 0, stream, (const uchar4*)d_part,
                     (uchar4*)d_frame, (long long)n, p->width, p->height, p->stripe_rows, p->num_parts, p->part);
  MIRT_HIP(hipGetLastError());
  return MIRT_OK;
}

int64_t render_num_pixels(const MirtRenderParams* p) { return local_pixels(p); }

int probe_math(int device, int which, int n, const float* in, float* out)
{
  MIRT_HIP(hipSetDevice(device));
  float *di = nullptr, *dout = nullptr;
  MIRT_HIP(hipMalloc(&di, 4 * (size_t)n)); MIRT_HIP(hipMalloc(&dout, 4 * (size_t)n));
  MIRT_HIP(hipMemcpy(di, in, 4 * (size_t)n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, which, n, di, dout);
  MIRT_HIP(hipGetLastError());
  MIRT_HIP(hipMemcpy(out, dout, 4 * (size_t)n, hipMemcpyDeviceToHost));
  hipFree(di); hipFree(dout);
  return MIRT_OK;
}

int probe_xorwow(int device, int spp, int nstreams, int draws, uint32_t* out)
{
  MIRT_HIP(hipSetDevice(device));
  RngCache cache;
  RngTablesDev t;
  int rc = ensure_rng_tables(&cache, spp, nstreams, nullptr, &t);
  if (rc != MIRT_OK) return rc;
  uint32_t* d = nullptr;
  MIRT_HIP(hipMalloc(&d, 4 * (size_t)nstreams * draws));
  hipLaunchKernelGGL(probe_xorwow_kernel, dim3((nstreams + 255) / 256), dim3(256), 0, 0, t, spp, nstreams, draws, d);
  MIRT_HIP(hipGetLastError());
  MIRT_HIP(hipMemcpy(out, d, 4 * (size_t)nstreams * draws, hipMemcpyDeviceToHost));
  hipFree(d);
  rng_cache_free(&cache);
  return MIRT_OK;
}

} // namespace mirt
