// Render on the device: replaces render() and the render kernels of draw.cu:94-239 together with the device code
// they call (shootPrimaryRay/hitNearest/diffuseLight/reflectionLight/refractionLight/globalIllumination/checkPlane,
// draw.cu:260-659; traverse_lbvh, bvh_traversal.cu:11-183; Ray::Ray and the primitive tests, struct.cu:16-163).
//
// trace_kernel    one lane = one sample.  A persistent grid walks the samples; every lane runs a small state machine
//                 (the reference's mutually recursive shading functions turned into an explicit ray-tree walk) around ONE
//                 shared BVH traversal loop, so that primary, shadow, reflection, refraction and GI rays of different
//                 lanes are traversed together.  Traversal stack: 32 entries per lane in LDS ([entry][lane], conflict
//                 free), spilling to global memory above that.  Nodes are 64-byte two-child records.  A workgroup
//                 is one wave; the loop header (who traverses, who waits, is the wave draining, are enough primitive
//                 tests pending) runs once per four traversal steps; only what the loop touches is passed by value.
// resolve_tree_kernel / resolve_kernel
//                 per pixel: sum the samples in the reference's xor-butterfly order (draw.cu:181-189), mean, sRGB,
//                 quantise (draw.cu:129-132 for spp <= 1, draw.cu:9-11,202-205 otherwise); one lane per sample when
//                 the butterfly fits a wave, one thread per pixel otherwise.
// order_kernel    sorts the frame's sample chunks by their measured cost: later frames hand them out longest first.
//
// The ray tree is evaluated top-down (every ray carries the product of the mixing weights above it) instead of the
// reference's bottom-up recursion; geometry and random-number consumption are identical, colours agree to rounding.
#include "scene_dev.h"
#include "host_scene.h"
#include "shade_common.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mirt {
namespace {

constexpr int RBLOCK = 256;
constexpr int MAX_SPP = 4096;         // resolve_kernel's partial-sum stack holds log2(4096) + 1 entries
constexpr int MAX_SLAB_ARGS = 4;      // device copies of RenderArgs a context cycles through (one per slab in flight on its stream)
// The trace kernel's waves never talk to each other, so a workgroup is one wave: a finished wave frees its slot (and its
// 10 KB of LDS) at once instead of waiting for the slowest of four, which is what lets the next frame's waves move in
// while this frame drains.
#ifndef MIRT_TRACE_BLOCK
#define MIRT_TRACE_BLOCK 64
#endif
constexpr int TRACE_BLOCK = MIRT_TRACE_BLOCK;
constexpr int MAX_CHUNK_SHIFT = 8, MIN_CHUNK_SHIFT = 6;   // a wave takes 64..256 consecutive samples from the frame per atomic
#ifndef MIRT_WAVES_PER_SIMD
#define MIRT_WAVES_PER_SIMD 4   // 128 VGPRs: measured best (2: 86 ms, 3: 76 ms, 4: 68 ms, 5: 79 ms on tenthousand 1080p16)
#endif
#ifndef MIRT_STACK_LDS
#define MIRT_STACK_LDS 24
#endif
constexpr int STACK_LDS = MIRT_STACK_LDS;
// The quantised walk found a triangle hit it is about to accept (scene_dev.h): does the reference's walk reach this leaf?  Yes,
// provably, if the triangle's exact leaf box passes the order-independent clauses of hit_aabb_adapted (bvh_traversal.cu:11-44)
// and the ray enters it before the hit: every ancestor's exact box contains the leaf box and the slab arithmetic is monotone in
// the box planes, so t_enter(ancestor) <= t_enter(leaf) < t <= the best distance at the time the ancestor was tested, and each of
// those tests passed.  A hit outside its box (the triangle test accepts up to 0.001 outside the triangle, struct.cu:155) does
// not qualify: the ray is then walked again over the exact records.  Rare: the scene-sized arguments come through `ap`.
MIRT_DEV bool triangle_leaf_reached(const RenderArgs* __restrict__ ap, uint32_t off16, const f3& o, const f3& d, float t, float tmin, f3& inv)
{
  const RenderArgs* aq = ap;
  asm volatile("" : "+s"(aq));
  const uint32_t id = aq->unit_prim[off16 - aq->prim_base16] & 0x7fffffffu;
  const float4 b0 = aq->tri_boxes[2 * (size_t)id], b1 = aq->tri_boxes[2 * (size_t)id + 1];
  inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const float tx1 = (b0.x - o.x) * inv.x, tx2 = (b0.y - o.x) * inv.x;
  const float ty1 = (b0.z - o.y) * inv.y, ty2 = (b0.w - o.y) * inv.y;
  const float tz1 = (b1.x - o.z) * inv.z, tz2 = (b1.y - o.z) * inv.z;
  const float te = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
  const float tx = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
  return te < tx && tx > tmin && te < t;
}

// QN: the walk starts on the 32-byte quantised node records (scene_dev.h); SPECX: SPEC_NOTRI / SPEC_NOBULB / SPEC_NOPEND of
// shade_common.h.  With QN and triangles a lane can also be on the 64-byte exact records (S.qsx == 0: its ray is being walked
// again, triangle_leaf_reached); then S.inv holds 1 / d.
// With QN on a scene that has triangles the quantised records are the wide ones (WIDE, scene_dev.h): a step tests the boxes of
// the node's four grandchildren and descends two levels.
template <bool COUNT, int TABLES, bool QN, int SPECX = 0>
__global__ void __launch_bounds__(TRACE_BLOCK, MIRT_WAVES_PER_SIMD) trace_kernel(const RenderArgs* __restrict__ ap, const HotArgs h)
{
  constexpr int SPEC = SPECX;
  constexpr bool NOTRI = (SPEC & SPEC_NOTRI) != 0;
  constexpr bool WIDE = QN && !NOTRI;
  // a finished shadow ray towards a point light that hit something, in a walk that is not the reference's own: the hit is vetted
  // by the shade phase before batch_next reads "occluded" off it (hit_needs_literal_walk); no such lanes in a kernel without bulbs
#define MIRT_HELD (QN && !(SPEC & SPEC_NOBULB) && h.reach_check && S.batch_pending && !S.trav && S.li >= h.num_suns && S.li < h.num_suns + h.num_bulbs && \
                   S.refbest != REF_NONE)
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[STACK_LDS * TRACE_BLOCK * 4];     // traversal stacks: STACK_LDS x TRACE_BLOCK words
  // The random-number state (8 words per lane) is only touched in the shade phase: it lives here during traversal so
  // that it does not occupy registers across the hot loop (the kernel runs at the 128-VGPR edge of 4 waves per SIMD).
  __shared__ uint4 lds_rng[2][TRACE_BLOCK];
  // Likewise eight words of shading state that the traversal loop and its batch transitions never touch (radiance so far,
  // alpha, the diffuse weight, the node's index of refraction).
  __shared__ uint4 lds_park[2][TRACE_BLOCK];
  uint32_t* const lds_stack = reinterpret_cast<uint32_t*>(lds_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  (void)lane;
  const unsigned char* const heap = reinterpret_cast<const unsigned char*>(h.nodes);
  // (32-bit: the grid has far fewer than 2^31 threads; the 64-bit products below are formed where they are used)
  const uint32_t gid32 = blockIdx.x * (uint32_t)TRACE_BLOCK + (uint32_t)tid;
  const long long gid = (long long)gid32;
  const long long gthreads = (long long)(gridDim.x * (uint32_t)TRACE_BLOCK);
  Counters cn = {0, 0, 0, 0, 0, 0, 0, 0, 0};

  // Work distribution: the sample range is cut into chunks of 2^chunk_shift consecutive samples (16 pixels at 16 spp); a wave
  // takes the next chunk from a global counter whenever its local one is used up (one atomic per chunk), so waves that
  // draw cheap samples simply draw more of them -- this is what keeps the tail short when a GPU renders only a stripe set.
  unsigned long long c_next = 0, c_end = 0;   // wave-uniform: this wave's current chunk
  bool exhausted = false;                      // wave-uniform: the global counter has run past the frame

  Lane S;
  S.g = -1; S.trav = false;
  S.rng.v0 = S.rng.v1 = S.rng.v2 = S.rng.v3 = S.rng.v4 = S.rng.d = 0; S.rng.bm_flag = 0; S.rng.bm_extra = 0.0f;
  S.L = mk3(0, 0, 0); S.alpha = 0.0f; S.steps = 0;
  S.Hdir = mk3(0, 0, 0); S.Hp = mk3(0, 0, 0); S.Hn = mk3(0, 0, 0); S.Hcolor = mk3(0, 0, 0);
  S.Hbounce = 0; S.Hior = 1.458f; S.Hrough = 0.0f; S.HtransNZ = false;
  S.wt = mk3(1, 1, 1); S.wD = mk3(0, 0, 0); S.pn = mk3(0, 0, 0);
  S.pc = 0; S.refr_bounce = 0; S.gi_n = 0; S.state = ST_PRIMARY;
  S.bo = mk3(0, 0, 0); S.rdir = mk3(0, 0, 1); S.li = 0; S.occl = 0ull; S.batch_pending = false; S.has_reflect = false;
  S.o = mk3(0, 0, 0); S.d = mk3(0, 0, 1); S.inv = mk3(0, 0, 1); S.bounce = 0; S.limit = INFINITY; S.shadow = false;
  S.tplane = INFINITY; S.plane_id = -1;
  S.qb = mk3(0, 0, 0); S.qc = mk3(0, 0, 0); S.qsx = S.qsy = S.qsz = 0;
  S.cur = REF_NONE; S.tos = REF_NONE; S.sp = 0; S.tbest = INFINITY; S.refbest = REF_NONE;

  lds_rng[0][tid] = make_uint4(0, 0, 0, 0);
  lds_rng[1][tid] = make_uint4(0, 0, 0, 0);
  lds_park[0][tid] = make_uint4(0, 0, 0, 0);
  lds_park[1][tid] = make_uint4(0, 0, 0, __float_as_uint(1.458f));
  for (;;) {
    // ================= shade / refill phase: lanes that are not traversing =================
    // The shade phase reads the frame's RenderArgs through a pointer the optimiser cannot see through, so that those
    // values are (scalar-)loaded here and do not live in registers across the traversal loop.
    const RenderArgs* aq = ap;
    asm volatile("" : "+s"(aq));
    const RenderArgs& a = *aq;
    // A lane whose nearest-hit walk ended on a sphere the reference may never have tested (hit_needs_literal_walk: a few rays per
    // frame) has its ray walked again here, the reference's way, before the hit is shaded below -- while the sixteen parked words
    // of the lane are still in LDS: with them in registers the walk's own pushed spills into the whole shade phase (+1..3 %).
    // (the lanes vetted are exactly those the loop below lets advance() consume: none is looked at twice)
    // (a batch lane: its reflection ray, or a shadow ray towards a point light -- one that hit something is left to this phase
    // by the traversal loop's header, MIRT_HELD)
    if (a.reach_check && !S.trav && S.g >= 0 && !(exhausted && S.batch_pending && !MIRT_HELD) && (!S.batch_pending || S.li >= a.num_suns)) {
      if (hit_needs_literal_walk<QN>(a, S)) walk_literally<COUNT, NOTRI>(a, S, cn, gid, gthreads);
    }
    {
      const uint4 r0 = lds_rng[0][tid], r1 = lds_rng[1][tid];
      S.rng.v0 = r0.x; S.rng.v1 = r0.y; S.rng.v2 = r0.z; S.rng.v3 = r0.w;
      S.rng.v4 = r1.x; S.rng.d = r1.y; S.rng.bm_extra = __uint_as_float(r1.z); S.rng.bm_flag = (int)r1.w;
      const uint4 p0 = lds_park[0][tid], p1 = lds_park[1][tid];
      S.L = mk3(__uint_as_float(p0.x), __uint_as_float(p0.y), __uint_as_float(p0.z)); S.alpha = __uint_as_float(p0.w);
      S.wD = mk3(__uint_as_float(p1.x), __uint_as_float(p1.y), __uint_as_float(p1.z)); S.Hior = __uint_as_float(p1.w);
    }
    // (once the frame's queue is empty the first ray of a new batch is started by the traversal loop's header instead)
    while (!S.trav && S.g >= 0 && !(exhausted && S.batch_pending && !MIRT_HELD)) {
      if (S.batch_pending) batch_next<COUNT, QN, RenderArgs, SPEC>(a, S, cn);
      else advance<COUNT, QN, SPEC>(a, S, cn, gid, gthreads);
    }
    if (!exhausted) {
      // lanes without a sample take new ones -- once init_k of them wait, or when the wave has nothing else to do: starting a
      // sample (RNG stream set-up, camera ray) costs the wave the same instructions for one lane as for forty
      const unsigned long long need = __ballot(!S.trav && S.g < 0);
      if (need && (__popcll(need) >= a.init_k || __ballot(S.trav || S.batch_pending) == 0)) {
        const int want = __popcll(need);
        const unsigned long long nsamples = (unsigned long long)a.num_samples;
        const unsigned long long chunk = 1ull << a.chunk_shift;
        const unsigned long long nchunks = (nsamples + chunk - 1) >> a.chunk_shift;
        const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
        int given = 0;
        long long my = -1;
        while (given < want) {
          if (c_next >= c_end) {
            unsigned long long k = 0;
            if (lane == 0) {
              k = atomicAdd(a.work_counter, 1ull);
              // longest-first hand-out order measured on an earlier frame (any order gives the same pixels)
              if (k < nchunks && a.chunk_order) k = a.chunk_order[k];
            }
            k = __shfl(k, 0);
            if (k >= nchunks) {
              // Every wave fetches past the end exactly once (and never again): the last one to do so puts both counters back
              // to zero for the next launch on this context -- no memset between launches.
              if (lane == 0 && atomicAdd(a.work_counter + 4, 1ull) == (unsigned long long)gridDim.x * (TRACE_BLOCK / 64) - 1ull) {
                a.work_counter[4] = 0ull;
                a.work_counter[0] = 0ull;
              }
              exhausted = true;
              break;
            }
            c_next = k << a.chunk_shift;
            c_end = (c_next + chunk < nsamples) ? c_next + chunk : nsamples;
          }
          const int avail = (int)(c_end - c_next);
          const int take = (want - given < avail) ? want - given : avail;
          if (!S.trav && S.g < 0 && r >= given && r < given + take) my = (long long)c_next + (r - given);
          c_next += (unsigned long long)take;
          given += take;
        }
        // (sched = 2: the hand-out position names its sample through the cost-ordered table)
        if (my >= 0) init_sample<COUNT, TABLES, QN>(a, S, cn, a.sample_order ? (long long)a.sample_order[my] : my);
      }
    }
    lds_rng[0][tid] = make_uint4(S.rng.v0, S.rng.v1, S.rng.v2, S.rng.v3);
    lds_rng[1][tid] = make_uint4(S.rng.v4, S.rng.d, __float_as_uint(S.rng.bm_extra), (uint32_t)S.rng.bm_flag);
    lds_park[0][tid] = make_uint4(__float_as_uint(S.L.x), __float_as_uint(S.L.y), __float_as_uint(S.L.z), __float_as_uint(S.alpha));
    lds_park[1][tid] = make_uint4(__float_as_uint(S.wD.x), __float_as_uint(S.wD.y), __float_as_uint(S.wD.z), __float_as_uint(S.Hior));
    if (__ballot(S.trav || S.batch_pending) == 0) {
      if (exhausted && __ballot(S.g >= 0) == 0) break;
      continue;
    }

    // ================= traversal phase: traverse_lbvh, bvh_traversal.cu:92-183 =================
    const float tmin = 0.0001f;
    for (;;) {
      const unsigned long long tm = __ballot(S.trav);
      // lanes whose batch ray finished and whose next one this header may start (not MIRT_HELD ones: the shade phase vets them)
      const bool pend = !S.trav && S.batch_pending && !MIRT_HELD;
      const unsigned long long bm = __ballot(pend);
      if (tm == 0 && bm == 0) break;
      // leave when enough lanes are waiting to shade (they cannot progress while the wave keeps traversing); a wave with
      // few live lanes (the drain at the end of the frame) does not wait for company: there the critical path is one
      // lane's bounce chain
      // lanes that wait: to shade, or (finished) for a new sample while the frame still has some
      const int nwait = __popcll(__ballot(!S.trav && !pend && (S.g >= 0 || !exhausted)));
      const int nlive = __popcll(__ballot(S.g >= 0));
      const bool drain = exhausted && nlive <= h.drain_lanes;
      if (nwait >= h.refill_k || (drain && nwait > 0)) break;
      // lanes whose batch ray finished move on to the next ray of their batch (cheap; done in groups)
      if (bm != 0 && (__popcll(bm) >= h.batch_k || tm == 0 || drain)) {
        if (pend) batch_next<COUNT, QN, HotArgs, SPEC>(h, S, cn);
      }
      // A wave executes the node path and the primitive path one after the other whenever its lanes are split between
      // them, and with ~45 live lanes nearly every iteration has a lane or two at a primitive.  So lanes that reach a
      // primitive wait until leaf_k of them have one pending (or nothing else is left to do): the primitive code then
      // runs in a fraction of the iterations.  Each ray still performs exactly the same sequence of steps.
      const bool leaf0 = S.trav && (S.cur & REF_LEAF) != 0;
      const unsigned long long lm = __ballot(leaf0);
      const bool do_leaf0 = __popcll(lm) >= h.leaf_k || __ballot(S.trav && !leaf0) == 0 || drain;
      // `reps` steps per pass through the header above: the bookkeeping is amortised; a lane whose ray ends in the first
      // step idles through the others, and a primitive reached in a later step waits for the next pass
      for (int rep = 0; rep < h.reps; ++rep) {
      const bool leaf = S.trav && (S.cur & REF_LEAF) != 0;
      const bool do_leaf = rep == 0 && do_leaf0;
      if (S.trav && (!leaf || do_leaf)) {
        if (COUNT) ++S.steps;      // (scheduling statistic: the frames that measure their chunks run the counting kernels)
        // one record per step: a node (64 B: two child boxes + two child references) or a primitive (sphere 16 B,
        // triangle 48 B), addressed by one scalar base (the record heap) + a 32-bit byte offset.  Each kind loads the
        // quarters it needs inside its own branch: no merged/zero-filled registers between the two paths.
        const float4* rec = reinterpret_cast<const float4*>(heap + (S.cur << 4));
        bool pop = false;
        if (leaf) {
          // intersect_leaf_primitives, bvh_traversal.cu:47-89
          float t = 0.0f;
          bool hit = false;
          const float4 q0 = rec[0];
          const bool tri = !NOTRI && (S.cur & REF_TRI) != 0;
          if (tri) {
            if (COUNT) cn.tri_tests++;
            const float4 q1 = rec[1], q2 = rec[2];
            hit = triangle_hit(q0, q1, q2, S.o, S.d, t);
          } else {
            if (COUNT) cn.sphere_tests++;
            float tc, t_far;
            hit = sphere_hit(q0, S.o, S.d, t, tc, t_far);
            if (QN && hit && (NOTRI || S.qsx != 0u)) hit = sphere_leaf_box_admits(q0, S.o, S.d, tc, t_far);
          }
          // (S.trav is true here)
          bool closer = closer_hit(hit, t, S.tbest, S.cur & REF_OFFMASK, S.refbest);
          if (QN && !NOTRI && closer && tri && S.qsx != 0u) {
            f3 inv;
            if (!triangle_leaf_reached(ap, S.cur & REF_OFFMASK, S.o, S.d, t, tmin, inv)) {
              // walk this ray again from the root, over the exact records (node 0 sits at heap offset 0), left child first
              closer = false;
              S.inv = inv; S.qsx = 0u;
              S.tos = 0u; S.sp = 1;      // (the pop below makes the root the current node)
              S.tbest = INFINITY; S.refbest = REF_NONE;
            }
          }
          S.tbest = closer ? t : S.tbest;
          S.refbest = closer ? S.cur : S.refbest;
          S.trav = !(closer && S.shadow && t < S.limit);      // any-hit exit
          pop = S.trav;
        } else {
          if (COUNT) cn.internal_visits++;
          // hit_aabb_adapted, bvh_traversal.cu:11-44, on both children
          // (the empty asm keeps the compiler from merging this branch's first load with the primitive branch's and
          // waiting for it before the other three are issued)
          uint32_t noff = S.cur << 4;
          asm volatile("" : "+v"(noff));
          const float4* nrec = reinterpret_cast<const float4*>(heap + noff);
          bool hl, hr;
          float tel, ter;
          uint32_t lref, rref;
          // push `v`: the previous top of stack goes to memory, the new top stays in a register.  With n entries on the stack,
          // entry k < n sits in slot k and entry n is S.tos (slot 0 only ever receives the dead S.tos of an empty stack), so
          // the slot to write is simply the current depth.
#define MIRT_PUSH(v) do { if (S.sp < h.lds_depth) lds_stack[S.sp * TRACE_BLOCK + tid] = S.tos; \
                          else h.stack_spill[(size_t)(S.sp - h.lds_depth) * gthreads + gid] = S.tos; \
                          S.tos = (v); ++S.sp; if (COUNT) cn.max_stack = max(cn.max_stack, (uint32_t)S.sp); } while (0)
          if (WIDE && S.qsx != 0u) {
            // four 16-byte requests: the boxes of up to four grandchildren in the reference's visiting order, their references.
            // The first one hit is descended, the others are pushed last first -- the reference's depth-first order over a
            // superset of the nodes it visits (it tests the children of the right child only when that is popped, against a
            // best distance that may have shrunk meanwhile).  The last push is the common one below.
            const uint4 w0 = *reinterpret_cast<const uint4*>(nrec), w1 = *reinterpret_cast<const uint4*>(nrec + 1);
            const uint4 w2 = *reinterpret_cast<const uint4*>(nrec + 2), w3 = *reinterpret_cast<const uint4*>(nrec + 3);
            const bool h0 = box_q(w0.x, w0.y, w0.z, S.inv, S.qb, S.qc, S.qsx, S.qsy, S.qsz, S.tbest, tmin);
            const bool h1 = box_q(w0.w, w1.x, w1.y, S.inv, S.qb, S.qc, S.qsx, S.qsy, S.qsz, S.tbest, tmin);
            const bool h2 = box_q(w1.z, w1.w, w2.x, S.inv, S.qb, S.qc, S.qsx, S.qsy, S.qsz, S.tbest, tmin);
            const bool h3 = box_q(w2.y, w2.z, w2.w, S.inv, S.qb, S.qc, S.qsx, S.qsy, S.qsz, S.tbest, tmin);
            uint32_t cand = w3.w;
            bool have = h3;
            if (h2 && have) MIRT_PUSH(cand);
            cand = h2 ? w3.z : cand; have = have || h2;
            if (h1 && have) MIRT_PUSH(cand);
            cand = h1 ? w3.y : cand; have = have || h1;
            hl = h0 || have; hr = h0 && have;
            lref = h0 ? w3.x : cand; rref = cand;
            tel = 0.0f; ter = 0.0f;
          } else if (QN && NOTRI) {
            // two 16-byte requests: twelve grid coordinates and the child references; every node of a sphere-only scene may be
            // descended near child first
            const uint4 w0 = *reinterpret_cast<const uint4*>(nrec), w1 = *reinterpret_cast<const uint4*>(nrec + 1);
            box_pair_q(w0, w1.x, w1.y, S.inv, S.qb, S.qc, S.qsx, S.qsy, S.qsz, S.tbest, tmin, hl, hr, tel, ter);
            lref = w1.z; rref = w1.w;
            order_children(hl, hr, tel, ter, NODE_SWAP_ANY | NODE_SWAP_PURE, h.swap_mask, lref, rref);
          } else {
            const float4 q0 = nrec[0], q1 = nrec[1], q2 = nrec[2];
            const uint4 ch = *reinterpret_cast<const uint4*>(nrec + 3);      // child references, NODE_SWAP_* flags
            box_pair(q0, q1, q2, S.o.x, S.o.y, S.o.z, S.inv.x, S.inv.y, S.inv.z, S.tbest, tmin, hl, hr, tel, ter);
            lref = ch.x; rref = ch.y;
            // (with QN every lane in this branch is walking its ray again: the reference's order, left first)
            if (!QN) order_children(hl, hr, tel, ter, ch.z, h.swap_mask, lref, rref);
          }
          // first child next, push the second (bvh_traversal.cu:149-157: left, right), written with selects: one short
          // branch for the push
          const bool both = hl && hr;
          // (no depth check: the stack cannot outgrow STACK_TOTAL.  A Karras tree over 30-bit codes with the index tie-break of
          // lbvh_builder.cu:76-101 is a radix tree over (code, index) keys of 30 + ceil(log2 N) bits, N < 2^28 (checked at scene
          // creation), so it is at most 58 levels deep, and the walk keeps at most one pending sibling per level.  The
          // reference's "stack overflow" warning, bvh_traversal.cu:154-164, is unreachable for the same reason.)
          if (both) MIRT_PUSH(rref);
#undef MIRT_PUSH
          S.cur = hl ? lref : (hr ? rref : S.cur);
          pop = !(hl || hr);
        }
        if (pop) {
          // an empty stack ends the traversal; otherwise the top becomes the current node and the new top is reloaded
          // (a dead read of slot 0 when the stack is now empty)
          S.trav = S.sp != 0;
          S.cur = S.tos;
          S.sp = S.sp > 0 ? S.sp - 1 : 0;
          S.tos = lds_stack[(S.sp < h.lds_depth ? S.sp : 0) * TRACE_BLOCK + tid];
          if (S.sp >= h.lds_depth) S.tos = h.stack_spill[(size_t)(S.sp - h.lds_depth) * gthreads + gid];
        }
      }
      }
    }
  }

  unsigned long long* const counters = COUNT ? ap->counters : nullptr;
  if (COUNT && counters) {
    uint32_t v[9] = {cn.samples, cn.rays, cn.shadow_rays, cn.internal_visits, cn.sphere_tests, cn.tri_tests, cn.mat_fetches, cn.max_stack, cn.traversed};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      unsigned long long x = v[k];
      if (k == 7) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { unsigned long long y = __shfl_xor(x, off); x = x > y ? x : y; }
        if ((tid & 63) == 0) atomicMax(&counters[k], x);
      } else {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if ((tid & 63) == 0) atomicAdd(&counters[k == 8 ? 11 : k], x);      // ([8..10]: work counter, overflow events, scratch)
      }
    }
  }
}
#undef MIRT_HELD

// draw.cu:9-11
MIRT_DEV unsigned char to_uchar_round(float f) { return (unsigned char)(fminf(fmaxf(f, 0.0f), 1.0f) * 255.0f + 0.5f); }
// draw.cu:129-132: plain float -> unsigned char conversion
MIRT_DEV unsigned char to_uchar_trunc(float f)
{
  if (!(f > 0.0f)) return 0;
  if (f >= 255.0f) return 255;
  return (unsigned char)f;
}

MIRT_DEV float4 mean_of(const float4 sum, int spp)
{
  const float inv = 1.0f / (float)spp;
  return make_float4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv);
}
// pixel_color_accum / uchar conversion, draw.cu:191-205 (spp > 1) and draw.cu:120-135 (spp <= 1)
MIRT_DEV void write_pixel(const ResolveArgs& a, long long lq, const float4 sum)
{
  const long long lp = a.pixel_base + lq;
  if (a.accum) {      // mirt_render_accumulate: one add per pixel and call
    const float4 o = a.accum[lp];
    a.accum[lp] = make_float4(o.x + sum.x, o.y + sum.y, o.z + sum.z, o.w + sum.w);
    return;
  }
  const float4 m = a.count > 1 ? mean_of(sum, a.count) : sum;
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

// spp <= 1, and the fallback for P > 64: one thread per pixel
__global__ void __launch_bounds__(RBLOCK) resolve_kernel(const ResolveArgs a)
{
  const long long lp = (long long)blockIdx.x * RBLOCK + threadIdx.x;
  if (lp >= a.num_local_pixels) return;
  float4 m;
  if (a.count <= 1) {
    m = a.samples[lp];
  } else {
    // Sum in the order of `for (mask = P/2; mask > 0; mask /= 2) v += shfl_xor(v, mask)` as lane 0 sees it
    // (draw.cu:181-189), P = next power of two >= spp, absent samples = 0: a pairwise tree over the samples in
    // bit-reversed order.
    int P = 1, lg = 0;
    while (P < a.count) { P <<= 1; ++lg; }
    const float4* s = a.samples + lp * a.count;
    float4 stk[13];
    int top = 0;
    for (int i = 0; i < P; ++i) {
      const int idx = (int)(__brev((unsigned)i) >> (32 - lg));
      float4 x = (idx < a.count) ? s[idx] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      int j = i;
      while (j & 1) {
        --top;
        const float4 l = stk[top];
        x = make_float4(l.x + x.x, l.y + x.y, l.z + x.z, l.w + x.w);
        j >>= 1;
      }
      stk[top++] = x;
    }
    m = stk[0];
  }
  write_pixel(a, lp, m);
}

// The same sum for P <= 64 with one lane per sample: coalesced 16-byte loads, then literally the reference's butterfly
// `for (mask = P/2; mask > 0; mask /= 2) v += shfl_xor(v, mask)` (draw.cu:181-189) inside each group of P lanes; the
// group's lane 0 parks its sum in LDS and the first threads of the block finish the block's pixels (full waves in the
// sRGB code instead of one lane in P).
constexpr int TBLOCK = 1024;
__global__ void __launch_bounds__(TBLOCK) resolve_tree_kernel(const ResolveArgs a, int P, int lg)
{
  __shared__ float4 sums[TBLOCK / 2];
  const int tid = threadIdx.x;
  const int ppb = TBLOCK >> lg;                       // pixels per block
  const int si = tid & (P - 1);
  const long long lp = (long long)blockIdx.x * ppb + (tid >> lg);
  float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (lp < a.num_local_pixels && si < a.count) v = a.samples[lp * a.count + si];
  for (int mask = P >> 1; mask > 0; mask >>= 1) {
    v.x += __shfl_xor(v.x, mask); v.y += __shfl_xor(v.y, mask); v.z += __shfl_xor(v.z, mask); v.w += __shfl_xor(v.w, mask);
  }
  if (si == 0) sums[tid >> lg] = v;
  __syncthreads();
  const long long lq = (long long)blockIdx.x * ppb + tid;
  if (tid < ppb && lq < a.num_local_pixels) write_pixel(a, lq, sums[tid]);
}

// Local pixel lp of part `part_id`'s compact buffer -> its place in the frame: the part holds its stripes (stripe i = rows
// [i stripe_rows, (i + 1) stripe_rows), owned by part i % num_parts) in increasing order, each row-major (include/mirt.h,
// MirtRenderParams).  init_sample_core does the same arithmetic in 32 bits.
__host__ __device__ inline void part_pixel_xy(long long lp, int width, int stripe_rows, int num_parts, int part_id, long long& x, long long& y)
{
  const long long stripe_pixels = (long long)stripe_rows * width;
  const long long ls = lp / stripe_pixels, within = lp - ls * stripe_pixels;
  const long long gs = ls * num_parts + part_id;
  y = gs * stripe_rows + within / width;
  x = within % width;
}

__global__ void __launch_bounds__(RBLOCK) scatter_kernel(const uchar4* __restrict__ part, uchar4* __restrict__ frame, long long n,
                                                         int width, int height, int stripe_rows, int num_parts, int part_id)
{
  const long long lp = (long long)blockIdx.x * RBLOCK + threadIdx.x;
  if (lp >= n) return;
  long long x, y;
  part_pixel_xy(lp, width, stripe_rows, num_parts, part_id, x, y);
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

// ---- longest-first scheduling: order the frame's chunks by the cost measured on a previous frame ------------------------
constexpr int SORT_BINS = 1024;
MIRT_DEV uint32_t cost_bin(uint32_t c) { const uint32_t b = c >> 3; return b < (uint32_t)SORT_BINS ? (uint32_t)(SORT_BINS - 1) - b : 0u; }   // bin 0 = most expensive
// One block: the per-bin counters are LDS atomics (global atomics on a handful of hot bins -- half the frame is sky --
// ran at ~90 per microsecond), and each thread walks a contiguous run of chunks and adds a run of equal bins at once.
__global__ void __launch_bounds__(SORT_BINS) order_kernel(const uint32_t* __restrict__ cost, uint32_t n, uint32_t* __restrict__ order)
{
  __shared__ uint32_t bins[SORT_BINS];
  const int t = threadIdx.x;
  bins[t] = 0;
  __syncthreads();
  const uint32_t per = (n + SORT_BINS - 1) / SORT_BINS;
  const uint32_t lo = min(n, (uint32_t)t * per), hi = min(n, lo + per);
  {
    uint32_t cb = 0, cnt = 0;
    for (uint32_t i = lo; i < hi; ++i) {
      const uint32_t b = cost_bin(cost[i]);
      if (cnt && b != cb) { atomicAdd(&bins[cb], cnt); cnt = 0; }
      cb = b; ++cnt;
    }
    if (cnt) atomicAdd(&bins[cb], cnt);
  }
  __syncthreads();
  const uint32_t own = bins[t];
  for (int o = 1; o < SORT_BINS; o <<= 1) {
    const uint32_t x = (t >= o) ? bins[t - o] : 0;
    __syncthreads();
    bins[t] += x;
    __syncthreads();
  }
  const uint32_t excl = bins[t] - own;
  __syncthreads();
  bins[t] = excl;
  __syncthreads();
  {
    uint32_t cb = 0, cnt = 0, first = lo;
    for (uint32_t i = lo; i <= hi; ++i) {
      const uint32_t b = (i < hi) ? cost_bin(cost[i]) : 0xffffffffu;
      if (cnt && b != cb) {
        const uint32_t base = atomicAdd(&bins[cb], cnt);
        for (uint32_t k = 0; k < cnt; ++k) order[base + k] = first + k;
        cnt = 0;
      }
      if (cnt == 0) first = i;
      cb = b; ++cnt;
    }
  }
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

// sample_tables >= 1: tables of the sequence skips of sample indices [0, sample_tables) (curand_init(1234 + pixel, sample, 0));
// 0: the per-pixel tables of curand_init(1234, pixel, 0).  allow_larger: tables for more sample indices serve as well.
int ensure_rng_tables(RngCache* rc, int sample_tables, long long frame_pixels, hipStream_t stream, RngTablesDev* out, bool allow_larger)
{
  const int spp = sample_tables;
  const long long key = spp >= 1 ? (long long)spp : -frame_pixels;
  if (rc->key != key && !(allow_larger && spp >= 1 && rc->key >= key)) {
    if (spp >= 1) build_sample_tables(spp, rc->host);
    else build_pixel_tables(frame_pixels, 1234, rc->host);
    MIRT_HIP(hipDeviceSynchronize());   // another stream may still be reading the old tables
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
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trace_kernel<false, 8, false>, TRACE_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 4 * 256 / TRACE_BLOCK;
  return prop.multiProcessorCount * per_cu;
}


// One call of mirt_render / mirt_render_accumulate.  The part's pixels are rendered in slabs of at most 2^slab_log2 samples,
// so the per-sample workspace is bounded (4 GiB by default) whatever the frame: BASELINE config 5 (3840x2160 x 256 spp,
// 2.1 G samples) takes 8 slabs instead of a 34 GB buffer.  Each slab is a trace launch + a resolve launch; a slab boundary
// costs one drain of the persistent grid (~1-2 ms per 64 M samples).
//   d_accum == null: pixels are written (mean, sRGB, quantise);  sample_first must be 0 and sample_count max(spp, 1)
//   d_accum != null: the sum of each pixel's samples [sample_first, sample_first + sample_count) is ADDED to d_accum
static int render_impl(MirtScene* sc, const MirtRenderParams* p, void* d_rgba8, void* d_rgba_f32, void* d_accum, int sample_first,
                       int sample_count, hipStream_t stream)
{
  const char* who = d_accum ? "mirt_render_accumulate" : "mirt_render";
  if (!sc->built) { set_error(std::string(who) + ": call mirt_build_lbvh first"); return MIRT_ERR_STATE; }
  const int64_t npix = local_pixels(p);
  if (npix < 0 || p->spp < 0 || (!d_rgba8 && !d_accum)) { set_error(std::string(who) + ": bad parameters"); return MIRT_ERR_ARG; }
  if (sample_first < 0 || sample_count < 1 || (long long)sample_first + sample_count > MAX_SPP) {
    set_error(std::string(who) + ": more than 4096 samples per pixel (mirt_render_accumulate renders any number in several calls of at most 4096)"); return MIRT_ERR_ARG;
  }
  if ((int64_t)p->width * p->height > 0x7fffffffll - 1234) { set_error(std::string(who) + ": frame too large for the 32-bit pixel seed"); return MIRT_ERR_ARG; }
  if (npix == 0) return MIRT_OK;
  if (npix >= 0x7fffffffll || (long long)p->stripe_rows * p->width >= 0x7fffffffll) { set_error(std::string(who) + ": part too large"); return MIRT_ERR_ARG; }
  const bool per_pixel_seed = d_accum != nullptr || p->spp > 1;      // draw.cu:74,162 vs draw.cu:105
  const int sppe = sample_count;
  const Options& opt = sc->opt;
  // slabs: whole pixels, at most 2^slab_log2 samples each
  long long slab_pixels = (1ll << opt.slab_log2) / sppe;
  if (slab_pixels < 1) slab_pixels = 1;
  if (slab_pixels > npix) slab_pixels = npix;
  const int nslabs = (int)((npix + slab_pixels - 1) / slab_pixels);
  const long long slab_samples_max = slab_pixels * sppe;
  const bool count = (p->flags & MIRT_RENDER_COUNTERS) != 0;

  if (!sc->grid_blocks) sc->grid_blocks = grid_blocks(sc->device);      // per scene, i.e. per device
  const int blocks_cached = sc->grid_blocks;
  long long want_blocks = (slab_samples_max + TRACE_BLOCK - 1) / TRACE_BLOCK;
  int blocks = (int)(want_blocks < blocks_cached ? want_blocks : blocks_cached);
  // A small frame (one GPU's stripe set of an 8-GPU job) rendered while another frame is in flight gets half the grid:
  // every wave ends with a drain -- its last samples, few live lanes, 0.5-2.5 ms -- during which it holds its slot, and
  // with two half-grid frames resident at a time there are half as many drains per frame (1/8 of 1080p x 16: 5.8 -> 5.4
  // ms per frame; no gain from 1/4 of a frame up, a loss for a frame rendered alone).
  if (!count && blocks == blocks_cached && slab_samples_max < 20ll * blocks_cached * TRACE_BLOCK) {
    for (int i = 0; i < MIRT_MAX_FRAMES; ++i) {
      const RenderCtx& c = sc->ctx[i];
      if (c.used && c.stream != stream && hipEventQuery(c.ev3) == hipErrorNotReady) { blocks = blocks_cached > 1 ? blocks_cached / 2 : 1; break; }   // (a frame on this same stream does not overlap)
    }
  }
  if (opt.trace_waves >= 1 && opt.trace_waves <= blocks_cached) blocks = opt.trace_waves;
  const size_t gthreads = (size_t)blocks * TRACE_BLOCK;

  // this frame's context; wait for the frame that used it MIRT_MAX_FRAMES renders ago (the frame counter moves only once
  // the frame is actually issued, below)
  RenderCtx& cx = sc->ctx[sc->frame_no % MIRT_MAX_FRAMES];
  if (cx.used) {
    MIRT_HIP(hipEventSynchronize(cx.ev3));
    if (!cx.timed) {   // fold the finished frame's trace-kernel time into the running mean (mirt_get_stats)
      float ms = 0.0f;
      int rc = trace_ms_of(cx, &ms);
      if (rc != MIRT_OK) return rc;
      sc->trace_ms_sum += ms; sc->trace_frames += 1; cx.timed = true;
    }
  }
  // workspace
  if (cx.samples_cap < (size_t)slab_samples_max) {
    MIRT_HIP(hipStreamSynchronize(stream));
    hipFree(cx.samples); cx.samples = nullptr; cx.samples_cap = 0;
    MIRT_HIP(hipMalloc(&cx.samples, sizeof(float4) * (size_t)slab_samples_max));
    cx.samples_cap = (size_t)slab_samples_max;
  }
  const size_t spill_need = (size_t)STACK_TOTAL_WIDE * gthreads;      // (the wide walk pushes up to three entries per two levels)
  if (cx.spill_cap < spill_need) {
    MIRT_HIP(hipStreamSynchronize(stream));
    hipFree(cx.stack_spill); cx.stack_spill = nullptr; cx.spill_cap = 0;
    MIRT_HIP(hipMalloc(&cx.stack_spill, sizeof(uint32_t) * spill_need));
    cx.spill_cap = spill_need;
  }
  const bool need_pending = sc->any_trans || sc->d.gi != 0;
  const int pending_slots = need_pending ? 2 * (sc->d.bounces + (sc->d.gi > 0 ? sc->d.gi : 0) + 2) : 0;
  const size_t pending_need = (size_t)pending_slots * PENDING_WORDS * gthreads;
  if (cx.pending_cap < pending_need) {
    MIRT_HIP(hipStreamSynchronize(stream));
    hipFree(cx.pending); cx.pending = nullptr; cx.pending_cap = 0;
    MIRT_HIP(hipMalloc(&cx.pending, sizeof(float) * pending_need));
    cx.pending_cap = pending_need;
  }

  RenderArgs a;
  memset(&a, 0, sizeof(a));
  a.width = p->width; a.height = p->height; a.bounces = sc->d.bounces; a.gi = sc->d.gi;
  a.spp = d_accum ? (p->spp > 1 ? p->spp : 2) : p->spp;      // only "is it >= 1" matters to the kernel: jittered samples (draw.cu:78-84,110-118,165-171)
  a.fisheye = sc->d.fisheye; a.panorama = sc->d.panorama;
  a.dof_focus = sc->d.dof_focus; a.dof_lens = sc->d.dof_lens; a.expose = sc->d.expose;
  a.forward.x = sc->d.forward.x; a.forward.y = sc->d.forward.y; a.forward.z = sc->d.forward.z;
  a.right.x = sc->d.right.x; a.right.y = sc->d.right.y; a.right.z = sc->d.right.z;
  a.up.x = sc->d.up.x; a.up.y = sc->d.up.y; a.up.z = sc->d.up.z;
  a.eye.x = sc->d.eye.x; a.eye.y = sc->d.eye.y; a.eye.z = sc->d.eye.z;
  a.stripe_rows = p->stripe_rows; a.num_parts = p->num_parts; a.part = p->part;
  a.sample_first = sample_first; a.sample_count = sample_count; a.seed_per_pixel = per_pixel_seed ? 1 : 0;
  a.nodes = sc->nodes; a.unit_prim = sc->unit_prim; a.mats = sc->mats;
  // Quantised node records: single-kernel path, any order but the reference's own.  A sphere-only scene: the 32-byte records,
  // always.  A scene with triangles: the wide records -- the reference's order at every node, so traversal = 1 only -- when the
  // scene is large enough for memory to matter (qnodes = 1: N >= 65536, the exact records no longer fit an L2; redchair.txt's
  // 1.7 k primitives are 12 % faster on the exact records, the 2 M-primitive scene 25 % faster on the wide ones) or always (2).
  const bool notri = sc->Nt == 0;
  const bool qwant = opt.qnodes != 0 && opt.wavefront == 0;
  // (and only if the grid of the quantised records resolves the scene's coordinates: grid_ok, lbvh_build.hip -- a scene that
  // sits hundreds of its own extents away from the world origin walks the exact records, in the reference's order)
  const bool qn = sc->grid_ok && (notri ? (qwant && opt.traversal >= 1 && sc->root_ref_q != REF_NONE)
                                        : (qwant && opt.traversal == 1 && sc->root_ref_w != REF_NONE && (opt.qnodes >= 2 || sc->N >= 65536)));
  // kernels specialised for what the scene does not have (SPEC_*, shade_common.h)
  const bool nobulb = opt.specialise != 0 && sc->d.num_bulbs == 0, nopend = opt.specialise != 0 && !need_pending;
  a.root_ref = qn ? (notri ? sc->root_ref_q : sc->root_ref_w) : sc->root_ref; a.num_spheres = sc->Ns; a.num_prims = sc->N;
  a.qparams = qn ? sc->qparams : nullptr;
  a.tri_boxes = sc->tri_boxes;
  a.prim_base16 = sc->prim_base / 16u;
  // traversal = 1: near child first on the quantised records of a sphere-only scene, nowhere else.  Over the exact boxes the
  // reordered walk can cull a box over a sphere whose hit distance rounds below that box's entry distance (one ulp is enough; the
  // reference, in its order, gets there first): 13 of 4 000 far-camera fuzz scenes differed by a pixel or a ray.  The quantised
  // boxes are rounded outwards by more than that rounding as long as the grid resolves it (grid_ok, a condition of qn) -- no differing byte
  // in 10 000 sphere scenes, 3 000 of them far-camera ones.  Everything else walks in the reference's order.
  a.swap_mask = opt.traversal == 1 ? ((qn && notri) ? NODE_SWAP_PURE : 0u) : (opt.traversal == 2 ? NODE_SWAP_ANY : 0u);
  a.skip_unlit = (opt.skip_unlit != 0 && sc->colors_finite && sc->d.num_suns + sc->d.num_bulbs <= 32) ? 1 : 0;
  a.shadow_anyhit = opt.shadow_anyhit != 0 ? 1 : 0;
  a.planes = sc->planes; a.num_planes = sc->d.num_planes;
  a.suns = sc->suns; a.num_suns = sc->d.num_suns;
  a.bulbs = sc->bulbs; a.num_bulbs = sc->d.num_bulbs;
  // random numbers are consumed only by jitter (spp >= 1), depth of field, rough normals and GI
  a.needs_rng = (a.spp >= 1) || (sc->d.dof_focus != 0.0f && !sc->d.fisheye && !sc->d.panorama) || sc->any_rough || sc->d.gi != 0;
  if (a.needs_rng) {
    int rc = ensure_rng_tables(&sc->rng, per_pixel_seed ? sample_first + sample_count : 0, (long long)p->width * p->height, stream, &a.rng, d_accum != nullptr);
    if (rc != MIRT_OK) return rc;
  }
  a.samples = cx.samples;
  a.stack_spill = cx.stack_spill;
  a.pending = cx.pending; a.pending_slots = pending_slots;
  a.counters = count ? cx.counters : nullptr;
  a.overflow = cx.counters + 9;
  a.lds_depth = (opt.stack_lds_depth >= 0 && opt.stack_lds_depth <= STACK_LDS) ? opt.stack_lds_depth : STACK_LDS;   // tests force the spill path
  // Thresholds of the two expensive divergent pieces of work, measured per kind of kernel (round 3, tools/r03_i.sh, r03_x.sh): lanes
  // wait to shade until refill_k of them do, lanes without a sample until init_k of them do.  Sphere-only scenes 32 / 10; wide
  // records (2 M-primitive scene) 24 / 8; exact records (redchair.txt) 64 / 64 -- with the samples handed out by cost class
  // (sched = 2) the lanes of a wave run samples of one kind, and redchair.txt's short ray trees are fastest in lock step: the whole
  // wave traverses, the whole wave shades, the whole wave takes 64 new samples (1080p16: 20.6 ms at 52 / 48, 18.1 at 64 / 64;
  // tenthousand.txt's deep reflection chains want the opposite: 28.0 ms at 64 / 64 against 21.7).  Refilling finished lanes at
  // every shade phase (init_k = 1, rounds 1-2) cost redchair.txt 14 % of its frame, the sphere scenes 1.5 %.
  a.refill_k = opt.refill_k > 0 ? opt.refill_k : (qn ? (notri ? 32 : 24) : 64);
  a.drain_lanes = opt.drain_lanes;
  const int init_k = opt.init_k > 0 ? opt.init_k : (qn ? (notri ? 10 : 8) : 64);
  a.init_k = init_k < a.refill_k ? init_k : a.refill_k;      // (<= refill_k: lanes waiting for a sample count as waiting in the loop header)
  a.batch_k = opt.batch_k;

  // ---- longest-first chunk order (single-kernel path, one-slab calls) ------------------------------------------------
  // chunk size: 256 samples, smaller for a small (part of a) frame so that every wave still gets a dozen chunks or more --
  // with four chunks per wave (1/8 of a 1080p frame) the waves finished up to a chunk apart
  int chunk_shift = MAX_CHUNK_SHIFT;
  while (chunk_shift > MIN_CHUNK_SHIFT && (slab_samples_max >> chunk_shift) < 16ll * blocks * (TRACE_BLOCK / 64)) --chunk_shift;
  if (opt.chunk_shift >= 4) chunk_shift = opt.chunk_shift;
  a.chunk_shift = chunk_shift;
  const size_t nchunks = (size_t)((slab_samples_max + (1ll << chunk_shift) - 1) >> chunk_shift);
  // sched = 1 (by chunk): one-slab calls only; sched = 2 (by sample): any call
  const bool wavefront = opt.wavefront != 0;
  const bool sched = opt.sched == 1 && nslabs == 1 && !wavefront;
  if (sched && cx.chunk_cap < nchunks) {
    MIRT_HIP(hipDeviceSynchronize());   // a frame on another stream may still be reading one of these orders
    hipFree(cx.chunk_cost); cx.chunk_cost = nullptr; cx.chunk_cap = 0; cx.order_key = -1;
    for (uint32_t*& o : cx.order_out) { hipFree(o); o = nullptr; }
    MIRT_HIP(hipMalloc(&cx.chunk_cost, 4 * nchunks));
    for (uint32_t*& o : cx.order_out) MIRT_HIP(hipMalloc(&o, 4 * nchunks));
    cx.chunk_cap = nchunks;
  }
  // The newest finished frame with the same sample count (and chunk size) that MEASURED its chunks provides the order; a frame
  // that is still running does not.  The scene is immutable and samples are seeded by pixel and sample index only, so a frame's
  // chunk costs are the same every time: once an order exists for this frame size it is reused, and the frame neither stamps
  // costs nor sorts them again (the sort took 0.34 ms of a 24 ms frame, on the stream's critical path).
  const long long okey = slab_samples_max * 16 + chunk_shift;
  const uint32_t* order = nullptr;
  if (sched) {
    unsigned long long best = 0;
    for (int i = 0; i < MIRT_MAX_FRAMES; ++i) {
      RenderCtx& c = sc->ctx[i];
      if (&c == &cx || !c.used || c.order_key != okey || c.order_frame <= best) continue;
      if (hipEventQuery(c.order_ev) != hipSuccess) continue;
      best = c.order_frame; order = c.order_out[(c.order_writes - 1) % RenderCtx::ORDER_BUFS];
    }
    if (!order && cx.used && cx.order_key == okey) order = cx.order_out[(cx.order_writes - 1) % RenderCtx::ORDER_BUFS];   // cx's own earlier frame (finished: synchronised above)
  }
  // sched = 2 (default): the same idea by SAMPLE -- the samples of every launch in order of decreasing cost class, positions within
  // a class kept (a stable one-byte radix pass: neighbours in the frame stay neighbours in the hand-out, and the lanes of a wave
  // work on samples of one kind).  One table per scene (4 B per sample of the call, at most 12 GiB), for the call shape rendered
  // last; measured by the first call of that shape (counting kernels + one sort per launch), reused afterwards.  Against the
  // chunk order: a 1/8 stripe share of the headline frame 4.16 -> 3.22 ms alone (its last expensive samples no longer start
  // late), redchair.txt 1080p16 23.8 -> 20.5 ms, tenthousand.txt 22.5 -> 21.7.
  const long long total_samples = (long long)npix * sppe;
  const bool by_sample = opt.sched == 2 && !wavefront && slab_samples_max < 0x7fffffffll && total_samples <= (3ll << 30);
  bool measure_samples = false, ordered_samples = false;
  a.sample_order = nullptr; a.sample_key = nullptr;
  if (by_sample) {
    order = nullptr;
    if (sc->so_busy && hipEventQuery(sc->so_ev) == hipSuccess) sc->so_busy = false;      // the measurement in flight has landed
    // (the table is a permutation of every launch's sample range: valid for exactly this total and this slab size)
    if (sc->so_key == okey && sc->so_total == total_samples && !sc->so_busy) ordered_samples = true;
    else if (!sc->so_busy) {
      // (no frame in flight reads the old table once every context's frame has finished: wait for them before rewriting it)
      for (int i = 0; i < MIRT_MAX_FRAMES; ++i) if (sc->ctx[i].used) MIRT_HIP(hipEventSynchronize(sc->ctx[i].ev3));
      if (sc->so_cap < (size_t)total_samples || sc->so_slab_cap < (size_t)slab_samples_max) {
        hipFree(sc->so_order); hipFree(sc->so_keys); hipFree(sc->so_keys2); hipFree(sc->so_ws);
        sc->so_order = sc->so_keys = sc->so_keys2 = sc->so_ws = nullptr; sc->so_cap = 0; sc->so_slab_cap = 0; sc->so_key = -1;
        // (the table is an optimisation: without the memory for it the call is rendered in frame order)
        const bool got = hipMalloc(&sc->so_order, 4 * (size_t)total_samples) == hipSuccess && hipMalloc(&sc->so_keys, 4 * (size_t)slab_samples_max) == hipSuccess &&
                         hipMalloc(&sc->so_keys2, 4 * (size_t)slab_samples_max) == hipSuccess && hipMalloc(&sc->so_ws, 4 * sort_low_byte_ws_words(slab_samples_max)) == hipSuccess;
        if (got) { sc->so_cap = (size_t)total_samples; sc->so_slab_cap = (size_t)slab_samples_max; }
        else {
          (void)hipGetLastError();
          hipFree(sc->so_order); hipFree(sc->so_keys); hipFree(sc->so_keys2); hipFree(sc->so_ws);
          sc->so_order = sc->so_keys = sc->so_keys2 = sc->so_ws = nullptr;
        }
      }
      if (sc->so_cap >= (size_t)total_samples && sc->so_slab_cap >= (size_t)slab_samples_max) {
        measure_samples = true;
        sc->so_key = -1;
        sc->so_pending_key = okey; sc->so_total = total_samples;
      }
    }
  }
  const bool measure = (sched && !by_sample && !order) || measure_samples;
  a.chunk_order = order;
  a.chunk_cost = (measure && !by_sample) ? cx.chunk_cost : nullptr;
  cx.frame_id = ++sc->frame_seq;
  MIRT_HIP(hipEventRecord(cx.ev0, stream));
  if (measure && !by_sample) MIRT_HIP(hipMemsetAsync(cx.chunk_cost, 0, 4 * nchunks, stream));
  if (count) { MIRT_HIP(hipMemsetAsync(cx.counters, 0, 8 * sizeof(unsigned long long), stream)); MIRT_HIP(hipMemsetAsync(cx.counters + 11, 0, sizeof(unsigned long long), stream)); }
  a.work_counter = cx.counters + 8;
  cx.wf_trace_ms = -1.0f;
  float wf_ms_total = 0.0f;
  HotArgs h;
  h.nodes = a.nodes; h.root_ref = a.root_ref; h.swap_mask = a.swap_mask; h.qparams = a.qparams;
  h.planes = a.planes; h.num_planes = a.num_planes; h.suns = a.suns; h.num_suns = a.num_suns; h.bulbs = a.bulbs; h.num_bulbs = a.num_bulbs; h.shadow_anyhit = a.shadow_anyhit;
  h.stack_spill = a.stack_spill; h.lds_depth = a.lds_depth; h.refill_k = a.refill_k; h.batch_k = a.batch_k; h.drain_lanes = a.drain_lanes;
  // (the quantised walks; the exact records are walked in the reference's own order, or -- traversal = 2 -- in one that promises nothing)
  a.reach_check = (qn && sc->N > 1) ? 1 : 0;
  a.reach_slack = 4.76837158203125e-07f * sc->coord_max;
  h.reach_check = a.reach_check;
  h.leaf_k = opt.leaf_k > 0 ? opt.leaf_k : (qn ? 8 : 4);      // (exact records, redchair.txt: 4 is 1.3 % better than 8)
  h.reps = opt.reps > 0 ? opt.reps : ((qn && !notri) ? 5 : 4);      // (wide records: 5 is 1 % better on the 2 M-primitive scene, worse elsewhere)
  if (!wavefront && !cx.args_dev) MIRT_HIP(hipMalloc(&cx.args_dev, sizeof(RenderArgs) * MAX_SLAB_ARGS));
  int P = 1, lg = 0;
  while (P < sample_count) { P <<= 1; ++lg; }

  MIRT_HIP(hipEventRecord(cx.ev1, stream));
  // one event pair per trace launch: a call of several slabs reports the SUM of its launches, not a bracket that would take in
  // the resolve kernels between them
  while ((int)cx.slab_ev.size() < 2 * nslabs) { hipEvent_t e = nullptr; MIRT_HIP(hipEventCreate(&e)); cx.slab_ev.push_back(e); }
  cx.launches = nslabs;
  cx.node_bytes = (qn && notri) ? 32 : 64;
  for (int slab = 0; slab < nslabs; ++slab) {
    const long long p0 = (long long)slab * slab_pixels;
    const long long pn = (p0 + slab_pixels < npix ? p0 + slab_pixels : npix) - p0;
    a.pixel_base = p0; a.num_local_pixels = pn; a.num_samples = pn * sppe;
    a.sample_order = ordered_samples ? sc->so_order + (size_t)p0 * sppe : nullptr;      // (this launch's part of the table)
    a.sample_key = measure_samples ? sc->so_keys : nullptr;
    // (the work counter, counters[8], and the count of waves that ran past its end, counters[12], are left at zero by the launch
    // itself; counters[9], the overflow events, is only reset by mirt_get_stats)
    if (wavefront) {
      a.refill_k = opt.wf_refill_k;
      float tms = 0.0f;
      int rc = wavefront_trace(sc, cx, a, count, stream, &tms);
      if (rc != MIRT_OK) return rc;
      wf_ms_total += tms;
      cx.wf_trace_ms = wf_ms_total;
    } else {
      // this slab's arguments: the kernel reads them from device memory; a ring of copies, so that the copy for slab k + 1
      // does not wait for the kernel of slab k (the stream orders a copy after the kernel that used the same slot)
      // (a frame like the one before it finds its arguments in place: nothing is copied)
      const int slot = slab % MAX_SLAB_ARGS;
      RenderArgs* adev = cx.args_dev + slot;
      if (!cx.args_valid[slot] || memcmp(&cx.args_host[slot], &a, sizeof(RenderArgs)) != 0) {
        MIRT_HIP(hipMemcpyAsync(adev, &a, sizeof(RenderArgs), hipMemcpyHostToDevice, stream));
        cx.args_host[slot] = a; cx.args_valid[slot] = true;
      }
      MIRT_HIP(hipEventRecord(cx.slab_ev[2 * slab], stream));
      // one instantiation per form of the random-number tables (device_common.h, xw_init) and per node format
      {
        const bool t8 = a.needs_rng && a.rng.mode == 0 && a.rng.chunk_bits == 8;
#define MIRT_LAUNCH(C, T, Q, P) hipLaunchKernelGGL((trace_kernel<C, T, Q, P>), dim3(blocks), dim3(TRACE_BLOCK), 0, stream, adev, h)
#define MIRT_LAUNCH_T(C, Q, P) do { if (t8) MIRT_LAUNCH(C, 8, Q, P); else MIRT_LAUNCH(C, 4, Q, P); } while (0)
#define MIRT_LAUNCH_S(C, Q) do { if (nobulb && nopend) MIRT_LAUNCH_T(C, Q, SPEC_NOBULB | SPEC_NOPEND); else if (nobulb) MIRT_LAUNCH_T(C, Q, SPEC_NOBULB); \
                                 else MIRT_LAUNCH_T(C, Q, 0); } while (0)
#define MIRT_LAUNCH_Q(C) do { if (qn && notri && nobulb && nopend) MIRT_LAUNCH_T(C, true, SPEC_NOTRI | SPEC_NOBULB | SPEC_NOPEND); \
                              else if (qn && notri) MIRT_LAUNCH_T(C, true, SPEC_NOTRI); \
                              else if (qn) MIRT_LAUNCH_S(C, true); else MIRT_LAUNCH_S(C, false); } while (0)
        if (count || measure) MIRT_LAUNCH_Q(true); else MIRT_LAUNCH_Q(false);      // (measure: cost stamps need the step counter)
#undef MIRT_LAUNCH_Q
#undef MIRT_LAUNCH_S
#undef MIRT_LAUNCH_T
#undef MIRT_LAUNCH
      }
    }
    MIRT_HIP(hipGetLastError());
    if (!wavefront) MIRT_HIP(hipEventRecord(cx.slab_ev[2 * slab + 1], stream));
    if (slab == nslabs - 1) MIRT_HIP(hipEventRecord(cx.ev2, stream));

    ResolveArgs ra;
    ra.samples = cx.samples; ra.rgba8 = (unsigned char*)d_rgba8; ra.rgba_f32 = (float4*)d_rgba_f32; ra.accum = (float4*)d_accum;
    ra.num_local_pixels = pn; ra.pixel_base = p0; ra.spp = p->spp; ra.count = sample_count;
    if (sample_count > 1 && P <= 64) {
      const long long ppb = TBLOCK >> lg;
      hipLaunchKernelGGL(resolve_tree_kernel, dim3((unsigned)((pn + ppb - 1) / ppb)), dim3(TBLOCK), 0, stream, ra, P, lg);
    } else {
      hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((pn + RBLOCK - 1) / RBLOCK)), dim3(RBLOCK), 0, stream, ra);
    }
    MIRT_HIP(hipGetLastError());
    if (measure_samples) {
      // this launch's part of the cost-ordered sample table, for later calls of this shape (the key buffer is reused by the next slab)
      int rc = sort_low_byte(sc->so_keys, sc->so_keys2, sc->so_order + (size_t)p0 * sppe, pn * sppe, sc->so_ws, stream);
      if (rc != MIRT_OK) return rc;
    }
  }
  if (measure_samples) {
    MIRT_HIP(hipEventRecord(sc->so_ev, stream));
    sc->so_key = sc->so_pending_key; sc->so_busy = true;
  } else if (measure) {
    // order for later frames.  It overwrites the buffer this context wrote three orders ago; every frame that could have read
    // that one has finished -- the host waited for each of them when it reused their contexts.
    uint32_t* out = cx.order_out[cx.order_writes % RenderCtx::ORDER_BUFS];
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(SORT_BINS), 0, stream, cx.chunk_cost, (uint32_t)nchunks, out);
    MIRT_HIP(hipGetLastError());
    MIRT_HIP(hipEventRecord(cx.order_ev, stream));
    cx.order_key = okey;
    cx.order_frame = cx.frame_id;
    ++cx.order_writes;
  }
  ++cx.uses;
  ++sc->frame_no;
  MIRT_HIP(hipEventRecord(cx.ev3, stream));
  cx.used = true; cx.counted = count; cx.timed = false; cx.stream = stream; sc->last = &cx;
  return MIRT_OK;
}

// trace-kernel time of the context's last (finished) call: the sum over its launches
int trace_ms_of(RenderCtx& cx, float* ms)
{
  *ms = 0.0f;
  if (cx.wf_trace_ms >= 0.0f) { *ms = cx.wf_trace_ms; return MIRT_OK; }
  for (int k = 0; k < cx.launches && 2 * k + 1 < (int)cx.slab_ev.size(); ++k) {
    float t = 0.0f;
    MIRT_HIP(hipEventElapsedTime(&t, cx.slab_ev[2 * k], cx.slab_ev[2 * k + 1]));
    *ms += t;
  }
  return MIRT_OK;
}

int render(MirtScene* sc, const MirtRenderParams* p, void* d_rgba8, void* d_rgba_f32, hipStream_t stream)
{
  if (p->spp > MAX_SPP) { set_error("mirt_render: more than 4096 samples per pixel in one call (use mirt_render_accumulate + mirt_finalize)"); return MIRT_ERR_ARG; }
  if (!d_rgba8) { set_error("mirt_render: bad parameters"); return MIRT_ERR_ARG; }
  return render_impl(sc, p, d_rgba8, d_rgba_f32, nullptr, 0, p->spp > 1 ? p->spp : 1, stream);
}

// render_kernel_atomic_aa, draw.cu:49-92: adds the samples [sample_first, sample_first + sample_count) of every pixel of the
// part to d_accum (float4 per pixel of the compact part buffer).  The reference adds sample by sample with atomicAdd, i.e. in
// no particular order; here a call adds one value per pixel, the sum of its samples in the xor-butterfly order of draw.cu:181-189.
int render_accumulate(MirtScene* sc, const MirtRenderParams* p, void* d_accum, int sample_first, int sample_count, hipStream_t stream)
{
  if (!d_accum) { set_error("mirt_render_accumulate: bad parameters"); return MIRT_ERR_ARG; }
  return render_impl(sc, p, nullptr, nullptr, d_accum, sample_first, sample_count, stream);
}

__global__ void __launch_bounds__(RBLOCK) finalize_kernel_dev(const float4* __restrict__ accum, uchar4* __restrict__ rgba8, long long n, int aa)
{
  // finalize_kernel, draw.cu:13-47
  const long long i = (long long)blockIdx.x * RBLOCK + threadIdx.x;
  if (i >= n) return;
  const float4 m = mean_of(accum[i], aa);
  uchar4 o;
  o.x = to_uchar_round(rgb_to_srgb(m.x)); o.y = to_uchar_round(rgb_to_srgb(m.y)); o.z = to_uchar_round(rgb_to_srgb(m.z)); o.w = to_uchar_round(m.w);
  rgba8[i] = o;
}

int finalize(const MirtRenderParams* p, const void* d_accum, int total_samples, void* d_rgba8, hipStream_t stream)
{
  const int64_t n = local_pixels(p);
  if (n < 0 || !d_accum || !d_rgba8 || total_samples < 1) { set_error("mirt_finalize: bad parameters"); return MIRT_ERR_ARG; }
  if (n == 0) return MIRT_OK;
  hipLaunchKernelGGL(finalize_kernel_dev, dim3((unsigned)((n + RBLOCK - 1) / RBLOCK)), dim3(RBLOCK), 0, stream, (const float4*)d_accum, (uchar4*)d_rgba8, (long long)n, total_samples);
  MIRT_HIP(hipGetLastError());
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

int part_pixel(const MirtRenderParams* p, int64_t local, int32_t* x, int32_t* y)
{
  const int64_t n = local_pixels(p);
  if (n < 0 || local < 0 || local >= n || !x || !y) { set_error("mirt_part_pixel_xy: bad argument"); return MIRT_ERR_ARG; }
  long long xx, yy;
  part_pixel_xy(local, p->width, p->stripe_rows, p->num_parts, p->part, xx, yy);
  *x = (int32_t)xx; *y = (int32_t)yy;
  return MIRT_OK;
}

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
  int rc = ensure_rng_tables(&cache, spp > 1 ? spp : 0, nstreams, nullptr, &t, false);
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
