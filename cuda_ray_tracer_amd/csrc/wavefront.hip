// Wavefront render path: the same shading state machine as render.hip's single kernel (shade_common.h), split into
//
//   wf_shade_kernel   one thread per pool slot.  A slot owns one sample at a time: it consumes the results of the rays it
//                     emitted in the previous round, shades (advance_core), emits the next rays -- a node's whole batch at
//                     once: its shadow rays and its reflection ray -- into a compact ray queue (wave-aggregated append),
//                     and, when its sample is finished, writes the RGBA and takes the next sample of the frame.
//   wf_trace_kernel   lean persistent kernel (64-VGPR class, 16-entry LDS stacks): waves pull rays from the queue in
//                     chunks; lanes whose ray is finished write its result and are refilled with `__ballot` +
//                     prefix-popcount compaction, so the traversal loop keeps its lanes busy.
//
// alternating until every sample of the frame is done.  Geometry, random-number consumption and counters are identical
// to the single-kernel path (the parity tests run both).
#include "scene_dev.h"
#include "host_scene.h"
#include "shade_common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mirt {
namespace {

constexpr int WBLOCK = 256;
constexpr int SBLOCK = 512;           // shade kernel block: one queue-append atomic per 512 slots
constexpr int WF_STACK_LDS = 16;
constexpr int WF_CHUNK = 256;          // rays a wave takes from the queue per atomic
#ifndef MIRT_WF_SHADE_WAVES
#define MIRT_WF_SHADE_WAVES 2
#endif
#ifndef MIRT_WF_WAVES_PER_SIMD
#define MIRT_WF_WAVES_PER_SIMD 8
#endif

// slot state, SoA: word w of slot s lives at state[w * pool + s]
enum : int {
  W_G_LO = 0, W_G_HI, W_RNG0, W_RNG1, W_RNG2, W_RNG3, W_RNG4, W_RNGD, W_BMFLAG, W_BMEXTRA,
  W_LX, W_LY, W_LZ, W_ALPHA,
  W_HDIR, W_HP = W_HDIR + 3, W_HN = W_HP + 3, W_HCOLOR = W_HN + 3, W_HBOUNCE = W_HCOLOR + 3, W_HIOR, W_HROUGH, W_FLAGS,
  W_WT, W_WD = W_WT + 3, W_PN = W_WD + 3, W_PC = W_PN + 3, W_REFRB, W_GIN, W_STATE,
  W_RO, W_RD = W_RO + 3, W_RBOUNCE = W_RD + 3,
  W_TBEST, W_REFBEST, W_TPLANE, W_PLANEID, W_OCCL_LO, W_OCCL_HI,
  W_NEXT_LO, W_NEXT_HI,             // next sample this slot will take (slot, slot + pool, slot + 2 pool, ...)
  W_COUNT
};

struct WfArgs {
  RenderArgs r;
  int pool;                         // number of slots
  uint32_t* state;                  // [W_COUNT][pool]
  float4* rays;                     // 2 x float4 per ray: (o.xyz, limit) (d.xyz, meta); meta = slot | j << 24 | shadow << 30
  unsigned long long* ctr;          // [0] next sample  [1] rays emitted this round  [2] trace queue head  [3] samples done
  unsigned int max_rays;
};

MIRT_DEV void st_f3(uint32_t* st, int pool, int slot, int w, const f3& v)
{
  st[(size_t)(w + 0) * pool + slot] = __float_as_uint(v.x);
  st[(size_t)(w + 1) * pool + slot] = __float_as_uint(v.y);
  st[(size_t)(w + 2) * pool + slot] = __float_as_uint(v.z);
}
MIRT_DEV f3 ld_f3(const uint32_t* st, int pool, int slot, int w)
{
  return mk3(__uint_as_float(st[(size_t)(w + 0) * pool + slot]), __uint_as_float(st[(size_t)(w + 1) * pool + slot]),
             __uint_as_float(st[(size_t)(w + 2) * pool + slot]));
}
#define WF_LD(w) (st[(size_t)(w) * pool + slot])
#define WF_ST(w, v) (st[(size_t)(w) * pool + slot] = (v))

MIRT_DEV void load_slot(const uint32_t* st, int pool, int slot, Lane& S)
{
  S.g = (int)WF_LD(W_G_LO);      // (the high word is its sign extension)
  S.rng.v0 = WF_LD(W_RNG0); S.rng.v1 = WF_LD(W_RNG1); S.rng.v2 = WF_LD(W_RNG2); S.rng.v3 = WF_LD(W_RNG3); S.rng.v4 = WF_LD(W_RNG4);
  S.rng.d = WF_LD(W_RNGD); S.rng.bm_flag = (int)WF_LD(W_BMFLAG); S.rng.bm_extra = __uint_as_float(WF_LD(W_BMEXTRA));
  S.L = ld_f3(st, pool, slot, W_LX); S.alpha = __uint_as_float(WF_LD(W_ALPHA));
  S.Hdir = ld_f3(st, pool, slot, W_HDIR); S.Hp = ld_f3(st, pool, slot, W_HP); S.Hn = ld_f3(st, pool, slot, W_HN);
  S.Hcolor = ld_f3(st, pool, slot, W_HCOLOR);
  S.Hbounce = (int)WF_LD(W_HBOUNCE); S.Hior = __uint_as_float(WF_LD(W_HIOR)); S.Hrough = __uint_as_float(WF_LD(W_HROUGH));
  const uint32_t fl = WF_LD(W_FLAGS);
  S.HtransNZ = fl & 1u; S.has_reflect = (fl >> 1) & 1u;
  S.wt = ld_f3(st, pool, slot, W_WT); S.wD = ld_f3(st, pool, slot, W_WD); S.pn = ld_f3(st, pool, slot, W_PN);
  S.pc = (int)WF_LD(W_PC); S.refr_bounce = (int)WF_LD(W_REFRB); S.gi_n = (int)WF_LD(W_GIN); S.state = (int)WF_LD(W_STATE);
  S.o = ld_f3(st, pool, slot, W_RO); S.d = ld_f3(st, pool, slot, W_RD); S.bounce = (int)WF_LD(W_RBOUNCE);
  S.tbest = __uint_as_float(WF_LD(W_TBEST)); S.refbest = WF_LD(W_REFBEST);
  S.tplane = __uint_as_float(WF_LD(W_TPLANE)); S.plane_id = (int)WF_LD(W_PLANEID);
  S.occl = ((unsigned long long)WF_LD(W_OCCL_HI) << 32) | WF_LD(W_OCCL_LO);
}

MIRT_DEV void store_slot(uint32_t* st, int pool, int slot, const Lane& S)
{
  WF_ST(W_G_LO, (uint32_t)S.g); WF_ST(W_G_HI, S.g < 0 ? 0xffffffffu : 0u);
  if (S.g < 0) return;
  WF_ST(W_RNG0, S.rng.v0); WF_ST(W_RNG1, S.rng.v1); WF_ST(W_RNG2, S.rng.v2); WF_ST(W_RNG3, S.rng.v3); WF_ST(W_RNG4, S.rng.v4);
  WF_ST(W_RNGD, S.rng.d); WF_ST(W_BMFLAG, (uint32_t)S.rng.bm_flag); WF_ST(W_BMEXTRA, __float_as_uint(S.rng.bm_extra));
  st_f3(st, pool, slot, W_LX, S.L); WF_ST(W_ALPHA, __float_as_uint(S.alpha));
  st_f3(st, pool, slot, W_HDIR, S.Hdir); st_f3(st, pool, slot, W_HP, S.Hp); st_f3(st, pool, slot, W_HN, S.Hn);
  st_f3(st, pool, slot, W_HCOLOR, S.Hcolor);
  WF_ST(W_HBOUNCE, (uint32_t)S.Hbounce); WF_ST(W_HIOR, __float_as_uint(S.Hior)); WF_ST(W_HROUGH, __float_as_uint(S.Hrough));
  WF_ST(W_FLAGS, (S.HtransNZ ? 1u : 0u) | (S.has_reflect ? 2u : 0u));
  st_f3(st, pool, slot, W_WT, S.wt); st_f3(st, pool, slot, W_WD, S.wD); st_f3(st, pool, slot, W_PN, S.pn);
  WF_ST(W_PC, (uint32_t)S.pc); WF_ST(W_REFRB, (uint32_t)S.refr_bounce); WF_ST(W_GIN, (uint32_t)S.gi_n); WF_ST(W_STATE, (uint32_t)S.state);
  st_f3(st, pool, slot, W_RO, S.o); st_f3(st, pool, slot, W_RD, S.d); WF_ST(W_RBOUNCE, (uint32_t)S.bounce);
  // results of the rays about to be traced: "nothing hit / nothing occluded" until the trace kernel says otherwise
  WF_ST(W_TBEST, __float_as_uint(INFINITY)); WF_ST(W_REFBEST, REF_NONE);
  WF_ST(W_TPLANE, __float_as_uint(INFINITY)); WF_ST(W_PLANEID, 0xffffffffu);
  WF_ST(W_OCCL_LO, 0u); WF_ST(W_OCCL_HI, 0u);
}

enum : int { WM_FREE = 100 };

template <bool COUNT>
__global__ void __launch_bounds__(SBLOCK, MIRT_WF_SHADE_WAVES) wf_shade_kernel(const WfArgs w)
{
  __shared__ unsigned long long blk_ray[SBLOCK / 64 + 1], blk_done[SBLOCK / 64];
  const RenderArgs& a = w.r;
  const int slot = blockIdx.x * SBLOCK + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int pool = w.pool;
  uint32_t* st = w.state;
  const bool valid = slot < pool;
  const int nlights = a.num_suns + a.num_bulbs;
  Counters cn = {0, 0, 0, 0, 0, 0, 0, 0, 0};

  Lane S;
  S.g = -1;
  S.rng.v0 = S.rng.v1 = S.rng.v2 = S.rng.v3 = S.rng.v4 = S.rng.d = 0; S.rng.bm_flag = 0; S.rng.bm_extra = 0.0f;
  S.L = mk3(0, 0, 0); S.alpha = 0.0f;
  S.Hdir = mk3(0, 0, 0); S.Hp = mk3(0, 0, 0); S.Hn = mk3(0, 0, 0); S.Hcolor = mk3(0, 0, 0);
  S.Hbounce = 0; S.Hior = 1.458f; S.Hrough = 0.0f; S.HtransNZ = false;
  S.wt = mk3(1, 1, 1); S.wD = mk3(0, 0, 0); S.pn = mk3(0, 0, 0);
  S.pc = 0; S.refr_bounce = 0; S.gi_n = 0; S.state = ST_PRIMARY;
  S.bo = mk3(0, 0, 0); S.rdir = mk3(0, 0, 1); S.li = 0; S.occl = 0ull; S.batch_pending = false; S.has_reflect = false;
  S.o = mk3(0, 0, 0); S.d = mk3(0, 0, 1); S.inv = mk3(0, 0, 1); S.bounce = 0; S.limit = INFINITY; S.shadow = false;
  S.tplane = INFINITY; S.plane_id = -1; S.trav = false;
  S.cur = REF_NONE; S.tos = REF_NONE; S.sp = 0; S.tbest = INFINITY; S.refbest = REF_NONE;

  int micro = WM_FREE;
  if (valid) {
    const long long g = (long long)(((unsigned long long)WF_LD(W_G_HI) << 32) | WF_LD(W_G_LO));
    if (g >= 0) {
      load_slot(st, pool, slot, S);
      micro = advance_core<COUNT>(a, S, cn, slot, pool);
    }
  }
  unsigned long long done_local = 0;
  if (micro == M_DONE) {
    a.samples[S.g] = make_float4(S.L.x, S.L.y, S.L.z, S.alpha);
    S.g = -1;
    micro = WM_FREE;
    ++done_local;
  }
  // Free slots take the next samples of the frame: consecutive samples go to consecutive free slots of the block (their
  // primary rays stay coherent); one atomic per block.  Any assignment gives the same pixels: a sample's RNG stream
  // depends only on its own index.
  {
    const bool want_sample = valid && micro == WM_FREE;
    const unsigned long long need = __ballot(want_sample);
    const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
    const int wv0 = threadIdx.x >> 6;
    if (lane == 0) blk_ray[wv0] = (unsigned long long)__popcll(need);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long run = 0;
      for (int i = 0; i < SBLOCK / 64; ++i) { const unsigned long long t = blk_ray[i]; blk_ray[i] = run; run += t; }
      // racy pre-check of a counter that only grows: once the frame is handed out no more atomics are issued
      blk_ray[SBLOCK / 64] = (run && (long long)w.ctr[0] < a.num_samples) ? atomicAdd(&w.ctr[0], run) : (unsigned long long)a.num_samples;
    }
    __syncthreads();
    const long long idx = (long long)(blk_ray[SBLOCK / 64] + blk_ray[wv0]) + r;
    __syncthreads();
    if (want_sample && idx < a.num_samples) {
      init_sample_core<COUNT>(a, S, cn, idx);
      S.pc = 0;
      if (S.bounce == 0) {   // a bounce-0 ray never hits (draw.cu:294): the sample is finished at once
        a.samples[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        S.g = -1;
        ++done_local;
      } else micro = M_TRACE;
    }
  }
  // (a node without lights and without a reflection ray emits nothing; it is consumed in the next round)

  // ---- emit this slot's rays: wave-level exclusive scan of the ray counts, one atomic per wave -------------------
  int nr = 0;
  unsigned long long lit = 0ull;      // lights whose shadow ray is traced (see batch_next: unlit ones are answered at once)
  if (micro == M_TRACE) nr = 1;
  else if (micro == M_BATCH) {
    const uint32_t unlit = (uint32_t)(S.occl >> 32);
    for (int j = 0; j < nlights; ++j) {
      if (j < 32 && ((unlit >> j) & 1u)) { if (COUNT) { cn.rays++; cn.shadow_rays++; } }
      else { lit |= 1ull << j; ++nr; }
    }
    nr += S.has_reflect ? 1 : 0;
  }
  int incl = nr;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  const int total = __shfl(incl, 63);
  // block-level aggregation: one atomic on the queue counter per block
  const int wv = threadIdx.x >> 6;
  if (lane == 63) blk_ray[wv] = (unsigned long long)total;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int i = 0; i < SBLOCK / 64; ++i) { const unsigned long long t = blk_ray[i]; blk_ray[i] = run; run += t; }
    blk_ray[SBLOCK / 64] = run ? atomicAdd(&w.ctr[1], run) : 0ull;
  }
  __syncthreads();
  const unsigned long long qbase = blk_ray[SBLOCK / 64] + blk_ray[wv];
  const unsigned int first = (unsigned int)qbase + (unsigned int)(incl - nr);
  if (micro == M_TRACE) {
    if (first < w.max_rays) {
      w.rays[2 * (size_t)first + 0] = make_float4(S.o.x, S.o.y, S.o.z, INFINITY);
      w.rays[2 * (size_t)first + 1] = make_float4(S.d.x, S.d.y, S.d.z, __uint_as_float((uint32_t)slot | (63u << 24)));
    }
  } else if (micro == M_BATCH) {
    // shadow rays (draw.cu:346 / 362-363), then the reflection ray (draw.cu:402)
    unsigned int q = first;
    for (int j = 0; j < nlights; ++j) {
      if (!((lit >> j) & 1ull)) continue;
      f3 dir; float limit;
      if (j < a.num_suns) { const LightDev& lt = a.suns[j]; dir = mk3(lt.nx, lt.ny, lt.nz); limit = INFINITY; }
      else { const LightDev& lt = a.bulbs[j - a.num_suns]; const f3 bd = mk3(lt.x, lt.y, lt.z) - S.Hp; dir = normalize(bd); limit = length(bd); }
      if (q < w.max_rays) {
        w.rays[2 * (size_t)q + 0] = make_float4(S.bo.x, S.bo.y, S.bo.z, limit);
        w.rays[2 * (size_t)q + 1] = make_float4(dir.x, dir.y, dir.z, __uint_as_float((uint32_t)slot | ((uint32_t)j << 24) | (1u << 30)));
      }
      ++q;
    }
    if (S.has_reflect) {
      S.o = S.bo; S.d = S.rdir; S.bounce = S.Hbounce - 1;
      if (q < w.max_rays) {
        w.rays[2 * (size_t)q + 0] = make_float4(S.o.x, S.o.y, S.o.z, INFINITY);
        w.rays[2 * (size_t)q + 1] = make_float4(S.d.x, S.d.y, S.d.z, __uint_as_float((uint32_t)slot | (63u << 24)));
      }
    }
  }
  if (valid) store_slot(st, pool, slot, S);

  // ---- counters ---------------------------------------------------------------------------------------------------
  {
    unsigned long long x = done_local;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    if (lane == 0) blk_done[wv] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long t = 0;
      for (int i = 0; i < SBLOCK / 64; ++i) t += blk_done[i];
      if (t) atomicAdd(&w.ctr[3], t);
    }
  }
  if (COUNT && a.counters) {
    unsigned long long s0 = cn.samples, s6 = cn.mat_fetches, s1 = cn.rays, s2 = cn.shadow_rays;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off); s6 += __shfl_xor(s6, off); s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    if (lane == 0) { if (s0) atomicAdd(&a.counters[0], s0); if (s6) atomicAdd(&a.counters[6], s6); if (s1) atomicAdd(&a.counters[1], s1); if (s2) atomicAdd(&a.counters[2], s2); }
  }
}

template <bool COUNT>
__global__ void __launch_bounds__(WBLOCK, MIRT_WF_WAVES_PER_SIMD) wf_trace_kernel(const WfArgs w)
{
  __shared__ uint32_t lds_stack[WF_STACK_LDS * WBLOCK];
  const RenderArgs& a = w.r;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const long long gid = (long long)blockIdx.x * WBLOCK + tid;
  const long long gthreads = (long long)gridDim.x * WBLOCK;
  const int pool = w.pool;
  uint32_t* st = w.state;
  const unsigned int nrays = (unsigned int)w.ctr[1];
  Counters cn = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const float tmin = 0.0001f;

  // per-lane ray + traversal state
  f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), inv = mk3(0, 0, 1);
  float limit = INFINITY, tbest = INFINITY, tplane = INFINITY;
  uint32_t meta = 0, cur = REF_NONE, tos = REF_NONE, refbest = REF_NONE;
  int sp = 0, plane_id = -1;
  bool trav = false, have = false, shadow = false;
  const bool anyhit = a.shadow_anyhit != 0;      // (0: shadow rays are nearest-hit queries, draw.cu:347-352)
  // wave-local part of the queue
  unsigned int c_next = 0, c_end = 0;
  bool queue_empty = false;

  for (;;) {
    // ================= write results of finished rays and refill idle lanes =================
    const unsigned long long idle = __ballot(!trav);
    if (idle == ~0ull || (!queue_empty && __popcll(idle) >= a.refill_k) || (queue_empty && idle != 0ull && __ballot(!trav && have) != 0ull)) {
      if (!trav && have) {
        const int slot = (int)(meta & 0xffffffu);
        if (shadow) {
          const bool occluded = (plane_id >= 0 && tplane < limit) || (refbest != REF_NONE && tbest < limit);
          if (occluded) {
            const uint32_t j = (meta >> 24) & 63u;
            atomicOr(&st[(size_t)(j < 32 ? W_OCCL_LO : W_OCCL_HI) * pool + slot], 1u << (j & 31u));
          }
        } else {
          WF_ST(W_TBEST, __float_as_uint(tbest)); WF_ST(W_REFBEST, refbest);
          WF_ST(W_TPLANE, __float_as_uint(tplane)); WF_ST(W_PLANEID, (uint32_t)plane_id);
        }
        have = false;
      }
      if (!queue_empty) {
        const unsigned long long need = __ballot(!trav);
        int want = __popcll(need);
        const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
        int given = 0;               // rays handed out so far in this refill (wave-uniform)
        unsigned int my = 0xffffffffu;
        while (given < want) {
          if (c_next >= c_end) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&w.ctr[2], (unsigned long long)WF_CHUNK);
            base = __shfl(base, 0);
            if (base >= nrays) { queue_empty = true; break; }
            c_next = (unsigned int)base;
            c_end = (base + WF_CHUNK < nrays) ? (unsigned int)base + WF_CHUNK : nrays;
          }
          const int avail = (int)(c_end - c_next);
          const int take = (want - given < avail) ? want - given : avail;
          if (!trav && r >= given && r < given + take) my = c_next + (unsigned int)(r - given);
          c_next += (unsigned int)take;
          given += take;
        }
        if (my != 0xffffffffu) {
          const float4 r0 = w.rays[2 * (size_t)my + 0], r1 = w.rays[2 * (size_t)my + 1];
          o = mk3(r0.x, r0.y, r0.z); limit = r0.w;
          d = mk3(r1.x, r1.y, r1.z); meta = __float_as_uint(r1.w);
          shadow = (meta >> 30) & 1u;
          have = true;
          if (COUNT) { cn.rays++; if (shadow) cn.shadow_rays++; }
          inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
          // hitNearest's plane half (checkPlane, draw.cu:581-615)
          tplane = INFINITY; plane_id = -1;
          for (int i = 0; i < a.num_planes; ++i) {
            const PlaneDev& pl = a.planes[i];
            const f3 pnor = mk3(pl.nx, pl.ny, pl.nz);
            const float t = dot(mk3(pl.px, pl.py, pl.pz) - o, pnor) / dot(d, pnor);
            if (t <= 1e-6f) continue;
            if (t < tplane && t > EPSILON) { tplane = t; plane_id = i; }
          }
          if (tplane >= (float)(INT_MAX - 10)) { tplane = INFINITY; plane_id = -1; }
          tbest = INFINITY; refbest = REF_NONE; cur = a.root_ref; sp = 0;
          trav = (a.root_ref != REF_NONE) && !(shadow && anyhit && plane_id >= 0 && tplane < limit);
          if (COUNT && trav) cn.traversed++;
        }
      }
    }
    if (__ballot(trav) == 0) {
      if (queue_empty && __ballot(have) == 0) break;
      continue;
    }

    // ================= one traversal step: traverse_lbvh, bvh_traversal.cu:92-183 =================
    if (trav) {
      const bool leaf = (cur & REF_LEAF) != 0;
      const bool tri = leaf && (cur & REF_TRI);
      const uint32_t roff = cur << 4;
      const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(a.nodes) + roff);
      const float4 q0 = rec[0];
      float4 q1 = make_float4(0, 0, 0, 0), q2 = q1, q3 = q1;
      if (!leaf || tri) { q1 = rec[1]; q2 = rec[2]; }
      if (!leaf) q3 = rec[3];
      bool pop = false;
      if (leaf) {
        // intersect_leaf_primitives, bvh_traversal.cu:47-89
        const uint32_t off16 = cur & REF_OFFMASK;
        if (tri) {
          if (COUNT) cn.tri_tests++;
          float t;
          const bool hit = triangle_hit(q0, q1, q2, o, d, t);
          if (closer_hit(hit, t, tbest, off16, refbest)) {
            tbest = t; refbest = cur;
            if (shadow && anyhit && tbest < limit) trav = false;      // any-hit exit (same boolean as draw.cu:347-352 / 365-370)
          }
        } else {
          if (COUNT) cn.sphere_tests++;
          float t;
          float tc_, tf_;
          const bool hit = sphere_hit(q0, o, d, t, tc_, tf_);
          if (closer_hit(hit, t, tbest, off16, refbest)) {
            tbest = t; refbest = cur;
            if (shadow && anyhit && tbest < limit) trav = false;
          }
        }
        pop = trav;
      } else {
        if (COUNT) cn.internal_visits++;
        // hit_aabb_adapted, bvh_traversal.cu:11-44, on both children
        bool hl, hr;
        float tel, ter;
        box_pair(q0, q1, q2, o.x, o.y, o.z, inv.x, inv.y, inv.z, tbest, tmin, hl, hr, tel, ter);
        uint32_t lref = __float_as_uint(q3.x), rref = __float_as_uint(q3.y);
        order_children(hl, hr, tel, ter, __float_as_uint(q3.z), a.swap_mask, lref, rref);
        if (hl && hr) {
          cur = lref;
          {   // (the stack cannot outgrow STACK_TOTAL: see render.hip)
            if (sp > 0) {
              const int s2 = sp - 1;
              if (s2 < WF_STACK_LDS) lds_stack[s2 * WBLOCK + tid] = tos;
              else a.stack_spill[(size_t)(s2 - WF_STACK_LDS) * gthreads + gid] = tos;
            }
            tos = rref;
            ++sp;
            if (COUNT) cn.max_stack = max(cn.max_stack, (uint32_t)sp);
          }
        } else if (hl) cur = lref;
        else if (hr) cur = rref;
        else pop = true;
      }
      if (pop) {
        if (sp == 0) trav = false;
        else {
          cur = tos;
          --sp;
          if (sp > 0) {
            const int s2 = sp - 1;
            tos = lds_stack[(s2 < WF_STACK_LDS ? s2 : 0) * WBLOCK + tid];
            if (s2 >= WF_STACK_LDS) tos = a.stack_spill[(size_t)(s2 - WF_STACK_LDS) * gthreads + gid];
          }
        }
      }
    }
  }

  if (COUNT && a.counters) {
    uint32_t v[9] = {0, cn.rays, cn.shadow_rays, cn.internal_visits, cn.sphere_tests, cn.tri_tests, 0, cn.max_stack, cn.traversed};
#pragma unroll
    for (int k = 1; k < 9; ++k) {
      if (k == 6) continue;
      unsigned long long x = v[k];
      if (k == 7) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { unsigned long long y = __shfl_xor(x, off); x = x > y ? x : y; }
        if (lane == 0) atomicMax(&a.counters[k], x);
      } else {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (lane == 0 && x) atomicAdd(&a.counters[k == 8 ? 11 : k], x);
      }
    }
  }
}

__global__ void wf_reset_kernel(uint32_t* state, int pool)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < pool) {
    state[(size_t)W_G_LO * pool + i] = 0xffffffffu; state[(size_t)W_G_HI * pool + i] = 0xffffffffu;
    state[(size_t)W_NEXT_LO * pool + i] = (uint32_t)i; state[(size_t)W_NEXT_HI * pool + i] = 0u;
  }
}

} // namespace

// Host side: runs shade / trace rounds until the frame's samples are all finished.
// Returns the summed trace-kernel time (ms, HIP events) in *trace_ms.
int wavefront_trace(MirtScene* sc, RenderCtx& cx, RenderArgs& a, bool count, hipStream_t stream, float* trace_ms)
{
  const long long nsamples = a.num_samples;
  int pool = sc->opt.wf_pool;
  if ((long long)pool > nsamples) pool = (int)((nsamples + SBLOCK - 1) / SBLOCK * SBLOCK);
  pool = (pool + SBLOCK - 1) / SBLOCK * SBLOCK;
  const int nlights = a.num_suns + a.num_bulbs;
  const size_t max_rays = (size_t)pool * (size_t)(nlights + 1);
  if (max_rays >= 0xffffffffull) { set_error("wavefront: ray queue too large"); return MIRT_ERR_ARG; }

  // workspace
  const size_t state_bytes = sizeof(uint32_t) * (size_t)W_COUNT * pool;
  const size_t rays_bytes = sizeof(float4) * 2 * max_rays;
  if (sc->wf_state_cap < state_bytes) {
    MIRT_HIP(hipStreamSynchronize(stream));
    hipFree(sc->wf_state); sc->wf_state = nullptr; sc->wf_state_cap = 0;
    MIRT_HIP(hipMalloc(&sc->wf_state, state_bytes));
    sc->wf_state_cap = state_bytes;
  }
  if (sc->wf_rays_cap < rays_bytes) {
    MIRT_HIP(hipStreamSynchronize(stream));
    hipFree(sc->wf_rays); sc->wf_rays = nullptr; sc->wf_rays_cap = 0;
    MIRT_HIP(hipMalloc(&sc->wf_rays, rays_bytes));
    sc->wf_rays_cap = rays_bytes;
  }
  if (!sc->wf_ctr) {
    MIRT_HIP(hipMalloc(&sc->wf_ctr, 8 * sizeof(unsigned long long)));
    MIRT_HIP(hipHostMalloc(&sc->wf_ctr_host, 8 * sizeof(unsigned long long)));
  }
  // pending-children LIFO is indexed by slot here
  const bool need_pending = sc->any_trans || sc->d.gi != 0;
  const int pending_slots = need_pending ? 2 * (sc->d.bounces + (sc->d.gi > 0 ? sc->d.gi : 0) + 2) : 0;
  const size_t pending_need = (size_t)pending_slots * PENDING_WORDS * pool;
  if (cx.pending_cap < pending_need) {
    MIRT_HIP(hipStreamSynchronize(stream));
    hipFree(cx.pending); cx.pending = nullptr; cx.pending_cap = 0;
    MIRT_HIP(hipMalloc(&cx.pending, sizeof(float) * pending_need));
    cx.pending_cap = pending_need;
  }
  a.pending = cx.pending; a.pending_slots = pending_slots;

  if (!sc->wf_trace_blocks) {
    hipDeviceProp_t prop;
    MIRT_HIP(hipGetDeviceProperties(&prop, sc->device));
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wf_trace_kernel<false>, WBLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    sc->wf_trace_blocks = prop.multiProcessorCount * per_cu;
  }
  const int trace_blocks = sc->wf_trace_blocks;
  const size_t gthreads = (size_t)trace_blocks * WBLOCK;
  const size_t spill_need = (size_t)STACK_TOTAL * gthreads;
  if (cx.spill_cap < spill_need) {
    MIRT_HIP(hipStreamSynchronize(stream));
    hipFree(cx.stack_spill); cx.stack_spill = nullptr; cx.spill_cap = 0;
    MIRT_HIP(hipMalloc(&cx.stack_spill, sizeof(uint32_t) * spill_need));
    cx.spill_cap = spill_need;
  }
  a.stack_spill = cx.stack_spill;

  WfArgs w;
  w.r = a; w.pool = pool; w.state = sc->wf_state; w.rays = sc->wf_rays; w.ctr = sc->wf_ctr; w.max_rays = (unsigned int)max_rays;

  MIRT_HIP(hipMemsetAsync(sc->wf_ctr, 0, 8 * sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(wf_reset_kernel, dim3((pool + 255) / 256), dim3(256), 0, stream, sc->wf_state, pool);

  const int shade_blocks = pool / SBLOCK;
  std::vector<hipEvent_t>& evs = sc->wf_events;
  size_t ev_used = 0;
  auto next_event = [&](hipEvent_t* out) -> int {
    if (ev_used == evs.size()) { hipEvent_t e; MIRT_HIP(hipEventCreate(&e)); evs.push_back(e); }
    *out = evs[ev_used++];
    return MIRT_OK;
  };
  const int check_every = 4;
  int rounds = 0;
  for (;;) {
    for (int k = 0; k < check_every; ++k, ++rounds) {
      MIRT_HIP(hipMemsetAsync(sc->wf_ctr + 1, 0, 2 * sizeof(unsigned long long), stream));   // rays emitted, queue head
      if (count) hipLaunchKernelGGL(wf_shade_kernel<true>, dim3(shade_blocks), dim3(SBLOCK), 0, stream, w);
      else hipLaunchKernelGGL(wf_shade_kernel<false>, dim3(shade_blocks), dim3(SBLOCK), 0, stream, w);
      hipEvent_t e0, e1;
      int rc = next_event(&e0); if (rc) return rc;
      rc = next_event(&e1); if (rc) return rc;
      MIRT_HIP(hipEventRecord(e0, stream));
      if (count) hipLaunchKernelGGL(wf_trace_kernel<true>, dim3(trace_blocks), dim3(WBLOCK), 0, stream, w);
      else hipLaunchKernelGGL(wf_trace_kernel<false>, dim3(trace_blocks), dim3(WBLOCK), 0, stream, w);
      MIRT_HIP(hipEventRecord(e1, stream));
    }
    MIRT_HIP(hipGetLastError());
    MIRT_HIP(hipMemcpyAsync(sc->wf_ctr_host, sc->wf_ctr, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    MIRT_HIP(hipStreamSynchronize(stream));
    if ((long long)sc->wf_ctr_host[3] >= nsamples && sc->wf_ctr_host[1] == 0) break;
    if (rounds > 100000) { set_error("wavefront: did not converge"); return MIRT_ERR_STATE; }
  }
  float total = 0.0f;
  for (size_t i = 0; i + 1 < ev_used; i += 2) {
    float ms = 0.0f;
    MIRT_HIP(hipEventElapsedTime(&ms, evs[i], evs[i + 1]));
    total += ms;
  }
  if (trace_ms) *trace_ms = total;
  sc->wf_rounds = rounds;
  return MIRT_OK;
}

} // namespace mirt
