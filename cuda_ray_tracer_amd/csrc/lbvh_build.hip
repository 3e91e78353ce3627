// LBVH build on the device: replaces build_lbvh_karas (lbvh_builder.cu:401-521) and
// build_morton_codes_and_sort_primitives (lbvh_utils.cu:77-129).
//
//   prim_bounds_kernel   scene bounds = union of primitive boxes (what parse.cpp:147-155,193-200 computes on the
//                        host and then drops -- reference bug #1; here the bounds are real)
//   morton_kernel        30-bit codes (lbvh_utils.cu:10-75)
//   radix_pass_kernel    stable LSD radix sort, 4 passes x 8 bits, wave64 ballot ranking (replaces thrust::sort_by_key)
//   karras_kernel        Karras 2012 hierarchy with the reference's index tie-break (lbvh_builder.cu:76-322)
//   tri_scan_*           exclusive scan of the triangle flags in sorted order: where each sorted leaf's record goes in the
//                        heap, and which subtrees hold spheres only
//   refit_pack_kernel    bottom-up boxes (lbvh_builder.cu:324-387, with the missing release/acquire added) and
//                        64-byte two-child node records for the traversal kernel
//   scatter_prims_kernel primitive records into the heap in sorted order (neighbours in space are neighbours in memory)
#include "scene_dev.h"
#include "host_scene.h"

#include <cstdio>
#include <cstring>

namespace mirt {

int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
  char buf[512];
  snprintf(buf, sizeof(buf), "HIP error in %s at line %d: %s (%s)", file, line, hipGetErrorString(e), what);
  set_error(buf);
  return MIRT_ERR_HIP;
}

namespace {

constexpr int BLOCK = 256;

struct Box { float xmin, xmax, ymin, ymax, zmin, zmax; };

// get_primitive_aabb_device, lbvh_builder.cu:33-57 with AABB ctors interval.cuh:55-81
MIRT_DEV Box prim_box(uint32_t type, uint32_t id, const float4* __restrict__ spheres, const float4* __restrict__ tri_verts)
{
  Box b;
  if (type == 0) {
    const float4 s = spheres[id];
    const float ax = s.x - s.w, bx = s.x + s.w;
    const float ay = s.y - s.w, by = s.y + s.w;
    const float az = s.z - s.w, bz = s.z + s.w;
    if (ax <= bx) { b.xmin = ax; b.xmax = bx; } else { b.xmin = bx; b.xmax = ax; }
    if (ay <= by) { b.ymin = ay; b.ymax = by; } else { b.ymin = by; b.ymax = ay; }
    if (az <= bz) { b.zmin = az; b.zmax = bz; } else { b.zmin = bz; b.zmax = az; }
  } else {
    const float4 p0 = tri_verts[3 * (size_t)id + 0], p1 = tri_verts[3 * (size_t)id + 1], p2 = tri_verts[3 * (size_t)id + 2];
    b.xmin = fminf(fminf(p0.x, p1.x), p2.x); b.xmax = fmaxf(fmaxf(p0.x, p1.x), p2.x);
    b.ymin = fminf(fminf(p0.y, p1.y), p2.y); b.ymax = fmaxf(fmaxf(p0.y, p1.y), p2.y);
    b.zmin = fminf(fminf(p0.z, p1.z), p2.z); b.zmax = fmaxf(fmaxf(p0.z, p1.z), p2.z);
    if (b.xmax - b.xmin < 0.01f) { b.xmin = b.xmin - 0.01f; b.xmax = b.xmax + 0.01f; }
    if (b.ymax - b.ymin < 0.01f) { b.ymin = b.ymin - 0.01f; b.ymax = b.ymax + 0.01f; }
    if (b.zmax - b.zmin < 0.01f) { b.zmin = b.zmin - 0.01f; b.zmax = b.zmax + 0.01f; }
  }
  return b;
}

// order-preserving float <-> uint map so that atomicMin/atomicMax on uints realise float min/max
MIRT_DEV uint32_t f2key(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
MIRT_DEV float key2f(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ void __launch_bounds__(BLOCK) prim_bounds_kernel(const MirtPrimRef* __restrict__ refs, const float4* __restrict__ spheres,
                                                            const float4* __restrict__ tri_verts, int n, uint32_t* __restrict__ keys)
{
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
    const MirtPrimRef r = refs[i];
    const Box b = prim_box(r.type, r.id, spheres, tri_verts);
    mn[0] = fminf(mn[0], b.xmin); mx[0] = fmaxf(mx[0], b.xmax);
    mn[1] = fminf(mn[1], b.ymin); mx[1] = fmaxf(mx[1], b.ymax);
    mn[2] = fminf(mn[2], b.zmin); mx[2] = fmaxf(mx[2], b.zmax);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      mn[k] = fminf(mn[k], __shfl_xor(mn[k], off));
      mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off));
    }
  // one set of atomics per block (the six words are hot: every block of the grid hits them)
  __shared__ float red[BLOCK / 64][6];
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { red[wv][k] = mn[k]; red[wv][3 + k] = mx[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (int i = 1; i < BLOCK / 64; ++i) v = (threadIdx.x < 3) ? fminf(v, red[i][threadIdx.x]) : fmaxf(v, red[i][threadIdx.x]);
    if (threadIdx.x < 3) atomicMin(&keys[threadIdx.x], f2key(v));
    else atomicMax(&keys[threadIdx.x], f2key(v));
  }
}

// lbvh_utils.cu:10-30
MIRT_DEV uint32_t expand_bits(uint32_t v)
{
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
MIRT_DEV uint32_t quantize_coordinate(float coord, float smin, float range)
{
  if (range <= 1e-6f) return 0;
  float normalized = (coord - smin) / range;
  normalized = fmaxf(0.0f, fminf(1.0f, normalized));
  return (uint32_t)(normalized * 1023);
}

// generate_morton_codes_kernel, lbvh_utils.cu:32-75
__global__ void __launch_bounds__(BLOCK) morton_kernel(const MirtPrimRef* __restrict__ refs, const float4* __restrict__ spheres,
                                                       const float4* __restrict__ tri_verts, int n, const uint32_t* __restrict__ bkeys,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const float mnx = key2f(bkeys[0]), mny = key2f(bkeys[1]), mnz = key2f(bkeys[2]);
  const float mxx = key2f(bkeys[3]), mxy = key2f(bkeys[4]), mxz = key2f(bkeys[5]);
  const MirtPrimRef r = refs[i];
  float cx, cy, cz;
  if (r.type == 0) {
    const float4 s = spheres[r.id];
    cx = s.x; cy = s.y; cz = s.z;
  } else {
    const float4 p0 = tri_verts[3 * (size_t)r.id + 0], p1 = tri_verts[3 * (size_t)r.id + 1], p2 = tri_verts[3 * (size_t)r.id + 2];
    cx = ((p0.x + p1.x) + p2.x) / 3.0f;
    cy = ((p0.y + p1.y) + p2.y) / 3.0f;
    cz = ((p0.z + p1.z) + p2.z) / 3.0f;
  }
  const uint32_t qx = quantize_coordinate(cx, mnx, mxx - mnx);
  const uint32_t qy = quantize_coordinate(cy, mny, mxy - mny);
  const uint32_t qz = quantize_coordinate(cz, mnz, mxz - mnz);
  keys[i] = expand_bits(qx) | (expand_bits(qy) << 1) | (expand_bits(qz) << 2);
  vals[i] = (uint32_t)i;
}

// ---- stable LSD radix sort -------------------------------------------------------------------------
constexpr int SORT_ITEMS = 8;
constexpr int SORT_TILE = BLOCK * SORT_ITEMS;   // 2048 keys per workgroup; each wave owns 512 consecutive keys

template <bool SCATTER>
__global__ void __launch_bounds__(BLOCK) radix_pass_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                           uint32_t* __restrict__ hist, const uint32_t* __restrict__ offsets,
                                                           const uint32_t* __restrict__ totals, int n, int shift, int nblocks)
{
  __shared__ uint32_t wcnt[4][256];
  __shared__ uint32_t dbase[256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 4 * 256; i += BLOCK) (&wcnt[0][0])[i] = 0;
  __syncthreads();

  const int base = blockIdx.x * SORT_TILE + wave * (SORT_ITEMS * 64);
  uint32_t key[SORT_ITEMS], rank[SORT_ITEMS];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    const int idx = base + it * 64 + lane;
    const bool valid = idx < n;
    key[it] = valid ? keys_in[idx] : 0xffffffffu;
    const uint32_t digit = (key[it] >> shift) & 255u;
    unsigned long long mask = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (digit >> b) & 1u;
      const unsigned long long bal = __ballot(valid && bit);
      mask &= bit ? bal : ~bal;
    }
    const uint32_t r = (uint32_t)__popcll(mask & lt);
    const uint32_t c = (uint32_t)__popcll(mask);
    uint32_t prev = 0;
    if (valid) prev = wcnt[wave][digit];
    rank[it] = prev + r;
    __builtin_amdgcn_wave_barrier();
    if (valid && r == 0) wcnt[wave][digit] = prev + c;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  const uint32_t c0 = wcnt[0][tid], c1 = wcnt[1][tid], c2 = wcnt[2][tid], c3 = wcnt[3][tid];
  if (!SCATTER) {
    hist[(size_t)tid * nblocks + blockIdx.x] = c0 + c1 + c2 + c3;
    return;
  }
  // exclusive scan of the 256 digit totals (every block repeats it: 256 values)
  const uint32_t mytot = totals[tid];
  dbase[tid] = mytot;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t v = (tid >= o) ? dbase[tid - o] : 0;
    __syncthreads();
    dbase[tid] += v;
    __syncthreads();
  }
  const uint32_t off = (dbase[tid] - mytot) + offsets[(size_t)tid * nblocks + blockIdx.x];
  __syncthreads();
  wcnt[0][tid] = off; wcnt[1][tid] = off + c0; wcnt[2][tid] = off + c0 + c1; wcnt[3][tid] = off + c0 + c1 + c2;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    const int idx = base + it * 64 + lane;
    if (idx < n) {
      const uint32_t digit = (key[it] >> shift) & 255u;
      const uint32_t pos = wcnt[wave][digit] + rank[it];
      keys_out[pos] = key[it];
      vals_out[pos] = vals_in ? vals_in[idx] : (uint32_t)idx;      // (no values: the position itself)
    }
  }
}

// hist is [256 digits][nblocks].  Block d scans row d in place (exclusive) and writes the row total to totals[d]; the
// scatter kernel adds the exclusive scan of the 256 totals (done per block in LDS).
__global__ void __launch_bounds__(BLOCK) row_scan_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ totals, int nblocks)
{
  __shared__ uint32_t sums[BLOCK];
  const int tid = threadIdx.x;
  uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
  const int chunk = (nblocks + BLOCK - 1) / BLOCK;
  const int lo = tid * chunk, hi = min(lo + chunk, nblocks);
  uint32_t s = 0;
  for (int i = lo; i < hi; ++i) s += row[i];
  sums[tid] = s;
  __syncthreads();
  for (int off = 1; off < BLOCK; off <<= 1) {
    const uint32_t v = (tid >= off) ? sums[tid - off] : 0;
    __syncthreads();
    sums[tid] += v;
    __syncthreads();
  }
  uint32_t run = sums[tid] - s;
  for (int i = lo; i < hi; ++i) { const uint32_t v = row[i]; row[i] = run; run += v; }
  if (tid == BLOCK - 1) totals[blockIdx.x] = sums[tid];
}

// ---- Karras hierarchy ------------------------------------------------------------------------------
MIRT_DEV int clz32(uint32_t x) { return x ? __clz((int)x) : 32; }
// adapted_delta, lbvh_builder.cu:76-101
MIRT_DEV int delta(int a, int b, int n, const uint32_t* __restrict__ codes)
{
  if (a < 0 || a >= n || b < 0 || b >= n) return -1;
  const uint32_t ka = codes[a], kb = codes[b];
  if (ka == kb) return 32 + clz32((uint32_t)a ^ (uint32_t)b);
  return clz32(ka ^ kb);
}

// generate_internal_nodes_karas_kernel, lbvh_builder.cu:224-322 (+ determine_range_adapted :103-182,
// find_split_adapted :186-221).  Children use the reference numbering: internal i in [0,N-2], leaf j -> N-1+j.
__global__ void __launch_bounds__(BLOCK) karras_kernel(const uint32_t* __restrict__ codes, int n, uint32_t* __restrict__ child_l,
                                                       uint32_t* __restrict__ child_r, int* __restrict__ parent, uint2* __restrict__ range)
{
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n - 1) return;
  // range
  const int dl = delta(i, i - 1, n, codes), dr = delta(i, i + 1, n, codes);
  int d, dmin;
  if (dr > dl) { d = 1; dmin = dl; } else { d = -1; dmin = dr; }
  uint32_t lmax = 1;
  int nb = (int)((uint32_t)i + lmax * (uint32_t)d);
  int cur = delta(i, nb, n, codes);
  while (cur > dmin) {
    lmax <<= 1;
    nb = (int)((uint32_t)i + lmax * (uint32_t)d);
    if (nb < 0 || nb >= n) break;
    cur = delta(i, nb, n, codes);
  }
  uint32_t l = 0;
  for (uint32_t t = lmax >> 1; t > 0; t >>= 1) {
    const int nbb = (int)((uint32_t)i + (l + t) * (uint32_t)d);
    if (nbb >= 0 && nbb < n) {
      if (delta(i, nbb, n, codes) > dmin) l += t;
    }
  }
  const int j = (int)((uint32_t)i + l * (uint32_t)d);
  const int first = (i < j) ? i : j, last = (i < j) ? j : i;
  range[i] = make_uint2((uint32_t)first, (uint32_t)last);
  // split
  int split = first;
  if (first != last) {
    const int common = delta(first, last, n, codes);
    int step = last - first;
    do {
      step = (step + 1) >> 1;
      const int cand = split + step;
      if (cand < last) {
        if (delta(first, cand, n, codes) > common) split = cand;
      }
    } while (step > 1);
  }
  if (split < first || split >= last) {   // cannot happen for a valid range; mirror the reference's marker
    child_l[i] = 0xffffffffu; child_r[i] = 0xffffffffu;
    return;
  }
  const uint32_t leaf_base = (uint32_t)(n - 1);
  const int d_split = delta(split, split + 1, n, codes);
  uint32_t lc, rc;
  if (split == first) lc = leaf_base + (uint32_t)split;
  else lc = (delta(first, split, n, codes) > d_split) ? (uint32_t)split : leaf_base + (uint32_t)first;
  if (split + 1 == last) rc = leaf_base + (uint32_t)last;
  else rc = (delta(split + 1, last, n, codes) > d_split) ? (uint32_t)(split + 1) : leaf_base + (uint32_t)last;
  child_l[i] = lc; child_r[i] = rc;
  parent[lc] = i; parent[rc] = i;
  if (i == 0) parent[0] = -1;
}

// ---- where the primitive records go: exclusive scan of the triangle flags over the sorted leaves ------------------------
// Sorted leaf j's record starts at unit j + 2 * tris_before[j] of the primitive region (a sphere is one 16-byte unit, a
// triangle three), and leaves [f, l] hold spheres only iff tris_before[l + 1] == tris_before[f].
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;

MIRT_DEV uint32_t block_exclusive_scan(uint32_t v, uint32_t* sh /* [BLOCK] */, uint32_t* total)
{
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int o = 1; o < BLOCK; o <<= 1) {
    const uint32_t x = (tid >= o) ? sh[tid - o] : 0;
    __syncthreads();
    sh[tid] += x;
    __syncthreads();
  }
  const uint32_t incl = sh[tid];
  if (total) *total = sh[BLOCK - 1];
  __syncthreads();
  return incl - v;
}

template <bool WRITE>
__global__ void __launch_bounds__(BLOCK) tri_scan_kernel(int n, const uint32_t* __restrict__ order, const MirtPrimRef* __restrict__ refs,
                                                         uint32_t* __restrict__ block_sums, uint32_t* __restrict__ tris_before)
{
  __shared__ uint32_t sh[BLOCK];
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t f[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const int j = base + k;
    f[k] = (j < n) ? (refs[order[j]].type != 0 ? 1u : 0u) : 0u;
    s += f[k];
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan(s, sh, &total);
  if (!WRITE) { if (threadIdx.x == 0) block_sums[blockIdx.x] = total; return; }
  run += block_sums[blockIdx.x];                // exclusive prefix of the blocks before this one
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const int j = base + k;
    if (j < n) tris_before[j] = run;
    run += f[k];
    if (j == n - 1) tris_before[n] = run;
  }
}
// one block: block_sums -> exclusive prefix, in place
__global__ void __launch_bounds__(BLOCK) scan_sums_kernel(uint32_t* __restrict__ block_sums, int nb)
{
  __shared__ uint32_t sh[BLOCK];
  uint32_t carry = 0;
  for (int b0 = 0; b0 < nb; b0 += BLOCK) {
    const int i = b0 + threadIdx.x;
    const uint32_t v = (i < nb) ? block_sums[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(v, sh, &total);
    if (i < nb) block_sums[i] = carry + ex;
    carry += total;
  }
}

// reference to child `node` (reference numbering: internal [0, N-2], leaf j -> N-1+j) as the traversal kernels want it
MIRT_DEV uint32_t make_ref(uint32_t node, uint32_t leaf_base, const uint32_t* __restrict__ order, const MirtPrimRef* __restrict__ refs,
                           const uint32_t* __restrict__ tris_before, const uint2* __restrict__ range, uint32_t prim_base16, bool* pure)
{
  if (node >= leaf_base) {
    const uint32_t j = node - leaf_base;
    const bool tri = refs[order[j]].type != 0;
    *pure = !tri;
    return REF_LEAF | (tri ? REF_TRI : 0u) | (prim_base16 + j + 2u * tris_before[j]);
  }
  const uint2 fl = range[node];
  *pure = tris_before[fl.y + 1] == tris_before[fl.x];
  return 4u * node;
}

// Quantised node records (scene_dev.h): child boxes on the scene-bounds grid, rounded outwards.
MIRT_DEV uint32_t qlo(float x, float smin, float step) { const float q = floorf((x - smin) / step) - 1.0f; return (uint32_t)fminf(fmaxf(q, 0.0f), QGRID); }
MIRT_DEV uint32_t qhi(float x, float smin, float step) { const float q = ceilf((x - smin) / step) + 1.0f; return (uint32_t)fminf(fmaxf(q, 0.0f), QGRID); }
__global__ void __launch_bounds__(BLOCK) pack_qnodes_kernel(int n, const uint32_t* __restrict__ bkeys, const uint32_t* __restrict__ child_l,
                                                            const uint32_t* __restrict__ child_r, const float* __restrict__ boxes,
                                                            const uint32_t* __restrict__ order, const MirtPrimRef* __restrict__ refs,
                                                            const uint32_t* __restrict__ tris_before, const uint2* __restrict__ range,
                                                            uint4* __restrict__ qnodes, float* __restrict__ qparams, uint32_t qbase16, uint32_t prim_base16)
{
  const int p = blockIdx.x * BLOCK + threadIdx.x;
  float smin[3], step[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    smin[k] = key2f(bkeys[k]);
    const float range = key2f(bkeys[3 + k]) - smin[k];
    step[k] = range > 0.0f ? range / QGRID : 1.0f;
  }
  if (p == 0) { for (int k = 0; k < 3; ++k) { qparams[k] = smin[k]; qparams[3 + k] = step[k]; qparams[6 + k] = QINV_STEPS / step[k]; } }
  if (p >= n - 1) return;
  const uint32_t leaf_base = (uint32_t)(n - 1);
  const uint32_t c[2] = {child_l[p], child_r[p]};
  uint32_t w[8];
  bool pure[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float* b = boxes + 6 * (size_t)c[s];
#pragma unroll
    for (int k = 0; k < 3; ++k) w[3 * s + k] = qlo(b[2 * k], smin[k], step[k]) | (qhi(b[2 * k + 1], smin[k], step[k]) << 16);
    const uint32_t r = make_ref(c[s], leaf_base, order, refs, tris_before, range, prim_base16, &pure[s]);
    w[6 + s] = c[s] >= leaf_base ? r : (qbase16 + 2u * c[s]);
  }
  if (pure[0] && pure[1]) w[6] |= REF_QPURE;
  qnodes[2 * (size_t)p + 0] = make_uint4(w[0], w[1], w[2], w[3]);
  qnodes[2 * (size_t)p + 1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// Wide quantised records (scene_dev.h): the grandchildren of every internal node, in the reference's visiting order.
__global__ void __launch_bounds__(BLOCK) pack_wnodes_kernel(int n, const uint32_t* __restrict__ bkeys, const uint32_t* __restrict__ child_l,
                                                            const uint32_t* __restrict__ child_r, const float* __restrict__ boxes,
                                                            const uint32_t* __restrict__ order, const MirtPrimRef* __restrict__ refs,
                                                            const uint32_t* __restrict__ tris_before, const uint2* __restrict__ range,
                                                            uint4* __restrict__ wnodes, float* __restrict__ qparams, uint32_t wbase16, uint32_t prim_base16)
{
  const int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n - 1) return;
  float smin[3], step[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    smin[k] = key2f(bkeys[k]);
    const float range = key2f(bkeys[3 + k]) - smin[k];
    step[k] = range > 0.0f ? range / QGRID : 1.0f;
  }
  if (p == 0) { for (int k = 0; k < 3; ++k) { qparams[k] = smin[k]; qparams[3 + k] = step[k]; qparams[6 + k] = QINV_STEPS / step[k]; } }
  const uint32_t leaf_base = (uint32_t)(n - 1);
  uint32_t kids[4];
  int nk = 0;
  const uint32_t two[2] = {child_l[p], child_r[p]};
  for (int s = 0; s < 2; ++s) {
    if (two[s] >= leaf_base) kids[nk++] = two[s];
    else { kids[nk++] = child_l[two[s]]; kids[nk++] = child_r[two[s]]; }
  }
  uint32_t w[16];
  for (int i = 0; i < 4; ++i) {
    if (i < nk) {
      const float* b = boxes + 6 * (size_t)kids[i];
      for (int k = 0; k < 3; ++k) w[3 * i + k] = qlo(b[2 * k], smin[k], step[k]) | (qhi(b[2 * k + 1], smin[k], step[k]) << 16);
      bool pure;
      const uint32_t r = make_ref(kids[i], leaf_base, order, refs, tris_before, range, prim_base16, &pure);
      w[12 + i] = kids[i] >= leaf_base ? r : (wbase16 + 4u * kids[i]);
    } else {
      w[3 * i + 0] = QBOX_NONE; w[3 * i + 1] = QBOX_NONE; w[3 * i + 2] = QBOX_NONE;
      w[12 + i] = REF_NONE;
    }
  }
  for (int i = 0; i < 4; ++i) wnodes[4 * (size_t)p + i] = make_uint4(w[4 * i + 0], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// primitive records in sorted order + the unit -> primitive map the shading code uses to find the material
__global__ void __launch_bounds__(BLOCK) scatter_prims_kernel(int n, const uint32_t* __restrict__ order, const MirtPrimRef* __restrict__ refs,
                                                              const uint32_t* __restrict__ tris_before, const float4* __restrict__ spheres,
                                                              const float4* __restrict__ tris, float4* __restrict__ prim_region,
                                                              uint32_t* __restrict__ unit_prim)
{
  const int j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= n) return;
  const MirtPrimRef r = refs[order[j]];
  const size_t unit = (size_t)j + 2 * (size_t)tris_before[j];
  if (r.type == 0) {
    prim_region[unit] = spheres[r.id];
    unit_prim[unit] = r.id;
  } else {
    prim_region[unit + 0] = tris[3 * (size_t)r.id + 0];
    prim_region[unit + 1] = tris[3 * (size_t)r.id + 1];
    prim_region[unit + 2] = tris[3 * (size_t)r.id + 2];
    unit_prim[unit + 0] = 0x80000000u | r.id; unit_prim[unit + 1] = 0x80000000u | r.id; unit_prim[unit + 2] = 0x80000000u | r.id;
  }
}

// set_aabb_kernel_adapted, lbvh_builder.cu:324-387.  One thread per leaf; the second thread to arrive at a parent
// merges the children.  The box stores are published before the arrival counter is bumped, see below (the reference
// has no fence there).  The merging thread also writes the parent's packed traversal record.
__global__ void __launch_bounds__(BLOCK) refit_pack_kernel(int n, const uint32_t* __restrict__ order, const MirtPrimRef* __restrict__ refs,
                                                           const float4* __restrict__ spheres, const float4* __restrict__ tri_verts,
                                                           const uint32_t* __restrict__ child_l, const uint32_t* __restrict__ child_r,
                                                           const int* __restrict__ parent, uint32_t* __restrict__ arrived,
                                                           float* boxes, float4* __restrict__ nodes, const uint32_t* __restrict__ tris_before,
                                                           const uint2* __restrict__ range, uint32_t prim_base16, float4* __restrict__ tri_boxes)
{
  const int j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= n) return;
  const uint32_t leaf_base = (uint32_t)(n - 1);
  uint32_t cur = leaf_base + (uint32_t)j;
  const MirtPrimRef r = refs[order[j]];
  const Box b = prim_box(r.type, r.id, spheres, tri_verts);
  if (r.type != 0) {      // the triangle's exact leaf box, by scene index (the quantised walk's triangle check, render.hip)
    tri_boxes[2 * (size_t)r.id + 0] = make_float4(b.xmin, b.xmax, b.ymin, b.ymax);
    tri_boxes[2 * (size_t)r.id + 1] = make_float4(b.zmin, b.zmax, 0.0f, 0.0f);
  }
  // Hand-off protocol (MI355X guide, "sc1 stores and loads on both sides"): every box word is written with a write-through
  // (agent-scope atomic) store and drained with s_waitcnt vmcnt(0) before the arrival counter is bumped; the second arriver
  // learns from the value its own add returned that the sibling's box is out, and reads it with L1-bypassing loads.
  // No cache-wide release/acquire fences: they cost microseconds per tree level.
  float* bp = boxes + 6 * (size_t)cur;
  __hip_atomic_store(bp + 0, b.xmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(bp + 1, b.xmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(bp + 2, b.ymin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(bp + 3, b.ymax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(bp + 4, b.zmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(bp + 5, b.zmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int p = parent[cur];
  while (p != -1 && p < n - 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t prev = __hip_atomic_fetch_add(&arrived[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == 0) break;
    const uint32_t lc = child_l[p], rc = child_r[p];
    const float* a = boxes + 6 * (size_t)lc;
    const float* c = boxes + 6 * (size_t)rc;
    const float a0 = __hip_atomic_load(a + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a1 = __hip_atomic_load(a + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float a2 = __hip_atomic_load(a + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a3 = __hip_atomic_load(a + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float a4 = __hip_atomic_load(a + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a5 = __hip_atomic_load(a + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float c0 = __hip_atomic_load(c + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), c1 = __hip_atomic_load(c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float c2 = __hip_atomic_load(c + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), c3 = __hip_atomic_load(c + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float c4 = __hip_atomic_load(c + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), c5 = __hip_atomic_load(c + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // traversal record: both child boxes as (min, max) pairs per axis (the slab test works on pairs) + child references
    float4* rec = nodes + 4 * (size_t)p;
    rec[0] = make_float4(a0, a1, a2, a3);
    rec[1] = make_float4(a4, a5, c0, c1);
    rec[2] = make_float4(c2, c3, c4, c5);
    bool pure_l, pure_r;
    const uint32_t ref_l = make_ref(lc, leaf_base, order, refs, tris_before, range, prim_base16, &pure_l);
    const uint32_t ref_r = make_ref(rc, leaf_base, order, refs, tris_before, range, prim_base16, &pure_r);
    rec[3] = make_float4(__uint_as_float(ref_l), __uint_as_float(ref_r),
                         __uint_as_float(NODE_SWAP_ANY | ((pure_l && pure_r) ? NODE_SWAP_PURE : 0u)), 0.0f);
    // AABB(AABB, AABB), interval.cuh:83-88
    float* pb = boxes + 6 * (size_t)p;
    __hip_atomic_store(pb + 0, fminf(a0, c0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(pb + 1, fmaxf(a1, c1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pb + 2, fminf(a2, c2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(pb + 3, fmaxf(a3, c3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pb + 4, fminf(a4, c4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(pb + 5, fmaxf(a5, c5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cur = (uint32_t)p;
    p = parent[cur];
  }
}

} // namespace

// One stable radix pass over the low byte of the keys (render.hip orders a frame's samples by cost class with it): keys_out /
// vals_out receive the keys and -- vals_in == null -- their original positions in ascending key order.  ws: 256 * nblocks + 256 words.
size_t sort_low_byte_ws_words(long long n) { return 256 * (size_t)((n + SORT_TILE - 1) / SORT_TILE) + 256; }
int sort_low_byte(const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, long long n, uint32_t* ws, hipStream_t stream)
{
  const int sblocks = (int)((n + SORT_TILE - 1) / SORT_TILE);
  uint32_t *hist = ws, *totals = ws + 256 * (size_t)sblocks;
  hipLaunchKernelGGL(radix_pass_kernel<false>, dim3(sblocks), dim3(BLOCK), 0, stream, keys_in, (const uint32_t*)nullptr, keys_out, vals_out, hist, hist, totals, (int)n, 0, sblocks);
  hipLaunchKernelGGL(row_scan_kernel, dim3(256), dim3(BLOCK), 0, stream, hist, totals, sblocks);
  hipLaunchKernelGGL(radix_pass_kernel<true>, dim3(sblocks), dim3(BLOCK), 0, stream, keys_in, (const uint32_t*)nullptr, keys_out, vals_out, hist, hist, totals, (int)n, 0, sblocks);
  MIRT_HIP(hipGetLastError());
  return MIRT_OK;
}

int build_lbvh(MirtScene* sc, hipStream_t stream)
{
  const int n = sc->N;
  sc->built = false;
  sc->root_ref = REF_NONE;
  if (n == 0) { sc->built = true; sc->build_ms = 0.0f; return MIRT_OK; }   // main.cu:44: build skipped when there are no primitives

  // sort / refit workspace: one allocation, made before the timed region and kept with the scene (a rebuild reuses it)
  const int sblocks = (n + SORT_TILE - 1) / SORT_TILE;
  const size_t wwords = 4 * (size_t)n + 256 * (size_t)sblocks + 256 + (size_t)(n > 1 ? n - 1 : 0);
  if (sc->build_ws_words < wwords) {
    (void)hipFree(sc->build_ws); sc->build_ws = nullptr; sc->build_ws_words = 0;
    MIRT_HIP(hipMalloc(&sc->build_ws, sizeof(uint32_t) * wwords));
    sc->build_ws_words = wwords;
  }
  MIRT_HIP(hipEventRecord(sc->ev0, stream));
  // scene bounds
  const int nblk = (n + BLOCK - 1) / BLOCK;
  if (!sc->opt.bounds_as_shipped) {
    static const uint32_t init_keys[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    MIRT_HIP(hipMemcpyAsync(sc->bounds_keys, init_keys, sizeof(init_keys), hipMemcpyHostToDevice, stream));
    const int bgrid = nblk < 512 ? nblk : 512;
    hipLaunchKernelGGL(prim_bounds_kernel, dim3(bgrid), dim3(BLOCK), 0, stream, sc->refs_in, sc->spheres, sc->tri_verts, n, sc->bounds_keys);
  } else {
    // the shipped reference never stores the bounds it accumulates (parse.cpp:28): min = +inf, max = -inf, every code 0
    static const uint32_t shipped_keys[6] = {0xff800000u, 0xff800000u, 0xff800000u, 0x007fffffu, 0x007fffffu, 0x007fffffu};
    MIRT_HIP(hipMemcpyAsync(sc->bounds_keys, shipped_keys, sizeof(shipped_keys), hipMemcpyHostToDevice, stream));
  }

  // morton codes + stable sort
  uint32_t* const ws = sc->build_ws;
  uint32_t *k0 = ws, *v0 = ws + (size_t)n, *k1 = ws + 2 * (size_t)n, *v1 = ws + 3 * (size_t)n;
  uint32_t *hist = ws + 4 * (size_t)n, *totals = hist + 256 * (size_t)sblocks;
  hipLaunchKernelGGL(morton_kernel, dim3(nblk), dim3(BLOCK), 0, stream, sc->refs_in, sc->spheres, sc->tri_verts, n, sc->bounds_keys, k0, v0);
  uint32_t *ki = k0, *vi = v0, *ko = k1, *vo = v1;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = pass * 8;
    hipLaunchKernelGGL(radix_pass_kernel<false>, dim3(sblocks), dim3(BLOCK), 0, stream, ki, vi, ko, vo, hist, hist, totals, n, shift, sblocks);
    hipLaunchKernelGGL(row_scan_kernel, dim3(256), dim3(BLOCK), 0, stream, hist, totals, sblocks);
    hipLaunchKernelGGL(radix_pass_kernel<true>, dim3(sblocks), dim3(BLOCK), 0, stream, ki, vi, ko, vo, hist, hist, totals, n, shift, sblocks);
    uint32_t* t = ki; ki = ko; ko = t; t = vi; vi = vo; vo = t;
  }
  // after 4 passes the result is back in k0/v0
  MIRT_HIP(hipMemcpyAsync(sc->codes, ki, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, stream));
  MIRT_HIP(hipMemcpyAsync(sc->order, vi, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, stream));

  uint32_t* arrived = totals + 256;
  if (n > 1) {
    MIRT_HIP(hipMemsetAsync(arrived, 0, sizeof(uint32_t) * (n - 1), stream));
    MIRT_HIP(hipMemsetAsync(sc->parent, 0xff, sizeof(int) * (2 * (size_t)n - 1), stream));
    const int kblk = (n - 1 + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(karras_kernel, dim3(kblk), dim3(BLOCK), 0, stream, sc->codes, n, sc->child_l, sc->child_r, sc->parent, sc->range);
  } else {
    MIRT_HIP(hipMemsetAsync(sc->parent, 0xff, sizeof(int), stream));
  }
  // heap placement of the sorted leaves (the sort workspace is free again: the block sums live in it)
  const int scan_blocks = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t* const block_sums = k1;
  hipLaunchKernelGGL(tri_scan_kernel<false>, dim3(scan_blocks), dim3(BLOCK), 0, stream, n, sc->order, sc->refs_in, block_sums, sc->tris_before);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(BLOCK), 0, stream, block_sums, scan_blocks);
  hipLaunchKernelGGL(tri_scan_kernel<true>, dim3(scan_blocks), dim3(BLOCK), 0, stream, n, sc->order, sc->refs_in, block_sums, sc->tris_before);
  hipLaunchKernelGGL(scatter_prims_kernel, dim3(nblk), dim3(BLOCK), 0, stream, n, sc->order, sc->refs_in, sc->tris_before, sc->spheres, sc->tris,
                     reinterpret_cast<float4*>(sc->heap + sc->prim_base), sc->unit_prim);
  hipLaunchKernelGGL(refit_pack_kernel, dim3(nblk), dim3(BLOCK), 0, stream, n, sc->order, sc->refs_in, sc->spheres, sc->tri_verts,
                     sc->child_l, sc->child_r, sc->parent, arrived, sc->boxes, sc->nodes, sc->tris_before, sc->range, sc->prim_base / 16u, sc->tri_boxes);
  sc->root_ref_q = REF_NONE;
  if (sc->qnode_base && n > 1 && !sc->opt.bounds_as_shipped) {
    const int kblk = (n - 1 + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(pack_qnodes_kernel, dim3(kblk), dim3(BLOCK), 0, stream, n, sc->bounds_keys, sc->child_l, sc->child_r, sc->boxes,
                       sc->order, sc->refs_in, sc->tris_before, sc->range, reinterpret_cast<uint4*>(sc->heap + sc->qnode_base), sc->qparams, sc->qnode_base / 16u, sc->prim_base / 16u);
    sc->root_ref_q = sc->qnode_base / 16u;
  }
  sc->root_ref_w = REF_NONE;
  if (sc->wnode_base && n > 1 && !sc->opt.bounds_as_shipped) {
    const int kblk = (n - 1 + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(pack_wnodes_kernel, dim3(kblk), dim3(BLOCK), 0, stream, n, sc->bounds_keys, sc->child_l, sc->child_r, sc->boxes,
                       sc->order, sc->refs_in, sc->tris_before, sc->range, reinterpret_cast<uint4*>(sc->heap + sc->wnode_base),
                       sc->qparams, sc->wnode_base / 16u, sc->prim_base / 16u);
    sc->root_ref_w = sc->wnode_base / 16u;
  }
  MIRT_HIP(hipGetLastError());
  MIRT_HIP(hipEventRecord(sc->ev1, stream));
  MIRT_HIP(hipStreamSynchronize(stream));   // lbvh_builder.cu:475
  MIRT_HIP(hipEventElapsedTime(&sc->build_ms, sc->ev0, sc->ev1));
  // the scene box, from the root record's two child boxes: its largest coordinate magnitude (RenderArgs::reach_slack), and whether
  // the grid of the quantised records resolves the scene's coordinates (grid_ok: on every axis they stay below 64 extents, i.e.
  // ulp(coordinate) below a quarter of a grid step).  Only then do the outward-rounded quantised boxes cover what float rounding
  // does to a primitive's own box planes and to a sphere's hit distance against them; a scene that fails
  // the test -- one that sits far from the world origin compared with its size -- is walked over the exact records.
  sc->coord_max = 0.0f;
  sc->grid_ok = false;
  if (n > 1) {
    float rec[12];      // left x, y | left z, right x | right y, z  (min, max pairs)
    MIRT_HIP(hipMemcpy(rec, sc->nodes, sizeof(rec), hipMemcpyDeviceToHost));
    for (float v : rec) sc->coord_max = fmaxf(sc->coord_max, fabsf(v));
    sc->grid_ok = true;
    for (int k = 0; k < 3; ++k) {
      const float lo = fminf(rec[2 * k], rec[6 + 2 * k]), hi = fmaxf(rec[2 * k + 1], rec[6 + 2 * k + 1]);
      if (!(fmaxf(fabsf(lo), fabsf(hi)) <= 64.0f * (hi - lo))) sc->grid_ok = false;
    }
  }
  // root: node 0, unless the whole scene is a single primitive
  sc->root_ref = (n == 1) ? (REF_LEAF | (sc->Nt ? REF_TRI : 0u) | (sc->prim_base / 16u)) : 0u;
  sc->built = true;
  return MIRT_OK;
}

int get_tree(MirtScene* sc, MirtTreeNode* nodes, uint32_t* codes, MirtPrimRef* refs, float* bounds)
{
  const int n = sc->N;
  if (!sc->built) { set_error("mirt_get_tree: LBVH not built"); return MIRT_ERR_STATE; }
  if (n == 0) return MIRT_OK;
  MIRT_HIP(hipDeviceSynchronize());
  std::vector<uint32_t> order(n), cl(n > 1 ? n - 1 : 0), cr(n > 1 ? n - 1 : 0);
  std::vector<MirtPrimRef> rin(n);
  std::vector<float> boxes(6 * (2 * (size_t)n - 1));
  MIRT_HIP(hipMemcpy(order.data(), sc->order, 4 * (size_t)n, hipMemcpyDeviceToHost));
  MIRT_HIP(hipMemcpy(rin.data(), sc->refs_in, sizeof(MirtPrimRef) * (size_t)n, hipMemcpyDeviceToHost));
  MIRT_HIP(hipMemcpy(boxes.data(), sc->boxes, 4 * boxes.size(), hipMemcpyDeviceToHost));
  if (n > 1) {
    MIRT_HIP(hipMemcpy(cl.data(), sc->child_l, 4 * (size_t)(n - 1), hipMemcpyDeviceToHost));
    MIRT_HIP(hipMemcpy(cr.data(), sc->child_r, 4 * (size_t)(n - 1), hipMemcpyDeviceToHost));
  }
  if (codes) MIRT_HIP(hipMemcpy(codes, sc->codes, 4 * (size_t)n, hipMemcpyDeviceToHost));
  if (refs) for (int i = 0; i < n; ++i) refs[i] = rin[order[i]];
  if (nodes) {
    for (int i = 0; i < 2 * n - 1; ++i) {
      MirtTreeNode& t = nodes[i];
      const float* b = &boxes[6 * (size_t)i];
      t.xmin = b[0]; t.xmax = b[1]; t.ymin = b[2]; t.ymax = b[3]; t.zmin = b[4]; t.zmax = b[5];
      if (i < n - 1) { t.left = cl[i]; t.right = cr[i]; t.prim_offset = 0; t.count = 0; }
      else { t.left = 0; t.right = 0; t.prim_offset = (uint32_t)(i - (n - 1)); t.count = 1; }
    }
  }
  if (bounds) {
    uint32_t k[6];
    MIRT_HIP(hipMemcpy(k, sc->bounds_keys, sizeof(k), hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; ++i) {
      uint32_t u = (k[i] & 0x80000000u) ? (k[i] & 0x7fffffffu) : ~k[i];
      memcpy(&bounds[i], &u, 4);
    }
  }
  return MIRT_OK;
}

} // namespace mirt
