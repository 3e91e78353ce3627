// Device-side building blocks shared by the LBVH build and the render kernels (gfx950 only).
//   * f3: float3 arithmetic with the reference's operator semantics (vec3.cuh:21-107)
//   * dm_*: deterministic logf/expf/sinf/cosf/powf built from IEEE double +,-,*,/ only, so that every
//     value the kernels compute is reproducible bit for bit on any IEEE machine (built with -ffp-contract=off;
//     hipcc's default correctly rounded fp32 divide/sqrt)
//   * Xorwow: cuRAND-compatible generator state kept in registers
#ifndef MIRT_DEVICE_COMMON_H
#define MIRT_DEVICE_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define MIRT_DEV __device__ __forceinline__

namespace mirt {

struct f3 { float x, y, z; };
MIRT_DEV f3 mk3(float x, float y, float z) { f3 v; v.x = x; v.y = y; v.z = z; return v; }
MIRT_DEV f3 operator+(const f3& a, const f3& b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
MIRT_DEV f3 operator-(const f3& a, const f3& b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
MIRT_DEV f3 operator*(const f3& a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
MIRT_DEV f3 operator*(float s, const f3& a) { return mk3(a.x * s, a.y * s, a.z * s); }
MIRT_DEV f3 operator*(const f3& a, const f3& b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }   // RGB*RGB, struct.cuh:30-33
MIRT_DEV f3 operator/(const f3& a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
MIRT_DEV f3 operator-(const f3& a) { return mk3(-a.x, -a.y, -a.z); }
MIRT_DEV float dot(const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vec3.cuh:7-18
MIRT_DEV bool fequal(float a, float b)
{
  float diff = fabsf(a - b);
  float largest = fmaxf(fabsf(a), fabsf(b));
  if (largest < 1e-6f) return diff < 1e-6f;
  return diff / largest < 1e-6f;
}
MIRT_DEV float length(const f3& v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
// fequal(x, 0) (vec3.cuh:7-18) without its division: largest = |x|; below 1e-6 the answer is |x| < 1e-6 = true, otherwise it
// is |x|/|x| < 1e-6 = false (inf/inf and NaN compare false on both routes) -- so it is exactly |x| < 1e-6.
MIRT_DEV bool is_zero(float x) { return fabsf(x) < 1e-6f; }
// vec3.cuh:72-82
MIRT_DEV f3 normalize(const f3& v)
{
  float mag = length(v);
  if (is_zero(mag)) return mk3(0.0f, 0.0f, 0.0f);
  float inv = 1.0f / mag;
  return mk3(v.x * inv, v.y * inv, v.z * inv);
}
// RGB == RGB(0,0,0), struct.cuh:20-23
MIRT_DEV bool is_black(const f3& c) { return is_zero(c.x) && is_zero(c.y) && is_zero(c.z); }

// ---- deterministic transcendental functions -----------------------------------------------------
MIRT_DEV double dm_frombits(uint64_t b) { return __longlong_as_double((long long)b); }
MIRT_DEV uint64_t dm_bits(double x) { return (uint64_t)__double_as_longlong(x); }
MIRT_DEV double dm_nan() { return dm_frombits(0x7ff8000000000000ULL); }
MIRT_DEV double dm_inf() { return dm_frombits(0x7ff0000000000000ULL); }

MIRT_DEV double dm_log(double x)
{
  if (x != x) return x;
  if (x < 0.0) return dm_nan();
  if (x == 0.0) return -dm_inf();
  if (x == dm_inf()) return x;
  uint64_t b = dm_bits(x);
  int e = (int)((b >> 52) & 0x7ff);
  if (e == 0) {
    x = x * 18014398509481984.0;
    b = dm_bits(x);
    e = (int)((b >> 52) & 0x7ff) - 54;
  }
  e -= 1023;
  double m = dm_frombits((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
  double s = (m - 1.0) / (m + 1.0);
  double z = s * s;
  double p = 1.0 / 23.0;
  p = p * z + 1.0 / 21.0;
  p = p * z + 1.0 / 19.0;
  p = p * z + 1.0 / 17.0;
  p = p * z + 1.0 / 15.0;
  p = p * z + 1.0 / 13.0;
  p = p * z + 1.0 / 11.0;
  p = p * z + 1.0 / 9.0;
  p = p * z + 1.0 / 7.0;
  p = p * z + 1.0 / 5.0;
  p = p * z + 1.0 / 3.0;
  p = p * z + 1.0;
  return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

MIRT_DEV double dm_exp(double x)
{
  if (x != x) return x;
  if (x > 709.0) return dm_inf();
  if (x < -745.0) return 0.0;
  double t = x * 1.4426950408889634;
  int k = (int)(t + (t < 0.0 ? -0.5 : 0.5));
  double kd = (double)k;
  double r = (x - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10;
  double p = 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  int k1 = k / 2, k2 = k - k1;
  double s1 = dm_frombits((uint64_t)(1023 + k1) << 52);
  double s2 = dm_frombits((uint64_t)(1023 + k2) << 52);
  return p * s1 * s2;
}

MIRT_DEV void dm_sincos(double x, double* sn, double* cs)
{
  if (x != x || x == dm_inf() || x == -dm_inf()) { *sn = dm_nan(); *cs = dm_nan(); return; }
  double t = x * 0.63661977236758134308;
  if (t > 1.0e9) t = 1.0e9;
  if (t < -1.0e9) t = -1.0e9;
  int k = (int)(t + (t < 0.0 ? -0.5 : 0.5));
  double kd = (double)k;
  double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
  double z = r * r;
  double ps = -1.0 / 355687428096000.0;
  ps = ps * z + 1.0 / 1307674368000.0;
  ps = ps * z - 1.0 / 6227020800.0;
  ps = ps * z + 1.0 / 39916800.0;
  ps = ps * z - 1.0 / 362880.0;
  ps = ps * z + 1.0 / 5040.0;
  ps = ps * z - 1.0 / 120.0;
  ps = ps * z + 1.0 / 6.0;
  double sr = r - r * z * ps;
  double pc = -1.0 / 6402373705728000.0;
  pc = pc * z + 1.0 / 20922789888000.0;
  pc = pc * z - 1.0 / 87178291200.0;
  pc = pc * z + 1.0 / 479001600.0;
  pc = pc * z - 1.0 / 3628800.0;
  pc = pc * z + 1.0 / 40320.0;
  pc = pc * z - 1.0 / 720.0;
  pc = pc * z + 1.0 / 24.0;
  pc = pc * z - 0.5;
  double cr = 1.0 + z * pc;
  switch (k & 3) {
    case 0: *sn = sr;  *cs = cr;  break;
    case 1: *sn = cr;  *cs = -sr; break;
    case 2: *sn = -sr; *cs = -cr; break;
    default: *sn = -cr; *cs = sr; break;
  }
}


/* ---- single-precision forms (logf, expf, sinf, cosf as the shading code calls them) -----------------------------------
 * Plain float arithmetic, one rounding per operation, Cody-Waite argument reduction, short Horner polynomials: within
 * ~2 ulp of the true value over the ranges the renderer uses (|angle| <= 2 pi, exp arguments in [-104, 88]) -- the accuracy
 * class of CUDA's own libm (sinf/cosf/expf 2 ulp, logf 1 ulp), four to five times cheaper than the double forms above, which
 * are kept for the sRGB power only. */
MIRT_DEV float dm_u2f(uint32_t b) { return __uint_as_float(b); }
MIRT_DEV uint32_t dm_f2u(float x) { return __float_as_uint(x); }

MIRT_DEV float dm_logf(float x)
{
  if (x != x) return x;
  if (x < 0.0f) return dm_u2f(0x7fc00000u);
  if (x == 0.0f) return dm_u2f(0xff800000u);
  if (x == dm_u2f(0x7f800000u)) return x;
  uint32_t b = dm_f2u(x);
  int e = (int)((b >> 23) & 0xffu);
  if (e == 0) {                                   /* subnormal */
    x = x * 8388608.0f;                           /* 2^23 */
    b = dm_f2u(x);
    e = (int)((b >> 23) & 0xffu) - 23;
  }
  e -= 127;
  float m = dm_u2f((b & 0x007fffffu) | 0x3f800000u);   /* [1, 2) */
  if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
  const float s = (m - 1.0f) / (m + 1.0f);        /* log m = 2 s (1 + z/3 + z^2/5 + z^3/7 + z^4/9 + ...), z = s^2 <= 0.0295 */
  const float z = s * s;
  float p = 1.0f / 9.0f;
  p = p * z + 1.0f / 7.0f;
  p = p * z + 1.0f / 5.0f;
  p = p * z + 1.0f / 3.0f;
  const float s2 = 2.0f * s;
  const float ef = (float)e;
  /* ln 2 = 0.693145752 (15 significant bits: e * hi is exact) + 1.42860677e-06 */
  return ef * 0.693145752f + (s2 + (s2 * z * p + ef * 1.42860677e-06f));
}

MIRT_DEV float dm_expf(float x)
{
  if (x != x) return x;
  if (x > 88.7228394f) return dm_u2f(0x7f800000u);
  if (x < -103.972076f) return 0.0f;
  const float t = x * 1.44269502f;
  const int k = (int)(t + (t < 0.0f ? -0.5f : 0.5f));
  const float kf = (float)k;
  const float r = (x - kf * 0.693145752f) - kf * 1.42860677e-06f;      /* |r| <= 0.3466 */
  float p = 1.0f / 5040.0f;
  p = p * r + 1.0f / 720.0f;
  p = p * r + 1.0f / 120.0f;
  p = p * r + 1.0f / 24.0f;
  p = p * r + 1.0f / 6.0f;
  p = p * r + 0.5f;
  p = p * r + 1.0f;
  p = p * r + 1.0f;
  const int k1 = k / 2, k2 = k - k1;               /* 2^k in two normal factors: reaches the subnormal results, never overflows early */
  return p * dm_u2f((uint32_t)(127 + k1) << 23) * dm_u2f((uint32_t)(127 + k2) << 23);
}

/* sin and cos together; intended for |x| up to a few thousand (angles here are within [-2 pi, 2 pi]) */
MIRT_DEV void dm_sincosf(float x, float* sn, float* cs)
{
  if (x != x || x == dm_u2f(0x7f800000u) || x == dm_u2f(0xff800000u)) { *sn = dm_u2f(0x7fc00000u); *cs = dm_u2f(0x7fc00000u); return; }
  float t = x * 0.636619747f;                      /* 2 / pi */
  if (t > 1.0e6f) t = 1.0e6f;
  if (t < -1.0e6f) t = -1.0e6f;
  const int k = (int)(t + (t < 0.0f ? -0.5f : 0.5f));
  const float kf = (float)k;
  /* pi / 2 = 1.57077026 (16 significant bits) + 2.60630623e-05 (16 bits) + 6.07709438e-11 */
  const float r = ((x - kf * 1.57077026f) - kf * 2.60630623e-05f) - kf * 6.07709438e-11f;   /* |r| <= pi / 4 */
  const float z = r * r;
  float ps = 1.0f / 362880.0f;
  ps = ps * z - 1.0f / 5040.0f;
  ps = ps * z + 1.0f / 120.0f;
  ps = ps * z - 1.0f / 6.0f;
  const float sr = r + r * z * ps;
  float pc = -1.0f / 3628800.0f;
  pc = pc * z + 1.0f / 40320.0f;
  pc = pc * z - 1.0f / 720.0f;
  pc = pc * z + 1.0f / 24.0f;
  pc = pc * z - 0.5f;
  const float cr = 1.0f + z * pc;
  switch (k & 3) {
    case 0: *sn = sr;  *cs = cr;  break;
    case 1: *sn = cr;  *cs = -sr; break;
    case 2: *sn = -sr; *cs = -cr; break;
    default: *sn = -cr; *cs = sr; break;
  }
}
MIRT_DEV float dm_sinf(float x) { float s, c; dm_sincosf(x, &s, &c); return s; }
MIRT_DEV float dm_cosf(float x) { float s, c; dm_sincosf(x, &s, &c); return c; }

MIRT_DEV float dm_powf(float x, float y)
{
  if (x != x || y != y) return (float)dm_nan();
  if (x < 0.0f) return (float)dm_nan();
  if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : (float)dm_inf());
  return (float)dm_exp((double)y * dm_log((double)x));
}
// RGBtosRGB, helper.cu:12-27
MIRT_DEV float rgb_to_srgb(float l)
{
  float sol;
  if (l < 0.0031308f) sol = 12.92f * l;
  else sol = 1.055f * dm_powf(l, 1 / 2.4f) - 0.055f;
  sol = fminf(1.0f, fmaxf(0.0f, sol));
  return sol;
}

// ---- XORWOW (curand_kernel.h semantics) -----------------------------------------
struct Xorwow {
  uint32_t v0, v1, v2, v3, v4, d;
  float bm_extra;
  int bm_flag;
};
MIRT_DEV uint32_t xw_next(Xorwow& s)
{
  uint32_t t = s.v0 ^ (s.v0 >> 2);
  s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4;
  s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
  s.d += 362437u;
  return s.v4 + s.d;
}
// curand_uniform: (0, 1]
MIRT_DEV float xw_uniform(Xorwow& s) { return (float)xw_next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f); }
// curand_normal: Box-Muller pair, second value cached
MIRT_DEV float xw_normal(Xorwow& s)
{
  if (s.bm_flag) { s.bm_flag = 0; return s.bm_extra; }
  uint32_t x = xw_next(s);
  uint32_t y = xw_next(s);
  float u = (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
  float v = (float)y * 1.46291807926e-9f + (1.46291807926e-9f / 2.0f);
  float sq = sqrtf(-2.0f * dm_logf(u));
  double sn, cs;
  dm_sincos((double)v, &sn, &cs);
  s.bm_extra = sq * (float)cs;
  s.bm_flag = 1;
  return sq * (float)sn;
}
// randD / standerdD, helper.cu:82-89
MIRT_DEV float randD(float start, float end, Xorwow& s) { float u = xw_uniform(s); return start + (end - start) * u; }
MIRT_DEV float standerdD(float stddev, Xorwow& s) { return xw_normal(s) * stddev; }

// Tables for the sequence skip-ahead (xorwow_tables.h), device copies.
struct RngTablesDev {
  const uint4* A;
  const uint32_t* B;
  const uint32_t* K;
  const uint32_t* R2;
  int mode, chunk_bits, nin_words, nchunks;
  uint32_t d0;
};

template <int NIN, int BITS>
MIRT_DEV void xw_apply_tables(const RngTablesDev& t, uint32_t m, const uint32_t* in, uint32_t* out)
{
  constexpr int PER_WORD = 32 / BITS;
  constexpr int NV = 1 << BITS;
  const size_t base = (size_t)m * (NIN * PER_WORD) * NV;
  uint32_t a0 = t.K[m * 5 + 0], a1 = t.K[m * 5 + 1], a2 = t.K[m * 5 + 2], a3 = t.K[m * 5 + 3], a4 = t.K[m * 5 + 4];
#pragma unroll
  for (int iw = 0; iw < NIN; ++iw) {
    const uint32_t w = in[iw];
#pragma unroll
    for (int c = 0; c < PER_WORD; ++c) {
      const uint32_t val = (w >> (c * BITS)) & (NV - 1);
      const size_t e = base + (size_t)(iw * PER_WORD + c) * NV + val;
      const uint4 q = t.A[e];
      a0 ^= q.x; a1 ^= q.y; a2 ^= q.z; a3 ^= q.w;
      a4 ^= t.B[e];
      // the 4-bit tables take 24 (40) lookups: issued all at once the loaded values do not fit the registers the shade phase
      // has, and the compiler spilled each of them to scratch behind its own wait -- six lookups in flight at a time instead
      if (BITS == 4 && ((iw * PER_WORD + c) % 6) == 5) asm volatile("" ::: "memory");      // (six: 128 spp 354 -> 286 ms; four: 295 ms)
    }
  }
  out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3; out[4] = a4;
}

// curand_init(1234 + pixel, sample, 0) for spp > 1 (draw.cu:162) / curand_init(1234, pixel, 0) for spp <= 1 (draw.cu:105)
// TABLES: 0 = whatever the tables are (run-time branches); 8 = per-sample tables with 8-bit chunks only (spp 2..16); 4 = the
// 4-bit forms only (per-sample tables of spp > 16, and the per-pixel tables of spp <= 1).  The trace kernel is instantiated
// per form: it runs at the register edge, and the code of the form it does not use moved its allocation by a per cent or two.
template <int TABLES = 0>
MIRT_DEV void xw_init(Xorwow& s, const RngTablesDev& t, uint32_t pixel, uint32_t sample)
{
  uint32_t out[5];
  if (TABLES == 8 || (TABLES != 4 && t.mode == 0) || (TABLES == 4 && t.mode == 0)) {
    const uint32_t seed_lo = 1234u + pixel;
    const uint32_t s0 = seed_lo ^ 0xaad26b49u;
    const uint32_t s1 = 0u ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    s.d = 6615241u + t1 + t0;
    uint32_t in[3] = {123456789u + t0, 362436069u ^ t0, 5783321u + t0};
    if (TABLES == 8 || (TABLES == 0 && t.chunk_bits == 8)) xw_apply_tables<3, 8>(t, sample, in, out);
    else xw_apply_tables<3, 4>(t, sample, in, out);
  } else {
    const uint32_t blk = pixel >> 8;
    uint32_t in[5] = {t.R2[blk * 5 + 0], t.R2[blk * 5 + 1], t.R2[blk * 5 + 2], t.R2[blk * 5 + 3], t.R2[blk * 5 + 4]};
    s.d = t.d0;
    xw_apply_tables<5, 4>(t, pixel & 255u, in, out);
  }
  s.v0 = out[0]; s.v1 = out[1]; s.v2 = out[2]; s.v3 = out[3]; s.v4 = out[4];
  s.bm_flag = 0; s.bm_extra = 0.0f;
}

} // namespace mirt
#endif
