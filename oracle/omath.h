/*
 * oracle/omath.h -- TEST INFRASTRUCTURE ONLY (part of the CPU oracle; see oracle/oracle.cpp).
 *
 * Deterministic transcendental functions for the oracle.  The reference calls CUDA's libm
 * (powf in helper.cu:21, expf in helper.cu:44, sinf/cosf in struct.cu:46-47,38 and helper.cu:97-98,
 * logf/sincosf inside curand_normal).  CUDA's libm is not available here, and glibc's and ROCm OCML's
 * versions differ from it and from each other in the last ulp, which is enough to send a 16-bounce
 * ray chain down a different path.  So the oracle and the HIP product each carry their own copy of the
 * same small algorithms, built only from IEEE +,-,*,/ on double (identical on x86-64 and gfx950 when
 * contraction is off).  The double forms (rounded once to float: < 0.501 ulp) serve the sRGB power; logf / expf / sinf /
 * cosf are single-precision forms of the same kind (IEEE +,-,*,/ on float), within ~2 ulp -- the accuracy class of CUDA's
 * libm -- because the shading code calls them per shading node and the double forms were 6-9 % of a frame.
 *
 * Compile with -ffp-contract=off.
 */
#ifndef ORACLE_OMATH_H
#define ORACLE_OMATH_H

#include <stdint.h>
#include <string.h>

static inline uint64_t o_bits(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
static inline double o_frombits(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }
static inline double o_nan(void) { return o_frombits(0x7ff8000000000000ULL); }
static inline double o_inf(void) { return o_frombits(0x7ff0000000000000ULL); }

/* natural log, double in / double out, |rel err| ~ 2e-16 */
static inline double o_log(double x)
{
  if (x != x) return x;
  if (x < 0.0) return o_nan();
  if (x == 0.0) return -o_inf();
  if (x == o_inf()) return x;
  uint64_t b = o_bits(x);
  int e = (int)((b >> 52) & 0x7ff);
  if (e == 0) { /* subnormal double */
    x = x * 18014398509481984.0; /* 2^54 */
    b = o_bits(x);
    e = (int)((b >> 52) & 0x7ff) - 54;
  }
  e -= 1023;
  double m = o_frombits((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
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

/* e^x, double in / double out */
static inline double o_exp(double x)
{
  if (x != x) return x;
  if (x > 709.0) return o_inf();
  if (x < -745.0) return 0.0;
  double t = x * 1.4426950408889634;
  int k = (int)(t + (t < 0.0 ? -0.5 : 0.5));
  double kd = (double)k;
  double r = (x - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10;
  double p = 1.0 / 6227020800.0;        /* 1/13! */
  p = p * r + 1.0 / 479001600.0;        /* 1/12! */
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
  double s1 = o_frombits((uint64_t)(1023 + k1) << 52);
  double s2 = o_frombits((uint64_t)(1023 + k2) << 52);
  return p * s1 * s2;
}

/* sin and cos together; intended for |x| < ~1e6 (angles here are within [-2pi, 2pi]) */
static inline void o_sincos(double x, double* sn, double* cs)
{
  if (x != x || x == o_inf() || x == -o_inf()) { *sn = o_nan(); *cs = o_nan(); return; }
  double t = x * 0.63661977236758134308; /* 2/pi */
  if (t > 1.0e9) t = 1.0e9;
  if (t < -1.0e9) t = -1.0e9;
  int k = (int)(t + (t < 0.0 ? -0.5 : 0.5));
  double kd = (double)k;
  double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
  double z = r * r;
  double ps = -1.0 / 355687428096000.0;   /* -1/17! */
  ps = ps * z + 1.0 / 1307674368000.0;    /*  1/15! */
  ps = ps * z - 1.0 / 6227020800.0;       /* -1/13! */
  ps = ps * z + 1.0 / 39916800.0;         /*  1/11! */
  ps = ps * z - 1.0 / 362880.0;           /* -1/9!  */
  ps = ps * z + 1.0 / 5040.0;
  ps = ps * z - 1.0 / 120.0;
  ps = ps * z + 1.0 / 6.0;
  /* sin r = r - r*z*ps' ; keep the literal form below */
  double sr = r - r * z * ps;
  double pc = -1.0 / 6402373705728000.0;  /* -1/18! */
  pc = pc * z + 1.0 / 20922789888000.0;   /*  1/16! */
  pc = pc * z - 1.0 / 87178291200.0;      /* -1/14! */
  pc = pc * z + 1.0 / 479001600.0;        /*  1/12! */
  pc = pc * z - 1.0 / 3628800.0;          /* -1/10! */
  pc = pc * z + 1.0 / 40320.0;            /*  1/8!  */
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
static inline float o_u2f(uint32_t b) { float x; memcpy(&x, &b, 4); return x; }
static inline uint32_t o_f2u(float x) { uint32_t b; memcpy(&b, &x, 4); return b; }

static inline float o_logf(float x)
{
  if (x != x) return x;
  if (x < 0.0f) return o_u2f(0x7fc00000u);
  if (x == 0.0f) return o_u2f(0xff800000u);
  if (x == o_u2f(0x7f800000u)) return x;
  uint32_t b = o_f2u(x);
  int e = (int)((b >> 23) & 0xffu);
  if (e == 0) {                                   /* subnormal */
    x = x * 8388608.0f;                           /* 2^23 */
    b = o_f2u(x);
    e = (int)((b >> 23) & 0xffu) - 23;
  }
  e -= 127;
  float m = o_u2f((b & 0x007fffffu) | 0x3f800000u);   /* [1, 2) */
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

static inline float o_expf(float x)
{
  if (x != x) return x;
  if (x > 88.7228394f) return o_u2f(0x7f800000u);
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
  return p * o_u2f((uint32_t)(127 + k1) << 23) * o_u2f((uint32_t)(127 + k2) << 23);
}

/* sin and cos together; intended for |x| up to a few thousand (angles here are within [-2 pi, 2 pi]) */
static inline void o_sincosf(float x, float* sn, float* cs)
{
  if (x != x || x == o_u2f(0x7f800000u) || x == o_u2f(0xff800000u)) { *sn = o_u2f(0x7fc00000u); *cs = o_u2f(0x7fc00000u); return; }
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
static inline float o_sinf(float x) { float s, c; o_sincosf(x, &s, &c); return s; }
static inline float o_cosf(float x) { float s, c; o_sincosf(x, &s, &c); return c; }

/* powf for x >= 0 (the only use is the sRGB curve, helper.cu:21); x < 0 -> NaN like a non-integer power */
static inline float o_powf(float x, float y)
{
  if (x != x || y != y) return (float)o_nan();
  if (x < 0.0f) return (float)o_nan();
  if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : (float)o_inf());
  return (float)o_exp((double)y * o_log((double)x));
}

#endif
