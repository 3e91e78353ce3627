/*
 * oracle/omath.h -- TEST INFRASTRUCTURE ONLY (part of the CPU oracle; see oracle/oracle.cpp).
 *
 * Deterministic transcendental functions for the oracle.  The reference calls CUDA's libm
 * (powf in helper.cu:21, expf in helper.cu:44, sinf/cosf in struct.cu:46-47,38 and helper.cu:97-98,
 * logf/sincosf inside curand_normal).  CUDA's libm is not available here, and glibc's and ROCm OCML's
 * versions differ from it and from each other in the last ulp, which is enough to send a 16-bounce
 * ray chain down a different path.  So the oracle and the HIP product each carry their own copy of the
 * same small algorithms, built only from IEEE +,-,*,/ on double (identical on x86-64 and gfx950 when
 * contraction is off), and the results are rounded once to float.  Accuracy: < 0.501 ulp of the float
 * result, i.e. equal to the correctly rounded value except in ~1e-8 of cases.
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

static inline float o_logf(float x) { return (float)o_log((double)x); }
static inline float o_expf(float x) { return (float)o_exp((double)x); }
static inline float o_sinf(float x) { double s, c; o_sincos((double)x, &s, &c); return (float)s; }
static inline float o_cosf(float x) { double s, c; o_sincos((double)x, &s, &c); return (float)c; }
/* powf for x >= 0 (the only use is the sRGB curve, helper.cu:21); x < 0 -> NaN like a non-integer power */
static inline float o_powf(float x, float y)
{
  if (x != x || y != y) return (float)o_nan();
  if (x < 0.0f) return (float)o_nan();
  if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : (float)o_inf());
  return (float)o_exp((double)y * o_log((double)x));
}

#endif
