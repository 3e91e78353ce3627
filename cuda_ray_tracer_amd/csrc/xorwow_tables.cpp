// Host-side precomputation for the cuRAND-compatible XORWOW generator (the reference seeds every sample with
// curand_init(seed, subsequence, 0): draw.cu:105 and draw.cu:162).  skipahead_sequence(n) advances the five
// LFSR words by n * 2^67 steps, a GF(2)-linear map; this file builds that map from the step function alone
// and turns it into chunked lookup tables the render kernel can apply with a few dozen loads per sample.
#include "xorwow_tables.h"

#include <cstring>

namespace mirt {
namespace {

inline void xw_step(uint32_t* v)
{
  uint32_t t = v[0] ^ (v[0] >> 2);
  v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
  v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

struct Mat160 { uint32_t col[160][5]; };   // column i = image of basis bit i

void apply(const Mat160& m, const uint32_t* in, uint32_t* out)
{
  uint32_t acc[5] = {0, 0, 0, 0, 0};
  for (int w = 0; w < 5; ++w)
    for (uint32_t bits = in[w]; bits; bits &= bits - 1) {
      const uint32_t* c = m.col[w * 32 + __builtin_ctz(bits)];
      for (int k = 0; k < 5; ++k) acc[k] ^= c[k];
    }
  memcpy(out, acc, sizeof(acc));
}
void mul(const Mat160& a, const Mat160& b, Mat160& out)   // out = a o b
{
  Mat160 t;
  for (int i = 0; i < 160; ++i) apply(a, b.col[i], t.col[i]);
  out = t;
}
void identity(Mat160& m)
{
  memset(&m, 0, sizeof(m));
  for (int i = 0; i < 160; ++i) m.col[i][i / 32] = 1u << (i % 32);
}

const Mat160& sequence_jump()   // step^(2^67)
{
  static Mat160 j;
  static bool ready = false;
  if (!ready) {
    Mat160 a;
    for (int i = 0; i < 160; ++i) {
      uint32_t v[5] = {0, 0, 0, 0, 0};
      v[i / 32] = 1u << (i % 32);
      xw_step(v);
      memcpy(a.col[i], v, sizeof(v));
    }
    for (int k = 0; k < 67; ++k) mul(a, a, a);
    j = a;
    ready = true;
  }
  return j;
}

// Expand matrix m into chunk tables for the listed state words.
void expand(const Mat160& m, const int* state_words, int nin, int bits, uint32_t* A, uint32_t* B)
{
  const int per_word = 32 / bits, nvals = 1 << bits;
  for (int iw = 0; iw < nin; ++iw)
    for (int c = 0; c < per_word; ++c)
      for (int val = 0; val < nvals; ++val) {
        uint32_t acc[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < bits; ++b)
          if (val & (1 << b)) {
            const uint32_t* col = m.col[state_words[iw] * 32 + c * bits + b];
            for (int k = 0; k < 5; ++k) acc[k] ^= col[k];
          }
        const size_t e = ((size_t)(iw * per_word + c)) * nvals + val;
        memcpy(A + e * 4, acc, 16);
        B[e] = acc[4];
      }
}

} // namespace

void xorwow_seed(uint64_t seed, uint32_t v[5], uint32_t* d)
{
  uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
  uint32_t s1 = ((uint32_t)(seed >> 32)) ^ 0xf7dcefddu;
  uint32_t t0 = 1099087573u * s0;
  uint32_t t1 = 2591861531u * s1;
  *d = 6615241u + t1 + t0;
  v[0] = 123456789u + t0;
  v[1] = 362436069u ^ t0;
  v[2] = 521288629u + t1;
  v[3] = 88675123u ^ t1;
  v[4] = 5783321u + t0;
}

void build_sample_tables(int spp, RngTables& t)
{
  t.mode = 0;
  t.num_mats = spp;
  // byte-indexed tables (61 KB per sample index, 12 lookups) while all of them stay in an XCD's L2 with the tree; nibble-indexed
  // ones (7.7 KB, 24 lookups) beyond: at 64 spp the byte tables are 3.9 MB and the frame ran 7-8 % slower (32 spp: 2 %; at
  // 16 spp the byte tables are 0.5 % faster)
#ifndef MIRT_RNG8_MAX_SPP
#define MIRT_RNG8_MAX_SPP 16
#endif
  t.chunk_bits = (spp <= MIRT_RNG8_MAX_SPP) ? 8 : 4;
  t.nin_words = 3;
  t.nchunks = t.nin_words * 32 / t.chunk_bits;
  const size_t per_mat = (size_t)t.nchunks << t.chunk_bits;
  t.A.assign(per_mat * 4 * spp, 0); t.B.assign(per_mat * spp, 0); t.K.assign((size_t)5 * spp, 0);
  t.R2.clear();
  // seed words that do not depend on the low 32 seed bits (seed = 1234 + pixel < 2^32): v2, v3
  uint32_t v[5], d;
  xorwow_seed(0, v, &d);
  const uint32_t fixed[5] = {0, 0, v[2], v[3], 0};
  static const int words[3] = {0, 1, 4};
  Mat160 cur; identity(cur);
  const Mat160& j1 = sequence_jump();
  for (int s = 0; s < spp; ++s) {
    expand(cur, words, 3, t.chunk_bits, &t.A[per_mat * 4 * s], &t.B[per_mat * s]);
    apply(cur, fixed, &t.K[(size_t)5 * s]);
    mul(j1, cur, cur);
  }
}

void build_pixel_tables(int64_t num_pixels, uint64_t seed, RngTables& t)
{
  t.mode = 1;
  t.num_mats = 256;
  t.chunk_bits = 4;
  t.nin_words = 5;
  t.nchunks = 40;
  const size_t per_mat = (size_t)t.nchunks << t.chunk_bits;
  t.A.assign(per_mat * 4 * 256, 0); t.B.assign(per_mat * 256, 0); t.K.assign((size_t)5 * 256, 0);
  static const int words[5] = {0, 1, 2, 3, 4};
  Mat160 cur; identity(cur);
  const Mat160& j1 = sequence_jump();
  for (int m = 0; m < 256; ++m) {
    expand(cur, words, 5, 4, &t.A[per_mat * 4 * m], &t.B[per_mat * m]);
    mul(j1, cur, cur);
  }
  // cur is now jump^256
  const int64_t nblk = (num_pixels + 255) / 256;
  t.R2.assign((size_t)nblk * 5, 0);
  uint32_t v[5], d;
  xorwow_seed(seed, v, &d);
  t.d0 = d;
  for (int64_t k = 0; k < nblk; ++k) {
    memcpy(&t.R2[(size_t)k * 5], v, 20);
    apply(cur, v, v);
  }
}

} // namespace mirt
