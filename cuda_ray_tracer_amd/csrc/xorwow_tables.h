// Lookup tables for the XORWOW sequence skip-ahead (see xorwow_tables.cpp).
#ifndef MIRT_XORWOW_TABLES_H
#define MIRT_XORWOW_TABLES_H
#include <cstdint>
#include <vector>

namespace mirt {

struct RngTables {
  int mode = -1;        // 0: one matrix per sample index (spp > 1); 1: one matrix per (pixel & 255) (spp <= 1)
  int num_mats = 0;
  int chunk_bits = 0;   // 4 or 8
  int nin_words = 0;    // 3 (state words 0,1,4 vary with the seed) or 5
  int nchunks = 0;
  uint32_t d0 = 0;      // mode 1: the Weyl word after seeding (constant seed)
  std::vector<uint32_t> A;   // [num_mats][nchunks][1<<bits][4]  output words 0..3
  std::vector<uint32_t> B;   // [num_mats][nchunks][1<<bits]     output word 4
  std::vector<uint32_t> K;   // [num_mats][5]                    image of the seed-independent words
  std::vector<uint32_t> R2;  // mode 1: [ceil(pixels/256)][5]    state after skipping 256*k subsequences
};

// cuRAND curand_init seeding of the XORWOW state (before any skip-ahead)
void xorwow_seed(uint64_t seed, uint32_t v[5], uint32_t* d);
// spp > 1: sample s of pixel p is curand_init(1234 + p, s, 0)  (draw.cu:162)
void build_sample_tables(int spp, RngTables& t);
// spp <= 1: pixel p is curand_init(seed = 1234, p, 0)          (draw.cu:105)
void build_pixel_tables(int64_t num_pixels, uint64_t seed, RngTables& t);

} // namespace mirt
#endif
