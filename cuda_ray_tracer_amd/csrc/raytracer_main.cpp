// raytracer -- the reference's command line (main.cu:25-94) over libmirt's C ABI:
//
//     raytracer scene.txt [--width W] [--height H] [--spp N] [--out file.png] [--device D] [--gpus N] [--traversal 0|1|2]
//
// Same contract: one positional scene file, the PNG is named by the scene's `png W H name` line and written to the
// current directory, the same phase timing lines go to stdout, and the reference's error messages + exit codes are
// kept ("Error opening file...", "One of the lines are not valid.": exit 1; device errors: EXIT_FAILURE).
// The optional flags override resolution / samples per pixel of the scene file (BASELINE.json's configs do).  --gpus N
// renders the frame on N GPUs of this node (image stripes, replicated BVH, RCCL framebuffer gather: mirt_multi_*).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/mirt.h"

static void die_on(int rc, const char* what)
{
  if (rc == MIRT_OK) return;
  if (rc == MIRT_ERR_IO || rc == MIRT_ERR_PARSE) { std::cout << mirt_last_error() << std::endl; exit(1); }
  std::cerr << "Error in " << what << " : " << mirt_last_error() << std::endl;
  exit(EXIT_FAILURE);
}
#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::cerr << "HIP Error in " << __FILE__ << " at line " << __LINE__ << " : " << hipGetErrorString(e_) << std::endl; exit(EXIT_FAILURE); } } while (0)

int main(int argc, char* argv[])
{
  if (argc < 2) { std::cout << "Error opening file..." << std::endl; return 1; }
  int ow = 0, oh = 0, ospp = -1, device = 0, gpus = 1, traversal = -1;
  std::string out_override;
  for (int i = 2; i < argc; ++i) {
    std::string a = argv[i];
    auto need = [&](int k) { if (i + k >= argc) { std::cerr << "missing value for " << a << std::endl; exit(2); } };
    if (a == "--width") { need(1); ow = atoi(argv[++i]); }
    else if (a == "--height") { need(1); oh = atoi(argv[++i]); }
    else if (a == "--spp") { need(1); ospp = atoi(argv[++i]); }
    else if (a == "--out") { need(1); out_override = argv[++i]; }
    else if (a == "--device") { need(1); device = atoi(argv[++i]); }
    else if (a == "--gpus") { need(1); gpus = atoi(argv[++i]); }
    else if (a == "--traversal") { need(1); traversal = atoi(argv[++i]); }
    else { std::cerr << "unknown option " << a << std::endl; return 2; }
  }

  MirtHostScene* hs = nullptr;
  die_on(mirt_parse_scene_file(argv[1], &hs), "parseInput");
  MirtSceneDesc desc;
  die_on(mirt_host_scene_desc(hs, &desc), "mirt_host_scene_desc");
  const int width = ow > 0 ? ow : desc.width, height = oh > 0 ? oh : desc.height;
  const int spp = ospp >= 0 ? ospp : desc.aa;

  if (gpus > 1) {
    // several GPUs: the same phases, each over all devices
    auto start = std::chrono::high_resolution_clock::now();
    MirtMulti* mm = nullptr;
    die_on(mirt_multi_create(&desc, gpus, nullptr, &mm), "copyConfigDataToDevice");
    if (traversal >= 0) die_on(mirt_multi_set_option(mm, "traversal", traversal), "mirt_multi_set_option");
    auto end = std::chrono::high_resolution_clock::now();
    std::chrono::duration<double> elapsed = end - start;
    std::cout << "Initialize raw config time: " << elapsed.count() << " seconds" << std::endl;
    std::vector<uint8_t> img((size_t)width * height * 4);
    MirtMultiStats st;
    start = std::chrono::high_resolution_clock::now();
    die_on(mirt_render_frame_multi(mm, width, height, spp, 4, img.data(), &st), "render");
    end = std::chrono::high_resolution_clock::now();
    elapsed = end - start;
    if (desc.num_prims > 0) {
      printf("LBVH Build time (N=%d): %.3f ms\n", desc.num_prims, st.build_ms);
      printf("LBVH Build (Karas algorithm) complete. Total nodes: %u\n", 2u * (unsigned)desc.num_prims - 1u);
    }
    std::cout << "Render time: " << elapsed.count() << " seconds" << std::endl;
    printf("GPUs: %d, framebuffer gather: %.3f ms, per-GPU render ms:", st.num_gpus, st.gather_ms);
    for (int r = 0; r < st.num_gpus && r < MIRT_MULTI_MAX_GPUS; ++r) printf(" %.3f", st.render_ms[r]);
    printf("\n");
    const std::string out = out_override.empty() ? std::string(mirt_host_scene_filename(hs)) : out_override;
    die_on(mirt_write_png(out.c_str(), img.data(), width, height), "Image::save");
    mirt_multi_destroy(mm);
    mirt_host_scene_destroy(hs);
    return 0;
  }

  auto start = std::chrono::high_resolution_clock::now();
  MirtScene* sc = nullptr;
  die_on(mirt_scene_create(&desc, device, &sc), "copyConfigDataToDevice");
  if (traversal >= 0) die_on(mirt_scene_set_option(sc, "traversal", traversal), "mirt_scene_set_option");
  auto end = std::chrono::high_resolution_clock::now();
  std::chrono::duration<double> elapsed = end - start;
  std::cout << "Initialize raw config time: " << elapsed.count() << " seconds" << std::endl;

  if (desc.num_prims > 0) {
    float ms = 0.0f;
    die_on(mirt_build_lbvh(sc, nullptr, &ms), "build_lbvh_karas");
    printf("LBVH Build time (N=%d): %.3f ms\n", desc.num_prims, ms);
    printf("LBVH Build (Karas algorithm) complete. Total nodes: %u\n", 2u * (unsigned)desc.num_prims - 1u);
  } else {
    die_on(mirt_build_lbvh(sc, nullptr, nullptr), "build_lbvh_karas");
  }

  start = std::chrono::high_resolution_clock::now();
  MirtRenderParams p;
  p.width = width; p.height = height; p.spp = spp; p.stripe_rows = height; p.num_parts = 1; p.part = 0; p.flags = 0;
  const size_t bytes = (size_t)width * height * 4;
  void* d_image = nullptr;
  HIP_CHECK(hipSetDevice(device));
  HIP_CHECK(hipMalloc(&d_image, bytes));
  end = std::chrono::high_resolution_clock::now();
  elapsed = end - start;
  std::cout << "Malloc and transfer to device time: " << elapsed.count() << " seconds" << std::endl;

  start = std::chrono::high_resolution_clock::now();
  die_on(mirt_render(sc, &p, d_image, nullptr, nullptr), "render");
  HIP_CHECK(hipDeviceSynchronize());
  {
    MirtStats st;      // (a capacity overflow during the render is an error, not a warning on stdout as bvh_traversal.cu:154-164)
    die_on(mirt_get_stats(sc, &st), "render");
  }
  end = std::chrono::high_resolution_clock::now();
  elapsed = end - start;
  std::cout << "Render time: " << elapsed.count() << " seconds" << std::endl;

  start = std::chrono::high_resolution_clock::now();
  std::vector<uint8_t> img(bytes);
  HIP_CHECK(hipMemcpy(img.data(), d_image, bytes, hipMemcpyDeviceToHost));
  end = std::chrono::high_resolution_clock::now();
  elapsed = end - start;
  std::cout << "Transfer to host time: " << elapsed.count() << " seconds" << std::endl;

  const std::string out = out_override.empty() ? std::string(mirt_host_scene_filename(hs)) : out_override;
  die_on(mirt_write_png(out.c_str(), img.data(), width, height), "Image::save");

  start = std::chrono::high_resolution_clock::now();
  HIP_CHECK(hipFree(d_image));
  mirt_scene_destroy(sc);
  mirt_host_scene_destroy(hs);
  end = std::chrono::high_resolution_clock::now();
  elapsed = end - start;
  std::cout << "hipFree time: " << elapsed.count() << " seconds" << std::endl;
  return 0;
}
