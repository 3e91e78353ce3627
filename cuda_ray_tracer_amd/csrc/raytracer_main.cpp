// raytracer -- the reference's command line (main.cu:25-94) over libmirt's C ABI:
//
//     raytracer scene.txt [--width W] [--height H] [--spp N] [--out file.png] [--device D] [--gpus N] [--frames K]
//                         [--traversal 0|1|2] [--bounds-as-shipped]
//
// Same contract: one positional scene file, the PNG is named by the scene's `png W H name` line and written to the
// current directory, the same lines go to stdout -- the phase timings (main.cu:39,61,71,80,93, with the reference's labels:
// the last one reads "cudaFree time:"), the build lines and, for at most 16 primitives, the "Node Info" dump of every node
// (lbvh_builder.cu:489-520), the "[DEBUG Render]" line of the many-samples kernel (draw.cu:232) -- and the reference's error
// messages + exit codes are kept ("Error opening file...", "One of the lines are not valid.": exit 1; device errors:
// EXIT_FAILURE).  The optional flags override resolution / samples per pixel of the scene file (BASELINE.json's configs do).
// --gpus N renders the frame on N GPUs of this node (image stripes, replicated BVH, RCCL framebuffer gather: mirt_multi_*);
// --frames K renders K frames back to back (consecutive frames overlap on every device) and reports ms per frame.
// --bounds-as-shipped builds the tree of the shipped reference (scene bounds never stored, parse.cpp:28: every Morton code 0).
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

// The reference's debug dump of a small tree (lbvh_builder.cu:496-520), same format.  visited_atomic_counter: the refit leaves 2
// in every internal node (both children arrived, lbvh_builder.cu:343-359); a leaf's counter and child offsets are never written
// by the reference (fresh cudaMalloc memory) and print as 0 here, an internal node's primitive_offset likewise.
static void print_node_info(MirtScene* sc, int n)
{
  std::vector<MirtTreeNode> nodes(2 * (size_t)n - 1);
  if (mirt_get_tree(sc, nodes.data(), nullptr, nullptr, nullptr) != MIRT_OK) return;
  printf("Node Info:\n");
  for (int i = 0; i < 2 * n - 1; ++i) {
    const MirtTreeNode& t = nodes[i];
    printf("  Node %d: num_primitives_in_leaf=%d, primitive_offset=%d, left_child_offset=%d, right_child_offset=%d, visited_atomic_counter=%u, bbox=(min: %.2f, %.2f, %.2f, max: %.2f, %.2f, %.2f)\n",
           i, (int)t.count, (int)t.prim_offset, (int)t.left, (int)t.right, t.count ? 0u : 2u, t.xmin, t.ymin, t.zmin, t.xmax, t.ymax, t.zmax);
  }
}

// draw.cu:222-232: the kernel for aa > 1 announces its launch (one thread per sample, blocks of 128)
static void print_debug_render(int width, int height, int spp)
{
  if (spp <= 1) return;
  const int block_size = 128;
  const int total_threads = width * height * spp;
  printf("[DEBUG Render] Launching AA Kernel. Total Threads: %d, Grid: %d, Block: %d\n", total_threads, (total_threads + block_size - 1) / block_size, block_size);
}

int main(int argc, char* argv[])
{
  if (argc < 2) { std::cout << "Error opening file..." << std::endl; return 1; }
  int ow = 0, oh = 0, ospp = -1, device = 0, gpus = 1, traversal = -1, frames = 1, shipped = 0;
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
    else if (a == "--frames") { need(1); frames = atoi(argv[++i]); if (frames < 1) frames = 1; }
    else if (a == "--bounds-as-shipped") shipped = 1;
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
    float ms_per_frame = 0.0f;
    start = std::chrono::high_resolution_clock::now();
    print_debug_render(width, height, spp);
    // (a capacity overflow on any device fails the call: mirt_multi_wait checks every device's scene)
    die_on(mirt_render_frames_multi(mm, width, height, spp, 4, frames, frames > 1 ? 2 : 1, img.data(), &st, &ms_per_frame), "render");
    end = std::chrono::high_resolution_clock::now();
    elapsed = end - start;
    if (desc.num_prims > 0) {
      printf("LBVH Build time (N=%d): %.3f ms\n", desc.num_prims, st.build_ms);
      printf("LBVH Build (Karas algorithm) complete. Total nodes: %u\n", 2u * (unsigned)desc.num_prims - 1u);
    }
    std::cout << "Render time: " << elapsed.count() << " seconds" << std::endl;
    printf("GPUs: %d, frames: %d (%.3f ms per frame), framebuffer gather: %.3f ms, per-GPU render ms of the last frame:", st.num_gpus, frames, ms_per_frame, st.gather_ms);
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
  if (shipped) die_on(mirt_scene_set_option(sc, "bounds_as_shipped", 1), "mirt_scene_set_option");
  auto end = std::chrono::high_resolution_clock::now();
  std::chrono::duration<double> elapsed = end - start;
  std::cout << "Initialize raw config time: " << elapsed.count() << " seconds" << std::endl;

  if (desc.num_prims > 0) {
    float ms = 0.0f;
    die_on(mirt_build_lbvh(sc, nullptr, &ms), "build_lbvh_karas");
    printf("LBVH Build time (N=%d): %.3f ms\n", desc.num_prims, ms);
    printf("LBVH Build (Karas algorithm) complete. Total nodes: %u\n", 2u * (unsigned)desc.num_prims - 1u);
    if (desc.num_prims <= 16) print_node_info(sc, desc.num_prims);
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
  print_debug_render(width, height, spp);
  for (int f = 0; f < frames; ++f) die_on(mirt_render(sc, &p, d_image, nullptr, nullptr), "render");
  HIP_CHECK(hipDeviceSynchronize());
  {
    MirtStats st;      // (a capacity overflow during the render is an error, not a warning on stdout as bvh_traversal.cu:154-164)
    die_on(mirt_get_stats(sc, &st), "render");
  }
  end = std::chrono::high_resolution_clock::now();
  elapsed = end - start;
  std::cout << "Render time: " << elapsed.count() << " seconds" << std::endl;
  if (frames > 1) printf("frames: %d (%.3f ms per frame)\n", frames, elapsed.count() * 1e3 / frames);

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
  std::cout << "cudaFree time: " << elapsed.count() << " seconds" << std::endl;      // (the reference's label, main.cu:93: the contract is the text)
  return 0;
}
