// Multi-GPU frames behind the C ABI (include/mirt.h, mirt_multi_*): the image-stripe data parallelism of SURVEY.md 8e in one
// process.  The scene is uploaded to every device and every device builds the identical LBVH (deterministic: no broadcast);
// the frame is cut into interleaved row stripes (MirtRenderParams); device r renders part r into a compact buffer; the parts
// are gathered to device 0 with grouped RCCL send/recv over xGMI (<= 1.04 MB per device at 1080p, each peer on its own link)
// and scattered into the row-major frame there.  One host thread drives all devices (every call is asynchronous on its
// device's stream).  The reference is single-GPU (main.cu:25-94); this replaces nothing in it.
//
// RCCL is loaded at run time (dlopen) and only when more than one device is used: libmirt.so has no link-time dependency on
// it, and a process that already holds an RCCL (PyTorch's) shares that one.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mirt.h"
#include "host_scene.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

bool load_rccl(Rccl& r)
{
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.lib) break;
  }
  if (!r.lib) return false;
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
  r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
  r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  return r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetErrorString;
}

int hip_err(hipError_t e, const char* what)
{
  mirt::set_error(std::string("mirt_multi: HIP error in ") + what + ": " + hipGetErrorString(e));
  return MIRT_ERR_HIP;
}
#define MM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_err(e_, #call); } while (0)

} // namespace

struct MirtMulti {
  int n = 0;
  std::vector<int> dev;
  std::vector<MirtScene*> scene;
  std::vector<hipStream_t> stream;
  std::vector<hipEvent_t> ev0, ev1;          // per device: part render start / end
  hipEvent_t gather_end = nullptr;           // device 0
  std::vector<void*> part;                   // per device: its compact part buffer
  std::vector<void*> gathered;               // device 0: where part r arrives (r >= 1)
  void* frame = nullptr;                     // device 0: the row-major frame
  size_t part_cap = 0, frame_cap = 0;
  Rccl rccl;
  std::vector<ncclComm_t> comm;
  float build_ms_max = 0.0f;
};

extern "C" {

int mirt_multi_num_parts(const MirtMulti* mm) { return mm ? mm->n : 0; }

void mirt_multi_destroy(MirtMulti* mm)
{
  if (!mm) return;
  for (int r = 0; r < (int)mm->scene.size(); ++r) {
    hipSetDevice(mm->dev[r]);
    if (r < (int)mm->stream.size() && mm->stream[r]) hipStreamSynchronize(mm->stream[r]);
    if (r < (int)mm->comm.size() && mm->comm[r]) mm->rccl.CommDestroy(mm->comm[r]);
    if (mm->scene[r]) mirt_scene_destroy(mm->scene[r]);
    hipSetDevice(mm->dev[r]);
    if (r < (int)mm->part.size()) hipFree(mm->part[r]);
    if (r < (int)mm->gathered.size()) hipFree(mm->gathered[r]);
    if (r < (int)mm->ev0.size() && mm->ev0[r]) hipEventDestroy(mm->ev0[r]);
    if (r < (int)mm->ev1.size() && mm->ev1[r]) hipEventDestroy(mm->ev1[r]);
    if (r < (int)mm->stream.size() && mm->stream[r]) hipStreamDestroy(mm->stream[r]);
  }
  if (!mm->dev.empty()) { hipSetDevice(mm->dev[0]); hipFree(mm->frame); if (mm->gather_end) hipEventDestroy(mm->gather_end); }
  delete mm;
}

int mirt_multi_create(const MirtSceneDesc* desc, int ngpu, const int* devices, MirtMulti** out)
{
  if (!desc || !out || ngpu < 1) { mirt::set_error("mirt_multi_create: bad argument"); return MIRT_ERR_ARG; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { mirt::set_error("mirt_multi_create: no HIP device available (libmirt has no CPU path)"); return MIRT_ERR_NO_DEVICE; }
  if (ngpu > ndev) { mirt::set_error("mirt_multi_create: more GPUs requested than present"); return MIRT_ERR_ARG; }
  MirtMulti* mm = new MirtMulti();
  mm->n = ngpu;
  for (int r = 0; r < ngpu; ++r) {
    const int d = devices ? devices[r] : r;
    if (d < 0 || d >= ndev) { mirt_multi_destroy(mm); mirt::set_error("mirt_multi_create: bad device index"); return MIRT_ERR_ARG; }
    for (int q = 0; q < r; ++q) if (mm->dev[q] == d) { mirt_multi_destroy(mm); mirt::set_error("mirt_multi_create: a device is listed twice"); return MIRT_ERR_ARG; }
    mm->dev.push_back(d);
  }
  mm->scene.assign(ngpu, nullptr); mm->stream.assign(ngpu, nullptr); mm->ev0.assign(ngpu, nullptr); mm->ev1.assign(ngpu, nullptr);
  mm->part.assign(ngpu, nullptr); mm->gathered.assign(ngpu, nullptr);
  // the BVH is replicated: every device gets the same arrays and builds the same tree
  for (int r = 0; r < ngpu; ++r) {
    int rc = mirt_scene_create(desc, mm->dev[r], &mm->scene[r]);
    if (rc != MIRT_OK) { mirt_multi_destroy(mm); return rc; }
    hipError_t e = hipSetDevice(mm->dev[r]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&mm->stream[r], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&mm->ev0[r]);
    if (e == hipSuccess) e = hipEventCreate(&mm->ev1[r]);
    if (e == hipSuccess && r == 0) e = hipEventCreate(&mm->gather_end);
    if (e != hipSuccess) { mirt_multi_destroy(mm); return hip_err(e, "stream / event creation"); }
  }
  for (int r = 0; r < ngpu; ++r) {
    float ms = 0.0f;
    int rc = mirt_build_lbvh(mm->scene[r], mm->stream[r], &ms);
    if (rc != MIRT_OK) { mirt_multi_destroy(mm); return rc; }
    if (ms > mm->build_ms_max) mm->build_ms_max = ms;
  }
  if (ngpu > 1) {
    if (!load_rccl(mm->rccl)) { mirt_multi_destroy(mm); mirt::set_error("mirt_multi_create: librccl.so not found (needed for more than one GPU)"); return MIRT_ERR_STATE; }
    mm->comm.assign(ngpu, nullptr);
    ncclResult_t nr = mm->rccl.CommInitAll(mm->comm.data(), ngpu, mm->dev.data());
    if (nr != ncclSuccess) {
      std::string msg = std::string("mirt_multi_create: ncclCommInitAll: ") + mm->rccl.GetErrorString(nr);
      mm->comm.clear(); mirt_multi_destroy(mm); mirt::set_error(msg); return MIRT_ERR_HIP;
    }
  }
  *out = mm;
  return MIRT_OK;
}

int mirt_multi_set_option(MirtMulti* mm, const char* name, int value)
{
  if (!mm) { mirt::set_error("mirt_multi_set_option: null argument"); return MIRT_ERR_ARG; }
  for (MirtScene* sc : mm->scene) { int rc = mirt_scene_set_option(sc, name, value); if (rc != MIRT_OK) return rc; }
  return MIRT_OK;
}

int mirt_render_frame_multi(MirtMulti* mm, int width, int height, int spp, int stripe_rows, uint8_t* host_rgba, MirtMultiStats* stats)
{
  if (!mm || width <= 0 || height <= 0 || spp < 0 || stripe_rows <= 0) { mirt::set_error("mirt_render_frame_multi: bad argument"); return MIRT_ERR_ARG; }
  const int n = mm->n;
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<MirtRenderParams> prm(n);
  std::vector<int64_t> npix(n);
  int64_t maxpix = 0;
  for (int r = 0; r < n; ++r) {
    MirtRenderParams& p = prm[r];
    p.width = width; p.height = height; p.spp = spp; p.stripe_rows = n > 1 ? stripe_rows : height; p.num_parts = n; p.part = r; p.flags = 0;
    npix[r] = mirt_render_num_pixels(&p);
    if (npix[r] < 0) { mirt::set_error("mirt_render_frame_multi: bad frame parameters"); return MIRT_ERR_ARG; }
    if (npix[r] > maxpix) maxpix = npix[r];
  }
  // buffers (grown on demand, kept)
  const size_t part_bytes = (size_t)maxpix * 4, frame_bytes = (size_t)width * height * 4;
  if (mm->part_cap < part_bytes) {
    for (int r = 0; r < n; ++r) {
      MM_HIP(hipSetDevice(mm->dev[r]));
      MM_HIP(hipStreamSynchronize(mm->stream[r]));
      hipFree(mm->part[r]); mm->part[r] = nullptr;
      MM_HIP(hipMalloc(&mm->part[r], part_bytes ? part_bytes : 4));
    }
    MM_HIP(hipSetDevice(mm->dev[0]));
    for (int r = 1; r < n; ++r) { hipFree(mm->gathered[r]); mm->gathered[r] = nullptr; MM_HIP(hipMalloc(&mm->gathered[r], part_bytes ? part_bytes : 4)); }
    mm->part_cap = part_bytes;
  }
  if (n > 1 && mm->frame_cap < frame_bytes) {
    MM_HIP(hipSetDevice(mm->dev[0]));
    MM_HIP(hipStreamSynchronize(mm->stream[0]));
    hipFree(mm->frame); mm->frame = nullptr;
    MM_HIP(hipMalloc(&mm->frame, frame_bytes));
    mm->frame_cap = frame_bytes;
  }
  // every device renders its stripes
  for (int r = 0; r < n; ++r) {
    MM_HIP(hipSetDevice(mm->dev[r]));
    MM_HIP(hipEventRecord(mm->ev0[r], mm->stream[r]));
    if (npix[r] > 0) { int rc = mirt_render(mm->scene[r], &prm[r], mm->part[r], nullptr, mm->stream[r]); if (rc != MIRT_OK) return rc; }
    MM_HIP(hipEventRecord(mm->ev1[r], mm->stream[r]));
  }
  const void* result = mm->part[0];          // one device: its part is the row-major frame
  if (n > 1) {
    // framebuffer gather to device 0: one grouped exchange per frame, each peer on its own xGMI link
    ncclResult_t nr = mm->rccl.GroupStart();
    for (int r = 1; r < n && nr == ncclSuccess; ++r) {
      if (npix[r] == 0) continue;
      nr = mm->rccl.Send(mm->part[r], (size_t)npix[r] * 4, ncclUint8, 0, mm->comm[r], mm->stream[r]);
      if (nr == ncclSuccess) nr = mm->rccl.Recv(mm->gathered[r], (size_t)npix[r] * 4, ncclUint8, r, mm->comm[0], mm->stream[0]);
    }
    const ncclResult_t ne = mm->rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) { mirt::set_error(std::string("mirt_render_frame_multi: RCCL: ") + mm->rccl.GetErrorString(nr)); return MIRT_ERR_HIP; }
    MM_HIP(hipSetDevice(mm->dev[0]));
    for (int r = 0; r < n; ++r) {
      if (npix[r] == 0) continue;
      int rc = mirt_scatter_part(&prm[r], r == 0 ? mm->part[0] : mm->gathered[r], mm->frame, mm->stream[0]);
      if (rc != MIRT_OK) return rc;
    }
    result = mm->frame;
  }
  MM_HIP(hipSetDevice(mm->dev[0]));
  MM_HIP(hipEventRecord(mm->gather_end, mm->stream[0]));
  if (host_rgba) MM_HIP(hipMemcpyAsync(host_rgba, result, frame_bytes, hipMemcpyDeviceToHost, mm->stream[0]));
  for (int r = 0; r < n; ++r) { MM_HIP(hipSetDevice(mm->dev[r])); MM_HIP(hipStreamSynchronize(mm->stream[r])); }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->num_gpus = n;
    stats->build_ms = mm->build_ms_max;
    for (int r = 0; r < n && r < MIRT_MULTI_MAX_GPUS; ++r) {
      MM_HIP(hipSetDevice(mm->dev[r]));
      MM_HIP(hipEventElapsedTime(&stats->render_ms[r], mm->ev0[r], mm->ev1[r]));
    }
    MM_HIP(hipSetDevice(mm->dev[0]));
    MM_HIP(hipEventElapsedTime(&stats->gather_ms, mm->ev1[0], mm->gather_end));
    stats->frame_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return MIRT_OK;
}

} // extern "C"
