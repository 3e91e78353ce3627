// Multi-GPU frames behind the C ABI (include/mirt.h, mirt_multi_*): the image-stripe data parallelism of SURVEY.md 8e in one
// process.  The scene is uploaded to every device and every device builds the identical LBVH (deterministic: no broadcast; the
// devices build concurrently, one host thread each); the frame is cut into interleaved row stripes (MirtRenderParams); device
// r renders part r into a compact buffer; the parts are gathered to device 0 with grouped RCCL send/recv over xGMI (<= 1.04 MB
// per device at 1080p, each peer on its own link) and scattered into the row-major frame there.
//
// Frames are pipelined: mirt_multi_submit issues a frame and returns, mirt_multi_wait collects it; up to
// MIRT_MULTI_MAX_IN_FLIGHT frames may be in flight, each on its own set of render streams (a scene keeps as many render
// workspaces), so that the next frame's waves move into the wave slots the draining frame frees -- on a stripe share of a
// frame the drain is a third of the time.  The gathers run in submission order on one communication stream per device (RCCL
// wants the calls on a communicator issued in one order).  One host thread drives all devices; every call is asynchronous on
// its device's streams.  The reference is single-GPU (main.cu:25-94); this replaces nothing in it.
//
// RCCL is loaded at run time (dlopen) and only when more than one device is used: libmirt.so has no link-time dependency on
// it, and a process that already holds an RCCL (PyTorch's) shares that one.  MIRT_MULTI_GATHER=copy gathers with peer-to-peer
// hipMemcpyAsync instead; a device may then be listed more than once (several parts time-sharing one GPU: a rehearsal of the
// N > 1 code path on a one-GPU box, never a scaling measurement).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
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

// everything one frame in flight owns
struct Slot {
  std::vector<hipStream_t> stream;           // per device: this slot's render stream
  std::vector<hipEvent_t> ev0, ev1;          // per device: part render start / end
  std::vector<hipEvent_t> sent;              // per device: its share of the gather has left (communication stream)
  hipEvent_t gather_end = nullptr;           // device 0 (communication stream): frame re-interleaved
  hipEvent_t done = nullptr;                 // device 0: host copy issued too
  std::vector<void*> part;                   // per device: its compact part buffer
  std::vector<void*> gathered;               // device 0: where part r arrives (r >= 1)
  void* frame = nullptr;                     // device 0: the row-major frame
  size_t part_cap = 0, frame_cap = 0;
  bool busy = false;
  unsigned long long ticket = 0;
  std::chrono::steady_clock::time_point t0;
};

} // namespace

struct MirtMulti {
  int n = 0;
  bool copy_gather = false;                  // peer-to-peer copies instead of RCCL (MIRT_MULTI_GATHER=copy)
  std::vector<int> dev;
  std::vector<MirtScene*> scene;
  std::vector<hipStream_t> comm_stream;      // per device: the gathers, in submission order
  Slot slot[MIRT_MULTI_MAX_IN_FLIGHT];
  unsigned long long next_ticket = 1;
  int outstanding = 0;
  Rccl rccl;
  std::vector<ncclComm_t> comm;
  float build_ms_max = 0.0f;
};

namespace {

// wait for everything the slot has in flight (error paths, reuse, destroy)
void drain_slot(MirtMulti* mm, Slot& s)
{
  for (int r = 0; r < mm->n; ++r) {
    hipSetDevice(mm->dev[r]);
    if (r < (int)s.stream.size() && s.stream[r]) hipStreamSynchronize(s.stream[r]);
    if (r < (int)mm->comm_stream.size() && mm->comm_stream[r]) hipStreamSynchronize(mm->comm_stream[r]);
  }
}

// Overflowed capacities on any device (a full pending-ray list: the image misses contributions)?  mirt_get_stats waits for
// every frame in flight of a scene, so this is only called once the pipeline is empty.
int check_overflow(MirtMulti* mm)
{
  int rc = MIRT_OK;
  for (int r = 0; r < mm->n; ++r) {
    MirtStats st;
    const int q = mirt_get_stats(mm->scene[r], &st);
    if (q != MIRT_OK && rc == MIRT_OK) rc = q;      // (the message of the first failing device stays in mirt_last_error)
  }
  return rc;
}

} // namespace

extern "C" {

int mirt_multi_num_parts(const MirtMulti* mm) { return mm ? mm->n : 0; }

void mirt_multi_destroy(MirtMulti* mm)
{
  if (!mm) return;
  for (Slot& s : mm->slot) drain_slot(mm, s);
  for (int r = 0; r < (int)mm->scene.size(); ++r) {
    hipSetDevice(mm->dev[r]);
    if (r < (int)mm->comm.size() && mm->comm[r]) mm->rccl.CommDestroy(mm->comm[r]);
    if (mm->scene[r]) mirt_scene_destroy(mm->scene[r]);
    hipSetDevice(mm->dev[r]);
    for (Slot& s : mm->slot) {
      if (r < (int)s.part.size()) hipFree(s.part[r]);
      if (r < (int)s.gathered.size()) hipFree(s.gathered[r]);
      if (r < (int)s.ev0.size() && s.ev0[r]) hipEventDestroy(s.ev0[r]);
      if (r < (int)s.ev1.size() && s.ev1[r]) hipEventDestroy(s.ev1[r]);
      if (r < (int)s.sent.size() && s.sent[r]) hipEventDestroy(s.sent[r]);
      if (r < (int)s.stream.size() && s.stream[r]) hipStreamDestroy(s.stream[r]);
    }
    if (r < (int)mm->comm_stream.size() && mm->comm_stream[r]) hipStreamDestroy(mm->comm_stream[r]);
  }
  if (!mm->dev.empty()) {
    hipSetDevice(mm->dev[0]);
    for (Slot& s : mm->slot) { hipFree(s.frame); if (s.gather_end) hipEventDestroy(s.gather_end); if (s.done) hipEventDestroy(s.done); }
  }
  delete mm;
}

int mirt_multi_create(const MirtSceneDesc* desc, int ngpu, const int* devices, MirtMulti** out)
{
  if (!desc || !out || ngpu < 1 || ngpu > MIRT_MULTI_MAX_GPUS) { mirt::set_error("mirt_multi_create: bad argument"); return MIRT_ERR_ARG; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { mirt::set_error("mirt_multi_create: no HIP device available (libmirt has no CPU path)"); return MIRT_ERR_NO_DEVICE; }
  const char* g = getenv("MIRT_MULTI_GATHER");
  const bool copy_gather = g && strcmp(g, "copy") == 0;
  if (ngpu > ndev && !copy_gather) { mirt::set_error("mirt_multi_create: more GPUs requested than present"); return MIRT_ERR_ARG; }
  MirtMulti* mm = new MirtMulti();
  mm->n = ngpu;
  mm->copy_gather = copy_gather;
  for (int r = 0; r < ngpu; ++r) {
    const int d = devices ? devices[r] : (copy_gather ? r % ndev : r);      // (rehearsal: more parts than GPUs time-share them)
    if (d < 0 || d >= ndev) { mirt_multi_destroy(mm); mirt::set_error("mirt_multi_create: bad device index"); return MIRT_ERR_ARG; }
    // (RCCL refuses a device listed twice; with peer copies several parts may time-share one GPU: a rehearsal, see above)
    if (!copy_gather) for (int q = 0; q < r; ++q) if (mm->dev[q] == d) { mirt_multi_destroy(mm); mirt::set_error("mirt_multi_create: a device is listed twice"); return MIRT_ERR_ARG; }
    mm->dev.push_back(d);
  }
  mm->scene.assign(ngpu, nullptr); mm->comm_stream.assign(ngpu, nullptr);
  for (Slot& s : mm->slot) {
    s.stream.assign(ngpu, nullptr); s.ev0.assign(ngpu, nullptr); s.ev1.assign(ngpu, nullptr); s.sent.assign(ngpu, nullptr);
    s.part.assign(ngpu, nullptr); s.gathered.assign(ngpu, nullptr);
  }
  // The BVH is replicated: every device gets the same arrays and builds the same tree -- all devices at once, one host thread
  // each (upload and build are synchronous calls, lbvh_builder.cu:475).
  std::vector<int> rcs(ngpu, MIRT_OK);
  std::vector<std::string> msgs(ngpu);
  std::vector<float> bms(ngpu, 0.0f);
  {
    std::vector<std::thread> th;
    for (int r = 0; r < ngpu; ++r) {
      th.emplace_back([&, r]() {
        int rc = mirt_scene_create(desc, mm->dev[r], &mm->scene[r]);
        hipError_t e = hipSuccess;
        if (rc == MIRT_OK) {
          e = hipSetDevice(mm->dev[r]);
          if (e == hipSuccess) e = hipStreamCreateWithFlags(&mm->comm_stream[r], hipStreamNonBlocking);
          for (Slot& s : mm->slot) {
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream[r], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreate(&s.ev0[r]);
            if (e == hipSuccess) e = hipEventCreate(&s.ev1[r]);
            if (e == hipSuccess) e = hipEventCreate(&s.sent[r]);
            if (e == hipSuccess && r == 0) e = hipEventCreate(&s.gather_end);
            if (e == hipSuccess && r == 0) e = hipEventCreate(&s.done);
          }
          if (e != hipSuccess) rc = hip_err(e, "stream / event creation");
        }
        if (rc == MIRT_OK) rc = mirt_build_lbvh(mm->scene[r], mm->slot[0].stream[r], &bms[r]);
        rcs[r] = rc;
        if (rc != MIRT_OK) msgs[r] = mirt_last_error();      // (the error string is per thread)
      });
    }
    for (std::thread& t : th) t.join();
  }
  for (int r = 0; r < ngpu; ++r) {
    if (rcs[r] != MIRT_OK) { const int rc = rcs[r]; const std::string m = msgs[r]; mirt_multi_destroy(mm); mirt::set_error(m); return rc; }
    if (bms[r] > mm->build_ms_max) mm->build_ms_max = bms[r];
  }
  if (ngpu > 1 && !copy_gather) {
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

int mirt_multi_get_stats(MirtMulti* mm, int part, MirtStats* out)
{
  if (!mm || part < 0 || part >= mm->n || !out) { mirt::set_error("mirt_multi_get_stats: bad argument"); return MIRT_ERR_ARG; }
  return mirt_get_stats(mm->scene[part], out);
}

static int submit_impl(MirtMulti* mm, Slot& s, int width, int height, int spp, int stripe_rows, uint8_t* host_rgba)
{
  const int n = mm->n;
  std::vector<MirtRenderParams> prm(n);
  std::vector<int64_t> npix(n);
  int64_t maxpix = 0;
  for (int r = 0; r < n; ++r) {
    MirtRenderParams& p = prm[r];
    p.width = width; p.height = height; p.spp = spp; p.stripe_rows = n > 1 ? stripe_rows : height; p.num_parts = n; p.part = r; p.flags = 0;
    npix[r] = mirt_render_num_pixels(&p);
    if (npix[r] < 0) { mirt::set_error("mirt_multi_submit: bad frame parameters"); return MIRT_ERR_ARG; }
    if (npix[r] > maxpix) maxpix = npix[r];
  }
  // buffers of this slot (grown on demand, kept; the slot is idle here)
  const size_t part_bytes = (size_t)maxpix * 4, frame_bytes = (size_t)width * height * 4;
  if (s.part_cap < part_bytes) {
    for (int r = 0; r < n; ++r) {
      MM_HIP(hipSetDevice(mm->dev[r]));
      hipFree(s.part[r]); s.part[r] = nullptr;
      MM_HIP(hipMalloc(&s.part[r], part_bytes ? part_bytes : 4));
    }
    MM_HIP(hipSetDevice(mm->dev[0]));
    for (int r = 1; r < n; ++r) { hipFree(s.gathered[r]); s.gathered[r] = nullptr; MM_HIP(hipMalloc(&s.gathered[r], part_bytes ? part_bytes : 4)); }
    s.part_cap = part_bytes;
  }
  if (n > 1 && s.frame_cap < frame_bytes) {
    MM_HIP(hipSetDevice(mm->dev[0]));
    hipFree(s.frame); s.frame = nullptr;
    MM_HIP(hipMalloc(&s.frame, frame_bytes));
    s.frame_cap = frame_bytes;
  }
  // every device renders its stripes on this slot's stream
  for (int r = 0; r < n; ++r) {
    MM_HIP(hipSetDevice(mm->dev[r]));
    MM_HIP(hipEventRecord(s.ev0[r], s.stream[r]));
    if (npix[r] > 0) { int rc = mirt_render(mm->scene[r], &prm[r], s.part[r], nullptr, s.stream[r]); if (rc != MIRT_OK) return rc; }
    MM_HIP(hipEventRecord(s.ev1[r], s.stream[r]));
    MM_HIP(hipStreamWaitEvent(mm->comm_stream[r], s.ev1[r], 0));      // the gather of this frame follows its part, and the gathers before it
  }
  const void* result = s.part[0];          // one device: its part is the row-major frame
  if (n > 1) {
    if (mm->copy_gather) {
      // peer-to-peer copies issued on the sender's communication stream; device 0 waits for each
      for (int r = 1; r < n; ++r) {
        if (npix[r] == 0) continue;
        MM_HIP(hipSetDevice(mm->dev[r]));
        MM_HIP(hipMemcpyPeerAsync(s.gathered[r], mm->dev[0], s.part[r], mm->dev[r], (size_t)npix[r] * 4, mm->comm_stream[r]));
        MM_HIP(hipEventRecord(s.ev1[r], mm->comm_stream[r]));
        MM_HIP(hipSetDevice(mm->dev[0]));
        MM_HIP(hipStreamWaitEvent(mm->comm_stream[0], s.ev1[r], 0));
      }
    } else {
      // framebuffer gather to device 0: one grouped exchange per frame, each peer on its own xGMI link
      ncclResult_t nr = mm->rccl.GroupStart();
      for (int r = 1; r < n && nr == ncclSuccess; ++r) {
        if (npix[r] == 0) continue;
        nr = mm->rccl.Send(s.part[r], (size_t)npix[r] * 4, ncclUint8, 0, mm->comm[r], mm->comm_stream[r]);
        if (nr == ncclSuccess) nr = mm->rccl.Recv(s.gathered[r], (size_t)npix[r] * 4, ncclUint8, r, mm->comm[0], mm->comm_stream[0]);
      }
      const ncclResult_t ne = mm->rccl.GroupEnd();
      if (nr == ncclSuccess) nr = ne;
      if (nr != ncclSuccess) { mirt::set_error(std::string("mirt_multi_submit: RCCL: ") + mm->rccl.GetErrorString(nr)); return MIRT_ERR_HIP; }
    }
    for (int r = 1; r < n; ++r) { MM_HIP(hipSetDevice(mm->dev[r])); MM_HIP(hipEventRecord(s.sent[r], mm->comm_stream[r])); }
    MM_HIP(hipSetDevice(mm->dev[0]));
    for (int r = 0; r < n; ++r) {
      if (npix[r] == 0) continue;
      int rc = mirt_scatter_part(&prm[r], r == 0 ? s.part[0] : s.gathered[r], s.frame, mm->comm_stream[0]);
      if (rc != MIRT_OK) return rc;
    }
    result = s.frame;
  }
  MM_HIP(hipSetDevice(mm->dev[0]));
  MM_HIP(hipEventRecord(s.gather_end, mm->comm_stream[0]));
  if (host_rgba) MM_HIP(hipMemcpyAsync(host_rgba, result, frame_bytes, hipMemcpyDeviceToHost, mm->comm_stream[0]));
  MM_HIP(hipEventRecord(s.done, mm->comm_stream[0]));
  return MIRT_OK;
}

int mirt_multi_submit(MirtMulti* mm, int width, int height, int spp, int stripe_rows, uint8_t* host_rgba, uint64_t* ticket)
{
  if (!mm || width <= 0 || height <= 0 || spp < 0 || stripe_rows <= 0 || !ticket) { mirt::set_error("mirt_multi_submit: bad argument"); return MIRT_ERR_ARG; }
  Slot& s = mm->slot[mm->next_ticket % MIRT_MULTI_MAX_IN_FLIGHT];
  if (s.busy) { mirt::set_error("mirt_multi_submit: MIRT_MULTI_MAX_IN_FLIGHT frames are in flight: mirt_multi_wait for the oldest first"); return MIRT_ERR_STATE; }
  s.t0 = std::chrono::steady_clock::now();
  const int rc = submit_impl(mm, s, width, height, spp, stripe_rows, host_rgba);
  if (rc != MIRT_OK) {
    // nothing of a half-issued frame stays in flight behind the caller's back
    const std::string msg = mirt_last_error();
    drain_slot(mm, s);
    mirt::set_error(msg);
    return rc;
  }
  s.busy = true;
  s.ticket = mm->next_ticket++;
  ++mm->outstanding;
  *ticket = s.ticket;
  return MIRT_OK;
}

int mirt_multi_wait(MirtMulti* mm, uint64_t ticket, MirtMultiStats* stats)
{
  if (!mm) { mirt::set_error("mirt_multi_wait: null argument"); return MIRT_ERR_ARG; }
  Slot& s = mm->slot[ticket % MIRT_MULTI_MAX_IN_FLIGHT];
  if (!s.busy || s.ticket != ticket) { mirt::set_error("mirt_multi_wait: no such frame in flight"); return MIRT_ERR_ARG; }
  const int n = mm->n;
  for (int r = 0; r < n; ++r) {
    MM_HIP(hipSetDevice(mm->dev[r]));
    MM_HIP(hipEventSynchronize(s.ev1[r]));
    if (r > 0 && n > 1) MM_HIP(hipEventSynchronize(s.sent[r]));      // its part buffer is free again
  }
  MM_HIP(hipSetDevice(mm->dev[0]));
  MM_HIP(hipEventSynchronize(s.done));
  s.busy = false;
  --mm->outstanding;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->num_gpus = n;
    stats->build_ms = mm->build_ms_max;
    if (!mm->copy_gather || n == 1) {      // (with peer copies ev1[r >= 1] was re-recorded after the copy)
      for (int r = 0; r < n && r < MIRT_MULTI_MAX_GPUS; ++r) {
        MM_HIP(hipSetDevice(mm->dev[r]));
        MM_HIP(hipEventElapsedTime(&stats->render_ms[r], s.ev0[r], s.ev1[r]));
      }
    } else {
      MM_HIP(hipSetDevice(mm->dev[0]));
      MM_HIP(hipEventElapsedTime(&stats->render_ms[0], s.ev0[0], s.ev1[0]));
    }
    MM_HIP(hipSetDevice(mm->dev[0]));
    MM_HIP(hipEventElapsedTime(&stats->gather_ms, s.ev1[0], s.gather_end));
    stats->frame_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - s.t0).count();
  }
  // a capacity overflow on any device makes the frame untrustworthy: checked whenever the pipeline is empty
  if (mm->outstanding == 0) return check_overflow(mm);
  return MIRT_OK;
}

int mirt_render_frame_multi(MirtMulti* mm, int width, int height, int spp, int stripe_rows, uint8_t* host_rgba, MirtMultiStats* stats)
{
  uint64_t t = 0;
  int rc = mirt_multi_submit(mm, width, height, spp, stripe_rows, host_rgba, &t);
  if (rc != MIRT_OK) return rc;
  return mirt_multi_wait(mm, t, stats);
}

int mirt_render_frames_multi(MirtMulti* mm, int width, int height, int spp, int stripe_rows, int nframes, int in_flight, uint8_t* host_rgba_last,
                             MirtMultiStats* last_stats, float* ms_per_frame)
{
  if (!mm || nframes < 1 || in_flight < 1 || in_flight > MIRT_MULTI_MAX_IN_FLIGHT) { mirt::set_error("mirt_render_frames_multi: bad argument"); return MIRT_ERR_ARG; }
  if (mm->outstanding != 0) { mirt::set_error("mirt_render_frames_multi: frames of an earlier mirt_multi_submit are still in flight"); return MIRT_ERR_STATE; }
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<uint64_t> tk(nframes, 0);
  int rc = MIRT_OK;
  int issued = 0, waited = 0;
  for (; issued < nframes && rc == MIRT_OK; ++issued) {
    if (issued - waited >= in_flight) { rc = mirt_multi_wait(mm, tk[waited], nullptr); ++waited; if (rc != MIRT_OK) break; }
    rc = mirt_multi_submit(mm, width, height, spp, stripe_rows, issued == nframes - 1 ? host_rgba_last : nullptr, &tk[issued]);
    if (rc != MIRT_OK) break;
  }
  const std::string msg = rc != MIRT_OK ? std::string(mirt_last_error()) : std::string();
  for (; waited < issued; ++waited) {      // drain, whatever happened
    const int q = mirt_multi_wait(mm, tk[waited], waited == nframes - 1 ? last_stats : nullptr);
    if (q != MIRT_OK && rc == MIRT_OK) rc = q;
  }
  if (!msg.empty()) mirt::set_error(msg);
  if (ms_per_frame) *ms_per_frame = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() / (float)nframes;
  return rc;
}

} // extern "C"
