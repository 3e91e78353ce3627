// C ABI glue for the device side of libmirt.so (include/mirt.h): scene upload, build, render, stats.
#include "scene_dev.h"
#include "host_scene.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace mirt;

namespace {

template <typename T>
int upload(T** dst, const std::vector<T>& src)
{
  *dst = nullptr;
  if (src.empty()) return MIRT_OK;
  MIRT_HIP(hipMalloc(dst, sizeof(T) * src.size()));
  MIRT_HIP(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
  return MIRT_OK;
}

bool nonzero(const MirtRGB& c) { return !(fabsf(c.r) < 1e-6f && fabsf(c.g) < 1e-6f && fabsf(c.b) < 1e-6f); }
bool finite3(const MirtRGB& c) { return std::isfinite(c.r) && std::isfinite(c.g) && std::isfinite(c.b); }

void pack_mat(const MirtMaterials& m, float4* out)
{
  out[0] = make_float4(m.color.r, m.color.g, m.color.b, m.shininess.r);
  out[1] = make_float4(m.shininess.g, m.shininess.b, m.trans.r, m.trans.g);
  out[2] = make_float4(m.trans.b, m.ior, m.roughness, 0.0f);
}

// mode: the option selects what is computed or visited (another tree, another walk, another render path) -- as opposed to a
// tuning value, which only moves time
struct OptionDesc { const char* name; int Options::*field; int lo, hi; bool mode; };
const OptionDesc OPTIONS[] = {
  {"bounds_as_shipped", &Options::bounds_as_shipped, 0, 1, true},
  {"traversal", &Options::traversal, 0, 2, true}, {"wavefront", &Options::wavefront, 0, 1, true},
  {"qnodes", &Options::qnodes, 0, 2, true}, {"shadow_anyhit", &Options::shadow_anyhit, 0, 1, true}, {"skip_unlit", &Options::skip_unlit, 0, 1, true},
  {"stack_lds_depth", &Options::stack_lds_depth, -1, 64, false}, {"refill_k", &Options::refill_k, 0, 64, false}, {"batch_k", &Options::batch_k, 1, 64, false},
  {"leaf_k", &Options::leaf_k, 0, 64, false}, {"init_k", &Options::init_k, 0, 64, false}, {"reps", &Options::reps, 0, 8, false}, {"drain_lanes", &Options::drain_lanes, 0, 64, false},
  {"chunk_shift", &Options::chunk_shift, 0, 12, false}, {"trace_waves", &Options::trace_waves, 0, 1 << 20, false}, {"sched", &Options::sched, 0, 2, false},
  {"specialise", &Options::specialise, 0, 1, false}, {"slab_log2", &Options::slab_log2, 8, 30, false},
  {"wf_pool", &Options::wf_pool, 256, 1 << 24, false}, {"wf_refill_k", &Options::wf_refill_k, 1, 64, false},
};

// MIRT_<NAME> (upper case) overrides an option's default; read once per scene, here.  Only the tuning values: a mode switch
// (traversal = 2 changes pixels on triangle silhouettes) is set through mirt_scene_set_option, or from the environment when
// MIRT_ALLOW_ENV=1 says that is meant (tools/ sweeps) -- a variable left over in a shell must not change an image.
void options_from_env(Options& o)
{
  const char* allow = getenv("MIRT_ALLOW_ENV");
  const bool modes = allow && atoi(allow) != 0;
  for (const OptionDesc& d : OPTIONS) {
    if (d.mode && !modes) continue;
    std::string env = "MIRT_";
    for (const char* c = d.name; *c; ++c) env += (char)toupper(*c);
    if (const char* e = getenv(env.c_str())) { const long v = atol(e); if (v >= d.lo && v <= d.hi) o.*(d.field) = (int)v; }
  }
}

int scene_create(const MirtSceneDesc* d, int device, MirtScene** out);

// mirt_get_stats: take a context's overflow count and zero it in one atomic step (a frame issued meanwhile from another host
// thread keeps every increment: it lands either in this reading or in the next)
__global__ void take_overflow_kernel(unsigned long long* counters)
{
  counters[10] = atomicExch(&counters[9], 0ull);
}

} // namespace

extern "C" {

int mirt_scene_set_option(MirtScene* sc, const char* name, int value)
{
  if (!sc || !name) { set_error("mirt_scene_set_option: null argument"); return MIRT_ERR_ARG; }
  for (const OptionDesc& d : OPTIONS)
    if (strcmp(d.name, name) == 0) {
      if (value < d.lo || value > d.hi) { set_error(std::string("mirt_scene_set_option: value out of range for ") + name); return MIRT_ERR_ARG; }
      sc->opt.*(d.field) = value;
      return MIRT_OK;
    }
  set_error(std::string("mirt_scene_set_option: unknown option ") + name);
  return MIRT_ERR_ARG;
}

int mirt_scene_get_option(const MirtScene* sc, const char* name, int* value)
{
  if (!sc || !name || !value) { set_error("mirt_scene_get_option: null argument"); return MIRT_ERR_ARG; }
  for (const OptionDesc& d : OPTIONS)
    if (strcmp(d.name, name) == 0) { *value = sc->opt.*(d.field); return MIRT_OK; }
  set_error(std::string("mirt_scene_get_option: unknown option ") + name);
  return MIRT_ERR_ARG;
}

int mirt_scene_create(const MirtSceneDesc* d, int device, MirtScene** out)
{
  // nothing may unwind across the C boundary (std::bad_alloc from the host-side staging vectors of a huge scene)
  try {
    return scene_create(d, device, out);
  } catch (const std::bad_alloc&) {
    set_error("mirt_scene_create: out of host memory");
    return MIRT_ERR_ARG;
  }
}

} // extern "C"

namespace {

int scene_create(const MirtSceneDesc* d, int device, MirtScene** out)
{
  if (!d || !out) { set_error("mirt_scene_create: null argument"); return MIRT_ERR_ARG; }
  *out = nullptr;
  if (d->num_spheres < 0 || d->num_triangles < 0 || d->num_prims < 0 || d->num_planes < 0 || d->num_suns < 0 || d->num_bulbs < 0) {
    set_error("mirt_scene_create: negative count"); return MIRT_ERR_ARG;
  }
  if ((d->num_spheres > 0 && !d->spheres) || (d->num_triangles > 0 && !d->triangles) || (d->num_prims > 0 && !d->prim_refs) ||
      (d->num_planes > 0 && !d->planes) || (d->num_suns > 0 && !d->suns) || (d->num_bulbs > 0 && !d->bulbs)) {
    set_error("mirt_scene_create: a count is positive but its array is null"); return MIRT_ERR_ARG;
  }
  if ((long long)d->num_spheres + d->num_triangles != d->num_prims) { set_error("mirt_scene_create: num_prims != num_spheres + num_triangles"); return MIRT_ERR_ARG; }
  if (d->num_suns + d->num_bulbs > 64) { set_error("mirt_scene_create: more than 64 lights are not supported"); return MIRT_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("mirt_scene_create: no HIP device available (libmirt has no CPU path)");
    return MIRT_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) { set_error("mirt_scene_create: bad device index"); return MIRT_ERR_ARG; }
  MIRT_HIP(hipSetDevice(device));

  MirtScene* sc = new MirtScene();
  sc->device = device;
  options_from_env(sc->opt);
  sc->d = *d;
  sc->N = d->num_prims; sc->Ns = d->num_spheres; sc->Nt = d->num_triangles;
  const int N = sc->N;

  // de-interleave the AoS inputs into the device layout (config_utils.cu:72-199 does the same job for the reference)
  std::vector<float4> spheres((size_t)sc->Ns), tris(3 * (size_t)sc->Nt), verts(3 * (size_t)sc->Nt), mats(3 * (size_t)N);
  for (int i = 0; i < sc->Ns; ++i) {
    const MirtSphere& s = d->spheres[i];
    spheres[i] = make_float4(s.c.x, s.c.y, s.c.z, s.r);
    pack_mat(s.mat, &mats[3 * (size_t)i]);
    if (nonzero(s.mat.trans)) sc->any_trans = true;
    if (s.mat.roughness > 0.0f) sc->any_rough = true;
    if (!finite3(s.mat.color)) sc->colors_finite = false;
  }
  for (int i = 0; i < sc->Nt; ++i) {
    const MirtTriangle& t = d->triangles[i];
    tris[3 * (size_t)i + 0] = make_float4(t.p0.x, t.p0.y, t.p0.z, t.nor.x);
    tris[3 * (size_t)i + 1] = make_float4(t.nor.y, t.nor.z, t.e1.x, t.e1.y);
    tris[3 * (size_t)i + 2] = make_float4(t.e1.z, t.e2.x, t.e2.y, t.e2.z);
    verts[3 * (size_t)i + 0] = make_float4(t.p0.x, t.p0.y, t.p0.z, 0.0f);
    verts[3 * (size_t)i + 1] = make_float4(t.p1.x, t.p1.y, t.p1.z, 0.0f);
    verts[3 * (size_t)i + 2] = make_float4(t.p2.x, t.p2.y, t.p2.z, 0.0f);
    pack_mat(t.mat, &mats[3 * ((size_t)sc->Ns + i)]);
    if (nonzero(t.mat.trans)) sc->any_trans = true;
    if (t.mat.roughness > 0.0f) sc->any_rough = true;
    if (!finite3(t.mat.color)) sc->colors_finite = false;
  }
  std::vector<MirtPrimRef> refs(d->prim_refs, d->prim_refs + N);
  for (int i = 0; i < N; ++i) {
    const MirtPrimRef& r = refs[i];
    if (r.type > 1 || (r.type == 0 && (int)r.id >= sc->Ns) || (r.type == 1 && (int)r.id >= sc->Nt)) {
      delete sc; set_error("mirt_scene_create: primitive reference out of range"); return MIRT_ERR_ARG;
    }
  }
  std::vector<PlaneDev> planes((size_t)d->num_planes);
  for (int i = 0; i < d->num_planes; ++i) {
    const MirtPlane& p = d->planes[i];
    PlaneDev& q = planes[i];
    q.nx = p.nor.x; q.ny = p.nor.y; q.nz = p.nor.z; q.px = p.point.x; q.py = p.point.y; q.pz = p.point.z;
    const float m[11] = {p.mat.color.r, p.mat.color.g, p.mat.color.b, p.mat.shininess.r, p.mat.shininess.g, p.mat.shininess.b,
                         p.mat.trans.r, p.mat.trans.g, p.mat.trans.b, p.mat.ior, p.mat.roughness};
    memcpy(q.mat, m, sizeof(m)); q.pad = 0.0f;
    if (nonzero(p.mat.trans)) sc->any_trans = true;
    if (p.mat.roughness > 0.0f) sc->any_rough = true;
    if (!finite3(p.mat.color)) sc->colors_finite = false;
  }
  for (int i = 0; i < d->num_suns; ++i) if (!finite3(d->suns[i].color)) sc->colors_finite = false;
  for (int i = 0; i < d->num_bulbs; ++i) if (!finite3(d->bulbs[i].color)) sc->colors_finite = false;
  if (!std::isfinite(d->expose) && d->expose != INFINITY) sc->colors_finite = false;
  std::vector<LightDev> suns((size_t)d->num_suns), bulbs((size_t)d->num_bulbs);
  for (int i = 0; i < d->num_suns; ++i) {
    const MirtVec3& v = d->suns[i].dir;
    // vec3::normalize (vec3.cuh:72-82) -- this translation unit is built with -ffp-contract=off
    const float mag = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    float nx = 0.0f, ny = 0.0f, nz = 0.0f;
    const float diff = fabsf(mag - 0.0f), largest = fmaxf(fabsf(mag), fabsf(0.0f));
    const bool zero = (largest < 1e-6f) ? (diff < 1e-6f) : (diff / largest < 1e-6f);
    if (!zero) { const float inv = 1.0f / mag; nx = v.x * inv; ny = v.y * inv; nz = v.z * inv; }
    suns[i] = {v.x, v.y, v.z, d->suns[i].color.r, d->suns[i].color.g, d->suns[i].color.b, nx, ny, nz, 1.0f / nx, 1.0f / ny, 1.0f / nz};
  }
  for (int i = 0; i < d->num_bulbs; ++i) bulbs[i] = {d->bulbs[i].point.x, d->bulbs[i].point.y, d->bulbs[i].point.z, d->bulbs[i].color.r, d->bulbs[i].color.g, d->bulbs[i].color.b, 0, 0, 0, 0, 0, 0};
  sc->d.spheres = nullptr; sc->d.triangles = nullptr; sc->d.prim_refs = nullptr; sc->d.planes = nullptr; sc->d.suns = nullptr; sc->d.bulbs = nullptr;

  int rc = MIRT_OK;
  auto chk = [&](int r) { if (rc == MIRT_OK) rc = r; };
  // record heap: [internal nodes 64 B each | primitive records in sorted order: sphere 16 B, triangle 48 B | 64 B pad] -- the
  // traversal kernel addresses any record with one 32-bit byte offset.  The build fills it (lbvh_build.hip).
  {
    const size_t nodes_bytes = N > 1 ? 64 * (size_t)(N - 1) : 0;
    const size_t sph_bytes = 16 * (size_t)sc->Ns, tri_bytes = 48 * (size_t)sc->Nt;
    // (quantised node records behind the primitives, scene_dev.h: 32-byte ones for a sphere-only scene, else the wide ones)
    const size_t qnode_bytes = (sc->Nt == 0 && N > 1) ? 32 * (size_t)(N - 1) : 0;
    const size_t wnode_bytes = (sc->Nt > 0 && N > 1) ? 64 * (size_t)(N - 1) + 128 : 0;      // wide records (scene_dev.h), line-aligned
    const size_t total = nodes_bytes + sph_bytes + tri_bytes + 64 + qnode_bytes + wnode_bytes;
    if (total > 0xfffffff0ull) { delete sc; set_error("mirt_scene_create: scene too large for 32-bit record offsets"); return MIRT_ERR_ARG; }
    hipError_t e = hipMalloc(&sc->heap, total);
    if (e == hipSuccess) e = hipMemset(sc->heap, 0, total);
    if (e != hipSuccess) { delete sc; return hip_fail(e, "hipMalloc(heap)", __FILE__, __LINE__); }
    sc->prim_base = (uint32_t)nodes_bytes;
    sc->qnode_base = qnode_bytes ? (uint32_t)(nodes_bytes + sph_bytes + tri_bytes + 64) : 0u;
    sc->wnode_base = wnode_bytes ? (uint32_t)((nodes_bytes + sph_bytes + tri_bytes + 64 + qnode_bytes + 127) / 128 * 128) : 0u;
    sc->nodes = reinterpret_cast<float4*>(sc->heap);
  }
  chk(upload(&sc->spheres, spheres)); chk(upload(&sc->tris, tris));
  chk(upload(&sc->tri_verts, verts)); chk(upload(&sc->mats, mats));
  chk(upload(&sc->refs_in, refs)); chk(upload(&sc->planes, planes)); chk(upload(&sc->suns, suns)); chk(upload(&sc->bulbs, bulbs));
  auto alloc = [&](void** p, size_t bytes) { if (rc == MIRT_OK && bytes) { hipError_t e = hipMalloc(p, bytes); if (e != hipSuccess) rc = hip_fail(e, "hipMalloc", __FILE__, __LINE__); } };
  if (N > 0) {
    alloc((void**)&sc->codes, 4 * (size_t)N); alloc((void**)&sc->order, 4 * (size_t)N);
    alloc((void**)&sc->parent, 4 * (2 * (size_t)N - 1)); alloc((void**)&sc->boxes, 24 * (2 * (size_t)N - 1));
    if (N > 1) { alloc((void**)&sc->child_l, 4 * (size_t)(N - 1)); alloc((void**)&sc->child_r, 4 * (size_t)(N - 1)); alloc((void**)&sc->range, 8 * (size_t)(N - 1)); }
    alloc((void**)&sc->unit_prim, 4 * ((size_t)sc->Ns + 3 * (size_t)sc->Nt));
    alloc((void**)&sc->tris_before, 4 * ((size_t)N + 1));
  }
  alloc((void**)&sc->bounds_keys, 6 * 4);
  alloc((void**)&sc->qparams, 9 * 4);
  alloc((void**)&sc->tri_boxes, 32 * (size_t)sc->Nt);
  for (int i = 0; i < mirt::MIRT_MAX_FRAMES; ++i) {
    alloc((void**)&sc->ctx[i].counters, 16 * sizeof(unsigned long long));
    if (rc == MIRT_OK && hipMemset(sc->ctx[i].counters, 0, 16 * sizeof(unsigned long long)) != hipSuccess) rc = MIRT_ERR_HIP;
  }
  if (rc == MIRT_OK) {
    hipError_t e = hipEventCreate(&sc->ev0);
    if (e == hipSuccess) e = hipEventCreate(&sc->ev1);
    if (e == hipSuccess) e = hipEventCreate(&sc->so_ev);
    for (int i = 0; i < mirt::MIRT_MAX_FRAMES && e == hipSuccess; ++i) {
      e = hipEventCreate(&sc->ctx[i].ev0);
      if (e == hipSuccess) e = hipEventCreate(&sc->ctx[i].ev1);
      if (e == hipSuccess) e = hipEventCreate(&sc->ctx[i].ev2);
      if (e == hipSuccess) e = hipEventCreate(&sc->ctx[i].ev3);
      if (e == hipSuccess) e = hipEventCreate(&sc->ctx[i].order_ev);
    }
    if (e != hipSuccess) rc = hip_fail(e, "hipEventCreate", __FILE__, __LINE__);
  }
  if (rc != MIRT_OK) { mirt_scene_destroy(sc); return rc; }
  *out = sc;
  return MIRT_OK;
}

} // namespace

extern "C" {

void mirt_scene_destroy(MirtScene* sc)
{
  if (!sc) return;
  hipSetDevice(sc->device);
  hipDeviceSynchronize();
  hipFree(sc->heap); hipFree(sc->spheres); hipFree(sc->tris); hipFree(sc->tri_verts); hipFree(sc->mats); hipFree(sc->refs_in);
  hipFree(sc->unit_prim); hipFree(sc->tris_before); hipFree(sc->range);
  hipFree(sc->planes); hipFree(sc->suns); hipFree(sc->bulbs);
  hipFree(sc->codes); hipFree(sc->order); hipFree(sc->child_l); hipFree(sc->child_r); hipFree(sc->parent); hipFree(sc->boxes); hipFree(sc->build_ws);
  hipFree(sc->bounds_keys); hipFree(sc->qparams); hipFree(sc->tri_boxes);
  for (int i = 0; i < mirt::MIRT_MAX_FRAMES; ++i) {
    mirt::RenderCtx& c = sc->ctx[i];
    hipFree(c.samples); hipFree(c.stack_spill); hipFree(c.pending); hipFree(c.counters); hipFree(c.args_dev);
    hipFree(c.chunk_cost); for (uint32_t* o : c.order_out) hipFree(o);
    if (c.ev0) hipEventDestroy(c.ev0);
    if (c.ev1) hipEventDestroy(c.ev1);
    if (c.ev2) hipEventDestroy(c.ev2);
    if (c.ev3) hipEventDestroy(c.ev3);
    if (c.order_ev) hipEventDestroy(c.order_ev);
    for (hipEvent_t e : c.slab_ev) hipEventDestroy(e);
  }
  rng_cache_free(&sc->rng);
  hipFree(sc->wf_state); hipFree(sc->wf_rays); hipFree(sc->wf_ctr);
  if (sc->wf_ctr_host) hipHostFree(sc->wf_ctr_host);
  for (hipEvent_t e : sc->wf_events) hipEventDestroy(e);
  if (sc->ev0) hipEventDestroy(sc->ev0);
  if (sc->ev1) hipEventDestroy(sc->ev1);
  if (sc->so_ev) hipEventDestroy(sc->so_ev);
  hipFree(sc->so_order); hipFree(sc->so_keys); hipFree(sc->so_keys2); hipFree(sc->so_ws);
  delete sc;
}

int mirt_build_lbvh(MirtScene* sc, void* stream, float* build_ms)
{
  if (!sc) { set_error("mirt_build_lbvh: null scene"); return MIRT_ERR_ARG; }
  MIRT_HIP(hipSetDevice(sc->device));
  int rc = build_lbvh(sc, (hipStream_t)stream);
  if (rc == MIRT_OK && build_ms) *build_ms = sc->build_ms;
  return rc;
}

int64_t mirt_render_num_pixels(const MirtRenderParams* p) { return p ? render_num_pixels(p) : -1; }

int mirt_part_pixel_xy(const MirtRenderParams* p, int64_t local, int32_t* x, int32_t* y)
{
  if (!p) { set_error("mirt_part_pixel_xy: null argument"); return MIRT_ERR_ARG; }
  return part_pixel(p, local, x, y);
}

int mirt_render(MirtScene* sc, const MirtRenderParams* p, void* d_rgba8, void* d_rgba_f32, void* stream)
{
  if (!sc || !p) { set_error("mirt_render: null argument"); return MIRT_ERR_ARG; }
  MIRT_HIP(hipSetDevice(sc->device));
  return render(sc, p, d_rgba8, d_rgba_f32, (hipStream_t)stream);
}

int mirt_render_accumulate(MirtScene* sc, const MirtRenderParams* p, void* d_accum_f32, int sample_first, int sample_count, void* stream)
{
  if (!sc || !p) { set_error("mirt_render_accumulate: null argument"); return MIRT_ERR_ARG; }
  MIRT_HIP(hipSetDevice(sc->device));
  return render_accumulate(sc, p, d_accum_f32, sample_first, sample_count, (hipStream_t)stream);
}

int mirt_finalize(const MirtRenderParams* p, const void* d_accum_f32, int total_samples, void* d_rgba8, void* stream)
{
  if (!p) { set_error("mirt_finalize: null argument"); return MIRT_ERR_ARG; }
  return finalize(p, d_accum_f32, total_samples, d_rgba8, (hipStream_t)stream);
}

int mirt_scatter_part(const MirtRenderParams* p, const void* d_part_rgba8, void* d_frame_rgba8, void* stream)
{
  if (!p) { set_error("mirt_scatter_part: null argument"); return MIRT_ERR_ARG; }
  return scatter_part(p, d_part_rgba8, d_frame_rgba8, (hipStream_t)stream);
}

int mirt_get_stats(MirtScene* sc, MirtStats* out)
{
  if (!sc || !out) { set_error("mirt_get_stats: null argument"); return MIRT_ERR_ARG; }
  memset(out, 0, sizeof(*out));
  MIRT_HIP(hipSetDevice(sc->device));
  out->build_ms = sc->build_ms;
  out->num_nodes = sc->N > 0 ? 2 * sc->N - 1 : 0;
  if (!sc->last) return MIRT_OK;
  for (int i = 0; i < mirt::MIRT_MAX_FRAMES; ++i) {   // finish and time every frame still in flight
    mirt::RenderCtx& c = sc->ctx[i];
    if (c.used) {     // capacity overflows of the frames this context rendered since the previous call
      MIRT_HIP(hipEventSynchronize(c.ev3));
      unsigned long long ov = 0;
      hipLaunchKernelGGL(take_overflow_kernel, dim3(1), dim3(1), 0, c.stream, c.counters);
      MIRT_HIP(hipGetLastError());
      MIRT_HIP(hipStreamSynchronize(c.stream));
      MIRT_HIP(hipMemcpy(&ov, c.counters + 10, sizeof(ov), hipMemcpyDeviceToHost));
      sc->overflow_events += ov;
    }
    if (c.used && !c.timed) {
      MIRT_HIP(hipEventSynchronize(c.ev3));
      float ms = 0.0f;
      int rc = trace_ms_of(c, &ms);
      if (rc != MIRT_OK) return rc;
      sc->trace_ms_sum += ms; sc->trace_frames += 1; c.timed = true;
    }
  }
  out->frames_timed = sc->trace_frames;
  out->trace_kernel_ms_mean = sc->trace_frames ? (float)(sc->trace_ms_sum / sc->trace_frames) : 0.0f;
  sc->trace_ms_sum = 0.0; sc->trace_frames = 0;
  mirt::RenderCtx& cx = *sc->last;
  { int rc = trace_ms_of(cx, &out->trace_kernel_ms); if (rc != MIRT_OK) return rc; }
  out->trace_launches = cx.launches;
  out->node_record_bytes = cx.node_bytes;
  MIRT_HIP(hipEventElapsedTime(&out->render_ms, cx.ev0, cx.ev3));
  if (cx.counted) {
    unsigned long long c[8];
    MIRT_HIP(hipMemcpy(c, cx.counters, sizeof(c), hipMemcpyDeviceToHost));
    out->samples = c[0]; out->rays = c[1]; out->shadow_rays = c[2]; out->internal_visits = c[3];
    out->sphere_tests = c[4]; out->tri_tests = c[5]; out->mat_fetches = c[6]; out->max_stack = c[7];
    MIRT_HIP(hipMemcpy(&out->rays_traversed, cx.counters + 11, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  }
  out->overflow_events = sc->overflow_events;
  sc->overflow_events = 0;
  if (out->overflow_events) {
    set_error("mirt_get_stats: capacity overflow during a render (the pending-children list of refraction / gi rays was full): the image is missing contributions");
    return MIRT_ERR_STATE;
  }
  return MIRT_OK;
}

int mirt_get_tree(MirtScene* sc, MirtTreeNode* nodes, uint32_t* codes, MirtPrimRef* refs, float* bounds)
{
  if (!sc) { set_error("mirt_get_tree: null scene"); return MIRT_ERR_ARG; }
  MIRT_HIP(hipSetDevice(sc->device));
  return get_tree(sc, nodes, codes, refs, bounds);
}

int mirt_probe_math(int device, int which, int n, const float* host_in, float* host_out)
{
  if (n <= 0 || !host_in || !host_out) { set_error("mirt_probe_math: bad argument"); return MIRT_ERR_ARG; }
  return probe_math(device, which, n, host_in, host_out);
}

int mirt_probe_xorwow(int device, int spp, int num_streams, int draws, uint32_t* host_out)
{
  if (num_streams <= 0 || draws <= 0 || !host_out) { set_error("mirt_probe_xorwow: bad argument"); return MIRT_ERR_ARG; }
  return probe_xorwow(device, spp, num_streams, draws, host_out);
}

} // extern "C"
