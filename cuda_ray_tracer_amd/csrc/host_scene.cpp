// Host-side scene front end: the reference's StlConfig + parseInput (config.hpp:24-73, parse.cpp:16-222),
// rewritten around the POD structs of include/mirt.h.  Same grammar, same defaults, same arithmetic
// (fp32, one rounding per operation; the library is built with -ffp-contract=off).
#include "host_scene.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace mirt {

thread_local std::string g_last_error;
void set_error(const std::string& s) { g_last_error = s; }

namespace {

inline MirtVec3 v3(float x, float y, float z) { MirtVec3 v; v.x = x; v.y = y; v.z = z; return v; }
inline MirtVec3 sub(const MirtVec3& a, const MirtVec3& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float dot(const MirtVec3& a, const MirtVec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline MirtVec3 cross(const MirtVec3& a, const MirtVec3& b)
{
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// vec3.cuh:7-18
inline bool fequal(float a, float b, float epsilon = 1e-6f)
{
  float diff = fabsf(a - b);
  float largest = fmaxf(fabsf(a), fabsf(b));
  if (largest < 1e-6f) return diff < epsilon;
  return diff / largest < epsilon;
}
// vec3.cuh:72-82
inline MirtVec3 normalize(const MirtVec3& v)
{
  float mag = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  if (fequal(mag, 0.0f)) return v3(0.0f, 0.0f, 0.0f);
  float inv = 1.0f / mag;
  return v3(v.x * inv, v.y * inv, v.z * inv);
}

bool to_int(const std::string& w, int* out)
{
  try { size_t pos = 0; *out = std::stoi(w, &pos); return true; } catch (...) { return false; }
}
bool to_float(const std::string& w, float* out)
{
  try { size_t pos = 0; *out = std::stof(w, &pos); return true; } catch (...) { return false; }
}

} // namespace

HostScene::HostScene()
{
  width = 0; height = 0; filename = "file.txt";
  color = {1.0f, 1.0f, 1.0f};
  bounces = 4; aa = 0; dof_focus = 0.0f; dof_lens = 0.0f;
  forward = v3(0.0f, 0.0f, -1.0f); right = v3(1.0f, 0.0f, 0.0f); up = v3(0.0f, 1.0f, 0.0f);
  eye = v3(0.0f, 0.0f, 0.0f); target_up = v3(0.0f, 1.0f, 0.0f);
  expose = INFINITY; fisheye = false; panorama = false;
  ior = 1.458f; rough = 0.0f; gi = 0;
  trans = {0.0f, 0.0f, 0.0f}; shine = {0.0f, 0.0f, 0.0f};
}

MirtMaterials HostScene::current_material() const
{
  MirtMaterials m;
  m.color = color; m.shininess = shine; m.trans = trans; m.ior = ior; m.roughness = rough;
  return m;
}

// Triangle(Vertex a, Vertex b, Vertex c, RGB), object.cuh:177-191
static MirtTriangle make_triangle(const MirtVec3& p0, const MirtVec3& p1, const MirtVec3& p2, const MirtMaterials& mat)
{
  MirtTriangle t;
  t.p0 = p0; t.p1 = p1; t.p2 = p2; t.mat = mat;
  t.nor = normalize(cross(sub(p1, p0), sub(p2, p0)));
  MirtVec3 a1 = cross(sub(p2, p0), t.nor);
  MirtVec3 a2 = cross(sub(p1, p0), t.nor);
  float k1 = 1 / (dot(a1, sub(p1, p0)));
  float k2 = 1 / (dot(a2, sub(p2, p0)));
  t.e1 = v3(a1.x * k1, a1.y * k1, a1.z * k1);
  t.e2 = v3(a2.x * k2, a2.y * k2, a2.z * k2);
  return t;
}

// Plane(a,b,c,d,rgb), object.cuh:136-141.  The host `pow(float,int)` is double; the sum is rounded at operator/.
static MirtPlane make_plane(float a, float b, float c, float d, const MirtMaterials& mat)
{
  MirtPlane p;
  p.a = a; p.b = b; p.c = c; p.d = d; p.mat = mat;
  p.nor = normalize(v3(a, b, c));
  float den = (float)(std::pow((double)a, 2) + std::pow((double)b, 2) + std::pow((double)c, 2));
  float nd = -d;
  p.point = v3((a * nd) / den, (b * nd) / den, (c * nd) / den);
  return p;
}

// parseLine, parse.cpp:41-222
int HostScene::parse_line(const std::vector<std::string>& w)
{
  if (w.empty()) return MIRT_OK;
  const std::string& k = w[0];
  const size_t n = w.size();
  float f[4]; int iv[3];
  auto floats = [&](int cnt) { for (int i = 0; i < cnt; ++i) if (!to_float(w[1 + i], &f[i])) return false; return true; };
  auto ints = [&](int cnt) { for (int i = 0; i < cnt; ++i) if (!to_int(w[1 + i], &iv[i])) return false; return true; };
  bool ok = true;

  if (k == "png" && n == 4) { ok = ints(2); if (ok) { width = iv[0]; height = iv[1]; filename = w[3]; } }
  else if (k == "bounces" && n == 2) { ok = ints(1); if (ok) bounces = iv[0]; }
  else if (k == "forward" && n == 4) {
    ok = floats(3);
    if (ok) { forward = v3(f[0], f[1], f[2]); right = normalize(cross(forward, up)); up = normalize(cross(right, forward)); }
  }
  else if (k == "up" && n == 4) {
    ok = floats(3);
    if (ok) { target_up = v3(f[0], f[1], f[2]); right = normalize(cross(forward, target_up)); up = normalize(cross(right, forward)); }
  }
  else if (k == "eye" && n == 4) { ok = floats(3); if (ok) eye = v3(f[0], f[1], f[2]); }
  else if (k == "expose" && n == 2) { ok = floats(1); if (ok) expose = f[0]; }
  else if (k == "dof" && n == 3) { ok = floats(2); if (ok) { dof_focus = f[0]; dof_lens = f[1]; } }
  else if (k == "aa" && n == 2) { ok = ints(1); if (ok) aa = iv[0]; }
  else if (k == "panorama" && n == 1) panorama = true;
  else if (k == "fisheye" && n == 1) fisheye = true;
  else if (k == "gi" && n == 2) { ok = ints(1); if (ok) gi = iv[0]; }
  else if (k == "color" && n == 4) { ok = floats(3); if (ok) color = {f[0], f[1], f[2]}; }
  else if (k == "roughness" && n == 2) { ok = floats(1); if (ok) rough = f[0]; }
  else if (k == "shininess" && n == 2) { ok = floats(1); if (ok) shine = {f[0], f[0], f[0]}; }
  else if (k == "shininess" && n == 4) { ok = floats(3); if (ok) shine = {f[0], f[1], f[2]}; }
  else if (k == "transparency" && n == 2) { ok = floats(1); if (ok) trans = {f[0], f[0], f[0]}; }
  else if (k == "transparency" && n == 4) { ok = floats(3); if (ok) trans = {f[0], f[1], f[2]}; }
  else if (k == "ior" && n == 2) { ok = floats(1); if (ok) ior = f[0]; }
  else if (k == "sphere" && n == 5) {
    ok = floats(4);
    if (ok) {
      MirtSphere s; s.c = v3(f[0], f[1], f[2]); s.r = f[3]; s.mat = current_material();
      spheres.push_back(s);
      MirtPrimRef r; r.type = 0; r.id = (uint32_t)(spheres.size() - 1);
      refs.push_back(r);
    }
  }
  else if (k == "plane" && n == 5) { ok = floats(4); if (ok) planes.push_back(make_plane(f[0], f[1], f[2], f[3], current_material())); }
  else if (k == "xyz" && n == 4) { ok = floats(3); if (ok) vertices.push_back(v3(f[0], f[1], f[2])); }
  else if (k == "tri" && n == 4) {
    ok = ints(3);
    if (ok) {
      const int size = (int)vertices.size();
      int idx[3];
      for (int i = 0; i < 3; ++i) idx[i] = (iv[i] > 0) ? iv[i] - 1 : size + iv[i];   // parse.cpp:178-180
      for (int i = 0; i < 3; ++i) if (idx[i] < 0 || idx[i] >= size) ok = false;       // UB in the reference
      if (ok) {
        triangles.push_back(make_triangle(vertices[idx[0]], vertices[idx[1]], vertices[idx[2]], current_material()));
        MirtPrimRef r; r.type = 1; r.id = (uint32_t)(triangles.size() - 1);
        refs.push_back(r);
      }
    }
  }
  else if (k == "sun" && n == 4) { ok = floats(3); if (ok) { MirtSun s; s.dir = v3(f[0], f[1], f[2]); s.color = color; suns.push_back(s); } }
  else if (k == "bulb" && n == 4) { ok = floats(3); if (ok) { MirtBulb b; b.point = v3(f[0], f[1], f[2]); b.color = color; bulbs.push_back(b); } }
  else ok = false;

  if (!ok) { set_error("One of the lines are not valid."); return MIRT_ERR_PARSE; }
  return MIRT_OK;
}

int HostScene::parse_stream(std::istream& in)
{
  std::string line;
  while (std::getline(in, line)) {
    std::stringstream ss(line);
    std::vector<std::string> words;
    std::string word;
    while (ss >> word) words.push_back(word);
    int rc = parse_line(words);
    if (rc != MIRT_OK) return rc;
  }
  return MIRT_OK;
}

void HostScene::fill_desc(MirtSceneDesc* d) const
{
  memset(d, 0, sizeof(*d));
  d->width = width; d->height = height; d->bounces = bounces; d->aa = aa;
  d->dof_focus = dof_focus; d->dof_lens = dof_lens;
  d->forward = forward; d->right = right; d->up = up; d->eye = eye;
  d->expose = expose;
  d->fisheye = fisheye ? 1 : 0; d->panorama = panorama ? 1 : 0; d->gi = gi;
  d->num_spheres = (int32_t)spheres.size(); d->num_triangles = (int32_t)triangles.size();
  d->num_prims = (int32_t)refs.size(); d->num_planes = (int32_t)planes.size();
  d->num_suns = (int32_t)suns.size(); d->num_bulbs = (int32_t)bulbs.size();
  d->spheres = spheres.empty() ? nullptr : spheres.data();
  d->triangles = triangles.empty() ? nullptr : triangles.data();
  d->prim_refs = refs.empty() ? nullptr : refs.data();
  d->planes = planes.empty() ? nullptr : planes.data();
  d->suns = suns.empty() ? nullptr : suns.data();
  d->bulbs = bulbs.empty() ? nullptr : bulbs.data();
}

// ---- synthetic scene (BASELINE config 5) ----------------------------------
namespace {
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }           // [0,1)
  float range(double lo, double hi) { return (float)(lo + (hi - lo) * unit()); }
};
} // namespace

void HostScene::make_synthetic(uint64_t seed, int ns, int nt)
{
  SplitMix64 g(seed);
  width = 3840; height = 2160; filename = "synthetic.png";
  bounces = 4; aa = 256;
  // camera: eye 0 25 0, forward 0 -0.2 -1 (same update rule as the `forward` keyword)
  eye = v3(0.0f, 25.0f, 0.0f);
  forward = v3(0.0f, -0.2f, -1.0f);
  right = normalize(cross(forward, up));
  up = normalize(cross(right, forward));
  color = {1.0f, 1.0f, 1.0f};
  { MirtSun s; s.dir = v3(1.0f, 1.0f, 1.0f); s.color = {1.0f, 1.0f, 1.0f}; suns.push_back(s); }
  { MirtSun s; s.dir = v3(-1.0f, 0.5f, 0.5f); s.color = {0.5f, 0.5f, 0.5f}; suns.push_back(s); }
  color = {0.6f, 0.6f, 0.6f};
  planes.push_back(make_plane(0.0f, 1.0f, 0.0f, 0.0f, current_material()));
  shine = {0.25f, 0.25f, 0.25f};
  rough = 0.0f;
  spheres.reserve((size_t)ns); triangles.reserve((size_t)nt); refs.reserve((size_t)ns + (size_t)nt);
  for (int i = 0; i < ns; ++i) {
    MirtSphere s;
    s.c = v3(g.range(-50, 50), g.range(0, 50), g.range(-150, -50));
    s.r = g.range(0.05, 0.25);
    color = {g.range(0.2, 1.0), g.range(0.2, 1.0), g.range(0.2, 1.0)};
    s.mat = current_material();
    spheres.push_back(s);
    MirtPrimRef r; r.type = 0; r.id = (uint32_t)i; refs.push_back(r);
  }
  for (int i = 0; i < nt; ++i) {
    MirtVec3 c = v3(g.range(-50, 50), g.range(0, 50), g.range(-150, -50));
    MirtVec3 p[3];
    for (int k = 0; k < 3; ++k) p[k] = v3(c.x + g.range(-0.3, 0.3), c.y + g.range(-0.3, 0.3), c.z + g.range(-0.3, 0.3));
    color = {g.range(0.2, 1.0), g.range(0.2, 1.0), g.range(0.2, 1.0)};
    triangles.push_back(make_triangle(p[0], p[1], p[2], current_material()));
    MirtPrimRef r; r.type = 1; r.id = (uint32_t)i; refs.push_back(r);
  }
}

} // namespace mirt

// ---- C ABI ---------------------------------------------------------------------------------------
struct MirtHostScene { mirt::HostScene hs; };

extern "C" {

const char* mirt_last_error(void) { return mirt::g_last_error.c_str(); }
int mirt_version(void) { return MIRT_VERSION; }

int mirt_parse_scene_file(const char* path, MirtHostScene** out)
{
  if (!path || !out) { mirt::set_error("null argument"); return MIRT_ERR_ARG; }
  std::ifstream in(path);
  if (!in) { mirt::set_error("Error opening file..."); return MIRT_ERR_IO; }
  MirtHostScene* h = new MirtHostScene();
  int rc = h->hs.parse_stream(in);
  if (rc != MIRT_OK) { delete h; return rc; }
  *out = h;
  return MIRT_OK;
}

int mirt_parse_scene_text(const char* text, size_t len, MirtHostScene** out)
{
  if (!text || !out) { mirt::set_error("null argument"); return MIRT_ERR_ARG; }
  std::istringstream in(std::string(text, len));
  MirtHostScene* h = new MirtHostScene();
  int rc = h->hs.parse_stream(in);
  if (rc != MIRT_OK) { delete h; return rc; }
  *out = h;
  return MIRT_OK;
}

int mirt_synthetic_scene(uint64_t seed, int num_spheres, int num_triangles, MirtHostScene** out)
{
  if (!out || num_spheres < 0 || num_triangles < 0) { mirt::set_error("bad argument"); return MIRT_ERR_ARG; }
  MirtHostScene* h = new MirtHostScene();
  h->hs.make_synthetic(seed, num_spheres, num_triangles);
  *out = h;
  return MIRT_OK;
}

void mirt_host_scene_destroy(MirtHostScene* hs) { delete hs; }

int mirt_host_scene_desc(const MirtHostScene* hs, MirtSceneDesc* out)
{
  if (!hs || !out) { mirt::set_error("null argument"); return MIRT_ERR_ARG; }
  hs->hs.fill_desc(out);
  return MIRT_OK;
}

const char* mirt_host_scene_filename(const MirtHostScene* hs) { return hs ? hs->hs.filename.c_str() : ""; }

} // extern "C"
