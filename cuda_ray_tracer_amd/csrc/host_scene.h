// Host-side scene container (the reference's StlConfig, config.hpp:24-73) and parser.
#ifndef MIRT_HOST_SCENE_H
#define MIRT_HOST_SCENE_H

#include <istream>
#include <string>
#include <vector>

#include "../../include/mirt.h"

namespace mirt {

extern thread_local std::string g_last_error;
void set_error(const std::string& s);

class HostScene {
public:
  HostScene();
  int parse_stream(std::istream& in);                          // parseInput, parse.cpp:16-39
  int parse_line(const std::vector<std::string>& words);      // parseLine, parse.cpp:41-222
  void make_synthetic(uint64_t seed, int num_spheres, int num_triangles);
  void fill_desc(MirtSceneDesc* d) const;
  MirtMaterials current_material() const;

  int width, height;
  std::string filename;
  MirtRGB color;
  int bounces, aa;
  float dof_focus, dof_lens;
  MirtVec3 forward, right, up, eye, target_up;
  float expose;
  bool fisheye, panorama;
  float ior, rough;
  int gi;
  MirtRGB trans, shine;

  std::vector<MirtSphere> spheres;
  std::vector<MirtTriangle> triangles;
  std::vector<MirtPrimRef> refs;
  std::vector<MirtPlane> planes;
  std::vector<MirtSun> suns;
  std::vector<MirtBulb> bulbs;
  std::vector<MirtVec3> vertices;
};

} // namespace mirt
#endif
