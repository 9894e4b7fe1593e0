// svsdf_points.hpp -- host-side query-point producer: point cloud -> occupancy grid -> occupied-voxel
// centres inside axis-aligned boxes around the trajectory waypoints.  
//
// Behavioural spec: reference
//   PCS = src/map_manager/src/PCSmap_manager.cpp        rcvGlobalMapHandler :88-210
//   PCH = src/map_manager/include/map_manager/PCSmap_manager.h   projInMap :128-135,
//         getPointsInAABBOutOfLastOne :184-219, unifiedID :118-125
//   GRD = src/map_manager/src/Gridmap3D.cpp              createGridMap :25-41, isInMap :43-71,
//         getGridIndex :137-177, getGridCubeCenter :184-195, isIndexOccupied :239-283
//   plan_manager.cpp:156-175 (waypoint loop, tmp_pos = (999,999,999), half extents bd/3)
// The reference collects into std::unordered_map<int, Vector3d> (iteration order unspecified); this
// producer returns the points ordered by the same unified voxel id -- the cost is a sum, so order
// does not matter to the optimizer.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace svsdf_host {

class OccupancyMap {
 public:
  // PCS:116-178: bounds from the cloud (floats widened to double), grid of `resolution`, a voxel is
  // occupied when it holds >= sta_threshold points.
  void build(const float *xyz, size_t n, double resolution, int sta_threshold) {
    res_ = resolution;
    for (int d = 0; d < 3; ++d) { bmin_[d] = 999999999.0; bmax_[d] = -999999999.0; }  // PCS:11-12
    for (size_t i = 0; i < n; ++i)
      for (int d = 0; d < 3; ++d) {
        const double v = (double)xyz[3 * i + d];
        if (v > bmax_[d]) bmax_[d] = v;
        if (v < bmin_[d]) bmin_[d] = v;
      }
    for (int d = 0; d < 3; ++d) size_[d] = (int)std::ceil((bmax_[d] - bmin_[d]) / res_);  // GRD:29-31
    const size_t total = (size_t)std::max(0, size_[0]) * std::max(0, size_[1]) * std::max(0, size_[2]);
    std::vector<double> cnt(total, 0.0);
    for (size_t i = 0; i < n; ++i) {
      int id[3];
      grid_index((double)xyz[3 * i], (double)xyz[3 * i + 1], (double)xyz[3 * i + 2], id);
      if (total) cnt[addr(id[0], id[1], id[2])] += 1.0;
    }
    occ_.assign(total, 0);
    occupied_ = 0;
    for (size_t a = 0; a < total; ++a)
      if (cnt[a] >= (double)sta_threshold) { occ_[a] = 1; ++occupied_; }
  }

  // PCH:184-219 applied to every centre in turn (plan_manager.cpp:156-167).  centres: n x 3.
  void gather(const double *centres, size_t n, const double halfbd[3], std::vector<double> &out_xyz) const {
    std::map<int, int> seen;  // unified id -> 1
    std::vector<int> ids;
    double last[3] = {999.0, 999.0, 999.0};  // tmp_pos
    for (size_t c = 0; c < n; ++c) {
      const double *ctr = centres + 3 * c;
      int c1[3], c2[3], l1[3], l2[3];
      box_ids(ctr, halfbd, c1, c2);
      box_ids(last, halfbd, l1, l2);
      for (int i = c1[0]; i <= c2[0]; ++i)
        for (int j = c1[1]; j <= c2[1]; ++j)
          for (int k = c1[2]; k <= c2[2]; ++k) {
            if (i > l2[0] || i < l1[0] || j > l2[1] || j < l1[1] || k > l2[2] || k < l1[2]) {
              if (occupied(i, j, k)) {
                const int uid = k * size_[0] * size_[1] + j * size_[0] + i;  // PCH:118-125
                if (seen.emplace(uid, 1).second) ids.push_back(uid);
              }
            }
          }
      std::memcpy(last, ctr, sizeof(last));
    }
    out_xyz.clear();
    for (const auto &kv : seen) {
      const int uid = kv.first;
      const int i = uid % size_[0], j = (uid / size_[0]) % size_[1], k = uid / (size_[0] * size_[1]);
      double p[3];
      cube_center(i, j, k, p);
      out_xyz.insert(out_xyz.end(), p, p + 3);
    }
  }

  const int *dims() const { return size_; }
  const double *bmin() const { return bmin_; }
  const double *bmax() const { return bmax_; }
  size_t occupied_count() const { return occupied_; }

 private:
  bool in_map(double x, double y, double z) const {  // GRD:43-71
    return !(x < bmin_[0] || y < bmin_[1] || z < bmin_[2] || x > bmax_[0] || y > bmax_[1] || z > bmax_[2]);
  }
  // GRD:137-177, including its clamping quirk (a negative iy / iz resets ix)
  void grid_index(double x, double y, double z, int id[3]) const {
    if (!in_map(x, y, z)) { id[0] = id[1] = id[2] = 0; return; }
    int ix = (int)std::floor((x - bmin_[0]) / res_);
    int iy = (int)std::floor((y - bmin_[1]) / res_);
    int iz = (int)std::floor((z - bmin_[2]) / res_);
    if (ix < 0) ix = 0;
    if (ix >= size_[0]) ix = size_[0] - 1;
    if (iy < 0) ix = 0;
    if (iy >= size_[1]) iy = size_[1] - 1;
    if (iz < 0) ix = 0;
    if (iz >= size_[2]) iz = size_[2] - 1;
    id[0] = ix; id[1] = iy; id[2] = iz;
  }
  void box_ids(const double ctr[3], const double halfbd[3], int c1[3], int c2[3]) const {
    double a[3], b[3];
    for (int d = 0; d < 3; ++d) {  // projInMap PCH:128-135
      a[d] = ctr[d] - halfbd[d];
      b[d] = ctr[d] + halfbd[d];
      if (a[d] < bmin_[d]) a[d] = bmin_[d];
      if (a[d] > bmax_[d]) a[d] = bmax_[d];
      if (b[d] < bmin_[d]) b[d] = bmin_[d];
      if (b[d] > bmax_[d]) b[d] = bmax_[d];
    }
    grid_index(a[0], a[1], a[2], c1);
    grid_index(b[0], b[1], b[2], c2);
  }
  size_t addr(int i, int j, int k) const { return ((size_t)i * size_[1] + j) * size_[2] + k; }  // GridMap3D.h:129-130
  bool occupied(int i, int j, int k) const {  // GRD:239-283: out-of-range counts as occupied
    if (i < 0 || i >= size_[0] || j < 0 || j >= size_[1] || k < 0 || k >= size_[2]) return true;
    return occ_[addr(i, j, k)] != 0;
  }
  void cube_center(int i, int j, int k, double p[3]) const {  // GRD:184-195
    if (i < 0 || i >= size_[0] || j < 0 || j >= size_[1] || k < 0 || k >= size_[2]) { p[0] = p[1] = p[2] = 0.0; return; }
    p[0] = (i + 0.5) * res_ + bmin_[0];
    p[1] = (j + 0.5) * res_ + bmin_[1];
    p[2] = (k + 0.5) * res_ + bmin_[2];
  }

  double res_ = 1.0;
  double bmin_[3] = {0, 0, 0}, bmax_[3] = {0, 0, 0};
  int size_[3] = {0, 0, 0};
  std::vector<unsigned char> occ_;
  size_t occupied_ = 0;
};

// ASCII PCD v0.7 with FIELDS x y z (src/plan_manager/pcds/map_*.pcd).  Returns false on any other layout.
inline bool read_pcd_ascii(const char *path, std::vector<float> &xyz) {
  std::FILE *f = std::fopen(path, "r");
  if (!f) return false;
  char line[512];
  bool data = false, fields_ok = false, ascii = false;
  long npoints = -1;
  xyz.clear();
  while (std::fgets(line, sizeof(line), f)) {
    if (!data) {
      if (std::strncmp(line, "FIELDS", 6) == 0) fields_ok = std::strncmp(line, "FIELDS x y z", 12) == 0;
      else if (std::strncmp(line, "POINTS", 6) == 0) npoints = std::atol(line + 6);
      else if (std::strncmp(line, "DATA", 4) == 0) { data = true; ascii = std::strstr(line, "ascii") != nullptr; }
      continue;
    }
    float x, y, z;
    if (std::sscanf(line, "%f %f %f", &x, &y, &z) == 3) { xyz.push_back(x); xyz.push_back(y); xyz.push_back(z); }
  }
  std::fclose(f);
  return fields_ok && ascii && (npoints < 0 || (long)(xyz.size() / 3) == npoints);
}

}  // namespace svsdf_host
