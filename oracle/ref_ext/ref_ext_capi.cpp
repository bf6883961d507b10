// TEST INFRASTRUCTURE ONLY.  C-ABI shim around the reference's own CPU sources
// (compiled where they lie under /root/reference, see Makefile) so that the
// genuine grid_subsampling_cpu() / radius_neighbors_cpu() can be called through
// ctypes without the pybind/ATen front end (which needs torch headers and a CUDA
// header the ROCm image lacks).  Mirrors what the reference front ends do:
//   geotransformer/extensions/cpu/radius_neighbors/radius_neighbors.cpp:5-76
//   geotransformer/extensions/cpu/grid_subsampling/grid_subsampling.cpp:5-83
#include <cstring>
#include <vector>
#include "cpu/grid_subsampling/grid_subsampling_cpu.h"
#include "cpu/radius_neighbors/radius_neighbors_cpu.h"

static std::vector<long> g_neighbors;
static std::vector<PointXYZ> g_s_points, g_s_normals;
static std::vector<long> g_s_lengths;

extern "C" {

long ref_radius_neighbors(const float* q, long nq, const float* s, long ns, const long* q_len,
                          const long* s_len, long batch, float radius) {
  std::vector<PointXYZ> vq(reinterpret_cast<const PointXYZ*>(q), reinterpret_cast<const PointXYZ*>(q) + nq);
  std::vector<PointXYZ> vs(reinterpret_cast<const PointXYZ*>(s), reinterpret_cast<const PointXYZ*>(s) + ns);
  std::vector<long> ql(q_len, q_len + batch), sl(s_len, s_len + batch);
  g_neighbors.clear();
  radius_neighbors_cpu(vq, vs, ql, sl, g_neighbors, radius);
  return nq ? (long)(g_neighbors.size() / nq) : 0;
}

void ref_fetch_neighbors(long* out) { std::memcpy(out, g_neighbors.data(), sizeof(long) * g_neighbors.size()); }

long ref_grid_subsampling(const float* p, const float* n, long np, const long* len, long batch, float voxel) {
  std::vector<PointXYZ> vp(reinterpret_cast<const PointXYZ*>(p), reinterpret_cast<const PointXYZ*>(p) + np);
  std::vector<PointXYZ> vn(reinterpret_cast<const PointXYZ*>(n), reinterpret_cast<const PointXYZ*>(n) + np);
  std::vector<long> l(len, len + batch);
  g_s_points.clear(); g_s_normals.clear(); g_s_lengths.clear();
  grid_subsampling_cpu(vp, g_s_points, l, g_s_lengths, vn, g_s_normals, voxel);
  return (long)g_s_points.size();
}

void ref_fetch_subsampled(float* sp, float* sn, long* sl) {
  std::memcpy(sp, g_s_points.data(), sizeof(PointXYZ) * g_s_points.size());
  std::memcpy(sn, g_s_normals.data(), sizeof(PointXYZ) * g_s_normals.size());
  std::memcpy(sl, g_s_lengths.data(), sizeof(long) * g_s_lengths.size());
}

}  // extern "C"
