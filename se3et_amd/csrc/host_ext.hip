// Host-pointer (CPU) variants of the two ops of the reference's pybind extension `geotransformer.ext` ("plus CPU
// (host-pointer) variants ... for DataLoader workers"): the reference runs them inside forked DataLoader workers
// (geotransformer/utils/data.py:159-209, utils/torch.py:65-75), where a HIP context cannot be used.  Plain C++ on host memory, stateless,
// safe to call concurrently from several processes / threads; same arithmetic contract as the device kernels (csrc/grid_subsample.hip,
// csrc/radius_neighbors.hip) and as the reference sources:
//   grid subsampling   extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-109, grid_subsampling_cpu.h:24-74
//   radius neighbours  extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91 (nanoflann radius search, sorted by distance)
// SE3_EXACT_FP: compiled with -ffp-contract=off (se3et_amd/build.py): the float expressions mirror unfused x86 arithmetic.
#define SE3_EXACT_FP 1
#include "common.h"
#include <math.h>
#include <algorithm>
#include <unordered_map>
#include <vector>

namespace {

// One cloud.  Output order = iteration order of a std::unordered_map<size_t, ...> filled in first-seen voxel order -- the very container
// (and libstdc++) the reference iterates, so the emission order is reproduced by construction.
int64_t grid_subsample_one(const float* pts, const float* nrm, int64_t n, float voxel, float* s_pts, float* s_nrm) {
  if (n == 0) return 0;
  float mn[3] = {pts[0], pts[1], pts[2]}, mx[3] = {pts[0], pts[1], pts[2]};
  for (int64_t i = 0; i < n; i++)
    for (int d = 0; d < 3; d++) {
      const float v = pts[3 * i + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
  const float inv = (float)(1.0 / (double)voxel);          // `minCorner * (1. / voxel_size)`: the scale is rounded to float by operator*
  float org[3];
  for (int d = 0; d < 3; d++) org[d] = floorf(mn[d] * inv) * voxel;
  const size_t nx = (size_t)(floorf((mx[0] - org[0]) / voxel) + 1), ny = (size_t)(floorf((mx[1] - org[1]) / voxel) + 1);
  struct Cell { int count; float sx, sy, sz; int64_t best; double best_d; };
  std::unordered_map<size_t, int64_t> slot_of;             // voxel id -> dense slot, keys inserted in first-seen order
  std::vector<Cell> cells;
  std::vector<int64_t> cell_of((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    // (through the signed conversion, explicitly: an index of -1 -- float rounding at the cloud's own minimum -- wraps as in the reference's
    // x86-64 build, grid_subsampling_cpu.cpp:47-49)
    const size_t ix = (size_t)(int64_t)floorf((pts[3 * i] - org[0]) / voxel), iy = (size_t)(int64_t)floorf((pts[3 * i + 1] - org[1]) / voxel),
                 iz = (size_t)(int64_t)floorf((pts[3 * i + 2] - org[2]) / voxel);
    const size_t key = ix + nx * iy + nx * ny * iz;
    auto it = slot_of.find(key);
    if (it == slot_of.end()) {
      it = slot_of.emplace(key, (int64_t)cells.size()).first;
      cells.push_back(Cell{0, 0.f, 0.f, 0.f, -1, 0.0});
    }
    Cell& c = cells[(size_t)it->second];
    c.count += 1;
    c.sx += pts[3 * i]; c.sy += pts[3 * i + 1]; c.sz += pts[3 * i + 2];            // float accumulation in input order
    cell_of[(size_t)i] = it->second;
  }
  for (int64_t i = 0; i < n; i++) {                                                  // member closest to the voxel mean, first minimum
    Cell& c = cells[(size_t)cell_of[(size_t)i]];
    const float a = (float)(1.0 / (double)c.count);
    const float dx = pts[3 * i] - c.sx * a, dy = pts[3 * i + 1] - c.sy * a, dz = pts[3 * i + 2] - c.sz * a;
    const double dist = (double)sqrtf(dx * dx + dy * dy + dz * dz);
    if (c.best < 0 || dist < c.best_d) { c.best = i; c.best_d = dist; }
  }
  int64_t o = 0;
  for (const auto& kv : slot_of) {
    const int64_t src = cells[(size_t)kv.second].best;
    for (int d = 0; d < 3; d++) {
      s_pts[3 * o + d] = pts[3 * src + d];
      if (s_nrm) s_nrm[3 * o + d] = nrm[3 * src + d];
    }
    o++;
  }
  return o;
}

}  // namespace
#include "kdtree_ref.h"
namespace {
using se3_kd::Hit;
using se3_kd::KdTree;

// All neighbours of every query of one cloud within `radius`: ascending distance, exactly tied distances in the reference's order.
template <typename Emit>
void radius_search_one(const float* q, int64_t nq, const float* s, int64_t ns, int64_t s_offset, float radius, Emit emit) {
  const float r2 = radius * radius;
  std::vector<Hit> hits;
  if (ns == 0) {
    for (int64_t i = 0; i < nq; i++) emit(i, hits);
    return;
  }
  const KdTree tree(s, ns);
  for (int64_t i = 0; i < nq; i++) {
    hits.clear();
    tree.radius(q + 3 * i, r2, hits);
    std::sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.d2 < b.d2; });      // (on the distance alone, as the reference)
    for (Hit& h : hits) h.idx += s_offset;
    emit(i, hits);
  }
}

}  // namespace

extern "C" int se3_grid_subsample_host(const float* points, const float* normals, int64_t n, const int64_t* lengths, int batch, float voxel_size,
                                       float* s_points, float* s_normals, int64_t* s_lengths) {
  SE3_REQUIRE(points && lengths && s_points && s_lengths, SE3_ERR_INVALID_ARG, "grid_subsample_host: null pointer");
  SE3_REQUIRE((normals == nullptr) == (s_normals == nullptr), SE3_ERR_INVALID_ARG, "grid_subsample_host: normals in and out must both be given or both NULL");
  SE3_REQUIRE(voxel_size > 0.f && batch >= 1, SE3_ERR_INVALID_ARG, "grid_subsample_host: bad voxel size / batch");
  int64_t start = 0, out = 0;
  for (int b = 0; b < batch; b++) {
    SE3_REQUIRE(lengths[b] >= 0 && start + lengths[b] <= n, SE3_ERR_INVALID_ARG, "grid_subsample_host: lengths exceed the point count");
    const int64_t m = grid_subsample_one(points + 3 * start, normals ? normals + 3 * start : nullptr, lengths[b], voxel_size,
                                         s_points + 3 * out, s_normals ? s_normals + 3 * out : nullptr);
    s_lengths[b] = m;
    out += m;
    start += lengths[b];
  }
  return SE3_OK;
}

// Two calls as for the reference's variable-width result: with out == NULL only *max_count is computed; with out (nq, limit) the rows
// are filled (ascending distance, padded with ns) and *max_count reports the largest in-radius count (may exceed limit).
extern "C" int se3_radius_neighbors_host(const float* q_points, int64_t nq, const float* s_points, int64_t ns, const int64_t* q_lengths,
                                         const int64_t* s_lengths, int batch, float radius, int64_t limit, int64_t* out, int64_t* max_count) {
  SE3_REQUIRE(q_points && s_points && q_lengths && s_lengths && max_count, SE3_ERR_INVALID_ARG, "radius_neighbors_host: null pointer");
  SE3_REQUIRE(batch >= 1 && (out == nullptr || limit >= 1), SE3_ERR_INVALID_ARG, "radius_neighbors_host: bad batch / limit");
  int64_t q0 = 0, s0 = 0, mc = 0;
  for (int b = 0; b < batch; b++) {
    SE3_REQUIRE(q0 + q_lengths[b] <= nq && s0 + s_lengths[b] <= ns, SE3_ERR_INVALID_ARG, "radius_neighbors_host: lengths exceed the point counts");
    radius_search_one(q_points + 3 * q0, q_lengths[b], s_points + 3 * s0, s_lengths[b], s0, radius, [&](int64_t i, const std::vector<Hit>& hits) {
      mc = std::max<int64_t>(mc, (int64_t)hits.size());
      if (out) {
        int64_t* row = out + (q0 + i) * limit;
        for (int64_t k = 0; k < limit; k++) row[k] = k < (int64_t)hits.size() ? hits[(size_t)k].idx : ns;
      }
    });
    q0 += q_lengths[b];
    s0 += s_lengths[b];
  }
  *max_count = mc;
  return SE3_OK;
}
