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

struct Hit { int64_t idx; float d2; };

// ---- radius search with the reference's ORDER of exactly tied distances -----------------------------------------------------------------
// The reference collects the matches of a query while it walks a k-d tree (nanoflann 1.3.x, KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float,
// ..>, .., 3>, leaf size 10: extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:31-33,56-59; extra/nanoflann/nanoflann.hpp) and then
// std::sorts them on the distance ALONE (nanoflann.hpp:208-214,1286-1287).  std::sort is unstable, so wherever distances tie exactly --
// 57 % of the stage-0 rows of data/demo -- the result depends on the order in which the walk met the matches, i.e. on the tree.  This is
// that tree, restated from the published algorithm (NOT the library's code: index arrays instead of pointer nodes, one recursive build):
//   build     (nanoflann.hpp:857-905)  a range of <= 10 indices is a leaf with the tight box of its points; otherwise cut it
//   cut       (:909-956)  the axis of the widest BOX side, among sides within 1e-5 of the widest the one whose POINTS spread most (first
//                         on ties); cut value = the middle of the box side, clamped to the points' range on that axis
//   partition (:967-1002) two sweeps of a two-pointer exchange: values < cut to the front, then values == cut behind them; the split index
//                         is the median position if it falls inside the run of equal values, else the nearer end of that run
//   node      (:893-903)  keeps the children's ACTUAL extents along the cut axis (their tight boxes), not the cut value; the box handed back
//                         to the parent is the union of the children's
//   walk      (:1348-1407) near child first (the side of the gap's midpoint the query lies on), the far child while the box distance, updated
//                         by the cut's own distance, is <= r^2; a leaf emits its indices in stored order when d2 < r^2 (:1355-1372,246-250)
// With the same libstdc++ std::sort on the same sequence the rows then equal the reference's bit for bit, ties included
// (tests/test_host_ext.py: order-sensitive checksums of all ten tables of data/demo; oracle/_ref where present).
class KdTree {
 public:
  KdTree(const float* pts, int64_t n) : p_(pts), perm_((size_t)n) {
    for (int64_t i = 0; i < n; i++) perm_[(size_t)i] = (size_t)i;
    for (int d = 0; d < 3; d++) box_lo_[d] = box_hi_[d] = pts[d];
    for (int64_t k = 1; k < n; k++)
      for (int d = 0; d < 3; d++) {
        const float v = pts[3 * k + d];
        if (v < box_lo_[d]) box_lo_[d] = v;
        if (v > box_hi_[d]) box_hi_[d] = v;
      }
    nodes_.reserve((size_t)(n / 4 + 8));
    build(0, (size_t)n, box_lo_, box_hi_);
  }

  // every index within r2 of q, in the order the reference's walk meets them
  void radius(const float* q, float r2, std::vector<Hit>& hits) const {
    float side[3] = {0.f, 0.f, 0.f}, outside = 0.f;
    for (int d = 0; d < 3; d++) {
      if (q[d] < box_lo_[d]) { side[d] = (q[d] - box_lo_[d]) * (q[d] - box_lo_[d]); outside += side[d]; }
      if (q[d] > box_hi_[d]) { side[d] = (q[d] - box_hi_[d]) * (q[d] - box_hi_[d]); outside += side[d]; }
    }
    walk(0, q, r2, outside, side, hits);
  }

 private:
  struct Node {
    int kid[2];              // -1: leaf
    size_t first, last;      // leaf: its run of perm_
    int axis;
    float low_end, high_start;       // the low child's largest / the high child's smallest coordinate on `axis` (box extents)
  };
  const float* p_;
  std::vector<size_t> perm_;
  std::vector<Node> nodes_;
  float box_lo_[3], box_hi_[3];

  float at(size_t k, int d) const { return p_[3 * perm_[k] + d]; }

  void range_of(size_t first, size_t last, int d, float& lo, float& hi) const {
    lo = hi = at(first, d);
    for (size_t k = first + 1; k < last; k++) {
      const float v = at(k, d);
      if (v < lo) lo = v;
      if (v > hi) hi = v;
    }
  }

  // (lo, hi): the box of the range on entry (as cut by the ancestors), the tight box of its points on return
  int build(size_t first, size_t last, float (&lo)[3], float (&hi)[3]) {
    const int me = (int)nodes_.size();
    nodes_.push_back(Node{{-1, -1}, first, last, 0, 0.f, 0.f});
    const size_t count = last - first;
    if (count <= 10) {
      for (int d = 0; d < 3; d++) range_of(first, last, d, lo[d], hi[d]);
      return me;
    }
    // the axis
    const float eps = 0.00001f;
    float widest = hi[0] - lo[0];
    for (int d = 1; d < 3; d++) widest = std::max(widest, hi[d] - lo[d]);
    int axis = 0;
    float best_spread = -1.f;
    for (int d = 0; d < 3; d++)
      if (hi[d] - lo[d] > (1 - eps) * widest) {
        float mn, mx;
        range_of(first, last, d, mn, mx);
        if (mx - mn > best_spread) { axis = d; best_spread = mx - mn; }
      }
    float mn, mx;
    range_of(first, last, axis, mn, mx);
    const float mid = (lo[axis] + hi[axis]) / 2;
    const float cut = mid < mn ? mn : (mid > mx ? mx : mid);
    // the two exchange sweeps (unsigned positions as in the reference: a right end that reaches 0 stops a sweep)
    size_t l = 0, r = count - 1;
    size_t* run = perm_.data() + first;
    auto val = [&](size_t k) { return p_[3 * run[k] + axis]; };
    for (;;) {
      while (l <= r && val(l) < cut) ++l;
      while (r && l <= r && val(r) >= cut) --r;
      if (l > r || !r) break;
      std::swap(run[l], run[r]);
      ++l;
      --r;
    }
    const size_t below = l;
    r = count - 1;
    for (;;) {
      while (l <= r && val(l) <= cut) ++l;
      while (r && l <= r && val(r) > cut) --r;
      if (l > r || !r) break;
      std::swap(run[l], run[r]);
      ++l;
      --r;
    }
    const size_t upto = l, half = count / 2;
    const size_t split = below > half ? below : (upto < half ? upto : half);
    float llo[3], lhi[3], rlo[3], rhi[3];
    for (int d = 0; d < 3; d++) { llo[d] = rlo[d] = lo[d]; lhi[d] = rhi[d] = hi[d]; }
    lhi[axis] = cut;
    rlo[axis] = cut;
    const int k0 = build(first, first + split, llo, lhi);
    const int k1 = build(first + split, last, rlo, rhi);
    Node& nd = nodes_[(size_t)me];
    nd.kid[0] = k0;
    nd.kid[1] = k1;
    nd.axis = axis;
    nd.low_end = lhi[axis];
    nd.high_start = rlo[axis];
    for (int d = 0; d < 3; d++) { lo[d] = std::min(llo[d], rlo[d]); hi[d] = std::max(lhi[d], rhi[d]); }
    return me;
  }

  void walk(int n, const float* q, float r2, float outside, float (&side)[3], std::vector<Hit>& hits) const {
    const Node& nd = nodes_[(size_t)n];
    if (nd.kid[0] < 0) {
      for (size_t k = nd.first; k < nd.last; k++) {
        const size_t j = perm_[k];
        const float dx = q[0] - p_[3 * j], dy = q[1] - p_[3 * j + 1], dz = q[2] - p_[3 * j + 2];
        float d2 = dx * dx;
        d2 += dy * dy;
        d2 += dz * dz;
        if (d2 < r2) hits.push_back(Hit{(int64_t)j, d2});
      }
      return;
    }
    const float v = q[nd.axis], to_low = v - nd.low_end, to_high = v - nd.high_start;
    const bool low_first = (to_low + to_high) < 0;
    const float gap = low_first ? to_high * to_high : to_low * to_low;
    walk(nd.kid[low_first ? 0 : 1], q, r2, outside, side, hits);
    const float kept = side[nd.axis];
    outside = outside + gap - kept;
    side[nd.axis] = gap;
    if (outside * 1.0f <= r2) walk(nd.kid[low_first ? 1 : 0], q, r2, outside, side, hits);
    side[nd.axis] = kept;
  }
};

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
