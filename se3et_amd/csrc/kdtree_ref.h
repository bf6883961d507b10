// The reference's k-d tree, restated (host side).  Shared by csrc/host_ext.hip (the host-memory geotransformer.ext twin: build + walk + sort on
// the CPU) and csrc/radius_ties.hip (the device search: the tree is built here, flattened, and WALKED ON THE GPU for the rows whose result
// depends on the order of exactly tied distances).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <algorithm>
#include <vector>

namespace se3_kd {

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

 public:
  struct Node {
    int kid[2];              // -1: leaf
    size_t first, last;      // leaf: its run of perm_
    int axis;
    float low_end, high_start;       // the low child's largest / the high child's smallest coordinate on `axis` (box extents)
  };
  const std::vector<Node>& nodes() const { return nodes_; }
  const std::vector<size_t>& perm() const { return perm_; }
  const float* box_lo() const { return box_lo_; }
  const float* box_hi() const { return box_hi_; }

 private:
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

}  // namespace se3_kd
