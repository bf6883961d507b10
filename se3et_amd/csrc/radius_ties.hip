// A2, exact ties: the reference's ORDER of exactly tied distances in the device radius search.
//
// The reference's row = the first `limit` entries of ALL in-radius matches of the query, collected while it walks a k-d tree
// (nanoflann KDTreeSingleIndexAdaptor, leaf size 10: extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:29-90,
// extra/nanoflann/nanoflann.hpp:857-1002,1348-1407) and then std::sort-ed on the distance ALONE (nanoflann.hpp:1286-1287; libstdc++'s
// introsort, unstable).  Where two matches have exactly the same float32 distance the result therefore depends on the walk order and on
// the sort's exchanges -- and a neighbour limit that cuts through such a group keeps another MEMBER than the index order of
// csrc/radius_neighbors.hip.  On jittered synthetic clouds no row holds a tie; on real scans (coordinates on a millimetre lattice: 57 %
// of the stage-0 rows of data/demo) most do.
//
// The search kernels flag every row whose kept entries -- or the entry right behind the cut -- hold an exact tie (a list of row numbers
// behind a device counter).  For the flagged rows only, this file reproduces the reference:
//   host    se3_kdtree_build_host: the reference's tree of every support cloud (csrc/kdtree_ref.h, the restatement the host-memory
//           twin csrc/host_ext.hip uses), flattened into index arrays; integer / structural preprocessing of 12 bytes per point that
//           the caller uploads;
//   device  radius_tie_rows_kernel: ONE LANE PER FLAGGED ROW walks the flattened tree with the reference's arithmetic (near child first,
//           far child while the box distance is <= r^2, leaves in stored order, d2 = dx*dx + dy*dy + dz*dz unfused, strict d2 < r^2),
//           appends the matches to its strip of a scratch buffer in walk order, sorts them with a restatement of libstdc++'s std::sort
//           (introsort: median-of-three to the front, unguarded Hoare partition, depth limit 2 floor(log2 n) with heap sort behind it,
//           threshold 16, final insertion sort) comparing the distance alone, and rewrites its row of the table.
// SE3_EXACT_FP: compiled with -ffp-contract=off (se3et_amd/build.py).
#define SE3_EXACT_FP 1
#include "common.h"
#include "kdtree_ref.h"
#include <string.h>
#include <memory>
#include <thread>

namespace {

// ---- flattened tree (host and device) --------------------------------------------------------------------------------------------------
struct KdCloud {            // 48 bytes
  int32_t node_base;        // first node of this cloud in the node array (its root)
  int32_t perm_base;        // first entry of this cloud in the permutation array (= its first support row)
  int32_t n_nodes, n_points;
  float lo[3], hi[3];       // the root box
  int32_t pad[2];
};
struct KdNode {             // 32 bytes
  int32_t kid0, kid1;       // node indices relative to node_base; kid0 < 0: leaf
  int32_t first, last;      // leaf: its run of the permutation, relative to perm_base
  int32_t axis;
  float low_end, high_start;
  int32_t pad;
};
struct KdHeader {           // 64 bytes
  int32_t magic, batch;
  int64_t total_nodes, total_points;
  int64_t off_clouds, off_perm, off_nodes;      // byte offsets from the start of the buffer
  int64_t pad[2];
};
constexpr int32_t kMagic = 0x6b645433;
static_assert(sizeof(KdCloud) == 48 && sizeof(KdNode) == 32 && sizeof(KdHeader) == 64, "flattened k-d tree layout");

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Batch {
  int64_t q_start[SE3_MAX_BATCH], q_count[SE3_MAX_BATCH], s_start[SE3_MAX_BATCH], s_count[SE3_MAX_BATCH];
  int n;
};

// ---- the device side -----------------------------------------------------------------------------------------------------------------
// the lane's strip of the scratch buffer: element k of row r at k * stride + r (lanes of a wave touch neighbouring words)
struct Strip {
  unsigned long long* base;
  int64_t stride;
  __host__ __device__ __forceinline__ unsigned long long get(int k) const { return base[(int64_t)k * stride]; }
  __host__ __device__ __forceinline__ void set(int k, unsigned long long v) const { base[(int64_t)k * stride] = v; }
};
// the comparison of the reference's sort: the distance alone (d2 >= 0: float order = order of the bit patterns)
__host__ __device__ __forceinline__ bool less_d2(unsigned long long a, unsigned long long b) { return (unsigned)(a >> 32) < (unsigned)(b >> 32); }

// libstdc++ std::__adjust_heap + std::__push_heap (bits/stl_heap.h), on the strip range [first, first + len)
__host__ __device__ void adjust_heap(const Strip& s, int first, int hole, int len, unsigned long long value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less_d2(s.get(first + child), s.get(first + child - 1))) child--;
    s.set(first + hole, s.get(first + child));
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    s.set(first + hole, s.get(first + child - 1));
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && less_d2(s.get(first + parent), value)) {
    s.set(first + hole, s.get(first + parent));
    hole = parent;
    parent = (hole - 1) / 2;
  }
  s.set(first + hole, value);
}
// std::__partial_sort(first, last, last) = __heap_select (make_heap; nothing behind `middle`) + __sort_heap
__host__ __device__ void heap_sort(const Strip& s, int first, int last) {
  const int len = last - first;
  if (len >= 2) {
    for (int parent = (len - 2) / 2;; parent--) {
      adjust_heap(s, first, parent, len, s.get(first + parent));
      if (parent == 0) break;
    }
  }
  for (int end = last; end - first > 1;) {
    --end;
    const unsigned long long value = s.get(end);
    s.set(end, s.get(first));
    adjust_heap(s, first, 0, end - first, value);
  }
}
__host__ __device__ __forceinline__ void swap_at(const Strip& s, int a, int b) {
  const unsigned long long x = s.get(a), y = s.get(b);
  s.set(a, y);
  s.set(b, x);
}
// std::__unguarded_linear_insert
__host__ __device__ __forceinline__ void linear_insert(const Strip& s, int last) {
  const unsigned long long val = s.get(last);
  int next = last - 1;
  while (less_d2(val, s.get(next))) {
    s.set(last, s.get(next));
    last = next;
    --next;
  }
  s.set(last, val);
}
// std::__insertion_sort
__host__ __device__ void insertion_sort(const Strip& s, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less_d2(s.get(i), s.get(first))) {
      const unsigned long long val = s.get(i);
      for (int k = i; k > first; k--) s.set(k, s.get(k - 1));          // move_backward(first, i, i + 1)
      s.set(first, val);
    } else {
      linear_insert(s, i);
    }
  }
}
// A lane's explicit stacks live outside its registers / private memory (the kernel keeps them in LDS: a per-lane array of a few hundred
// bytes is scratch memory, and a dispatch with that much scratch stalled the HOST for ~10 ms per launch while the runtime resized the scratch
// pool -- 100 ms per pyramid of the real pair for 5 ms of kernels).  Slot d of the lane at [d * stride].
struct Stacks {
  int* i;          // walk: node | stage << 30          sort: first | last << 16
  float* a;        // walk: the frame's `outside`       sort: depth (integer bits)
  float* b;        // walk: the saved `side` value
  int stride;
};
constexpr int kMaxDepth = 48;

// std::sort(first, last, comp) of libstdc++ (bits/stl_algo.h: __sort -> __introsort_loop + __final_insertion_sort)
__host__ __device__ void std_sort(const Strip& s, int n, const Stacks& K) {
  if (n <= 0) return;
  constexpr int kThreshold = 16;
  int depth0 = 0;
  while ((2 << depth0) <= n) depth0++;                                  // floor(log2 n)
  // the loop recurses on the right part and iterates on the left; the two parts are disjoint, so an explicit stack in any order
  // performs the same exchanges (n < 65536: first | last << 16; at most 2 floor(log2 n) <= 30 pending parts)
  int sp = 1;
  K.i[0] = 0 | (n << 16);
  K.a[0] = __builtin_bit_cast(float, 2 * depth0);
  while (sp > 0) {
    --sp;
    const int packed = K.i[sp * K.stride];
    int first = packed & 0xffff, last = (int)((unsigned)packed >> 16), depth = __builtin_bit_cast(int, K.a[sp * K.stride]);
    while (last - first > kThreshold) {
      if (depth == 0) {
        heap_sort(s, first, last);
        break;
      }
      --depth;
      // __unguarded_partition_pivot: median of (first + 1, mid, last - 1) to *first, then the unguarded Hoare partition of [first + 1, last)
      const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
      const unsigned long long va = s.get(a), vb = s.get(b), vc = s.get(c);
      int med;
      if (less_d2(va, vb)) med = less_d2(vb, vc) ? b : (less_d2(va, vc) ? c : a);
      else med = less_d2(va, vc) ? a : (less_d2(vb, vc) ? c : b);
      swap_at(s, first, med);
      const unsigned long long pivot = s.get(first);
      int lo = first + 1, hi = last;
      for (;;) {
        while (less_d2(s.get(lo), pivot)) ++lo;
        --hi;
        while (less_d2(pivot, s.get(hi))) --hi;
        if (!(lo < hi)) break;
        swap_at(s, lo, hi);
        ++lo;
      }
      if (sp < kMaxDepth) {
        K.i[sp * K.stride] = lo | (last << 16);
        K.a[sp * K.stride] = __builtin_bit_cast(float, depth);
        sp++;
      }
      last = lo;
    }
  }
  if (n > kThreshold) {                                                 // __final_insertion_sort
    insertion_sort(s, 0, kThreshold);
    for (int i = kThreshold; i < n; ++i) linear_insert(s, i);           // __unguarded_insertion_sort
  } else {
    insertion_sort(s, 0, n);
  }
}


// One flagged row: walk, collect, sort, rewrite.  Returns false when the strip (max_hits) or the walk stack overflowed (the row is left as
// it is).  Host-callable as well: se3_debug_radius_tie_order_host runs exactly this code on the CPU (tests/test_radius_ties_cpu.py checks
// it against the host twin's std::sort there, where no GPU is needed).
__host__ __device__ bool tie_row(const float* __restrict__ q, const float* __restrict__ s, const Batch& bt, const unsigned char* __restrict__ tree,
                                 float r2, int limit, int64_t row, int max_hits, const Strip& strip, const Stacks& K, int64_t ns_total,
                                 int64_t* __restrict__ out) {
  int b = 0;
  while (b + 1 < bt.n && row >= bt.q_start[b] + bt.q_count[b]) b++;
  const KdHeader* H = reinterpret_cast<const KdHeader*>(tree);
  const KdCloud cl = reinterpret_cast<const KdCloud*>(tree + H->off_clouds)[b];
  const int32_t* perm = reinterpret_cast<const int32_t*>(tree + H->off_perm) + cl.perm_base;
  const KdNode* nodes = reinterpret_cast<const KdNode*>(tree + H->off_nodes) + cl.node_base;
  const float* sp = s + 3 * bt.s_start[b];
  const float qv[3] = {q[3 * row], q[3 * row + 1], q[3 * row + 2]};
  int n = 0;
  bool over = false;
  if (cl.n_points > 0) {
    float side[3] = {0.f, 0.f, 0.f}, outside = 0.f;
    for (int d = 0; d < 3; d++) {
      if (qv[d] < cl.lo[d]) { side[d] = (qv[d] - cl.lo[d]) * (qv[d] - cl.lo[d]); outside += side[d]; }
      if (qv[d] > cl.hi[d]) { side[d] = (qv[d] - cl.hi[d]) * (qv[d] - cl.hi[d]); outside += side[d]; }
    }
    // frames of the walk: (node, stage) in K.i, the box distance `outside` the node was entered with in K.a, the saved side value in K.b
    int sp_ = 1;
    K.i[0] = 0;
    K.a[0] = outside;
    while (sp_ > 0) {
      const int top = (sp_ - 1) * K.stride;
      const int word = K.i[top], node = word & 0x3fffffff, stage = (int)((unsigned)word >> 30);
      const float f_outside = K.a[top];
      const KdNode nd = nodes[node];
      if (nd.kid0 < 0) {
        for (int k = nd.first; k < nd.last; k++) {
          const int j = perm[k];
          const float dx = qv[0] - sp[3 * j], dy = qv[1] - sp[3 * j + 1], dz = qv[2] - sp[3 * j + 2];
          float d2 = dx * dx;
          d2 += dy * dy;
          d2 += dz * dz;
          if (d2 < r2) {
            if (n < max_hits) strip.set(n, ((unsigned long long)__builtin_bit_cast(unsigned, d2) << 32) | (unsigned)(bt.s_start[b] + j));
            else over = true;
            n += n < max_hits ? 1 : 0;
          }
        }
        sp_--;
        continue;
      }
      const float v = nd.axis == 0 ? qv[0] : (nd.axis == 1 ? qv[1] : qv[2]);
      const float to_low = v - nd.low_end, to_high = v - nd.high_start;
      const bool low_first = (to_low + to_high) < 0;
      const float gap = low_first ? to_high * to_high : to_low * to_low;
      if (stage == 0) {
        K.i[top] = node | (1 << 30);
        if (sp_ < kMaxDepth) { K.i[sp_ * K.stride] = low_first ? nd.kid0 : nd.kid1; K.a[sp_ * K.stride] = f_outside; sp_++; } else { over = true; sp_--; }
      } else if (stage == 1) {
        const float kept = nd.axis == 0 ? side[0] : (nd.axis == 1 ? side[1] : side[2]);
        const float far_outside = f_outside + gap - kept;
        if (far_outside * 1.0f <= r2) {
          K.b[top] = kept;
          K.i[top] = node | (2 << 30);
          if (nd.axis == 0) side[0] = gap; else if (nd.axis == 1) side[1] = gap; else side[2] = gap;
          if (sp_ < kMaxDepth) { K.i[sp_ * K.stride] = low_first ? nd.kid1 : nd.kid0; K.a[sp_ * K.stride] = far_outside; sp_++; } else { over = true; sp_--; }
        } else {
          sp_--;
        }
      } else {
        const float kept = K.b[top];
        if (nd.axis == 0) side[0] = kept; else if (nd.axis == 1) side[1] = kept; else side[2] = kept;
        sp_--;
      }
    }
  }
  if (n >= 65536) over = true;                 // (the sort's stack packs positions into 16 bits)
  if (over) return false;                      // (cannot happen with max_hits = the search's own largest count: the row keeps its index order)
  std_sort(strip, n, K);
  for (int k = 0; k < limit; k++) out[row * limit + k] = k < n ? (int64_t)(unsigned)(strip.get(k) & 0xffffffffull) : ns_total;
  return true;
}

__global__ __launch_bounds__(64) void radius_tie_rows_kernel(const float* __restrict__ q, const float* __restrict__ s, Batch bt,
                                                             const unsigned char* __restrict__ tree, float r2, int limit,
                                                             const int32_t* __restrict__ rows, int64_t num_rows, int max_hits,
                                                             unsigned long long* __restrict__ scratch, int64_t ns_total,
                                                             int64_t* __restrict__ out, int32_t* __restrict__ overflow) {
  __shared__ int stk_i[kMaxDepth * 64];
  __shared__ float stk_a[kMaxDepth * 64], stk_b[kMaxDepth * 64];
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= num_rows) return;
  const Stacks K{stk_i + threadIdx.x, stk_a + threadIdx.x, stk_b + threadIdx.x, 64};
  if (!tie_row(q, s, bt, tree, r2, limit, rows[r], max_hits, Strip{scratch + r, num_rows}, K, ns_total, out)) atomicAdd(overflow, 1);
}

int fill_batch(Batch* bt, const int64_t* q_len, const int64_t* s_len, int batch, int64_t nq, int64_t ns) {
  int64_t qs = 0, ss = 0;
  bt->n = batch;
  for (int b = 0; b < batch; b++) {
    if (q_len[b] < 0 || s_len[b] < 0) return 1;
    bt->q_start[b] = qs; bt->q_count[b] = q_len[b];
    bt->s_start[b] = ss; bt->s_count[b] = s_len[b];
    qs += q_len[b]; ss += s_len[b];
  }
  return qs != nq || ss != ns;
}

}  // namespace

// Upper bound of the flattened trees of `batch` clouds with `ns` points in total (a node per point is never reached: a split leaves
// at least one point on each side, so there are at most 2 ns - batch nodes).
extern "C" size_t se3_kdtree_max_bytes(int64_t ns, int batch) {
  if (ns < 0 || batch < 1 || batch > SE3_MAX_BATCH) return 0;
  return align256(sizeof(KdHeader)) + align256(sizeof(KdCloud) * (size_t)batch) + align256(sizeof(int32_t) * (size_t)(ns + 1)) +
         align256(sizeof(KdNode) * (size_t)(2 * ns + batch));
}

// HOST memory in, HOST memory out: the reference's k-d tree of every support cloud (nanoflann.hpp:857-1002 as restated in csrc/kdtree_ref.h),
// flattened.  *used_bytes <= capacity is what has to be uploaded.
extern "C" int se3_kdtree_build_host(const float* s_points_host, int64_t ns, const int64_t* s_lengths_host, int batch, void* tree_host,
                                     size_t capacity, size_t* used_bytes) {
  SE3_REQUIRE(s_points_host && s_lengths_host && tree_host && used_bytes, SE3_ERR_INVALID_ARG, "kdtree_build_host: null pointer");
  SE3_REQUIRE(batch >= 1 && batch <= SE3_MAX_BATCH && ns >= 0 && ns < (1ll << 30), SE3_ERR_INVALID_ARG, "kdtree_build_host: bad batch / size");
  SE3_REQUIRE(capacity >= se3_kdtree_max_bytes(ns, batch), SE3_ERR_WORKSPACE, "kdtree_build_host: buffer too small");
  unsigned char* base = (unsigned char*)tree_host;
  KdHeader H{};
  H.magic = kMagic;
  H.batch = batch;
  H.off_clouds = (int64_t)align256(sizeof(KdHeader));
  H.off_perm = H.off_clouds + (int64_t)align256(sizeof(KdCloud) * (size_t)batch);
  H.off_nodes = H.off_perm + (int64_t)align256(sizeof(int32_t) * (size_t)(ns + 1));
  KdCloud* clouds = (KdCloud*)(base + H.off_clouds);
  int32_t* perm = (int32_t*)(base + H.off_perm);
  KdNode* nodes = (KdNode*)(base + H.off_nodes);
  // the trees of the clouds are independent: built by one host thread each (at most 16 at a time; small batches of small clouds stay on the
  // caller's thread), flattened in cloud order afterwards
  std::vector<int64_t> start((size_t)batch + 1, 0);
  for (int b = 0; b < batch; b++) {
    SE3_REQUIRE(s_lengths_host[b] >= 0 && start[(size_t)b] + s_lengths_host[b] <= ns, SE3_ERR_INVALID_ARG, "kdtree_build_host: lengths exceed the point count");
    start[(size_t)b + 1] = start[(size_t)b] + s_lengths_host[b];
  }
  std::vector<std::unique_ptr<se3_kd::KdTree>> trees((size_t)batch);
  auto build_one = [&](int b) {
    const int64_t n = s_lengths_host[b];
    if (n > 0) trees[(size_t)b].reset(new se3_kd::KdTree(s_points_host + 3 * start[(size_t)b], n));
  };
  if (batch > 1 && ns >= 4096) {
    for (int b0 = 0; b0 < batch; b0 += 16) {
      std::vector<std::thread> pool;
      for (int b = b0; b < batch && b < b0 + 16; b++) pool.emplace_back(build_one, b);
      for (auto& t : pool) t.join();
    }
  } else {
    for (int b = 0; b < batch; b++) build_one(b);
  }
  int64_t s0 = 0, n0 = 0;
  for (int b = 0; b < batch; b++) {
    const int64_t n = s_lengths_host[b];
    KdCloud c{};
    c.node_base = (int32_t)n0;
    c.perm_base = (int32_t)s0;
    c.n_points = (int32_t)n;
    if (n > 0) {
      const se3_kd::KdTree& tree = *trees[(size_t)b];
      const auto& tn = tree.nodes();
      const auto& tp = tree.perm();
      c.n_nodes = (int32_t)tn.size();
      for (int d = 0; d < 3; d++) { c.lo[d] = tree.box_lo()[d]; c.hi[d] = tree.box_hi()[d]; }
      for (size_t k = 0; k < tp.size(); k++) perm[s0 + (int64_t)k] = (int32_t)tp[k];
      for (size_t k = 0; k < tn.size(); k++) {
        KdNode o{};
        o.kid0 = tn[k].kid[0]; o.kid1 = tn[k].kid[1];
        o.first = (int32_t)tn[k].first; o.last = (int32_t)tn[k].last;
        o.axis = tn[k].axis; o.low_end = tn[k].low_end; o.high_start = tn[k].high_start;
        nodes[n0 + (int64_t)k] = o;
      }
    }
    clouds[b] = c;
    s0 += n;
    n0 += c.n_nodes;
  }
  SE3_REQUIRE(s0 == ns, SE3_ERR_INVALID_ARG, "kdtree_build_host: lengths do not sum to ns");
  H.total_nodes = n0;
  H.total_points = ns;
  memcpy(base, &H, sizeof(H));
  *used_bytes = (size_t)H.off_nodes + sizeof(KdNode) * (size_t)n0;
  return SE3_OK;
}

extern "C" size_t se3_radius_tie_scratch_bytes(int64_t num_rows, int max_hits) {
  if (num_rows < 0 || max_hits < 0) return 0;
  return align256(sizeof(unsigned long long) * (size_t)num_rows * (size_t)(max_hits > 0 ? max_hits : 1)) + 256;
}

// Rewrites the rows `tie_rows` (DEVICE int32 list, num_tie_rows entries: what se3_radius_neighbors_ties / _grid_ties flagged) of the
// (nq, limit) table with the reference's order of exactly tied distances.  tree_dev: the upload of se3_kdtree_build_host's buffer for the
// SAME support clouds; max_hits: at least the largest in-radius count of the search (its max_count output); scratch: DEVICE,
// se3_radius_tie_scratch_bytes(num_tie_rows, max_hits).  The last 4 bytes of the scratch count rows that overflowed (must stay 0).
extern "C" int se3_radius_neighbors_tie_order(const float* q_points, int64_t nq, const float* s_points, int64_t ns, const int64_t* q_lengths_host,
                                              const int64_t* s_lengths_host, int batch, const void* tree_dev, float radius, int limit,
                                              const int32_t* tie_rows, int64_t num_tie_rows, int max_hits, void* scratch, size_t scratch_bytes,
                                              int64_t* neighbors, void* stream) {
  SE3_REQUIRE(q_points && s_points && q_lengths_host && s_lengths_host && tree_dev && tie_rows && scratch && neighbors, SE3_ERR_INVALID_ARG,
              "radius_neighbors_tie_order: null pointer");
  SE3_REQUIRE(batch >= 1 && batch <= SE3_MAX_BATCH && limit >= 1 && limit <= SE3_MAX_NEIGHBOR_LIMIT && max_hits >= 1, SE3_ERR_INVALID_ARG,
              "radius_neighbors_tie_order: bad batch / limit / max_hits");
  SE3_REQUIRE(scratch_bytes >= se3_radius_tie_scratch_bytes(num_tie_rows, max_hits), SE3_ERR_WORKSPACE, "radius_neighbors_tie_order: scratch too small");
  Batch bt;
  SE3_REQUIRE(fill_batch(&bt, q_lengths_host, s_lengths_host, batch, nq, ns) == 0, SE3_ERR_INVALID_ARG,
              "radius_neighbors_tie_order: lengths do not sum to the sizes");
  if (num_tie_rows <= 0) return SE3_OK;
  hipStream_t st = (hipStream_t)stream;
  int32_t* overflow = (int32_t*)((char*)scratch + se3_radius_tie_scratch_bytes(num_tie_rows, max_hits) - 256);
  if (hipMemsetAsync(overflow, 0, sizeof(int32_t), st) != hipSuccess) {
    se3_set_error("radius_neighbors_tie_order: memset failed");
    return SE3_ERR_LAUNCH;
  }
  radius_tie_rows_kernel<<<(unsigned)se3_cdiv(num_tie_rows, 64), 64, 0, st>>>(q_points, s_points, bt, (const unsigned char*)tree_dev, radius * radius, limit,
                                                                               tie_rows, num_tie_rows, max_hits, (unsigned long long*)scratch, ns,
                                                                               neighbors, overflow);
  SE3_CHECK_LAUNCH("radius_neighbors_tie_order");
  return SE3_OK;
}

// The code of the device pass on HOST memory (every pointer a host pointer, tree_host = se3_kdtree_build_host's buffer): the CPU check of the
// walk and of the std::sort restatement (tests/test_radius_ties_cpu.py).  Returns the number of rows that overflowed through *overflowed.
extern "C" int se3_debug_radius_tie_order_host(const float* q_points, int64_t nq, const float* s_points, int64_t ns, const int64_t* q_lengths,
                                               const int64_t* s_lengths, int batch, const void* tree_host, float radius, int limit,
                                               const int32_t* tie_rows, int64_t num_tie_rows, int max_hits, int64_t* neighbors, int* overflowed) {
  SE3_REQUIRE(q_points && s_points && q_lengths && s_lengths && tree_host && neighbors && overflowed && (tie_rows || num_tie_rows == 0),
              SE3_ERR_INVALID_ARG, "debug_radius_tie_order_host: null pointer");
  SE3_REQUIRE(batch >= 1 && batch <= SE3_MAX_BATCH && limit >= 1 && max_hits >= 1, SE3_ERR_INVALID_ARG, "debug_radius_tie_order_host: bad arguments");
  Batch bt;
  SE3_REQUIRE(fill_batch(&bt, q_lengths, s_lengths, batch, nq, ns) == 0, SE3_ERR_INVALID_ARG, "debug_radius_tie_order_host: lengths do not sum to the sizes");
  std::vector<unsigned long long> strip((size_t)max_hits);
  int si[kMaxDepth];
  float sa[kMaxDepth], sb[kMaxDepth];
  const Stacks K{si, sa, sb, 1};
  *overflowed = 0;
  for (int64_t r = 0; r < num_tie_rows; r++)
    if (!tie_row(q_points, s_points, bt, (const unsigned char*)tree_host, radius * radius, limit, tie_rows[r], max_hits, Strip{strip.data(), 1}, K, ns, neighbors))
      *overflowed += 1;
  return SE3_OK;
}

// The sort restatement alone against libstdc++ (host): mode 0 -- std_sort against std::sort, mode 1 -- heap_sort against
// std::partial_sort(first, last, last) (what the introsort loop calls once its depth limit is spent), both on copies of keys[0..n) with the
// comparison of the search (the high 32 bits alone).  Returns the number of positions where the two results differ (0 = identical), or -1.
extern "C" int64_t se3_debug_std_sort_host(const unsigned long long* keys, int64_t n, int mode) {
  if (keys == nullptr || n < 0 || n >= 65536) return -1;
  std::vector<unsigned long long> a(keys, keys + n), b(keys, keys + n);
  auto cmp = [](unsigned long long x, unsigned long long y) { return (unsigned)(x >> 32) < (unsigned)(y >> 32); };
  const Strip strip{a.data(), 1};
  int si[kMaxDepth];
  float sa[kMaxDepth], sb[kMaxDepth];
  if (mode == 0) {
    std_sort(strip, (int)n, Stacks{si, sa, sb, 1});
    std::sort(b.begin(), b.end(), cmp);
  } else {
    heap_sort(strip, 0, (int)n);
    std::partial_sort(b.begin(), b.end(), b.end(), cmp);
  }
  int64_t diff = 0;
  for (int64_t i = 0; i < n; i++) diff += a[(size_t)i] != b[(size_t)i];
  return diff;
}
