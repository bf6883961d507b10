// B1, round 5: the fused KPConvInterSO3 kernel with UNION-STAGED producers.
//
// Reference: geotransformer/modules/e2pn/blocks_epn.py:334-390 (feat_gather_by_perm), :454-546 (forward); arithmetic and operand
// formats as csrc/kpconv_mfma.hip (16 kernel-point orbits, f16 hi + lo pieces, three products in f32; the same neighbour table, the same
// weight fragments, the same tile image, the same consumer waves).  What changes is how the producers get the neighbours' feature rows.
//
// csrc/kpconv_mfma.hip: every producer lane gathers ITS column of its point's 32-40 neighbour rows with 8-byte requests -- 464 vector-memory
// instructions per 16-channel step and compute unit, the kernel's bound (one request per ~16-20 cycles whatever its width), and the same
// row is fetched again by every point of the tile that has it as a neighbour (3.9 GB fetched per step for 0.5 GB of features).
// Here a workgroup owns 16 points that are SPATIAL neighbours (se3_kpconv_union_plan: 16 consecutive points of a per-cloud Morton order --
// tile membership only; no tensor is reordered, every point still walks its own table and writes its own output row).  Their neighbour
// lists overlap: 46 / 84 / 115 distinct rows per tile on average at the bench shape instead of 126 / 304 / 503 list entries
// (tools/r5/union_sizes.py).  Per 8-channel chunk:
//   loaders   the tile's DISTINCT support rows ("union", <= 128) are read once, whole: 192 contiguous bytes per row and chunk, 16 B per lane
//             (<= 30 wave requests per chunk instead of 232), split into f16 hi / lo and left in LDS as MFMA B fragments (B image)
//   gather    H[p, o, (a, c)] = sum_u A_p[o, u] X[u, (a, c)]: the point's orbit weights scattered over the union's index u as the A operand
//             (zero where u is not a neighbour of p; built ONCE per tile -- it does not depend on the channel -- and kept in registers),
//             the B image shared by all 16 points: v_mfma_f32_16x16x32_f16, three products.  No per-lane gather, no per-point split of x.
//   result    split into f16 hi / lo and stored into the tile image the consumers read, exactly as before.
// Two tile images instead of three (a producer step is one chunk, not a pair), two B images: 148.9 KB of LDS.
// A tile whose union exceeds 128 rows is processed in several passes over halves of its points (the plan's sub-tiles), all passes adding
// into the same accumulators (rows of points outside a pass are zero): rare at the bench shapes except the coarsest strided layer.
// Summation order: a point's neighbours are added in the order of their support row numbers inside 32-row K-steps; the plan is a pure
// function of (order, table), so runs are bit-identical; results differ from csrc/kpconv_mfma.hip in the last bits (f32 association).
// Non-finite features: 0 x Inf = NaN reaches every point of a tile whose union holds the row, not only the row's neighbours.
#include "common.h"
#include "kpconv_sums.h"

namespace {

using namespace kpsum;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kHeaderB = 256;                // weight-fragment buffer header (csrc/kpconv_mfma.hip)
#ifndef SE3_UCAP
#define SE3_UCAP 128
#endif
constexpr int kUCap = SE3_UCAP;                   // distinct support rows per (sub-)tile (160 measured: 64 -> 80 registers of A fragments, 27 instead of 10 spilled, 13 KB
                                             // more LDS -- the narrow layers the policy gives this kernel ran 0.401 / 0.283 / 0.544 ms instead of 0.345 / 0.236 / 0.526;
                                             // 96 (-DSE3_UCAP=96, tools/r5/build_variant.sh): the 32-wide layers another 3-4 % faster, the 64-wide
                                             // ones 20 % slower -- 1.27 instead of 1.01 passes per group)
constexpr int kKSM = kUCap / 32;             // K32-steps of the gather product
constexpr int kBFragB = 1024;                // one B fragment: 64 lanes x 16 B
constexpr int kBImgB = 2 * 3 * kKSM * kBFragB;   // [piece][anchor pair][K32-step][lane]: 30 720 B
constexpr int kScrRow = kUCap + 4;           // floats per orbit row of the A-build scratch
constexpr int kMaxSub = 16;

__device__ __forceinline__ int row_point(int i) { return ((i >> 3) << 2) + (i & 3); }
__device__ __forceinline__ int row_rsel(int i) { return (0x96 >> (i >> 2)) & 1; }

// ---- plan --------------------------------------------------------------------------------------------------------------------------------
// [order int32 G*16][nsub int32 G][desc int4 G*16: {first point, points, first union row, union rows}][urow int32 G*16*NNp][loc u8 G*16*NNp]
struct PlanViews {
  int* order;
  int* nsub;
  int4* desc;
  int* urow;
  unsigned char* loc;
};
__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
inline PlanViews plan_views(void* plan, int64_t G, int NNp) {
  unsigned char* p = static_cast<unsigned char*>(plan);
  PlanViews v;
  v.order = reinterpret_cast<int*>(p);
  p += align16((size_t)G * 16 * sizeof(int));
  v.nsub = reinterpret_cast<int*>(p);
  p += align16((size_t)G * sizeof(int));
  v.desc = reinterpret_cast<int4*>(p);
  p += (size_t)G * kMaxSub * sizeof(int4);
  v.urow = reinterpret_cast<int*>(p);
  p += align16((size_t)G * 16 * NNp * sizeof(int));
  v.loc = p;
  return v;
}
inline size_t plan_bytes(int64_t G, int NNp) {
  return align16((size_t)G * 16 * sizeof(int)) + align16((size_t)G * sizeof(int)) + (size_t)G * kMaxSub * sizeof(int4) +
         align16((size_t)G * 16 * NNp * sizeof(int)) + align16((size_t)G * 16 * NNp) + 256;
}

struct NeighborTable {
  const float* hwt;
  const int *nbr, *cnt;
  int NNp;
};
NeighborTable table_views(const void* table, int64_t P, int NN) {
  NeighborTable t;
  t.NNp = NN <= 32 ? 32 : (NN + 39) / 40 * 40;      // (csrc/kpconv_mfma.hip: table_views)
  t.hwt = static_cast<const float*>(table);
  t.nbr = reinterpret_cast<const int*>(t.hwt + (size_t)P * t.NNp * 16);
  t.cnt = t.nbr + (size_t)P * t.NNp;
  return t;
}

// One WAVE per group of 16 order positions: its list entries (row, owner, slot) sorted by row in LDS; per candidate range of owners the
// distinct rows counted; a range whose union fits (or a single point) becomes a sub-tile, any other is halved.
__global__ __launch_bounds__(64) void kpconv_union_plan_kernel(const int* __restrict__ nbr, const int* __restrict__ cnt, int NNp,
                                                               const int* __restrict__ order_in, PlanViews pv) {
  __shared__ unsigned long long keys[1024];
  const int lane = threadIdx.x;
  const int64_t group = blockIdx.x;
  // lanes 0-15: the group's points and their list lengths; exclusive prefix = first entry of every owner
  const int pid = lane < 16 ? order_in[group * 16 + lane] : -1;
  int c = pid >= 0 ? cnt[pid] : 0;
  c = c < 64 ? c : 64;
  if (lane < 16) pv.order[group * 16 + lane] = pid;
  int off = c;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    const int v = __shfl_up(off, o);
    if (lane >= o) off += v;
  }
  const int E = __shfl(off, 15);
  off -= c;
  int n2 = 64;
  while (n2 < E) n2 <<= 1;
  for (int e = lane; e < n2; e += 64) keys[e] = ~0ull;
  __syncthreads();
  for (int i = 0; i < 16; i++) {
    const int ci = __shfl(c, i), oi = __shfl(off, i), pi = __shfl(pid, i);
    if (lane < ci) keys[oi + lane] = ((unsigned long long)(unsigned)nbr[(int64_t)pi * NNp + lane] << 10) | (unsigned)(i << 6) | (unsigned)lane;
  }
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < (n2 >> 1); t += 64) {
        const int a = ((t & ~(j - 1)) << 1) | (t & (j - 1)), b = a | j;
        const bool up = (a & k) == 0;
        const unsigned long long x = keys[a], y = keys[b];
        if ((x > y) == up) {
          keys[a] = y;
          keys[b] = x;
        }
      }
      __syncthreads();
    }
  // uniform work stack of owner ranges (depth <= 5); a lane owns the entries lane * per .. + per - 1
  const int per = n2 >> 6;
  int st_lo[6], st_n[6], sp = 0, ustart = 0, nsub = 0;
  st_lo[0] = 0;
  st_n[0] = 16;
  sp = 1;
  int* urow = pv.urow + group * 16 * NNp;
  unsigned char* loc = pv.loc + group * 16 * NNp;
  while (sp > 0) {
    sp--;
    const int lo = st_lo[sp], n = st_n[sp];
    // pass 1: distinct rows of the range among this lane's entries (an entry is FIRST when no earlier entry of the range has its row)
    int mine = 0;
    for (int q = 0; q < per; q++) {
      const int e = lane * per + q;
      if (e >= E) break;
      const unsigned long long kq = keys[e];
      const int owner = (int)((kq >> 6) & 15);
      if (owner < lo || owner >= lo + n) continue;
      bool first = true;
      for (int j = e - 1; j >= 0; j--) {
        const unsigned long long o = keys[j];
        if ((o >> 10) != (kq >> 10)) break;
        const int ow = (int)((o >> 6) & 15);
        if (ow >= lo && ow < lo + n) {
          first = false;
          break;
        }
      }
      mine += first ? 1 : 0;
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    const int U = __shfl(incl, 63);
    if (U <= kUCap || n == 1) {
      int run = incl - mine;                                             // firsts in front of this lane's entries
      for (int q = 0; q < per; q++) {
        const int e = lane * per + q;
        if (e >= E) break;
        const unsigned long long kq = keys[e];
        const int owner = (int)((kq >> 6) & 15);
        if (owner < lo || owner >= lo + n) continue;
        bool first = true;
        for (int j = e - 1; j >= 0; j--) {
          const unsigned long long o = keys[j];
          if ((o >> 10) != (kq >> 10)) break;
          const int ow = (int)((o >> 6) & 15);
          if (ow >= lo && ow < lo + n) {
            first = false;
            break;
          }
        }
        if (first) {
          if (run < kUCap) urow[ustart + run] = (int)(kq >> 10);
          run++;
        }
        loc[owner * NNp + (int)(kq & 63)] = (unsigned char)(run - 1 < kUCap ? run - 1 : kUCap - 1);
      }
      if (lane == 0) pv.desc[group * kMaxSub + nsub] = make_int4(lo, n, ustart, U < kUCap ? U : kUCap);
      ustart += U < kUCap ? U : kUCap;
      nsub++;
    } else {
      st_lo[sp] = lo + n / 2;
      st_n[sp] = n / 2;
      sp++;
      st_lo[sp] = lo;
      st_n[sp] = n / 2;
      sp++;
    }
  }
  if (lane == 0) pv.nsub[group] = nsub;
}

// ---- spatial order of a stage's points (tile membership) -----------------------------------------------------------------------------
struct CloudOffsets {
  int n;
  int64_t start[SE3_MAX_BATCH + 1];
};
__device__ __forceinline__ unsigned spread10(unsigned v) {      // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
// key = cloud << 32 | 30-bit Morton code of floor(p / cell) (10 bits per axis, wrapping: only the quality of the order depends on it)
__global__ void point_order_keys_kernel(const float* __restrict__ pts, int64_t n, CloudOffsets co, float inv_cell, int64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cloud = 0;
  for (int c = 1; c < co.n; c++) cloud += i >= co.start[c] ? 1 : 0;
  const int x = (int)floorf(pts[3 * i] * inv_cell) + 512, y = (int)floorf(pts[3 * i + 1] * inv_cell) + 512,
            z = (int)floorf(pts[3 * i + 2] * inv_cell) + 512;
  const unsigned m = spread10((unsigned)x) | (spread10((unsigned)y) << 1) | (spread10((unsigned)z) << 2);
  keys[i] = ((int64_t)cloud << 32) | (int64_t)m;
}
// rank r of the sorted keys -> position r + padding in front of its cloud (every cloud starts a new group of 16)
__global__ void point_order_place_kernel(const int64_t* __restrict__ sorted_keys, const int64_t* __restrict__ sorted_idx, int64_t n,
                                         CloudOffsets co, int* __restrict__ order) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int cloud = (int)(sorted_keys[r] >> 32);
  int64_t pad = 0;
  for (int c = 0; c < cloud; c++) {
    const int64_t len = co.start[c + 1] - co.start[c];
    pad += (16 - (len & 15)) & 15;
  }
  order[r + pad] = (int)sorted_idx[r];
}

// The same order in ONE launch for clouds of up to 8192 points: one workgroup per cloud sorts (Morton code, local index) in LDS.
constexpr int kOrderCap = 8192;
struct OrderStages {                   // up to four stages of one pyramid per launch (blockIdx.y)
  const float* pts[4];
  int* order[4];
  float inv_cell[4];
  CloudOffsets co[4];
};
__global__ __launch_bounds__(1024) void point_order_cloud_kernel(const OrderStages S) {
  extern __shared__ __align__(16) unsigned long long okeys[];
  const int cloud = blockIdx.x, tid = threadIdx.x;
  const float* __restrict__ pts = S.pts[0];
  int* __restrict__ order = S.order[0];
  float inv_cell = S.inv_cell[0];
  const CloudOffsets* cop = &S.co[0];
#pragma unroll
  for (int y = 1; y < 4; y++)
    if ((int)blockIdx.y == y) {
      pts = S.pts[y];
      order = S.order[y];
      inv_cell = S.inv_cell[y];
      cop = &S.co[y];
    }
  const CloudOffsets& co = *cop;
  if (cloud >= co.n) return;
  const int64_t start = co.start[cloud];
  const int n = (int)(co.start[cloud + 1] - start);
  int64_t pos0 = 0;                                                      // first order position of this cloud: whole groups of the clouds before
  for (int c = 0; c < cloud; c++) pos0 += ((co.start[c + 1] - co.start[c] + 15) >> 4) << 4;
  int n2 = 64;
  while (n2 < n) n2 <<= 1;
  for (int i = tid; i < n2; i += 1024) {
    unsigned long long k = ~0ull;
    if (i < n) {
      const float* p = pts + 3 * (start + i);
      const int x = (int)floorf(p[0] * inv_cell) + 512, y = (int)floorf(p[1] * inv_cell) + 512, z = (int)floorf(p[2] * inv_cell) + 512;
      const unsigned m = spread10((unsigned)x) | (spread10((unsigned)y) << 1) | (spread10((unsigned)z) << 2);
      k = ((unsigned long long)m << 16) | (unsigned)i;
    }
    okeys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (n2 >> 1); t += 1024) {
        const int a = ((t & ~(j - 1)) << 1) | (t & (j - 1)), b = a | j;
        const bool up = (a & k) == 0;
        const unsigned long long x = okeys[a], y = okeys[b];
        if ((x > y) == up) {
          okeys[a] = y;
          okeys[b] = x;
        }
      }
      __syncthreads();
    }
  const int npad = ((n + 15) >> 4) << 4;
  for (int r = tid; r < npad; r += 1024) order[pos0 + r] = r < n ? (int)(start + (int64_t)(okeys[r] & 0xffffu)) : -1;
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    hi[i] = (_Float16)v[i];
    lo[i] = (_Float16)(v[i] - (float)hi[i]);
  }
}

struct UnionArgs {
  const float* x;
  const float* hwt;
  const int* cnt;
  int NNp;
  const int* order;
  const int* nsub;
  const int4* desc;
  const int* urow;
  const unsigned char* loc;
  const u32x4* Wf;
  const float* hdr;
  int64_t P;
  int64_t G;                 // groups of 16 order positions
  int Cin, Cout;
  float* out;
  float* split_part;
  int* split_count;
  const float* x_amax;       // largest |x| of the tensor (se3_group_norm_apply_amax) or null: kpsum::x_split_scale
  int variant;               // diagnostic bits (se3_debug_set_kpconv_union_variant; results are wrong with any of them set): 1 producers skip the gather
                             // product and the image stores, 2 skip the row loads and the B image, 4 skip the A fragments, 8 consumers skip their MFMAs
};

// PERSISTENT workgroups: workgroup b walks the groups b, b + gridDim.x, ...; an ITEM is one pass (sub-tile) of a group over the workgroup's C
// chunks.  One barrier per step; the producers run one step ahead of the consumers ACROSS items, so a group's prologue (its metadata, the A
// fragments, the first rows) and the previous group's epilogue never stop the other role:
//   producers  item k: P(-1) A fragments (scratch: the tile image the consumers are NOT reading), rows of chunk 0 -> B image 0, rows of chunk 1
//              requested; [KS > 1: one idle step]; P(u), u = 0 .. C - 1: rows of chunk u + 1 -> B image (u + 1) & 1, rows of chunk u + 2 requested,
//              gather product of chunk u from B image u & 1 -> tile image u & 1.  The next item's metadata travels during P(0) (order, plan
//              records), P(C - 2) (row numbers, neighbour counts) and P(C - 1) (rows of its chunk 0).
//   consumers  during P(k, -1): contraction of chunk C - 1 of item k - 1; [KS > 1, idle step: partial sums of the K split -> tile image 1];
//              during P(k, 0): epilogue of item k - 1's group if that was its last pass; during P(k, u), u >= 1: contraction of chunk u - 1.
// Barriers per workgroup with K items: K (C + 1 + [KS > 1]) + 1 + [KS > 1], the same count on both sides.
template <int NCW, int KS, int CT, bool BLK>      // consumer waves: NCW column groups x KS K-split groups; CT column tiles per wave; BLK: x as [point][Cin / 8][6 anchors][8 channels]
__global__ __launch_bounds__(64 * (NCW * KS + 8)) void kpconv_union_kernel(const UnionArgs a) {
  constexpr int NC = NCW * KS;
  constexpr int NPW = 8;
  constexpr int kSPW = kSteps / KS;
  constexpr bool kIdle = KS > 1;
  static_assert(kSteps % KS == 0, "K split must divide the 18 K16-steps of a chunk");
  static_assert((KS - 1) * NCW * CT * 48 * 64 * 4 <= kTileB, "the K-split partial sums live in one tile image");
  extern __shared__ __align__(16) unsigned char lds[];                 // [2 tile images][2 B images][table][point numbers x 2]
  unsigned char* const bimg0 = lds + 2 * kTileB;
  unsigned* tab = reinterpret_cast<unsigned*>(bimg0 + 2 * kBImgB);
  int* pid_s = reinterpret_cast<int*>(tab + kSteps * 4);               // [group parity][16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Cin = a.Cin, Cout = a.Cout, NNp = a.NNp;
  const int64_t G = a.G;
  const int chunks = Cin / kCC / (int)gridDim.z, chunk0 = blockIdx.z * chunks;
  for (int e = tid; e < kSteps * 4; e += 64 * (NC + NPW)) {
    const int h = e & 1, rsel = (e >> 1) & 1, st = e >> 2;
    const int u = 2 * st + h, s = u / kA, t = u % kA;
    unsigned v = 0;
    for (int rt = 0; rt < 3; rt++) {
      const int r = 2 * rt + rsel;
      v |= (unsigned)run_of(kOrb.id[s][r], kOrb.anchor[t][r]) << (8 * rt);
    }
    tab[e] = v;
  }
  __syncthreads();
  if (wave >= NC) {
    // ---------------- producer ----------------
    const int pw = wave - NC;
    const int g = lane >> 4, col = lane & 15, odd = col & 1, half = col >> 3, cpair = (col & 7) >> 1;
    const unsigned rowlen = (unsigned)(kA * Cin);
    const bool v_gather = !(a.variant & 1), v_load = !(a.variant & 2), v_abuild = !(a.variant & 4);
    const float xs = x_split_scale(a.x_amax);
    // loader task of this lane: rows 8 oct + 4 jh .. + 3 of the union, 16-byte part q of a row's 192-byte chunk (60 lanes per wave)
    const int task = pw * 60 + lane, oh = task / 12, q = task - oh * 12, oct = oh >> 1, jh = oh & 1;
    const unsigned qoff = BLK ? (unsigned)(q * 4) : (unsigned)((q >> 1) * Cin + (q & 1) * 4);
    const unsigned cstride = BLK ? 48u : 8u;
    // destination inside a B image (piece 0): fragment (anchor pair q >> 2, K32-step oct >> 2), unit (oct & 3, column), second half of the unit for jh
    const int nt_l = q >> 2, g_l = oct & 3;
    const int bdst = (nt_l * kKSM + (oct >> 2)) * kBFragB + g_l * 256 + jh * 8;
    const int bsw = (nt_l + g_l) & 3;
    // A-build scratch: 8 orbits x kScrRow floats per wave, in the tile image the consumers are not reading during P(-1)
    float* const scr = reinterpret_cast<float*>(lds + (chunks & 1) * kTileB) + pw * (8 * kScrRow);
    // ---- item metadata, fetched with vector loads (one dword per lane: lanes 0-15 the group's order, 16-19 the plan record, 20 the group's pass count)
    auto meta_load = [&](int64_t grp, int sub) -> int {
      const int* p = lane < 16 ? a.order + grp * 16 + lane
                               : (lane < 20 ? reinterpret_cast<const int*>(a.desc + grp * kMaxSub + sub) + (lane - 16) : a.nsub + grp);
      return lane < 21 ? *p : 0;
    };
    int64_t grp = blockIdx.x;
    int sub = 0, gpar = 0;
    int meta = meta_load(grp, sub);
    unsigned roff[4], roff_n[4];
    int meta2 = 0;                                                        // lanes 0, 1: neighbour counts of this wave's two points
    f32x4 V[4];
#pragma unroll
    for (int j = 0; j < 4; j++) V[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // stage B of an item: row numbers of this lane's loader task, neighbour counts of the wave's points
    auto stage_b = [&](int64_t grp_, int meta_, unsigned (&ro)[4]) -> int {
      const int ustart = __builtin_amdgcn_readlane(meta_, 18), ucount = __builtin_amdgcn_readlane(meta_, 19);
      const int* urow = a.urow + grp_ * 16 * NNp + ustart;
      const bool lt = lane < 60 && oct < 4 * ((ucount + 31) >> 5);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int u = 8 * oct + 4 * jh + j;
        u = u < ucount ? u : ucount - 1;
        ro[j] = lt ? (unsigned)urow[u] * rowlen + qoff : 0u;
      }
      const int p0 = __builtin_amdgcn_readlane(meta_, pw), p1 = __builtin_amdgcn_readlane(meta_, pw + NPW);
      const int pp = lane == 0 ? p0 : p1;
      return (lane < 2 && pp >= 0) ? a.cnt[pp] : 0;
    };
    auto request = [&](int c, bool lt, const unsigned (&ro)[4]) {
      const unsigned co = (unsigned)(chunk0 + c) * cstride;
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (lt) V[j] = *reinterpret_cast<const f32x4*>(a.x + ro[j] + co);
    };
    using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
    auto convert = [&](int c, bool lt) {
      if (!lt) return;
      unsigned char* dst = bimg0 + (c & 1) * kBImgB + bdst;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        f16x4 hi, lo4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float v = V[j][i] * xs;
          hi[j] = (_Float16)v;
          lo4[j] = (_Float16)(v - (float)hi[j]);
        }
        const int unit = (((q & 3) * 4 + i) ^ bsw) * 16;
        *reinterpret_cast<f16x4*>(dst + unit) = hi;
        *reinterpret_cast<f16x4*>(dst + 3 * kKSM * kBFragB + unit) = lo4;
      }
    };
    meta2 = stage_b(grp, meta, roff);
    {
      const int ucount0 = __builtin_amdgcn_readlane(meta, 19);
      if (v_load) request(0, lane < 60 && oct < 4 * ((ucount0 + 31) >> 5), roff);
    }
    for (;;) {
      // ---------------- P(-1) ----------------
      const int lo = __builtin_amdgcn_readlane(meta, 16), np = __builtin_amdgcn_readlane(meta, 17);
      const int ucount = __builtin_amdgcn_readlane(meta, 19), nsub = __builtin_amdgcn_readlane(meta, 20);
      const int ksn = (ucount + 31) >> 5;
      const bool ltask = lane < 60 && oct < 4 * ksn;
      if (pw == 0 && sub == 0 && lane < 16) pid_s[gpar * 16 + lane] = meta;       // the group's point numbers, for the consumers' epilogue
      // A fragments of this wave's two points: lane = list slot; the point's 16 orbit weights of that slot land at the slot's union index in a
      // wave-private scratch (8 orbits at a time), and come back in MFMA A layout (row = orbit, 8 consecutive union rows per lane), split
      f16x8 ah[2][kKSM], al[2][kKSM];
#pragma unroll
      for (int pt = 0; pt < 2; pt++) {
        const int i = pw + NPW * pt;
        const int pid = __builtin_amdgcn_readlane(meta, pw + NPW * pt);
        const int c = __builtin_amdgcn_readlane(meta2, pt);
        const bool active = pid >= 0 && i >= lo && i < lo + np && v_abuild;
        const bool slot = active && lane < c;
        const int lcv = slot ? (int)a.loc[(grp * 16 + i) * NNp + lane] : 0;
        float hwv[16];
        {
          const float* hw = a.hwt + (int64_t)(pid >= 0 ? pid : 0) * 16 * NNp + lane;
#pragma unroll
          for (int o = 0; o < 16; o++) hwv[o] = slot ? hw[o * NNp] : 0.f;
        }
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          for (int e = lane; e < 8 * kScrRow / 4; e += 64) reinterpret_cast<f32x4*>(scr)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (slot) {
#pragma unroll
            for (int o = 0; o < 8; o++) scr[o * kScrRow + lcv] = hwv[8 * hf + o];
          }
          if (half == hf) {
#pragma unroll
            for (int ks = 0; ks < kKSM; ks++) {
              const f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + (col & 7) * kScrRow + 32 * ks + 8 * g);
              const f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + (col & 7) * kScrRow + 32 * ks + 8 * g + 4);
              const float v8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
              split8(v8, ah[pt][ks], al[pt][ks]);
            }
          }
        }
      }
      if (v_load) convert(0, ltask);
      if (chunks > 1 && v_load) request(1, ltask, roff);
      __syncthreads();
      if (kIdle) __syncthreads();
      // the item after this one
      const bool last_sub = sub + 1 >= nsub;
      const int64_t ngrp = last_sub ? grp + gridDim.x : grp;
      const int nsubi = last_sub ? 0 : sub + 1;
      const bool has_next = ngrp < G;
      int meta_n = 0, meta2_n = 0;
      for (int u = 0; u < chunks; u++) {
        if (u + 1 < chunks && v_load) convert(u + 1, ltask);
        if (u + 2 < chunks && v_load) request(u + 2, ltask, roff);
        if (u == 0 && has_next) meta_n = meta_load(ngrp, nsubi);
        if (u == (chunks >= 2 ? chunks - 2 : 0) && has_next) meta2_n = stage_b(ngrp, meta_n, roff_n);
        if (v_gather) {
          const unsigned char* bsrc = bimg0 + (u & 1) * kBImgB + g * 256;
          unsigned char* img = lds + (u & 1) * kTileB;
          f32x4 acc[2][3];
#pragma unroll
          for (int pt = 0; pt < 2; pt++)
#pragma unroll
            for (int nt = 0; nt < 3; nt++) acc[pt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < kKSM; ks++)
            if (ks < ksn) {
#pragma unroll
              for (int nt = 0; nt < 3; nt++) {
                const int unit = (col ^ ((nt + g) & 3)) * 16;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(bsrc + (nt * kKSM + ks) * kBFragB + unit);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(bsrc + ((3 + nt) * kKSM + ks) * kBFragB + unit);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[0][ks], bh, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[1][ks], bh, acc[1][nt], 0, 0, 0);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0][ks], bl, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1][ks], bl, acc[1][nt], 0, 0, 0);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0][ks], bh, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1][ks], bh, acc[1][nt], 0, 0, 0);
              }
            }
          // D[orbit 4 g + r][column col] = H[orbit][anchor 2 nt + half][channel col & 7]: f16 hi | lo, channel pairs exchanged inside lane
          // pairs (even lanes keep the hi dword of channels (c, c + 1), odd lanes the lo dword of (c - 1, c))
#pragma unroll
          for (int pt = 0; pt < 2; pt++) {
            unsigned char* dst = img + (pw + NPW * pt) * kRowB + odd * kPieceB + cpair * 4 + g * 64;
#pragma unroll
            for (int nt = 0; nt < 3; nt++)
#pragma unroll
              for (int r = 0; r < 4; r++) {
                const float v = acc[pt][nt][r];
                const _Float16 hi = (_Float16)v;
                const _Float16 lo16 = (_Float16)(v - (float)hi);
                const unsigned w = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo16) << 16);
                const unsigned w2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, false);
                const unsigned word = odd ? ((w2 >> 16) | (w & 0xffff0000u)) : ((w & 0xffffu) | (w2 << 16));
                *reinterpret_cast<unsigned*>(dst + ((2 * nt + half) * 16 + r) * 16) = word;
              }
          }
        }
        if (u == chunks - 1 && has_next && v_load) {
          const int ucn = __builtin_amdgcn_readlane(meta_n, 19);
          request(0, lane < 60 && oct < 4 * ((ucn + 31) >> 5), roff_n);
        }
        __syncthreads();
      }
      if (!has_next) break;
      if (last_sub) gpar ^= 1;
      grp = ngrp;
      sub = nsubi;
      meta = meta_n;
      meta2 = meta2_n;
#pragma unroll
      for (int j = 0; j < 4; j++) roff[j] = roff_n[j];
    }
    __syncthreads();                                                     // the consumers' last contraction
    if (kIdle) __syncthreads();
    return;
  }
  // ---------------- consumer (arithmetic of csrc/kpconv_mfma.hip) ----------------
  const int cw = wave % NCW, ksp = wave / NCW;
  const int NCT = Cout / 32, ct0 = (blockIdx.y * NCW + cw) * CT;
  const int i32 = lane & 31, h = lane >> 5;
  const int a_base = row_point(i32) * kRowB;
  const int tab_lane = row_rsel(i32) * 2 + h;
  f32x16 acc[CT][3];
#pragma unroll
  for (int n = 0; n < CT; n++)
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
      for (int v = 0; v < 16; v++) acc[n][rt][v] = 0.f;
  constexpr int BD = (CT == 2 && KS > 1) ? 2 : (CT == 1 ? 6 : 3);
  constexpr int U = BD == 2 ? 2 : 6;
  static_assert(kSPW % U == 0, "K16-steps per wave and chunk must be a multiple of the unroll");
  const int64_t wstep = (int64_t)NCT * 2 * 64;
  const u32x4* wbase = a.Wf + (int64_t)chunk0 * kSteps * wstep + (int64_t)ct0 * 2 * 64 + lane;
  const int total_steps = chunks * kSteps;                              // the weight stream of an item; the ring wraps into the next item
  u32x4 bq[BD][CT][2];
#pragma unroll
  for (int j = 0; j < BD; j++) {
    int gs = ksp + j * KS;
    gs = gs < total_steps ? gs : gs - total_steps;
#pragma unroll
    for (int n = 0; n < CT; n++) {
      bq[j][n][0] = wbase[gs * wstep + n * 128];
      bq[j][n][1] = wbase[gs * wstep + n * 128 + 64];
    }
  }
  constexpr int kAgent = 16;
  const bool v_mfma = !(a.variant & 8);
  auto contract = [&](int cc) {
    const unsigned char* img = lds + (cc & 1) * kTileB;
    f16x8 av[2][3][2];
    {
      const unsigned runs = tab[ksp * 4 + tab_lane];
#pragma unroll
      for (int rt = 0; rt < 3; rt++) {
        const int off = a_base + (int)((runs >> (8 * rt)) & 0xff) * 16;
        av[0][rt][0] = *reinterpret_cast<const f16x8*>(img + off);
        av[0][rt][1] = *reinterpret_cast<const f16x8*>(img + off + kPieceB);
      }
    }
#pragma unroll 1
    for (int q0 = v_mfma ? 0 : kSPW; q0 < kSPW; q0 += U) {
#pragma unroll
      for (int j = 0; j < U; j++) {
        const int ja = j & 1, jb = j % BD;
        const int st = ksp + (q0 + j) * KS;
        {
          const int sn = st + KS < kSteps ? st + KS : st;
          const unsigned runs = tab[sn * 4 + tab_lane];
#pragma unroll
          for (int rt = 0; rt < 3; rt++) {
            const int off = a_base + (int)((runs >> (8 * rt)) & 0xff) * 16;
            av[ja ^ 1][rt][0] = *reinterpret_cast<const f16x8*>(img + off);
            av[ja ^ 1][rt][1] = *reinterpret_cast<const f16x8*>(img + off + kPieceB);
          }
          // (the six reads stay HERE, a whole K-step ahead of their use: left alone the scheduler sinks each ds_read_b128 to the MFMA that
          // consumes it, which then waits out the LDS latency)
          if constexpr (CT == 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int n = 0; n < CT; n++) {
          const f16x8 b0 = __builtin_bit_cast(f16x8, bq[jb][n][0]);
#pragma unroll
          for (int rt = 0; rt < 3; rt++) acc[n][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ja][rt][1], b0, acc[n][rt], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < CT; n++) {
          const f16x8 b1 = __builtin_bit_cast(f16x8, bq[jb][n][1]);
#pragma unroll
          for (int rt = 0; rt < 3; rt++) acc[n][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ja][rt][0], b1, acc[n][rt], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < CT; n++) {
          const f16x8 b0 = __builtin_bit_cast(f16x8, bq[jb][n][0]);
#pragma unroll
          for (int rt = 0; rt < 3; rt++) acc[n][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ja][rt][0], b0, acc[n][rt], 0, 0, 0);
        }
        {
          int gs = cc * kSteps + st + BD * KS;
          gs = gs < total_steps ? gs : gs - total_steps;
#pragma unroll
          for (int n = 0; n < CT; n++) {
            bq[jb][n][0] = wbase[gs * wstep + n * 128];
            bq[jb][n][1] = wbase[gs * wstep + n * 128 + 64];
          }
        }
      }
    }
  };
  // Output of one group: accumulator register v of lane (column i32, half h) of row tile rt = (point v of the group, rotation 2 rt + (h ^ s[v >> 2])),
  // s = 0, 1, 1, 0; the point's row in `out` comes from the order (a scalar per v); absent points are skipped.
  auto epilogue = [&](int64_t group, int par) {
    const float inv_scale = a.hdr[0] / x_split_scale(a.x_amax);
    const int row_b = Cout * 4;
    int voff[CT][2];
#pragma unroll
    for (int n = 0; n < CT; n++) {
      voff[n][0] = ((ct0 + n) * 32 + i32) * 4 + h * row_b;
      voff[n][1] = ((ct0 + n) * 32 + i32) * 4 + (1 - h) * row_b;
    }
#pragma unroll
    for (int n = 0; n < CT; n++)
#pragma unroll
      for (int rt = 0; rt < 3; rt++) acc[n][rt] *= inv_scale;
    bool owner = true;
    if (gridDim.z > 1) {
      // Split input channels: this wave's slice goes to the partial buffer of its split; the wave that finds itself last of the gridDim.z
      // that own the slice adds the partials in split order and writes the output (csrc/kpconv_mfma.hip; no atomics on the output)
      const int64_t rows_all = G * kTP * kA;
      {
        const __amdgpu_buffer_rsrc_t prs =
            __builtin_amdgcn_make_buffer_rsrc(a.split_part + (blockIdx.z * rows_all + group * kTP * kA) * Cout, 0, kTP * kA * row_b, 0x00020000);
#pragma unroll
        for (int n = 0; n < CT; n++)
#pragma unroll
          for (int rt = 0; rt < 3; rt++)
#pragma unroll
            for (int v = 0; v < 16; v++) {
              const float val = acc[n][rt][v];
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), prs, voff[n][(0x6 >> (v >> 2)) & 1],
                                                    (v * kA + 2 * rt) * row_b, kAgent);
            }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int* counter = a.split_count + group * (NCT / CT) + ct0 / CT;
      int ticket = 0;
      if (lane == 0) ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      owner = ticket == (int)gridDim.z - 1;
      if (owner) {
        if (lane == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int n = 0; n < CT; n++)
#pragma unroll
          for (int rt = 0; rt < 3; rt++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[n][rt][v] = 0.f;
#pragma unroll 1
        for (int z = 0; z < (int)gridDim.z; z++) {
          const __amdgpu_buffer_rsrc_t prs =
              __builtin_amdgcn_make_buffer_rsrc(a.split_part + (z * rows_all + group * kTP * kA) * Cout, 0, kTP * kA * row_b, 0x00020000);
#pragma unroll
          for (int n = 0; n < CT; n++)
#pragma unroll
            for (int rt = 0; rt < 3; rt++)
#pragma unroll
              for (int v = 0; v < 16; v++)
                acc[n][rt][v] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, voff[n][(0x6 >> (v >> 2)) & 1],
                                                                                                (v * kA + 2 * rt) * row_b, kAgent));
        }
      }
    }
    if (owner) {
#pragma unroll
      for (int v = 0; v < 16; v++) {
        const int pid = __builtin_amdgcn_readfirstlane(pid_s[par * 16 + v]);
        if (pid < 0) continue;
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)pid * kA * Cout, 0, kA * row_b, 0x00020000);
#pragma unroll
        for (int n = 0; n < CT; n++)
#pragma unroll
          for (int rt = 0; rt < 3; rt++) {
            const float val = acc[n][rt][v];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ors, voff[n][(0x6 >> (v >> 2)) & 1], 2 * rt * row_b, 0);
          }
      }
    }
  };
  float* const red = reinterpret_cast<float*>(lds + kTileB);            // K-split partial sums: tile image 1 (free from the idle step to the end of P(0))
  int64_t grp = blockIdx.x, prev_grp = 0;
  int sub = 0, par = 0, prev_par = 0;
  int nsub = grp < G ? a.nsub[grp] : 0;
  bool prev_last = false;
  __syncthreads();                                                       // P(0, -1)
  if (kIdle) __syncthreads();
  for (;;) {
    // during P(k, 0) (or after the last item): the epilogue of the group the previous item finished
    if (prev_last) {
      if (ksp == 0) {
        if (KS > 1) {
#pragma unroll 1
          for (int kk = 1; kk < KS; kk++)
#pragma unroll
            for (int n = 0; n < CT; n++)
#pragma unroll
              for (int rt = 0; rt < 3; rt++)
#pragma unroll
                for (int v = 0; v < 16; v++) acc[n][rt][v] += red[((((kk - 1) * NCW + cw) * CT + n) * 48 + rt * 16 + v) * 64 + lane];
        }
        epilogue(prev_grp, prev_par);
      }
#pragma unroll
      for (int n = 0; n < CT; n++)
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
          for (int v = 0; v < 16; v++) acc[n][rt][v] = 0.f;
    }
    if (grp >= G) break;
    __syncthreads();                                                     // end of P(k, 0)
#pragma unroll 1
    for (int cc = 0; cc < chunks; cc++) {                                // during P(k, cc + 1); the last one during P(k + 1, -1)
      contract(cc);
      __syncthreads();
    }
    prev_last = sub + 1 >= nsub;
    prev_grp = grp;
    prev_par = par;
    if (kIdle) {                                                         // the idle step of the next item
      if (prev_last && ksp > 0) {
#pragma unroll
        for (int n = 0; n < CT; n++)
#pragma unroll
          for (int rt = 0; rt < 3; rt++)
#pragma unroll
            for (int v = 0; v < 16; v++) red[((((ksp - 1) * NCW + cw) * CT + n) * 48 + rt * 16 + v) * 64 + lane] = acc[n][rt][v];
      }
      __syncthreads();
    }
    if (prev_last) {
      grp += gridDim.x;
      sub = 0;
      par ^= 1;
      nsub = grp < G ? a.nsub[grp] : 0;
    } else {
      sub++;
    }
  }
}

// (csrc/kpconv_mfma.hip: fused_splits, with the groups as tiles)
int union_splits(int64_t tiles, int colblocks, int chunks) {
  const int64_t wg = tiles * colblocks;
  int best = 1;
  int64_t best_cost = ((wg + 255) / 256) * (chunks + 3);
  for (int z = 2; z <= 8; z *= 2) {
    if (chunks % (2 * z) != 0 || chunks / z < 4) break;
    const int64_t cost = ((wg * z + 255) / 256) * (chunks / z + 3);
    if (10 * cost <= 9 * best_cost) {
      best = z;
      best_cost = cost;
    }
  }
  return best;
}
int union_colblocks(int out_channels) {
  const int NCT = out_channels / 32;
  return NCT % 4 == 0 ? NCT / 4 : NCT % 2 == 0 ? NCT / 2 : NCT;
}
constexpr size_t kSplitCounterB = 64 * 1024;
int g_union_variant = 0;
int g_union_wgs = 0;              // workgroups per launch; 0: 256 (one per compute unit, each walking ~G / 256 groups) for layers up to 64 wide, 1024 beyond
                                  // (fewer groups per workgroup: the hardware's dispatch balances the uneven ones); se3_debug_set_kpconv_union_variant(v | wgs << 8) overrides

}  // namespace

extern "C" void se3_debug_set_kpconv_union_variant(int variant) {
  g_union_variant = variant & 0xff;
  g_union_wgs = variant >> 8;
}

extern "C" int64_t se3_point_order_groups(const int64_t* cloud_lengths_host, int num_clouds) {
  int64_t g = 0;
  for (int c = 0; c < num_clouds; c++) g += (cloud_lengths_host[c] + 15) / 16;
  return g;
}

extern "C" int se3_point_order_stages(const float* const* points, const int64_t* num_points, const int64_t* const* cloud_lengths_host,
                                      const int* num_clouds, const float* cell, int32_t* const* order, int num_stages, void* stream) {
  SE3_REQUIRE(points && num_points && cloud_lengths_host && num_clouds && cell && order, SE3_ERR_INVALID_ARG, "point_order: null pointer");
  SE3_REQUIRE(num_stages >= 1 && num_stages <= 4, SE3_ERR_UNSUPPORTED, "point_order: %d stages per call (max 4)", num_stages);
  OrderStages S;
  int64_t longest = 0, total = 0;
  int max_clouds = 0;
  for (int y = 0; y < 4; y++) {
    const int yy = y < num_stages ? y : 0;
    SE3_REQUIRE(points[yy] && order[yy] && cloud_lengths_host[yy], SE3_ERR_INVALID_ARG, "point_order: null pointer");
    SE3_REQUIRE(num_clouds[yy] >= 1 && num_clouds[yy] <= SE3_MAX_BATCH && cell[yy] > 0.f, SE3_ERR_UNSUPPORTED, "point_order: %d clouds (max %d)",
                num_clouds[yy], SE3_MAX_BATCH);
    S.pts[y] = points[yy];
    S.order[y] = order[yy];
    S.inv_cell[y] = 1.f / cell[yy];
    S.co[y].n = num_clouds[yy];
    S.co[y].start[0] = 0;
    for (int c = 0; c < num_clouds[yy]; c++) {
      S.co[y].start[c + 1] = S.co[y].start[c] + cloud_lengths_host[yy][c];
      longest = cloud_lengths_host[yy][c] > longest ? cloud_lengths_host[yy][c] : longest;
    }
    if (y < num_stages) {
      SE3_REQUIRE(S.co[y].start[num_clouds[yy]] == num_points[yy], SE3_ERR_INVALID_ARG, "point_order: cloud lengths do not add up to the point count");
      total += num_points[yy];
      max_clouds = num_clouds[yy] > max_clouds ? num_clouds[yy] : max_clouds;
    }
  }
  SE3_REQUIRE(longest <= kOrderCap, SE3_ERR_UNSUPPORTED, "point_order: a cloud of %lld points (one-launch form: at most %d; use se3_point_order_keys + a sort + _place)",
              (long long)longest, kOrderCap);
  if (total == 0) return SE3_OK;
  int n2 = 64;
  while (n2 < longest) n2 <<= 1;
  const size_t lds = (size_t)n2 * sizeof(unsigned long long);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&point_order_cloud_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(kOrderCap * sizeof(unsigned long long)));
    attr_set = true;
  }
  point_order_cloud_kernel<<<dim3((unsigned)max_clouds, (unsigned)num_stages), 1024, lds, (hipStream_t)stream>>>(S);
  SE3_CHECK_LAUNCH("point_order");
  return SE3_OK;
}

extern "C" int se3_point_order(const float* points, int64_t num_points, const int64_t* cloud_lengths_host, int num_clouds, float cell,
                               int32_t* order, void* stream) {
  return se3_point_order_stages(&points, &num_points, &cloud_lengths_host, &num_clouds, &cell, &order, 1, stream);
}

extern "C" int se3_point_order_keys(const float* points, int64_t num_points, const int64_t* cloud_lengths_host, int num_clouds, float cell,
                                    int64_t* keys, void* stream) {
  SE3_REQUIRE(points && keys && cloud_lengths_host, SE3_ERR_INVALID_ARG, "point_order_keys: null pointer");
  SE3_REQUIRE(num_clouds >= 1 && num_clouds <= SE3_MAX_BATCH && cell > 0.f, SE3_ERR_UNSUPPORTED, "point_order_keys: %d clouds (max %d)", num_clouds,
              SE3_MAX_BATCH);
  if (num_points == 0) return SE3_OK;
  CloudOffsets co;
  co.n = num_clouds;
  co.start[0] = 0;
  for (int c = 0; c < num_clouds; c++) co.start[c + 1] = co.start[c] + cloud_lengths_host[c];
  SE3_REQUIRE(co.start[num_clouds] == num_points, SE3_ERR_INVALID_ARG, "point_order_keys: cloud lengths do not add up to the point count");
  point_order_keys_kernel<<<(unsigned)se3_cdiv(num_points, 256), 256, 0, (hipStream_t)stream>>>(points, num_points, co, 1.f / cell, keys);
  SE3_CHECK_LAUNCH("point_order_keys");
  return SE3_OK;
}

extern "C" int se3_point_order_place(const int64_t* sorted_keys, const int64_t* sorted_index, int64_t num_points,
                                     const int64_t* cloud_lengths_host, int num_clouds, int32_t* order, void* stream) {
  SE3_REQUIRE(sorted_keys && sorted_index && order && cloud_lengths_host, SE3_ERR_INVALID_ARG, "point_order_place: null pointer");
  SE3_REQUIRE(num_clouds >= 1 && num_clouds <= SE3_MAX_BATCH, SE3_ERR_UNSUPPORTED, "point_order_place: %d clouds (max %d)", num_clouds, SE3_MAX_BATCH);
  const int64_t G = se3_point_order_groups(cloud_lengths_host, num_clouds);
  if (G == 0) return SE3_OK;
  if (hipMemsetAsync(order, 0xff, (size_t)G * 16 * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) {
    se3_set_error("point_order_place: memset failed");
    return SE3_ERR_LAUNCH;
  }
  if (num_points == 0) return SE3_OK;
  CloudOffsets co;
  co.n = num_clouds;
  co.start[0] = 0;
  for (int c = 0; c < num_clouds; c++) co.start[c + 1] = co.start[c] + cloud_lengths_host[c];
  point_order_place_kernel<<<(unsigned)se3_cdiv(num_points, 256), 256, 0, (hipStream_t)stream>>>(sorted_keys, sorted_index, num_points, co, order);
  SE3_CHECK_LAUNCH("point_order_place");
  return SE3_OK;
}

extern "C" size_t se3_kpconv_union_plan_bytes(int64_t num_groups, int num_neighbors) {
  if (num_groups < 0 || num_neighbors < 1 || num_neighbors > 64) return 0;
  const int NNp = num_neighbors <= 32 ? 32 : (num_neighbors + 39) / 40 * 40;
  return plan_bytes(num_groups, NNp);
}

extern "C" int se3_kpconv_union_plan(const void* table, int64_t num_queries, int num_neighbors, const int32_t* order, int64_t num_groups,
                                     void* plan, size_t plan_bytes_given, void* stream) {
  SE3_REQUIRE(table && order && plan, SE3_ERR_INVALID_ARG, "kpconv_union_plan: null pointer");
  SE3_REQUIRE(num_neighbors >= 1 && num_neighbors <= 64, SE3_ERR_UNSUPPORTED, "kpconv_union_plan: %d neighbours (max 64)", num_neighbors);
  SE3_REQUIRE(plan_bytes_given >= se3_kpconv_union_plan_bytes(num_groups, num_neighbors), SE3_ERR_INVALID_ARG, "kpconv_union_plan: plan buffer too small");
  if (num_groups == 0) return SE3_OK;
  const NeighborTable t = table_views(table, num_queries, num_neighbors);
  const PlanViews pv = plan_views(plan, num_groups, t.NNp);
  kpconv_union_plan_kernel<<<(unsigned)num_groups, 64, 0, (hipStream_t)stream>>>(t.nbr, t.cnt, t.NNp, order, pv);
  SE3_CHECK_LAUNCH("kpconv_union_plan");
  return SE3_OK;
}

extern "C" size_t se3_kpconv_union_split_workspace_bytes(int64_t num_groups, int in_channels, int out_channels) {
  if (num_groups <= 0 || in_channels % kCC || out_channels % 32) return 0;
  const int z = union_splits(num_groups, union_colblocks(out_channels), in_channels / kCC);
  if (z == 1 || (size_t)num_groups * (out_channels / 32) * sizeof(int) > kSplitCounterB) return 0;
  return kSplitCounterB + (size_t)z * num_groups * kTP * kA * out_channels * sizeof(float);
}

extern "C" int se3_kpconv_so3_union(const float* x, const void* table, const void* plan, int64_t num_groups, int64_t num_queries,
                                    int64_t num_support, int num_neighbors, int in_channels, int out_channels, const void* weight_pieces,
                                    float* out, void* split_workspace, size_t split_workspace_bytes, int x_chunked, const float* x_amax,
                                    void* stream) {
  SE3_REQUIRE(x && table && plan && weight_pieces && out, SE3_ERR_INVALID_ARG, "kpconv_so3_union: null pointer");
  SE3_REQUIRE(num_neighbors >= 1 && num_neighbors <= 64, SE3_ERR_UNSUPPORTED, "kpconv_so3_union: %d neighbours (max 64)", num_neighbors);
  SE3_REQUIRE(in_channels > 0 && in_channels % kCC == 0 && out_channels >= 32 && out_channels % 32 == 0, SE3_ERR_UNSUPPORTED,
              "kpconv_so3_union: channels (%d, %d) must be multiples of (8, 32)", in_channels, out_channels);
  SE3_REQUIRE((int64_t)num_support * kA * in_channels < (1ll << 31), SE3_ERR_UNSUPPORTED, "kpconv_so3_union: support features exceed 2^31 elements");
  if (num_queries == 0 || num_groups == 0) return SE3_OK;
  hipStream_t st = (hipStream_t)stream;
  const NeighborTable t = table_views(table, num_queries, num_neighbors);
  const PlanViews pv = plan_views(const_cast<void*>(plan), num_groups, t.NNp);
  const int NCT = out_channels / 32;
  UnionArgs a;
  a.x = x;
  a.hwt = t.hwt;
  a.cnt = t.cnt;
  a.NNp = t.NNp;
  a.order = pv.order;
  a.nsub = pv.nsub;
  a.desc = pv.desc;
  a.urow = pv.urow;
  a.loc = pv.loc;
  a.hdr = static_cast<const float*>(weight_pieces);
  a.Wf = reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(weight_pieces) + kHeaderB);
  a.P = num_queries;
  a.G = num_groups;
  a.Cin = in_channels;
  a.Cout = out_channels;
  a.out = out;
  a.split_part = nullptr;
  a.split_count = nullptr;
  a.variant = g_union_variant;
  a.x_amax = x_amax;
  const size_t lds = (size_t)2 * kTileB + 2 * kBImgB + kSteps * 4 * sizeof(unsigned) + 32 * sizeof(int);
  int splits = 1;
  if (split_workspace != nullptr) {
    const size_t need = se3_kpconv_union_split_workspace_bytes(num_groups, in_channels, out_channels);
    if (need != 0) {
      SE3_REQUIRE(split_workspace_bytes >= need, SE3_ERR_WORKSPACE, "kpconv_so3_union: split workspace too small");
      splits = union_splits(num_groups, union_colblocks(out_channels), in_channels / kCC);
      a.split_count = static_cast<int*>(split_workspace);
      a.split_part = reinterpret_cast<float*>(static_cast<unsigned char*>(split_workspace) + kSplitCounterB);
    }
  }
  // persistent workgroups: one per compute unit (160 KB of LDS each), dealt over (column blocks, channel splits, groups)
  auto wgx = [&](int colblocks) {
    const int wgs = g_union_wgs > 0 ? g_union_wgs : (out_channels <= 64 ? 256 : 1024);
    int64_t per = wgs / ((int64_t)colblocks * splits);
    per = per < 1 ? 1 : per;
    return per < num_groups ? per : num_groups;
  };
#define SE3_UNION_X(NCW_, KS_, CT_, BLK_)                                                                                                \
  {                                                                                                                                      \
    static bool attr_set = false;                                                                                                        \
    if (!attr_set) {                                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kpconv_union_kernel<NCW_, KS_, CT_, BLK_>),                               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                                   \
      attr_set = true;                                                                                                                   \
    }                                                                                                                                    \
    kpconv_union_kernel<NCW_, KS_, CT_, BLK_>                                                                                            \
        <<<dim3((unsigned)wgx(NCT / (NCW_ * CT_)), (unsigned)(NCT / (NCW_ * CT_)), (unsigned)splits), 64 * (NCW_ * KS_ + 8), lds, st>>>(a); \
  }
#define SE3_UNION(NCW_, KS_, CT_)                      \
  {                                                    \
    if (x_chunked) SE3_UNION_X(NCW_, KS_, CT_, true)   \
    else SE3_UNION_X(NCW_, KS_, CT_, false)            \
  }
  // (the two-tile form (4, 1, 2) of csrc/kpconv_mfma.hip does not fit the persistent kernel's registers: the scheduler sinks the weight ring's
  // requests to their use and the consumers lose half their rate; wide layers run as column blocks of 128, the cheap gather done per block)
  if (NCT % 4 == 0) SE3_UNION(4, 1, 1)
  else if (NCT % 2 == 0) SE3_UNION(1, 3, 2)
  else SE3_UNION(1, 3, 1)
#undef SE3_UNION_X
#undef SE3_UNION
  SE3_CHECK_LAUNCH("kpconv_so3_union");
  return SE3_OK;
}
