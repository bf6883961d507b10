// Shared by csrc/kpconv_so3.hip (producer) and csrc/kpconv_mfma.hip (consumer): the "orbit sum" form of the KPConvInterSO3 operand.
//
// Reference: geotransformer/modules/e2pn/blocks_epn.py:228-332 (kidx_rot / ridx_rot) and :454-546 (forward).  The slot sums
//   G[p, r, (s, t), c] = sum_{k: kidx[k, r] = s} F[p, k, a: ridx[a, r] = t, c]
// take only 16 DISTINCT kernel-point sums over all (s, r): the 6 vertices and the centre alone (slots 0, 2, 5), the 3 equators (slot 1:
// the four vertices around the axis of r) and the 6 face quadruples (slots 3 / 4: the faces around vertex r / around its opposite).  So
//   H[p, hh, a, c] = sum_{k in orbit hh} F[p, k, a, c]                         (16 x 6 values per point and channel instead of 36 x 6)
// is formed ONCE per point (by the thread that holds F[p, :, a, c] in registers), split ONCE into two f16 pieces (hi + lo = 22 significant
// bits) and the matrix-core kernel only READS fragments: G[p, r, (s, t), c] = H[p, orbit_id[s][r], anchor_of[t][r], c].
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace kpsum {

constexpr int kK = 15, kA = 6, kS = 6;
constexpr int kOrbits = 16;                    // distinct kernel-point sums
constexpr int kCC = 8;                         // channels per chunk (one 16-byte f16 run)
constexpr int kTP = 16;                        // points per tile
constexpr int kRowB = (kOrbits * kA + 1) * 16; // bytes per (piece, point) row of a tile: 96 runs of 16 B + 16 B pad (odd number of 16-B
                                               // units: the 16 points of a b128 read fall into different bank quads)
constexpr int kPieceB = kTP * kRowB + 32;      // hi piece, 32 B, lo piece, 48 B: the lo piece sits 8 banks and the next image 4 banks further, so
constexpr int kTileB = 2 * kPieceB + 16;       // that the producers' dword stores (hi / lo pieces of two images per instruction) spread over all banks
                                               // one (channel chunk, point tile) image: [piece][point][row] = 49 744 B, the same in HBM and LDS
constexpr int kSteps = kS * kA / 2;            // K16-steps per channel chunk: 2 weight slots x 8 channels

// slot tables of the SE3ET configuration (se3et_amd/tables.py kernel_slot_table / anchor_slot_table)
__device__ constexpr int kKidx[kK][kA] = {{0, 1, 1, 1, 1, 2}, {1, 0, 1, 2, 1, 1}, {1, 1, 0, 1, 2, 1}, {1, 2, 1, 0, 1, 1},
                                          {1, 1, 2, 1, 0, 1}, {2, 1, 1, 1, 1, 0}, {3, 3, 3, 4, 4, 4}, {3, 4, 3, 3, 4, 4},
                                          {3, 4, 4, 3, 3, 4}, {3, 3, 4, 4, 3, 4}, {4, 3, 3, 4, 4, 3}, {4, 4, 3, 3, 4, 3},
                                          {4, 4, 4, 3, 3, 3}, {4, 3, 4, 4, 3, 3}, {5, 5, 5, 5, 5, 5}};
__device__ constexpr int kRidx[kA][kA] = {{0, 3, 3, 3, 3, 5}, {1, 0, 4, 5, 2, 1}, {2, 2, 0, 4, 5, 4},
                                          {3, 5, 2, 0, 4, 3}, {4, 4, 5, 2, 0, 2}, {5, 1, 1, 1, 1, 0}};

struct OrbitTable {
  unsigned mask[kOrbits];      // bit k set: kernel point k belongs to the orbit
  int count;                   // distinct orbits found (must be kOrbits)
  int id[kS][kA];              // [s][r] -> orbit
  int anchor[kA][kA];          // [t][r] -> a with ridx[a][r] = t
};

constexpr OrbitTable make_orbits() {
  OrbitTable T{};
  for (int s = 0; s < kS; s++)
    for (int r = 0; r < kA; r++) {
      unsigned m = 0;
      for (int k = 0; k < kK; k++)
        if (kKidx[k][r] == s) m |= 1u << k;
      int at = -1;
      for (int o = 0; o < T.count; o++)
        if (T.mask[o] == m) at = o;
      if (at < 0) {
        at = T.count;
        if (at < kOrbits) T.mask[at] = m;
        T.count++;
      }
      T.id[s][r] = at;
    }
  for (int t = 0; t < kA; t++)
    for (int r = 0; r < kA; r++) {
      int a = 0;
      for (int aa = 0; aa < kA; aa++)
        if (kRidx[aa][r] == t) a = aa;
      T.anchor[t][r] = a;
    }
  return T;
}
__device__ constexpr OrbitTable kOrb = make_orbits();
static_assert(make_orbits().count == kOrbits, "the SE3ET slot tables have 16 distinct kernel-point orbits");

// Power-of-two scale of the input features of the fused kernels before their f16 split, from the largest magnitude of the tensor (written by
// the GroupNorm apply pass that produces it, se3_group_norm_apply_amax): inside [2^-4, 2^7) -- or unknown, zero, NaN / Inf -- the features are
// split as they are (bit-identical to the unscaled kernels); outside, the largest magnitude goes to [2^6, 2^7) (exact), the orbit sums of
// up to 64 neighbours x 4 kernel points stay below 2^15, and the epilogue takes the scale out again.
__device__ __forceinline__ float x_split_scale(const float* amax) {
  if (amax == nullptr) return 1.f;
  const unsigned b = *reinterpret_cast<const unsigned*>(amax);
  const int e = (int)((b >> 23) & 0xff);
  if (e == 0 || e == 0xff || (e >= 123 && e <= 133)) return 1.f;
  const int k = 133 - e > 126 ? 126 : 133 - e;                           // (a largest magnitude below 2^-120: 2^126, the scale stays a normal float)
  return __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
}

// 16-byte run of (orbit, anchor) inside a point's row: anchor-major, so that the 16 orbits a producer wave holds for one anchor are contiguous
__host__ __device__ constexpr int run_of(int orbit, int a) { return a * kOrbits + orbit; }

}  // namespace kpsum
