// A1: stack-mode grid subsampling on gfx950.
//
// Replaces geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-109 (+ grid_subsampling_cpu.h:24-74).
// Contract (bit-exact, the output rows are copies of input rows):
//   origin  = floor(min * float(1/voxel)) * voxel ; nx, ny = floor((max - origin) / voxel) + 1      (float32, unfused)
//   key     = ix + nx*iy + nx*ny*iz with i* = floor((p - origin) / voxel)                            (size_t)
//   choice  = the member closest to the voxel mean: mean = (sum in input order) * float(1/count); distance
//             sqrtf((dx*dx + dy*dy) + dz*dz) in float32; first strict minimum in input order
//   order   = iteration order of libstdc++ std::unordered_map<size_t,...> filled in first-seen order of the keys.
//
// Pipeline (all on the device, no host round trip):
//   bounds_kernel      one workgroup per cloud: min/max -> origin, nx, ny
//   hash_kernel        per point: key, insert into an open-addressing table (64-bit CAS), first-seen index (atomicMin),
//                      member count (atomicAdd)
//   rank_kernel        one workgroup per cloud: voxels ranked by first-seen index with a prefix sum (no sort), CSR offsets
//   fill_kernel        per point: append to its voxel's member segment
//   select_kernel      per voxel: order the (few) members by input index, sequential float32 mean, closest member
//   order_kernel       one workgroup per cloud: emulation of the unordered_map iteration order.  The libstdc++ layout
//                      (one forward list, a node entering an empty bucket goes to the list front, otherwise to the front
//                      of its bucket; bucket counts 13, 29, 59, ... doubling at load factor 1) makes the list after a
//                      (re)hash a pure function of the insertion sequence: buckets ordered by DEscending first touch,
//                      members by DEscending position.  Each of the O(log V) growth stages is therefore evaluated in
//                      parallel with atomics + one prefix sum instead of walking a list.
//   gather_kernel      writes the selected rows in emission order, clouds back to back
#define SE3_EXACT_FP 1
#include "common.h"

namespace {

constexpr int kBlock = 1024;
// Empty-slot marker of the voxel hash table.  Voxel keys are ix + nx iy + nx ny iz evaluated in size_t arithmetic with every index in
// [-1, n] (see voxel_index: an index of -1 wraps, as in the reference), i.e. within nx ny (nz + 2) of 0 (mod 2^64): 2^63 is never a key
// (all-ones IS one: ix = -1, iy = iz = 0).
constexpr unsigned long long kEmptyKey = 0x8000000000000000ull;

// (size_t)floor(x) as the reference's x86-64 build evaluates it (grid_subsampling_cpu.cpp:47-49): through the SIGNED conversion, so the
// -1 that float rounding can produce for the cloud's own minimum (origin = floor(min * float(1 / v)) * v can land one ulp ABOVE min:
// data/demo/src.npy holds such a point) becomes 2^64 - 1 and the key wraps.  The GPU's float -> unsigned conversion saturates at 0 instead,
// which put that point's voxel into another bucket of the order emulation.
__device__ inline unsigned long long voxel_index(float q) { return (unsigned long long)(long long)floorf(q); }
__constant__ unsigned long long kBucketSeq[24] = {13ull, 29ull, 59ull, 127ull, 257ull, 541ull, 1109ull, 2357ull, 5087ull,
    10273ull, 20753ull, 42043ull, 85229ull, 172933ull, 351061ull, 712697ull, 1447153ull, 2938679ull, 5967347ull,
    12117689ull, 24607243ull, 49969847ull, 101473717ull, 0ull};

struct CloudMeta {          // per cloud, device resident
  float org[3];
  float voxel;
  unsigned long long nx, ny;
  int n_vox;
  int pad;
};

struct BatchInfo;
struct Layout {             // device workspace carve-up (per cloud arrays are indexed with the cloud's point offset)
  BatchInfo* info;                    // the batch descriptor every kernel reads (written by init_kernel: host or device lengths)
  CloudMeta* meta;                    // [batch]
  unsigned long long* table_key;      // [cap_total]
  int* table_first;                   // [cap_total]  first-seen local point index of the slot's voxel
  int* table_cnt;                     // [cap_total]
  int* table_vox;                     // [cap_total]  voxel id (first-seen rank) of the slot
  int* slot_of_point;                 // [n]
  int* vox_at_point;                  // [n]   slot+1 if the point is the first of its voxel else 0
  unsigned long long* vox_key;        // [n]   key per voxel id
  int* vox_cnt;                       // [n]
  int* vox_off;                       // [n]   CSR offsets (local)
  int* vox_fill;                      // [n]
  int* members;                       // [n]   local point indices grouped by voxel
  int* sel;                           // [n]   selected local point index per voxel id
  int* seq_a;                         // [n]   order emulation ping
  int* seq_b;                         // [n]   pong
  int* pos_bkt;                       // [n]
  int* pos_next;                      // [n]
  int* pos_w;                         // [n]
  int* bk_first;                      // [bk_total]
  int* bk_cnt;                        // [bk_total]
  int* bk_head;                       // [bk_total]
};

struct BatchInfo {
  int64_t start[SE3_MAX_BATCH];
  int64_t count[SE3_MAX_BATCH];
  int64_t cap_start[SE3_MAX_BATCH];
  int64_t cap[SE3_MAX_BATCH];        // power of two
  int64_t bk_start[SE3_MAX_BATCH];
};

__host__ __device__ inline int64_t pow2_at_least(int64_t v) {
  int64_t c = 16;
  while (c < v) c <<= 1;
  return c;
}

__host__ __device__ inline unsigned long long bucket_cap_for(int64_t n) {
  // smallest member of the libstdc++ growth sequence that holds n elements
  const unsigned long long seq[] = {13ull, 29ull, 59ull, 127ull, 257ull, 541ull, 1109ull, 2357ull, 5087ull, 10273ull,
      20753ull, 42043ull, 85229ull, 172933ull, 351061ull, 712697ull, 1447153ull, 2938679ull, 5967347ull, 12117689ull,
      24607243ull, 49969847ull, 101473717ull};
  for (int i = 0; i < 23; i++)
    if ((unsigned long long)n <= seq[i]) return seq[i];
  return 0ull;
}

// ---- block-wide helpers (kBlock threads) ---------------------------------------------------------------------------
__device__ float block_reduce(float v, bool is_max, float* sh) {
  for (int o = 32; o > 0; o >>= 1) {
    float t = __shfl_xor(v, o);
    v = is_max ? fmaxf(v, t) : fminf(v, t);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
  for (int w = 1; w < kBlock / 64; w++) r = is_max ? fmaxf(r, sh[w]) : fminf(r, sh[w]);
  return r;
}

// exclusive prefix sum of a[0..n) in place (or suffix sum when `reverse`), one workgroup; returns the total
__device__ int block_scan(int* a, int n, bool reverse, int* sh) {
  const int t = threadIdx.x;
  const int chunk = (n + kBlock - 1) / kBlock;
  const int lo = t * chunk, hi = min(n, lo + chunk);
  int s = 0;
  for (int i = lo; i < hi; i++) s += reverse ? a[n - 1 - i] : a[i];
  __syncthreads();
  sh[t] = s;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {          // Hillis-Steele inclusive scan over the 1024 partials
    int v = (t >= off) ? sh[t - off] : 0;
    __syncthreads();
    sh[t] += v;
    __syncthreads();
  }
  const int total = sh[kBlock - 1];
  int run = sh[t] - s;
  for (int i = lo; i < hi; i++) {
    int idx = reverse ? n - 1 - i : i;
    int v = a[idx];
    a[idx] = run;
    run += v;
  }
  __syncthreads();
  return total;
}

// ---- kernels -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bounds_kernel(const float* __restrict__ pts, const BatchInfo* __restrict__ bip, float voxel,
                                                        Layout L) {
  const BatchInfo& bi = *bip;
  __shared__ float sh[kBlock / 64];
  const int b = blockIdx.x;
  const int64_t n = bi.count[b];
  const float* p = pts + 3 * bi.start[b];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = threadIdx.x; i < n; i += kBlock)
    for (int d = 0; d < 3; d++) {
      float v = p[3 * i + d];
      mn[d] = fminf(mn[d], v);
      mx[d] = fmaxf(mx[d], v);
    }
  float rmn[3], rmx[3];
  for (int d = 0; d < 3; d++) {
    rmn[d] = block_reduce(mn[d], false, sh);
    rmx[d] = block_reduce(mx[d], true, sh);
  }
  if (threadIdx.x == 0) {
    CloudMeta m;
    const float inv = (float)(1.0 / (double)voxel);
    for (int d = 0; d < 3; d++) m.org[d] = __fmul_rn(floorf(__fmul_rn(rmn[d], inv)), voxel);
    m.voxel = voxel;
    m.nx = n > 0 ? (unsigned long long)__fadd_rn(floorf(se3_exact_div(__fsub_rn(rmx[0], m.org[0]), voxel)), 1.0f) : 1ull;
    m.ny = n > 0 ? (unsigned long long)__fadd_rn(floorf(se3_exact_div(__fsub_rn(rmx[1], m.org[1]), voxel)), 1.0f) : 1ull;
    m.n_vox = 0;
    m.pad = 0;
    L.meta[b] = m;
  }
}

__global__ void hash_kernel(const float* __restrict__ pts, const BatchInfo* __restrict__ bip, Layout L) {
  const BatchInfo& bi = *bip;
  const int b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bi.count[b]) return;
  const CloudMeta m = L.meta[b];
  const float* p = pts + 3 * (bi.start[b] + i);
  const unsigned long long ix = voxel_index(se3_exact_div(__fsub_rn(p[0], m.org[0]), m.voxel));
  const unsigned long long iy = voxel_index(se3_exact_div(__fsub_rn(p[1], m.org[1]), m.voxel));
  const unsigned long long iz = voxel_index(se3_exact_div(__fsub_rn(p[2], m.org[2]), m.voxel));
  const unsigned long long key = ix + m.nx * iy + m.nx * m.ny * iz;
  const unsigned long long mask = (unsigned long long)bi.cap[b] - 1ull;
  unsigned long long h = (key * 0x9E3779B97F4A7C15ull) >> 20 & mask;
  unsigned long long* tk = L.table_key + bi.cap_start[b];
  while (true) {
    unsigned long long cur = tk[h];
    if (cur == kEmptyKey) {
      const unsigned long long prev = atomicCAS(&tk[h], kEmptyKey, key);
      cur = (prev == kEmptyKey) ? key : prev;
    }
    if (cur == key) break;
    h = (h + 1) & mask;
  }
  const int64_t slot = bi.cap_start[b] + (int64_t)h;
  atomicMin(&L.table_first[slot], (int)i);
  atomicAdd(&L.table_cnt[slot], 1);
  L.slot_of_point[bi.start[b] + i] = (int)h;
}

__global__ void mark_kernel(const BatchInfo* __restrict__ bip, Layout L) {       // per table slot: flag the first-seen point of each voxel
  const BatchInfo& bi = *bip;
  const int b = blockIdx.y;
  const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= bi.cap[b]) return;
  const int64_t slot = bi.cap_start[b] + h;
  if (L.table_key[slot] != kEmptyKey) L.vox_at_point[bi.start[b] + L.table_first[slot]] = (int)h + 1;
}

__global__ __launch_bounds__(kBlock) void rank_kernel(const BatchInfo* __restrict__ bip, Layout L) {
  const BatchInfo& bi = *bip;
  __shared__ int sh[kBlock];
  const int b = blockIdx.x;
  const int n = (int)bi.count[b];
  const int64_t p0 = bi.start[b], c0 = bi.cap_start[b];
  // 1. rank = number of voxels first seen before point i ; vox_off is used as scratch for the flags
  int* flag = L.vox_off + p0;
  for (int i = threadIdx.x; i < n; i += kBlock) flag[i] = L.vox_at_point[p0 + i] ? 1 : 0;
  __syncthreads();
  const int nv = block_scan(flag, n, false, sh);
  for (int i = threadIdx.x; i < n; i += kBlock) {
    const int h1 = L.vox_at_point[p0 + i];
    if (h1) {
      const int v = flag[i];
      L.table_vox[c0 + h1 - 1] = v;
      L.vox_key[p0 + v] = L.table_key[c0 + h1 - 1];
      L.vox_cnt[p0 + v] = L.table_cnt[c0 + h1 - 1];
    }
  }
  __syncthreads();
  // 2. CSR offsets over voxel ids
  for (int v = threadIdx.x; v < nv; v += kBlock) L.vox_off[p0 + v] = L.vox_cnt[p0 + v];
  __syncthreads();
  block_scan(L.vox_off + p0, nv, false, sh);
  if (threadIdx.x == 0) L.meta[b].n_vox = nv;
}

__global__ void fill_kernel(const BatchInfo* __restrict__ bip, Layout L) {
  const BatchInfo& bi = *bip;
  const int b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bi.count[b]) return;
  const int64_t p0 = bi.start[b];
  const int v = L.table_vox[bi.cap_start[b] + L.slot_of_point[p0 + i]];
  const int k = atomicAdd(&L.vox_fill[p0 + v], 1);
  L.members[p0 + L.vox_off[p0 + v] + k] = (int)i;
}

__global__ void select_kernel(const float* __restrict__ pts, const BatchInfo* __restrict__ bip, Layout L) {
  const BatchInfo& bi = *bip;
  const int b = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= L.meta[b].n_vox) return;
  const int64_t p0 = bi.start[b];
  int* mem = L.members + p0 + L.vox_off[p0 + v];
  const int c = L.vox_cnt[p0 + v];
  for (int i = 1; i < c; i++) {            // members arrive in atomic order: restore input order (tiny lists)
    int x = mem[i], j = i - 1;
    while (j >= 0 && mem[j] > x) { mem[j + 1] = mem[j]; j--; }
    mem[j + 1] = x;
  }
  const float* P = pts + 3 * p0;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int i = 0; i < c; i++) {
    const float* p = P + 3 * (int64_t)mem[i];
    sx = __fadd_rn(sx, p[0]); sy = __fadd_rn(sy, p[1]); sz = __fadd_rn(sz, p[2]);
  }
  const float a = (float)(1.0 / (double)c);
  const float ax = __fmul_rn(sx, a), ay = __fmul_rn(sy, a), az = __fmul_rn(sz, a);
  int best = mem[0];
  float bestd = INFINITY;
  for (int i = 0; i < c; i++) {
    const float* p = P + 3 * (int64_t)mem[i];
    const float dx = __fsub_rn(p[0], ax), dy = __fsub_rn(p[1], ay), dz = __fsub_rn(p[2], az);
    const float d = se3_exact_sqrt(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    if (i == 0 || d < bestd) { bestd = d; best = mem[i]; }
  }
  L.sel[p0 + v] = best;
}

// Empty hash table (keys all ones, first-seen index 0x7f7f7f7f, no members) and per-point counters at zero: one launch.  Its first thread
// also publishes the batch descriptor the other kernels read: the host's (lengths known on the host) or, with lengths_dev, one built from
// the per-cloud counts a previous launch left in device memory (se3_grid_subsample_dev: a pyramid stage that follows another without a host
// synchronisation in between).
__global__ void init_kernel(Layout L, int64_t cap_total, int64_t n, BatchInfo host_info, const int64_t* __restrict__ lengths_dev, int batch) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (lengths_dev == nullptr) {
      *L.info = host_info;
    } else {
      BatchInfo bi;
      int64_t tot = 0, cap = 0, bk = 0;
      for (int b = 0; b < SE3_MAX_BATCH; b++) {
        const int64_t len = b < batch ? lengths_dev[b] : 0;
        bi.start[b] = tot; bi.count[b] = len;
        bi.cap_start[b] = cap; bi.cap[b] = pow2_at_least(2 * len);
        bi.bk_start[b] = bk;
        tot += len; cap += bi.cap[b];
        bk += (int64_t)bucket_cap_for(len > 0 ? len : 1);
      }
      *L.info = bi;
    }
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap_total; i += stride) {
    L.table_key[i] = kEmptyKey;
    L.table_first[i] = 0x7f7f7f7f;
    L.table_cnt[i] = 0;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    L.vox_at_point[i] = 0;
    L.vox_fill[i] = 0;
  }
}

// One workgroup per cloud.  seq_a/seq_b hold the node sequence (voxel ids) of the current growth stage.
__global__ __launch_bounds__(kBlock) void order_kernel(const BatchInfo* __restrict__ bip, Layout L) {
  const BatchInfo& bi = *bip;
  __shared__ int sh[kBlock];
  const int b = blockIdx.x;
  const int64_t p0 = bi.start[b];
  const int V = L.meta[b].n_vox;
  const unsigned long long* key = L.vox_key + p0;
  int* cur = L.seq_a + p0;
  int* nxt = L.seq_b + p0;
  int* pbkt = L.pos_bkt + p0;
  int* pnext = L.pos_next + p0;
  int* pw = L.pos_w + p0;
  int* first = L.bk_first + bi.bk_start[b];
  int* cnt = L.bk_cnt + bi.bk_start[b];
  int* head = L.bk_head + bi.bk_start[b];
  int done = 0;                                    // nodes already inside the list (a prefix of `cur` in list order)
  for (int level = 0; done < V; level++) {
    const unsigned long long B = kBucketSeq[level];
    const int n = (int)min((unsigned long long)V, B);          // list length when this stage ends
    for (int i = done + threadIdx.x; i < n; i += kBlock) cur[i] = i;   // newcomers enter in first-seen order
    for (unsigned long long k = threadIdx.x; k < B; k += kBlock) { first[k] = 0x7fffffff; cnt[k] = 0; head[k] = -1; }
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += kBlock) {
      const int bk = (int)(key[cur[p]] % B);
      pbkt[p] = bk;
      atomicMin(&first[bk], p);
      atomicAdd(&cnt[bk], 1);
      pnext[p] = atomicExch(&head[bk], p);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += kBlock) pw[p] = (first[pbkt[p]] == p) ? cnt[pbkt[p]] : 0;
    __syncthreads();
    block_scan(pw, n, true, sh);                   // pw[p] = number of nodes in buckets first touched after p
    for (int p = threadIdx.x; p < n; p += kBlock) {
      const int bk = pbkt[p];
      int later = 0;
      for (int q = head[bk]; q >= 0; q = pnext[q]) later += (q > p);
      nxt[pw[first[bk]] + later] = cur[p];
    }
    __syncthreads();
    int* t = cur; cur = nxt; nxt = t;
    done = n;
  }
  // final emission order lives in `cur`; publish it in seq_a
  if (cur != L.seq_a + p0)
    for (int i = threadIdx.x; i < V; i += kBlock) L.seq_a[p0 + i] = cur[i];
}

__global__ void gather_kernel(const float* __restrict__ pts, const float* __restrict__ nrm, const BatchInfo* __restrict__ bip, int batch,
                              Layout L, float* __restrict__ s_pts, float* __restrict__ s_nrm,
                              int64_t* __restrict__ s_len) {
  const BatchInfo& bi = *bip;
  const int b = blockIdx.y;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  const int V = L.meta[b].n_vox;
  if (b == 0 && o < batch) s_len[o] = L.meta[o].n_vox;
  if (o >= V) return;
  int64_t base = 0;
  for (int c = 0; c < b; c++) base += L.meta[c].n_vox;
  const int64_t p0 = bi.start[b];
  const int64_t src = p0 + L.sel[p0 + L.seq_a[p0 + o]];
  for (int d = 0; d < 3; d++) {
    s_pts[3 * (base + o) + d] = pts[3 * src + d];
    if (nrm) s_nrm[3 * (base + o) + d] = nrm[3 * src + d];
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct Sizes {
  int64_t cap_total, bk_total;
};

size_t carve(int64_t n, int batch, int64_t cap_total, int64_t bk_total, char* base, Layout* L) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    char* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  };
  Layout l;
  l.info = (BatchInfo*)take(sizeof(BatchInfo));
  l.meta = (CloudMeta*)take(sizeof(CloudMeta) * batch);
  l.table_key = (unsigned long long*)take(8 * cap_total);
  l.table_first = (int*)take(4 * cap_total);
  l.table_cnt = (int*)take(4 * cap_total);
  l.table_vox = (int*)take(4 * cap_total);
  l.slot_of_point = (int*)take(4 * n);
  l.vox_at_point = (int*)take(4 * n);
  l.vox_key = (unsigned long long*)take(8 * n);
  l.vox_cnt = (int*)take(4 * n);
  l.vox_off = (int*)take(4 * n);
  l.vox_fill = (int*)take(4 * n);
  l.members = (int*)take(4 * n);
  l.sel = (int*)take(4 * n);
  l.seq_a = (int*)take(4 * n);
  l.seq_b = (int*)take(4 * n);
  l.pos_bkt = (int*)take(4 * n);
  l.pos_next = (int*)take(4 * n);
  l.pos_w = (int*)take(4 * n);
  l.bk_first = (int*)take(4 * bk_total);
  l.bk_cnt = (int*)take(4 * bk_total);
  l.bk_head = (int*)take(4 * bk_total);
  if (L) *L = l;
  return (off + 255) & ~(size_t)255;
}

}  // namespace

extern "C" size_t se3_grid_subsample_workspace_bytes(int64_t n, int batch) {
  if (n < 0 || batch < 1 || batch > SE3_MAX_BATCH) return 0;
  // worst case over any split of n points into `batch` clouds
  int64_t cap_total = 4 * n + 32 * batch;        // sum over clouds of pow2_at_least(2 n_b) < 4 n + 32 batch
  int64_t bk_total = 3 * n + 16 * batch;         // growth-sequence member holding n_b is < 2.25 n_b + 13
  return carve(n > 0 ? n : 1, batch, cap_total, bk_total, nullptr, nullptr);
}

namespace {
// lengths_host: the clouds' sizes are known (they sum to n, grids and tables are sized by them).  lengths_dev (lengths_host == nullptr): the
// sizes live in device memory (the s_lengths of a previous stage); n is then only an UPPER bound of their sum (the rows of `points` that
// exist), the launches cover n points per cloud and the tables take their worst-case sizes -- the kernels themselves bound every access by
// the descriptor init_kernel builds on the device.
int run_grid_subsample(const float* points, const float* normals, int64_t n, const int64_t* lengths_host, const int64_t* lengths_dev,
                       int batch, float voxel_size, float* s_points, float* s_normals, int64_t* s_lengths, void* workspace,
                       size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(batch >= 1 && batch <= SE3_MAX_BATCH, SE3_ERR_INVALID_ARG, "grid_subsample: batch %d not in [1,%d]", batch,
              SE3_MAX_BATCH);
  SE3_REQUIRE(points && s_points && s_lengths && (lengths_host || lengths_dev) && workspace, SE3_ERR_INVALID_ARG,
              "grid_subsample: null pointer");
  SE3_REQUIRE(voxel_size > 0.f, SE3_ERR_INVALID_ARG, "grid_subsample: voxel size must be positive");
  SE3_REQUIRE(n >= 0 && n < (1ll << 30), SE3_ERR_UNSUPPORTED, "grid_subsample: %lld points not supported", (long long)n);
  BatchInfo bi{};
  int64_t cap_total = 0, bk_total = 0, nmax = 0, capmax = 0;
  if (lengths_host) {
    int64_t tot = 0;
    for (int b = 0; b < batch; b++) {
      SE3_REQUIRE(lengths_host[b] >= 0, SE3_ERR_INVALID_ARG, "grid_subsample: negative length");
      bi.start[b] = tot; bi.count[b] = lengths_host[b];
      bi.cap_start[b] = cap_total; bi.cap[b] = pow2_at_least(2 * lengths_host[b]);
      bi.bk_start[b] = bk_total;
      tot += lengths_host[b]; cap_total += bi.cap[b];
      bk_total += (int64_t)bucket_cap_for(lengths_host[b] > 0 ? lengths_host[b] : 1);
      if (lengths_host[b] > nmax) nmax = lengths_host[b];
      capmax = bi.cap[b] > capmax ? bi.cap[b] : capmax;
    }
    SE3_REQUIRE(tot == n, SE3_ERR_INVALID_ARG, "grid_subsample: lengths sum %lld != n %lld", (long long)tot, (long long)n);
  } else {
    cap_total = 4 * n + 32 * batch;              // the worst case over any split (se3_grid_subsample_workspace_bytes)
    bk_total = 3 * n + 16 * batch;
    nmax = n;
    capmax = pow2_at_least(2 * n);
  }
  Layout L;
  size_t need = carve(n > 0 ? n : 1, batch, cap_total, bk_total, (char*)workspace, &L);
  SE3_REQUIRE(need <= workspace_bytes, SE3_ERR_WORKSPACE, "grid_subsample: workspace %zu < %zu bytes", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const int tpb = 256;
  {
    const int64_t most = cap_total > n ? cap_total : (n > 0 ? n : 1);
    const int64_t blocks = se3_cdiv(most, (int64_t)tpb);
    init_kernel<<<(unsigned)(blocks < 2048 ? blocks : 2048), tpb, 0, st>>>(L, cap_total, n > 0 ? n : 1, bi, lengths_host ? nullptr : lengths_dev,
                                                                         batch);
  }
  bounds_kernel<<<batch, kBlock, 0, st>>>(points, L.info, voxel_size, L);
  if (nmax > 0) {
    dim3 gp((unsigned)se3_cdiv(nmax, tpb), (unsigned)batch);
    hash_kernel<<<gp, tpb, 0, st>>>(points, L.info, L);
    mark_kernel<<<dim3((unsigned)se3_cdiv(capmax, tpb), (unsigned)batch), tpb, 0, st>>>(L.info, L);
  }
  rank_kernel<<<batch, kBlock, 0, st>>>(L.info, L);
  if (nmax > 0) {
    dim3 gp((unsigned)se3_cdiv(nmax, tpb), (unsigned)batch);
    fill_kernel<<<gp, tpb, 0, st>>>(L.info, L);
    select_kernel<<<gp, tpb, 0, st>>>(points, L.info, L);
  }
  order_kernel<<<batch, kBlock, 0, st>>>(L.info, L);
  gather_kernel<<<dim3((unsigned)se3_cdiv(nmax > 0 ? nmax : 1, tpb), (unsigned)batch), tpb, 0, st>>>(
      points, normals, L.info, batch, L, s_points, s_normals, s_lengths);
  SE3_CHECK_LAUNCH("grid_subsample");
  return SE3_OK;
}
}  // namespace

extern "C" int se3_grid_subsample(const float* points, const float* normals, int64_t n, const int64_t* lengths_host,
                                  int batch, float voxel_size, float* s_points, float* s_normals, int64_t* s_lengths,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(lengths_host, SE3_ERR_INVALID_ARG, "grid_subsample: null pointer");
  return run_grid_subsample(points, normals, n, lengths_host, nullptr, batch, voxel_size, s_points, s_normals, s_lengths, workspace,
                            workspace_bytes, stream);
}

extern "C" int se3_grid_subsample_dev(const float* points, const float* normals, int64_t n_rows, const int64_t* lengths_dev, int batch,
                                      float voxel_size, float* s_points, float* s_normals, int64_t* s_lengths, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  SE3_REQUIRE(lengths_dev, SE3_ERR_INVALID_ARG, "grid_subsample_dev: null pointer");
  return run_grid_subsample(points, normals, n_rows, nullptr, lengths_dev, batch, voxel_size, s_points, s_normals, s_lengths, workspace,
                            workspace_bytes, stream);
}
