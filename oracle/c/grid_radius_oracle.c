/* TEST INFRASTRUCTURE ONLY (oracle).  Plain-C restatement of the two CPU ops of the reference's
 * `geotransformer.ext`, used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker -- never by the product path (se3et_amd/), which runs these ops as HIP kernels.
 *
 *   A1 grid subsampling   restates geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-109
 *                         and grid_subsampling_cpu.h:24-74 (SampledListPoints::choose)
 *   A2 radius neighbours  restates geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91
 *                         with nanoflann's L2_Simple_Adaptor metric (extra/nanoflann/nanoflann.hpp:249-253,
 *                         radius result set :432-440, sort by distance :1286-1287) replaced by an exhaustive
 *                         scan that applies the same float32 arithmetic ((dx*dx + dy*dy) + dz*dz < r*r).
 *
 * Pinned against the genuine reference build (oracle/_ref/libref_ext.so) in tests/test_oracle_vs_reference.py
 * and against the committed fixtures tests/golden/precompute_*.npz.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no -march flags: the float expressions must not be fused).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- libstdc++ std::unordered_map<size_t, T> iteration-order model -------------------------------------
 * The reference emits voxels in the iteration order of a default-constructed std::unordered_map keyed by
 * the voxel id (grid_subsampling_cpu.cpp:28-29,66-70).  libstdc++ keeps all nodes in one singly linked
 * list; a bucket stores the node *before* its first node; a node entering an empty bucket goes to the
 * global front, otherwise to the front of its bucket; std::hash<size_t> is the identity; the bucket count
 * follows _Prime_rehash_policy (max load factor 1, growth 2) which, for one-at-a-time insertion from an
 * empty map, yields the sequence below (observed with g++ 11.4, and validated against oracle/_ref). */
static const uint64_t kBucketSeq[] = {13ull, 29ull, 59ull, 127ull, 257ull, 541ull, 1109ull, 2357ull, 5087ull,
    10273ull, 20753ull, 42043ull, 85229ull, 172933ull, 351061ull, 712697ull, 1447153ull, 2938679ull,
    5967347ull, 12117689ull, 24607243ull, 49969847ull, 101473717ull, 0ull};

typedef struct {
  int64_t *bucket;   /* per bucket: -2 empty, -1 "before begin", else node index */
  int64_t *next;     /* per node */
  const uint64_t *key;
  int64_t head;      /* before_begin.next */
  uint64_t nb;
  int level;
  int64_t count;
} OrderModel;

static void om_place(OrderModel *m, int64_t node) {
  uint64_t b = m->key[node] % m->nb;
  if (m->bucket[b] != -2) {
    int64_t prev = m->bucket[b];
    if (prev == -1) { m->next[node] = m->head; m->head = node; }
    else { m->next[node] = m->next[prev]; m->next[prev] = node; }
  } else {
    m->next[node] = m->head;
    m->head = node;
    if (m->next[node] >= 0) m->bucket[m->key[m->next[node]] % m->nb] = node;
    m->bucket[b] = -1;
  }
}

static void om_rehash(OrderModel *m, uint64_t nb) {
  m->bucket = (int64_t *)realloc(m->bucket, sizeof(int64_t) * nb);
  for (uint64_t i = 0; i < nb; i++) m->bucket[i] = -2;
  m->nb = nb;
  int64_t p = m->head;
  m->head = -1;
  while (p >= 0) {
    int64_t nx = m->next[p];
    om_place(m, p);
    p = nx;
  }
}

static void om_insert(OrderModel *m, int64_t node) {
  if (m->count + 1 > (int64_t)m->nb || m->level < 0) {
    m->level++;
    om_rehash(m, kBucketSeq[m->level]);
  }
  om_place(m, node);
  m->count++;
}

/* ---- A1 -------------------------------------------------------------------------------------------------*/
typedef struct { uint64_t key; int64_t slot; } KeySlot;

static int64_t single_grid_subsample(const float *pts, const float *nrm, int64_t n, float voxel,
                                     float *s_pts, float *s_nrm) {
  if (n == 0) return 0;
  float mn[3] = {pts[0], pts[1], pts[2]}, mx[3] = {pts[0], pts[1], pts[2]};
  for (int64_t i = 0; i < n; i++)
    for (int d = 0; d < 3; d++) {
      float v = pts[3 * i + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
  /* originCorner = floor(minCorner * (1. / voxel)) * voxel : the scale is rounded to float by operator* */
  float inv = (float)(1.0 / (double)voxel);
  float org[3];
  for (int d = 0; d < 3; d++) org[d] = floorf(mn[d] * inv) * voxel;
  uint64_t nx = (uint64_t)(floorf((mx[0] - org[0]) / voxel) + 1);
  uint64_t ny = (uint64_t)(floorf((mx[1] - org[1]) / voxel) + 1);

  uint64_t *pkey = (uint64_t *)malloc(sizeof(uint64_t) * n);
  for (int64_t i = 0; i < n; i++) {
    /* (size_t)floor(..) of the reference on x86-64 goes through the signed conversion: the -1 that float rounding can give the cloud's own
     * minimum (data/demo/src.npy) wraps to 2^64 - 1; written out so that it does not rest on the compiler's choice for a negative input */
    uint64_t ix = (uint64_t)(int64_t)floorf((pts[3 * i + 0] - org[0]) / voxel);
    uint64_t iy = (uint64_t)(int64_t)floorf((pts[3 * i + 1] - org[1]) / voxel);
    uint64_t iz = (uint64_t)(int64_t)floorf((pts[3 * i + 2] - org[2]) / voxel);
    pkey[i] = ix + nx * iy + nx * ny * iz;
  }
  /* distinct voxels in first-seen order, via an open-addressing table */
  uint64_t cap = 16;
  while (cap < (uint64_t)(2 * n)) cap <<= 1;
  int64_t *table = (int64_t *)malloc(sizeof(int64_t) * cap);
  for (uint64_t i = 0; i < cap; i++) table[i] = -1;
  uint64_t *vkey = (uint64_t *)malloc(sizeof(uint64_t) * n);
  int64_t *vox_of = (int64_t *)malloc(sizeof(int64_t) * n);
  int64_t nv = 0;
  for (int64_t i = 0; i < n; i++) {
    uint64_t h = (pkey[i] * 0x9E3779B97F4A7C15ull) & (cap - 1);
    while (table[h] >= 0 && vkey[table[h]] != pkey[i]) h = (h + 1) & (cap - 1);
    if (table[h] < 0) { table[h] = nv; vkey[nv] = pkey[i]; nv++; }
    vox_of[i] = table[h];
  }
  /* per voxel: float accumulation in input order, then the member closest to the mean (first minimum) */
  float *sum = (float *)calloc(3 * nv, sizeof(float));
  int32_t *cnt = (int32_t *)calloc(nv, sizeof(int32_t));
  for (int64_t i = 0; i < n; i++) {
    int64_t v = vox_of[i];
    cnt[v] += 1;
    for (int d = 0; d < 3; d++) sum[3 * v + d] += pts[3 * i + d];
  }
  int64_t *best = (int64_t *)malloc(sizeof(int64_t) * nv);
  double *bestd = (double *)malloc(sizeof(double) * nv);
  for (int64_t v = 0; v < nv; v++) best[v] = -1;
  for (int64_t i = 0; i < n; i++) {
    int64_t v = vox_of[i];
    float a = (float)(1.0 / (double)cnt[v]);
    float ax = sum[3 * v] * a, ay = sum[3 * v + 1] * a, az = sum[3 * v + 2] * a;
    float dx = pts[3 * i] - ax, dy = pts[3 * i + 1] - ay, dz = pts[3 * i + 2] - az;
    double dist = (double)sqrtf(dx * dx + dy * dy + dz * dz);
    if (best[v] < 0 || dist < bestd[v]) { best[v] = i; bestd[v] = dist; }
  }
  /* emission order */
  OrderModel m;
  m.bucket = NULL; m.next = (int64_t *)malloc(sizeof(int64_t) * (nv + 1)); m.key = vkey;
  m.head = -1; m.nb = 1; m.level = -1; m.count = 0;
  for (int64_t v = 0; v < nv; v++) om_insert(&m, v);
  int64_t o = 0;
  for (int64_t p = m.head; p >= 0; p = m.next[p], o++) {
    memcpy(s_pts + 3 * o, pts + 3 * best[p], 3 * sizeof(float));
    memcpy(s_nrm + 3 * o, nrm + 3 * best[p], 3 * sizeof(float));
  }
  free(m.bucket); free(m.next); free(pkey); free(table); free(vkey); free(vox_of); free(sum); free(cnt);
  free(best); free(bestd);
  return nv;
}

/* s_points / s_normals must hold n rows; returns the total number of sampled points. */
int64_t oracle_grid_subsample(const float *points, const float *normals, int64_t n, const int64_t *lengths,
                              int64_t batch, float voxel, float *s_points, float *s_normals,
                              int64_t *s_lengths) {
  int64_t start = 0, out = 0;
  (void)n;
  for (int64_t b = 0; b < batch; b++) {
    int64_t m = single_grid_subsample(points + 3 * start, normals + 3 * start, lengths[b], voxel,
                                      s_points + 3 * out, s_normals + 3 * out);
    s_lengths[b] = m;
    out += m;
    start += lengths[b];
  }
  return out;
}

/* ---- A2 -------------------------------------------------------------------------------------------------*/
typedef struct { float d2; int64_t idx; } Hit;
static int hit_cmp(const void *a, const void *b) {
  const Hit *x = (const Hit *)a, *y = (const Hit *)b;
  if (x->d2 < y->d2) return -1;
  if (x->d2 > y->d2) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);   /* ties: unspecified in the reference (std::sort) */
}

/* Writes, per query, its first min(count, cap) neighbours (ascending d2) into out[nq*cap], padded with the
 * total support size ns; returns the maximum neighbour count over all queries (may exceed cap). */
int64_t oracle_radius_neighbors(const float *q, int64_t nq, const float *s, int64_t ns, const int64_t *q_len,
                                const int64_t *s_len, int64_t batch, float radius, int64_t cap, int64_t *out) {
  float r2 = radius * radius;
  int64_t max_count = 0, q0 = 0, s0 = 0;
  Hit *hits = (Hit *)malloc(sizeof(Hit) * (ns > 0 ? ns : 1));
  for (int64_t b = 0; b < batch; b++) {
    for (int64_t i = q0; i < q0 + q_len[b]; i++) {
      int64_t c = 0;
      for (int64_t j = 0; j < s_len[b]; j++) {
        float dx = q[3 * i] - s[3 * (s0 + j)];
        float dy = q[3 * i + 1] - s[3 * (s0 + j) + 1];
        float dz = q[3 * i + 2] - s[3 * (s0 + j) + 2];
        float d2 = dx * dx;
        d2 += dy * dy;
        d2 += dz * dz;
        if (d2 < r2) { hits[c].d2 = d2; hits[c].idx = s0 + j; c++; }
      }
      qsort(hits, (size_t)c, sizeof(Hit), hit_cmp);
      if (c > max_count) max_count = c;
      for (int64_t k = 0; k < cap; k++) out[i * cap + k] = k < c ? hits[k].idx : ns;
    }
    q0 += q_len[b];
    s0 += s_len[b];
  }
  (void)nq;
  free(hits);
  return max_count;
}
