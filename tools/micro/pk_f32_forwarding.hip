// Probe for the packed-f32 sequence behind round 4's intermittent difference: the Hermite weights of a table interval
// computed the way the embedding's record kernel used to (the vectoriser turns (2 t^3, 3 t^2) into v_pk_mul_f32 and the next-but-one
// instruction into a v_pk_fma_f32 whose LOW result reads the HIGH half of that product through op_sel) against the scalar chains that replaced
// it (se3et_amd/csrc/geo_records.h).  Both forms round identically, so every bitwise mismatch of h00 is a wrong result of the packed form.
// Each trial lets the GPU idle, then starts the probe on three streams at once beside a streaming kernel (the conditions under which the
// model showed it); mismatches are counted per quarter of the wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/pk_f32_forwarding.hip -o tools/micro/pk_f32_forwarding && tools/micro/pk_f32_forwarding [trials] [idle ms]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// the record kernel's old arithmetic, verbatim (pair_term_record of round 3)
__device__ __forceinline__ void record_packed(float x, float inv_h, int entries, int& j_out, float4& w_out) {
  const float u = x * inv_h;
  const int j = (int)floorf(u);
  const bool ok = (j >= 0) && (j + 1 < entries);
  const float tt = u - (float)j, h = 1.0f / inv_h;
  const float t2 = tt * tt, t3 = t2 * tt;
  j_out = ok ? j : -1;
  w_out = ok ? make_float4(2.f * t3 - 3.f * t2 + 1.f, -2.f * t3 + 3.f * t2, h * (t3 - 2.f * t2 + tt), h * (t3 - t2)) : make_float4(x, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ float h00_scalar(float x, float inv_h) {
  const float u = x * inv_h;
  const float tt = u - floorf(u);
  const float t2 = tt * tt, t3 = t2 * tt;
  float a = 2.f * t3, b = 3.f * t2;
  asm volatile("" : "+v"(a));
  asm volatile("" : "+v"(b));
  float wx = (a - b) + 1.f;
  asm volatile("" : "+v"(wx));
  return wx;
}

// one thread per sample, four records per thread as in geo_pair_terms_kernel (distance + three angles), results stored (the stores keep the
// instruction mix of the original: 5 x 16 bytes per thread)
__global__ __launch_bounds__(256) void probe_kernel(const float* __restrict__ xs, int n, float inv_h_d, float inv_h_a, int entries,
                                                    int4* __restrict__ jrec, float4* __restrict__ wrec, unsigned long long* __restrict__ bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int j[4];
  float4 w[4];
  float x[4];
#pragma unroll
  for (int t = 0; t < 4; t++) x[t] = xs[(size_t)t * n + i];
  record_packed(x[0], inv_h_d, entries, j[0], w[0]);
#pragma unroll
  for (int t = 1; t < 4; t++) record_packed(x[t], inv_h_a, entries, j[t], w[t]);
  jrec[i] = make_int4(j[0], j[1], j[2], j[3]);
#pragma unroll
  for (int t = 0; t < 4; t++) wrec[(size_t)i * 4 + t] = w[t];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const float want = h00_scalar(x[t], t == 0 ? inv_h_d : inv_h_a);
    if (j[t] >= 0 && __float_as_uint(want) != __float_as_uint(w[t].x)) atomicAdd(&bad[(threadIdx.x & 63) >> 4], 1ull);
  }
}

// the three instructions themselves, in the order and distance of the record kernel (inline asm: the compiler schedules the C form above with
// two instructions between the product and its reader, the record kernel had one)
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void probe_asm_kernel(const float* __restrict__ xs, int n, int reps, unsigned long long* __restrict__ bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float tt = xs[i] - floorf(xs[i]);
  const unsigned long long k = ((unsigned long long)0x40400000u << 32) | 0x40000000u;      // (2.0, 3.0) in an SGPR pair
  unsigned long long wrong = 0;
  for (int r = 0; r < reps; r++) {
    const float t2 = tt * tt, t3 = t2 * tt;
    f2v t32 = {t3, t2}, p, q;
    float d;
    asm volatile("v_pk_mul_f32 %0, %3, %4\n\t"
                 "v_sub_f32_e32 %1, %5, %6\n\t"
                 "v_pk_fma_f32 %2, %3, %4, %0 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]"
                 : "=&v"(p), "=&v"(d), "=&v"(q)
                 : "v"(t32), "s"(k), "v"(t3), "v"(t2));
    float a = 2.f * t3, b = 3.f * t2;
    asm volatile("" : "+v"(a));
    asm volatile("" : "+v"(b));
    const float want_lo = a - b, want_hi = __builtin_fmaf(t2, 3.f, -a);
    wrong += (__float_as_uint(want_lo) != __float_as_uint(q.x)) + (__float_as_uint(want_hi) != __float_as_uint(q.y));
    tt = tt * 0.61803f + 0.17f + d * 1e-9f;
    tt -= floorf(tt);
  }
  if (wrong) atomicAdd(&bad[(threadIdx.x & 63) >> 4], wrong);
}

__global__ void stream_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = a[i];
    v.x += 1.f;
    b[i] = v;
  }
}

int main(int argc, char** argv) {
  const int trials = argc > 1 ? atoi(argv[1]) : 200, idle_ms = argc > 2 ? atoi(argv[2]) : 300;
  const int churn = argc > 3 ? atoi(argv[3]) : 0;        // 1: a host thread allocates and frees device memory while the probes run (a cold start's allocator traffic)
  const int n = 385 * 385, S = 3;
  std::vector<float> h((size_t)4 * n);
  srand(1);
  for (auto& v : h) v = 60.f * (rand() / (float)RAND_MAX);
  float* xs[S]; int4* jr[S]; float4* wr[S]; unsigned long long* bad; hipStream_t st[S];
  float4 *sa, *sb;
  const size_t sn = (size_t)1 << 24;                       // 256 MB each way
  CHECK(hipMalloc(&sa, sn * 16)); CHECK(hipMalloc(&sb, sn * 16)); CHECK(hipMemset(sa, 0, sn * 16));
  CHECK(hipMalloc(&bad, 4 * sizeof(unsigned long long))); CHECK(hipMemset(bad, 0, 32));
  for (int s = 0; s < S; s++) {
    CHECK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
    CHECK(hipMalloc(&xs[s], h.size() * 4)); CHECK(hipMemcpy(xs[s], h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&jr[s], (size_t)n * 16)); CHECK(hipMalloc(&wr[s], (size_t)n * 64));
  }
  unsigned long long total[4] = {0, 0, 0, 0}, records = 0;
  int bad_trials = 0;
  for (int trial = 0; trial < trials; trial++) {
    CHECK(hipDeviceSynchronize());
    std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
    volatile bool stop = false;
    std::thread churner;
    if (churn)
      churner = std::thread([&]() {
        (void)hipSetDevice(0);
        while (!stop) {
          void* pbuf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
          for (int a = 0; a < 6; a++) (void)hipMalloc(&pbuf[a], (size_t)(16 + 48 * a) << 20);
          for (int a = 0; a < 6; a++) (void)hipFree(pbuf[a]);
        }
      });
    for (int rep = 0; rep < (churn ? 16 : 4); rep++)
      for (int s = 0; s < S; s++) {
        probe_kernel<<<(n + 255) / 256, 256, 0, st[s]>>>(xs[s], n, 64.f, 32.f, 4000, jr[s], wr[s], bad);
        probe_asm_kernel<<<(n + 255) / 256, 256, 0, st[s]>>>(xs[s], n, 64, bad);
        if (s == 0) stream_kernel<<<1024, 256, 0, st[s]>>>(sa, sb, sn);
      }
    CHECK(hipDeviceSynchronize());
    stop = true;
    if (churn) churner.join();
    unsigned long long got[4];
    CHECK(hipMemcpy(got, bad, 32, hipMemcpyDeviceToHost));
    CHECK(hipMemset(bad, 0, 32));
    records += (unsigned long long)(churn ? 16 : 4) * S * (4 + 64) * n;
    if (got[0] | got[1] | got[2] | got[3]) {
      bad_trials++;
      printf("trial %d: wrong h00 in lanes 0-15: %llu, 16-31: %llu, 32-47: %llu, 48-63: %llu\n", trial, got[0], got[1], got[2], got[3]);
    }
    for (int q = 0; q < 4; q++) total[q] += got[q];
  }
  printf("%d of %d trials with wrong results; %llu records; wrong by wave quarter: %llu %llu %llu %llu\n", bad_trials, trials, records, total[0], total[1],
         total[2], total[3]);
  return 0;
}
