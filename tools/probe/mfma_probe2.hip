// MFMA issue rate vs waves per SIMD (4x4 tiles of 16x16x32 f16, registers only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>  // 0: 16x16x32, 1: 32x32x16
__global__ void __launch_bounds__(1024) probe(unsigned long long* out, int iters, float seed) {
  h8 fa[4], fb[4];
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 8; ++e) { fa[a][e] = (_Float16)(seed * (threadIdx.x % 7 + a + e)); fb[a][e] = (_Float16)(seed * (threadIdx.x % 5 + a * e)); }
  unsigned long long t0, t1;
  float s = 0;
  if (KIND == 0) {
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
            acc[ct][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ct], fb[j], acc[ct][j], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
  } else {
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 12; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            acc[ct][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ct], fb[j], acc[ct][j], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) s += acc[a][b][0] + acc[a][b][15];
  }
  if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
  if (s == 12345.f) out[0] = 0;
}

template <int KIND>
void run(int threads, int iters, float seed) {
  unsigned long long* d;
  (void)hipMalloc(&d, 256 * 16 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) probe<KIND><<<256, threads>>>(d, iters, seed);
  (void)hipEventRecord(e0);
  probe<KIND><<<256, threads>>>(d, iters, seed);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 16);
  (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; int n = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { sum += h[b * 16 + w]; ++n; }
  const double cyc = sum / n;
  const double n_mfma = (double)iters * 48;
  const double per_simd = n_mfma * (threads / 64) / 4.0;
  const double flop = (KIND == 0 ? 16384.0 : 32768.0) * n_mfma * (threads / 64) * 256;
  printf("%s seed %.3f waves/SIMD %d: %.2f ticks per MFMA per SIMD; kernel %.3f ms -> %.0f TFLOP/s; ticks/us %.0f\n",
         KIND == 0 ? "16x16x32" : "32x32x16", seed, threads / 256, cyc / per_simd, ms, flop / ms / 1e9, cyc / (ms * 1e3));
  (void)hipFree(d);
}

int main() {
  const int iters = 4000;
  for (float seed : {0.0f, 0.37f}) {
    for (int t : {256, 512, 768, 1024}) run<0>(t, iters, seed);
    for (int t : {256, 512, 1024}) run<1>(t, iters, seed);
  }
  return 0;
}
