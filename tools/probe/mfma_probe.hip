// Micro-probe: MFMA 16x16x32 f16 issue rate with 1 or 2 waves per SIMD, with and without the
// K loop's LDS read pattern (10 ds_read_b128 per 24 MFMAs) and barriers.  Prints cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: MFMA only; 1: + 10 ds_read_b128 per 24 MFMAs (counted waits); 2: 1 + s_barrier every 48 MFMAs
__global__ void __launch_bounds__(512, 2) probe(unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((float*)smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x4 acc[4][6];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 6; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
  h8 fa[4], fb[6];
  const uint32_t base = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  for (int a = 0; a < 4; ++a) fa[a] = *(h8*)(smem + base + a * 1024);
  for (int b = 0; b < 6; ++b) fb[b] = *(h8*)(smem + ((base + 8192 + b * 1024) & 65535));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (MODE >= 1) {
          if (j == 0) asm volatile("s_waitcnt lgkmcnt(4)");
          else if (j >= 2) asm volatile("s_waitcnt lgkmcnt(9)");
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[ct][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ct], fb[j], acc[ct][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 1) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j]) : "v"((base + 8192u) & 65535u), "i"(0));
          if (j < 2) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[2 * j]) : "v"(base), "i"(0));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[2 * j + 1]) : "v"(base), "i"(1024));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE == 2) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 6; ++b) s += acc[a][b][0] + acc[a][b][3];
  if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
  if (s == 12345.f) out[0] = 0;
}

template <int MODE>
void run(const char* name, int threads, int iters) {
  unsigned long long* d;
  hipMalloc(&d, 256 * 16 * 8);
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 3; ++rep) probe<MODE><<<256, threads, 65536>>>(d, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 16);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; int n = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { sum += h[b * 16 + w]; ++n; }
  const double cyc = sum / n;
  const double mfma_per_simd = (double)iters * 48 * (threads / 64) / 4.0;
  printf("%-28s threads %4d: %.0f cycles, %.2f cycles per MFMA per SIMD\n", name, threads, cyc, cyc / mfma_per_simd);
  hipFree(d);
}

int main() {
  const int iters = 400;
  run<0>("mfma only", 256, iters);
  run<0>("mfma only", 512, iters);
  run<1>("mfma + 10 ds_read/24", 256, iters);
  run<1>("mfma + 10 ds_read/24", 512, iters);
  run<2>("  + barrier per 48", 256, iters);
  run<2>("  + barrier per 48", 512, iters);
  return 0;
}
