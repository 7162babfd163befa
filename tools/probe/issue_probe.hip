// Can a SIMD run one wave's MFMAs and another wave's BN+mish-like VALU work at the same time?
// Roles per wave of a 512-thread workgroup (one per CU, waves w and w+4 share SIMD w):
//   'M' = 24 independent 16x16x32 f16 MFMAs per iteration, 'V' = the epilogue's instruction mix
//   (per 16 values: 16 v_exp_f32, 16 v_rcp_f32, 28 packed fp32 ops), 'B' = both interleaved in one
//   wave (24 MFMAs + VMIX values of VALU per iteration), '-' = exit at once.
// Prints per role the wall time (100 MHz counter) of its waves and the shader clock it saw.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Out { unsigned long long wall, clk; };

__device__ __forceinline__ void valu_mix(f32x2 (&v)[8], f32x2 sc, f32x2 sh) {
  // the shape of the kernel's BN + mish on 16 values: y = x*sc+sh; n = exp2(y*log2e); d = n*n+2n;
  // r = rcp(d+2); out = y*d*r
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f32x2 y = v[i] * sc + sh;                       // v_pk_fma
    f32x2 t = y * f32x2{1.442695f, 1.442695f};      // v_pk_mul
    f32x2 n = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    f32x2 d = n * n + (n + n);                      // v_pk_add, v_pk_fma
    f32x2 e = d + f32x2{2.f, 2.f};                  // v_pk_add
    f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    v[i] = (y * d) * r;                             // 2 v_pk_mul
  }
}

template <int VMIX>   // VALU groups of 16 values interleaved per 24 MFMAs in role 'B'
__global__ void __launch_bounds__(512, 1) probe(Out* out, const char* roles, int iters_m, int iters_v, float seed) {
  extern __shared__ char lds_force_one_wg[];
  const int w = threadIdx.x / 64;
  const char role = roles[w];
  if (role == '-') return;
  h8 fa[4], fb[6];
  if (seed < 0.f) {   // random mantissas: values uniform in [-1, 1) from an integer hash (what trained weights / activations toggle)
    auto rnd = [&](unsigned k) {
      unsigned h = (threadIdx.x * 2654435761u) ^ (k * 40503u + 0x9e3779b9u);
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
      return (_Float16)((float)(int)(h & 0xffff) / 32768.0f - 1.0f);
    };
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 8; ++e) fa[a][e] = rnd(a * 8 + e);
    for (int a = 0; a < 6; ++a) for (int e = 0; e < 8; ++e) fb[a][e] = rnd(100 + a * 8 + e);
  } else {
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 8; ++e) fa[a][e] = (_Float16)(seed * (threadIdx.x % 7 + a + e) - seed * 3);
  for (int a = 0; a < 6; ++a) for (int e = 0; e < 8; ++e) fb[a][e] = (_Float16)(seed * (threadIdx.x % 5 + a * e) - seed * 5);
  }
  f32x4 acc[4][6];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 6; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{seed * (threadIdx.x % 11) + i, seed * i - 1.f};
  const f32x2 sc = {0.5f + seed, 0.25f}, sh = {0.1f, -0.2f};
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  if (role == 'M') {
#pragma unroll 1
    for (int it = 0; it < iters_m; ++it) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ct], fb[j], acc[ct][j], 0, 0, 0);
    }
  } else if (role == 'V') {
#pragma unroll 1
    for (int it = 0; it < iters_v; ++it) valu_mix(v, sc, sh);
  } else {  // 'B'
#pragma unroll 1
    for (int it = 0; it < iters_m; ++it) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ct], fb[j], acc[ct][j], 0, 0, 0);
        if (j < VMIX) valu_mix(v, sc, sh);
      }
    }
  }
  const unsigned long long w1 = wall_clock64(), c1 = clock64();
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 6; ++b) s += acc[a][b][0] + acc[a][b][3];
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  if (threadIdx.x % 64 == 0) out[blockIdx.x * 8 + w] = Out{w1 - w0, c1 - c0};
  if (s == 12345.f) out[0].wall = 0;
}

template <int VMIX>
void run(const char* roles, int iters_m, int iters_v, float seed, const char* label) {
  Out* d; char* dr;
  (void)hipMalloc(&d, 256 * 8 * sizeof(Out));
  (void)hipMalloc(&dr, 8);
  (void)hipMemset(d, 0, 256 * 8 * sizeof(Out));
  (void)hipMemcpy(dr, roles, 8, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)probe<VMIX>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) probe<VMIX><<<256, 512, 100 * 1024>>>(d, dr, iters_m, iters_v, seed);
  (void)hipEventRecord(e0);
  probe<VMIX><<<256, 512, 100 * 1024>>>(d, dr, iters_m, iters_v, seed);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<Out> h(256 * 8);
  (void)hipMemcpy(h.data(), d, h.size() * sizeof(Out), hipMemcpyDeviceToHost);
  printf("%-34s roles %.8s seed %.2f kernel %7.3f ms |", label, roles, seed, ms);
  for (char role : {'M', 'V', 'B'}) {
    double wall = 0, clk = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) if (roles[w] == role) { wall += h[b * 8 + w].wall; clk += h[b * 8 + w].clk; ++n; }
    if (!n) continue;
    wall /= n; clk /= n;
    const double us = wall / 100.0;   // 100 MHz counter
    printf(" %c: %8.1f us, clock counter %.0f MHz", role, us, clk / us);
    if (role != 'V') {
      int waves = 0; for (int w = 0; w < 8; ++w) waves += roles[w] == role;
      const double flop = 16384.0 * 24 * iters_m * waves * 256;
      printf(", %.0f TFLOP/s", flop / us / 1e6);
    }
    if (role == 'V') printf(", %.2f ns per 16 values per wave", us * 1e3 / iters_v);
  }
  printf("\n");
  (void)hipFree(d); (void)hipFree(dr);
}

int main() {
  const int im = 30000, iv = 60000;
  for (float seed : {0.0f, 0.37f, -1.0f}) {
    run<0>("MMMMMMMM", im, iv, seed, "MFMA on both waves of a SIMD");
    run<0>("MMMM----", 2 * im, iv, seed, "MFMA on one wave per SIMD");
    run<0>("VVVVVVVV", im, iv, seed, "VALU on both waves");
    run<0>("VVVV----", im, 2 * iv, seed, "VALU on one wave per SIMD");
    run<0>("MMMMVVVV", 2 * im, 2 * iv, seed, "MFMA wave + VALU wave per SIMD");
    run<0>("MMMMVVVV", 2 * im, iv, seed, "same, half the VALU work");
    run<1>("BBBB----", 2 * im, iv, seed, "one wave, 16 values per 24 MFMAs");
    run<2>("BBBB----", 2 * im, iv, seed, "one wave, 32 values per 24 MFMAs");
    run<4>("BBBB----", 2 * im, iv, seed, "one wave, 64 values per 24 MFMAs");
    run<2>("BBBBBBBB", im, iv, seed, "two waves, 32 values per 24 MFMAs");
    run<4>("BBBBBBBB", im, iv, seed, "two waves, 64 values per 24 MFMAs");
  }
  return 0;
}
