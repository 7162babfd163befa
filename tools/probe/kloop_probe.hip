// K-loop probe (wall clock, non-zero data): 8 waves per CU, wave tile 4x6 MFMA 16x16x32 f16 per k32 step.
//  mode 0: MFMA only
//  mode 1: A and B fragments from LDS (10 ds_read_b128 per step), no barrier
//  mode 2: mode 1 + s_barrier every 2 steps (the weight ring's cadence)
//  mode 4: mode 2 + the weight ring's LDS-DMA (2 x 1 KiB per wave per 2 steps, counted vmcnt before the barrier)
//  mode 5: mode 2 + the weight stream staged through registers (2 x global_load_dwordx4 per wave per 2 steps,
//          written to LDS with ds_write_b128 one macro-step later)
//  mode 3: B from LDS (6 reads), A from global memory (4 x 16 B per lane per step, waves of one cout group share
//          addresses, 2 steps ahead), no barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512, 2) kloop(const h8* __restrict__ w, float* out, int steps, int wsteps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 131072 / 2; i += blockDim.x) ((_Float16*)smem)[i] = (_Float16)(0.01f * ((i * 7 + blockIdx.x) % 61) - 0.3f);
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int cg = wid & 1, lg = wid >> 1;
  f32x4 acc[4][6];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 6; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
  h8 fa[3][4], fb[6];
  h8 stg[2] = {};
  const uint32_t abase = cg * 4096 + lane * 16;                 // A panel image in LDS: [cg][ct][lane]
  const uint32_t bbase = 16384 + lg * 24576 + lane * 16;        // B: [lg][j][lane]
  const h8* wp = w + cg * 256 + lane;                           // global A: [step][cg][ct][lane]
  for (int ct = 0; ct < 4; ++ct) { fa[0][ct] = *(h8*)(smem + abase + ct * 1024); fa[1][ct] = fa[0][ct]; fa[2][ct] = fa[0][ct]; }
  for (int j = 0; j < 6; ++j) fb[j] = *(h8*)(smem + bbase + j * 1024);
  if (MODE == 3) {
    for (int ct = 0; ct < 4; ++ct) fa[1][ct] = wp[ct * 64];
    for (int ct = 0; ct < 4; ++ct) fa[2][ct] = wp[512 + ct * 64];
  }
  __syncthreads();
  int ws = 2;
#pragma unroll 1
  for (int s = 0; s < steps; s += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (MODE == 3) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // step u's fragments (issued two steps ago) landed
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5) {
          if (j == 0) asm volatile("s_waitcnt lgkmcnt(4)");
          else if (j >= 2) asm volatile("s_waitcnt lgkmcnt(9)");
        } else if (MODE == 3) {
          if (j == 0) asm volatile("s_waitcnt lgkmcnt(5)");
          else asm volatile("s_waitcnt lgkmcnt(5)");
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[ct][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[MODE == 3 ? u : (u & 1)][ct], fb[j], acc[ct][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 1) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j]) : "v"(bbase + ((s + u) & 7) * 16), "i"(0));
          if (MODE != 3 && j < 2) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[(u + 1) & 1][2 * j]) : "v"(abase), "i"(0));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[(u + 1) & 1][2 * j + 1]) : "v"(abase), "i"(1024));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 3) {   // refill this step's buffer with step +3
        ws = (ws + 1 == wsteps) ? 0 : ws + 1;
        const h8* p = wp + (size_t)ws * 512;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(fa[u][ct]) : "v"(p), "i"(0));
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 2 && (u & 1)) __builtin_amdgcn_s_barrier();
      if (MODE == 5 && ((s + u) & 1)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pair requested one macro-step ago
        char* lp = smem + 100000 + wid * 2048 + lane * 16;
        *(h8*)lp = stg[0];
        *(h8*)(lp + 1024) = stg[1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ws = (ws + 1 == wsteps / 2) ? 0 : ws + 1;
        const h8* gp = (const h8*)((const char*)w + (size_t)ws * 16384 + wid * 2048 + lane * 16);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg[0]) : "v"(gp));
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(stg[1]) : "v"(gp));
      }
      if (MODE == 4 && ((s + u) & 1)) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // two macro-steps stay in flight
        __builtin_amdgcn_s_barrier();
        ws = (ws + 1 == wsteps / 2) ? 0 : ws + 1;
        const char* gp = (const char*)w + (size_t)ws * 16384 + wid * 2048 + lane * 16;
        char* lp = smem + 100000 + (ws % 3) * 0 + wid * 2048;   // scratch area past the images
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + 1024), (__attribute__((address_space(3))) void*)(lp + 1024), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float sum = 0;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 6; ++b) sum += acc[a][b][0] + acc[a][b][3];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}

template <int MODE>
void run(const char* name, const h8* w, float* out, int steps, int wsteps, int waves = 8) {
  (void)hipFuncSetAttribute((const void*)kloop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) kloop<MODE><<<256, waves * 64, 131072>>>(w, out, steps, wsteps);
  (void)hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep) kloop<MODE><<<256, waves * 64, 131072>>>(w, out, steps, wsteps);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = 16384.0 * 24 * steps * waves * 256;
  printf("%-40s (%d waves/CU) %.3f ms  %.0f TFLOP/s\n", name, waves, ms, flop / ms / 1e9);
}

// 4 waves per CU (one per SIMD, up to 512 registers each), wave tile 8 x 6: 14 ds_read_b128 per 48 MFMAs.
__global__ void __launch_bounds__(256, 1) kloop_big(const h8* __restrict__ w, float* out, int steps, int barrier) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 131072 / 2; i += blockDim.x) ((_Float16*)smem)[i] = (_Float16)(0.01f * ((i * 7 + blockIdx.x) % 61) - 0.3f);
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  f32x4 acc[8][6];
  for (int a = 0; a < 8; ++a) for (int b = 0; b < 6; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
  h8 fa[2][8], fb[6];
  const uint32_t abase = lane * 16;
  const uint32_t bbase = 16384 + wid * 24576 + lane * 16;
  for (int ct = 0; ct < 8; ++ct) { fa[0][ct] = *(h8*)(smem + abase + ct * 1024); fa[1][ct] = fa[0][ct]; }
  for (int j = 0; j < 6; ++j) fb[j] = *(h8*)(smem + bbase + j * 1024);
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < steps; s += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (j == 0) asm volatile("s_waitcnt lgkmcnt(4)");
        else if (j >= 2) asm volatile("s_waitcnt lgkmcnt(13)");
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
          acc[ct][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[u][ct], fb[j], acc[ct][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j]) : "v"(bbase + ((s + u) & 7) * 16), "i"(0));
        if (j < 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[u ^ 1][4 * j + k]) : "v"(abase + k * 1024), "i"(0));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (barrier) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float sum = 0;
  for (int a = 0; a < 8; ++a) for (int b = 0; b < 6; ++b) sum += acc[a][b][0] + acc[a][b][3];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}

void run_big(const h8* w, float* out, int steps, int barrier) {
  (void)hipFuncSetAttribute((const void*)kloop_big, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) kloop_big<<<256, 256, 131072>>>(w, out, steps, barrier);
  (void)hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep) kloop_big<<<256, 256, 131072>>>(w, out, steps, barrier);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = 16384.0 * 48 * steps * 4 * 256;
  printf("%-40s %.3f ms  %.0f TFLOP/s\n", barrier ? "4 waves/CU, 8x6 tiles, LDS + barrier" : "4 waves/CU, 8x6 tiles, A+B from LDS", ms, flop / ms / 1e9);
}

int main() {
  const int steps = 36 * 300, wsteps = 36;
  h8* w; float* out;
  std::vector<_Float16> hw((size_t)wsteps * 512 * 8);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)(0.02f * (float)((i * 13) % 41) - 0.4f);
  (void)hipMalloc(&w, hw.size() * 2);
  (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 256 * 512 * 4);
  for (int round = 0; round < 3; ++round) {
    run<0>("MFMA only", w, out, steps, wsteps);
    run<1>("A+B from LDS", w, out, steps, wsteps);
    run<2>("A+B from LDS + barrier / 2 steps", w, out, steps, wsteps);
    run<3>("B from LDS, A from global (2 ahead)", w, out, steps, wsteps);
    run<4>("A+B from LDS + barrier + LDS-DMA ring", w, out, steps, wsteps);
    run<5>("A+B from LDS + barrier + reg-staged ring", w, out, steps, wsteps);
    // one wave per SIMD with the same 4 x 6 tile: can a single wave keep the MFMA pipe busy?
    run<0>("MFMA only", w, out, steps, wsteps, 4);
    run<1>("A+B from LDS", w, out, steps, wsteps, 4);
    run<3>("B from LDS, A from global (2 ahead)", w, out, steps, wsteps, 4);
    run_big(w, out, steps, 0);
    run_big(w, out, steps, 1);
  }
  return 0;
}
