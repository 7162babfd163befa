// ds_read_b128 cost per lane -> address pattern: which of the kernels' fragment layouts pay LDS bank conflicts?
// 8 waves per workgroup (one workgroup per CU), each wave issues `iters` x 8 independent ds_read_b128 with the
// pattern's addresses (plus a per-iteration slot shift, as the K loop's taps do) and sums what it reads.
// Prints LDS cycles per wave instruction per CU (128 B/clk = 8 cycles for 64 lanes x 16 B when conflict free).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ h8 lds_read128(uint32_t addr) {
  h8 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}

template <int PATTERN>
__global__ void __launch_bounds__(512, 1) probe(unsigned long long* out, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x * 16; i < 150 * 1024; i += 512 * 16) *(float4*)(smem + i) = float4{1, 2, 3, 4};
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t base;
  if (PATTERN == 0) base = lane * 16;                                             // linear
  else if (PATTERN == 1) base = ((lane & 15) + wid * 16) * 272 + (lane >> 4) * 16;   // conv16 B fragment (C_b = 128): 16 slots x 4 chunks
  else if (PATTERN == 2) base = ((lane & 31) + wid * 32) * 272 + (lane >> 5) * 16;   // conv_core B fragment: 32 slots x 2 chunks
  else if (PATTERN == 3) base = ((lane >> 5) * 4096 + (((lane >> 4) & 1) * 128 + (wid & 1) * 64 + (lane & 15)) * 16);  // conv16 A fragment
  else if (PATTERN == 4) base = ((lane & 15) + wid * 16) * 144 + (lane >> 4) * 16;   // conv16 B fragment, C_b = 64
  else if (PATTERN == 5) base = ((lane & 15) + wid * 16) * 272 + ((lane >> 4) ^ ((lane >> 2) & 3)) * 16;   // B fragment, chunk swizzled by n/4
  else if (PATTERN == 6) base = ((lane & 15) + wid * 16) * 272 + (lane >> 4) * 64;   // B fragment, chunks 64 B apart
  else if (PATTERN == 7) base = ((lane & 15) + wid * 16) * 288 + (lane >> 4) * 16;   // B fragment, slot stride 288
  else if (PATTERN == 8) base = (2 * (lane & 15) + wid * 32) * 272 + (lane >> 4) * 16;   // k_block's B fragment: every other slot
  else base = (2 * (lane & 15) + wid * 32) * 144 + (lane >> 4) * 16;                 // the same at C_b = 64
  float acc = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const uint32_t a = base + (it & 7) * 272;
    h8 v0 = lds_read128(a), v1 = lds_read128(a + 32), v2 = lds_read128(a + 64), v3 = lds_read128(a + 96);
    h8 v4 = lds_read128(a + 128), v5 = lds_read128(a + 160), v6 = lds_read128(a + 192), v7 = lds_read128(a + 224);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    acc += (float)v0[0] + (float)v1[1] + (float)v2[2] + (float)v3[3] + (float)v4[4] + (float)v5[5] + (float)v6[6] + (float)v7[7];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
  if (acc == 12345.f) *sink = acc;
}

template <int PATTERN>
__global__ void __launch_bounds__(512, 1) wprobe(unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  uint32_t base;
  if (PATTERN == 0) base = (2 * n + wid * 32) * 272 + (q >> 1) * 16 + (q & 1) * 8;
  else if (PATTERN == 1) base = (2 * n + wid * 32) * 144 + (q >> 1) * 16 + (q & 1) * 8;
  else base = wid * 8192 + lane * 8;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const uint32_t a = base + (it & 1) * 272;
    const float2 v = {1.f, (float)it};
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(k * 32) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int P>
void wr(const char* name) {
  unsigned long long* d;
  (void)hipMalloc(&d, 256 * 8 * 8);
  (void)hipFuncSetAttribute((const void*)wprobe<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  const int iters = 20000;
  wprobe<P><<<256, 512, 150 * 1024>>>(d, iters);
  wprobe<P><<<256, 512, 150 * 1024>>>(d, iters);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v; s /= h.size();
  printf("%-78s %.2f LDS cycles per ds_write_b64 (8 waves)\n", name, s / iters / 64.0);
  (void)hipFree(d);
}

template <int P>
void run(const char* name) {
  unsigned long long* d; float* sink;
  (void)hipMalloc(&d, 256 * 8 * 8); (void)hipMalloc(&sink, 4);
  (void)hipFuncSetAttribute((const void*)probe<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  const int iters = 20000;
  probe<P><<<256, 512, 150 * 1024>>>(d, iters, sink);
  probe<P><<<256, 512, 150 * 1024>>>(d, iters, sink);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v; s /= h.size();
  // 8 waves x 8 reads per iteration share one LDS
  printf("%-62s %.2f LDS cycles per ds_read_b128 (8 waves)\n", name, s / iters / 64.0);
  (void)hipFree(d); (void)hipFree(sink);
}

int main() {
  run<0>("linear, lane * 16");
  run<1>("conv16 B fragment: slot (lane & 15) * 272 + chunk (lane >> 4) * 16");
  run<2>("conv_core B fragment: slot (lane & 31) * 272 + chunk (lane >> 5) * 16");
  run<3>("conv16 A fragment (weights, 16-byte rows)");
  run<4>("conv16 B fragment at C_b = 64: slot stride 144");
  run<5>("conv16 B fragment, chunk index xor (n / 4)");
  run<6>("conv16 B fragment, chunks 64 bytes apart");
  run<7>("conv16 B fragment, slot stride 288");
  run<8>("k_block B fragment as coded: slot 2 * (lane & 15), stride 272");
  run<9>("k_block B fragment as coded at C_b = 64: stride 144");
  wr<0>("epilogue_write16 ds_write_b64: row 2n, stride 272, piece (q>>1)*16 + (q&1)*8");
  wr<1>("the same, stride 144 (C_b = 64)");
  wr<2>("ds_write_b64 linear (lane * 8)");
  return 0;
}
