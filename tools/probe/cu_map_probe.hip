// Which workgroups of a 512-workgroup launch (256 threads, ~73 KB of LDS: two per CU) share a CU?
// Every workgroup records XCC_ID and HW_ID (cu / sh / se fields) and its start time, then idles long enough
// that the whole grid is resident at once.  Prints, per CU, the block ids that ran on it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
struct Rec { unsigned hw, xcc; unsigned long long t0; };
__global__ void __launch_bounds__(256, 2) probe(Rec* out) {
  extern __shared__ char lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    out[blockIdx.x] = Rec{hw, xcc, t0};
  }
  lds[threadIdx.x] = 1;
  while (__builtin_amdgcn_s_memtime() < t0 + 2000000ull) __builtin_amdgcn_s_sleep(32);
}
int main() {
  const int n = 512;
  Rec* d; (void)hipMalloc(&d, n * sizeof(Rec));
  (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 73 * 1024);
  probe<<<n, 256, 73 * 1024>>>(d);
  (void)hipDeviceSynchronize();
  probe<<<n, 256, 73 * 1024>>>(d);
  (void)hipDeviceSynchronize();
  std::vector<Rec> h(n);
  (void)hipMemcpy(h.data(), d, n * sizeof(Rec), hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cus;
  unsigned long long tmin = ~0ull;
  for (auto& r : h) tmin = r.t0 < tmin ? r.t0 : tmin;
  for (int i = 0; i < n; ++i) cus[((h[i].xcc & 0xf) << 8) | ((h[i].hw >> 8) & 0x7f)].push_back(i);
  printf("%zu distinct (xcc, se, sh, cu); block ids per CU and their start offsets (cycles):\n", cus.size());
  int shown = 0, pair256 = 0, pairadj = 0, other = 0;
  for (auto& kv : cus) {
    if (shown++ < 24) {
      printf("  xcc %u se/sh/cu 0x%02x:", kv.first >> 8, kv.first & 0xff);
      for (int i : kv.second) printf(" %d(+%llu)", i, h[i].t0 - tmin);
      printf("\n");
    }
    if (kv.second.size() == 2) {
      const int d = kv.second[1] - kv.second[0];
      if (d == 256) ++pair256; else if (d == 8) ++pairadj; else ++other;
    } else ++other;
  }
  printf("CUs whose two workgroups are b and b+256: %d, b and b+8: %d, anything else: %d\n", pair256, pairadj, other);
  return 0;
}
