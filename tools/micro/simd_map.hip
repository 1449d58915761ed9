// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13])
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/simd_map.hip -o /tmp/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void k(unsigned* out) {
  extern __shared__ unsigned char smem[];
  const int wid = threadIdx.x >> 6;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wid] = id;
  if (threadIdx.x == 0) smem[0] = 1;
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 8 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 163840, 0, d);
    unsigned h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    int pair_w4 = 0, pair_w1 = 0;
    for (int b = 0; b < 256; ++b) {
      for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
      int s[8]; for (int w = 0; w < 8; ++w) s[w] = (h[b * 8 + w] >> 4) & 3;
      bool p4 = true, p1 = true;
      for (int w = 0; w < 4; ++w) { if (s[w] != s[w + 4]) p4 = false; }
      for (int w = 0; w < 8; w += 2) { if (s[w] != s[w + 1]) p1 = false; }
      pair_w4 += p4; pair_w1 += p1;
    }
    printf("launch %d: blocks where wave w and w+4 share a SIMD: %d / 256; where waves 2k, 2k+1 share a SIMD: %d / 256\n", rep, pair_w4, pair_w1);
    for (int b = 0; b < 4; ++b) { printf(" block %d simd ids:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("   cu %u se %u\n", (h[b*8] >> 8) & 15, (h[b*8] >> 13) & 7); }
    for (int w = 0; w < 8; ++w) printf(" wave %d on SIMD0..3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  }
  return 0;
}
