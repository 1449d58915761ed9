// LDS read throughput of gemm256's fragment access patterns, in bytes per clock per CU (s_memtime: clock-independent).
// One workgroup per CU, W waves, each issuing R ds_read_b128 per iteration (then s_waitcnt lgkmcnt(0)), no barrier.
//   pattern 0: linear (lane * 16 B, consecutive 1 KiB blocks)
//   pattern 1: the KC fragment image of gemm256 (row = lane & 15, 16-B chunk (lane >> 4) ^ ((row >> 1) & 7), 128-B rows)
//   pattern 2: ds_read_b64_tr_b16 pairs as the KS image reads them (two 8-byte transposed reads per fragment)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/lds_read_rate.hip -o tools/micro/lds_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s8v __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));

template <int PATTERN, int R>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 131072 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
  __syncthreads();
  unsigned base;
  if (PATTERN == 0) base = lane * 16;
  else if (PATTERN == 1) { const int row = lane & 15; base = row * 128 + ((((lane >> 4)) ^ ((row >> 1) & 7)) << 4); }
  else { const int p = lane & 15; const int r = (lane >> 4) * 8 + (p >> 2); base = r * 512 + (((0 ^ ((r & 3) | ((r >> 1) & 4)))) << 5) + ((p & 3) << 3); }
  base += (wid & 3) * 16384 * (PATTERN == 2 ? 0 : 1);
  s8v acc = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    s8v f[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (PATTERN == 2) {
        const unsigned char* a = smem + base + (i & 7) * 64 + (i >> 3) * 16384;
        const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a));
        const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a + 4 * 512));
        f[i][0] = lo[0]; f[i][1] = lo[1]; f[i][2] = lo[2]; f[i][3] = lo[3]; f[i][4] = hi[0]; f[i][5] = hi[1]; f[i][6] = hi[2]; f[i][7] = hi[3];
      } else {
        f[i] = *reinterpret_cast<const s8v*>(smem + base + (i & 7) * 2048 + (i >> 3) * 64 * (PATTERN == 1 ? 1 : 0) + (PATTERN == 0 ? (i >> 3) * 1024 : 0));
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) acc ^= f[i];
    asm volatile("" : "+v"(acc));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc[0] == 12345 && acc[3] == 999) out[1000] = 1;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int PATTERN, int R>
static void run(const char* what, int waves, unsigned long long* d) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<PATTERN, R>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int iters = 2000;
  hipLaunchKernelGGL((k<PATTERN, R>), dim3(256), dim3(waves * 64), 131072, 0, d, iters);
  hipDeviceSynchronize();
  unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0; for (int i = 0; i < 256; ++i) cyc += (double)h[i]; cyc /= 256;
  const double bytes = (double)iters * R * 1024.0 * waves;
  printf("%-44s waves %d  reads/iter %2d : %6.1f B/clk/CU  (%.1f cycles per wave-instruction per CU)\n", what, waves, R, bytes / cyc, cyc / ((double)iters * R * waves));
}

int main() {
  unsigned long long* d; hipMalloc(&d, 8 * 1024 + 64);
  for (int waves : {4, 8}) {
    run<0, 24>("ds_read_b128 linear", waves, d);
    run<1, 24>("ds_read_b128 gemm256 KC fragment pattern", waves, d);
    run<1, 8>("ds_read_b128 gemm256 KC fragment pattern", waves, d);
    run<2, 16>("2 x ds_read_b64_tr_b16 (KS fragment pattern)", waves, d);
  }
  return 0;
}
