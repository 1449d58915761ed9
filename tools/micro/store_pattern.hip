// Micro-benchmark: per-CU global-store throughput of a GEMM-epilogue-like burst for three lane->address patterns.
//   A: 16 rows x 64 B per wave-instruction (what gemm256's epilogue does: lane&15 = row, lane>>4 = 16-B chunk)
//   B:  8 rows x 128 B (full cache lines)
//   C:  1 row  x 1 KiB (fully contiguous)
//   D:  8 rows x 128 B with the lane order a DPP row_ror:8 exchange of the MFMA layout yields (row = lane&7)
// Every workgroup (512 threads, one per CU) stores a 256 x 256 bf16 tile (128 KiB) per iteration into a [M, ldc] matrix.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int PAT>
__global__ __launch_bounds__(512) void k(unsigned short* C, int ldc, int tiles_n, int iters) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
  for (int it = 0; it < iters; ++it) {
    const int tile = blockIdx.x + it * gridDim.x;
    const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
    unsigned short* base = C + (size_t)(m0 + wm * 128) * ldc + n0 + wn * 64;  // this wave's 128 x 64 sub-tile
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int row, col;
      if (PAT == 0) { row = (j >> 1) * 16 + (lane & 15); col = (j & 1) * 32 + (lane >> 4) * 8; }
      else if (PAT == 1) { row = j * 8 + (lane >> 3); col = (lane & 7) * 8; }
      else if (PAT == 3) { row = j * 8 + (lane & 7); col = (((lane >> 3) & 1) * 4 + (lane >> 4)) * 8; }  // D: full lines, lanes strided
      else { row = j * 8 + (lane >> 3); col = (lane & 7) * 8; base = C + (size_t)tile * 65536 + wid * 8192; }
      if (PAT == 2) *reinterpret_cast<uint4*>(base + j * 512 + lane * 8) = v;
      else *reinterpret_cast<uint4*>(base + (size_t)row * ldc + col) = v;
      v.x += 1;
    }
  }
}

int main() {
  const int M = 65536, N = 4096, iters = 16;
  unsigned short* C;
  hipMalloc(&C, (size_t)M * N * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {256, 64, 8};
  for (int gi = 0; gi < 3; ++gi)
  for (int pat = 0; pat < 4; ++pat) {
    const int G = grids[gi];
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(512), 0, 0, C, N, N / 256, iters);
      if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(512), 0, 0, C, N, N / 256, iters);
      if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(G), dim3(512), 0, 0, C, N, N / 256, iters);
      if (pat == 3) hipLaunchKernelGGL(k<3>, dim3(G), dim3(512), 0, 0, C, N, N / 256, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)G * iters * 131072.0;
      if (rep == 2) printf("grid %3d pattern %c: %8.1f us  %6.2f TB/s  %5.1f GB/s per CU  (%.2f us per 128-KiB tile)\n", G, "ABCD"[pat],
                           ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / G, ms * 1e3 / iters);
    }
  }
  return 0;
}
