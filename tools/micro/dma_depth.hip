// Micro-benchmark: LDS-DMA (global_load_lds_dwordx4) throughput per CU as a function of the bytes kept in flight.
// Every workgroup (512 threads = 8 waves, one per CU) streams "tiles" of 64 KiB (A 32 KiB: 256 rows x 128 B at row stride
// lda; B 32 KiB likewise) exactly like gemm256's stage256, into LDS (the same LDS bytes are overwritten: this measures the
// memory pipe, not a GEMM).  DEPTH = number of 64-KiB tiles issued before the oldest is waited for (counted vmcnt).
//   mode 0: A streams through a huge matrix (HBM / MALL misses), B cycles over an 8 MiB weight (L2 hits)  ~ forward GEMM
//   mode 1: both cycle over small L2-resident buffers
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/dma_depth.hip -o /tmp/dma_depth
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((address_space(3))) void lds_void;
static __device__ __forceinline__ void glds16(const void* sbase, unsigned voff, void* l) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)l);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
template <int N> static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DEPTH>
__global__ __launch_bounds__(512) void k(const unsigned short* A, const unsigned short* B, int lda, int ldb, int ktiles, int mode,
                                         int a_rows_total) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile_m = blockIdx.x;  // each workgroup owns a 256-row panel of A, walks K
  auto issue = [&](int t) {
    // A: rows tile_m*256.., k-slice t ; B: rows (tile_m % 16)*256.., k-slice t
    const unsigned short* abase = A + (size_t)((tile_m * 256) % a_rows_total) * lda + (size_t)t * 64;
    const unsigned short* bbase = B + (size_t)((tile_m & 15) * 256) * ldb + (size_t)t * 64;
    unsigned char* s = smem + (t % DEPTH) * 65536 % (2 * 65536);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = wid * 4 + j;
      const int row = q * 8 + (lane >> 3), pos = lane & 7;
      glds16(abase, (unsigned)(row * lda + ((pos ^ ((row >> 1) & 7)) << 3)) * 2u, s + q * 1024);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = wid * 4 + j;
      const int row = q * 8 + (lane >> 3), pos = lane & 7;
      glds16(bbase, (unsigned)(row * ldb + ((pos ^ ((row >> 1) & 7)) << 3)) * 2u, s + 32768 + q * 1024);
    }
  };
  for (int t = 0; t < DEPTH - 1 && t < ktiles; ++t) issue(t);
  for (int t = 0; t < ktiles; ++t) {
    if (t + DEPTH - 1 < ktiles) issue(t + DEPTH - 1);
    // wait until tile t has landed: (DEPTH-1) younger tiles x 8 pieces may stay in flight
    if (t + DEPTH - 1 < ktiles) wait_vm<8 * (DEPTH - 1)>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
  }
}

int main() {
  const int M = 65536, K = 4096;                 // A: 512 MiB streamed; B: 4096 x 4096 (32 MiB) -> rows reused by 16 panels
  unsigned short *A, *B;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)4096 * K * 2);
  hipMemset(A, 1, (size_t)M * K * 2); hipMemset(B, 1, (size_t)4096 * K * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ktiles = K / 64;
  for (int mode = 0; mode < 2; ++mode)
    for (int depth = 2; depth <= 4; ++depth) {
      const int a_rows = mode == 0 ? M : 4096;   // mode 1: A panels wrap inside 4096 rows (32 MiB: L2/MALL resident)
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (depth == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 131072, 0, A, B, K, K, ktiles, mode, a_rows);
        if (depth == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 131072, 0, A, B, K, K, ktiles, mode, a_rows);
        if (depth == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 131072, 0, A, B, K, K, ktiles, mode, a_rows);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const double bytes = 256.0 * ktiles * 65536.0;
      printf("mode %d (A %s) tiles in flight %d: %7.1f us  %5.2f TB/s  %5.1f GB/s per CU  %.2f us per 64-KiB K tile\n", mode,
             mode == 0 ? "streams from HBM" : "L2/MALL resident", depth - 1, best * 1e3, bytes / best / 1e9, bytes / best / 1e6 / 256,
             best * 1e3 / ktiles);
    }
  return 0;
}
