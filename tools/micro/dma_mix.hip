// Micro-benchmark: what sets the landing time of gemm256's 64-KiB LDS-DMA stage?  One workgroup per CU (512 threads), the
// GEMM's two-stage skeleton -- issue stage t+1, work on stage t, s_waitcnt vmcnt(0), barrier -- with each ingredient switchable:
//   src    0: A panels wrap inside 32 MiB (L2 / MALL resident)      1: A streams from a 512-MiB matrix (HBM; rotated per launch)
//   share  CUs that read the same A panel at the same time (GEMM: the 4 column tiles of an XCD's 8 x 4 patch) -- 1 or 4
//   reads  ds_read_b128 per wave per step on the current stage (GEMM: 24 = 192 KiB per workgroup)
//   mfma   v_mfma_f32_16x16x32_bf16 per wave per step (GEMM: 64)
//   touch  L2 prefetch of the K slice `dist` steps ahead, one dword per 128-B line: 0 none, 1 every workgroup touches its A and B
//          slices (8 wave-instructions), 2 only one workgroup per sharing group (A: column 0 of the patch, B: row 0)
// B (256 x 64 slice of a 4096 x 4096 weight, 32 MiB) is shared by all workgroups with the same column index, as in the GEMM.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/dma_mix.hip -o /tmp/dma_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f4v;
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s8v;

static __device__ __forceinline__ void glds16x4(const void* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned v3, void* l) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)l);
  asm volatile(
      "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %4\n\t"
      "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
      "global_load_lds_dwordx4 %2, %4 offset:2048\n\t"
      "global_load_lds_dwordx4 %3, %4 offset:3072"
      :
      : "v"(v0), "v"(v1 - 1024u), "v"(v2 - 2048u), "v"(v3 - 3072u), "s"(sbase), "s"(dst)
      : "memory", "m0");
}

template <int READS, int MFMA>
__global__ __launch_bounds__(512, 2) void k(const unsigned short* A, const unsigned short* B, int ld, int ktiles, int share,
                                            int a_rows_total, int a_row_base, float* sink, int touch, int dist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // GEMM tile map: XCD = block & 7; inside an XCD 32 workgroups form an 8 (A panels) x 4 (B panels) patch
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int apanel = share == 4 ? xcd * 8 + (local & 7) : (int)blockIdx.x;
  const int bpanel = share == 4 ? (local >> 3) : (int)(blockIdx.x & 15);
  const size_t arow0 = (size_t)a_row_base + (size_t)(apanel * 256) % a_rows_total;
  unsigned voffa[4], voffb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wid * 4 + j;
    const int row = q * 8 + (lane >> 3), pos = lane & 7;
    voffa[j] = (unsigned)(row * ld + ((pos ^ ((row >> 1) & 7)) << 3)) * 2u;
    voffb[j] = voffa[j];
  }
  auto issue = [&](int t, unsigned char* s) {
    const unsigned short* abase = A + arow0 * ld + (size_t)t * 64;
    const unsigned short* bbase = B + (size_t)(bpanel * 256) * ld + (size_t)t * 64;
    glds16x4(abase, voffa[0], voffa[1], voffa[2], voffa[3], s + wid * 4096);
    glds16x4(bbase, voffb[0], voffb[1], voffb[2], voffb[3], s + 32768 + wid * 4096);
  };
  unsigned pf_sink = 0;
  f4v acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f4v){0.f, 0.f, 0.f, 0.f};
  issue(0, smem);
  for (int t = 0; t < ktiles; ++t) {
    if (touch && touch != 9 && t > 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // (non-touching waves: conservative, their last op is a DMA piece -- wait for it below)
    if (!touch || touch == 9 || t == 0 || !(touch == 1 || (wid >= 4 ? (local & 7) == 0 : (local >> 3) == 0))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned char* cur = smem + (t & 1) * 65536;
    unsigned char* nxt = smem + ((t + 1) & 1) * 65536;
    if (t + 1 < ktiles && touch != 9) issue(t + 1, nxt);
    if (touch && touch != 9) {
      const int tt = min(t + dist, ktiles - 1);
      const bool isb = wid >= 4;
      const bool mine = touch == 1 || (isb ? (local & 7) == 0 : (local >> 3) == 0);
      if (mine) {
        const int j = (wid & 3) * 64 + lane;
        const unsigned short* base = isb ? B + (size_t)(bpanel * 256) * ld + (size_t)tt * 64 : A + arow0 * ld + (size_t)tt * 64;
        const unsigned voff = (unsigned)(j * ld) * 2u;
        asm volatile("global_load_dword %0, %1, %2" : "+v"(pf_sink) : "v"(voff), "s"(base) : "memory");
      }
    }
    if constexpr (READS > 0) {
      // the GEMM's fragment volume: READS x 1 KiB per wave, conflict-free 16-byte reads spread over the stage
      constexpr int NR = READS > 0 ? READS : 1;
      bf16x8 f[NR];
#pragma unroll
      for (int i = 0; i < READS; ++i) {
        const s8v v = *reinterpret_cast<const s8v*>(cur + ((wid * 7 + i) & 63) * 1024 + lane * 16);
        f[i] = __builtin_bit_cast(bf16x8, v);
      }
      if constexpr (MFMA > 0) {
#pragma unroll
        for (int i = 0; i < MFMA; ++i) acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i % NR], f[(i * 5 + 1) % NR], acc[i & 15], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < READS; ++i) asm volatile("" ::"v"(f[i]));
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::"v"(pf_sink));
  if (MFMA > 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    if (s == 123.456f) sink[0] = s;
  }
}


// Ring variant (round 2's five-slot idea, here with all 160 KiB of LDS: 3 A slots + 2 B slots of 32 KiB): the A slice is issued TWO
// steps ahead, the B slice one; issue order inside a step is B(t+1) then A(t+2) so that the counted wait at the next step
// (vmcnt = the 4 A pieces just issued) covers B(t+1) and the older A(t+1).  SPREAD 1: A(t+2) is issued after half of the step's
// MFMAs instead of right after the barrier.
template <int READS, int MFMA, int SPREAD>
__global__ __launch_bounds__(512, 2) void kring(const unsigned short* A, const unsigned short* B, int ld, int ktiles, int share,
                                                int a_rows_total, int a_row_base, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int apanel = share == 4 ? xcd * 8 + (local & 7) : (int)blockIdx.x;
  const int bpanel = share == 4 ? (local >> 3) : (int)(blockIdx.x & 15);
  const size_t arow0 = (size_t)a_row_base + (size_t)(apanel * 256) % a_rows_total;
  unsigned voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wid * 4 + j;
    const int row = q * 8 + (lane >> 3), pos = lane & 7;
    voff[j] = (unsigned)(row * ld + ((pos ^ ((row >> 1) & 7)) << 3)) * 2u;
  }
  auto issueA = [&](int t) {
    glds16x4(A + arow0 * ld + (size_t)t * 64, voff[0], voff[1], voff[2], voff[3], smem + (t % 3) * 32768 + wid * 4096);
  };
  auto issueB = [&](int t) {
    glds16x4(B + (size_t)(bpanel * 256) * ld + (size_t)t * 64, voff[0], voff[1], voff[2], voff[3], smem + 98304 + (t & 1) * 32768 + wid * 4096);
  };
  f4v acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f4v){0.f, 0.f, 0.f, 0.f};
  issueA(0);
  issueB(0);
  if (ktiles > 1) issueA(1);
  for (int t = 0; t < ktiles; ++t) {
    if (t + 1 < ktiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // only A(t+1) may still fly
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned char* ca = smem + (t % 3) * 32768;
    const unsigned char* cb = smem + 98304 + (t & 1) * 32768;
    if (t + 1 < ktiles) issueB(t + 1);
    if (SPREAD == 0 && t + 2 < ktiles) issueA(t + 2);
    constexpr int NR = READS > 0 ? READS : 2;
    bf16x8 f[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const unsigned char* src = (i & 1) ? cb : ca;
      const s8v v = *reinterpret_cast<const s8v*>(src + ((wid * 7 + i) & 31) * 1024 + lane * 16);
      f[i] = __builtin_bit_cast(bf16x8, v);
    }
    if constexpr (MFMA > 0) {
#pragma unroll
      for (int i = 0; i < MFMA / 2; ++i) acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i % NR], f[(i * 5 + 1) % NR], acc[i & 15], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (SPREAD == 1 && t + 2 < ktiles) issueA(t + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = MFMA / 2; i < MFMA; ++i) acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i % NR], f[(i * 5 + 1) % NR], acc[i & 15], 0, 0, 0);
    } else {
      if (SPREAD == 1 && t + 2 < ktiles) issueA(t + 2);
#pragma unroll
      for (int i = 0; i < NR; ++i) asm volatile("" ::"v"(f[i]));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MFMA > 0) {
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s2 += acc[i][0];
    if (s2 == 123.456f) sink[0] = s2;
  }
}

template <int READS, int MFMA, int SPREAD>
static void run_ring(const char* what, const unsigned short* A, const unsigned short* B, int K, int src, float* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kring<READS, MFMA, SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  const int ktiles = K / 64;
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    const int a_rows = src == 0 ? 4096 : 16384;
    const int base = src == 0 ? 0 : (rep & 3) * 16384;
    hipEventRecord(e0);
    hipLaunchKernelGGL((kring<READS, MFMA, SPREAD>), dim3(256), dim3(512), 163840, 0, A, B, K, ktiles, 4, a_rows, base, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("%-34s A %-8s share 4 reads %2d mfma %2d A-issue %s : %.2f us per K step\n", what, src ? "HBM" : "resident", READS, MFMA,
         SPREAD ? "mid-step" : "after the barrier", best * 1e3 / ktiles);
}

template <int READS, int MFMA>
static void run(const char* what, const unsigned short* A, const unsigned short* B, int K, int src, int share, float* sink, int touch = 0,
                int dist = 2, int pad = 0) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<READS, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int ktiles = K / 64;
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    const int a_rows = src == 0 ? 4096 : 16384 * (share == 4 ? 1 : 4);
    const int base = src == 0 ? 0 : (share == 4 ? (rep & 3) * 16384 : 0);   // rotate the 128-MiB window so MALL cannot keep it
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<READS, MFMA>), dim3(256), dim3(512), 131072, 0, A, B, K + pad, ktiles, share, a_rows, base, sink, touch, dist);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("%-34s A %-8s share %d reads %2d mfma %2d touch %d dist %d ld K+%d : %.2f us per K step  (%5.1f GB/s per CU)\n", what, src ? "HBM" : "resident", share,
         READS, MFMA, touch, dist, pad, best * 1e3 / ktiles, 65536.0 * ktiles / best / 1e6);
}

int main() {
  const int M = 65536, K = 4096;
  unsigned short *A, *B; float* sink;
  hipMalloc(&A, (size_t)M * (K + 512) * 2); hipMalloc(&B, (size_t)4096 * (K + 512) * 2); hipMalloc(&sink, 64);
  hipMemset(A, 0x3c, (size_t)M * (K + 512) * 2); hipMemset(B, 0x3c, (size_t)4096 * (K + 512) * 2);
  run<0, 0>("DMA only", A, B, K, 0, 1, sink);
  run<0, 0>("DMA only", A, B, K, 0, 4, sink);
  run<0, 0>("DMA only", A, B, K, 1, 4, sink);
  run<0, 0>("DMA only", A, B, K, 1, 1, sink);
  run<24, 0>("DMA + fragment reads", A, B, K, 0, 4, sink);
  run<24, 0>("DMA + fragment reads", A, B, K, 1, 4, sink);
  run<12, 0>("DMA + half the fragment reads", A, B, K, 1, 4, sink);
  run<24, 64>("DMA + reads + MFMA", A, B, K, 0, 4, sink);
  run<24, 64>("DMA + reads + MFMA", A, B, K, 1, 4, sink);
  run<24, 64>("DMA + reads + MFMA", A, B, K, 1, 1, sink);
  run<24, 64>("reads + MFMA, NO in-loop DMA (touch 9)", A, B, K, 0, 4, sink, 9);
  run<24, 0>("reads only, NO in-loop DMA (touch 9)", A, B, K, 0, 4, sink, 9);
  run_ring<0, 0, 0>("RING: DMA only", A, B, K, 1, sink);
  run_ring<0, 0, 0>("RING: DMA only", A, B, K, 0, sink);
  run_ring<24, 64, 0>("RING: DMA + reads + MFMA", A, B, K, 1, sink);
  run_ring<24, 64, 1>("RING: DMA + reads + MFMA", A, B, K, 1, sink);
  run_ring<24, 64, 0>("RING: DMA + reads + MFMA", A, B, K, 0, sink);
  run_ring<24, 64, 1>("RING: DMA + reads + MFMA", A, B, K, 0, sink);
  run_ring<24, 64, 1>("RING: DMA + reads + MFMA", A, B, K, 1, sink);
  // leading dimension K + pad elements (row stride not a power of two): do the 256 rows of a K slice spread over more channels?
  for (int pad : {0, 64, 128, 192, 256, 8}) {
    run<0, 0>("DMA only", A, B, K, 1, 4, sink, 0, 2, pad);
    run<0, 0>("DMA only", A, B, K, 0, 4, sink, 0, 2, pad);
    run<24, 64>("DMA + reads + MFMA", A, B, K, 1, 4, sink, 0, 2, pad);
  }
  return 0;
}
