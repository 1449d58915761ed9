// Issue-cost probe for the GEMM K step on gfx950: what does each non-MFMA ingredient of gemm256's main loop cost next to 64
// v_mfma_f32_16x16x32_bf16 per wave, with one wave per SIMD (4 waves) and with two in lockstep (8 waves: identical code from the
// same barrier, as in the GEMM)?  Instruction order is pinned (sched_barrier between every statement, fillers in inline asm).
// Per pattern: cycles (s_memtime) per iteration for wave 0 and wave 4.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/issue_probe.hip -o tools/micro/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF(i) { acc[(i) & 31] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(i) & 3], fb[((i) >> 2) & 3], acc[(i) & 31], 0, 0, 0); SB(); }
#define SALU() { asm volatile("s_add_u32 %0, %0, 1" : "+s"(sd)); SB(); }
#define READ(i) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(rd[(i) & 3]) : "v"(laddr), "n"((((i) * 5) & 31) * 1024) : "memory"); SB(); }
#define WAITL() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB(); }
#define PIECE(j) { asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" :: "v"(goff), "s"(gsrc), "s"(ldst), "n"(((j) & 3) * 1024) : "memory", "m0"); SB(); }
#define QUAD() { asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(goff), "s"(gsrc), "s"(ldst) : "memory", "m0"); SB(); }
#define BAR() { asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); SB(); }

enum { P_MFMA = 0, P_SALU_CHUNK, P_SALU_SPREAD, P_READ_CHUNK, P_READ_SPREAD, P_DMA_BURST, P_DMA_QUADS, P_DMA_SPREAD, P_BARRIER, P_OLD, P_SPREAD_ALL,
       P_OLD_NODMA, P_SALU_ONLY, P_READ_ONLY, P_DMA_ONLY, P_BAR_ONLY, P_HALF_MFMA_SALU, NPAT };
static const char* kNames[NPAT] = {"64 mfma", "64 mfma + 40 salu, 8 x (5 salu, 8 mfma)", "64 mfma + 40 salu spread", "64 mfma + 24 ds_read_b128, 8 x (3 reads, 8 mfma)",
  "64 mfma + 24 ds_read_b128 spread", "64 mfma + 8 LDS-DMA pieces (burst, own m0 each)", "64 mfma + 2 x (m0 + 4 pieces) bursts", "64 mfma + 8 LDS-DMA pieces, one per 8 mfma",
  "64 mfma + barrier", "two-stage loop shape: barrier, 2 quads, 8 x (3 reads, 5 salu, 8 mfma)", "the same ingredients, every one between two mfma",
  "two-stage loop shape without the DMA", "40 salu alone", "24 ds_read_b128 alone", "8 LDS-DMA pieces alone (2 quads)", "barrier alone", "32 mfma + 40 salu spread"};

template <int P>
__global__ __launch_bounds__(512, 2) void k(const unsigned short* src, unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f4v acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = (f4v){0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4], fb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = 0.001f * (lane + i * 7 + wid);
    fa[i] = (bf16x8){(__bf16)v, (__bf16)(v * 2), (__bf16)(v * 3), (__bf16)-v, (__bf16)v, (__bf16)(1 - v), (__bf16)v, (__bf16)(v + 1)};
    fb[i] = (bf16x8){(__bf16)(1 - v), (__bf16)v, (__bf16)-v, (__bf16)(v * 3), (__bf16)v, (__bf16)v, (__bf16)(2 - v), (__bf16)v};
  }
  i4v rd[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  unsigned sd = 0;
  const unsigned laddr = (unsigned)(size_t)(lds_void*)smem + lane * 16;
  const unsigned goff = lane * 16;
  const unsigned short* gsrc = src + (size_t)(blockIdx.x * 8 + wid) * 2048;   // 4 KiB per wave, L2-resident
  const unsigned ldst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem + 65536 + wid * 4096);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    SB();
    if (P == P_MFMA) {
#pragma unroll
      for (int i = 0; i < 64; ++i) MF(i)
    } else if (P == P_SALU_CHUNK) {
#pragma unroll
      for (int g = 0; g < 8; ++g) { SALU() SALU() SALU() SALU() SALU()
#pragma unroll
        for (int i = 0; i < 8; ++i) MF(g * 8 + i) }
    } else if (P == P_SALU_SPREAD || P == P_HALF_MFMA_SALU) {
#pragma unroll
      for (int i = 0; i < 64; ++i) { if (P == P_SALU_SPREAD || (i & 1)) MF(i) if ((i & 7) < 5) SALU() }
    } else if (P == P_READ_CHUNK) {
#pragma unroll
      for (int g = 0; g < 8; ++g) { READ(g * 3) READ(g * 3 + 1) READ(g * 3 + 2)
#pragma unroll
        for (int i = 0; i < 8; ++i) MF(g * 8 + i) }
      WAITL()
    } else if (P == P_READ_SPREAD) {
#pragma unroll
      for (int i = 0; i < 64; ++i) { MF(i) if ((i & 7) == 1 || (i & 7) == 4 || (i & 7) == 6) READ(i) }
      WAITL()
    } else if (P == P_DMA_BURST) {
#pragma unroll
      for (int j = 0; j < 8; ++j) PIECE(j)
#pragma unroll
      for (int i = 0; i < 64; ++i) MF(i)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (P == P_DMA_QUADS) {
      QUAD() QUAD()
#pragma unroll
      for (int i = 0; i < 64; ++i) MF(i)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (P == P_DMA_SPREAD) {
#pragma unroll
      for (int i = 0; i < 64; ++i) { MF(i) if ((i & 7) == 3) PIECE(i >> 3) }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (P == P_BARRIER) {
#pragma unroll
      for (int i = 0; i < 64; ++i) MF(i)
      BAR()
    } else if (P == P_OLD || P == P_OLD_NODMA) {
      BAR()
      if (P == P_OLD) { QUAD() QUAD() }
#pragma unroll
      for (int g = 0; g < 8; ++g) { READ(g * 3) READ(g * 3 + 1) READ(g * 3 + 2) SALU() SALU() SALU() SALU() SALU()
#pragma unroll
        for (int i = 0; i < 8; ++i) MF(g * 8 + i) }
    } else if (P == P_SPREAD_ALL) {
      BAR()
#pragma unroll
      for (int i = 0; i < 64; ++i) { MF(i) if ((i & 7) == 1 || (i & 7) == 4 || (i & 7) == 6) READ(i) else if ((i & 7) == 3) PIECE(i >> 3) else if ((i & 7) != 7) SALU() else if (i < 48) SALU() }
    } else if (P == P_SALU_ONLY) {
#pragma unroll
      for (int i = 0; i < 40; ++i) SALU()
    } else if (P == P_READ_ONLY) {
#pragma unroll
      for (int i = 0; i < 24; ++i) READ(i)
      WAITL()
    } else if (P == P_DMA_ONLY) {
      QUAD() QUAD()
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (P == P_BAR_ONLY) {
      BAR()
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i][0];
  if (s == 123.4567f || sd == 0x7fffffffu || rd[0][0] + rd[1][1] + rd[2][2] + rd[3][3] == 0x12345678) out[4096] = 1;
  if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int P>
static void run(const unsigned short* src, unsigned long long* d) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<P>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int iters = 400;
  double res[2][2];
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int waves = cfg ? 8 : 4;
    hipLaunchKernelGGL((k<P>), dim3(256), dim3(waves * 64), 131072, 0, src, d, iters);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double w0 = 0, w4 = 0;
    for (int b = 0; b < 256; ++b) { w0 += (double)h[b * 8]; w4 += (double)h[b * 8 + (cfg ? 4 : 3)]; }
    res[cfg][0] = w0 / 256 / iters; res[cfg][1] = w4 / 256 / iters;
  }
  printf("%-76s 4 waves: %6.0f %6.0f   8 waves: wave0 %6.0f wave4 %6.0f\n", kNames[P], res[0][0], res[0][1], res[1][0], res[1][1]);
}

int main() {
  unsigned short* src; unsigned long long* d;
  hipMalloc(&src, (size_t)256 * 8 * 4096 + 65536); hipMemset(src, 0x3c, (size_t)256 * 8 * 4096 + 65536);
  hipMalloc(&d, 8 * 5000);
  printf("cycles per iteration (one K step's worth of work per wave)\n");
  run<P_MFMA>(src, d); run<P_SALU_ONLY>(src, d); run<P_SALU_CHUNK>(src, d); run<P_SALU_SPREAD>(src, d); run<P_HALF_MFMA_SALU>(src, d);
  run<P_READ_ONLY>(src, d); run<P_READ_CHUNK>(src, d); run<P_READ_SPREAD>(src, d);
  run<P_DMA_ONLY>(src, d); run<P_DMA_BURST>(src, d); run<P_DMA_QUADS>(src, d); run<P_DMA_SPREAD>(src, d);
  run<P_BAR_ONLY>(src, d); run<P_BARRIER>(src, d); run<P_OLD_NODMA>(src, d); run<P_OLD>(src, d); run<P_SPREAD_ALL>(src, d);
  return 0;
}
