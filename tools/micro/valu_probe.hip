// VALU / MFMA issue-cost probe for gfx950: cycles (s_memtime) per loop body of the attention kernels' two phases, alone and with a
// partner wave on the same SIMD.  One workgroup per CU; waves 0-3 land on SIMDs 0-3, waves 4-7 are their partners.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/valu_probe.hip -o tools/micro/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// body kinds
enum { K_EXP = 0, K_PHASEA = 1, K_MFMA = 2, K_FMA = 3, K_PKFMA = 4, K_CVT = 5, K_NONE = 6, K_EXP_IND = 7, K_MIX1 = 8, K_MIX2 = 9, K_MIX3 = 10, K_MIXE = 11, K_MIXA = 12, K_MIXE2 = 13, K_M32 = 14, K_M32F2 = 15, K_M32F4 = 16, K_M32F5 = 17, K_M32F6 = 18, K_M32E2 = 19, K_M32E3 = 20, K_M32A = 21 };

template <int KIND>
static __device__ __forceinline__ void body32(f4v (&x)[8], f16v (&a32)[3], bf16x8 (&fa)[4], uint32_t& sink, float c0, float c1) {
  constexpr int NV = KIND == K_M32 ? 0 : KIND == K_M32F2 ? 2 : KIND == K_M32F4 ? 4 : KIND == K_M32F5 ? 5 : KIND == K_M32F6 ? 6
                   : KIND == K_M32E2 ? 2 : KIND == K_M32E3 ? 3 : 0;
  constexpr bool E = KIND == K_M32E2 || KIND == K_M32E3;
  if (KIND == K_M32A) {
    // phase A's instruction mix spread over 16 of 18 MFMAs: per 2 MFMAs one fragment = 2 pk_fma + 4 exp + 2 cvt + or
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a32[0] = MFMA32(fa[i & 3], fa[(i + 1) & 3], a32[0]);
      a32[1] = MFMA32(fa[(i + 2) & 3], fa[(i + 1) & 3], a32[1]);
      f2v lo = (f2v){x[i][0], x[i][1]} * (f2v){c0, c0} - (f2v){c1, c1};
      f2v hi = (f2v){x[i][2], x[i][3]} * (f2v){c0, c0} - (f2v){c1, c1};
      const float e0 = __builtin_amdgcn_exp2f(lo[0]), e1 = __builtin_amdgcn_exp2f(lo[1]);
      const float e2 = __builtin_amdgcn_exp2f(hi[0]), e3 = __builtin_amdgcn_exp2f(hi[1]);
      const uint32_t w0 = pack2bf(e0, e1), w1 = pack2bf(e2, e3);
      o |= w0 | w1;
      x[i][0] += __uint_as_float(w0 << 16) * 1e-30f;
    }
    a32[2] = MFMA32(fa[0], fa[1], a32[2]);
    a32[2] = MFMA32(fa[2], fa[3], a32[2]);
    sink |= o;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    a32[i % 3] = MFMA32(fa[i & 3], fa[(i + 1) & 3], a32[i % 3]);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = (i * NV + v) & 31;
      const float t = x[idx >> 2][idx & 3];
      x[idx >> 2][idx & 3] = E ? __builtin_amdgcn_exp2f(t) : t * c0 + c1;
    }
  }
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
  }
}

template <int KIND>
static __device__ __forceinline__ void body(f4v (&x)[8], f4v (&acc)[9], bf16x8 (&fa)[4], uint32_t& sink, float c0, float c1) {
  if (KIND == K_EXP) {   // 32 dependent-free exps (each on its own register)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[i][r] = __builtin_amdgcn_exp2f(x[i][r]);
  } else if (KIND == K_FMA) {   // 32 plain fmas
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[i][r] = x[i][r] * c0 + c1;
  } else if (KIND == K_PKFMA) {   // 16 packed fmas (same 32 elements)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f2v lo = (f2v){x[i][0], x[i][1]} * (f2v){c0, c0} + (f2v){c1, c1};
      f2v hi = (f2v){x[i][2], x[i][3]} * (f2v){c0, c0} + (f2v){c1, c1};
      x[i] = (f4v){lo[0], lo[1], hi[0], hi[1]};
    }
  } else if (KIND == K_CVT) {   // 16 cvt_pk
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sink ^= pack2bf(x[i][0], x[i][1]);
      sink ^= pack2bf(x[i][2], x[i][3]);
    }
  } else if (KIND == K_PHASEA) {   // the forward kernel's phase A: 16 pk_fma + 32 exp + 16 cvt + or-reduce
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f2v lo = (f2v){x[i][0], x[i][1]} * (f2v){c0, c0} - (f2v){c1, c1};
      f2v hi = (f2v){x[i][2], x[i][3]} * (f2v){c0, c0} - (f2v){c1, c1};
      const float e0 = __builtin_amdgcn_exp2f(lo[0]), e1 = __builtin_amdgcn_exp2f(lo[1]);
      const float e2 = __builtin_amdgcn_exp2f(hi[0]), e3 = __builtin_amdgcn_exp2f(hi[1]);
      const uint32_t w0 = pack2bf(e0, e1), w1 = pack2bf(e2, e3);
      o |= w0 | w1;
      x[i][0] += __uint_as_float(w0 << 16) * 1e-30f;   // keep a loop-carried dependence so nothing is hoisted
    }
    sink |= o;
  } else if (KIND == K_MFMA) {   // 36 MFMAs on 9 independent accumulators
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 9; ++i) acc[i] = MFMA(fa[k], fa[(k + 1) & 3], acc[i]);
  } else if (KIND == K_MIX1 || KIND == K_MIX2 || KIND == K_MIX3 || KIND == K_MIXE || KIND == K_MIXE2) {
    // 36 MFMAs, each followed by n plain fmas (or exps) on independent registers, order pinned by sched_group_barrier
    constexpr int NV = KIND == K_MIX1 ? 1 : KIND == K_MIX2 ? 2 : KIND == K_MIX3 ? 3 : KIND == K_MIXE ? 1 : 2;
    constexpr bool E = KIND == K_MIXE || KIND == K_MIXE2;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        acc[i] = MFMA(fa[k], fa[(k + 1) & 3], acc[i]);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int idx = ((k * 9 + i) * NV + v) & 31;
          const float t = x[idx >> 2][idx & 3];
          x[idx >> 2][idx & 3] = E ? __builtin_amdgcn_exp2f(t) : t * c0 + c1;
        }
      }
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    }
  } else if (KIND == K_MIXA) {
    // phase A's instruction mix spread over 32 MFMAs: per 4 MFMAs one fragment = 2 pk_fma + 4 exp + 2 cvt + or
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k + (i & 1) * 4] = MFMA(fa[k], fa[(k + 1) & 3], acc[k + (i & 1) * 4]);
      f2v lo = (f2v){x[i][0], x[i][1]} * (f2v){c0, c0} - (f2v){c1, c1};
      f2v hi = (f2v){x[i][2], x[i][3]} * (f2v){c0, c0} - (f2v){c1, c1};
      const float e0 = __builtin_amdgcn_exp2f(lo[0]), e1 = __builtin_amdgcn_exp2f(lo[1]);
      const float e2 = __builtin_amdgcn_exp2f(hi[0]), e3 = __builtin_amdgcn_exp2f(hi[1]);
      const uint32_t w0 = pack2bf(e0, e1), w1 = pack2bf(e2, e3);
      o |= w0 | w1;
      x[i][0] += __uint_as_float(w0 << 16) * 1e-30f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[8] = MFMA(fa[k], fa[(k + 1) & 3], acc[8]);
    sink |= o;
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
  } else if (KIND == K_EXP_IND) {   // 32 exps whose results feed 16 cvts immediately (dependent pairs)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float e0 = __builtin_amdgcn_exp2f(x[i][0]), e1 = __builtin_amdgcn_exp2f(x[i][1]);
      const float e2 = __builtin_amdgcn_exp2f(x[i][2]), e3 = __builtin_amdgcn_exp2f(x[i][3]);
      sink ^= pack2bf(e0, e1) ^ pack2bf(e2, e3);
    }
  }
}

// waves 0-3 run KA, waves 4-7 (if present) run KB; out[wave] = cycles per iteration
template <int KA, int KB>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, float c0, float c1, const float* in) {
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f4v x[8], acc[9];
  f16v a32[3];
  bf16x8 fa[4];
  uint32_t sink = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) a32[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (f4v){in[threadIdx.x & 63], in[1], in[2], in[3]} * (float)(i + 1);
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s8v t = {(short)(0x3c00 + threadIdx.x + k), 0x3f80, 0x3e00, 0x3d80, 0x3f00, 0x3e80, 0x3c80, (short)(0x3f80 + k)};
    fa[k] = __builtin_bit_cast(bf16x8, t);
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wid < 4) {
    for (int it = 0; it < iters; ++it) {
      if (KA >= K_M32) body32<KA>(x, a32, fa, sink, c0, c1); else body<KA>(x, acc, fa, sink, c0, c1);
      asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      if (KB >= K_M32) body32<KB>(x, a32, fa, sink, c0, c1); else body<KB>(x, acc, fa, sink, c0, c1);
      asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1] + x[i][2] + x[i][3];
#pragma unroll
  for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) s += a32[i][0] + a32[i][15];
  if (s == 12345.678f || sink == 0x12345u) out[100] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wid] = (float)(t1 - t0) / (float)iters;
}

template <int KA, int KB>
static void run(const char* name, int nthreads, float* dout, const float* din) {
  const int iters = 2000;
  hipLaunchKernelGGL((probe<KA, KB>), dim3(256), dim3(nthreads), 0, 0, dout, iters, 0.18f, 3.0f, din);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<KA, KB>), dim3(256), dim3(nthreads), 0, 0, dout, iters, 0.18f, 3.0f, din);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  float h[8];
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-34s %d waves: cycles/iter wave0 %.0f wave3 %.0f", name, nthreads / 64, h[0], h[3]);
  if (nthreads > 256) printf(" | wave4 %.0f wave7 %.0f", h[4], h[7]);
  printf("   (%.1f us total, %.2f GHz-equivalent)\n", ms * 1e3, h[0] * iters / (ms * 1e-3) * 1e-9);
}

int main() {
  float *dout, *din;
  hipMalloc(&dout, 1024);
  hipMalloc(&din, 1024);
  float hin[64];
  for (int i = 0; i < 64; ++i) hin[i] = 0.001f * (i + 1);
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  run<K_MFMA, K_MFMA>("36 mfma16", 256, dout, din);
  run<K_M32, K_M32>("18 mfma32", 256, dout, din);
  run<K_M32F2, K_M32F2>("18x(mfma32+2 fma)", 256, dout, din);
  run<K_M32F4, K_M32F4>("18x(mfma32+4 fma)", 256, dout, din);
  run<K_M32F5, K_M32F5>("18x(mfma32+5 fma)", 256, dout, din);
  run<K_M32F6, K_M32F6>("18x(mfma32+6 fma)", 256, dout, din);
  run<K_M32E2, K_M32E2>("18x(mfma32+2 exp)", 256, dout, din);
  run<K_M32E3, K_M32E3>("18x(mfma32+3 exp)", 256, dout, din);
  run<K_M32A, K_M32A>("18 mfma32 interleaved with phase A", 256, dout, din);
  run<K_M32A, K_M32A>("mfma32+phase A interleaved x2 waves", 512, dout, din);
  run<K_M32, K_PHASEA>("18 mfma32 | phase A", 512, dout, din);
  run<K_PHASEA, K_M32>("phase A | 18 mfma32", 512, dout, din);
  run<K_M32, K_FMA>("18 mfma32 | 32 fma", 512, dout, din);
  return 0;
}
