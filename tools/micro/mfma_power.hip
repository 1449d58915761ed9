// Micro-benchmark: sustained bf16 MFMA rate of the whole chip from registers only (no LDS, no memory in the loop), for
//   variant 0: v_mfma_f32_16x16x32_bf16  (what gemm256 / attention use; 8192 MACs, 16 operand registers read per 8192 MACs)
//   variant 1: v_mfma_f32_32x32x16_bf16  (16384 MACs per instruction for the same 16 operand registers)
// with random or all-zero operand data.  The point: DESIGN.md section 3 found the training step pinned at the 1400 W package
// cap (2.04 GHz instead of 2.4); this measures what the matrix pipe alone sustains under that cap and whether the 32x32
// shape -- half the register-file operand traffic per MAC -- sustains more.  Each run lasts ~1.5 s so the clock settles.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o /tmp/mfma_power ; run: /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f4v;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int VAR>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ ops, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  // 4 A and 4 B operand fragments per lane, loaded once
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint4 ua = ops[(size_t)tid * 8 + i], ub = ops[(size_t)tid * 8 + 4 + i];
    a[i] = *reinterpret_cast<const bf16x8*>(&ua);
    b[i] = *reinterpret_cast<const bf16x8*>(&ub);
  }
  float s = 0.f;
  if (VAR == 0) {
    f4v acc[4][4];   // 16 independent chains = a 64x64 wave tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f4v){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    f16v acc[2][2];  // 4 independent chains = the same 64x64 wave tile, same 64 accumulator registers
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      // two k-halves per trip so that one trip is the same 64x64x32 MACs as variant 0's 16 instructions
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * h], b[j + 2 * h], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  }
  if (s == 123.456f) out[tid] = s;   // keep the chains alive
}

static unsigned short bf16_of(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  const int blocks = ncu, threads = 512;   // 8 waves per CU = 2 per SIMD, as in the GEMM
  const size_t nthr = (size_t)blocks * threads;
  std::vector<unsigned short> h(nthr * 64);
  uint4* d;
  float* out;
  hipMalloc(&d, nthr * 128);
  hipMalloc(&out, nthr * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int data = 0; data < 2; ++data) {
    srand(1234);
    for (size_t i = 0; i < h.size(); ++i) h[i] = data ? bf16_of((float)rand() / RAND_MAX * 2.f - 1.f) : 0;
    hipMemcpy(d, h.data(), nthr * 128, hipMemcpyHostToDevice);
    for (int var = 0; var < 2; ++var) {
      for (int rep = 0; rep < 2; ++rep) {
        const int iters = 6000000;   // 64x64x32 MACs per wave per iteration
        hipEventRecord(e0);
        if (var == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, out, iters);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, d, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 64 * 64 * 32 * (double)iters * (nthr / 64);
        printf("%s data  %s  rep %d : %8.1f ms  %7.1f TFLOP/s\n", data ? "random" : "zero  ",
               var ? "32x32x16" : "16x16x32", rep, ms, flops / ms / 1e9);
        fflush(stdout);
      }
    }
  }
  return 0;
}
