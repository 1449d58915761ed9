// Price of the cross-wave dQ reduction a single-pass attention backward would need ("Attention, round 3"):
// 8 waves of one workgroup each add a 32 x 64 fp32 partial tile (32 ds_add_f32 wave-instructions) into ONE shared 8-KiB LDS
// window per 32-query chunk, next to the chunk's 40 MFMAs and ~74 VALU instructions.  Cycles per chunk (s_memtime), one
// workgroup per CU, 8 waves (2 per SIMD).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/lds_atomic_probe.hip -o tools/micro/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

enum { V_MFMA = 0, V_MFMA_ATOM_SHARED = 1, V_MFMA_ATOM_PRIVATE = 2, V_MFMA_STORE = 3, V_FULL = 4, V_FULL_ATOM = 5, V_ATOM_ONLY = 6,
       V_FULL_ATOM_PK = 7 };

template <int VAR>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* win = reinterpret_cast<float*>(smem);               // shared 8 KiB dQ window (2 windows alternate)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < 16384; i += 512) win[i] = 0.0f;
  __syncthreads();
  bf16x8 fa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) fa[i][k] = (__bf16)(0.001f * (float)((lane + i * 7 + k) & 15));
  f4v acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = (f4v){0.f, 0.f, 0.f, 0.f};
  f4v x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = (f4v){0.1f * lane, 0.2f, 0.3f, 0.4f};
  const float c0 = 0.999f, c1 = 1e-3f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    float* w = win + (it & 1) * 2048 + (VAR == V_MFMA_ATOM_PRIVATE ? wid * 2048 : 0);
    // 40 MFMAs (S, dP, dV, dK, dQ of one 32 x 32 block), the last 8 produce the dQ partial tile
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i % 10] = MFMA(fa[i & 3], fa[(i + 1) & 3], acc[i % 10]);
    if (VAR == V_FULL || VAR == V_FULL_ATOM || VAR == V_FULL_ATOM_PK) {
      // the softmax-backward VALU mix of a chunk: 16 x (mul, exp2, mul) + packs + adds (~74 instructions)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(x[i][r] * c0);
          x[i][r] = e * acc[i][r] + c1;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[i] = x[i] * c0 + (f4v){c1, c1, c1, c1};
        acc[i + 4] += x[i];
      }
    }
    if (VAR == V_MFMA_ATOM_SHARED || VAR == V_MFMA_ATOM_PRIVATE || VAR == V_FULL_ATOM || VAR == V_ATOM_ONLY) {
      // 8 accumulators x 4 values per lane = 32 ds_add_f32; lane-contiguous addresses (conflict-free within the instruction)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(w + (i * 4 + r) * 64 + lane, acc[i][r]);
    }
    if (VAR == V_FULL_ATOM_PK) {
      // the same tile through the packed form (ds_pk_add_f32 is not on gfx950's LDS; use 16 x 2 adds of neighbouring values)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          unsafeAtomicAdd(w + (i * 4 + r) * 64 + lane, acc[i][r]);
          unsafeAtomicAdd(w + (i * 4 + r + 1) * 64 + lane, acc[i][r + 1]);
        }
    }
    if (VAR == V_MFMA_STORE) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) w[(i * 4 + r) * 64 + lane + wid * 0] = acc[i][r];
    }
    if (VAR != V_MFMA && (it & 3) == 3) __syncthreads();   // the window is flushed every few chunks in the real kernel
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  s += x[0][0] + win[tid];
  out[blockIdx.x * 512 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int VAR>
static void run(const char* name, int iters) {
  float* out;
  long long* cyc;
  const int nb = 256;
  hipMalloc(&out, nb * 512 * 4);
  hipMalloc(&cyc, nb * 8 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<VAR>, dim3(nb), dim3(512), 131072, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<VAR>, dim3(nb), dim3(512), 131072, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s cycles/chunk wave0 %6.0f wave7 %6.0f   (%.1f us total for %d chunks)\n", name, (double)h[0] / iters, (double)h[7] / iters,
         ms * 1e3, iters);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  const int iters = 2000;
  run<V_MFMA>("40 mfma16", iters);
  run<V_ATOM_ONLY>("32 ds_add_f32 (shared window) only... +40 mfma", iters);
  run<V_MFMA_ATOM_SHARED>("40 mfma16 + 32 ds_add_f32, shared window", iters);
  run<V_MFMA_ATOM_PRIVATE>("40 mfma16 + 32 ds_add_f32, private windows", iters);
  run<V_MFMA_STORE>("40 mfma16 + 32 ds_write_b32", iters);
  run<V_FULL>("40 mfma16 + softmax-backward VALU mix", iters);
  run<V_FULL_ATOM>("40 mfma16 + VALU mix + 32 ds_add_f32 shared", iters);
  return 0;
}
