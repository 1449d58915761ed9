// Hardware-semantics probes (debug entry points, used by tests/test_gpu_probe.py): they pin the two
// gfx950 facts every MFMA kernel in this library relies on -- the ds_read_b64_tr_b16 lane/element
// mapping and the mfma_f32_16x16x32_bf16 operand/result layouts -- so a layout mistake shows up as
// one failing 64-lane probe instead of a wrong GEMM.
#include "common.h"

// in: 2048 uint16 staged linearly into LDS as a [16 rows][128 B] image; every lane p of a 16-lane
// group supplies the address of (row (p>>2) + 4*g, 4 elements at column (p&3)*4) and stores the 4
// returned elements.  Expected (if the documented semantics hold): out[lane][j] = in[(4*g + j) * 64 + (lane & 15)].
__global__ void probe_tr_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x, p = lane & 15, g = lane >> 4;
  const unsigned char* a = reinterpret_cast<const unsigned char*>(lds) + (4 * g + (p >> 2)) * 128 + (p & 3) * 8;
  const s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}

// a, b: bf16 [16][32] row-major (k contiguous).  c[16][16] = a b^T computed by ONE MFMA with the
// fragment maps this library assumes: lane (g,i) holds x[i][g*8 .. g*8+7]; D[row = g*4 + r][col = i].
__global__ void probe_mfma_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, float* __restrict__ c) {
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  const s8v av = *reinterpret_cast<const s8v*>(a + i * 32 + g * 8);
  const s8v bv = *reinterpret_cast<const s8v*>(b + i * 32 + g * 8);
  f4v acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) c[(g * 4 + r) * 16 + i] = acc[r];
}

extern "C" {

int kbner_probe_tr(const uint16_t* in, uint16_t* out, void* stream) {
  hipLaunchKernelGGL(probe_tr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
  KBNER_LAUNCH_RET();
}

int kbner_probe_mfma(const bf16_t* a, const bf16_t* b, float* c, void* stream) {
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c);
  KBNER_LAUNCH_RET();
}

int kbner_abi_version(void) { return 2; }

// returns the number of visible HIP devices, or -(hipError)
int kbner_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  return e == hipSuccess ? n : -(int)e;
}

}  // extern "C"
