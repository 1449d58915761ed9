#include <hip/hip_runtime.h>
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, float* back) {
  int i = threadIdx.x;
  float a = in[4*i], b = in[4*i+1], c = in[4*i+2], d = in[4*i+3];
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
  out[i] = w;
  f2v lo = __builtin_amdgcn_cvt_pk_f32_bf8(w, false);
  f2v hi = __builtin_amdgcn_cvt_pk_f32_bf8(w, true);
  back[4*i] = lo[0]; back[4*i+1] = lo[1]; back[4*i+2] = hi[0]; back[4*i+3] = hi[1];
}
int main() {
  float h[256]; for (int i = 0; i < 256; ++i) h[i] = (i - 100) * 0.37f * (i % 7 == 0 ? 1e-4f : 1.f);
  float *di, *db; unsigned* dout;
  hipMalloc(&di, 1024); hipMalloc(&db, 1024); hipMalloc(&dout, 256);
  hipMemcpy(di, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, db);
  float b[256]; hipMemcpy(b, db, 1024, hipMemcpyDeviceToHost);
  for (int i = 0; i < 24; ++i) printf("%g -> %g\n", h[i], b[i]);
  return 0;
}
