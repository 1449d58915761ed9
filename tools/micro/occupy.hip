// A stand-in for the CUs a collective's kernels hold (tools/contention_lab.py): `blocks` workgroups of 256 threads that spin for
// `cycles` shader cycles (s_memtime) and do nothing else.  A CU that hosts one of them cannot host a 256 x 256 GEMM workgroup
// (8 waves x 236-251 VGPRs fill all four SIMDs' register files; the whole 160 KiB of LDS).
//   build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/occupy.hip -o tools/micro/liboccupy.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long cycles, unsigned* sink) {
  // 4 KiB of LDS, as a collective's kernel would hold some: the ring GEMM's workgroup needs the CU's whole 160 KiB, so the CU is
  // taken for it whatever the register file still has free (without this the 236-VGPR NT kernel co-resides with the squatter)
  __shared__ unsigned pad[1024];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned x = pad[(threadIdx.x * 7) & 1023];
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
    __builtin_amdgcn_s_sleep(16);
    x = x * 1664525u + 1013904223u;
  }
  if (x == 0xdeadbeefu) sink[0] = x;
}

extern "C" int occupy_launch(int blocks, unsigned long long cycles, unsigned* sink, void* stream) {
  if (blocks <= 0) return 0;
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cycles, sink);
  return (int)hipGetLastError();
}
