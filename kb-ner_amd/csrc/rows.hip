// Row-indexed helper kernels for gfx950: first-subtoken gather (+scatter backward), the T-wide
// emission head (fwd / dX / dW,db) and bf16 column sums (bias gradients).  All HBM/L2-bound
// integer-indexed row moves with 16-byte vector accesses; none of this is GEMM-shaped enough
// (N = T = 29) to belong on MFMA.
//
// Replaces: the per-token pooling loop + assign_batch_features + .cpu()/.to(device) bounce
// (flair/embeddings.py:3288-3345,108-124; sequence_tagger_model.py:909), the remove_x compaction
// loop (sequence_tagger_model.py:2474-2488: the host passes row indices of the kept tokens so
// gather and compaction are ONE gather), and self.linear (sequence_tagger_model.py:1027).
#include "common.h"

#define HEAD_MAXT 64

// out[r,:] = idx[r] >= 0 ? src[idx[r],:] : 0        (bf16 rows, H % 8 == 0)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx,
                                                          bf16_t* __restrict__ out, int R, int H) {
  const int lane = threadIdx.x % 64;
  const int wave = blockIdx.x * 4 + threadIdx.x / 64;
  const int nwave = gridDim.x * 4;
  for (int r = wave; r < R; r += nwave) {
    const int s = idx[r];
    for (int h0 = lane * 8; h0 < H; h0 += 512) {
      uint4 u = make_uint4(0, 0, 0, 0);
      if (s >= 0) u = *reinterpret_cast<const uint4*>(src + (size_t)s * H + h0);
      *reinterpret_cast<uint4*>(out + (size_t)r * H + h0) = u;
    }
  }
}

// dsrc[idx[r],:] = dout[r,:] for idx[r] >= 0 (indices are unique by construction: one first
// subtoken per word token); the caller zero-fills dsrc beforehand.
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ dout, const int* __restrict__ idx,
                                                           bf16_t* __restrict__ dsrc, int R, int H) {
  const int lane = threadIdx.x % 64;
  const int wave = blockIdx.x * 4 + threadIdx.x / 64;
  const int nwave = gridDim.x * 4;
  for (int r = wave; r < R; r += nwave) {
    const int s = idx[r];
    if (s < 0) continue;
    for (int h0 = lane * 8; h0 < H; h0 += 512)
      *reinterpret_cast<uint4*>(dsrc + (size_t)s * H + h0) = *reinterpret_cast<const uint4*>(dout + (size_t)r * H + h0);
  }
}

static __device__ __forceinline__ void unpack8(const uint4 u, float* f) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = __uint_as_float(w[j] << 16);
    f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
  }
}

// emissions: out[r,t] = sum_h x[r,h] * w[t,h] + b[t]    (x bf16, w/b/out fp32; one wave per row)
__global__ __launch_bounds__(256) void head_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int R, int H,
                                                       int T) {
  const int lane = threadIdx.x % 64;
  const int wave = blockIdx.x * 4 + threadIdx.x / 64;
  const int nwave = gridDim.x * 4;
  for (int r = wave; r < R; r += nwave) {
    float xv[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int h0 = (lane + 64 * c) * 8;
      if (h0 < H) {
        unpack8(*reinterpret_cast<const uint4*>(x + (size_t)r * H + h0), xv[c]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[c][j] = 0.0f;
      }
    }
    for (int t = 0; t < T; ++t) {
      float acc = 0.0f;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int h0 = (lane + 64 * c) * 8;
        if (h0 < H) {
          const float4 a = *reinterpret_cast<const float4*>(w + (size_t)t * H + h0);
          const float4 b = *reinterpret_cast<const float4*>(w + (size_t)t * H + h0 + 4);
          acc += xv[c][0] * a.x + xv[c][1] * a.y + xv[c][2] * a.z + xv[c][3] * a.w + xv[c][4] * b.x + xv[c][5] * b.y +
                 xv[c][6] * b.z + xv[c][7] * b.w;
        }
      }
      acc = wave_sum(acc);
      if (lane == 0) out[(size_t)r * T + t] = acc + bias[t];
    }
  }
}

// The same for FEW rows (round 6: a 4-sentence step has ~64 kept tokens): one wave per (row, tag) instead of one wave per row walking
// the T tags one after the other -- 29 dependent load + reduce rounds on 16 workgroups were 33 us of latency for 2 MFLOP.
__global__ __launch_bounds__(256) void head_fwd_rt_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int R, int H,
                                                          int T) {
  const int lane = threadIdx.x % 64;
  const int id = blockIdx.x * 4 + threadIdx.x / 64;
  if (id >= R * T) return;
  const int r = id / T, t = id % T;
  float acc = 0.0f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int h0 = (lane + 64 * c) * 8;
    if (h0 < H) {
      float xv[8];
      unpack8(*reinterpret_cast<const uint4*>(x + (size_t)r * H + h0), xv);
      const float4 a = *reinterpret_cast<const float4*>(w + (size_t)t * H + h0);
      const float4 b = *reinterpret_cast<const float4*>(w + (size_t)t * H + h0 + 4);
      // (the same expression and chunk order as head_fwd_kernel: the same bits)
      acc += xv[0] * a.x + xv[1] * a.y + xv[2] * a.z + xv[3] * a.w + xv[4] * b.x + xv[5] * b.y + xv[6] * b.z + xv[7] * b.w;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) out[(size_t)r * T + t] = acc + bias[t];
}

// dx[r,h] = sum_t de[r,t] * w[t,h]        (one wave per row; bf16 out)
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* __restrict__ de, const float* __restrict__ w,
                                                          bf16_t* __restrict__ dx, int R, int H, int T) {
  const int lane = threadIdx.x % 64;
  const int wave = blockIdx.x * 4 + threadIdx.x / 64;
  const int nwave = gridDim.x * 4;
  for (int r = wave; r < R; r += nwave) {
    float acc[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[c][j] = 0.0f;
    for (int t = 0; t < T; ++t) {
      const float d = de[(size_t)r * T + t];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int h0 = (lane + 64 * c) * 8;
        if (h0 < H) {
          const float4 a = *reinterpret_cast<const float4*>(w + (size_t)t * H + h0);
          const float4 b = *reinterpret_cast<const float4*>(w + (size_t)t * H + h0 + 4);
          acc[c][0] += d * a.x; acc[c][1] += d * a.y; acc[c][2] += d * a.z; acc[c][3] += d * a.w;
          acc[c][4] += d * b.x; acc[c][5] += d * b.y; acc[c][6] += d * b.z; acc[c][7] += d * b.w;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int h0 = (lane + 64 * c) * 8;
      if (h0 < H) {
        uint4 u;
        u.x = pack2bf(acc[c][0], acc[c][1]);
        u.y = pack2bf(acc[c][2], acc[c][3]);
        u.z = pack2bf(acc[c][4], acc[c][5]);
        u.w = pack2bf(acc[c][6], acc[c][7]);
        *reinterpret_cast<uint4*>(dx + (size_t)r * H + h0) = u;
      }
    }
  }
}

// dw[t,h] += sum_r de[r,t] * x[r,h] ; db[t] += sum_r de[r,t]
// grid = (ceil(H/256), ceil(R/64)); thread owns column h; 64 rows of de staged in LDS.
// TMAX (32 for T <= 32: the 29 tags of the KB-NER dictionaries; 64 otherwise): the staged rows have a FIXED stride of TMAX floats,
// zero-padded behind T, so that a row's tag values leave LDS as TMAX / 4 broadcast ds_read_b128 and the tag loop is straight-line code
// (round 6: with the runtime stride T the compiler issued one ds_read_b32 per (row, tag) and waited for each -- 1856 exposed LDS
// latencies per block, 110 us per launch whatever the batch, 0.9 % of a 4-sentence optimizer step).
// RB rows per block: 64, or 16 when there are few rows (round 6: 64 kept tokens on 4 workgroups were 64 dependent rounds each, 31 us).
template <int TMAX, int RB = 64>
__global__ __launch_bounds__(256) void head_bwd_dw_kernel(const float* __restrict__ de, const bf16_t* __restrict__ x,
                                                          float* __restrict__ dw, float* __restrict__ db, int R, int H, int T) {
  __shared__ __attribute__((aligned(16))) float sde[RB * TMAX];
  const int h = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * RB;
  const int nr = min(RB, R - r0);
  for (int i = threadIdx.x; i < RB * TMAX; i += 256) {
    const int r = i / TMAX, t = i % TMAX;
    sde[i] = (r < nr && t < T) ? de[(size_t)(r0 + r) * T + t] : 0.0f;
  }
  __syncthreads();
  float acc[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) acc[t] = 0.0f;
  if (h < H) {
    for (int r = 0; r < nr; ++r) {
      const float xv = bf2f(x[(size_t)(r0 + r) * H + h]);
#pragma unroll
      for (int t4 = 0; t4 < TMAX / 4; ++t4) {
        const float4 d = *reinterpret_cast<const float4*>(&sde[r * TMAX + t4 * 4]);
        acc[t4 * 4 + 0] += d.x * xv;
        acc[t4 * 4 + 1] += d.y * xv;
        acc[t4 * 4 + 2] += d.z * xv;
        acc[t4 * 4 + 3] += d.w * xv;
      }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < T) atomicAdd(dw + (size_t)t * H + h, acc[t]);
  }
  if (blockIdx.x == 0 && threadIdx.x < T) {
    float s = 0.0f;
    for (int r = 0; r < nr; ++r) s += sde[r * TMAX + threadIdx.x];
    atomicAdd(db + threadIdx.x, s);
  }
}

// out[n] += sum_m x[m,n]   (bf16 in, fp32 atomics out).  Block = 64 column-threads (16-byte loads, 512 columns)
// x 4 row-threads; the 4 row partials are combined in LDS so a block issues one atomic per column.
// grid = (ceil(N/512), ceil(M/rows_per_block))
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int M, int N,
                                                     int ld, int rows_per_block) {
  __shared__ float red[4][512];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 64 + tx) * 8;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
  if (n0 < N) {
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += 4) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(x + (size_t)r * ld + n0), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ty][tx * 8 + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < 512; c += 256) {
    const int n = blockIdx.x * 512 + c;
    if (n < N) atomicAdd(out + n, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
  }
}

// same for wide rows (H > 1024: linear(2 * hidden -> T) behind the BiLSTM): the row is walked in 512-column chunks per tag
// instead of being held in registers
__global__ __launch_bounds__(256) void head_fwd_wide_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out, int R,
                                                            int H, int T) {
  const int lane = threadIdx.x % 64;
  const int wave = blockIdx.x * 4 + threadIdx.x / 64;
  const int nwave = gridDim.x * 4;
  for (int r = wave; r < R; r += nwave) {
    for (int t = 0; t < T; ++t) {
      float acc = 0.0f;
      for (int h0 = lane * 8; h0 < H; h0 += 512) {
        float xv[8];
        unpack8(*reinterpret_cast<const uint4*>(x + (size_t)r * H + h0), xv);
        const float4 a = *reinterpret_cast<const float4*>(w + (size_t)t * H + h0);
        const float4 b = *reinterpret_cast<const float4*>(w + (size_t)t * H + h0 + 4);
        acc += xv[0] * a.x + xv[1] * a.y + xv[2] * a.z + xv[3] * a.w + xv[4] * b.x + xv[5] * b.y + xv[6] * b.z + xv[7] * b.w;
      }
      acc = wave_sum(acc);
      if (lane == 0) out[(size_t)r * T + t] = acc + bias[t];
    }
  }
}

static inline int rows_grid(int R) {
  int g = (R + 3) / 4;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return g;
}

extern "C" {

int kbner_gather_rows(const bf16_t* src, const int* idx, bf16_t* out, int R, int H, void* stream) {
  KBNER_CHECK_ARG(R >= 0 && H > 0 && H % 8 == 0);
  if (R == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, src, idx, out, R, H);
  KBNER_LAUNCH_RET();
}

int kbner_scatter_rows(const bf16_t* dout, const int* idx, bf16_t* dsrc, int R, int H, void* stream) {
  KBNER_CHECK_ARG(R >= 0 && H > 0 && H % 8 == 0);
  if (R == 0) return 0;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, dout, idx, dsrc, R, H);
  KBNER_LAUNCH_RET();
}

int kbner_head_fwd(const bf16_t* x, const float* w, const float* bias, float* out, int R, int H, int T, void* stream) {
  KBNER_CHECK_ARG(R >= 0 && H > 0 && H % 8 == 0 && H <= 8192 && T > 0 && T <= HEAD_MAXT);
  if (R == 0) return 0;
  if (H <= 1024 && R <= 512)
    hipLaunchKernelGGL(head_fwd_rt_kernel, dim3((R * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, R, H, T);
  else if (H <= 1024)
    hipLaunchKernelGGL(head_fwd_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, R, H, T);
  else   // the BiLSTM tagger head of config 5: 2 x 1024 (padded) hidden columns
    hipLaunchKernelGGL(head_fwd_wide_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, R, H, T);
  KBNER_LAUNCH_RET();
}

int kbner_head_bwd_dx(const float* de, const float* w, bf16_t* dx, int R, int H, int T, void* stream) {
  KBNER_CHECK_ARG(R >= 0 && H > 0 && H % 8 == 0 && H <= 1024 && T > 0 && T <= HEAD_MAXT);
  if (R == 0) return 0;
  hipLaunchKernelGGL(head_bwd_dx_kernel, dim3(rows_grid(R)), dim3(256), 0, (hipStream_t)stream, de, w, dx, R, H, T);
  KBNER_LAUNCH_RET();
}

int kbner_head_bwd_dw(const float* de, const bf16_t* x, float* dw, float* db, int R, int H, int T, void* stream) {
  KBNER_CHECK_ARG(R >= 0 && H > 0 && T > 0 && T <= HEAD_MAXT);
  if (R == 0) return 0;
  if (T <= 32 && R <= 512)
    hipLaunchKernelGGL((head_bwd_dw_kernel<32, 16>), dim3((H + 255) / 256, (R + 15) / 16), dim3(256), 0, (hipStream_t)stream, de, x, dw,
                       db, R, H, T);
  else if (T <= 32)
    hipLaunchKernelGGL(head_bwd_dw_kernel<32>, dim3((H + 255) / 256, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream, de, x, dw,
                       db, R, H, T);
  else
    hipLaunchKernelGGL(head_bwd_dw_kernel<HEAD_MAXT>, dim3((H + 255) / 256, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream, de, x,
                       dw, db, R, H, T);
  KBNER_LAUNCH_RET();
}

int kbner_colsum(const bf16_t* x, float* out, int M, int N, int ld, void* stream) {
  KBNER_CHECK_ARG(M >= 0 && N > 0 && N % 8 == 0 && ld >= N && ld % 8 == 0);
  if (M == 0) return 0;
  int rpb = 128;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 511) / 512, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, x, out,
                     M, N, ld, rpb);
  KBNER_LAUNCH_RET();
}

}  // extern "C"

// fp32 row gather (evaluation path: compaction of the [B*n, T] emissions to the non-S-X rows before Viterbi / CRF loss):
// out[r,:] = idx[r] >= 0 ? src[idx[r],:] : 0
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                              float* __restrict__ out, int R, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R * W) return;
  const int r = i / W, c = i % W;
  const int s = idx[r];
  out[i] = s >= 0 ? src[(size_t)s * W + c] : 0.0f;
}

extern "C" int kbner_gather_rows_f32(const float* src, const int* idx, float* out, int R, int W, void* stream) {
  KBNER_CHECK_ARG(R >= 0 && W > 0);
  if (R == 0) return 0;
  hipLaunchKernelGGL(gather_rows_f32_kernel, dim3((R * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, idx, out, R, W);
  KBNER_LAUNCH_RET();
}

// Strided variant of gather_rows for the stacked-embedding concat of BASELINE config 5 (sequence_tagger_model.py:879-891:
// torch.cat of every embedding's [B, n, D_i] features): embedding i's pooled rows are written straight into its column block
// of the concatenated [rows, ld_out] matrix -- out[r, 0:H] = idx[r] >= 0 ? src[idx[r], 0:H] * mul : 0 -- so the concatenation
// never exists as a separate copy.  16-byte accesses (H % 8 == 0, ld_out % 8 == 0, out 16-byte aligned).
__global__ __launch_bounds__(256) void gather_rows_ld_kernel(const bf16_t* __restrict__ src, int ld_src, const int* __restrict__ idx,
                                                             bf16_t* __restrict__ out, int ld_out, int R, int H8) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)R * H8) return;
  const int r = (int)(i / H8), c = (int)(i % H8);
  const int s = idx[r];
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (s >= 0) v = *reinterpret_cast<const uint4*>(src + (size_t)s * ld_src + c * 8);
  *reinterpret_cast<uint4*>(out + (size_t)r * ld_out + c * 8) = v;
}

extern "C" int kbner_gather_rows_ld(const bf16_t* src, int ld_src, const int* idx, bf16_t* out, int ld_out, int R, int H,
                                    void* stream) {
  KBNER_CHECK_ARG(R >= 0 && H > 0 && H % 8 == 0 && ld_src % 8 == 0 && ld_out % 8 == 0);
  if (R == 0) return 0;
  const size_t n = (size_t)R * (H / 8);
  hipLaunchKernelGGL(gather_rows_ld_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, idx,
                     out, ld_out, R, H / 8);
  KBNER_LAUNCH_RET();
}

// out[n] += sum_r ws[r, n]: folds the per-(tile row, wave row) column sums a GEMM launched with KBNER_EPI_COLSUM_WS left in
// its workspace (rows = 2 * M / 256).  Two deterministic passes of one kernel: grid (ceil(N / 256), rows / chunk) blocks each
// fold `chunk` consecutive rows IN PLACE into the first row of their chunk (a block reads only its own chunk), then one row of
// blocks folds those first rows into out.  (A single pass with 16 blocks ran at 150 GB/s: 55 us for 8 MB.)
// block 256 columns x 4 row-lanes.
__global__ __launch_bounds__(1024) void colsum_fold_kernel(float* __restrict__ ws, int first_stride, int stride, int count, int N,
                                                           float* __restrict__ out) {
  __shared__ float part[4][256];
  const int tx = threadIdx.x & 255, ty = threadIdx.x >> 8;
  const int n = blockIdx.x * 256 + tx;
  float* base = ws + (size_t)blockIdx.y * first_stride * N;
  float acc = 0.0f;
  if (n < N)
    for (int i = ty; i < count; i += 4) acc += base[(size_t)i * stride * N + n];
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && n < N) {
    const float t = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
    if (out != nullptr) out[n] += t;
    else base[n] = t;
  }
}

extern "C" int kbner_colsum_rows_f32(float* ws, int rows, int N, float* out, void* stream) {
  KBNER_CHECK_ARG(ws != nullptr && out != nullptr && rows > 0 && N > 0);
  const int chunk = 16;
  const dim3 gx((N + 255) / 256);
  // (two passes only when there are enough partial rows to be worth a second launch: at 4 sentences per step the workspace has 32 rows
  // and the second launch was half of this call's 9.6 us, 48 times per step)
  if (rows > 4 * chunk && rows % chunk == 0) {
    hipLaunchKernelGGL(colsum_fold_kernel, dim3(gx.x, rows / chunk), dim3(1024), 0, (hipStream_t)stream, ws, chunk, 1, chunk, N,
                       (float*)nullptr);
    hipLaunchKernelGGL(colsum_fold_kernel, gx, dim3(1024), 0, (hipStream_t)stream, ws, 0, chunk, rows / chunk, N, out);
  } else {
    hipLaunchKernelGGL(colsum_fold_kernel, gx, dim3(1024), 0, (hipStream_t)stream, ws, 0, 1, rows, N, out);
  }
  KBNER_LAUNCH_RET();
}

// The single-pass fold for up to COLSUM_BATCH_MAX workspaces of one width in ONE launch (blockIdx.y picks the item): the FFN-up bias
// gradients of all layers at the end of a small-batch backward pass instead of one 4.8-us launch per layer.  Same summation order
// per column as the single pass above.
#define COLSUM_BATCH_MAX 64
struct ColsumItem {
  const float* ws;
  float* out;
  long long rows;
};
struct ColsumBatch {
  ColsumItem it[COLSUM_BATCH_MAX];
};
__global__ __launch_bounds__(1024) void colsum_fold_batched_kernel(const ColsumBatch batch, int N) {
  __shared__ float part[4][256];
  const ColsumItem& q = batch.it[blockIdx.y];
  const int tx = threadIdx.x & 255, ty = threadIdx.x >> 8;
  const int n = blockIdx.x * 256 + tx;
  const int count = (int)q.rows;
  float acc = 0.0f;
  if (n < N)
    for (int i = ty; i < count; i += 4) acc += q.ws[(size_t)i * N + n];
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && n < N) q.out[n] += (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
}

// items (HOST memory, n <= 64 records of 3 x 64 bits: ws, out -- device pointers -- and the number of rows): out[c] += sum_r ws[r, c]
extern "C" int kbner_colsum_rows_f32_batched(const long long* items, int n, int N, void* stream) {
  KBNER_CHECK_ARG(items != nullptr && n >= 0 && n <= COLSUM_BATCH_MAX && N > 0);
  if (n == 0) return 0;
  ColsumBatch b;
  for (int i = 0; i < n; ++i) {
    b.it[i].ws = reinterpret_cast<const float*>(items[3 * i]);
    b.it[i].out = reinterpret_cast<float*>(items[3 * i + 1]);
    b.it[i].rows = items[3 * i + 2];
    KBNER_CHECK_ARG(b.it[i].ws != nullptr && b.it[i].out != nullptr && b.it[i].rows > 0);
  }
  for (int i = n; i < COLSUM_BATCH_MAX; ++i) b.it[i] = b.it[0];
  hipLaunchKernelGGL(colsum_fold_batched_kernel, dim3((N + 255) / 256, n), dim3(1024), 0, (hipStream_t)stream, b, N);
  KBNER_LAUNCH_RET();
}

// fp32 row scatter (data-parallel exchange of the touched word-embedding gradient rows, kbner/dp.py): dst[idx[r],:] = rows[r,:];
// indices unique, 16-byte accesses (W % 4 == 0)
__global__ __launch_bounds__(256) void scatter_rows_f32_kernel(const float4* __restrict__ rows, const int* __restrict__ idx,
                                                               float4* __restrict__ dst, int R, int W4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)R * W4) return;
  const int r = (int)(i / W4), c = (int)(i % W4);
  const int d = idx[r];
  if (d >= 0) dst[(size_t)d * W4 + c] = rows[i];
}

extern "C" int kbner_scatter_rows_f32(const float* rows, const int* idx, float* dst, int R, int W, void* stream) {
  KBNER_CHECK_ARG(R >= 0 && W > 0 && W % 4 == 0);
  if (R == 0) return 0;
  const size_t n = (size_t)R * (W / 4);
  hipLaunchKernelGGL(scatter_rows_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(rows), idx, reinterpret_cast<float4*>(dst), R, W / 4);
  KBNER_LAUNCH_RET();
}

// fp32 row scatter-ADD, any width: dst[idx[r],:] += rows[r,:] for idx[r] >= 0 (indices unique: plain read-modify-write).
// The knowledge-distillation step (kbner/engine.py:Tagger.kd_loss) folds the gradient of the gold-label NLL, computed on the
// rows left after the remove_x compaction, back into the gradient of the all-token emissions the KD terms are defined on
// (the backward of the masked_select compaction, sequence_tagger_model.py:2474-2488, under autograd in the reference).
__global__ __launch_bounds__(256) void scatter_add_rows_f32_kernel(const float* __restrict__ rows, const int* __restrict__ idx,
                                                                   float* __restrict__ dst, int R, int W) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)R * W) return;
  const int r = (int)(i / W), c = (int)(i % W);
  const int d = idx[r];
  if (d >= 0) dst[(size_t)d * W + c] += rows[i];
}

extern "C" int kbner_scatter_add_rows_f32(const float* rows, const int* idx, float* dst, int R, int W, void* stream) {
  KBNER_CHECK_ARG(R >= 0 && W > 0);
  if (R == 0) return 0;
  KBNER_CHECK_ARG(rows != nullptr && idx != nullptr && dst != nullptr);
  const size_t n = (size_t)R * W;
  hipLaunchKernelGGL(scatter_add_rows_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, idx,
                     dst, R, W);
  KBNER_LAUNCH_RET();
}

// Multi-view training's representation term (FastSequenceTagger._calculate_multi_view_loss with calculate_l2_loss,
// sequence_tagger_model.py:1988-1996,2026-2035): mse_loss(orig view's token representations, the context view's (detached)) at the
// sentence's real tokens, forward and backward in one pass over the rows:
//     loss += sum_r w[r] * sum_h (a[r,h] - b[r,h])^2 ;   da[r,:] += 2 * gscale * w[r] * (a[r,:] - b[r,:])
// (w[r] = sentence weight / H, 0 for padding rows; da is the bf16 gradient of the pooled rows the head's backward just wrote).
// One wavefront per row.
__global__ __launch_bounds__(256) void l2_rows_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                      const float* __restrict__ w, float gscale, bf16_t* __restrict__ da,
                                                      float* __restrict__ loss, int R, int H) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float wr = w[r];
  if (wr == 0.0f) return;
  const bf16_t* ar = a + (size_t)r * H;
  const bf16_t* br = b + (size_t)r * H;
  bf16_t* dr = da + (size_t)r * H;
  float acc = 0.0f;
  for (int h = lane * 2; h < H; h += 128) {
    const f2v x = unpack2bf(*reinterpret_cast<const uint32_t*>(ar + h));
    const f2v y = unpack2bf(*reinterpret_cast<const uint32_t*>(br + h));
    const float d0 = x[0] - y[0], d1 = x[1] - y[1];
    acc += d0 * d0 + d1 * d1;
    if (da != nullptr) {
      const f2v g = unpack2bf(*reinterpret_cast<const uint32_t*>(dr + h));
      *reinterpret_cast<uint32_t*>(dr + h) = pack2bf(g[0] + 2.0f * gscale * wr * d0, g[1] + 2.0f * gscale * wr * d1);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) atomicAdd(loss, wr * acc);
}

extern "C" int kbner_l2_rows(const bf16_t* a, const bf16_t* b, const float* w, float gscale, bf16_t* da, float* loss, int R, int H,
                             void* stream) {
  KBNER_CHECK_ARG(R >= 0 && H > 0 && H % 2 == 0);
  if (R == 0) return 0;
  KBNER_CHECK_ARG(a != nullptr && b != nullptr && w != nullptr && loss != nullptr);
  hipLaunchKernelGGL(l2_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, w, gscale, da, loss, R,
                     H);
  KBNER_LAUNCH_RET();
}

// Materialise a dropout site's multiplier (tests / debugging only: the product kernels regenerate it in registers):
// out[z,i,j] = drop_keep(rowkey(seed, z*M+i), colkey(seed, z*N+j)) ? 1/(1-p) : 0.  Hidden-state sites: Z=1, [M tokens, H];
// attention-probability sites: Z = B*A heads, M = N = S.
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, int Z, int M, int N, uint32_t seed,
                                                           uint32_t thresh) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)Z * M * N) return;
  const int j = (int)(i % N);
  const int r = (int)((i / N) % M);
  const int z = (int)(i / ((size_t)M * N));
  const bool keep = drop_keep(drop_rowkey(seed, (uint32_t)(z * M + r)), drop_colkey(seed, (uint32_t)(z * N + j)), thresh);
  out[i] = keep ? drop_scale(thresh) : 0.0f;
}

extern "C" int kbner_dropout_mask(float* out, int Z, int M, int N, uint32_t seed, uint32_t thresh, void* stream) {
  KBNER_CHECK_ARG(out != nullptr && Z > 0 && M > 0 && N > 0);
  const size_t n = (size_t)Z * M * N;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, Z, M, N, seed,
                     thresh);
  KBNER_LAUNCH_RET();
}


// Split-K finish: C[m,n] = bf16( dropout( sum_s ws[s][m][n] + bias[n] ) + addend[m,n] ).  The forward / dgrad GEMMs of a small
// micro-batch have a few dozen 256x256 tiles, each a serial chain of K/64 DMA round trips; splitting a long K over up to 4
// workgroups (fp32 slabs written with plain stores by KBNER_EPI_STORE32) shortens that chain, this pass folds the slabs.
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, int splits, size_t slab, const float* __restrict__ bias,
                                                            const bf16_t* __restrict__ addend, int ldadd, bf16_t* __restrict__ C, int ldc, int M,
                                                            int N, uint32_t drop_seed, uint32_t drop_thresh) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;  // 8 consecutive columns per thread
  if (i >= (size_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  *reinterpret_cast<uint4*>(C + (size_t)m * ldc + n) = splitk_fold8_pack(ws, splits, slab, bias, addend, ldadd, m, n, N, drop_seed, drop_thresh);
}

extern "C" int kbner_splitk_finish(const float* ws, int splits, const float* bias, const bf16_t* addend, int ldadd, bf16_t* C, int ldc,
                                   int M, int N, uint32_t drop_seed, uint32_t drop_thresh, void* stream) {
  KBNER_CHECK_ARG(ws != nullptr && C != nullptr && splits >= 1 && splits <= 8 && M > 0 && N > 0 && N % 8 == 0 && ldc % 8 == 0);
  KBNER_CHECK_ARG(addend == nullptr || ldadd % 8 == 0);
  const size_t n8 = (size_t)M * N / 8;
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, splits, (size_t)M * N,
                     bias, addend, ldadd, C, ldc, M, N, drop_seed, drop_thresh);
  KBNER_LAUNCH_RET();
}
