// Optimiser-side kernels for gfx950: fused HF-semantics AdamW over ONE flat fp32 arena (params,
// grads, m, v laid out identically), global grad-norm reduction, and small flat utilities.
// Pure HBM streaming: 16-byte vector accesses, 16-KiB chunks per workgroup, >> 256 workgroups; the clip
// coefficient is read from device memory so clip_grad_norm_ + step + zero_grad are one pass with
// no host synchronisation (algorithmic traffic 28 B/param + 2 B/param bf16 shadow + 4 B zeroing where the caller asks for it:
// since round 5 the engine leaves the GEMM-weight gradients for the next backward pass to overwrite, kbner/engine.py FusedAdamW).
//
// Replaces transformers==3.0.0 AdamW.step's per-tensor Python loop and
// torch.nn.utils.clip_grad_norm_ (flair/trainers/finetune_trainer.py:1010,1018):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= step_size * m / (sqrt(v) + eps) ; p -= lr*wd*p
//   step_size = lr * sqrt(1-b2^t) / (1-b1^t) is computed on the host in double and passed in.
#include "common.h"

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, size_t n,
                                                    size_t n_shadow, float step_size, float lr_wd, float b1, float b2, float eps,
                                                    const float* __restrict__ gnorm_sq, float max_norm, float grad_scale,
                                                    int zero_grad) {
  float gs = grad_scale;
  if (gnorm_sq) {
    const float norm = sqrtf(*gnorm_sq) * grad_scale;
    const float coef = max_norm / (norm + 1e-6f);
    if (coef < 1.0f) gs *= coef;
  }
  const size_t n4 = n / 4;
  // Each workgroup walks CONTIGUOUS chunks of ADAMW_CHUNK float4 per array: it touches 16 consecutive KiB of each of the eight
  // streams before it moves on, instead of 4 KiB per grid stride.  Round 5, tools/adamw_bench.py,
  // alternating libraries on two boxes: 4.27 -> 3.95 ms and 3.68 -> 3.48 ms per optimizer step (chunks of 512 / 2048 / 4096
  // float4: 3.52 / 3.49 / 4.01 ms).
  constexpr size_t ADAMW_CHUNK = 1024;
  const size_t nchunk = (n4 + ADAMW_CHUNK - 1) / ADAMW_CHUNK;
  for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x)
  for (size_t i = c * ADAMW_CHUNK + threadIdx.x; i < min(n4, (c + 1) * ADAMW_CHUNK); i += blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pa = &pp.x;
    const float* ga = &gg.x;
    float* ma = &mm.x;
    float* va = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = ga[k] * gs;
      ma[k] = ma[k] * b1 + (1.0f - b1) * gk;
      va[k] = va[k] * b2 + (1.0f - b2) * gk * gk;
      pa[k] -= step_size * (ma[k] / (sqrtf(va[k]) + eps));
      if (lr_wd != 0.0f) pa[k] -= lr_wd * pa[k];
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (shadow && (i * 4 + 3) < n_shadow) {
      uint2 u;
      u.x = pack2bf(pp.x, pp.y);
      u.y = pack2bf(pp.z, pp.w);
      reinterpret_cast<uint2*>(shadow)[i] = u;
    }
  }
}

// Row-gated variants for an EMBEDDING table [rows, width]: a row whose flag is 0 has never received a gradient, so its g, m and
// v are exactly 0 and (weight decay 0) the AdamW update leaves p unchanged: the row is not read at all.  XLM-R's word embedding
// is 46 % of the parameters and a corpus touches a small part of its 250 002 rows, so with the YAMLs' 4 sentences per optimiser
// step the dense update was 24 % of the step.  One wave per row.
// Round 6: the flag is two bits.  Bit 0 (KBNER_ROW_LIVE): the row has EVER received a gradient -- its moments are non-zero and every
// step moves it.  Bit 1 (KBNER_ROW_TOUCHED): it has received one SINCE THE LAST ZEROING update -- only then can g be non-zero.  A
// live row that is not touched has g == 0 exactly (the previous update zeroed it and nothing has written since), so the update
// neither reads nor re-zeroes its gradient (24 instead of 32 B per element moved) and the clip norm skips it altogether; the
// arithmetic is the same expression with g = 0, so the results are the dense update's bit for bit.  With the YAMLs' 4 sentences
// per step at most 2 048 of 250 002 rows are touched.
__global__ __launch_bounds__(256) void adamw_rows_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, unsigned char* __restrict__ flags, int rows,
                                                         int width, float step_size, float b1, float b2, float eps,
                                                         const float* __restrict__ gnorm_sq, float max_norm, float grad_scale,
                                                         int zero_grad) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const unsigned fl = __builtin_amdgcn_readfirstlane((unsigned)flags[row]);
  if (!fl) return;
  const bool touched = (fl & 2u) != 0;
  float gs = grad_scale;
  if (gnorm_sq) {
    const float norm = sqrtf(*gnorm_sq) * grad_scale;
    const float coef = max_norm / (norm + 1e-6f);
    if (coef < 1.0f) gs *= coef;
  }
  const size_t base = (size_t)row * width;
  for (int i = (threadIdx.x & 63) * 4; i < width; i += 256) {
    float4 pp = *reinterpret_cast<float4*>(p + base + i);
    const float4 gg = touched ? *reinterpret_cast<const float4*>(g + base + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 mm = *reinterpret_cast<float4*>(m + base + i);
    float4 vv = *reinterpret_cast<float4*>(v + base + i);
    float* pa = &pp.x;
    const float* ga = &gg.x;
    float* ma = &mm.x;
    float* va = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = ga[k] * gs;
      ma[k] = ma[k] * b1 + (1.0f - b1) * gk;
      va[k] = va[k] * b2 + (1.0f - b2) * gk * gk;
      pa[k] -= step_size * (ma[k] / (sqrtf(va[k]) + eps));
    }
    *reinterpret_cast<float4*>(p + base + i) = pp;
    *reinterpret_cast<float4*>(m + base + i) = mm;
    *reinterpret_cast<float4*>(v + base + i) = vv;
    if (zero_grad && touched) *reinterpret_cast<float4*>(g + base + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (zero_grad && touched && (threadIdx.x & 63) == 0) flags[row] = (unsigned char)(fl & ~2u);
}


// ---- LAZY rows (round 6).  A live row that receives no gradient for k steps is moved by k updates that depend on nothing but
// its own p, m, v and the steps' scalars: m <- b1 m, v <- b2 v, p <- p - step_s * m / (sqrt(v) + eps).  The dense optimizer
// streams 24 B per element through HBM for each of them (7 GB per step for XLM-R's 250 002 x 1024 table once every row is live:
// 1.0 ms of the YAMLs' 11-ms step); the only READER of a row is the embedding lookup of a batch that contains its id.  So the k
// updates are applied -- the same fp32 operations in the same order, k trips through one loop in registers -- when the row is next
// needed: by the lookup (kbner_adamw_rows_catchup on the batch's ids, before the forward pass) or by an update that brings it a
// gradient.  row_t[r] = the last step applied to row r (-1: never live), *clock = the optimizer's step count, hist[s & mask] =
// step s's step_size.  Bit-identical to the eager kernels (tests: test_lazy_embedding_rows_equal_eager).
struct RowState {
  float4 p[4], m[4], v[4];
};

// (m * b1 + (1 - b1) * 0 and v * b2 + (1 - b2) * 0 * 0 of the eager kernel, without the terms that are zero: the same values -- at
//  most the sign of a zero differs, which no later operation of the update can turn into a different number)
__device__ __forceinline__ void adam_zero_grad_step(float& p, float& m, float& v, float step, float b1, float b2, float eps) {
  m = m * b1;
  v = v * b2;
  p -= step * (m / (sqrtf(v) + eps));
}

// one wave: the zero-gradient updates of steps (from, to] on one row of width <= 1024 held in registers
template <int NV>
__device__ __forceinline__ void row_catch_up(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, size_t base, int width,
                                             int lane, int from, int to, const float* __restrict__ hist, int mask, float b1, float b2,
                                             float eps, RowState& st, bool load) {
  if (load) {
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int i = lane * 4 + c * 256;
      if (i < width) {
        st.p[c] = *reinterpret_cast<const float4*>(p + base + i);
        st.m[c] = *reinterpret_cast<const float4*>(m + base + i);
        st.v[c] = *reinterpret_cast<const float4*>(v + base + i);
      }
    }
  }
  for (int s = from + 1; s <= to; ++s) {
    const float step = hist[s & mask];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      adam_zero_grad_step(st.p[c].x, st.m[c].x, st.v[c].x, step, b1, b2, eps);
      adam_zero_grad_step(st.p[c].y, st.m[c].y, st.v[c].y, step, b1, b2, eps);
      adam_zero_grad_step(st.p[c].z, st.m[c].z, st.v[c].z, step, b1, b2, eps);
      adam_zero_grad_step(st.p[c].w, st.m[c].w, st.v[c].w, step, b1, b2, eps);
    }
  }
}

template <int NV>
__device__ __forceinline__ void row_store(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, size_t base, int width,
                                          int lane, const RowState& st) {
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int i = lane * 4 + c * 256;
    if (i < width) {
      *reinterpret_cast<float4*>(p + base + i) = st.p[c];
      *reinterpret_cast<float4*>(m + base + i) = st.m[c];
      *reinterpret_cast<float4*>(v + base + i) = st.v[c];
    }
  }
}

// rows ids[0..n) (ids == nullptr: every row) brought to step *clock.  One WORKGROUP per entry, one column per thread: what a row
// owes is geometric in the steady state (mean rows / distinct-ids-per-step: 121 steps at the YAMLs' 2 048 sub-tokens, the longest of
// a step's 2 048 rows ~900), and the launch lasts as long as its longest row -- with one wave per row (16 columns per lane) that
// was 0.9 ms, as much as the eager update it replaces; spread over 16 waves it is ~60 us.  The step sizes of up to 1024 owed steps
// are staged in LDS at a time.  An id that occurs several times is claimed by the first workgroup to swap the clock into row_t
// (the others find it current and leave: the claimant finishes inside this launch, the reader is a later launch).
__global__ __launch_bounds__(1024) void adamw_rows_catchup_kernel(const int* __restrict__ ids, int n, float* __restrict__ p,
                                                                  float* __restrict__ m, float* __restrict__ v,
                                                                  const unsigned char* __restrict__ flags, int* __restrict__ row_t,
                                                                  const int* __restrict__ clock, const float* __restrict__ hist,
                                                                  int mask, int rows, int width, float b1, float b2, float eps) {
  __shared__ int s_from;
  __shared__ float s_step[1024];
  const int row = ids ? ids[blockIdx.x] : (int)blockIdx.x;
  if (row < 0 || row >= rows) return;
  const int now = *clock;
  if (threadIdx.x == 0) s_from = (flags[row] & 1) ? atomicExch(row_t + row, now) : now;
  __syncthreads();
  int from = s_from;
  if (from < 0 || from >= now) return;   // never live (m = v = 0: no step moves it) / current / claimed by another workgroup
  const int col = threadIdx.x;
  const bool mine = col < width;
  const size_t at = (size_t)row * width + col;
  float pp = 0.f, mm = 0.f, vv = 0.f;
  if (mine) {
    pp = p[at];
    mm = m[at];
    vv = v[at];
  }
  while (from < now) {
    const int cnt = min(now - from, 1024);
    __syncthreads();
    if ((int)threadIdx.x < cnt) s_step[threadIdx.x] = hist[(from + 1 + (int)threadIdx.x) & mask];
    __syncthreads();
    for (int k = 0; k < cnt; ++k) adam_zero_grad_step(pp, mm, vv, s_step[k], b1, b2, eps);
    from += cnt;
  }
  if (mine) {
    p[at] = pp;
    m[at] = mm;
    v[at] = vv;
  }
}

// step t of the optimizer on the TOUCHED rows (the others wait for their next reader): whatever a row still owes up to t - 1,
// then the update proper with its gradient, which is zeroed; row_t <- t, TOUCHED cleared.  Workgroup 0 also advances the clock.
__global__ __launch_bounds__(256) void adamw_rows_lazy_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                              float* __restrict__ v, unsigned char* __restrict__ flags,
                                                              int* __restrict__ row_t, float* __restrict__ hist, int mask, int t,
                                                              int rows, int width, float step_size, float b1, float b2, float eps,
                                                              const float* __restrict__ gnorm_sq, float max_norm, float grad_scale,
                                                              int* __restrict__ clock) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // the clock advances with this launch: nothing in it reads `clock` (the step index is the argument t) or hist[t] (a row owes
  // steps up to t - 1 at most); readers are later launches
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hist[t & mask] = step_size;
    *clock = t;
  }
  if (row >= rows) return;
  const unsigned fl = __builtin_amdgcn_readfirstlane((unsigned)flags[row]);
  if (!(fl & 2u)) return;
  float gs = grad_scale;
  if (gnorm_sq) {
    const float norm = sqrtf(*gnorm_sq) * grad_scale;
    const float coef = max_norm / (norm + 1e-6f);
    if (coef < 1.0f) gs *= coef;
  }
  const size_t base = (size_t)row * width;
  const int from = __builtin_amdgcn_readfirstlane(row_t[row]);
  RowState st;
  row_catch_up<4>(p, m, v, base, width, lane, from < 0 ? t - 1 : from, t - 1, hist, mask, b1, b2, eps, st, true);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int i = lane * 4 + c * 256;
    if (i < width) {
      const float4 gg = *reinterpret_cast<const float4*>(g + base + i);
      float* pa = &st.p[c].x;
      const float* ga = &gg.x;
      float* ma = &st.m[c].x;
      float* va = &st.v[c].x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gk = ga[k] * gs;
        ma[k] = ma[k] * b1 + (1.0f - b1) * gk;
        va[k] = va[k] * b2 + (1.0f - b2) * gk * gk;
        pa[k] -= step_size * (ma[k] / (sqrtf(va[k]) + eps));
      }
      *reinterpret_cast<float4*>(g + base + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  row_store<4>(p, m, v, base, width, lane, st);
  if (lane == 0) {
    row_t[row] = t;
    flags[row] = (unsigned char)(fl & ~2u);
  }
}

__global__ void rows_clock_kernel(int* __restrict__ clock, float* __restrict__ hist, int mask, int t, float step_size) {
  hist[t & mask] = step_size;
  *clock = t;
}

// partial[b] = sum of g^2 over the TOUCHED rows among [b * rpb, (b+1) * rpb)
__global__ __launch_bounds__(256) void sqnorm_rows_kernel(const float* __restrict__ g, const unsigned char* __restrict__ flags,
                                                          int rows, int width, int rpb, float* __restrict__ partial) {
  __shared__ float red[4];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r1 = min(rows, (int)(blockIdx.x + 1) * rpb);
  float acc = 0.0f;
  // the flags of this wave's rows (blockIdx.x * rpb + wid + 4 j) 64 at a time, then only the touched ones in ascending order --
  // a live row nothing has written since the last step holds g == 0 and adds nothing to any partial sum
  for (int j0 = 0; blockIdx.x * rpb + wid + 4 * j0 < r1; j0 += 64) {
    const int myrow = blockIdx.x * rpb + wid + 4 * (j0 + lane);
    unsigned long long todo = __ballot(myrow < r1 && (flags[myrow] & 2));
    while (todo) {
      const int j = __builtin_ctzll(todo);
      todo &= todo - 1;
      const float* gr = g + (size_t)(blockIdx.x * rpb + wid + 4 * (j0 + j)) * width;
      for (int i = lane * 4; i < width; i += 256) {
        const float4 x = *reinterpret_cast<const float4*>(gr + i);
        acc += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
      }
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wid] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// flags[ids[i]] = LIVE | TOUCHED for i < n (ids < 0 ignored): which embedding rows have ever / now received a gradient
__global__ __launch_bounds__(256) void mark_rows_kernel(const int* __restrict__ ids, int n, unsigned char* __restrict__ flags, int rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int r = ids[i];
    if (r >= 0 && r < rows) flags[r] = 3;
  }
}

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 x = reinterpret_cast<const float4*>(g)[i];
    acc += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ partial, int np, float* __restrict__ out,
                                                           int accumulate) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) acc += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s = (red[0] + red[1]) + (red[2] + red[3]);
    out[0] = accumulate ? (float)((double)out[0] + s) : (float)s;
  }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    uint2 u;
    u.x = pack2bf(a.x, a.y);
    u.y = pack2bf(a.z, a.w);
    reinterpret_cast<uint2*>(y)[i] = u;
  }
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, size_t n) {
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const uint2 u = reinterpret_cast<const uint2*>(x)[i];
    float4 a;
    a.x = __uint_as_float(u.x << 16);
    a.y = __uint_as_float(u.x & 0xffff0000u);
    a.z = __uint_as_float(u.y << 16);
    a.w = __uint_as_float(u.y & 0xffff0000u);
    reinterpret_cast<float4*>(y)[i] = a;
  }
}

// out[0] = sum_i w[i] * (a[i] - b[i])      (CRF loss: mean over sentences of logZ - gold)
__global__ __launch_bounds__(64) void wdiff_sum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ w, int n, float* __restrict__ out) {
  float acc = 0.0f;
  for (int i = threadIdx.x; i < n; i += 64) acc += w[i] * (a[i] - b[i]);
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[0] = acc;
}

#define SQN_BLOCKS 2048

extern "C" {

int kbner_sqnorm_ws_floats(void) { return SQN_BLOCKS; }

// n % 4 == 0 (the arena pads every tensor to 4 floats); n_shadow % 4 == 0
int kbner_adamw_hf(float* p, float* g, float* m, float* v, bf16_t* shadow, size_t n, size_t n_shadow, float step_size,
                   float lr_wd, float b1, float b2, float eps, const float* gnorm_sq, float max_norm, float grad_scale,
                   int zero_grad, void* stream) {
  KBNER_CHECK_ARG(n % 4 == 0 && n_shadow % 4 == 0 && n_shadow <= n);
  if (n == 0) return 0;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, shadow, n, n_shadow,
                     step_size, lr_wd, b1, b2, eps, gnorm_sq, max_norm, grad_scale, zero_grad);
  KBNER_LAUNCH_RET();
}

// AdamW (weight decay 0) on the flagged rows of an embedding table p/g/m/v f32[rows, width]; width % 4 == 0.  Unflagged rows are
// not touched: exact as long as a row's flag is set (kbner_mark_rows) before its first non-zero gradient is applied.
int kbner_adamw_hf_rows(float* p, float* g, float* m, float* v, unsigned char* flags, int rows, int width, float step_size,
                        float b1, float b2, float eps, const float* gnorm_sq, float max_norm, float grad_scale, int zero_grad,
                        void* stream) {
  KBNER_CHECK_ARG(p != nullptr && g != nullptr && m != nullptr && v != nullptr && flags != nullptr);
  KBNER_CHECK_ARG(rows >= 0 && width > 0 && width % 4 == 0);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(adamw_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, p, g, m, v, flags, rows, width,
                     step_size, b1, b2, eps, gnorm_sq, max_norm, grad_scale, zero_grad);
  KBNER_LAUNCH_RET();
}

// Lazy variant of kbner_adamw_hf_rows (see the kernels): step `t` (1-based, = *clock + 1) on the touched rows, then hist[t & mask]
// <- step_size, *clock <- t.  hist holds mask + 1 floats (a power of two); the caller brings every row up to date
// (kbner_adamw_rows_catchup with ids == nullptr) at least once every `mask` steps.  width <= 1024.
int kbner_adamw_hf_rows_lazy(float* p, float* g, float* m, float* v, unsigned char* flags, int* row_t, int* clock, float* hist,
                             int hist_len, int t, int rows, int width, float step_size, float b1, float b2, float eps,
                             const float* gnorm_sq, float max_norm, float grad_scale, void* stream) {
  KBNER_CHECK_ARG(p != nullptr && g != nullptr && m != nullptr && v != nullptr && flags != nullptr && row_t != nullptr);
  KBNER_CHECK_ARG(clock != nullptr && hist != nullptr && hist_len >= 2 && (hist_len & (hist_len - 1)) == 0 && t >= 1);
  KBNER_CHECK_ARG(rows >= 0 && width > 0 && width % 4 == 0 && width <= 1024);
  if (rows > 0)
    hipLaunchKernelGGL(adamw_rows_lazy_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, p, g, m, v, flags, row_t, hist,
                       hist_len - 1, t, rows, width, step_size, b1, b2, eps, gnorm_sq, max_norm, grad_scale, clock);
  else
    hipLaunchKernelGGL(rows_clock_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, clock, hist, hist_len - 1, t, step_size);
  KBNER_LAUNCH_RET();
}

// rows ids[0..n) (device i32; entries < 0 ignored; nullptr: all `rows` rows) brought to step *clock: what the eager kernel would
// have done to them in every step since row_t.  Before every embedding lookup of a table under kbner_adamw_hf_rows_lazy.
int kbner_adamw_rows_catchup(const int* ids, int n, float* p, float* m, float* v, const unsigned char* flags, int* row_t,
                             const int* clock, const float* hist, int hist_len, int rows, int width, float b1, float b2, float eps,
                             void* stream) {
  KBNER_CHECK_ARG(p != nullptr && m != nullptr && v != nullptr && flags != nullptr && row_t != nullptr && clock != nullptr);
  KBNER_CHECK_ARG(hist != nullptr && hist_len >= 2 && (hist_len & (hist_len - 1)) == 0 && rows > 0 && width > 0 && width % 4 == 0 &&
                  width <= 1024 && n >= 0);
  const int cnt = ids ? n : rows;
  if (cnt == 0) return 0;
  hipLaunchKernelGGL(adamw_rows_catchup_kernel, dim3(cnt), dim3(1024), 0, (hipStream_t)stream, ids, cnt, p, m, v, flags, row_t, clock, hist,
                     hist_len - 1, rows, width, b1, b2, eps);
  KBNER_LAUNCH_RET();
}

// out[0] (+)= sum of g^2 over the flagged rows of g f32[rows, width]; ws holds kbner_sqnorm_ws_floats() floats
int kbner_grad_sqnorm_rows(const float* g, const unsigned char* flags, int rows, int width, float* ws, float* out, int accumulate,
                           void* stream) {
  KBNER_CHECK_ARG(g != nullptr && flags != nullptr && ws != nullptr && out != nullptr && rows > 0 && width > 0 && width % 4 == 0);
  int rpb = (rows + SQN_BLOCKS - 1) / SQN_BLOCKS;
  rpb = (rpb + 3) / 4 * 4;
  const int blocks = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(sqnorm_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, flags, rows, width, rpb, ws);
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, blocks, out, accumulate);
  KBNER_LAUNCH_RET();
}

int kbner_mark_rows(const int* ids, int n, unsigned char* flags, int rows, void* stream) {
  KBNER_CHECK_ARG(ids != nullptr && flags != nullptr && n >= 0 && rows > 0);
  if (n == 0) return 0;
  hipLaunchKernelGGL(mark_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids, n, flags, rows);
  KBNER_LAUNCH_RET();
}

// out[0] (+)= sum g^2 ; ws holds kbner_sqnorm_ws_floats() floats
int kbner_grad_sqnorm(const float* g, size_t n, float* ws, float* out, int accumulate, void* stream) {
  KBNER_CHECK_ARG(n % 4 == 0);
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > SQN_BLOCKS) blocks = SQN_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, ws);
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, (int)blocks, out, accumulate);
  KBNER_LAUNCH_RET();
}

int kbner_f32_to_bf16(const float* x, bf16_t* y, size_t n, void* stream) {
  KBNER_CHECK_ARG(n % 4 == 0);
  if (n == 0) return 0;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n);
  KBNER_LAUNCH_RET();
}

int kbner_bf16_to_f32(const bf16_t* x, float* y, size_t n, void* stream) {
  KBNER_CHECK_ARG(n % 4 == 0);
  if (n == 0) return 0;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n);
  KBNER_LAUNCH_RET();
}

int kbner_wdiff_sum(const float* a, const float* b, const float* w, int n, float* out, void* stream) {
  KBNER_CHECK_ARG(n >= 0);
  hipLaunchKernelGGL(wdiff_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, w, n, out);
  KBNER_LAUNCH_RET();
}

}  // extern "C"
