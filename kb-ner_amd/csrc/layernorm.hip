// LayerNorm(+embedding) kernels for gfx950.  HBM-bound: one 64-lane wavefront owns a token row and
// reduces it with DPP/shuffle butterflies; every global access is a 16-byte vector of 8 bf16
// (or 2 x float4 for fp32 tables).  fp32 statistics, bf16 storage.
//
// Replaces the torch ops reached through transformers' BertEmbeddings / BertSelfOutput /
// BertOutput LayerNorm (called from flair/embeddings.py:3269) and their autograd backward.
//   fwd : y = (h - mean) * rstd * gamma + beta           (eps inside the sqrt, eps = 1e-5)
//   bwd : dh = rstd * (g - mean_H(g) - xhat * mean_H(g * xhat)),  g = dy * gamma
//         dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy ; dbias += sum_rows dh (optional:
//         the bias gradient of the GEMM whose output fed this LayerNorm's input)
// H must be a multiple of 8 and <= 1024 (XLM-R base 768 / large 1024).
#include "common.h"

#define LN_MAXCH 2  // 16-byte chunks per lane: H <= 64 * 8 * 2

template <int NCH>
struct RowF {
  float v[NCH][8];
};

template <int NCH>
static __device__ __forceinline__ void load_row_bf16(const bf16_t* __restrict__ p, int H, int lane, RowF<NCH>& r) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int h0 = (lane + 64 * c) * 8;
    if (h0 < H) {
      const uint4 u = *reinterpret_cast<const uint4*>(p + h0);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r.v[c][2 * j] = __uint_as_float(w[j] << 16);
        r.v[c][2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[c][j] = 0.0f;
    }
  }
}

// raw (unconverted) row image: lets the backward kernel fetch row r+stride while it reduces row r
template <int NCH>
struct RowRaw {
  uint4 v[NCH];
};
template <int NCH>
static __device__ __forceinline__ void load_row_raw(const bf16_t* __restrict__ p, int H, int lane, RowRaw<NCH>& r) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int h0 = (lane + 64 * c) * 8;
    r.v[c] = (h0 < H) ? *reinterpret_cast<const uint4*>(p + h0) : make_uint4(0u, 0u, 0u, 0u);
  }
}
template <int NCH>
static __device__ __forceinline__ void unpack_row(const RowRaw<NCH>& raw, RowF<NCH>& r) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const uint32_t w[4] = {raw.v[c].x, raw.v[c].y, raw.v[c].z, raw.v[c].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r.v[c][2 * j] = __uint_as_float(w[j] << 16);
      r.v[c][2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
    }
  }
}

template <int NCH>
static __device__ __forceinline__ void store_row_bf16(bf16_t* __restrict__ p, int H, int lane, const RowF<NCH>& r) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int h0 = (lane + 64 * c) * 8;
    if (h0 < H) {
      uint4 u;
      u.x = pack2bf(r.v[c][0], r.v[c][1]);
      u.y = pack2bf(r.v[c][2], r.v[c][3]);
      u.z = pack2bf(r.v[c][4], r.v[c][5]);
      u.w = pack2bf(r.v[c][6], r.v[c][7]);
      *reinterpret_cast<uint4*>(p + h0) = u;
    }
  }
}

template <int NCH>
static __device__ __forceinline__ void load_row_f32(const float* __restrict__ p, int H, int lane, RowF<NCH>& r) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int h0 = (lane + 64 * c) * 8;
    if (h0 < H) {
      const float4 a = *reinterpret_cast<const float4*>(p + h0);
      const float4 b = *reinterpret_cast<const float4*>(p + h0 + 4);
      r.v[c][0] = a.x; r.v[c][1] = a.y; r.v[c][2] = a.z; r.v[c][3] = a.w;
      r.v[c][4] = b.x; r.v[c][5] = b.y; r.v[c][6] = b.z; r.v[c][7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[c][j] = 0.0f;
    }
  }
}

// dropout column keys of this lane's columns (common.h: drop_colkey), kept as raw bits in a RowF
template <int NCH>
static __device__ __forceinline__ void load_colkeys(uint32_t seed, int lane, RowF<NCH>& k) {
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) k.v[c][j] = __uint_as_float(drop_colkey(seed, (uint32_t)((lane + 64 * c) * 8 + j)));
}
template <int NCH>
static __device__ __forceinline__ void drop_row(RowF<NCH>& x, const RowF<NCH>& ck, uint32_t seed, uint32_t thresh, float scale,
                                                int row) {
  const uint32_t rk = drop_rowkey(seed, (uint32_t)row);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) x.v[c][j] = drop_keep(rk, __float_as_uint(ck.v[c][j]), thresh) ? x.v[c][j] * scale : 0.0f;
}

// round a row through bf16 (what is stored is what is normalised: fwd/bwd stay consistent)
template <int NCH>
static __device__ __forceinline__ void round_row(RowF<NCH>& r) {
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[c][j] = bf2f(f2bf(r.v[c][j]));
}

template <int NCH>
static __device__ __forceinline__ void row_stats(const RowF<NCH>& x, int H, float eps, float& mean, float& rstd) {
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x.v[c][j];
  mean = wave_sum(s) / (float)H;
  float q = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int h0 = (threadIdx.x % 64 + 64 * c) * 8;
    if (h0 < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = x.v[c][j] - mean;
        q += d * d;
      }
    }
  }
  rstd = rsqrtf(wave_sum(q) / (float)H + eps);
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ h, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o, int M, int H) {
  const int lane = threadIdx.x % 64;
  const int wave = blockIdx.x * 4 + threadIdx.x / 64;
  const int nwave = gridDim.x * 4;
  RowF<NCH> g, b;
  load_row_f32<NCH>(gamma, H, lane, g);
  load_row_f32<NCH>(beta, H, lane, b);
  // row r + stride is requested while row r is reduced (as in the backward kernel): one row in flight per wave left the kernel at
  // 5.0-5.2 TB/s -- 16 waves x 2 KiB per CU is about what the HBM latency needs in flight, with nothing to spare
  RowRaw<NCH> raw;
  if (wave < M) load_row_raw<NCH>(h + (size_t)wave * H, H, lane, raw);
  for (int r = wave; r < M; r += nwave) {
    RowRaw<NCH> nxt = raw;
    if (r + nwave < M) load_row_raw<NCH>(h + (size_t)(r + nwave) * H, H, lane, nxt);
    RowF<NCH> x;
    unpack_row<NCH>(raw, x);
    float mean, rstd;
    row_stats<NCH>(x, H, eps, mean, rstd);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) x.v[c][j] = (x.v[c][j] - mean) * rstd * g.v[c][j] + b.v[c][j];
    store_row_bf16<NCH>(y + (size_t)r * H, H, lane, x);
    if (lane == 0) {
      mean_o[r] = mean;
      rstd_o[r] = rstd;
    }
    raw = nxt;
  }
}

// The same with the input row FOLDED from split-K slabs on the way in (round 6, small micro-batches): h = bf16(dropout(sum_s ws[s] +
// bias) + addend) is what kbner_splitk_finish would have written -- splitk_fold8_pack, the same bits --; it is stored (the backward
// pass reads it) and normalised without the launch and the read-back in between.  M <= a few thousand rows: one row per wave.
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_slabs_kernel(const float* __restrict__ ws, int splits, const float* __restrict__ bias,
                                                           const bf16_t* __restrict__ addend, int ldadd, uint32_t drop_seed,
                                                           uint32_t drop_thresh, bf16_t* __restrict__ h, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, bf16_t* __restrict__ y,
                                                           float* __restrict__ mean_o, float* __restrict__ rstd_o, int M, int H) {
  const int lane = threadIdx.x % 64;
  const int wave = blockIdx.x * 4 + threadIdx.x / 64;
  const int nwave = gridDim.x * 4;
  const size_t slab = (size_t)M * H;
  RowF<NCH> g, b;
  load_row_f32<NCH>(gamma, H, lane, g);
  load_row_f32<NCH>(beta, H, lane, b);
  for (int r = wave; r < M; r += nwave) {
    RowRaw<NCH> raw;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int h0 = (lane + 64 * c) * 8;
      raw.v[c] = make_uint4(0u, 0u, 0u, 0u);
      if (h0 < H) {
        raw.v[c] = splitk_fold8_pack(ws, splits, slab, bias, addend, ldadd, r, h0, H, drop_seed, drop_thresh);
        *reinterpret_cast<uint4*>(h + (size_t)r * H + h0) = raw.v[c];
      }
    }
    RowF<NCH> x;
    unpack_row<NCH>(raw, x);
    float mean, rstd;
    row_stats<NCH>(x, H, eps, mean, rstd);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) x.v[c][j] = (x.v[c][j] - mean) * rstd * g.v[c][j] + b.v[c][j];
    store_row_bf16<NCH>(y + (size_t)r * H, H, lane, x);
    if (lane == 0) {
      mean_o[r] = mean;
      rstd_o[r] = rstd;
    }
  }
}

// word[ids] + pos[pos_ids] + type[0] -> h0 (bf16, saved) -> LayerNorm -> y
template <int NCH>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int* __restrict__ ids, const int* __restrict__ pos_ids,
                                                           const float* __restrict__ word, const float* __restrict__ pos,
                                                           const float* __restrict__ type0, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, bf16_t* __restrict__ h0,
                                                           bf16_t* __restrict__ y, float* __restrict__ mean_o,
                                                           float* __restrict__ rstd_o, int M, int H, uint32_t drop_seed,
                                                           uint32_t drop_thresh) {
  const int lane = threadIdx.x % 64;
  const int wave = blockIdx.x * 4 + threadIdx.x / 64;
  const int nwave = gridDim.x * 4;
  RowF<NCH> g, b, ty, ck;
  if (drop_thresh) load_colkeys<NCH>(drop_seed, lane, ck);
  const float dscale = drop_scale(drop_thresh);
  load_row_f32<NCH>(gamma, H, lane, g);
  load_row_f32<NCH>(beta, H, lane, b);
  load_row_f32<NCH>(type0, H, lane, ty);
  for (int r = wave; r < M; r += nwave) {
    RowF<NCH> x, p;
    load_row_f32<NCH>(word + (size_t)ids[r] * H, H, lane, x);
    load_row_f32<NCH>(pos + (size_t)pos_ids[r] * H, H, lane, p);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) x.v[c][j] = (x.v[c][j] + p.v[c][j]) + ty.v[c][j];
    round_row<NCH>(x);
    store_row_bf16<NCH>(h0 + (size_t)r * H, H, lane, x);
    float mean, rstd;
    row_stats<NCH>(x, H, eps, mean, rstd);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) x.v[c][j] = (x.v[c][j] - mean) * rstd * g.v[c][j] + b.v[c][j];
    if (drop_thresh) drop_row<NCH>(x, ck, drop_seed, drop_thresh, dscale, r);  // BertEmbeddings.dropout
    store_row_bf16<NCH>(y + (size_t)r * H, H, lane, x);
    if (lane == 0) {
      mean_o[r] = mean;
      rstd_o[r] = rstd;
    }
  }
}

// dst[0..H) += r (fp32 atomics).  In the row layout a lane owns 8 CONSECUTIVE elements, so an atomic instruction issued from
// it would touch 64 addresses 32 bytes apart (16 cache lines, 4 lanes each).  The row is therefore transposed through `stage`
// (this wave's 64 * 8 * NCH floats of LDS) first: each atomic instruction then covers 64 consecutive floats = two cache lines,
// which the L2 retires several times faster (the embedding backward went from 1.91 ms to 0.25 ms per step at 65 536 tokens).
template <int NCH>
static __device__ __forceinline__ void flush_row_atomic(float* __restrict__ dst, int H, int lane, const RowF<NCH>& r,
                                                        float* __restrict__ stage) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float4* s4 = reinterpret_cast<float4*>(stage + (lane + 64 * c) * 8);
    s4[0] = make_float4(r.v[c][0], r.v[c][1], r.v[c][2], r.v[c][3]);
    s4[1] = make_float4(r.v[c][4], r.v[c][5], r.v[c][6], r.v[c][7]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int i = 0; i < 8 * NCH; ++i) {
    const int e = i * 64 + lane;
    if (e < H) atomicAdd(dst + e, stage[e]);
  }
  __builtin_amdgcn_wave_barrier();   // the next row's stores must not overtake these reads
}

// Backward.  EMBED: additionally scatter-add dh into the embedding-table gradients (fp32 atomics:
// token ids repeat) and accumulate sum_rows dh into dtype0 through the dbias path.
template <int NCH, bool EMBED, bool SLABS = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ h,
                                                     const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                     const float* __restrict__ gamma, bf16_t* __restrict__ dh,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ dbias, const int* __restrict__ ids,
                                                     const int* __restrict__ pos_ids, float* __restrict__ dword,
                                                     float* __restrict__ dpos, float* __restrict__ ws, int M, int H,
                                                     bf16_t* __restrict__ dhm, uint32_t drop_seed, uint32_t drop_thresh,
                                                     const float* __restrict__ dy_ws = nullptr, int dy_splits = 0,
                                                     const bf16_t* __restrict__ dy_add = nullptr, int dy_ldadd = 0,
                                                     unsigned char* __restrict__ row_flags = nullptr) {
  // SLABS (round 6, small micro-batches; a separate instantiation, the other one is untouched): the incoming gradient row is folded from the split-K slabs of the GEMM that
  // produced it -- bf16(sum_s dy_ws[s] + dy_add), kbner_splitk_finish's bits (splitk_fold8_pack) -- instead of read from `dy`.
  // Dropout replay (drop_thresh != 0): EMBED -> the incoming dy is masked first (y = drop(LN(h0)));
  // otherwise -> the GEMM whose (dropped) output fed this LayerNorm's input gets dhm = mask * dh as its dY, the
  // residual branch keeps the unmasked dh, and the GEMM's bias gradient (dbias) sums the masked rows.
  __shared__ float red[3][4][64 * 8 * NCH];
  const int lane = threadIdx.x % 64;
  const int wid = threadIdx.x / 64;
  const int wave = blockIdx.x * 4 + wid;
  const int nwave = gridDim.x * 4;
  RowF<NCH> g, ag, ab, ah, ck;
  load_row_f32<NCH>(gamma, H, lane, g);
  if (drop_thresh) load_colkeys<NCH>(drop_seed, lane, ck);
  const float dscale = drop_scale(drop_thresh);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) ag.v[c][j] = ab.v[c][j] = ah.v[c][j] = 0.0f;
  const float invH = 1.0f / (float)H;
  RowF<NCH> pacc;  // EMBED: running position-embedding gradient of the current run of equal position ids
  int pcur = -1;
  // one row ahead: the loads of row r + nwave are in flight while row r is reduced and stored
  RowRaw<NCH> xraw, draw;
  float mean_n = 0.0f, rstd_n = 0.0f;
  const size_t dy_slab = (size_t)M * H;
  auto load_dy = [&](int row, RowRaw<NCH>& out) {
    if constexpr (!SLABS) {
      load_row_raw<NCH>(dy + (size_t)row * H, H, lane, out);
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int h0 = (lane + 64 * c) * 8;
        out.v[c] = (h0 < H) ? splitk_fold8_pack(dy_ws, dy_splits, dy_slab, nullptr, dy_add, dy_ldadd, row, h0, H, 0u, 0u)
                            : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  if (wave < M) {
    load_row_raw<NCH>(h + (size_t)wave * H, H, lane, xraw);
    load_dy(wave, draw);
    mean_n = mean_i[wave];
    rstd_n = rstd_i[wave];
  }
  for (int r = wave; r < M; r += nwave) {
    RowF<NCH> x, d;
    unpack_row<NCH>(xraw, x);
    unpack_row<NCH>(draw, d);
    const float mean = mean_n, rstd = rstd_n;
    const int rn = r + nwave;
    if (rn < M) {
      load_row_raw<NCH>(h + (size_t)rn * H, H, lane, xraw);
      load_dy(rn, draw);
      mean_n = mean_i[rn];
      rstd_n = rstd_i[rn];
    }
    if (EMBED && drop_thresh) drop_row<NCH>(d, ck, drop_seed, drop_thresh, dscale, r);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int h0 = (lane + 64 * c) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (h0 < H) ? (x.v[c][j] - mean) * rstd : 0.0f;
        const float gd = d.v[c][j] * g.v[c][j];
        x.v[c][j] = xh;
        ag.v[c][j] += d.v[c][j] * xh;
        ab.v[c][j] += d.v[c][j];
        s1 += gd;
        s2 += gd * xh;
      }
    }
    s1 = wave_sum(s1) * invH;
    s2 = wave_sum(s2) * invH;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        d.v[c][j] = rstd * (d.v[c][j] * g.v[c][j] - s1 - x.v[c][j] * s2);
      }
    if (dh) store_row_bf16<NCH>(dh + (size_t)r * H, H, lane, d);
    if (!EMBED && drop_thresh) {
      drop_row<NCH>(d, ck, drop_seed, drop_thresh, dscale, r);
      store_row_bf16<NCH>(dhm + (size_t)r * H, H, lane, d);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) ah.v[c][j] += d.v[c][j];
    if (EMBED) {
      float* stage = &red[0][wid][0];   // free until the column-sum reduction after the row loop
      flush_row_atomic<NCH>(dword + (size_t)ids[r] * H, H, lane, d, stage);
      // EMBED: the optimizer's row flags (KBNER_ROW_LIVE | KBNER_ROW_TOUCHED, include/kbner.h) of the rows this pass writes
      if (row_flags != nullptr && lane == 0) row_flags[ids[r]] = 3;
      // Position rows repeat B times per step (514 rows shared by every sentence): with the grid a multiple of the sentence
      // length a wave's successive rows r, r + nwave, ... are the SAME position of different sentences, so their
      // gradients are summed in registers and flushed once per run of equal ids instead of once per token
      // (B-fold fewer atomics on B-way contended addresses); any other id pattern just flushes more often.
      const int p = pos_ids[r];
      if (p != pcur) {
        if (pcur >= 0) flush_row_atomic<NCH>(dpos + (size_t)pcur * H, H, lane, pacc, stage);
        pcur = p;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) pacc.v[c][j] = 0.0f;
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) pacc.v[c][j] += d.v[c][j];
    }
  }
  if (EMBED && pcur >= 0) flush_row_atomic<NCH>(dpos + (size_t)pcur * H, H, lane, pacc, &red[0][wid][0]);
  // Column sums: combine the block's 4 waves in LDS, then write the block's partial row to the workspace with plain
  // coalesced stores; ln_colreduce_kernel sums the partial rows.  (Per-block global atomics -- 3 x H per block -- were
  // the bottleneck of this kernel: with enough blocks to hide HBM latency they outnumber the useful traffic.)
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = (lane + 64 * c) * 8 + j;  // < 64*8*NCH
      red[0][wid][col] = ag.v[c][j];
      red[1][wid][col] = ab.v[c][j];
      red[2][wid][col] = ah.v[c][j];
    }
  __syncthreads();
  float* wrow = ws + (size_t)blockIdx.x * 3 * H;
  for (int col = threadIdx.x; col < H; col += 256) {
    wrow[col] = (red[0][0][col] + red[0][1][col]) + (red[0][2][col] + red[0][3][col]);
    wrow[H + col] = (red[1][0][col] + red[1][1][col]) + (red[1][2][col] + red[1][3][col]);
    wrow[2 * H + col] = (red[2][0][col] + red[2][1][col]) + (red[2][2][col] + red[2][3][col]);
  }
}

// out_k[col] += sum_blocks ws[block][k][col], k = 0..2 (dgamma, dbeta, dbias).  grid = (ceil(3H/64), 8): a block owns 64
// consecutive entries of the 3H-wide partial row and one eighth of the partial rows; 4 row-groups of threads x 8-way
// unrolled loads keep ~32 loads in flight per thread-column, the 8 slices meet through one atomic per entry.
__global__ __launch_bounds__(256) void ln_colreduce_kernel(const float* __restrict__ ws, int nblocks, int H, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ dbias) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;  // over 3*H
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  float acc = 0.0f;
  if (i < 3 * H) {
#pragma unroll 8
    for (int b = b0 + ty; b < b1; b += 4) acc += ws[(size_t)b * 3 * H + i];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && i < 3 * H) {
    const int k = i / H, col = i % H;
    float* out = k == 0 ? dgamma : (k == 1 ? dbeta : dbias);
    if (out != nullptr) atomicAdd(out + col, (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]));
  }
}

// The same reduction for up to LN_BATCH_MAX LayerNorms in ONE launch (blockIdx.z picks the item): with a few sentences per step
// every ln_bwd's own reduce launch is a 4.7-us kernel at the launch floor, 49 times per backward pass; the partial rows of a whole
// pass are kept (one workspace per LayerNorm) and reduced together at its end.  The descriptors travel as kernel arguments.
#define LN_BATCH_MAX 64
struct LnPartialItem {
  const float* ws;
  float* dgamma;
  float* dbeta;
  float* dbias;
  long long nblocks;
};
struct LnPartialBatch {
  LnPartialItem it[LN_BATCH_MAX];
};
__global__ __launch_bounds__(256) void ln_colreduce_batched_kernel(const LnPartialBatch batch, int H) {
  __shared__ float red[4][64];
  const LnPartialItem& q = batch.it[blockIdx.z];
  const float* __restrict__ ws = q.ws;
  const int nblocks = (int)q.nblocks;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;  // over 3*H
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  float acc = 0.0f;
  if (i < 3 * H) {
#pragma unroll 8
    for (int b = b0 + ty; b < b1; b += 4) acc += ws[(size_t)b * 3 * H + i];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && i < 3 * H) {
    const int k = i / H, col = i % H;
    float* out = k == 0 ? q.dgamma : (k == 1 ? q.dbeta : q.dbias);
    if (out != nullptr) atomicAdd(out + col, (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]));
  }
}

#define LN_BWD_MAXBLOCKS 1024

static inline int ln_grid(int M) {
  int g = (M + 3) / 4;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return g;
}

extern "C" {

int kbner_ln_bwd_ws_floats(int H) { return LN_BWD_MAXBLOCKS * 3 * H; }

int kbner_ln_fwd(const bf16_t* h, const float* gamma, const float* beta, float eps, bf16_t* y, float* mean, float* rstd,
                 int M, int H, void* stream) {
  KBNER_CHECK_ARG(M >= 0 && H > 0 && H % 8 == 0 && H <= 64 * 8 * LN_MAXCH);
  if (M == 0) return 0;
  if (H <= 512)
    hipLaunchKernelGGL(ln_fwd_kernel<1>, dim3(ln_grid(M)), dim3(256), 0, (hipStream_t)stream, h, gamma, beta, eps, y, mean, rstd, M, H);
  else
    hipLaunchKernelGGL(ln_fwd_kernel<2>, dim3(ln_grid(M)), dim3(256), 0, (hipStream_t)stream, h, gamma, beta, eps, y, mean, rstd, M, H);
  KBNER_LAUNCH_RET();
}

int kbner_embed_ln_fwd(const int* ids, const int* pos_ids, const float* word, const float* pos, const float* type0,
                       const float* gamma, const float* beta, float eps, bf16_t* h0, bf16_t* y, float* mean, float* rstd,
                       int M, int H, uint32_t drop_seed, uint32_t drop_thresh, void* stream) {
  KBNER_CHECK_ARG(M >= 0 && H > 0 && H % 8 == 0 && H <= 64 * 8 * LN_MAXCH);
  if (M == 0) return 0;
  if (H <= 512)
    hipLaunchKernelGGL(embed_ln_fwd_kernel<1>, dim3(ln_grid(M)), dim3(256), 0, (hipStream_t)stream, ids, pos_ids, word, pos,
                       type0, gamma, beta, eps, h0, y, mean, rstd, M, H, drop_seed, drop_thresh);
  else
    hipLaunchKernelGGL(embed_ln_fwd_kernel<2>, dim3(ln_grid(M)), dim3(256), 0, (hipStream_t)stream, ids, pos_ids, word, pos,
                       type0, gamma, beta, eps, h0, y, mean, rstd, M, H, drop_seed, drop_thresh);
  KBNER_LAUNCH_RET();
}

// dh may be NULL (embedding LayerNorm: nothing upstream).  dbias may be NULL.
// ws: kbner_ln_bwd_ws_floats(H) floats of scratch (per-block partial column sums)
// drop_thresh != 0: also write dhm = dropout-mask(drop_seed) * dh (the dY of the GEMM that produced the dropped branch)
int kbner_ln_bwd(const bf16_t* dy, const bf16_t* h, const float* mean, const float* rstd, const float* gamma, bf16_t* dh,
                 float* dgamma, float* dbeta, float* dbias, float* ws, int M, int H, bf16_t* dhm, uint32_t drop_seed,
                 uint32_t drop_thresh, void* stream) {
  KBNER_CHECK_ARG(M >= 0 && H > 0 && H % 8 == 0 && H <= 64 * 8 * LN_MAXCH && ws != nullptr);
  KBNER_CHECK_ARG(drop_thresh == 0 || dhm != nullptr);
  if (M == 0) return 0;
  int grid = ln_grid(M);
  if (grid > LN_BWD_MAXBLOCKS) grid = LN_BWD_MAXBLOCKS;
  if (H <= 512)
    hipLaunchKernelGGL((ln_bwd_kernel<1, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, h, mean, rstd, gamma, dh,
                       dgamma, dbeta, dbias, nullptr, nullptr, nullptr, nullptr, ws, M, H, dhm, drop_seed, drop_thresh);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<2, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, h, mean, rstd, gamma, dh,
                       dgamma, dbeta, dbias, nullptr, nullptr, nullptr, nullptr, ws, M, H, dhm, drop_seed, drop_thresh);
  // (dgamma == NULL: the partial rows stay in ws -- kbner_ln_bwd_blocks(M) of them -- for kbner_ln_colreduce_batched)
  if (dgamma != nullptr)
    hipLaunchKernelGGL(ln_colreduce_kernel, dim3((3 * H + 63) / 64, 8), dim3(256), 0, (hipStream_t)stream, ws, grid, H, dgamma, dbeta,
                       dbias);
  KBNER_LAUNCH_RET();
}

// kbner_ln_fwd / kbner_ln_bwd with the input row folded from split-K slabs (ws f32 [splits][M, H]) instead of read as bf16: what
// kbner_splitk_finish would have written, without its launch.  Forward: h (the folded row, bf16) is stored as well.
int kbner_ln_fwd_slabs(const float* ws, int splits, const float* bias, const bf16_t* addend, int ldadd, uint32_t drop_seed,
                       uint32_t drop_thresh, bf16_t* h, const float* gamma, const float* beta, float eps, bf16_t* y, float* mean,
                       float* rstd, int M, int H, void* stream) {
  KBNER_CHECK_ARG(M >= 0 && H > 0 && H % 8 == 0 && H <= 64 * 8 * LN_MAXCH && ws != nullptr && splits >= 1 && splits <= 16);
  KBNER_CHECK_ARG(h != nullptr && y != nullptr && (addend == nullptr || ldadd % 8 == 0));
  if (M == 0) return 0;
  if (H <= 512)
    hipLaunchKernelGGL(ln_fwd_slabs_kernel<1>, dim3(ln_grid(M)), dim3(256), 0, (hipStream_t)stream, ws, splits, bias, addend, ldadd,
                       drop_seed, drop_thresh, h, gamma, beta, eps, y, mean, rstd, M, H);
  else
    hipLaunchKernelGGL(ln_fwd_slabs_kernel<2>, dim3(ln_grid(M)), dim3(256), 0, (hipStream_t)stream, ws, splits, bias, addend, ldadd,
                       drop_seed, drop_thresh, h, gamma, beta, eps, y, mean, rstd, M, H);
  KBNER_LAUNCH_RET();
}

int kbner_ln_bwd_slabs(const float* dy_ws, int splits, const bf16_t* dy_add, int ldadd, const bf16_t* h, const float* mean,
                       const float* rstd, const float* gamma, bf16_t* dh, float* dgamma, float* dbeta, float* dbias, float* ws, int M,
                       int H, bf16_t* dhm, uint32_t drop_seed, uint32_t drop_thresh, void* stream) {
  KBNER_CHECK_ARG(M >= 0 && H > 0 && H % 8 == 0 && H <= 64 * 8 * LN_MAXCH && ws != nullptr && dy_ws != nullptr);
  KBNER_CHECK_ARG(splits >= 1 && splits <= 16 && (dy_add == nullptr || ldadd % 8 == 0) && (drop_thresh == 0 || dhm != nullptr));
  if (M == 0) return 0;
  int grid = ln_grid(M);
  if (grid > LN_BWD_MAXBLOCKS) grid = LN_BWD_MAXBLOCKS;
  if (H <= 512)
    hipLaunchKernelGGL((ln_bwd_kernel<1, false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)nullptr, h, mean, rstd,
                       gamma, dh, dgamma, dbeta, dbias, nullptr, nullptr, nullptr, nullptr, ws, M, H, dhm, drop_seed, drop_thresh, dy_ws,
                       splits, dy_add, ldadd);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<2, false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)nullptr, h, mean, rstd,
                       gamma, dh, dgamma, dbeta, dbias, nullptr, nullptr, nullptr, nullptr, ws, M, H, dhm, drop_seed, drop_thresh, dy_ws,
                       splits, dy_add, ldadd);
  if (dgamma != nullptr)
    hipLaunchKernelGGL(ln_colreduce_kernel, dim3((3 * H + 63) / 64, 8), dim3(256), 0, (hipStream_t)stream, ws, grid, H, dgamma, dbeta,
                       dbias);
  KBNER_LAUNCH_RET();
}

// how many partial rows (of 3 H floats) kbner_ln_bwd leaves in its workspace for M rows
int kbner_ln_bwd_blocks(int M) {
  int grid = ln_grid(M);
  return grid > LN_BWD_MAXBLOCKS ? LN_BWD_MAXBLOCKS : grid;
}

// items (HOST memory, n <= 64 records of 5 x 64 bits: ws, dgamma, dbeta, dbias -- device pointers, dbias may be 0 -- and the number of
// partial rows): dgamma / dbeta / dbias += the column sums of each item's partial rows, all in one launch.
int kbner_ln_colreduce_batched(const long long* items, int n, int H, void* stream) {
  KBNER_CHECK_ARG(items != nullptr && n >= 0 && n <= LN_BATCH_MAX && H > 0);
  if (n == 0) return 0;
  LnPartialBatch b;
  for (int i = 0; i < n; ++i) {
    b.it[i].ws = reinterpret_cast<const float*>(items[5 * i]);
    b.it[i].dgamma = reinterpret_cast<float*>(items[5 * i + 1]);
    b.it[i].dbeta = reinterpret_cast<float*>(items[5 * i + 2]);
    b.it[i].dbias = reinterpret_cast<float*>(items[5 * i + 3]);
    b.it[i].nblocks = items[5 * i + 4];
    KBNER_CHECK_ARG(b.it[i].ws != nullptr && b.it[i].dgamma != nullptr && b.it[i].dbeta != nullptr && b.it[i].nblocks > 0);
  }
  for (int i = n; i < LN_BATCH_MAX; ++i) b.it[i] = b.it[0];
  hipLaunchKernelGGL(ln_colreduce_batched_kernel, dim3((3 * H + 63) / 64, 8, n), dim3(256), 0, (hipStream_t)stream, b, H);
  KBNER_LAUNCH_RET();
}

static int embed_ln_bwd_impl(const bf16_t* dy, const bf16_t* h0, const float* mean, const float* rstd, const float* gamma,
                             const int* ids, const int* pos_ids, float* dgamma, float* dbeta, float* dword, float* dpos,
                             float* dtype0, float* ws, int M, int H, uint32_t drop_seed, uint32_t drop_thresh, unsigned char* row_flags,
                             void* stream) {
  KBNER_CHECK_ARG(M >= 0 && H > 0 && H % 8 == 0 && H <= 64 * 8 * LN_MAXCH && ws != nullptr);
  if (M == 0) return 0;
  int grid = ln_grid(M);
  if (grid > LN_BWD_MAXBLOCKS) grid = LN_BWD_MAXBLOCKS;
  if (H <= 512)
    hipLaunchKernelGGL((ln_bwd_kernel<1, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, h0, mean, rstd, gamma,
                       (bf16_t*)nullptr, dgamma, dbeta, dtype0, ids, pos_ids, dword, dpos, ws, M, H, (bf16_t*)nullptr, drop_seed,
                       drop_thresh, (const float*)nullptr, 0, (const bf16_t*)nullptr, 0, row_flags);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<2, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, h0, mean, rstd, gamma,
                       (bf16_t*)nullptr, dgamma, dbeta, dtype0, ids, pos_ids, dword, dpos, ws, M, H, (bf16_t*)nullptr, drop_seed,
                       drop_thresh, (const float*)nullptr, 0, (const bf16_t*)nullptr, 0, row_flags);
  // (dgamma == NULL: the partial rows stay in ws for kbner_ln_colreduce_batched, as for kbner_ln_bwd)
  if (dgamma != nullptr)
    hipLaunchKernelGGL(ln_colreduce_kernel, dim3((3 * H + 63) / 64, 8), dim3(256), 0, (hipStream_t)stream, ws, grid, H, dgamma, dbeta,
                       dtype0);
  KBNER_LAUNCH_RET();
}

int kbner_embed_ln_bwd(const bf16_t* dy, const bf16_t* h0, const float* mean, const float* rstd, const float* gamma,
                       const int* ids, const int* pos_ids, float* dgamma, float* dbeta, float* dword, float* dpos,
                       float* dtype0, float* ws, int M, int H, uint32_t drop_seed, uint32_t drop_thresh, void* stream) {
  return embed_ln_bwd_impl(dy, h0, mean, rstd, gamma, ids, pos_ids, dgamma, dbeta, dword, dpos, dtype0, ws, M, H, drop_seed, drop_thresh,
                           nullptr, stream);
}

// the same + the optimizer's embedding-row flags set by the kernel itself (row_flags u8[rows of dword], may be NULL): every row this
// pass adds a gradient to becomes KBNER_ROW_LIVE | KBNER_ROW_TOUCHED -- what a kbner_mark_rows launch on `ids` would do
int kbner_embed_ln_bwd_mark(const bf16_t* dy, const bf16_t* h0, const float* mean, const float* rstd, const float* gamma,
                            const int* ids, const int* pos_ids, float* dgamma, float* dbeta, float* dword, float* dpos,
                            float* dtype0, float* ws, unsigned char* row_flags, int M, int H, uint32_t drop_seed, uint32_t drop_thresh,
                            void* stream) {
  return embed_ln_bwd_impl(dy, h0, mean, rstd, gamma, ids, pos_ids, dgamma, dbeta, dword, dpos, dtype0, ws, M, H, drop_seed, drop_thresh,
                           row_flags, stream);
}

}  // extern "C"
