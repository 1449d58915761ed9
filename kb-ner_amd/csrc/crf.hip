// Linear-chain CRF kernels for gfx950: Viterbi decode, forward-algorithm NLL, and its analytic
// backward (forward-backward marginals).  One 64-lane wavefront per sentence; lane t owns the
// "to" tag t; the [T,T] transition matrix ([to,from], live-code convention of
// flair/models/sequence_tagger_model.py:402-410) is held one row per lane in registers (see "Register-resident scan")
// for the whole scan.  The scans are sequential-latency bound (n' steps of a T x T max-plus /
// log-sum-exp); throughput comes from one launch covering every sentence of the batch.
//
// Reference semantics restated (never copied):
//   _viterbi_decode   sequence_tagger_model.py:1248-1327
//   _forward_alg      sequence_tagger_model.py:1329-1394
//   _score_sentence   sequence_tagger_model.py:2544-2591
// This file is compiled WITHOUT fast-math: Viterbi must reproduce the reference's fp32 adds
// bit-for-bit ((v[f] + trans[t,f]) -> first max over f -> + emit[t]).
#include "common.h"

#define CRF_NEG (-1e12f)
#define CRF_MAXT 64

// Register-resident scan (all three kernels): lane t keeps row t of the transition matrix (and, for the backward
// kernel, column t and its row of the transition-gradient accumulator) in VGPRs, the running score vector lives one
// value per lane and is broadcast with v_readlane (an SGPR operand of the add), so a step is ~5 VALU ops per (to,from)
// pair with no LDS round trip and no barrier; the next step's emission / alpha rows are prefetched from HBM while the
// current step computes.  TT = 32 or 64 is the compile-time padded tag count (T = 29 in every KB-NER dictionary);
// padding entries hold -inf transitions so they never win a max and add exp(-inf) = 0 to a sum.
static __device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ------------------------------------------------------------------------------------------
// Viterbi
// ------------------------------------------------------------------------------------------
template <int TT>
__global__ __launch_bounds__(64) void crf_viterbi_kernel(const float* __restrict__ emit, const float* __restrict__ trans,
                                                         const int* __restrict__ lens, int n, int T, int start, int stop,
                                                         int* __restrict__ tags, float* __restrict__ conf,
                                                         int* __restrict__ popped) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* svv = reinterpret_cast<float*>(smem);                          // [n][T] Viterbi scores v'_i (for the confidences)
  unsigned char* sbp = smem + (size_t)n * T * sizeof(float);              // [n][T] backpointers
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int L = lens[b];
  const bool live = t < T;
  float row[TT];
#pragma unroll
  for (int f = 0; f < TT; ++f) row[f] = (live && f < T) ? trans[t * T + f] : -INFINITY;
  const float tstop = live ? trans[stop * T + t] : 0.0f;
  int* tg = tags + (size_t)b * n;
  float* cf = conf + (size_t)b * n;
  for (int i = L + t; i < n; i += 64) {
    tg[i] = -1;
    cf[i] = 0.0f;
  }
  const float* e = emit + (size_t)b * n * T;
  float vcur = (t == start) ? 0.0f : CRF_NEG;
  float enext = (live && L > 0) ? e[t] : 0.0f;
  for (int i = 0; i < L; ++i) {
    const float et = enext;
    if (live && i + 1 < L) enext = e[(size_t)(i + 1) * T + t];
    float best = lane_bcast(vcur, 0) + row[0];
    int arg = 0;
#pragma unroll
    for (int f = 1; f < TT; ++f) {
      const float c = lane_bcast(vcur, f) + row[f];
      if (c > best) {  // strict: first maximal index wins, as torch.max(dim) on CPU
        best = c;
        arg = f;
      }
    }
    const float vnew = best + et;
    if (live) {
      sbp[i * T + t] = (unsigned char)arg;
      svv[i * T + t] = vnew;
    }
    vcur = live ? vnew : CRF_NEG;
  }
  // terminal (:1279-1287): + trans[STOP,:], then STOP and START entries forced to -1e12
  float term = -INFINITY;
  if (live) {
    term = vcur + tstop;
    if (t == stop || t == start) term = CRF_NEG;
  }
  float bv = term;
  int bi = live ? t : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  __syncthreads();  // scores / backpointers written by all lanes
  // confidence of token i = max(softmax(v'_i)) = 1 / sum_t exp(v'_i[t] - max_t v'_i[t])   (:1295-1300); off the scan's
  // critical path: one lane per token, after the recurrence
  for (int i = t; i < L; i += 64) {
    const float* v = svv + i * T;
    float m = v[0];
    for (int k = 1; k < T; ++k) m = fmaxf(m, v[k]);
    float sum = 0.0f;
    for (int k = 0; k < T; ++k) sum += expf(v[k] - m);
    cf[i] = 1.0f / sum;
  }
  if (t == 0) {
    int best = bi;
    for (int i = L - 1; i >= 0; --i) {
      tg[i] = best;
      best = sbp[i * T + best];
    }
    if (popped) popped[b] = (L > 0) ? best : start;  // the reference asserts this == START (:1302-1303)
  }
}

// ------------------------------------------------------------------------------------------
// NLL forward: logZ (forward algorithm), gold path score, alpha saved for backward
// ------------------------------------------------------------------------------------------
template <int TT>
__global__ __launch_bounds__(64) void crf_nll_fwd_kernel(const float* __restrict__ emit, const float* __restrict__ trans,
                                                         const int* __restrict__ tags, const int* __restrict__ lens, int n, int T,
                                                         int start, int stop, float* __restrict__ logz, float* __restrict__ gold,
                                                         float* __restrict__ alpha) {
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int L = lens[b];
  const bool live = t < T;
  float row[TT];
#pragma unroll
  for (int f = 0; f < TT; ++f) row[f] = (live && f < T) ? trans[t * T + f] : -INFINITY;
  const float tstop = live ? trans[stop * T + t] : 0.0f;
  const float* e = emit + (size_t)b * n * T;
  float* al = alpha + (size_t)b * (n + 1) * T;
  float acur = (t == start) ? 0.0f : CRF_NEG;
  if (live) al[t] = acur;
  float enext = (live && L > 0) ? e[t] : 0.0f;
  for (int i = 0; i < L; ++i) {
    const float et = enext;
    if (live && i + 1 < L) enext = e[(size_t)(i + 1) * T + t];
    // tag_var[t,f] = (emit[t] + trans[t,f]) + alpha[f]   (:1361-1367, that association)
    float x[TT];
#pragma unroll
    for (int f = 0; f < TT; ++f) x[f] = (et + row[f]) + lane_bcast(acur, f);
    float m = x[0];
#pragma unroll
    for (int f = 1; f < TT; ++f) m = fmaxf(m, x[f]);
    float s = 0.0f;
#pragma unroll
    for (int f = 0; f < TT; ++f) s += __expf(x[f] - m);
    const float anew = m + logf(s);
    if (live) al[(size_t)(i + 1) * T + t] = anew;
    acur = live ? anew : CRF_NEG;
  }
  // terminal: lse(alpha_L + trans[STOP,:])   (:1383-1393)
  const float term = live ? acur + tstop : -INFINITY;
  const float m = wave_max(term);
  const float s = wave_sum(live ? __expf(term - m) : 0.0f);
  // gold path score (:2544-2591) on compacted rows: mask[k] = k < L
  const int* tg = tags + (size_t)b * n;
  float g = 0.0f;
  for (int k = t; k < L; k += 64) {
    const int tk = tg[k];
    const int prev = (k == 0) ? start : tg[k - 1];
    g += e[(size_t)k * T + tk] + trans[tk * T + prev];
  }
  g = wave_sum(g);
  if (t == 0) {
    const int last = (L > 0) ? tg[L - 1] : start;
    logz[b] = m + logf(s);
    gold[b] = g + trans[stop * T + last];
  }
}

// ------------------------------------------------------------------------------------------
// NLL backward: d/d emit and d/d trans of sum_b dloss[b] * (logZ_b - gold_b)
// ------------------------------------------------------------------------------------------
// POSTERIOR = true reuses the same beta scan to emit the token marginals p_i(t) = softmax_t(alpha_i[t] + beta_i[t]) instead of
// gradients (SequenceTagger._obtain_labels' predict_posterior branch, sequence_tagger_model.py:1182-1192, which adds
// _forward_alg(distill_mode=True) and _backward_alg :1396-1470): demit receives the marginals, tags / dloss / dtrans are unused.
template <int TT, bool POSTERIOR>
__global__ __launch_bounds__(64) void crf_nll_bwd_kernel(const float* __restrict__ emit, const float* __restrict__ trans,
                                                         const int* __restrict__ tags, const int* __restrict__ lens,
                                                         const float* __restrict__ alpha, const float* __restrict__ logz,
                                                         const float* __restrict__ dloss, int n, int T, int start, int stop,
                                                         float* __restrict__ demit, float* __restrict__ dtrans) {
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int L = lens[b];
  const bool live = t < T;
  const float w = POSTERIOR ? 1.0f : dloss[b];
  const float lz = logz[b];
  float row[TT], col[TT], dacc[TT];  // trans[t,:], trans[:,t], d trans[t,:]
#pragma unroll
  for (int f = 0; f < TT; ++f) {
    row[f] = (live && f < T) ? trans[t * T + f] : -INFINITY;
    col[f] = (live && f < T) ? trans[f * T + t] : -INFINITY;
    dacc[f] = 0.0f;
  }
  const float* e = emit + (size_t)b * n * T;
  const float* al = alpha + (size_t)b * (n + 1) * T;
  float* de = demit + (size_t)b * n * T;
  const int* tg = POSTERIOR ? nullptr : tags + (size_t)b * n;
  for (int i = L * T + t; i < n * T; i += 64) de[i] = 0.0f;
  // beta_L = trans[STOP,:]; terminal marginal d trans[STOP, t] (kept by lane t, added at the end)
  float beta = live ? trans[stop * T + t] : CRF_NEG;
  float dstop = live ? w * __expf(al[(size_t)L * T + t] + beta - lz) : 0.0f;
  float anext = (live && L > 0) ? al[(size_t)(L - 1) * T + t] : CRF_NEG;
  float enext = (live && L > 0) ? e[(size_t)(L - 1) * T + t] : 0.0f;
  for (int i = L - 1; i >= 0; --i) {
    const float ai = live ? anext : CRF_NEG;
    const float ei = enext;
    if (live && i > 0) {
      anext = al[(size_t)(i - 1) * T + t];
      enext = e[(size_t)(i - 1) * T + t];
    }
    // role "to" = t: pairwise marginals p[t,f] = exp(emit[t] + beta_{i+1}[t] - logZ + trans[t,f] + alpha_i[f])
    const float base = ei + beta - lz;
    float rs = 0.0f;
#pragma unroll
    for (int f = 0; f < TT; ++f) {
      const float p = __expf(base + row[f] + lane_bcast(ai, f));
      rs += p;
      if (!POSTERIOR) dacc[f] += w * p;
    }
    if (live) de[(size_t)i * T + t] = POSTERIOR ? rs : w * rs - ((tg[i] == t) ? w : 0.0f);
    // role "from" = t: beta_i[t] = lse_to(emit[to] + trans[to,t] + beta_{i+1}[to])
    float x[TT];
#pragma unroll
    for (int u = 0; u < TT; ++u) x[u] = (lane_bcast(ei, u) + col[u]) + lane_bcast(beta, u);
    float m = x[0];
#pragma unroll
    for (int u = 1; u < TT; ++u) m = fmaxf(m, x[u]);
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < TT; ++u) s += __expf(x[u] - m);
    beta = live ? m + logf(s) : CRF_NEG;
  }
  if (POSTERIOR) return;
  // d trans: lane t owns row t (pair marginals) ; row STOP additionally gets the terminal marginals; gold path: -w per used
  // transition (lane 0 walks the tags; contention-free through the same atomics)
  if (live) {
#pragma unroll
    for (int f = 0; f < TT; ++f)
      if (f < T && dacc[f] != 0.0f) atomicAdd(dtrans + t * T + f, dacc[f]);
    if (dstop != 0.0f) atomicAdd(dtrans + stop * T + t, dstop);
  }
  if (t == 0) {
    int prev = start;
    for (int k = 0; k < L; ++k) {
      const int tk = tg[k];
      atomicAdd(dtrans + tk * T + prev, -w);
      prev = tk;
    }
    atomicAdd(dtrans + stop * T + prev, -w);
  }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

size_t kbner_crf_viterbi_lds_bytes(int n, int T) { return (size_t)n * T * 5 + 16; }  // fp32 scores + u8 backpointers

int kbner_crf_viterbi(const float* emit, const float* trans, const int* lens, int B, int n, int T, int start, int stop,
                      int* tags, float* conf, int* popped, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= CRF_MAXT && T <= 255);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  const size_t lds = kbner_crf_viterbi_lds_bytes(n, T);
  KBNER_CHECK_ARG(lds <= 160 * 1024);
  const void* fn = T <= 32 ? reinterpret_cast<const void*>(crf_viterbi_kernel<32>) : reinterpret_cast<const void*>(crf_viterbi_kernel<64>);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
  }
  if (T <= 32)
    hipLaunchKernelGGL(crf_viterbi_kernel<32>, dim3(B), dim3(64), lds, (hipStream_t)stream, emit, trans, lens, n, T, start, stop,
                       tags, conf, popped);
  else
    hipLaunchKernelGGL(crf_viterbi_kernel<64>, dim3(B), dim3(64), lds, (hipStream_t)stream, emit, trans, lens, n, T, start, stop,
                       tags, conf, popped);
  KBNER_LAUNCH_RET();
}

int kbner_crf_nll_fwd(const float* emit, const float* trans, const int* tags, const int* lens, int B, int n, int T, int start,
                      int stop, float* logz, float* gold, float* alpha, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= CRF_MAXT);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  if (T <= 32)
    hipLaunchKernelGGL(crf_nll_fwd_kernel<32>, dim3(B), dim3(64), 0, (hipStream_t)stream, emit, trans, tags, lens, n, T, start,
                       stop, logz, gold, alpha);
  else
    hipLaunchKernelGGL(crf_nll_fwd_kernel<64>, dim3(B), dim3(64), 0, (hipStream_t)stream, emit, trans, tags, lens, n, T, start,
                       stop, logz, gold, alpha);
  KBNER_LAUNCH_RET();
}

int kbner_crf_nll_bwd(const float* emit, const float* trans, const int* tags, const int* lens, const float* alpha,
                      const float* logz, const float* dloss, int B, int n, int T, int start, int stop, float* demit,
                      float* dtrans, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= CRF_MAXT);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  if (T <= 32)
    hipLaunchKernelGGL((crf_nll_bwd_kernel<32, false>), dim3(B), dim3(64), 0, (hipStream_t)stream, emit, trans, tags, lens, alpha,
                       logz, dloss, n, T, start, stop, demit, dtrans);
  else
    hipLaunchKernelGGL((crf_nll_bwd_kernel<64, false>), dim3(B), dim3(64), 0, (hipStream_t)stream, emit, trans, tags, lens, alpha,
                       logz, dloss, n, T, start, stop, demit, dtrans);
  KBNER_LAUNCH_RET();
}

// token marginals from the alpha / logZ that kbner_crf_nll_fwd saved: marg f32[B,n,T] (rows >= lens[b] are zero)
int kbner_crf_posterior(const float* emit, const float* trans, const int* lens, const float* alpha, const float* logz, int B,
                        int n, int T, int start, int stop, float* marg, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= CRF_MAXT);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  if (T <= 32)
    hipLaunchKernelGGL((crf_nll_bwd_kernel<32, true>), dim3(B), dim3(64), 0, (hipStream_t)stream, emit, trans, (const int*)nullptr,
                       lens, alpha, logz, (const float*)nullptr, n, T, start, stop, marg, (float*)nullptr);
  else
    hipLaunchKernelGGL((crf_nll_bwd_kernel<64, true>), dim3(B), dim3(64), 0, (hipStream_t)stream, emit, trans, (const int*)nullptr,
                       lens, alpha, logz, (const float*)nullptr, n, T, start, stop, marg, (float*)nullptr);
  KBNER_LAUNCH_RET();
}

}  // extern "C"
