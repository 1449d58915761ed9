// Linear-chain CRF kernels for gfx950: Viterbi decode, forward-algorithm NLL, and its analytic
// backward (forward-backward marginals).  One 64-lane wavefront per sentence; lane t owns the
// "to" tag t; the [T,T] transition matrix ([to,from], live-code convention of
// flair/models/sequence_tagger_model.py:402-410) and the running score vector stay LDS-resident
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

// ------------------------------------------------------------------------------------------
// Viterbi
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void crf_viterbi_kernel(const float* __restrict__ emit, const float* __restrict__ trans,
                                                         const int* __restrict__ lens, int n, int T, int start, int stop,
                                                         int* __restrict__ tags, float* __restrict__ conf,
                                                         int* __restrict__ popped) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int TP = T | 1;  // odd row stride: conflict-free row reads across lanes
  float* sT = reinterpret_cast<float*>(smem);            // [T][TP]
  float* sv = sT + T * TP;                               // [64]
  unsigned char* sbp = reinterpret_cast<unsigned char*>(sv + 64);  // [n][T]
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int L = lens[b];
  for (int i = t; i < T * T; i += 64) sT[(i / T) * TP + (i % T)] = trans[i];
  if (t < 64) sv[t] = (t == start) ? 0.0f : CRF_NEG;
  int* tg = tags + (size_t)b * n;
  float* cf = conf + (size_t)b * n;
  for (int i = L + t; i < n; i += 64) {
    tg[i] = -1;
    cf[i] = 0.0f;
  }
  __syncthreads();
  const float* e = emit + (size_t)b * n * T;
  float vcur = (t < T) ? sv[t] : CRF_NEG;
  for (int i = 0; i < L; ++i) {
    float vnew = -INFINITY;
    if (t < T) {
      const float* row = sT + t * TP;
      float best = sv[0] + row[0];
      int arg = 0;
      for (int f = 1; f < T; ++f) {
        const float c = sv[f] + row[f];
        if (c > best) {  // strict: first maximal index wins, as torch.max(dim) on CPU
          best = c;
          arg = f;
        }
      }
      vnew = best + e[(size_t)i * T + t];
      sbp[i * T + t] = (unsigned char)arg;
    }
    // confidence = max(softmax(v')) = 1 / sum(exp(v' - max))   (:1295-1300)
    const float m = wave_max(vnew);
    const float ex = (t < T) ? expf(vnew - m) : 0.0f;
    const float s = wave_sum(ex);
    if (t == 0) cf[i] = 1.0f / s;
    __syncthreads();
    if (t < T) sv[t] = vnew;
    vcur = vnew;
    __syncthreads();
  }
  // terminal (:1279-1287): + trans[STOP,:], then STOP and START entries forced to -1e12
  float term = -INFINITY;
  if (t < T) {
    term = vcur + sT[stop * TP + t];
    if (t == stop || t == start) term = CRF_NEG;
  }
  float bv = term;
  int bi = (t < T) ? t : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
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
__global__ __launch_bounds__(64) void crf_nll_fwd_kernel(const float* __restrict__ emit, const float* __restrict__ trans,
                                                         const int* __restrict__ tags, const int* __restrict__ lens, int n, int T,
                                                         int start, int stop, float* __restrict__ logz, float* __restrict__ gold,
                                                         float* __restrict__ alpha) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int TP = T | 1;
  float* sT = reinterpret_cast<float*>(smem);  // [T][TP]
  float* sa = sT + T * TP;                     // [64]
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int L = lens[b];
  for (int i = t; i < T * T; i += 64) sT[(i / T) * TP + (i % T)] = trans[i];
  sa[t] = (t == start) ? 0.0f : CRF_NEG;
  __syncthreads();
  const float* e = emit + (size_t)b * n * T;
  float* al = alpha + (size_t)b * (n + 1) * T;
  if (t < T) al[t] = sa[t];
  float acur = (t < T) ? sa[t] : CRF_NEG;
  for (int i = 0; i < L; ++i) {
    float anew = CRF_NEG;
    if (t < T) {
      const float* row = sT + t * TP;
      const float et = e[(size_t)i * T + t];
      // tag_var[t,f] = (emit[t] + trans[t,f]) + alpha[f]   (:1361-1367, that association)
      float m = (et + row[0]) + sa[0];
      for (int f = 1; f < T; ++f) m = fmaxf(m, (et + row[f]) + sa[f]);
      float s = 0.0f;
      for (int f = 0; f < T; ++f) s += expf(((et + row[f]) + sa[f]) - m);
      anew = m + logf(s);
      al[(size_t)(i + 1) * T + t] = anew;
    }
    __syncthreads();
    if (t < T) sa[t] = anew;
    acur = anew;
    __syncthreads();
  }
  // terminal: lse(alpha_L + trans[STOP,:])   (:1383-1393)
  const float term = (t < T) ? acur + sT[stop * TP + t] : -INFINITY;
  const float m = wave_max(term);
  const float s = wave_sum((t < T) ? expf(term - m) : 0.0f);
  // gold path score (:2544-2591) on compacted rows: mask[k] = k < L
  const int* tg = tags + (size_t)b * n;
  float g = 0.0f;
  for (int k = t; k < L; k += 64) {
    const int tk = tg[k];
    const int prev = (k == 0) ? start : tg[k - 1];
    g += e[(size_t)k * T + tk] + sT[tk * TP + prev];
  }
  g = wave_sum(g);
  if (t == 0) {
    const int last = (L > 0) ? tg[L - 1] : start;
    logz[b] = m + logf(s);
    gold[b] = g + sT[stop * TP + last];
  }
}

// ------------------------------------------------------------------------------------------
// NLL backward: d/d emit and d/d trans of sum_b dloss[b] * (logZ_b - gold_b)
// ------------------------------------------------------------------------------------------
// POSTERIOR = true reuses the same beta scan to emit the token marginals p_i(t) = softmax_t(alpha_i[t] + beta_i[t]) instead of
// gradients (SequenceTagger._obtain_labels' predict_posterior branch, sequence_tagger_model.py:1182-1192, which adds
// _forward_alg(distill_mode=True) and _backward_alg :1396-1470): demit receives the marginals, tags / dloss / dtrans are unused.
template <bool POSTERIOR>
__global__ __launch_bounds__(64) void crf_nll_bwd_kernel(const float* __restrict__ emit, const float* __restrict__ trans,
                                                         const int* __restrict__ tags, const int* __restrict__ lens,
                                                         const float* __restrict__ alpha, const float* __restrict__ logz,
                                                         const float* __restrict__ dloss, int n, int T, int start, int stop,
                                                         float* __restrict__ demit, float* __restrict__ dtrans) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int TP = T | 1;
  float* sT = reinterpret_cast<float*>(smem);  // [T][TP]
  float* sD = sT + T * TP;                     // [T][TP] d trans accumulator (lane t owns row t)
  float* sa = sD + T * TP;                     // [64] alpha_i
  float* sb = sa + 64;                         // [64] beta_{i+1}
  float* se = sb + 64;                         // [64] emit_i
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int L = lens[b];
  const float w = POSTERIOR ? 1.0f : dloss[b];
  const float lz = logz[b];
  for (int i = t; i < T * T; i += 64) sT[(i / T) * TP + (i % T)] = trans[i];
  for (int i = t; i < T * TP; i += 64) sD[i] = 0.0f;
  __syncthreads();
  const float* e = emit + (size_t)b * n * T;
  const float* al = alpha + (size_t)b * (n + 1) * T;
  float* de = demit + (size_t)b * n * T;
  const int* tg = POSTERIOR ? nullptr : tags + (size_t)b * n;
  for (int i = L * T + t; i < n * T; i += 64) de[i] = 0.0f;
  // beta_L = trans[STOP,:]; terminal marginal into d trans[STOP,:]
  if (t < T) {
    sb[t] = sT[stop * TP + t];
    sD[stop * TP + t] += w * expf(al[(size_t)L * T + t] + sT[stop * TP + t] - lz);
  }
  __syncthreads();
  for (int i = L - 1; i >= 0; --i) {
    if (t < T) {
      sa[t] = al[(size_t)i * T + t];
      se[t] = e[(size_t)i * T + t];
    }
    __syncthreads();
    float nb = 0.0f;
    if (t < T) {
      // role "to" = t: pairwise marginals p[t,f]
      const float* row = sT + t * TP;
      float* drow = sD + t * TP;
      const float base = se[t] + sb[t] - lz;
      float rs = 0.0f;
      for (int f = 0; f < T; ++f) {
        const float p = expf(base + row[f] + sa[f]);
        rs += p;
        if (!POSTERIOR) drow[f] += w * p;
      }
      de[(size_t)i * T + t] = POSTERIOR ? rs : w * rs - ((tg[i] == t) ? w : 0.0f);
      // role "from" = t: beta_i[t] = lse_to(emit[to] + trans[to,t] + beta_{i+1}[to])
      float m = -INFINITY;
      for (int u = 0; u < T; ++u) m = fmaxf(m, se[u] + sT[u * TP + t] + sb[u]);
      float s = 0.0f;
      for (int u = 0; u < T; ++u) s += expf(se[u] + sT[u * TP + t] + sb[u] - m);
      nb = m + logf(s);
    }
    __syncthreads();
    if (t < T) sb[t] = nb;
    __syncthreads();
  }
  if (POSTERIOR) return;
  // gold path: -w on each used transition
  if (t == 0) {
    int prev = start;
    for (int k = 0; k < L; ++k) {
      const int tk = tg[k];
      sD[tk * TP + prev] -= w;
      prev = tk;
    }
    sD[stop * TP + prev] -= w;
  }
  __syncthreads();
  for (int i = t; i < T * T; i += 64) {
    const float v = sD[(i / T) * TP + (i % T)];
    if (v != 0.0f) atomicAdd(dtrans + i, v);
  }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

size_t kbner_crf_viterbi_lds_bytes(int n, int T) { return (size_t)(T * (T | 1) + 64) * 4 + (size_t)n * T; }

int kbner_crf_viterbi(const float* emit, const float* trans, const int* lens, int B, int n, int T, int start, int stop,
                      int* tags, float* conf, int* popped, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= CRF_MAXT && T <= 255);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  const size_t lds = kbner_crf_viterbi_lds_bytes(n, T);
  KBNER_CHECK_ARG(lds <= 160 * 1024);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(crf_viterbi_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
  }
  hipLaunchKernelGGL(crf_viterbi_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, emit, trans, lens, n, T, start, stop,
                     tags, conf, popped);
  KBNER_LAUNCH_RET();
}

int kbner_crf_nll_fwd(const float* emit, const float* trans, const int* tags, const int* lens, int B, int n, int T, int start,
                      int stop, float* logz, float* gold, float* alpha, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= CRF_MAXT);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  const size_t lds = (size_t)(T * (T | 1) + 64) * 4;
  hipLaunchKernelGGL(crf_nll_fwd_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, emit, trans, tags, lens, n, T, start,
                     stop, logz, gold, alpha);
  KBNER_LAUNCH_RET();
}

int kbner_crf_nll_bwd(const float* emit, const float* trans, const int* tags, const int* lens, const float* alpha,
                      const float* logz, const float* dloss, int B, int n, int T, int start, int stop, float* demit,
                      float* dtrans, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= CRF_MAXT);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  const size_t lds = (size_t)(2 * T * (T | 1) + 3 * 64) * 4;
  hipLaunchKernelGGL(crf_nll_bwd_kernel<false>, dim3(B), dim3(64), lds, (hipStream_t)stream, emit, trans, tags, lens, alpha, logz,
                     dloss, n, T, start, stop, demit, dtrans);
  KBNER_LAUNCH_RET();
}

// token marginals from the alpha / logZ that kbner_crf_nll_fwd saved: marg f32[B,n,T] (rows >= lens[b] are zero)
int kbner_crf_posterior(const float* emit, const float* trans, const int* lens, const float* alpha, const float* logz, int B,
                        int n, int T, int start, int stop, float* marg, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= CRF_MAXT);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  const size_t lds = (size_t)(2 * T * (T | 1) + 3 * 64) * 4;
  hipLaunchKernelGGL(crf_nll_bwd_kernel<true>, dim3(B), dim3(64), lds, (hipStream_t)stream, emit, trans, (const int*)nullptr, lens,
                     alpha, logz, (const float*)nullptr, n, T, start, stop, marg, (float*)nullptr);
  KBNER_LAUNCH_RET();
}

}  // extern "C"
