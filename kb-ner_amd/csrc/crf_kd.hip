// Multi-view ("cooperative learning") posterior distillation loss of the CRF tagger, forward AND backward in one kernel.
//
// Reference semantics restated (never copied): FastSequenceTagger._calculate_multi_view_loss, `distill_posterior` branch
// (flair/models/sequence_tagger_model.py:2080-2093) as driven by ModelFinetuner.train (flair/trainers/finetune_trainer.py:909-966):
// the sentence WITH its retrieved context is the teacher view (its emissions at the real tokens, detached), the bare sentence
// (`sentence.orig_sent`) the student view, and
//     loss = sum_b sum_i T^2 * KL( softmax(g^t_{b,i} / T) || softmax(g^s_{b,i} / T) ) / B        (:2384-2398, use_crf => / B)
// with g = forward_var + backward_var of `_forward_alg(distill_mode=True)` (:1329-1380, alpha INCLUDING token i's emission) and
// `_backward_alg` (:1396-1470, beta EXCLUDING it; beta_{L-1} = trans[STOP,:]), rows at or past the sentence length masked.
// The reference differentiates this through autograd; here the chain rule through both log-sum-exp recursions is explicit:
//     G_i = dL/dg_i = w_b * T * (q_i - p_i)                                  (q, p: student / teacher tempered marginals)
//     alpha: Abar_i = G_i + sum_t' Abar_{i+1}[t'] W_{i+1}[t',.],  W_i[t,f] = exp(alpha_{i-1}[f] + trans[t,f] - (alpha_i[t] - e_i[t]))
//            d e_i += Abar_i ;  d trans[t,f] += Abar_i[t] W_i[t,f] ;  d trans[t,START] += Abar_0[t]
//     beta : Bbar_j = G_j + S_{j-1},  S_j[u] = sum_t Bbar_j[t] V_j[t,u],  V_j[t,u] = exp(e_{j+1}[u] + beta_{j+1}[u] + trans[u,t] - beta_j[t])
//            d e_{j+1} += S_j ;  d trans[u,t] += Bbar_j[t] V_j[t,u] ;  d trans[STOP,t] += Bbar_{L-1}[t]
//
// One 64-lane wavefront per sentence, lane = tag (T <= 32), exactly like csrc/crf.hip: row t and column t of the transitions
// and of their gradient live in VGPRs, score vectors one value per lane, broadcast by v_readlane; four sequential scans of L
// steps each (alpha, beta + loss, alpha-adjoint, beta-adjoint) over a per-sentence workspace of 4 * n * T floats.
#include "common.h"

#define KD_NEG (-1e12f)
#define KD_TT 32

static __device__ __forceinline__ float kd_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// log-sum-exp over x[0..KD_TT)
static __device__ __forceinline__ float kd_lse(const float (&x)[KD_TT]) {
  float m = x[0];
#pragma unroll
  for (int f = 1; f < KD_TT; ++f) m = fmaxf(m, x[f]);
  float s = 0.0f;
#pragma unroll
  for (int f = 0; f < KD_TT; ++f) s += __expf(x[f] - m);
  return m + logf(s);
}

__global__ __launch_bounds__(64) void crf_posterior_kl_kernel(const float* __restrict__ emit_s, const float* __restrict__ emit_t,
                                                              const float* __restrict__ trans, const int* __restrict__ lens,
                                                              const float* __restrict__ wgt, float tau, int n, int T, int start,
                                                              int stop, float* __restrict__ loss, float* __restrict__ demit,
                                                              float* __restrict__ dtrans, float* __restrict__ ws) {
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int L = lens[b];
  const bool live = t < T;
  const float w = wgt[b];
  const float inv_tau = 1.0f / tau;
  float row[KD_TT], col[KD_TT];   // trans[t,:], trans[:,t]
#pragma unroll
  for (int f = 0; f < KD_TT; ++f) {
    row[f] = (live && f < T) ? trans[t * T + f] : -INFINITY;
    col[f] = (live && f < T) ? trans[f * T + t] : -INFINITY;
  }
  const float* es = emit_s + (size_t)b * n * T;
  const float* et = emit_t + (size_t)b * n * T;
  float* de = demit + (size_t)b * n * T;
  float* As = ws + (size_t)b * 4 * n * T;   // alpha of the student view
  float* Bs = As + (size_t)n * T;            // beta of the student view
  float* G = Bs + (size_t)n * T;             // dL/dg
  float* At = G + (size_t)n * T;             // alpha of the teacher view
  for (int i = L * T + t; i < n * T; i += 64) de[i] = 0.0f;
  if (L <= 0) {
    if (t == 0) loss[b] = 0.0f;
    return;
  }

  // ---- scan 1: alpha of both views (emission of token i included)
  {
    float as = (t == start) ? 0.0f : KD_NEG, at = as;
    for (int i = 0; i < L; ++i) {
      const float e1 = live ? es[(size_t)i * T + t] : 0.0f;
      const float e2 = live ? et[(size_t)i * T + t] : 0.0f;
      float x[KD_TT];
#pragma unroll
      for (int f = 0; f < KD_TT; ++f) x[f] = (e1 + row[f]) + kd_bcast(as, f);
      const float n1 = kd_lse(x);
#pragma unroll
      for (int f = 0; f < KD_TT; ++f) x[f] = (e2 + row[f]) + kd_bcast(at, f);
      const float n2 = kd_lse(x);
      as = live ? n1 : KD_NEG;
      at = live ? n2 : KD_NEG;
      if (live) {
        As[(size_t)i * T + t] = as;
        At[(size_t)i * T + t] = at;
      }
    }
  }

  // ---- scan 2: beta of both views, tempered marginals, loss, G
  float lacc = 0.0f;
  {
    float bs = live ? trans[stop * T + t] : KD_NEG, bt = bs;
    for (int i = L - 1; i >= 0; --i) {
      const float gs = live ? (As[(size_t)i * T + t] + bs) * inv_tau : -INFINITY;
      const float gt = live ? (At[(size_t)i * T + t] + bt) * inv_tau : -INFINITY;
      const float ms = wave_max(gs), mt = wave_max(gt);
      const float xs = live ? __expf(gs - ms) : 0.0f, xt = live ? __expf(gt - mt) : 0.0f;
      const float zs = wave_sum(xs), zt = wave_sum(xt);
      const float q = xs / zs, p = xt / zt;
      if (live) {
        const float logq = gs - ms - logf(zs), logp = gt - mt - logf(zt);
        if (p > 0.0f) lacc += p * (logp - logq);   // kl_div's 0 * log 0 = 0 convention
        G[(size_t)i * T + t] = w * tau * (q - p);
        Bs[(size_t)i * T + t] = bs;
      }
      if (i > 0) {
        const float e1 = live ? es[(size_t)i * T + t] : 0.0f;
        const float e2 = live ? et[(size_t)i * T + t] : 0.0f;
        float x[KD_TT];
#pragma unroll
        for (int u = 0; u < KD_TT; ++u) x[u] = (kd_bcast(e1, u) + col[u]) + kd_bcast(bs, u);
        const float n1 = kd_lse(x);
#pragma unroll
        for (int u = 0; u < KD_TT; ++u) x[u] = (kd_bcast(e2, u) + col[u]) + kd_bcast(bt, u);
        const float n2 = kd_lse(x);
        bs = live ? n1 : KD_NEG;
        bt = live ? n2 : KD_NEG;
      }
    }
  }
  lacc = wave_sum(lacc);
  if (t == 0) loss[b] = tau * tau * lacc;

  float drow[KD_TT], dcol[KD_TT];   // d trans[t,:] (alpha adjoint), d trans[:,t] (beta adjoint)
#pragma unroll
  for (int f = 0; f < KD_TT; ++f) drow[f] = dcol[f] = 0.0f;

  // ---- scan 3: adjoint of the alpha recursion, i = L-1 .. 0
  float dstart = 0.0f;
  {
    float carry = 0.0f;
    for (int i = L - 1; i >= 0; --i) {
      const float abar = live ? G[(size_t)i * T + t] + carry : 0.0f;
      if (live) de[(size_t)i * T + t] = abar;
      if (i == 0) {
        dstart = abar;   // alpha_0[t] = e_0[t] + trans[t,START]
        break;
      }
      const float c = live ? As[(size_t)i * T + t] - es[(size_t)i * T + t] : 0.0f;   // lse_f(alpha_{i-1}[f] + trans[t,f])
      const float ap = live ? As[(size_t)(i - 1) * T + t] : KD_NEG;
      float nc = 0.0f;
#pragma unroll
      for (int f = 0; f < KD_TT; ++f) {
        drow[f] += abar * __expf((kd_bcast(ap, f) + row[f]) - c);                        // lane = "to" tag
        nc += kd_bcast(abar, f) * __expf((ap + col[f]) - kd_bcast(c, f));                 // lane = "from" tag
      }
      carry = nc;
    }
  }

  // ---- scan 4: adjoint of the beta recursion, j = 0 .. L-1
  float dstop = 0.0f;
  {
    float carry = 0.0f;
    for (int j = 0; j < L; ++j) {
      const float bbar = live ? G[(size_t)j * T + t] + carry : 0.0f;
      if (j == L - 1) {
        dstop = bbar;   // beta_{L-1}[t] = trans[STOP,t]
        break;
      }
      const float bj = live ? Bs[(size_t)j * T + t] : 0.0f;
      const float y = live ? es[(size_t)(j + 1) * T + t] + Bs[(size_t)(j + 1) * T + t] : 0.0f;
      float s = 0.0f;
#pragma unroll
      for (int u = 0; u < KD_TT; ++u) {
        dcol[u] += bbar * __expf((kd_bcast(y, u) + col[u]) - bj);                          // lane = tag of token j ("from")
        s += kd_bcast(bbar, u) * __expf((y + row[u]) - kd_bcast(bj, u));                    // lane = tag of token j+1 ("to")
      }
      if (live) de[(size_t)(j + 1) * T + t] += s;
      carry = s;
    }
  }

  if (live) {
#pragma unroll
    for (int f = 0; f < KD_TT; ++f)
      if (f < T) {
        if (drow[f] != 0.0f) atomicAdd(dtrans + t * T + f, drow[f]);
        if (dcol[f] != 0.0f) atomicAdd(dtrans + f * T + t, dcol[f]);
      }
    if (dstart != 0.0f) atomicAdd(dtrans + t * T + start, dstart);
    if (dstop != 0.0f) atomicAdd(dtrans + stop * T + t, dstop);
  }
}

extern "C" {

// floats of workspace kbner_crf_posterior_kl needs
size_t kbner_crf_posterior_kl_ws_floats(int B, int n, int T) { return (size_t)B * 4 * n * T; }

// Multi-view posterior distillation: loss[b] = T^2 sum_i KL(teacher || student tempered token marginals) for the student
// emissions emit_s f32[B,n,T] against the (constant) teacher emissions emit_t f32[B,n,T]; the gradient of sum_b wgt[b] * loss[b]
// with respect to emit_s is WRITTEN to demit f32[B,n,T] (rows >= lens[b] zero) and with respect to the transitions ADDED to
// dtrans f32[T,T] (atomics).  T <= 32; tau > 0.
int kbner_crf_posterior_kl(const float* emit_s, const float* emit_t, const float* trans, const int* lens, const float* wgt,
                           float tau, int B, int n, int T, int start, int stop, float* loss, float* demit, float* dtrans,
                           float* ws, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= KD_TT && tau > 0.0f);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  KBNER_CHECK_ARG(emit_s != nullptr && emit_t != nullptr && trans != nullptr && lens != nullptr && wgt != nullptr &&
                  loss != nullptr && demit != nullptr && dtrans != nullptr && ws != nullptr);
  hipLaunchKernelGGL(crf_posterior_kl_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, emit_s, emit_t, trans, lens, wgt, tau, n,
                     T, start, stop, loss, demit, dtrans, ws);
  KBNER_LAUNCH_RET();
}

}  // extern "C"
