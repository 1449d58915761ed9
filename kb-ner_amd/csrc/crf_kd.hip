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

// TSCORE: emit_t holds the teacher's forward-backward SCORES (alpha + beta of ITS OWN CRF, computed once before training by
// kbner_crf_fb_score) instead of emissions to be scanned with the student's transitions -- the teacher-student `distill_posterior`
// branch of simple_forward_distillation_loss (sequence_tagger_model.py:2120-2136) as opposed to the multi-view one (:2080-2093).
template <bool TSCORE>
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
      float x[KD_TT];
#pragma unroll
      for (int f = 0; f < KD_TT; ++f) x[f] = (e1 + row[f]) + kd_bcast(as, f);
      const float n1 = kd_lse(x);
      as = live ? n1 : KD_NEG;
      if (live) As[(size_t)i * T + t] = as;
      if (!TSCORE) {
        const float e2 = live ? et[(size_t)i * T + t] : 0.0f;
#pragma unroll
        for (int f = 0; f < KD_TT; ++f) x[f] = (e2 + row[f]) + kd_bcast(at, f);
        const float n2 = kd_lse(x);
        at = live ? n2 : KD_NEG;
        if (live) At[(size_t)i * T + t] = at;
      }
    }
  }

  // ---- scan 2: beta of both views, tempered marginals, loss, G
  float lacc = 0.0f;
  {
    float bs = live ? trans[stop * T + t] : KD_NEG, bt = bs;
    for (int i = L - 1; i >= 0; --i) {
      const float gs = live ? (As[(size_t)i * T + t] + bs) * inv_tau : -INFINITY;
      const float gt = live ? (TSCORE ? et[(size_t)i * T + t] : At[(size_t)i * T + t] + bt) * inv_tau : -INFINITY;
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
        float x[KD_TT];
#pragma unroll
        for (int u = 0; u < KD_TT; ++u) x[u] = (kd_bcast(e1, u) + col[u]) + kd_bcast(bs, u);
        const float n1 = kd_lse(x);
        bs = live ? n1 : KD_NEG;
        if (!TSCORE) {
          const float e2 = live ? et[(size_t)i * T + t] : 0.0f;
#pragma unroll
          for (int u = 0; u < KD_TT; ++u) x[u] = (kd_bcast(e2, u) + col[u]) + kd_bcast(bt, u);
          const float n2 = kd_lse(x);
          bt = live ? n2 : KD_NEG;
        }
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

// Forward-backward scores of a (teacher) CRF: score[b,i,t] = alpha_i[t] + beta_i[t] for i < lens[b], 0 past it -- what
// ModelFinetuner.assign_pretrained_teacher_targets stores per sentence as the `distill_posterior` target
// (finetune_trainer.py:1627-1634: `(forward_var + backward_var) * mask` with the teacher's own transitions, after the logits of
// START / STOP / <unk> were lowered by 1e12 -- here the tags of the `suppress` bit mask).  alpha is written into `score` by the
// forward scan and beta added in place by the backward scan: no workspace.
__global__ __launch_bounds__(64) void crf_fb_score_kernel(const float* __restrict__ emit, const float* __restrict__ trans,
                                                          const int* __restrict__ lens, unsigned suppress, int n, int T, int start,
                                                          int stop, float* __restrict__ score) {
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int L = lens[b];
  const bool live = t < T;
  const float off = (live && ((suppress >> t) & 1u)) ? 1e12f : 0.0f;
  float row[KD_TT], col[KD_TT];
#pragma unroll
  for (int f = 0; f < KD_TT; ++f) {
    row[f] = (live && f < T) ? trans[t * T + f] : -INFINITY;
    col[f] = (live && f < T) ? trans[f * T + t] : -INFINITY;
  }
  const float* e = emit + (size_t)b * n * T;
  float* sc = score + (size_t)b * n * T;
  for (int i = max(L, 0) * T + t; i < n * T; i += 64) sc[i] = 0.0f;
  if (L <= 0) return;
  float a = (t == start) ? 0.0f : KD_NEG;
  for (int i = 0; i < L; ++i) {
    const float e1 = live ? e[(size_t)i * T + t] - off : 0.0f;
    float x[KD_TT];
#pragma unroll
    for (int f = 0; f < KD_TT; ++f) x[f] = (e1 + row[f]) + kd_bcast(a, f);
    const float n1 = kd_lse(x);
    a = live ? n1 : KD_NEG;
    if (live) sc[(size_t)i * T + t] = a;
  }
  float bt = live ? trans[stop * T + t] : KD_NEG;
  for (int i = L - 1; i >= 0; --i) {
    if (live) sc[(size_t)i * T + t] += bt;
    if (i > 0) {
      const float e1 = live ? e[(size_t)i * T + t] - off : 0.0f;
      float x[KD_TT];
#pragma unroll
      for (int u = 0; u < KD_TT; ++u) x[u] = (kd_bcast(e1, u) + col[u]) + kd_bcast(bt, u);
      const float n1 = kd_lse(x);
      bt = live ? n1 : KD_NEG;
    }
  }
}

// Teacher side of `distill_exact` (finetune_trainer.py:1705-1722,1885): for every adjacent token pair (i-1, i), i < lens[b], the
// softmax over the T*T (to, from) tag pairs of (alpha_{i-1}[from] + beta_i[to] + e_i[to] + trans[to, from]) / tau -- the
// teacher's tempered pairwise posterior -- plus start_score[t] = (e_0[t] + trans[t, START] + beta_0[t]) / tau and
// end_score[t] = (trans[STOP, t] + alpha_{L-1}[t]) / tau.  Emissions of the `suppress` tags lowered by 1e12 as in
// crf_fb_score_kernel.  pair rows at or past lens[b] - 1 are written as zeros (the reference stores softmax(0) = 1 / T^2 there
// and multiplies them by a zero mask in the loss, sequence_tagger_model.py:2170,2414).  ws: n * T floats per sentence (alpha).
__global__ __launch_bounds__(64) void crf_pair_posterior_kernel(const float* __restrict__ emit, const float* __restrict__ trans,
                                                                const int* __restrict__ lens, unsigned suppress, float tau, int n,
                                                                int T, int start, int stop, float* __restrict__ pair,
                                                                float* __restrict__ start_score, float* __restrict__ end_score,
                                                                float* __restrict__ ws) {
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int L = lens[b];
  const bool live = t < T;
  const float off = (live && ((suppress >> t) & 1u)) ? 1e12f : 0.0f;
  const float inv_tau = 1.0f / tau;
  float row[KD_TT], col[KD_TT];
#pragma unroll
  for (int f = 0; f < KD_TT; ++f) {
    row[f] = (live && f < T) ? trans[t * T + f] : -INFINITY;
    col[f] = (live && f < T) ? trans[f * T + t] : -INFINITY;
  }
  const float* e = emit + (size_t)b * n * T;
  float* A = ws + (size_t)b * n * T;
  float* P = pair + (size_t)b * (n > 0 ? n - 1 : 0) * T * T;
  for (size_t i = (size_t)max(L - 1, 0) * T * T + t; i < (size_t)(n > 0 ? n - 1 : 0) * T * T; i += 64) P[i] = 0.0f;
  if (L <= 0) {
    if (live) start_score[(size_t)b * T + t] = end_score[(size_t)b * T + t] = 0.0f;
    return;
  }
  float a = (t == start) ? 0.0f : KD_NEG;
  for (int i = 0; i < L; ++i) {
    const float e1 = live ? e[(size_t)i * T + t] - off : 0.0f;
    float x[KD_TT];
#pragma unroll
    for (int f = 0; f < KD_TT; ++f) x[f] = (e1 + row[f]) + kd_bcast(a, f);
    const float n1 = kd_lse(x);
    a = live ? n1 : KD_NEG;
    if (live) A[(size_t)i * T + t] = a;
  }
  if (live) end_score[(size_t)b * T + t] = (trans[stop * T + t] + a) * inv_tau;
  float bt = live ? trans[stop * T + t] : KD_NEG;
  for (int i = L - 1; i >= 1; --i) {
    const float e1 = live ? e[(size_t)i * T + t] - off : 0.0f;
    const float ap = live ? A[(size_t)(i - 1) * T + t] : KD_NEG;   // alpha_{i-1}[lane]
    float x[KD_TT];
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < KD_TT; ++f) {
      x[f] = live ? (((kd_bcast(ap, f) + bt) + (e1 + row[f])) * inv_tau) : -INFINITY;
      mx = fmaxf(mx, x[f]);
    }
    mx = wave_max(mx);
    float sm = 0.0f;
#pragma unroll
    for (int f = 0; f < KD_TT; ++f) {
      x[f] = __expf(x[f] - mx);
      sm += x[f];
    }
    sm = wave_sum(sm);
    const float inv = 1.0f / sm;
    if (live) {
      float* dst = P + (size_t)(i - 1) * T * T + (size_t)t * T;
#pragma unroll
      for (int f = 0; f < KD_TT; ++f)
        if (f < T) dst[f] = x[f] * inv;
    }
    float y[KD_TT];
#pragma unroll
    for (int u = 0; u < KD_TT; ++u) y[u] = (kd_bcast(e1, u) + col[u]) + kd_bcast(bt, u);
    const float n1 = kd_lse(y);
    bt = live ? n1 : KD_NEG;
  }
  if (live) start_score[(size_t)b * T + t] = (((e[t] - off) + trans[t * T + start]) + bt) * inv_tau;
}

// Student side of `distill_exact` (simple_forward_distillation_loss, sequence_tagger_model.py:2139-2244, and
// _calculate_xstruct_distillation_loss, :2400-2425), forward AND backward:
//     loss[b] = max(0, -(E_b - logZ_tau) * tau^2)
//     E_b = sum_{i>=1} sum_{t,f} pair_i[t,f] (e_i[t] + trans[t,f]) / tau + sum_t softmax(start_score)[t] (e_0[t] + trans[t,START]) / tau
//           + sum_t softmax(end_score)[t] trans[STOP,t] / tau,          logZ_tau = the partition of the CRF (e / tau, trans / tau)
// The reference differentiates it through autograd; here d loss / d e_i[t] = tau (mu_i[t] - sum_f pair_i[t,f]) with mu the token
// marginals of the tempered CRF, d loss / d trans[t,f] = tau sum_i (nu_i[t,f] - pair_i[t,f]) with nu its pairwise marginals, and
// the START column / STOP row likewise; a sentence whose loss was clamped to 0 contributes no gradient (the reference overwrites
// it with a constant).  demit WRITTEN, dtrans ADDED, each scaled by wgt[b].  ws: n * T floats per sentence.
__global__ __launch_bounds__(64) void crf_exact_kd_kernel(const float* __restrict__ emit, const float* __restrict__ trans,
                                                          const int* __restrict__ lens, const float* __restrict__ pair,
                                                          const float* __restrict__ start_score, const float* __restrict__ end_score,
                                                          const float* __restrict__ wgt, float tau, int n, int T, int start,
                                                          int stop, float* __restrict__ loss, float* __restrict__ demit,
                                                          float* __restrict__ dtrans, float* __restrict__ ws) {
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int L = lens[b];
  const bool live = t < T;
  const float inv_tau = 1.0f / tau;
  float row[KD_TT], col[KD_TT];   // trans[t,:] / tau, trans[:,t] / tau
#pragma unroll
  for (int f = 0; f < KD_TT; ++f) {
    row[f] = (live && f < T) ? trans[t * T + f] * inv_tau : -INFINITY;
    col[f] = (live && f < T) ? trans[f * T + t] * inv_tau : -INFINITY;
  }
  const float* e = emit + (size_t)b * n * T;
  const float* P = pair + (size_t)b * (n > 0 ? n - 1 : 0) * T * T;
  float* de = demit + (size_t)b * n * T;
  float* A = ws + (size_t)b * n * T;
  for (int i = t; i < n * T; i += 64) de[i] = 0.0f;
  if (L <= 0) {
    if (t == 0) loss[b] = 0.0f;
    return;
  }
  // softmax of the teacher's start / end scores
  float ps, pe;
  {
    const float s1 = live ? start_score[(size_t)b * T + t] : -INFINITY, s2 = live ? end_score[(size_t)b * T + t] : -INFINITY;
    const float m1 = wave_max(s1), m2 = wave_max(s2);
    const float x1 = live ? __expf(s1 - m1) : 0.0f, x2 = live ? __expf(s2 - m2) : 0.0f;
    ps = x1 / wave_sum(x1);
    pe = x2 / wave_sum(x2);
  }
  const float tstart = live ? trans[t * T + start] * inv_tau : 0.0f;   // trans[t, START] / tau
  const float tstop = live ? trans[stop * T + t] * inv_tau : 0.0f;     // trans[STOP, t] / tau
  // ---- forward scan: tempered alpha + the teacher's expected score
  float expect = 0.0f;
  float a = (t == start) ? 0.0f : KD_NEG;
  for (int i = 0; i < L; ++i) {
    const float e1 = live ? e[(size_t)i * T + t] * inv_tau : 0.0f;
    float x[KD_TT];
#pragma unroll
    for (int f = 0; f < KD_TT; ++f) x[f] = (e1 + row[f]) + kd_bcast(a, f);
    const float n1 = kd_lse(x);
    a = live ? n1 : KD_NEG;
    if (live) {
      A[(size_t)i * T + t] = a;
      if (i == 0) {
        expect += ps * (e1 + tstart);
      } else {
        const float* src = P + (size_t)(i - 1) * T * T + (size_t)t * T;
#pragma unroll
        for (int f = 0; f < KD_TT; ++f)
          if (f < T) expect += src[f] * (e1 + row[f]);
      }
    }
  }
  if (live) expect += pe * tstop;
  expect = wave_sum(expect);
  float logz;
  {
    const float v = live ? a + tstop : -INFINITY;
    const float m = wave_max(v);
    logz = m + logf(wave_sum(live ? __expf(v - m) : 0.0f));
  }
  const float lb = -(expect - logz) * tau * tau;
  if (t == 0) loss[b] = lb < 0.0f ? 0.0f : lb;
  if (lb < 0.0f) return;
  const float w = wgt[b] * tau;   // -tau^2 * (1 / tau): both E and logZ_tau depend on (e, trans) through (e, trans) / tau
  // ---- backward scan: beta, marginals, gradients
  float drow[KD_TT];
#pragma unroll
  for (int f = 0; f < KD_TT; ++f) drow[f] = 0.0f;
  float bt = live ? tstop : KD_NEG;
  float dstop = 0.0f, dstart = 0.0f;
  for (int i = L - 1; i >= 0; --i) {
    const float e1 = live ? e[(size_t)i * T + t] * inv_tau : 0.0f;
    const float ai = live ? A[(size_t)i * T + t] : KD_NEG;
    const float mu = live ? __expf((ai + bt) - logz) : 0.0f;
    if (i == L - 1) dstop = w * (mu - pe);
    float pm = 0.0f;
    if (i == 0) {
      pm = ps;
      dstart = w * (mu - ps);
    } else {
      const float ap = live ? A[(size_t)(i - 1) * T + t] : KD_NEG;
      const float c = (e1 + bt) - logz;
      const float* src = P + (size_t)(i - 1) * T * T + (size_t)t * T;
#pragma unroll
      for (int f = 0; f < KD_TT; ++f) {
        const float pv = (live && f < T) ? src[f] : 0.0f;
        const float nu = live ? __expf((kd_bcast(ap, f) + row[f]) + c) : 0.0f;
        drow[f] += w * (nu - pv);
        pm += pv;
      }
    }
    if (live) de[(size_t)i * T + t] = w * (mu - pm);
    if (i > 0) {
      float y[KD_TT];
#pragma unroll
      for (int u = 0; u < KD_TT; ++u) y[u] = (kd_bcast(e1, u) + col[u]) + kd_bcast(bt, u);
      const float n1 = kd_lse(y);
      bt = live ? n1 : KD_NEG;
    }
  }
  if (live) {
#pragma unroll
    for (int f = 0; f < KD_TT; ++f)
      if (f < T && drow[f] != 0.0f) atomicAdd(dtrans + t * T + f, drow[f]);
    if (dstart != 0.0f) atomicAdd(dtrans + t * T + start, dstart);
    if (dstop != 0.0f) atomicAdd(dtrans + stop * T + t, dstop);
  }
}

// Emission-level distillation of a CRF student (`distill_emission`, sequence_tagger_model.py:2311-2365 -> :2384-2398): per token
// T^2 KL(p || softmax(emit / tau)) with p = softmax(teacher / tau) or, when the trainer stored probabilities (distill_prob), the
// teacher row itself; a term with p = 0 contributes 0 (torch's kl_div).  One wave per sentence walks its tokens, lane = tag:
// no scan, the sentence-level grouping only keeps the per-sentence loss free of atomics.
//   loss[b] = tau^2 sum_{i < len} sum_t p_t (log p_t - log q_t);   demit[b,i,t] = wgt[b] tau (q_t sum_t' p_t' - p_t), 0 behind the end
template <bool PROB>
__global__ __launch_bounds__(64) void emission_kl_kernel(const float* __restrict__ emit, const float* __restrict__ teacher,
                                                         const int* __restrict__ lens, const float* __restrict__ wgt, float tau,
                                                         int n, int T, float* __restrict__ loss, float* __restrict__ demit) {
  const int b = blockIdx.x, t = threadIdx.x;
  const bool live = t < T;
  const int len = min(max(lens[b], 0), n);
  const float w = wgt[b], itau = 1.0f / tau;
  const float* e = emit + (size_t)b * n * T;
  const float* tc = teacher + (size_t)b * n * T;
  float* d = demit + (size_t)b * n * T;
  float acc = 0.0f;
  for (int i = 0; i < len; ++i) {
    const float s = live ? e[(size_t)i * T + t] * itau : -INFINITY;
    const float smax = wave_max(s);
    const float lq = s - smax - __logf(wave_sum(live ? __expf(s - smax) : 0.0f));      // log softmax(emit / tau)
    float p, lp;
    if (PROB) {
      p = live ? tc[(size_t)i * T + t] : 0.0f;
      lp = p > 0.0f ? __logf(p) : 0.0f;
    } else {
      const float u = live ? tc[(size_t)i * T + t] * itau : -INFINITY;
      const float umax = wave_max(u);
      lp = u - umax - __logf(wave_sum(live ? __expf(u - umax) : 0.0f));
      p = live ? __expf(lp) : 0.0f;
    }
    const float psum = PROB ? wave_sum(p) : 1.0f;
    if (live) {
      acc += p > 0.0f ? p * (lp - lq) : 0.0f;
      d[(size_t)i * T + t] = w * tau * (__expf(lq) * psum - p);
    }
  }
  for (int i = len; i < n; ++i)
    if (live) d[(size_t)i * T + t] = 0.0f;
  acc = wave_sum(acc);
  if (t == 0) loss[b] = acc * tau * tau;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Softmax head -- FastSequenceTagger(use_crf=False), the reference's "softmax student" (sequence_tagger_model.py:2523-2539: token-level
// cross entropy of the emissions under the (remove_x-narrowed) mask; :1177-1180, 1212-1218: arg-max of the emissions, confidence =
// softmax probability of that tag).  One wave per sentence, lane = tag (T <= 64), tokens in order: the per-sentence loss is a
// deterministic sum, no atomics.
//   loss[b] = sum_{i < len} (logsumexp(e_i) - e_i[tag_i]);   demit[b,i,t] = wgt[b] (softmax(e_i)_t - [t == tag_i]), 0 behind the end
__global__ __launch_bounds__(64) void softmax_ce_kernel(const float* __restrict__ emit, const int* __restrict__ tags,
                                                        const int* __restrict__ lens, const float* __restrict__ wgt, int n, int T,
                                                        float* __restrict__ loss, float* __restrict__ demit) {
  const int b = blockIdx.x, t = threadIdx.x;
  const bool live = t < T;
  const int len = min(max(lens[b], 0), n);
  const float w = wgt[b];
  const float* e = emit + (size_t)b * n * T;
  const int* tg = tags + (size_t)b * n;
  float* d = demit + (size_t)b * n * T;
  float acc = 0.0f;
  for (int i = 0; i < len; ++i) {
    const float s = live ? e[(size_t)i * T + t] : -INFINITY;
    const float smax = wave_max(s);
    const float lq = s - smax - __logf(wave_sum(live ? __expf(s - smax) : 0.0f));      // log softmax(e_i)
    const int gold = tg[i];
    if (live) {
      if (t == gold) acc -= lq;
      d[(size_t)i * T + t] = w * (__expf(lq) - (t == gold ? 1.0f : 0.0f));
    }
  }
  for (int i = len; i < n; ++i)
    if (live) d[(size_t)i * T + t] = 0.0f;
  acc = wave_sum(acc);
  if (t == 0) loss[b] = acc;
}

// tags[b,i] = first arg-max of e_i (torch.max's choice among equal values on contiguous rows), conf[b,i] = softmax(e_i)[tags], and --
// dist != nullptr -- the whole distribution; positions behind a sentence's end: tag 0, confidence 0, distribution 0
__global__ __launch_bounds__(64) void softmax_decode_kernel(const float* __restrict__ emit, const int* __restrict__ lens, int n, int T,
                                                            int* __restrict__ tags, float* __restrict__ conf,
                                                            float* __restrict__ dist) {
  const int b = blockIdx.x, t = threadIdx.x;
  const bool live = t < T;
  const int len = min(max(lens[b], 0), n);
  const float* e = emit + (size_t)b * n * T;
  for (int i = 0; i < n; ++i) {
    if (i >= len) {
      if (t == 0) {
        tags[(size_t)b * n + i] = 0;
        conf[(size_t)b * n + i] = 0.0f;
      }
      if (dist != nullptr && live) dist[((size_t)b * n + i) * T + t] = 0.0f;
      continue;
    }
    const float s = live ? e[(size_t)i * T + t] : -INFINITY;
    const float smax = wave_max(s);
    const float ex = live ? __expf(s - smax) : 0.0f;
    const float p = ex / wave_sum(ex);
    const unsigned long long hit = __ballot(live && s == smax);
    const int best = __ffsll((long long)hit) - 1;
    const float pb = __shfl(p, best, 64);
    if (t == 0) {
      tags[(size_t)b * n + i] = best;
      conf[(size_t)b * n + i] = pb;
    }
    if (dist != nullptr && live) dist[((size_t)b * n + i) * T + t] = p;
  }
}

extern "C" {

// Emission-level distillation term of a CRF student (see emission_kl_kernel): emit, teacher f32[B,n,T]; teacher_is_prob: the
// teacher rows are probabilities (distill_prob) instead of scores; loss f32[B] WRITTEN (unweighted), demit f32[B,n,T] WRITTEN with
// d(sum_b wgt[b] loss[b]) / d emit.  T <= 64.
int kbner_emission_kl(const float* emit, const float* teacher, const int* lens, const float* wgt, float tau, int teacher_is_prob,
                      int B, int n, int T, float* loss, float* demit, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= 64 && tau > 0.0f);
  if (B == 0) return 0;
  KBNER_CHECK_ARG(emit != nullptr && teacher != nullptr && lens != nullptr && wgt != nullptr && loss != nullptr && demit != nullptr);
  if (teacher_is_prob)
    hipLaunchKernelGGL(emission_kl_kernel<true>, dim3(B), dim3(64), 0, (hipStream_t)stream, emit, teacher, lens, wgt, tau, n, T, loss,
                       demit);
  else
    hipLaunchKernelGGL(emission_kl_kernel<false>, dim3(B), dim3(64), 0, (hipStream_t)stream, emit, teacher, lens, wgt, tau, n, T,
                       loss, demit);
  KBNER_LAUNCH_RET();
}

// Softmax head (use_crf = false): token-level cross entropy.  emit f32[B,n,T], tags i32[B,n] (read below lens[b] only), wgt f32[B]:
// loss f32[B] WRITTEN per sentence (unweighted sum over its tokens), demit f32[B,n,T] WRITTEN with d(sum_b wgt[b] loss[b]) / d emit.
int kbner_softmax_ce(const float* emit, const int* tags, const int* lens, const float* wgt, int B, int n, int T, float* loss,
                     float* demit, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= 64);
  if (B == 0) return 0;
  KBNER_CHECK_ARG(emit != nullptr && tags != nullptr && lens != nullptr && wgt != nullptr && loss != nullptr && demit != nullptr);
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, emit, tags, lens, wgt, n, T, loss, demit);
  KBNER_LAUNCH_RET();
}

// Softmax head, decode: tags i32[B,n] = arg-max tag per token, conf f32[B,n] = its softmax probability, dist f32[B,n,T] (nullable) =
// the distributions (get_all_tags).
int kbner_softmax_decode(const float* emit, const int* lens, int B, int n, int T, int* tags, float* conf, float* dist, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= 64);
  if (B == 0 || n == 0) return 0;
  KBNER_CHECK_ARG(emit != nullptr && lens != nullptr && tags != nullptr && conf != nullptr);
  hipLaunchKernelGGL(softmax_decode_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, emit, lens, n, T, tags, conf, dist);
  KBNER_LAUNCH_RET();
}

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
  hipLaunchKernelGGL(crf_posterior_kl_kernel<false>, dim3(B), dim3(64), 0, (hipStream_t)stream, emit_s, emit_t, trans, lens, wgt,
                     tau, n, T, start, stop, loss, demit, dtrans, ws);
  KBNER_LAUNCH_RET();
}

// Teacher-student posterior distillation (simple_forward_distillation_loss, `distill_posterior` branch,
// sequence_tagger_model.py:2120-2136): as kbner_crf_posterior_kl, but the teacher side is given as its forward-backward SCORES
// score_t f32[B,n,T] (kbner_crf_fb_score of the teacher's emissions under the teacher's transitions), not as emissions.
// Same workspace size, same outputs (demit WRITTEN, dtrans ADDED).
int kbner_crf_posterior_kl_scores(const float* emit_s, const float* score_t, const float* trans, const int* lens, const float* wgt,
                                  float tau, int B, int n, int T, int start, int stop, float* loss, float* demit, float* dtrans,
                                  float* ws, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= KD_TT && tau > 0.0f);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  KBNER_CHECK_ARG(emit_s != nullptr && score_t != nullptr && trans != nullptr && lens != nullptr && wgt != nullptr &&
                  loss != nullptr && demit != nullptr && dtrans != nullptr && ws != nullptr);
  hipLaunchKernelGGL(crf_posterior_kl_kernel<true>, dim3(B), dim3(64), 0, (hipStream_t)stream, emit_s, score_t, trans, lens, wgt,
                     tau, n, T, start, stop, loss, demit, dtrans, ws);
  KBNER_LAUNCH_RET();
}

// score f32[B,n,T] = alpha + beta (log domain) of the CRF (emit, trans) at the tokens below lens[b], 0 past them; the emissions
// of the tags set in `suppress` (bit t = tag t) are lowered by 1e12 first (finetune_trainer.py:1627-1634).  T <= 32.
int kbner_crf_fb_score(const float* emit, const float* trans, const int* lens, unsigned suppress, int B, int n, int T, int start,
                       int stop, float* score, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 0 && T > 0 && T <= KD_TT);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0 || n == 0) return 0;
  KBNER_CHECK_ARG(emit != nullptr && trans != nullptr && lens != nullptr && score != nullptr);
  hipLaunchKernelGGL(crf_fb_score_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, emit, trans, lens, suppress, n, T, start, stop,
                     score);
  KBNER_LAUNCH_RET();
}

// floats of workspace kbner_crf_pair_posterior / kbner_crf_exact_kd need
size_t kbner_crf_pair_ws_floats(int B, int n, int T) { return (size_t)B * n * T; }

// Teacher targets of `distill_exact` (finetune_trainer.py:1705-1722,1885): pair f32[B, n-1, T*T] (index to * T + from),
// start_score / end_score f32[B,T] of the CRF (emit - 1e12 on the `suppress` tags, trans) at temperature tau.  T <= 32.
int kbner_crf_pair_posterior(const float* emit, const float* trans, const int* lens, unsigned suppress, float tau, int B, int n,
                             int T, int start, int stop, float* pair, float* start_score, float* end_score, float* ws,
                             void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 1 && T > 0 && T <= KD_TT && tau > 0.0f);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  KBNER_CHECK_ARG(emit != nullptr && trans != nullptr && lens != nullptr && (pair != nullptr || n == 1) &&
                  start_score != nullptr && end_score != nullptr && ws != nullptr);
  hipLaunchKernelGGL(crf_pair_posterior_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, emit, trans, lens, suppress, tau, n, T,
                     start, stop, pair, start_score, end_score, ws);
  KBNER_LAUNCH_RET();
}

// Student loss of `distill_exact` against those targets (sequence_tagger_model.py:2139-2244,2400-2425): loss f32[B] (clamped
// at 0), demit f32[B,n,T] WRITTEN with d(sum_b wgt[b] loss[b]) / d emit, the transition gradient ADDED to dtrans.  T <= 32.
int kbner_crf_exact_kd(const float* emit, const float* trans, const int* lens, const float* pair, const float* start_score,
                       const float* end_score, const float* wgt, float tau, int B, int n, int T, int start, int stop, float* loss,
                       float* demit, float* dtrans, float* ws, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 1 && T > 0 && T <= KD_TT && tau > 0.0f);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T);
  if (B == 0) return 0;
  KBNER_CHECK_ARG(emit != nullptr && trans != nullptr && lens != nullptr && (pair != nullptr || n == 1) &&
                  start_score != nullptr && end_score != nullptr && wgt != nullptr && loss != nullptr && demit != nullptr &&
                  dtrans != nullptr && ws != nullptr);
  hipLaunchKernelGGL(crf_exact_kd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, emit, trans, lens, pair, start_score,
                     end_score, wgt, tau, n, T, start, stop, loss, demit, dtrans, ws);
  KBNER_LAUNCH_RET();
}

}  // extern "C"
