// n-best Viterbi decode of the linear-chain CRF (SequenceTagger._viterbi_decode_nbest, flair/models/sequence_tagger_model.py
// :1660-1818 -- the NCRF++ decoder the knowledge-distillation trainers run on their teachers).  One wavefront per sentence,
// lane j = target tag: per step every lane keeps the `nbest` largest of its T * nbest candidates
//     cand(i, k -> j) = (emit[t, j] + trans[i, j]) + part[i][k]            (same association order as the reference)
// in a register-resident sorted list (insertion with strict '>', so equal values keep the lower flat index i * nbest + k
// first: torch.topk leaves that order unspecified, oracle/crf.py:viterbi_nbest pins it the same way), publishes its new
// partition row through LDS and its flat back-pointers to a global int16 workspace; the backtrace is done by lanes k < nbest.
// The decoder's conventions and defects are reproduced as they are (see the oracle's docstring): trans indexed [from, to],
// back-pointers of padded steps zero, end pointers written over position len-1 for every tag row (so a sentence shorter than
// the batch maximum reads pointer[pointer[k] % nbest] there), padded positions decode to 0 except the last column.
#include "common.h"

#define NBEST_MAXT 64

template <int NB>
static __device__ __forceinline__ void nb_insert(float (&val)[NB], int (&idx)[NB], float v, int id) {
  if (v > val[NB - 1]) {
    val[NB - 1] = v;
    idx[NB - 1] = id;
#pragma unroll
    for (int p = NB - 1; p > 0; --p) {
      const bool up = val[p] > val[p - 1];
      const float a = val[p - 1], b = val[p];
      const int ia = idx[p - 1], ib = idx[p];
      val[p - 1] = up ? b : a;
      val[p] = up ? a : b;
      idx[p - 1] = up ? ib : ia;
      idx[p] = up ? ia : ib;
    }
  }
}

template <int NB>
__global__ __launch_bounds__(64) void crf_viterbi_nbest_kernel(const float* __restrict__ emit, const float* __restrict__ trans,
                                                               const int* __restrict__ lens, int n, int T, int start, int stop,
                                                               int nbest, short* __restrict__ bpws, int* __restrict__ decode,
                                                               float* __restrict__ pscore) {
  __shared__ float part[2][NBEST_MAXT][NB];
  __shared__ int s_ptr[NB];
  __shared__ float s_end[NB];
  const int b = blockIdx.x, j = threadIdx.x;
  const int L = lens[b];
  const bool live = j < T;
  const float* em = emit + (size_t)b * n * T;
  short* bp = bpws + (size_t)b * n * T * nbest;   // bp[t][j][k]: flat back-pointer of step t + 1 (the reference's back_points[t])
  if (live) {
    const float p0 = em[j] + trans[start * T + j];
    for (int k = 0; k < nbest; ++k) part[0][j][k] = p0;
  }
  __syncthreads();
  int cur = 0;
  for (int t = 1; t < L; ++t) {
    if (live) {
      float val[NB];
      int idx[NB];
#pragma unroll
      for (int p = 0; p < NB; ++p) {
        val[p] = -INFINITY;
        idx[p] = 0;
      }
      const float e = em[(size_t)t * T + j];
      if (t == 1) {
        for (int i = 0; i < T; ++i) nb_insert<NB>(val, idx, (e + trans[i * T + j]) + part[cur][i][0], i * nbest);
      } else {
        for (int i = 0; i < T; ++i) {
          const float s = e + trans[i * T + j];
          for (int k = 0; k < nbest; ++k) nb_insert<NB>(val, idx, s + part[cur][i][k], i * nbest + k);
        }
      }
#pragma unroll
      for (int p = 0; p < NB; ++p)
        if (p < nbest) {
          part[cur ^ 1][j][p] = val[p];
          bp[((size_t)(t - 1) * T + j) * nbest + p] = (short)idx[p];
        }
    }
    cur ^= 1;
    __syncthreads();
  }
  // last_partition + trans[:, j], column STOP (:1747-1763)
  if (j == stop) {
    float val[NB];
    int idx[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      val[p] = -INFINITY;
      idx[p] = 0;
    }
    for (int i = 0; i < T; ++i)
      for (int k = 0; k < nbest; ++k) nb_insert<NB>(val, idx, part[cur][i][k] + trans[i * T + j], i * nbest + k);
    float mx = val[0], sum = 0.0f;
#pragma unroll
    for (int p = 0; p < NB; ++p)
      if (p < nbest) sum += __expf(val[p] - mx);
#pragma unroll
    for (int p = 0; p < NB; ++p)
      if (p < nbest) {
        s_ptr[p] = idx[p];
        s_end[p] = __expf(val[p] - mx) / sum;
      }
  }
  __threadfence_block();
  __syncthreads();
  if (j < nbest) {
    const int k = j;
    pscore[(size_t)b * nbest + k] = s_end[k];
    int* dec = decode + (size_t)b * n * nbest;
    int pointer = s_ptr[k];
    dec[(size_t)(n - 1) * nbest + k] = pointer / nbest;
    for (int t = n - 2; t >= 0; --t) {
      int nw;
      if (t >= L) nw = 0;                                            // masked back-pointers (:1742)
      else if (t == L - 1) nw = s_ptr[pointer % nbest];               // the end pointers, scattered over every tag row (:1766-1775)
      else nw = (int)bp[((size_t)t * T + pointer / nbest) * nbest + pointer % nbest];
      dec[(size_t)t * nbest + k] = nw / nbest;
      pointer = (t >= L) ? nw + pointer : nw;
    }
  }
}

extern "C" {

// emit f32 [B, n, T], trans f32 [T, T], lens i32 [B] (1 <= lens <= n) -> decode i32 [B, n, nbest], path_score f32 [B, nbest].
// ws: int16 [B * n * T * nbest] back-pointer workspace (kbner_crf_viterbi_nbest_ws_bytes).  T <= 64, 1 <= nbest <= 16.
size_t kbner_crf_viterbi_nbest_ws_bytes(int B, int n, int T, int nbest) { return (size_t)B * n * T * nbest * sizeof(short); }

int kbner_crf_viterbi_nbest(const float* emit, const float* trans, const int* lens, int B, int n, int T, int start, int stop,
                            int nbest, void* ws, int* decode, float* path_score, void* stream) {
  KBNER_CHECK_ARG(B >= 0 && n >= 1 && T > 0 && T <= NBEST_MAXT && nbest >= 1 && nbest <= 16 && T * nbest <= 32767);
  // the first step offers only T candidates (one per source tag): the reference's torch.topk(nbest) over them raises for
  // nbest > T (sequence_tagger_model.py:2071-2237), and the merged lists would carry -inf / duplicate paths here
  KBNER_CHECK_ARG(nbest <= T);
  KBNER_CHECK_ARG(start >= 0 && start < T && stop >= 0 && stop < T && ws != nullptr && decode != nullptr && path_score != nullptr);
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  short* w = reinterpret_cast<short*>(ws);
  if (nbest <= 4)
    hipLaunchKernelGGL(crf_viterbi_nbest_kernel<4>, dim3(B), dim3(64), 0, st, emit, trans, lens, n, T, start, stop, nbest, w, decode, path_score);
  else if (nbest <= 8)
    hipLaunchKernelGGL(crf_viterbi_nbest_kernel<8>, dim3(B), dim3(64), 0, st, emit, trans, lens, n, T, start, stop, nbest, w, decode, path_score);
  else
    hipLaunchKernelGGL(crf_viterbi_nbest_kernel<16>, dim3(B), dim3(64), 0, st, emit, trans, lens, n, T, start, stop, nbest, w, decode, path_score);
  KBNER_LAUNCH_RET();
}

}  // extern "C"
