/* kbner.h -- C ABI of libkbner_hip.so: the MI355X (gfx950) kernels behind KB-NER's token-classification
 * hot path (XLM-R encoder fwd/bwd + emission head + linear-chain CRF + AdamW).
 *
 * The reference (Alibaba-NLP/KB-NER) is pure Python and has NO FFI for this path: the "interface" each
 * entry point replaces is the torch / transformers op call site listed next to it (file:line under the
 * reference tree).  The binding a maintainer adds is a ctypes stub; the build's own
 * binding is kb-ner_amd/kbner/lib.py.
 *
 * Conventions (all functions):
 *   - return 0 on success, -22 (EINVAL) on a bad argument, -(hipError_t) on a launch failure; never throw/exit
 *   - every pointer is a DEVICE pointer owned by the caller (torch) and kept alive until the stream drains
 *   - `stream` is a hipStream_t (NULL = default stream); work is enqueued asynchronously on it
 *   - no device allocation: callers pass workspaces.  Mutable process state, all of it in atomics (entry points may be called
 *     from several host threads and on several devices): once-per-device flags (a kernel's dynamic-LDS limit has been raised on
 *     device d; the CU count of device d), the GEMM main-loop switch (kbner_gemm_set_variant) and, in device memory, the eight
 *     per-XCD round counters of the ring GEMM's long-K launches (monotonic, never reset; csrc/gemm256.hip xcd_tile_sync: two such
 *     launches running CONCURRENTLY on one device would only lose their L2 sharing).  The library reads no environment variable.
 *   - bf16 tensors are raw uint16_t storage, row-major; fp32 statistics / optimizer state / CRF
 */
#ifndef KBNER_H
#define KBNER_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t kbner_bf16;

int kbner_abi_version(void);
int kbner_device_count(void);

/* ---------------- linear-chain CRF (transitions[to,from], START/STOP are tag ids) ---------------- */
/* SequenceTagger._viterbi_decode, flair/models/sequence_tagger_model.py:1248-1327 (called per sentence
 * from _obtain_labels :1193-1210).  emit f32[B,n,T], lens i32[B] -> tags i32[B,n] (-1 past lens),
 * conf f32[B,n] (max of softmax of the Viterbi scores, :1295-1300), popped i32[B] (nullable; the start tag
 * the reference asserts == START at :1302-1303).  Bit-exact tag indices. */
size_t kbner_crf_viterbi_lds_bytes(int n, int T);
int kbner_crf_viterbi(const float* emit, const float* trans, const int* lens, int B, int n, int T, int start, int stop,
                      int* tags, float* conf, int* popped, void* stream);
/* _forward_alg :1329-1394 + FastSequenceTagger._score_sentence :2544-2591 on compacted rows.
 * -> logz f32[B], gold f32[B], alpha f32[B,n+1,T] (saved for backward). */
int kbner_crf_nll_fwd(const float* emit, const float* trans, const int* tags, const int* lens, int B, int n, int T, int start,
                      int stop, float* logz, float* gold, float* alpha, void* stream);
/* autograd backward of the two above (loss.backward(), flair/trainers/finetune_trainer.py:957):
 * demit f32[B,n,T] (overwritten), dtrans f32[T,T] (+=, atomics) for sum_b dloss[b]*(logz_b - gold_b). */
int kbner_crf_nll_bwd(const float* emit, const float* trans, const int* tags, const int* lens, const float* alpha,
                      const float* logz, const float* dloss, int B, int n, int T, int start, int stop, float* demit,
                      float* dtrans, void* stream);

/* token marginals p_i(t) = softmax_t(alpha_i + beta_i): the predict_posterior branch of _obtain_labels (:1182-1192 =
 * _forward_alg(distill_mode=True) :1329 + _backward_alg :1396-1470).  alpha / logz come from kbner_crf_nll_fwd (its tags
 * argument may be all zeros).  marg f32[B,n,T], rows at or past lens[b] are zero-filled. */
int kbner_crf_posterior(const float* emit, const float* trans, const int* lens, const float* alpha, const float* logz, int B,
                        int n, int T, int start, int stop, float* marg, void* stream);

/* Multi-view ("cooperative learning") posterior distillation between two views of the same sentences -- the `distill_posterior`
   branch of FastSequenceTagger._calculate_multi_view_loss (flair/models/sequence_tagger_model.py:2080-2093, loss form
   :2384-2398), called by ModelFinetuner.train (flair/trainers/finetune_trainer.py:959-966).  loss[b] = T^2 * sum_i KL(softmax(
   g^t_i / T) || softmax(g^s_i / T)), g = forward_var + backward_var of the view's emissions; emit_t (the context view's emissions at
   the real tokens) is a constant.  Forward and backward in one launch: d(sum_b wgt[b] loss[b]) / d emit_s is WRITTEN to demit
   f32[B,n,T] (rows >= lens[b] zero), the transition gradient ADDED to dtrans f32[T,T].  T <= 32.
   ws: kbner_crf_posterior_kl_ws_floats(B, n, T) floats. */
size_t kbner_crf_posterior_kl_ws_floats(int B, int n, int T);
int kbner_crf_posterior_kl(const float* emit_s, const float* emit_t, const float* trans, const int* lens, const float* wgt,
                           float tau, int B, int n, int T, int start, int stop, float* loss, float* demit, float* dtrans,
                           float* ws, void* stream);
/* ---- teacher-student knowledge distillation of the CRF (`distill_mode: true`) ----
 * Teacher side, run once per training sentence before the first epoch (ModelFinetuner.assign_pretrained_teacher_targets,
 * flair/trainers/finetune_trainer.py:1515-1910); `suppress` = bit mask of the tags whose emissions are lowered by 1e12 first
 * (START, STOP, <unk>: :1627-1629,1705-1707):
 *   kbner_crf_fb_score        score f32[B,n,T] = forward_var + backward_var of the teacher's CRF below lens[b], 0 past it
 *                             (:1631-1634) -- the `distill_posterior` target;
 *   kbner_crf_pair_posterior  the `distill_exact` targets (:1709-1722,1885): pair f32[B,n-1,T*T] = softmax over (to, from) of
 *                             (alpha_{i-1}[from] + beta_i[to] + e_i[to] + trans[to,from]) / tau (rows past lens[b]-1 zero),
 *                             start_score, end_score f32[B,T]; ws: kbner_crf_pair_ws_floats(B, n, T) floats.
 * Student side, every training step (FastSequenceTagger.simple_forward_distillation_loss,
 * flair/models/sequence_tagger_model.py:2110-2372), forward and backward in one launch, d(sum_b wgt[b] loss[b]) / d emit
 * WRITTEN to demit f32[B,n,T], the transition gradient ADDED to dtrans f32[T,T]:
 *   kbner_crf_posterior_kl_scores  :2120-2136 -- kbner_crf_posterior_kl with the teacher given as its fb scores score_t
 *                                  (its own transitions went into them), same workspace;
 *   kbner_crf_exact_kd             :2139-2244 + _calculate_xstruct_distillation_loss :2400-2425 -- loss[b] = max(0, -(E_teacher[
 *                                  score / tau] - logZ_tau) * tau^2); ws: kbner_crf_pair_ws_floats(B, n, T) floats.
 * (The `distill_crf` branch, :2249-2309, is kbner_crf_nll_fwd / _bwd over the n-best paths of kbner_crf_viterbi_nbest.)  T <= 32. */
int kbner_crf_fb_score(const float* emit, const float* trans, const int* lens, unsigned suppress, int B, int n, int T, int start,
                       int stop, float* score, void* stream);
int kbner_crf_posterior_kl_scores(const float* emit_s, const float* score_t, const float* trans, const int* lens, const float* wgt,
                                  float tau, int B, int n, int T, int start, int stop, float* loss, float* demit, float* dtrans,
                                  float* ws, void* stream);
size_t kbner_crf_pair_ws_floats(int B, int n, int T);
int kbner_crf_pair_posterior(const float* emit, const float* trans, const int* lens, unsigned suppress, float tau, int B, int n,
                             int T, int start, int stop, float* pair, float* start_score, float* end_score, float* ws,
                             void* stream);
int kbner_crf_exact_kd(const float* emit, const float* trans, const int* lens, const float* pair, const float* start_score,
                       const float* end_score, const float* wgt, float tau, int B, int n, int T, int start, int stop, float* loss,
                       float* demit, float* dtrans, float* ws, void* stream);
/* `distill_emission` (sequence_tagger_model.py:2311-2365 -> _calculate_distillation_loss :2384-2398) for a CRF student: per token
 * tau^2 KL(p || softmax(emit / tau)), p = softmax(teacher / tau) or the teacher row itself (teacher_is_prob: `distill_prob`, the
 * trainer stored softmax(logits), finetune_trainer.py:1474); loss f32[B] WRITTEN per sentence (unweighted), demit f32[B,n,T]
 * WRITTEN with d(sum_b wgt[b] loss[b]) / d emit (0 behind a sentence's end).  T <= 64. */
int kbner_emission_kl(const float* emit, const float* teacher, const int* lens, const float* wgt, float tau, int teacher_is_prob,
                      int B, int n, int T, float* loss, float* demit, void* stream);
/* Softmax head -- FastSequenceTagger(use_crf=False), sequence_tagger_model.py:2523-2539 (loss: token-level cross entropy of the
 * emissions under the remove_x-narrowed mask) and :1177-1180, 1212-1218 (decode: arg-max of the emissions, confidence = its softmax
 * probability).  kbner_softmax_ce: loss f32[B] WRITTEN per sentence (unweighted sum over its tokens below lens[b]), demit f32[B,n,T]
 * WRITTEN with d(sum_b wgt[b] loss[b]) / d emit (0 behind a sentence's end).  kbner_softmax_decode: tags i32[B,n], conf f32[B,n],
 * dist f32[B,n,T] (nullable: get_all_tags); behind a sentence's end tag 0 / 0.0.  T <= 64. */
int kbner_softmax_ce(const float* emit, const int* tags, const int* lens, const float* wgt, int B, int n, int T, float* loss,
                     float* demit, void* stream);
int kbner_softmax_decode(const float* emit, const int* lens, int B, int n, int T, int* tags, float* conf, float* dist, void* stream);
/* n-best Viterbi (SequenceTagger._viterbi_decode_nbest, sequence_tagger_model.py:1660-1818; called on KD teachers at
 * finetune_trainer.py:1600, distillation_trainer.py:819): decode i32 [B, n, nbest] tag indices and path_score f32 [B, nbest]
 * (softmax over the nbest end scores).  The NCRF++ decoder's conventions are kept as they are: trans indexed [from, to],
 * padded positions decode to 0 except the last column.  ws: kbner_crf_viterbi_nbest_ws_bytes(B, n, T, nbest) bytes.
 * 1 <= nbest <= min(16, T): the first step offers T candidates, the reference's topk raises beyond that (-> -22 here). */
size_t kbner_crf_viterbi_nbest_ws_bytes(int B, int n, int T, int nbest);
int kbner_crf_viterbi_nbest(const float* emit, const float* trans, const int* lens, int B, int n, int T, int start, int stop,
                            int nbest, void* ws, int* decode, float* path_score, void* stream);

/* ---------------- row moves: pooling, compaction, emission head ---------------- */
/* first-subtoken pooling + assign_batch_features (flair/embeddings.py:3288-3345,108-124) and the remove_x
 * compaction loop (sequence_tagger_model.py:2474-2488) as ONE gather: out[r] = idx[r] >= 0 ? src[idx[r]] : 0 */
int kbner_gather_rows(const kbner_bf16* src, const int* idx, kbner_bf16* out, int R, int H, void* stream);
/* strided variant: out[r, 0:H] (row stride ld_out) = idx[r] >= 0 ? src[idx[r], 0:H] (row stride ld_src) : 0.  One embedding's
 * pooled features written into its column block of the stacked-embedding matrix (torch.cat at sequence_tagger_model.py:879-891) */
int kbner_gather_rows_ld(const kbner_bf16* src, int ld_src, const int* idx, kbner_bf16* out, int ld_out, int R, int H, void* stream);
/* fp32 variant for the evaluation path: emissions [B*n,T] -> the rows _obtain_labels decodes (sequence_tagger_model.py:1198-1200) */
int kbner_gather_rows_f32(const float* src, const int* idx, float* out, int R, int W, void* stream);
/* fp32 row scatter dst[idx[r],:] = rows[r,:] (unique indices, W % 4 == 0): the data-parallel exchange of the touched
 * word-embedding gradient rows -- new capability, the reference has no distributed path (finetune_trainer.py:466,699-700) */
int kbner_scatter_rows_f32(const float* rows, const int* idx, float* dst, int R, int W, void* stream);
/* fp32 row scatter-ADD, any W: dst[idx[r],:] += rows[r,:] (idx[r] < 0 skipped, indices unique) -- backward of the remove_x
 * compaction (sequence_tagger_model.py:2474-2488) when the KD terms need the gradient on the all-token emissions */
int kbner_scatter_add_rows_f32(const float* rows, const int* idx, float* dst, int R, int W, void* stream);
/* calculate_l2_loss of multi-view training (sequence_tagger_model.py:1988-1996,2026-2035), forward + backward:
 * loss[0] += sum_r w[r] sum_h (a[r,h] - b[r,h])^2 and (da != NULL) da[r,:] += 2 * gscale * w[r] * (a[r,:] - b[r,:]);
 * a, b, da bf16 [R,H] (H even), w f32 [R] (0 = skip the row); b is the constant (detached) view */
int kbner_l2_rows(const kbner_bf16* a, const kbner_bf16* b, const float* w, float gscale, kbner_bf16* da, float* loss, int R, int H,
                  void* stream);
/* its backward (unique indices; caller zero-fills dsrc) */
int kbner_scatter_rows(const kbner_bf16* dout, const int* idx, kbner_bf16* dsrc, int R, int H, void* stream);
/* self.linear, sequence_tagger_model.py:1027: out f32[R,T] = x bf16[R,H] . w f32[T,H]^T + bias */
int kbner_head_fwd(const kbner_bf16* x, const float* w, const float* bias, float* out, int R, int H, int T, void* stream);
int kbner_head_bwd_dx(const float* de, const float* w, kbner_bf16* dx, int R, int H, int T, void* stream);
int kbner_head_bwd_dw(const float* de, const kbner_bf16* x, float* dw, float* db, int R, int H, int T, void* stream);
/* bias gradients: out f32[N] += column sums of x bf16[M,N] (row stride ld) */
int kbner_colsum(const kbner_bf16* x, float* out, int M, int N, int ld, void* stream);

/* ---------------- LayerNorm / embeddings (transformers BertEmbeddings, BertSelfOutput, BertOutput) ---------------- */
int kbner_ln_fwd(const kbner_bf16* h, const float* gamma, const float* beta, float eps, kbner_bf16* y, float* mean,
                 float* rstd, int M, int H, void* stream);
/* ws: kbner_ln_bwd_ws_floats(H) floats of scratch for the per-block partial column sums (dgamma, dbeta, dbias are +=) */
int kbner_ln_bwd_ws_floats(int H);
/* drop_thresh != 0 (training with hidden dropout): the branch that fed this LayerNorm's input was dropout(GEMM out);
 * additionally write dhm = mask(drop_seed) * dh / (1-p), the dY of that GEMM; dbias then sums dhm. */
int kbner_ln_bwd(const kbner_bf16* dy, const kbner_bf16* h, const float* mean, const float* rstd, const float* gamma,
                 kbner_bf16* dh, float* dgamma, float* dbeta, float* dbias, float* ws, int M, int H, kbner_bf16* dhm,
                 uint32_t drop_seed, uint32_t drop_thresh, void* stream);
/* kbner_ln_fwd / kbner_ln_bwd whose input row is FOLDED from the fp32 slabs of a split-K GEMM (ws f32 [splits][M, H], the grouped
 * KBNER_EPI_STORE32 launch) on the way in: bf16(dropout(sum_s ws[s] + bias) + addend) -- kbner_splitk_finish's arithmetic, the same
 * bits -- without that launch and without reading its output back.  Forward: the folded row is also stored to `h` (the backward pass
 * reads it).  Backward: the incoming gradient is sum_s dy_ws[s] + dy_add; dgamma == NULL defers the column sums as for kbner_ln_bwd. */
int kbner_ln_fwd_slabs(const float* ws, int splits, const float* bias, const kbner_bf16* addend, int ldadd, uint32_t drop_seed,
                       uint32_t drop_thresh, kbner_bf16* h, const float* gamma, const float* beta, float eps, kbner_bf16* y, float* mean,
                       float* rstd, int M, int H, void* stream);
int kbner_ln_bwd_slabs(const float* dy_ws, int splits, const kbner_bf16* dy_add, int ldadd, const kbner_bf16* h, const float* mean,
                       const float* rstd, const float* gamma, kbner_bf16* dh, float* dgamma, float* dbeta, float* dbias, float* ws, int M,
                       int H, kbner_bf16* dhm, uint32_t drop_seed, uint32_t drop_thresh, void* stream);
/* Small batches: kbner_ln_bwd with dgamma == NULL (dbeta / dbias ignored) leaves its kbner_ln_bwd_blocks(M) partial rows in `ws` instead
 * of reducing them with a launch of its own; kbner_ln_colreduce_batched then adds the column sums of up to 64 such workspaces to their
 * gradients in ONE launch (49 LayerNorm backward passes per encoder backward pass at 4.7 us each otherwise).  items: HOST memory,
 * n records of five 64-bit words: ws, dgamma, dbeta, dbias (device pointers; dbias may be 0), number of partial rows. */
int kbner_ln_bwd_blocks(int M);
int kbner_ln_colreduce_batched(const long long* items, int n, int H, void* stream);
/* word[ids] + pos[pos_ids] + type[0] -> h0 (saved) -> LayerNorm -> dropout -> y (BertEmbeddings.forward) */
int kbner_embed_ln_fwd(const int* ids, const int* pos_ids, const float* word, const float* pos, const float* type0,
                       const float* gamma, const float* beta, float eps, kbner_bf16* h0, kbner_bf16* y, float* mean,
                       float* rstd, int M, int H, uint32_t drop_seed, uint32_t drop_thresh, void* stream);
int kbner_embed_ln_bwd(const kbner_bf16* dy, const kbner_bf16* h0, const float* mean, const float* rstd, const float* gamma,
                       const int* ids, const int* pos_ids, float* dgamma, float* dbeta, float* dword, float* dpos,
                       float* dtype0, float* ws, int M, int H, uint32_t drop_seed, uint32_t drop_thresh, void* stream);
/* the same with the optimizer's embedding-row flags (u8 per row of dword; KBNER_ROW_LIVE | KBNER_ROW_TOUCHED below) set by the kernel
 * for every row it adds a gradient to -- what a kbner_mark_rows launch on `ids` does --, and, like kbner_ln_bwd, with dgamma == NULL
 * leaving the partial column sums in `ws` for kbner_ln_colreduce_batched.  row_flags may be NULL. */
int kbner_embed_ln_bwd_mark(const kbner_bf16* dy, const kbner_bf16* h0, const float* mean, const float* rstd, const float* gamma,
                            const int* ids, const int* pos_ids, float* dgamma, float* dbeta, float* dword, float* dpos,
                            float* dtype0, float* ws, unsigned char* row_flags, int M, int H, uint32_t drop_seed, uint32_t drop_thresh,
                            void* stream);

/* ---------------- dropout (torch.nn.Dropout inside transformers' BertEmbeddings / BertSelfAttention / BertSelfOutput /
 * BertOutput, active while ModelFinetuner trains: finetune_trainer.py:938 model.train()) ----------------
 * Counter-based and replayable: element (i,j) of a site is kept iff
 *   mul24(mix(seed + i) ^ mix((seed*0x9E3779B1 + 0x7F4A7C15) ^ j), 0x9E3779) >= drop_thresh,  drop_thresh = p * 2^32 (mul24: the low 24 bits of both factors, low 32 bits of the product),
 * kept values are scaled by 1/(1-p).  No mask is stored: backward passes the same (seed, thresh).  drop_thresh = 0 disables.
 * kbner_dropout_mask materialises the multiplier (tests): out f32[Z,M,N], element (z,i,j) -> keys (z*M+i, z*N+j). */
int kbner_dropout_mask(float* out, int Z, int M, int N, uint32_t seed, uint32_t thresh, void* stream);

/* ---------------- bf16 MFMA GEMM (torch.nn.Linear fwd/bwd inside transformers' BertLayer) ---------------- */
#define KBNER_GEMM_NT 0 /* C[M,N] = A[M,K] . B[N,K]^T        forward  */
#define KBNER_GEMM_NN 1 /* C[M,N] = A[M,K] . Bmem[K,N]        dgrad    */
#define KBNER_GEMM_TN 2 /* C[M,N] = Amem[K,M]^T . Bmem[K,N]   wgrad    */
#define KBNER_EPI_BIAS 1
#define KBNER_EPI_GELU 2 /* C = gelu(pre), out2 = gelu'(pre) as bf16, pre = bf16(alpha*acc + bias): the derivative is saved, not the pre-activation */
#define KBNER_EPI_GELU_FWD 1024 /* C = gelu(pre) alone, no derivative output: the forward of inference (evaluate, frozen encoders) */
#define KBNER_EPI_ADD 4
#define KBNER_EPI_DGELU 8 /* C = (alpha*acc) * aux, aux = the gelu'(pre) saved by KBNER_EPI_GELU (BertIntermediate's gelu backward) */
#define KBNER_EPI_ATOMIC32 16
#define KBNER_EPI_DROP 128 /* C = dropout(alpha*acc + bias) + addend : hidden dropout of BertSelfOutput / BertOutput */
int kbner_gemm_bf16(int layout, const kbner_bf16* A, int lda, const kbner_bf16* B, int ldb, int M, int N, int K,
                    kbner_bf16* C, int ldc, float* C32, int ldc32, const float* bias, const kbner_bf16* addend, int ldadd,
                    const kbner_bf16* aux, int ldaux, kbner_bf16* out2, int ldout2, int epi, int splitk, float alpha,
                    uint32_t drop_seed, uint32_t drop_thresh, void* stream);

/* Grouped GEMM on the 256x256x64 / 8-wave kernel: up to 16 problems of one layout per launch (the weight-
 * gradient GEMMs of four encoder layers = 768 tiles fill the chip without split-K).  Per problem M,N % 256 == 0, K % 64 == 0. */
#define KBNER_EPI_RMW32 32 /* C32 += result by non-atomic 16-byte read-modify-write */
#define KBNER_EPI_STORE32 256 /* C32 = result, fp32 plain stores (must be the only flag): one split-K slab */
#define KBNER_EPI_COLSUM 64 /* also accumulate the output's column sums (the producing layer's bias gradient) */
#define KBNER_EPI_COLSUM_WS 512 /* with KBNER_EPI_COLSUM: `colsum` is a workspace f32 [2 * M/256, N] written by plain stores (one
                                   line per tile row and wave row) instead of atomics; fold it with kbner_colsum_rows_f32 */
typedef struct kbner_gemm_problem {
  const kbner_bf16* A;
  const kbner_bf16* B;
  kbner_bf16* C;
  float* C32;
  const float* bias;
  const kbner_bf16* addend;
  const kbner_bf16* aux;
  kbner_bf16* out2;
  float* colsum; /* KBNER_EPI_COLSUM: colsum[n] += sum_m out[m,n] (fp32 atomics) */
  int M, N, K;
  int lda, ldb, ldc, ldc32, ldadd, ldaux, ldout2;
  int epi;
  float alpha;
  uint32_t drop_seed, drop_thresh; /* KBNER_EPI_DROP */
} kbner_gemm_problem;
int kbner_gemm_bf16_grouped(int layout, int nprob, const kbner_gemm_problem* probs, void* stream);
/* rows of an output tile the grouped launch uses for ONE problem of this layout and shape: 256, or 128 when the 256 x 256 tiling
   would give at most half of the CUs a tile (small micro-batches).  A KBNER_EPI_COLSUM_WS workspace has 2 * M / rows lines. */
int kbner_gemm_tile_rows(int layout, int M, int N);
/* out[n] += sum_r ws[r, n], rows = 2 * M / tile rows: the second half of KBNER_EPI_COLSUM_WS (ws is folded in place: clobbered) */
int kbner_colsum_rows_f32(float* ws, int rows, int N, float* out, void* stream);
/* the same single-pass fold for up to 64 workspaces of one width in one launch.  items: HOST memory, n records of three 64-bit words:
 * ws, out (device pointers), number of rows.  out[c] += sum_r ws[r, c]; ws is left as it is. */
int kbner_colsum_rows_f32_batched(const long long* items, int n, int N, void* stream);
/* The same launch with DYNAMIC tile scheduling, for steps whose CUs are shared with another kernel (an RCCL collective of the
 * overlapped gradient exchange): a persistent static walk would run the share of every workgroup that finds no CU after all the
 * others have finished -- twice the launch time with 8 CUs held (profiles/round5_cu_contention.txt).  Round 5: launches whose
 * K loops are at least 16 steps long run the ring kernel with ONE workgroup per tile -- the hardware dispatcher hands tiles to
 * CUs as they become free; `sched` is not touched --; shorter ones run the two-stage loop whose workgroups DRAW their tiles from
 * the 8 per-XCD counters in `sched` (device ints the caller zeroed on this stream since their last use).  256-row tiles always.
 * Bit-identical outputs (each output tile is computed the same way whoever computes it).  New capability: the reference has no
 * data-parallel path (flair/trainers/finetune_trainer.py:466,699-700). */
int kbner_gemm_bf16_grouped_dyn(int layout, int nprob, const kbner_gemm_problem* probs, int* sched, void* stream);
/* Which main loops the static launches of the three calls above use (process-wide, atomic; the A/B switch of tools/gemm_pp_lab.py,
 * tools/ab_step.sh, tools/wgrad_lab.py and of tests).  A bit field, default 3; bit-identical outputs in every setting (same MFMA
 * order per accumulator, same epilogue arithmetic):
 *   bit 0  the ring kernels: 256-row tiles on the interleaved ring loop (gemm256f_kernel, round 4: 3 + 2 operand slots, every
 *          fragment read / LDS-DMA piece / cursor operation between two MFMAs), 128-row tiles on the same schedule with three
 *          48-KiB stages and one tile per workgroup (gemm128i_kernel, round 6; see bit 6).  Clear = the two-stage loop of rounds
 *          1-3 for both tile heights and for every dynamic launch.
 *   bit 1  ring, long-K launches (every K >= 16384, at least two tiles per CU -- the grouped weight gradients): the workgroups of
 *          an XCD meet between tiles so that the sharers of an operand panel stay within what their L2 holds (round 5: L2 misses
 *          of that launch 18.5 -> 13.9 GB, -0.5 ms per step).  Assumes the 32 workgroups with equal blockIdx & 7 are co-resident
 *          on one XCD (a 256-workgroup grid on an otherwise idle device); when they are not -- CUs masked, or held by another
 *          stream's kernel -- a meeting times out (bounded, ~0.2 ms per tile boundary; correctness never depends on a meeting).
 *          Dynamic launches (the ones that share CUs with a collective) never meet.
 *   bit 2  + a meeting every 256 K steps inside a tile (11.25 GB = the two-stage loop's traffic exactly; no faster in the step).
 *   bit 3  128-row tiles on the two-stage loop although bit 0 is set (the deep ring's A/B).
 *   bit 4  (round 6, NOT default) a single K = 1024 forward problem with the bias + GELU + GELU' epilogue and at least two 128 x 256
 *          tiles per CU runs on csrc/gemm128x.hip: half-height tiles whose epilogue is spread over the NEXT tile's 16 K steps (the
 *          previous tile's 64 accumulators stay alive beside the current ones).  Bit-identical; measured SLOWER than the 256-row
 *          ring kernel on MI355X (1290-1340 against 1150-1225 us at 256 sentences: profiles/round6_gemm128x_lab.txt), kept as the
 *          A/B it is.
 *   bit 5  (round 6, NOT default) the same launches on csrc/gemm128s.hip: 128 x 256 tiles with WAVE-SPECIALISED epilogues (four MFMA
 *          waves hand the finished tile to four epilogue waves as bf16 through LDS; the pre-activation is rounded to bf16 before GELU, so
 *          results agree with the other kernels to one rounding, not bit for bit).  Also slower on MI355X (1271-1280 us): the loop is
 *          bound by the CU's vector-memory path, which the 128-row tile loads with 1.5 x the operand bytes.
 *   bit 6  128-row tiles on round 5's deep ring (gemm128r_kernel: 4 + 3 slots, A three K steps ahead, plain wait - barrier - burst -
 *          compute loop) instead of gemm128i_kernel: the A/B of the small-batch regime (4 sentences per step: 24.2 / 25.0 against
 *          25.1 / 25.7 us per 256-tile launch inside the step, 14.7 against 17.1 us for the 64-tile o-projection alone).
 *   bits 8-11  lab builds of gemm128x.hip / gemm128s.hip only (-DX128_LAB).  bits 12-15  trace builds only.
 * kbner_gemm_get_variant returns the current value. */
int kbner_gemm_set_variant(int variant);
int kbner_gemm_get_variant(void);
/* Split-K for small micro-batches (a few dozen output tiles, long K): the K range is cut into `splits` problems of ONE grouped
 * launch, each writing an fp32 slab ws[s] (f32[M,N], KBNER_EPI_STORE32); this folds them:
 * C = bf16(dropout(sum_s ws[s] + bias) + addend).  bias / addend nullable, drop_thresh 0 = no dropout. */
int kbner_splitk_finish(const float* ws, int splits, const float* bias, const kbner_bf16* addend, int ldadd, kbner_bf16* C, int ldc,
                        int M, int N, uint32_t drop_seed, uint32_t drop_thresh, void* stream);

/* ---------------- fused self-attention (transformers BertSelfAttention), head_dim 64, S<=512 ---------------- */
/* drop_*: attention-probability dropout, element (i,j) = (bh*S + query, bh*S + key) with bh = b*A + head.
 * ctx_lo (nullable), B*S*H bytes: the forward also stores the rounding residual O - bf16(O), one e5m2 byte per element in a
 * layout private to the two calls, and the backward, given the same buffer, takes the softmax-backward correction
 * D = rowdot(dO, O) from bf16 O + residual instead of the bf16 O alone.  Where the value rows of a head are nearly parallel,
 * dS = P (dP - D) cancels and the 2^-9 rounding of O is amplified (16 % of a query.weight gradient at L = 24 on random
 * weights; 2.6 % with the residual: DESIGN.md section 3). */
int kbner_attn_fwd(const kbner_bf16* qkv, const float* maskbias, kbner_bf16* ctx, uint8_t* ctx_lo, float* lse, int B, int S, int H,
                   int A, uint32_t drop_seed, uint32_t drop_thresh, void* stream);
/* dbias_qkv f32[3H] (nullable): += column sums of dqkv, i.e. the fused QKV projection's bias gradient */
int kbner_attn_bwd(const kbner_bf16* qkv, const kbner_bf16* ctx, const uint8_t* ctx_lo, const kbner_bf16* dctx,
                   const float* maskbias, const float* lse, float* Dws, kbner_bf16* dqkv, int B, int S, int H, int A,
                   uint32_t drop_seed, uint32_t drop_thresh, float* dbias_qkv, void* stream);

/* ---------------- LSTM recurrence (inference): BiLSTM tagger head + FlairEmbeddings character LMs ---------------- */
/* One time step of torch.nn.LSTM's recurrent half for a whole batch and `ndir` directions / models
 * (flair/models/sequence_tagger_model.py:324-357,969-994 `self.rnn`; flair/models/language_model.py:41-44,71-95):
 *   gates = gx[gxi[d][b]] (= Wih x_t + bih + bhh, bf16, gate order i|f|g|o, direction d at column d*4*Hp) + h_in Whh^T
 *   c = sigmoid(f) c + sigmoid(i) tanh(g) ; h_out = sigmoid(o) tanh(c) ; out[outi[d][b], d*out_dir_stride + :] = h_out
 * gxi < 0: the sequence is finished (pack_padded_sequence semantics): state carried unchanged.  outi < 0: h not stored.
 * whh bf16 [ndir, 4Hp, Hp]; h_in / h_out bf16 [ndir, B, Hp]; c f32 [ndir, B, Hp].  Hp % 32 == 0 (zero-pad hidden 1000 -> 1024). */
int kbner_lstm_step(const kbner_bf16* gx, int ld_gx, const int* gxi, const kbner_bf16* whh, const kbner_bf16* h_in,
                    kbner_bf16* h_out, float* c, kbner_bf16* out, int ldo, int out_dir_stride, const int* outi, int B, int Hp,
                    int ndir, void* stream);
/* The whole recurrence in one call: `steps` time steps of `ndir` LSTMs in lockstep, one launch per step enqueued back to back
 * (4 waves per workgroup split K and own every sequence of a batch chunk, so Whh is streamed once per step).  gxi / outi
 * i32 [steps, ndir, B]; out_col i32 [ndir] (device): first column of direction d's h inside `out` -- character LMs whose column
 * blocks are not equidistant run as ONE group; h bf16 [2, ndir, B, Hp] ping-pong (h[0] = h_0, result in h[steps & 1]). */
int kbner_lstm_seq(const kbner_bf16* gx, int ld_gx, const int* gxi, const kbner_bf16* whh, kbner_bf16* h, float* c, kbner_bf16* out,
                   int ldo, const int* out_col, const int* outi, int steps, int B, int Hp, int ndir, void* stream);

/* ---------------- optimiser (transformers==3.0.0 AdamW + clip_grad_norm_, finetune_trainer.py:1010,1018) ------------- */
int kbner_sqnorm_ws_floats(void);
int kbner_grad_sqnorm(const float* g, size_t n, float* ws, float* out, int accumulate, void* stream);
int kbner_adamw_hf(float* p, float* g, float* m, float* v, kbner_bf16* shadow, size_t n, size_t n_shadow, float step_size,
                   float lr_wd, float b1, float b2, float eps, const float* gnorm_sq, float max_norm, float grad_scale,
                   int zero_grad, void* stream);
/* The same update restricted to the rows of an embedding table that have ever received a gradient (flags u8[rows], set by
   kbner_mark_rows from the looked-up ids): an unflagged row has g = m = v = 0, so with weight decay 0 HF AdamW leaves it
   unchanged and it is not read.  (The reference's dense optimizer.step() walks all 250 002 rows of XLM-R's word embedding,
   46 % of the parameters, finetune_trainer.py:1018.)  kbner_grad_sqnorm_rows is the clip norm's share of those rows.
   A flag is two bits: KBNER_ROW_LIVE = the row has ever received a gradient (its moments are non-zero, every step moves it),
   KBNER_ROW_TOUCHED = it has received one since the last zeroing update.  kbner_mark_rows sets both; kbner_adamw_hf_rows with
   zero_grad clears TOUCHED on the rows it zeroes.  A live row that is not touched holds g == 0 exactly: the clip norm skips it and
   the update neither reads nor re-zeroes its gradient (same expression with g = 0: bit-identical to the dense update).  A caller
   that writes a row's gradient by any other route sets both bits itself. */
#define KBNER_ROW_LIVE 1
#define KBNER_ROW_TOUCHED 2
int kbner_mark_rows(const int* ids, int n, unsigned char* flags, int rows, void* stream);
int kbner_grad_sqnorm_rows(const float* g, const unsigned char* flags, int rows, int width, float* ws, float* out, int accumulate,
                           void* stream);
int kbner_adamw_hf_rows(float* p, float* g, float* m, float* v, unsigned char* flags, int rows, int width, float step_size,
                        float b1, float b2, float eps, const float* gnorm_sq, float max_norm, float grad_scale, int zero_grad,
                        void* stream);
/* LAZY rows (round 6).  A live row that receives no gradient in a step is moved by an update that reads nothing but its own p, m, v
   and the step's step_size; the only reader of a row is the embedding lookup of a batch that holds its id.  Instead of streaming
   every live row through HBM in every step (24 B per element: 7 GB for XLM-R's table), kbner_adamw_hf_rows_lazy applies step `t`
   (1-based; = clock[0] + 1) to the TOUCHED rows only -- each first brought up to step t - 1 --, zeroes their gradients, clears
   TOUCHED, then records hist[t & (hist_len - 1)] = step_size and clock[0] = t.  kbner_adamw_rows_catchup applies to the rows
   ids[0..n) (device i32, entries < 0 ignored, repeats allowed; NULL: all rows) the zero-gradient steps (row_t[r], clock[0]] they
   owe: call it on a batch's ids before the lookup, and with NULL before anything else reads the table or its moments.  The k owed
   updates are the eager kernel's k updates -- the same fp32 operations in the same order, in registers: bit-identical state.
   row_t i32[rows] (last step applied; -1 = never live), clock i32[1], hist f32[hist_len] (a power of two; the caller runs the
   all-rows catch-up at least once every hist_len - 1 steps), width <= 1024.  (The reference's optimizer.step() walks every
   parameter in every step: finetune_trainer.py:1018.) */
int kbner_adamw_hf_rows_lazy(float* p, float* g, float* m, float* v, unsigned char* flags, int* row_t, int* clock, float* hist,
                             int hist_len, int t, int rows, int width, float step_size, float b1, float b2, float eps,
                             const float* gnorm_sq, float max_norm, float grad_scale, void* stream);
int kbner_adamw_rows_catchup(const int* ids, int n, float* p, float* m, float* v, const unsigned char* flags, int* row_t,
                             const int* clock, const float* hist, int hist_len, int rows, int width, float b1, float b2, float eps,
                             void* stream);
int kbner_f32_to_bf16(const float* x, kbner_bf16* y, size_t n, void* stream);
int kbner_bf16_to_f32(const kbner_bf16* x, float* y, size_t n, void* stream); /* n % 4 == 0 */
int kbner_wdiff_sum(const float* a, const float* b, const float* w, int n, float* out, void* stream);

/* ---------------- hardware-semantics probes (debug) ---------------- */
int kbner_probe_tr(const uint16_t* in, uint16_t* out, void* stream);
int kbner_probe_mfma(const kbner_bf16* a, const kbner_bf16* b, float* c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KBNER_H */
