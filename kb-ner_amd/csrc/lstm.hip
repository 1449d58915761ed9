// LSTM recurrence for gfx950 (inference): the BiLSTM tagger head of BASELINE config 5 (`use_rnn: true`, hidden_size 1000,
// flair/models/sequence_tagger_model.py:324-357,969-994 -> torch.nn.LSTM) and the character language models behind
// FlairEmbeddings (flair/models/language_model.py:41-44,71-95 -> torch.nn.LSTM(100, 2048)).
//
//   i, f, g, o = split(Wih x_t + bih + Whh h_{t-1} + bhh)        (torch gate order i | f | g | o)
//   c_t = sigmoid(f) * c_{t-1} + sigmoid(i) * tanh(g) ;  h_t = sigmoid(o) * tanh(c_t)
//
// The input half (Wih x_t + biases) has no sequential dependence and is done for all time steps at once by the caller (one
// MFMA GEMM for the tagger head; a [chars, 4H] lookup table for the character LMs).  This kernel is ONE time step of the
// recurrent half for a whole batch and both directions: gates += h_{t-1} Whh^T on the matrix cores, then the cell update in
// registers.  A wavefront owns 16 hidden units x 16 sequences x the 4 gates (four 16x16 accumulators whose lanes hold the same
// (unit, sequence) element, so the cell update needs no data movement); it streams its 64 rows of Whh (bf16, [4H, H]) and the
// 16 h_{t-1} rows straight from L2 as k-contiguous MFMA fragments -- Whh is re-read every step by construction (33 MB for a
// 2048-unit LM, resident in the 256 MB Infinity Cache), there is nothing to stage through LDS that a second wave would share.
// Time steps are separate launches (the step-to-step dependence is a grid-wide one); sequences of different length and the
// backward direction are expressed by per-step row tables: which pre-activation row each sequence consumes at this step
// (-1 = the sequence is finished: state carried unchanged) and where its h_t goes in the output (-1 = not needed).
#include "common.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

static __device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
static __device__ __forceinline__ float tanh_f(float x) {
  // tanh(x) = 1 - 2 / (exp(2x) + 1): exact to fp32 rounding, saturates cleanly at +-1 (exp overflow -> inf -> 1 - 0)
  return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f);
}

// grid (Hp / 16, ceil(B / 16), ndir); block 64.
// gx   bf16 [rows_gx, ld_gx]  : pre-activations Wih x + bih + bhh, direction d at column d * 4 * Hp, gate q at + q * Hp
// gxi  i32  [ndir, B]         : row of gx this sequence consumes at this step, -1 = inactive
// whh  bf16 [ndir, 4 * Hp, Hp]
// h_in / h_out bf16 [ndir, B, Hp] ; c f32 [ndir, B, Hp] (in place)
// out  bf16 [rows_out, ldo]   : h_t written at row outi[d][b] (>= 0), columns d * out_dir_stride + [0, Hp)
__global__ __launch_bounds__(64) void lstm_step_kernel(const bf16_t* __restrict__ gx, int ld_gx, const int* __restrict__ gxi,
                                                      const bf16_t* __restrict__ whh, const bf16_t* __restrict__ h_in,
                                                      bf16_t* __restrict__ h_out, float* __restrict__ c, bf16_t* __restrict__ out,
                                                      int ldo, int out_dir_stride, const int* __restrict__ outi, int B, int Hp) {
  const int lane = threadIdx.x;
  const int li = lane & 15, g = lane >> 4;
  const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16, d = blockIdx.z;
  const int b = b0 + li;                       // the sequence this lane's accumulator column belongs to
  const bool bvalid = b < B;
  const int brow = bvalid ? b : B - 1;         // clamp: rows past B are computed and discarded
  const bf16_t* hrow = h_in + ((size_t)d * B + brow) * Hp + g * 8;
  const bf16_t* wrow = whh + ((size_t)d * 4 * Hp + j0 + li) * Hp + g * 8;   // + q * Hp * Hp for gate q
  const size_t gstride = (size_t)Hp * Hp;
  f4v acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k0 = 0; k0 < Hp; k0 += 32) {
    const bf16x8 hf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s8v*>(hrow + k0));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bf16x8 wf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s8v*>(wrow + q * gstride + k0));
      acc[q] = MFMA16(wf, hf, acc[q]);   // D[unit g*4+r][sequence li]
    }
  }
  if (!bvalid) return;
  const int row = gxi[d * B + b];
  const size_t sidx = ((size_t)d * B + b) * Hp + j0 + g * 4;
  if (row < 0) {   // finished sequence: carry the state
    *reinterpret_cast<uint2*>(h_out + sidx) = *reinterpret_cast<const uint2*>(h_in + sidx);
    return;
  }
  const bf16_t* gp = gx + (size_t)row * ld_gx + (size_t)d * 4 * Hp + j0 + g * 4;
  float pre[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint2 u = *reinterpret_cast<const uint2*>(gp + q * Hp);
    const f2v a = unpack2bf(u.x), bq = unpack2bf(u.y);
    pre[q][0] = acc[q][0] + a[0];
    pre[q][1] = acc[q][1] + a[1];
    pre[q][2] = acc[q][2] + bq[0];
    pre[q][3] = acc[q][3] + bq[1];
  }
  float4 cv = *reinterpret_cast<const float4*>(c + sidx);
  float cc[4] = {cv.x, cv.y, cv.z, cv.w}, hh[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ig = sigmoid_f(pre[0][r]), fg = sigmoid_f(pre[1][r]), gg = tanh_f(pre[2][r]), og = sigmoid_f(pre[3][r]);
    cc[r] = fg * cc[r] + ig * gg;
    hh[r] = og * tanh_f(cc[r]);
  }
  *reinterpret_cast<float4*>(c + sidx) = make_float4(cc[0], cc[1], cc[2], cc[3]);
  uint2 ho;
  ho.x = pack2bf(hh[0], hh[1]);
  ho.y = pack2bf(hh[2], hh[3]);
  *reinterpret_cast<uint2*>(h_out + sidx) = ho;
  const int orow = outi[d * B + b];
  if (orow >= 0) *reinterpret_cast<uint2*>(out + (size_t)orow * ldo + (size_t)d * out_dir_stride + j0 + g * 4) = ho;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The step kernel the sequence driver (kbner_lstm_seq) launches.  Differences from lstm_step_kernel above, all aimed at the
// character-LM case (hidden 2048: Whh is 33.5 MB per model and does not fit any on-chip level, so a step is a stream of Whh
// out of the Infinity Cache / HBM and the kernel's job is to keep enough of that stream in flight):
//   * a workgroup is 4 waves that split K (wave w takes the 64-wide k slices w, w + 4, ...: whole 128-byte lines of every Whh
//     row) and owns ALL sequences of its batch chunk (NT tiles of 16), so every Whh byte is read once per step instead of once
//     per 16 sequences, and the dependent-load chain per wave is K / 256 iterations instead of K / 32;
//   * partial gate sums meet in LDS (4 waves x 4 gates x NT tiles x 1 KB), wave `nt` finishes tile nt (cell update as above);
//   * the output column of every direction / model comes from a table, so character LMs whose column blocks are not
//     equidistant in the concatenated feature matrix run as ONE group (one launch per time step for all of them).
// Why not a persistent kernel with Whh slices stationary in LDS and a grid barrier per step (2048-unit LM: 131 KB per CU):
// the guide's price list (MI355X_MICROARCH.md, rows barrier-xcd / boundary / handoff-payload) puts a 256-WG grid barrier at
// 4.1-4.7 us and the re-read of the 131 KB h_t by every CU at >= 1.3 us, against 1.5-1.9 us for a dependent kernel boundary; only
// one 2048-unit model fits the chip's LDS at a time, so the four LMs would run back to back at ~7 us per step = 4 ms, no better
// than streaming all four through one launch per step (measured in profiles/round3_cfg5_stack_bench.jsonl).
template <int NT>
__global__ __launch_bounds__(256) void lstm_seq_step_kernel(const bf16_t* __restrict__ gx, int ld_gx, const int* __restrict__ gxi,
                                                           const bf16_t* __restrict__ whh, const bf16_t* __restrict__ h_in,
                                                           bf16_t* __restrict__ h_out, float* __restrict__ c,
                                                           bf16_t* __restrict__ out, int ldo, const int* __restrict__ out_col,
                                                           const int* __restrict__ outi, int B, int Hp) {
  __shared__ f4v red[4 * 4 * NT * 64];
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int j0 = blockIdx.x * 16, d = blockIdx.y, b0 = blockIdx.z * (16 * NT);
  const bf16_t* wrow = whh + ((size_t)d * 4 * Hp + j0 + li) * Hp + g * 8;   // + q * Hp * Hp for gate q
  const size_t gstride = (size_t)Hp * Hp;
  const bf16_t* hrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int b = b0 + nt * 16 + li;
    hrow[nt] = h_in + ((size_t)d * B + (b < B ? b : B - 1)) * Hp + g * 8;
  }
  f4v acc[4][NT];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[q][nt] = (f4v){0.f, 0.f, 0.f, 0.f};
  // software pipeline: the fragments of k slice i + 1 are requested before the MFMAs of slice i are issued, so every wave keeps
  // 8 Whh + 2 NT h loads (16 B per lane each) in flight across its dependent chain of Hp / 256 slices
  s8v wf[2][4], hf[2][NT];
  auto load_slice = [&](int kk, s8v (&wv)[2][4], s8v (&hv)[2][NT]) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) hv[half][nt] = *reinterpret_cast<const s8v*>(hrow[nt] + kk + half * 32);
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[half][q] = *reinterpret_cast<const s8v*>(wrow + q * gstride + kk + half * 32);
    }
  };
  int kk = w * 64;
  if (kk < Hp) load_slice(kk, wf, hf);
  for (; kk < Hp; kk += 256) {
    s8v wn[2][4], hn[2][NT];
    // (the last slice re-requests itself instead of branching: the loads stay unconditional and ahead of the MFMAs)
    load_slice(kk + 256 < Hp ? kk + 256 : kk, wn, hn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)   // D[unit g*4+r][sequence li of tile nt]
          acc[q][nt] = MFMA16(__builtin_bit_cast(bf16x8, wf[half][q]), __builtin_bit_cast(bf16x8, hf[half][nt]), acc[q][nt]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int q = 0; q < 4; ++q) wf[half][q] = wn[half][q];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) hf[half][nt] = hn[half][nt];
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) red[((w * 4 + q) * NT + nt) * 64 + lane] = acc[q][nt];
  __syncthreads();
  if (w >= NT) return;
  const int nt = w;
  const int b = b0 + nt * 16 + li;
  if (b >= B) return;
  const int row = gxi[d * B + b];
  const size_t sidx = ((size_t)d * B + b) * Hp + j0 + g * 4;
  if (row < 0) {   // finished sequence: carry the state
    *reinterpret_cast<uint2*>(h_out + sidx) = *reinterpret_cast<const uint2*>(h_in + sidx);
    return;
  }
  const bf16_t* gp = gx + (size_t)row * ld_gx + (size_t)d * 4 * Hp + j0 + g * 4;
  float pre[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f4v sum = red[((0 * 4 + q) * NT + nt) * 64 + lane];
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) {
      const f4v p = red[((ww * 4 + q) * NT + nt) * 64 + lane];
      sum[0] += p[0];
      sum[1] += p[1];
      sum[2] += p[2];
      sum[3] += p[3];
    }
    const uint2 u = *reinterpret_cast<const uint2*>(gp + q * Hp);
    const f2v a = unpack2bf(u.x), bq = unpack2bf(u.y);
    pre[q][0] = sum[0] + a[0];
    pre[q][1] = sum[1] + a[1];
    pre[q][2] = sum[2] + bq[0];
    pre[q][3] = sum[3] + bq[1];
  }
  float4 cv = *reinterpret_cast<const float4*>(c + sidx);
  float cc[4] = {cv.x, cv.y, cv.z, cv.w}, hh[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ig = sigmoid_f(pre[0][r]), fg = sigmoid_f(pre[1][r]), gg = tanh_f(pre[2][r]), og = sigmoid_f(pre[3][r]);
    cc[r] = fg * cc[r] + ig * gg;
    hh[r] = og * tanh_f(cc[r]);
  }
  *reinterpret_cast<float4*>(c + sidx) = make_float4(cc[0], cc[1], cc[2], cc[3]);
  uint2 ho;
  ho.x = pack2bf(hh[0], hh[1]);
  ho.y = pack2bf(hh[2], hh[3]);
  *reinterpret_cast<uint2*>(h_out + sidx) = ho;
  const int orow = outi[d * B + b];
  if (orow >= 0) *reinterpret_cast<uint2*>(out + (size_t)orow * ldo + out_col[d] + j0 + g * 4) = ho;
}

template <int NT>
static void lstm_seq_launch(const bf16_t* gx, int ld_gx, const int* gxi, const bf16_t* whh, const bf16_t* h_in, bf16_t* h_out,
                            float* c, bf16_t* out, int ldo, const int* out_col, const int* outi, int B, int Hp, int ndir,
                            hipStream_t stream) {
  hipLaunchKernelGGL(lstm_seq_step_kernel<NT>, dim3(Hp / 16, ndir, (B + 16 * NT - 1) / (16 * NT)), dim3(256), 0, stream, gx, ld_gx,
                     gxi, whh, h_in, h_out, c, out, ldo, out_col, outi, B, Hp);
}

extern "C" {

// One LSTM time step for `ndir` independent directions / models that share the batch size and hidden width.
// Constraints: Hp % 32 == 0, ld_gx % 4 == 0, ldo % 4 == 0, out_dir_stride % 4 == 0; hidden sizes that are not a multiple of 32
// are zero-padded by the caller (a padded unit has zero weights and biases: its gates are 0.5 / 0 and its c and h stay 0).
int kbner_lstm_step(const bf16_t* gx, int ld_gx, const int* gxi, const bf16_t* whh, const bf16_t* h_in, bf16_t* h_out, float* c,
                    bf16_t* out, int ldo, int out_dir_stride, const int* outi, int B, int Hp, int ndir, void* stream) {
  KBNER_CHECK_ARG(gx != nullptr && gxi != nullptr && whh != nullptr && h_in != nullptr && h_out != nullptr && c != nullptr &&
                  out != nullptr && outi != nullptr);
  KBNER_CHECK_ARG(B > 0 && ndir > 0 && Hp > 0 && Hp % 32 == 0 && ld_gx % 4 == 0 && ldo % 4 == 0 && out_dir_stride % 4 == 0);
  hipLaunchKernelGGL(lstm_step_kernel, dim3(Hp / 16, (B + 15) / 16, ndir), dim3(64), 0, (hipStream_t)stream, gx, ld_gx, gxi, whh,
                     h_in, h_out, c, out, ldo, out_dir_stride, outi, B, Hp);
  KBNER_LAUNCH_RET();
}

// The whole recurrence of `ndir` single-layer LSTMs in lockstep (the two directions of the tagger's BiLSTM; all character LMs of
// one hidden width): `steps` time steps, one launch each, enqueued back to back by this one call.
//   gx  bf16 [rows_gx, ld_gx]: pre-activations, direction d at column d * 4 * Hp (gate q at + q * Hp)
//   gxi / outi i32 [steps, ndir, B]: per step the row of gx each sequence consumes (-1 = finished: state carried) and the row of
//        `out` that receives h_t (-1 = not needed);  out_col i32 [ndir] (device): first column of direction d's h inside `out`
//   h   bf16 [2, ndir, B, Hp]: ping-pong state, h[0] = h_0 on entry; the final state is h[steps & 1];  c f32 [ndir, B, Hp] in place
// Constraints as kbner_lstm_step, and Hp % 64 == 0 (a wave's k slice is a whole 128-byte line of every Whh row); out_col[d] % 4 == 0.
int kbner_lstm_seq(const bf16_t* gx, int ld_gx, const int* gxi, const bf16_t* whh, bf16_t* h, float* c, bf16_t* out, int ldo,
                   const int* out_col, const int* outi, int steps, int B, int Hp, int ndir, void* stream) {
  KBNER_CHECK_ARG(gx != nullptr && gxi != nullptr && whh != nullptr && h != nullptr && c != nullptr && out != nullptr &&
                  out_col != nullptr && outi != nullptr);
  KBNER_CHECK_ARG(steps >= 0 && B > 0 && ndir > 0 && Hp > 0 && Hp % 64 == 0 && ld_gx % 4 == 0 && ldo % 4 == 0);
  const size_t hs = (size_t)ndir * B * Hp;
  const int per = ndir * B;
  for (int s = 0; s < steps; ++s) {
    const bf16_t* h_in = h + (size_t)(s & 1) * hs;
    bf16_t* h_out = h + (size_t)((s + 1) & 1) * hs;
    const int* gi = gxi + (size_t)s * per;
    const int* oi = outi + (size_t)s * per;
    if (B <= 16)
      lstm_seq_launch<1>(gx, ld_gx, gi, whh, h_in, h_out, c, out, ldo, out_col, oi, B, Hp, ndir, (hipStream_t)stream);
    else if (B <= 32)
      lstm_seq_launch<2>(gx, ld_gx, gi, whh, h_in, h_out, c, out, ldo, out_col, oi, B, Hp, ndir, (hipStream_t)stream);
    else if (B <= 48)
      lstm_seq_launch<3>(gx, ld_gx, gi, whh, h_in, h_out, c, out, ldo, out_col, oi, B, Hp, ndir, (hipStream_t)stream);
    else
      lstm_seq_launch<4>(gx, ld_gx, gi, whh, h_in, h_out, c, out, ldo, out_col, oi, B, Hp, ndir, (hipStream_t)stream);
  }
  KBNER_LAUNCH_RET();
}

}  // extern "C"
