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

}  // extern "C"
