// Fused multi-head self-attention for gfx950 (CDNA4), head_dim 64, S <= 512 (XLM-R's window),
// forward + backward, bf16 MFMA with fp32 softmax statistics.
//
//   P = softmax(Q K^T / sqrt(d) + maskbias[key]) ; O = P V            (maskbias = (1-mask) * -10000,
//   transformers 3.0.0 BertSelfAttention, reached from flair/embeddings.py:3269)
//
// Design (MI355X-first, not a flash-attention port): XLM-R never exceeds 512 positions, so the two
// [S,64] operand panels a workgroup needs (64 KiB each at S=512) fit the CU's 160 KiB LDS whole.
// Every kernel keeps a 16-row "stationary" fragment set in registers per wave and streams the two
// LDS-resident panels past it: no online-softmax rescaling, no K/V re-fetch per tile, one DMA
// (global_load_lds, 16 B/lane, XOR-swizzled on the source address) of each panel per workgroup.
// Products are computed transposed where that makes the softmax axis lane-local (S^T = K Q^T) so
// the probabilities feed the next MFMA as a B operand straight from registers; k-strided operands
// (V^T, K^T, Q^T, dO^T) come from the same row-major panels via ds_read_b64_tr_b16.
//
//   attn_fwd     grid (S/128, A, B): panels K,V ; stationary Q        -> O [M,H], lse [B,A,S]
//   attn_bwd_dq  grid (S/128, A, B): panels K,V ; stationary Q,dO     -> dQ
//   attn_bwd_dkv grid (S/128, A, B): panels Q,dO; stationary K,V      -> dK, dV
// (backward recomputes S and dP once per kernel: 7 matmuls instead of 5, no atomics, no LDS
// round trip of computed tiles.)  qkv / dqkv are token-major [M, 3H] (Q | K | V), head h at
// column h*64, exactly what the fused QKV GEMM produces / consumes: no permute kernels.
#include "common.h"
#include <cstdlib>

#include "attn_common.h"

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// NKB = S / 16 is a template parameter: with a compile-time key count the whole 16-row pass is straight-line code
// (no per-fragment branches), so hipcc interleaves the K-panel ds_reads, the 2*NKB QK^T MFMAs, the softmax VALU work
// and the V transpose-reads instead of serialising them block by block.  LDS addresses are one per-lane base per
// (k-step | d-block) plus compile-time offsets (the XOR swizzle of a row depends only on the lane, not on the tile).
// DROP: attention-probability dropout (BertSelfAttention.dropout).  Element (query q, key k) of head (b,h) is kept iff
// drop_keep(rowkey(bh*S + q), colkey(bh*S + k)) (common.h); the normaliser `sum` / the stored lse are those of the
// UNdropped softmax, the kept probabilities are scaled by 1/(1-p) through `inv`.
// PERSIST (whole heads per workgroup, many heads per CU): grid.x workgroups walk the (batch, head) items w, w + grid.x, ... and
// the NEXT head's panels are DMA'd while the current head is still being computed -- K as soon as every wave has finished the
// last pass's Q.K^T (one barrier), V when the last P.V is done -- so the ~2.3 us a head's 128 KiB of panels take from L2 are no
// longer exposed once per head with a single workgroup per CU.
template <int NKB, bool DROP, bool PERSIST = false>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ maskbias,
                                                          bf16_t* __restrict__ ctx, uint8_t* __restrict__ ctx_lo, float* __restrict__ lse, int H,
                                                          int A, float scale, int rpw, uint32_t drop_seed, uint32_t drop_thresh,
                                                          int nitems) {
  constexpr int S = NKB * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = smem + AT_MAXS * 128;
  float* sMask = reinterpret_cast<float*>(smem + 2 * AT_MAXS * 128);
  uint32_t* sCk = reinterpret_cast<uint32_t*>(sMask + AT_MAXS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qt = PERSIST ? 0 : blockIdx.x;
  const int ld = 3 * H;
  const float scale2 = scale * 1.4426950408889634f;
  const float dscale = DROP ? drop_scale(drop_thresh) : 1.0f;
  // sMask holds maskbias / scale: it is added to the RAW score sums q.k (the softmax runs on scale2 * (q.k + mask / scale)), and
  // only in 16-key fragments at or behind the first masked key (`nfree16` leading fragments are mask-free: for the prefix masks
  // of padded batches that is every fragment but the tail, for full-length sentences all of them)
  __shared__ int s_first_masked[2];   // two slots: the next item's is filled while the current one's is still read
  // mask / dropout column keys of item (b_, h_) -> sMask, sCk, s_first_masked[slot_] (which thread 0 has reset to S beforehand)
  auto stage_mask = [&](int b_, int h_, int slot_) {
    int fm = S;
    const uint32_t bhS_ = (uint32_t)((b_ * A + h_) * S);
    for (int i = tid; i < S; i += 512) {
      const float m = maskbias[(size_t)b_ * S + i];
      sMask[i] = m * (1.0f / scale);
      if (m != 0.0f && i < fm) fm = i;
      if (DROP) sCk[i] = drop_colkey(drop_seed, bhS_ + (uint32_t)i);
    }
    if (fm < S) atomicMin(&s_first_masked[slot_], fm);
  };
  int item = PERSIST ? (int)blockIdx.x : 0;
  int h = PERSIST ? item % A : (int)blockIdx.y, b = PERSIST ? item / A : (int)blockIdx.z;
  int slot = 0;
  if (tid == 0) s_first_masked[0] = s_first_masked[1] = S;
  __syncthreads();
  stage_mask(b, h, 0);
  stage_panel(qkv + (size_t)b * S * ld + h * AT_D + H, ld, S, sK, wid, lane);
  stage_panel(qkv + (size_t)b * S * ld + h * AT_D + 2 * H, ld, S, sV, wid, lane);
  const int g = lane >> 4, li = lane & 15;
  // per-lane LDS bases (see kc_frag / tr_frag): K rows f*16 + li -> + f*2048 ; V rows kc*32 + g*4 + (li>>2) -> + kc*4096
  const unsigned char* kb0 = sK + li * 128 + (((0 * 4 + g) ^ kc_swz(li)) << 4);
  const unsigned char* kb1 = sK + li * 128 + (((1 * 4 + g) ^ kc_swz(li)) << 4);
  const int vrow = g * 4 + (li >> 2);
  const unsigned char* vb[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
    vb[db] = sV + vrow * 128 + (((db * 2 + ((li & 3) >> 1)) ^ kc_swz(vrow)) << 4) + ((li & 1) << 3);
  for (;;) {   // items of this workgroup (one iteration unless PERSIST)
  const bf16_t* base = qkv + (size_t)b * S * ld + h * AT_D;
  const uint32_t bhS = (uint32_t)((b * A + h) * S);
  const int next = item + (int)gridDim.x;
  const bool has_next = PERSIST && next < nitems;
  // K has landed when at most the V pieces (S/64 per wave, issued after K) are still in flight
  wait_vm(S / 64);
  __syncthreads();
  const int nfree16 = s_first_masked[slot] >> 4;   // fragments f < nfree16 hold no masked key
  if (has_next && tid == 0) s_first_masked[slot ^ 1] = S;   // (published by the barrier before the next item's mask is staged)
  bool v_ready = false, k_next = false;
#pragma unroll 1
  for (int pass = 0; pass < rpw / 128; ++pass) {
    const int q0 = qt * rpw + wid * (rpw / 8) + pass * 16;
    const bool active = q0 < S;
    f4v st[NKB];
    float mx = -INFINITY, sum = 0.0f;
    if (active) {
      const bf16x8 qf0 = glb_frag(base, ld, q0, 0, lane);
      const bf16x8 qf1 = glb_frag(base, ld, q0, 1, lane);
      // two straight-line copies of the score pass (a per-fragment test would cut the unrolled body into 32 basic blocks):
      // full-length sentences -- no masked key anywhere -- skip the mask add altogether
      if (nfree16 >= NKB) {
#pragma unroll
        for (int f = 0; f < NKB; ++f) {
          f4v a = (f4v){0.f, 0.f, 0.f, 0.f};
          a = MFMA(__builtin_bit_cast(bf16x8, *reinterpret_cast<const s8v*>(kb0 + f * 2048)), qf0, a);
          a = MFMA(__builtin_bit_cast(bf16x8, *reinterpret_cast<const s8v*>(kb1 + f * 2048)), qf1, a);
          // the row maximum is taken on the RAW sums (scale2 > 0 commutes with max): two v_max3 per fragment, no multiply
          mx = fmaxf(fmaxf(mx, a[0]), a[1]);
          mx = fmaxf(fmaxf(mx, a[2]), a[3]);
          st[f] = a;
        }
      } else {
#pragma unroll
        for (int f = 0; f < NKB; ++f) {
          f4v a = *reinterpret_cast<const f4v*>(sMask + f * 16 + g * 4);   // accumulators start at mask / scale
          a = MFMA(__builtin_bit_cast(bf16x8, *reinterpret_cast<const s8v*>(kb0 + f * 2048)), qf0, a);
          a = MFMA(__builtin_bit_cast(bf16x8, *reinterpret_cast<const s8v*>(kb1 + f * 2048)), qf1, a);
          mx = fmaxf(fmaxf(mx, a[0]), a[1]);
          mx = fmaxf(fmaxf(mx, a[2]), a[3]);
          st[f] = a;
        }
      }
      mx = group4_max(mx);
      const float nmx = -mx * scale2;            // exp2(scale2 * s - scale2 * max): one fma + one v_exp per probability
      mx = mx * scale2;                          // log2-domain maximum, what the stored lse needs
      const uint32_t rk = DROP ? drop_rowkey(drop_seed, bhS + (uint32_t)(q0 + li)) : 0u;
#pragma unroll
      for (int f = 0; f < NKB; ++f) {
        f4v a = st[f];
        a[0] = __builtin_amdgcn_exp2f(a[0] * scale2 + nmx);
        a[1] = __builtin_amdgcn_exp2f(a[1] * scale2 + nmx);
        a[2] = __builtin_amdgcn_exp2f(a[2] * scale2 + nmx);
        a[3] = __builtin_amdgcn_exp2f(a[3] * scale2 + nmx);
        if (DROP) {
          sum += (a[0] + a[1]) + (a[2] + a[3]);  // normaliser of the UNdropped softmax
          const uint4 ck = *reinterpret_cast<const uint4*>(sCk + f * 16 + g * 4);
          a[0] = drop_keep(rk, ck.x, drop_thresh) ? a[0] : 0.0f;
          a[1] = drop_keep(rk, ck.y, drop_thresh) ? a[1] : 0.0f;
          a[2] = drop_keep(rk, ck.z, drop_thresh) ? a[2] : 0.0f;
          a[3] = drop_keep(rk, ck.w, drop_thresh) ? a[3] : 0.0f;
        }
        st[f] = a;
      }
      if (DROP) sum = group4_sum(sum);
    }
    if (has_next && pass == rpw / 128 - 1) {
      // every wave is past the last Q.K^T / softmax of this head: sK, sMask and sCk are dead -> the next head's mask and K panel
      __syncthreads();
      const int hn = next % A, bn = next / A;
      stage_mask(bn, hn, slot ^ 1);
      stage_panel(qkv + (size_t)bn * S * ld + hn * AT_D + H, ld, S, sK, wid, lane);
      k_next = true;
    }
    if (!v_ready) {  // first pass only (uniform): V is needed from here on
      wait_vm(k_next ? S / 64 : 0);
      __syncthreads();
      v_ready = true;
    }
    if (active) {
      // The softmax VALU work bounds this kernel, so two per-probability operations are moved off the vector pipe: the
      // UNnormalised exponentials (all <= 1) go to the P.V MFMAs and the 16 outputs are scaled by 1/sum instead of the
      // 4*NKB probabilities, and (without dropout) the row sums come from one extra MFMA per key chunk against an all-ones
      // A fragment -- the sum of exactly the bf16 probabilities that multiply V.
      f4v o[4], osum = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] = (f4v){0.f, 0.f, 0.f, 0.f};
      const s8v ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
      const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
#pragma unroll
      for (int kc = 0; kc < NKB / 2; ++kc) {
        const bf16x8 pb = pack_b(st[2 * kc], st[2 * kc + 1]);
        if (!DROP) osum = MFMA(ones, pb, osum);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const unsigned char* a = vb[db] + kc * 4096;
          const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a));
          const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a + 16 * 128));
          s8v v;
          v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
          v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
          o[db] = MFMA(__builtin_bit_cast(bf16x8, v), pb, o[db]);
        }
      }
      // O^T fragment: lane holds O[q0+li][db*16 + g*4 .. +3]
      if (!DROP) sum = osum[0];  // every row of ones.P^T holds the column (= query) sums
      const float inv = dscale / sum;
      bf16_t* orow = ctx + (size_t)(b * S + q0 + li) * H + h * AT_D;
      uint32_t res[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        uint2 u;
        u.x = pack2bf_res8(o[db][0] * inv, o[db][1] * inv, res[db], false);
        u.y = pack2bf_res8(o[db][2] * inv, o[db][3] * inv, res[db], true);
        *reinterpret_cast<uint2*>(orow + db * 16 + g * 4) = u;
      }
      if (ctx_lo)   // O - bf16(O), one byte per element, 16 B per lane and 16-row block (at_res_block): see kbner_attn_bwd
        *reinterpret_cast<uint4*>(at_res_block(ctx_lo, b, A, h, S, q0) + lane * 16) = make_uint4(res[0], res[1], res[2], res[3]);
      if (g == 0) lse[((size_t)b * A + h) * S + q0 + li] = (mx + __log2f(sum)) * 0.6931471805599453f;
    }
  }
  if (!has_next) break;
  __syncthreads();   // every wave is done with sV: the next head's V panel
  item = next;
  h = item % A;
  b = item / A;
  slot ^= 1;
  stage_panel(qkv + (size_t)b * S * ld + h * AT_D + 2 * H, ld, S, sV, wid, lane);
  }  // items
}

// Round 3: the same panel-resident forward with 32 query rows per wave and pass.  With 16-row passes every K / V fragment read
// out of LDS feeds ONE MFMA, and a pass moves both 64-KiB panels through the LDS pipe for 16 rows: 4 MB per head, ~260 K cycles
// of LDS bandwidth per CU and B = 128 launch next to ~280 K cycles of MFMA + VALU issue per SIMD -- the kernel is co-limited by
// the two.  Two 16-row blocks per wave share every fragment read (half the LDS bytes per flop, as the
// backward kernels do since round 2).  The scores of 32 rows x 512 keys do not fit the register file, so a pass runs the keys in
// TWO halves -- scores, exponentials and P.V of keys [0, S/2), then of [S/2, S) -- joined by ONE online-softmax step per row
// (outputs and row sums of the first half rescaled by exp2(m_old - m_new)): not a streaming softmax, a two-block one.
// Everything else (accumulators started at mask / scale, ones-MFMA row sums, unnormalised P into the P.V MFMAs, persistent walk
// over heads with the next head's K panel requested after the last Q.K^T and its V panel after the last P.V) is the 16-row
// kernel's.  Needs rpw % 256 == 0 (every wave owns whole 32-row passes) and NKB % 4 == 0.
template <int NKB, bool DROP, bool PERSIST = false, int NP = 2>
__global__ __launch_bounds__(512) void attn_fwd32_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ maskbias,
                                                         bf16_t* __restrict__ ctx, uint8_t* __restrict__ ctx_lo, float* __restrict__ lse, int H, int A,
                                                         float scale,
                                                         int rpw, uint32_t drop_seed, uint32_t drop_thresh, int nitems) {
  constexpr int S = NKB * 16;
  constexpr int NH = NKB / NP;  // 16-key fragments per part (NP parts of the key axis, joined by NP - 1 online-softmax steps per row)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = smem + AT_MAXS * 128;
  float* sMask = reinterpret_cast<float*>(smem + 2 * AT_MAXS * 128);
  uint32_t* sCk = reinterpret_cast<uint32_t*>(sMask + AT_MAXS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qt = PERSIST ? 0 : blockIdx.x;
  const int ld = 3 * H;
  const float scale2 = scale * 1.4426950408889634f;
  const float dscale = DROP ? drop_scale(drop_thresh) : 1.0f;
  __shared__ int s_first_masked[2];
  auto stage_mask = [&](int b_, int h_, int slot_) {
    int fm = S;
    const uint32_t bhS_ = (uint32_t)((b_ * A + h_) * S);
    for (int i = tid; i < S; i += 512) {
      const float m = maskbias[(size_t)b_ * S + i];
      sMask[i] = m * (1.0f / scale);
      if (m != 0.0f && i < fm) fm = i;
      if (DROP) sCk[i] = drop_colkey(drop_seed, bhS_ + (uint32_t)i);
    }
    if (fm < S) atomicMin(&s_first_masked[slot_], fm);
  };
  int item = PERSIST ? (int)blockIdx.x : 0;
  int h = PERSIST ? item % A : (int)blockIdx.y, b = PERSIST ? item / A : (int)blockIdx.z;
  int slot = 0;
  if (tid == 0) s_first_masked[0] = s_first_masked[1] = S;
  __syncthreads();
  stage_mask(b, h, 0);
  stage_panel(qkv + (size_t)b * S * ld + h * AT_D + H, ld, S, sK, wid, lane);
  stage_panel(qkv + (size_t)b * S * ld + h * AT_D + 2 * H, ld, S, sV, wid, lane);
  const int g = lane >> 4, li = lane & 15;
  const unsigned char* kb0 = sK + li * 128 + (((0 * 4 + g) ^ kc_swz(li)) << 4);
  const unsigned char* kb1 = sK + li * 128 + (((1 * 4 + g) ^ kc_swz(li)) << 4);
  const int vrow = g * 4 + (li >> 2);
  const unsigned char* vb[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
    vb[db] = sV + vrow * 128 + (((db * 2 + ((li & 3) >> 1)) ^ kc_swz(vrow)) << 4) + ((li & 1) << 3);
  const s8v ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
  for (;;) {   // items of this workgroup (one iteration unless PERSIST)
  const bf16_t* base = qkv + (size_t)b * S * ld + h * AT_D;
  const uint32_t bhS = (uint32_t)((b * A + h) * S);
  const int next = item + (int)gridDim.x;
  const bool has_next = PERSIST && next < nitems;
  wait_vm(S / 64);   // K has landed when at most the V pieces are still in flight
  __syncthreads();
  const int nfree16 = s_first_masked[slot] >> 4;
  if (has_next && tid == 0) s_first_masked[slot ^ 1] = S;
  bool v_ready = false, k_next = false;
  const int npass = rpw / 256;
#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
    const int q0 = qt * rpw + wid * (rpw / 8) + pass * 32;
    const bool active = q0 < S;
    bf16x8 qf[2][2];
    uint32_t rk[2] = {0u, 0u};
    float m[2] = {-INFINITY, -INFINITY}, sum[2] = {0.0f, 0.0f};
    f4v o[2][4], osum[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      osum[j] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int db = 0; db < 4; ++db) o[j][db] = (f4v){0.f, 0.f, 0.f, 0.f};
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        qf[j][0] = glb_frag(base, ld, q0 + j * 16, 0, lane);
        qf[j][1] = glb_frag(base, ld, q0 + j * 16, 1, lane);
        if (DROP) rk[j] = drop_rowkey(drop_seed, bhS + (uint32_t)(q0 + j * 16 + li));
      }
    }
#pragma unroll
    for (int half = 0; half < NP; ++half) {   // (`half`: the part index; NP = 2 in the first version)
      f4v st[2][NH];
      bf16x8 pbuf[2][NH / 2];
      if (active) {
        float mx[2] = {-INFINITY, -INFINITY};
        const bool plain = nfree16 >= (half + 1) * NH;   // no masked key in this half (wave-uniform)
        if (plain) {
#pragma unroll
          for (int f = 0; f < NH; ++f) {
            const bf16x8 k0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s8v*>(kb0 + (half * NH + f) * 2048));
            const bf16x8 k1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s8v*>(kb1 + (half * NH + f) * 2048));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              f4v a = (f4v){0.f, 0.f, 0.f, 0.f};
              a = MFMA(k0, qf[j][0], a);
              a = MFMA(k1, qf[j][1], a);
              mx[j] = fmaxf(fmaxf(mx[j], a[0]), a[1]);
              mx[j] = fmaxf(fmaxf(mx[j], a[2]), a[3]);
              st[j][f] = a;
            }
          }
        } else {
#pragma unroll
          for (int f = 0; f < NH; ++f) {
            const bf16x8 k0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s8v*>(kb0 + (half * NH + f) * 2048));
            const bf16x8 k1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s8v*>(kb1 + (half * NH + f) * 2048));
            const f4v mk = *reinterpret_cast<const f4v*>(sMask + (half * NH + f) * 16 + g * 4);   // mask / scale
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              f4v a = MFMA(k0, qf[j][0], mk);
              a = MFMA(k1, qf[j][1], a);
              mx[j] = fmaxf(fmaxf(mx[j], a[0]), a[1]);
              mx[j] = fmaxf(fmaxf(mx[j], a[2]), a[3]);
              st[j][f] = a;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float mh = group4_max(mx[j]);
          const float mn = fmaxf(m[j], mh);
          if (half > 0) {   // online-softmax step: the earlier parts' outputs and sums to the new reference
            const float alpha = __builtin_amdgcn_exp2f((m[j] - mn) * scale2);
#pragma unroll
            for (int db = 0; db < 4; ++db) o[j][db] *= alpha;
            osum[j] *= alpha;
            sum[j] *= alpha;
          }
          m[j] = mn;
          const float nmx = -mn * scale2;
          // exponentials are packed to the bf16 B fragments of P.V as they are produced: the fp32 score registers die here,
          // two per packed fragment (peak 128 -> 64 live registers for the probabilities of a half)
#pragma unroll
          for (int kc = 0; kc < NH / 2; ++kc) {
            f4v e[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int f = 2 * kc + t;
              f4v a = st[j][f];
              a[0] = __builtin_amdgcn_exp2f(a[0] * scale2 + nmx);
              a[1] = __builtin_amdgcn_exp2f(a[1] * scale2 + nmx);
              a[2] = __builtin_amdgcn_exp2f(a[2] * scale2 + nmx);
              a[3] = __builtin_amdgcn_exp2f(a[3] * scale2 + nmx);
              if (DROP) {
                sum[j] += (a[0] + a[1]) + (a[2] + a[3]);   // normaliser of the UNdropped softmax (reduced over g at the end)
                const uint4 ck = *reinterpret_cast<const uint4*>(sCk + (half * NH + f) * 16 + g * 4);
                a[0] = drop_keep(rk[j], ck.x, drop_thresh) ? a[0] : 0.0f;
                a[1] = drop_keep(rk[j], ck.y, drop_thresh) ? a[1] : 0.0f;
                a[2] = drop_keep(rk[j], ck.z, drop_thresh) ? a[2] : 0.0f;
                a[3] = drop_keep(rk[j], ck.w, drop_thresh) ? a[3] : 0.0f;
              }
              e[t] = a;
            }
            pbuf[j][kc] = pack_b(e[0], e[1]);
          }
        }
      }
      if (half == NP - 1 && has_next && pass == npass - 1) {
        // every wave is past the last Q.K^T / softmax of this head: sK, sMask and sCk are dead -> the next head's mask and K panel
        __syncthreads();
        const int hn = next % A, bn = next / A;
        stage_mask(bn, hn, slot ^ 1);
        stage_panel(qkv + (size_t)bn * S * ld + hn * AT_D + H, ld, S, sK, wid, lane);
        k_next = true;
      }
      if (!v_ready) {  // first P.V of the item (uniform): V is needed from here on
        wait_vm(0);
        __syncthreads();
        v_ready = true;
      }
      if (active) {
#pragma unroll
        for (int kc = 0; kc < NH / 2; ++kc) {
          const bf16x8 pb0 = pbuf[0][kc];
          const bf16x8 pb1 = pbuf[1][kc];
          if (!DROP) {
            osum[0] = MFMA(ones, pb0, osum[0]);
            osum[1] = MFMA(ones, pb1, osum[1]);
          }
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            const unsigned char* a = vb[db] + (half * (NH / 2) + kc) * 4096;
            const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a));
            const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a + 16 * 128));
            s8v v;
            v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
            v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
            const bf16x8 vf = __builtin_bit_cast(bf16x8, v);
            o[0][db] = MFMA(vf, pb0, o[0][db]);
            o[1][db] = MFMA(vf, pb1, o[1][db]);
          }
        }
      }
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float sm = DROP ? group4_sum(sum[j]) : osum[j][0];
        const float inv = dscale / sm;
        bf16_t* orow = ctx + (size_t)(b * S + q0 + j * 16 + li) * H + h * AT_D;
        uint32_t res[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          uint2 u;
          u.x = pack2bf_res8(o[j][db][0] * inv, o[j][db][1] * inv, res[db], false);
          u.y = pack2bf_res8(o[j][db][2] * inv, o[j][db][3] * inv, res[db], true);
          *reinterpret_cast<uint2*>(orow + db * 16 + g * 4) = u;
        }
        if (ctx_lo)   // O - bf16(O), one byte per element, 16 B per lane and 16-row block (at_res_block): see kbner_attn_bwd
          *reinterpret_cast<uint4*>(at_res_block(ctx_lo, b, A, h, S, q0 + j * 16) + lane * 16) =
              make_uint4(res[0], res[1], res[2], res[3]);
        if (g == 0) lse[((size_t)b * A + h) * S + q0 + j * 16 + li] = (m[j] * scale2 + __log2f(sm)) * 0.6931471805599453f;
      }
    }
  }
  if (!has_next) break;
  __syncthreads();   // every wave is done with sV: the next head's V panel
  item = next;
  h = item % A;
  b = item / A;
  slot ^= 1;
  stage_panel(qkv + (size_t)b * S * ld + h * AT_D + 2 * H, ld, S, sV, wid, lane);
  }  // items
}

// Column sums of a [16 rows x 64 cols] output fragment set accumulated over a workgroup's passes: the bias gradient of the
// fused QKV projection (d qkv.bias = column sums of dQ | dK | dV), so no separate pass re-reads the 3H-wide dqkv.
// acc[db][r] is this lane's running sum for column db*16 + g*4 + r (rows li); reduce over li, over the 8 waves, one atomic
// per column per workgroup.
template <int NW>
static __device__ __forceinline__ void flush_colsum(f4v (&acc)[4], float (*red)[64], float* __restrict__ out, int wid, int lane,
                                                    int tid) {
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = acc[db][r];
      x += __shfl_xor(x, 1, 64);
      x += __shfl_xor(x, 2, 64);
      x += __shfl_xor(x, 4, 64);
      x += __shfl_xor(x, 8, 64);
      if ((lane & 15) == 0) red[wid][db * 16 + (lane >> 4) * 4 + r] = x;
    }
  __syncthreads();
  if (tid < 64) {
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w][tid];
    atomicAdd(out + tid, t);
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// backward: dQ   (owner = query rows; panels K, V)
// ------------------------------------------------------------------------------------------
// DROP replays the forward mask: dP_eff = mask * dP / (1-p) before the softmax backward (D = rowdot(dO, O) already
// contains the dropped probabilities through O).
// It also produces D[b,h,q] = rowdot(dO, O) (the softmax-backward correction) for its own queries from the dO / O
// fragments it already needs, and writes it for the dK/dV kernel that runs next: no separate row-dot pass.
template <bool DROP, bool RES>
__global__ __launch_bounds__(AT_NWB * 64) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                             const bf16_t* __restrict__ ctx, const uint8_t* __restrict__ ctx_lo,
                                                             const float* __restrict__ maskbias, const float* __restrict__ lse,
                                                             float* __restrict__ Dv, bf16_t* __restrict__ dqkv, int S,
                                                             int H, int A, float scale, int rpw, uint32_t drop_seed,
                                                             uint32_t drop_thresh, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[AT_NWB][64];
  unsigned char* sK = smem;
  unsigned char* sV = smem + AT_MAXS * 128;
  float* sMask = reinterpret_cast<float*>(smem + 2 * AT_MAXS * 128);
  uint32_t* sCk = reinterpret_cast<uint32_t*>(sMask + AT_MAXS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int ld = 3 * H;
  const bf16_t* base = qkv + (size_t)b * S * ld + h * AT_D;
  const uint32_t bhS = (uint32_t)((b * A + h) * S);
  const float dscale = DROP ? drop_scale(drop_thresh) : 1.0f;
  f4v bsum[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) bsum[db] = (f4v){0.f, 0.f, 0.f, 0.f};
  stage_panel<AT_NWB>(base + H, ld, S, sK, wid, lane);
  stage_panel<AT_NWB>(base + 2 * H, ld, S, sV, wid, lane);
  for (int i = tid; i < S; i += AT_NWB * 64) {
    sMask[i] = maskbias[(size_t)b * S + i] * 1.4426950408889634f;
    if (DROP) sCk[i] = drop_colkey(drop_seed, bhS + (uint32_t)i);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int g = lane >> 4, li = lane & 15;
  const int nkc = S / 32;
  const float scale2 = scale * 1.4426950408889634f;
  const PanelBases pK = panel_bases(sK, lane), pV = panel_bases(sV, lane);
  const bf16_t* dob = dctx + (size_t)b * S * H + h * AT_D;
  const bf16_t* ob = ctx + (size_t)b * S * H + h * AT_D;
#pragma unroll 1
  for (int pass = 0; pass < rpw / (16 * AT_NWB); ++pass) {
    const int q0 = qt * rpw + wid * (rpw / AT_NWB) + pass * 16;
    if (q0 >= S) break;
    const bf16x8 qf0 = glb_frag(base, ld, q0, 0, lane);
    const bf16x8 qf1 = glb_frag(base, ld, q0, 1, lane);
    const bf16x8 do0 = glb_frag(dob, H, q0, 0, lane);
    const bf16x8 do1 = glb_frag(dob, H, q0, 1, lane);
    const size_t sidx = ((size_t)b * A + h) * S + q0 + li;
    const float l_q = lse[sidx] * 1.4426950408889634f;  // log2 domain
    // D = sum_d dO[q,d] O[q,d]: each lane group g holds 16 of the 64 d of row q0+li in its two fragments
    const bf16x8 of0 = glb_frag(ob, H, q0, 0, lane), of1 = glb_frag(ob, H, q0, 1, lane);
    uint32_t rw[4];
    if (RES) res_words(at_res_block(ctx_lo, b, A, h, S, q0), lane, rw);
    float d_part = dot8(do0, of0) + dot8(do1, of1);
    if (RES) d_part += res_dot16(do0, do1, rw);   // the residual of O
    const float d_true = group4_sum(d_part);
    if (g == 0) Dv[sidx] = d_true;
    // dropout: dS = (1 / (1-p)) P (m dP - (1-p) D) -- the 1 / (1-p) leaves through the final dQ scale (see attn_bwd_dq2_kernel)
    const float d_q = DROP ? d_true * (1.0f / dscale) : d_true;
    const uint32_t rk = DROP ? drop_rowkey(drop_seed, bhS + (uint32_t)(q0 + li)) : 0u;
    f4v dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = (f4v){0.f, 0.f, 0.f, 0.f};
    // two-stage software pipeline: the 8 score / dP MFMAs of chunk kc+1 are issued before the softmax VALU work
    // of chunk kc, so the matrix pipe runs under the exp2 / multiply stream of the same wave
#define DQ_SP(S0, S1, P0, P1, CO)                 \
  S0 = MFMA(kc_at(pK.kc[0], (CO)), qf0, zero4);          \
  S0 = MFMA(kc_at(pK.kc[1], (CO)), qf1, S0);             \
  S1 = MFMA(kc_at(pK.kc[0], (CO) + 2048), qf0, zero4);   \
  S1 = MFMA(kc_at(pK.kc[1], (CO) + 2048), qf1, S1);      \
  P0 = MFMA(kc_at(pV.kc[0], (CO)), do0, pinit);          \
  P0 = MFMA(kc_at(pV.kc[1], (CO)), do1, P0);             \
  P1 = MFMA(kc_at(pV.kc[0], (CO) + 2048), do0, pinit);   \
  P1 = MFMA(kc_at(pV.kc[1], (CO) + 2048), do1, P1)
    const f4v zero4 = (f4v){0.f, 0.f, 0.f, 0.f};
    // without dropout the dP accumulators START at -D (one query per lane): dS = P * (dP - D) loses its subtraction
    const f4v pinit = (f4v){-d_q, -d_q, -d_q, -d_q};
    f4v s0, s1, p0, p1;
    DQ_SP(s0, s1, p0, p1, 0);
    for (int kc = 0; kc < nkc; ++kc) {
      const int co = kc * 4096;
      // branch-free prefetch of the next chunk (the last iteration recomputes chunk 0 and discards it) so the whole
      // body is ONE basic block the scheduler can interleave
      f4v n0, n1, m0_, m1_;
      const int con = (kc + 1 < nkc) ? co + 4096 : 0;
      DQ_SP(n0, n1, m0_, m1_, con);
      const float4 m0 = *reinterpret_cast<const float4*>(sMask + kc * 32 + g * 4);
      const float4 m1 = *reinterpret_cast<const float4*>(sMask + kc * 32 + 16 + g * 4);
      const float mb0[4] = {m0.x, m0.y, m0.z, m0.w};
      const float mb1[4] = {m1.x, m1.y, m1.z, m1.w};
      f4v ds0, ds1;
      uint32_t ck0[4] = {0u, 0u, 0u, 0u}, ck1[4] = {0u, 0u, 0u, 0u};
      if (DROP) {
        const uint4 c0 = *reinterpret_cast<const uint4*>(sCk + kc * 32 + g * 4);
        const uint4 c1 = *reinterpret_cast<const uint4*>(sCk + kc * 32 + 16 + g * 4);
        ck0[0] = c0.x; ck0[1] = c0.y; ck0[2] = c0.z; ck0[3] = c0.w;
        ck1[0] = c1.x; ck1[1] = c1.y; ck1[2] = c1.z; ck1[3] = c1.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr0 = __builtin_amdgcn_exp2f(s0[r] * scale2 + mb0[r] - l_q);
        const float pr1 = __builtin_amdgcn_exp2f(s1[r] * scale2 + mb1[r] - l_q);
        float dp0 = p0[r], dp1 = p1[r];
        if (DROP) {
          dp0 = drop_keep(rk, ck0[r], drop_thresh) ? dp0 : -d_q;
          dp1 = drop_keep(rk, ck1[r], drop_thresh) ? dp1 : -d_q;
        }
        ds0[r] = pr0 * dp0;
        ds1[r] = pr1 * dp1;
      }
      const bf16x8 dsb = pack_b(ds0, ds1);
#pragma unroll
      for (int db = 0; db < 4; ++db) dq[db] = MFMA(tr_at(pK.tr[db], co), dsb, dq[db]);
      s0 = n0; s1 = n1; p0 = m0_; p1 = m1_;
    }
#undef DQ_SP
    bf16_t* orow = dqkv + (size_t)(b * S + q0 + li) * ld + h * AT_D;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      uint2 u;
      const float os = scale * dscale;
      u.x = pack2bf(dq[db][0] * os, dq[db][1] * os);
      u.y = pack2bf(dq[db][2] * os, dq[db][3] * os);
      *reinterpret_cast<uint2*>(orow + db * 16 + g * 4) = u;
      bsum[db] += dq[db] * os;
    }
  }
  if (dbias != nullptr) flush_colsum<AT_NWB>(bsum, red, dbias + h * AT_D, wid, lane, tid);
}

// ------------------------------------------------------------------------------------------
// backward: dK, dV   (owner = key rows; panels Q, dO)
// ------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(AT_NWB * 64) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                              const float* __restrict__ maskbias, const float* __restrict__ lse,
                                                              const float* __restrict__ Dv, bf16_t* __restrict__ dqkv, int S,
                                                              int H, int A, float scale, int rpw, uint32_t drop_seed,
                                                              uint32_t drop_thresh, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[AT_NWB][64];
  f4v bsk[4], bsv[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) bsk[db] = bsv[db] = (f4v){0.f, 0.f, 0.f, 0.f};
  unsigned char* sQ = smem;
  unsigned char* sO = smem + AT_MAXS * 128;
  float* sL = reinterpret_cast<float*>(smem + 2 * AT_MAXS * 128);
  float* sD = sL + AT_MAXS;
  uint32_t* sRk = reinterpret_cast<uint32_t*>(sD + AT_MAXS);  // dropout row keys of the head's queries
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int ld = 3 * H;
  const bf16_t* base = qkv + (size_t)b * S * ld + h * AT_D;
  const bf16_t* dob = dctx + (size_t)b * S * H + h * AT_D;
  stage_panel<AT_NWB>(base, ld, S, sQ, wid, lane);
  stage_panel<AT_NWB>(dob, H, S, sO, wid, lane);
  const size_t sbase = ((size_t)b * A + h) * S;
  const uint32_t bhS = (uint32_t)sbase;
  const float dscale = DROP ? drop_scale(drop_thresh) : 1.0f;
  for (int i = tid; i < S; i += AT_NWB * 64) {
    sL[i] = lse[sbase + i] * 1.4426950408889634f;  // log2 domain
    sD[i] = -Dv[sbase + i] * (1.0f / dscale);  // the dP accumulators start at -(1-p) D (= -D without dropout)
    if (DROP) sRk[i] = drop_rowkey(drop_seed, bhS + (uint32_t)i);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int g = lane >> 4, li = lane & 15;
  const int nqc = S / 32;
  const float scale2 = scale * 1.4426950408889634f;
  const PanelBases pQ = panel_bases(sQ, lane), pO = panel_bases(sO, lane);
#pragma unroll 1
  for (int pass = 0; pass < rpw / (16 * AT_NWB); ++pass) {
    const int k0 = kt * rpw + wid * (rpw / AT_NWB) + pass * 16;
    if (k0 >= S) break;
    const bf16x8 kf0 = glb_frag(base + H, ld, k0, 0, lane);
    const bf16x8 kf1 = glb_frag(base + H, ld, k0, 1, lane);
    const bf16x8 vf0 = glb_frag(base + 2 * H, ld, k0, 0, lane);
    const bf16x8 vf1 = glb_frag(base + 2 * H, ld, k0, 1, lane);
    const float mb = maskbias[(size_t)b * S + k0 + li] * 1.4426950408889634f;
    const uint32_t ck = DROP ? drop_colkey(drop_seed, bhS + (uint32_t)(k0 + li)) : 0u;
    f4v dk[4], dv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      dk[db] = (f4v){0.f, 0.f, 0.f, 0.f};
      dv[db] = (f4v){0.f, 0.f, 0.f, 0.f};
    }
#define nd_at(CO, off) (*reinterpret_cast<const f4v*>(sD + ((CO) >> 7) + (off) + g * 4))
#define DKV_SP(S0, S1, P0, P1, CO)                \
  S0 = MFMA(kc_at(pQ.kc[0], (CO)), kf0, zero4);          \
  S0 = MFMA(kc_at(pQ.kc[1], (CO)), kf1, S0);             \
  S1 = MFMA(kc_at(pQ.kc[0], (CO) + 2048), kf0, zero4);   \
  S1 = MFMA(kc_at(pQ.kc[1], (CO) + 2048), kf1, S1);      \
  P0 = MFMA(kc_at(pO.kc[0], (CO)), vf0, nd_at((CO), 0));  \
  P0 = MFMA(kc_at(pO.kc[1], (CO)), vf1, P0);             \
  P1 = MFMA(kc_at(pO.kc[0], (CO) + 2048), vf0, nd_at((CO), 16)); \
  P1 = MFMA(kc_at(pO.kc[1], (CO) + 2048), vf1, P1)
    const f4v zero4 = (f4v){0.f, 0.f, 0.f, 0.f};
    f4v s0, s1, p0, p1;
    DKV_SP(s0, s1, p0, p1, 0);
    for (int qc = 0; qc < nqc; ++qc) {
      // S and dP in [q rows, key cols] orientation: lane holds q = qc*32 + f*16 + g*4 + r, key = k0 + li
      const int co = qc * 4096;
      f4v n0, n1, m0_, m1_;
      const int con = (qc + 1 < nqc) ? co + 4096 : 0;  // branch-free (see attn_bwd_dq_kernel)
      DKV_SP(n0, n1, m0_, m1_, con);
      const float4 l0 = *reinterpret_cast<const float4*>(sL + qc * 32 + g * 4);
      const float4 l1 = *reinterpret_cast<const float4*>(sL + qc * 32 + 16 + g * 4);
      const float4 d0 = *reinterpret_cast<const float4*>(sD + qc * 32 + g * 4);
      const float4 d1 = *reinterpret_cast<const float4*>(sD + qc * 32 + 16 + g * 4);
      const float la[4] = {l0.x, l0.y, l0.z, l0.w}, lb[4] = {l1.x, l1.y, l1.z, l1.w};
      const float da[4] = {d0.x, d0.y, d0.z, d0.w}, dbv[4] = {d1.x, d1.y, d1.z, d1.w};
      f4v pr0, pr1, ds0, ds1;
      uint32_t rk0[4] = {0u, 0u, 0u, 0u}, rk1[4] = {0u, 0u, 0u, 0u};
      if (DROP) {
        const uint4 c0 = *reinterpret_cast<const uint4*>(sRk + qc * 32 + g * 4);
        const uint4 c1 = *reinterpret_cast<const uint4*>(sRk + qc * 32 + 16 + g * 4);
        rk0[0] = c0.x; rk0[1] = c0.y; rk0[2] = c0.z; rk0[3] = c0.w;
        rk1[0] = c1.x; rk1[1] = c1.y; rk1[2] = c1.z; rk1[3] = c1.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e0 = __builtin_amdgcn_exp2f(s0[r] * scale2 + mb - la[r]);
        const float e1 = __builtin_amdgcn_exp2f(s1[r] * scale2 + mb - lb[r]);
        float dp0 = p0[r], dp1 = p1[r];
        pr0[r] = e0;
        pr1[r] = e1;
        if (DROP) {  // dV sees mask * P / (1-p); dS = P * (mask * dP / (1-p) - D)
          const bool k0_ = drop_keep(rk0[r], ck, drop_thresh), k1_ = drop_keep(rk1[r], ck, drop_thresh);
          pr0[r] = k0_ ? e0 : 0.0f;
          pr1[r] = k1_ ? e1 : 0.0f;
          dp0 = k0_ ? dp0 : da[r];     // da / dbv hold -(1-p) D, the accumulators' start value
          dp1 = k1_ ? dp1 : dbv[r];
        }
        ds0[r] = e0 * dp0;
        ds1[r] = e1 * dp1;
      }
      const bf16x8 pb = pack_b(pr0, pr1);
      const bf16x8 dsb = pack_b(ds0, ds1);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        dv[db] = MFMA(tr_at(pO.tr[db], co), pb, dv[db]);
        dk[db] = MFMA(tr_at(pQ.tr[db], co), dsb, dk[db]);
      }
      s0 = n0; s1 = n1; p0 = m0_; p1 = m1_;
    }
#undef DKV_SP
#undef nd_at
    bf16_t* krow = dqkv + (size_t)(b * S + k0 + li) * ld + H + h * AT_D;
    bf16_t* vrow = krow + H;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      uint2 u;
      const float ks = scale * dscale;
      u.x = pack2bf(dk[db][0] * ks, dk[db][1] * ks);
      u.y = pack2bf(dk[db][2] * ks, dk[db][3] * ks);
      *reinterpret_cast<uint2*>(krow + db * 16 + g * 4) = u;
      uint2 w;
      w.x = pack2bf(dv[db][0] * dscale, dv[db][1] * dscale);
      w.y = pack2bf(dv[db][2] * dscale, dv[db][3] * dscale);
      *reinterpret_cast<uint2*>(vrow + db * 16 + g * 4) = w;
      bsk[db] += dk[db] * ks;
      bsv[db] += dv[db] * dscale;
    }
  }
  if (dbias != nullptr) {
    flush_colsum<AT_NWB>(bsk, red, dbias + H + h * AT_D, wid, lane, tid);
    flush_colsum<AT_NWB>(bsv, red, dbias + 2 * H + h * AT_D, wid, lane, tid);
  }
}

// ------------------------------------------------------------------------------------------
// backward, second structure: 32 stationary rows per wave (two 16-row fragments)
// ------------------------------------------------------------------------------------------
// The 16-row kernels above read every LDS operand fragment for ONE MFMA: per 32-key chunk a wave moves 8 KiB (S, dP) +
// 4 KiB (transposed) out of LDS for 12 MFMAs.  Here every fragment read feeds TWO MFMAs (row blocks j = 0, 1 of the wave's
// 32 rows), halving the LDS bytes per flop, and the two independent row blocks give the scheduler an MFMA stream (block 1's
// scores) to run under block 0's exp2 / multiply work.  Measured on MI355X (B=128 x 16 heads, S=512): dQ + dK/dV 813 -> 735 us;
// with the score accumulators started at -lse/scale (P = exp2(scale2 * acc): no subtract / fma per score), the mask add only in
// chunks that hold a masked key and two chunks per loop trip (LDS address adds 30 -> 10 per chunk) 717 us -- a third fewer VALU
// instructions bought 2.5 %: rocprofv3 --pmc (profiles/round2_attn_pmc.txt) shows neither pipe saturated (MFMA busy 33 %, VALU
// issue 26 % per wave, LDS 22 %, 36-42 % of the wave cycles parked in s_waitcnt): at two waves per SIMD the loop is bound by
// its own dependency chain (LDS read -> MFMA pair -> exp2 -> cvt -> MFMA), not by a throughput limit.
// (Register double-buffering of the fragment stream one chunk ahead -- 32 more VGPRs + sched_barriers -- measured SLOWER,
// 783 us.)
// Key chunks that lie entirely behind the sentence's last unmasked key are skipped: their probabilities are exp(-10000 + x)
// = 0 exactly in fp32 (as in the reference), so they contribute exactly nothing -- length-sorted real batches are padded
// (711 us at 450 real keys of 512).
//
// klen: 1 + index of the last key whose mask bias is 0 (prefix masks: the number of real sub-tokens).
template <int NT>
static __device__ __forceinline__ int stage_mask_klen(const float* __restrict__ maskbias, size_t off, int S, float inv_scale,
                                                       float* sMask, int* sKlen, int tid, int& nfree) {
  // sMask[i] = maskbias[i] / scale: added to the RAW score sums (see the accumulator-init note in the kernels).
  // klen = 1 + last unmasked key; nfree = leading 32-key chunks that hold no masked key at all (prefix masks: klen / 32)
  if (tid < 2) sKlen[tid] = 0;
  __syncthreads();
  int last = 0, cnt = 0;
  for (int i = tid; i < S; i += NT) {
    const float m = maskbias[off + i];
    sMask[i] = m * inv_scale;
    if (m > -1.0f) {   // additive bias 0 = attend (anything near -10000 = masked)
      last = i + 1;
      cnt += 1;
    }
  }
  if (last) {
    atomicMax(sKlen, last);
    atomicAdd(sKlen + 1, cnt);
  }
  __syncthreads();
  const int klen = sKlen[0];
  nfree = (sKlen[1] == klen) ? (klen >> 5) : 0;
  return klen;
}

// RES: ctx_lo is given.  A template parameter, not a run-time test: behind `if (ctx_lo)` the residual loads could not be issued
// with the pass's other loads and their HBM latency came on top of the O fragments' (in situ, where nothing of ctx_lo is cached,
// the kernel ran 375 us per layer against 337 without the residual; the lab, whose buffers stay in the memory-side cache, had
// shown +8 us)
template <bool DROP, bool RES>
__global__ __launch_bounds__(512) void attn_bwd_dq2_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                          const bf16_t* __restrict__ ctx, const uint8_t* __restrict__ ctx_lo,
                                                          const float* __restrict__ maskbias,
                                                          const float* __restrict__ lse, float* __restrict__ Dv,
                                                          bf16_t* __restrict__ dqkv, int S, int H, int A, float scale, int rpw,
                                                          uint32_t drop_seed, uint32_t drop_thresh, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[8][64];
  __shared__ int sKlen[2];
  unsigned char* sK = smem;
  unsigned char* sV = smem + AT_MAXS * 128;
  float* sMask = reinterpret_cast<float*>(smem + 2 * AT_MAXS * 128);
  uint32_t* sCk = reinterpret_cast<uint32_t*>(sMask + AT_MAXS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int ld = 3 * H;
  const bf16_t* base = qkv + (size_t)b * S * ld + h * AT_D;
  const uint32_t bhS = (uint32_t)((b * A + h) * S);
  const float dscale = DROP ? drop_scale(drop_thresh) : 1.0f;
  f4v bsum[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) bsum[db] = (f4v){0.f, 0.f, 0.f, 0.f};
  stage_panel<8>(base + H, ld, S, sK, wid, lane);
  stage_panel<8>(base + 2 * H, ld, S, sV, wid, lane);
  int nfree;
  const int klen = stage_mask_klen<512>(maskbias, (size_t)b * S, S, 1.0f / scale, sMask, sKlen, tid, nfree);
  if (DROP)
    for (int i = tid; i < S; i += 512) sCk[i] = drop_colkey(drop_seed, bhS + (uint32_t)i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int g = lane >> 4, li = lane & 15;
  const int nkc = ((klen + 63) >> 6) << 1;   // key chunks holding an unmasked key, rounded up to the loop's unroll of 2
                                             // (a fully masked extra chunk has P = 0: it adds exactly nothing)
  const float scale2 = scale * 1.4426950408889634f;
  const float oscale = scale * dscale;   // dQ = scale / (1-p) * (dS' K), see the accumulator-start note below
  const PanelBases pK = panel_bases(sK, lane), pV = panel_bases(sV, lane);
  const bf16_t* dob = dctx + (size_t)b * S * H + h * AT_D;
  const bf16_t* ob = ctx + (size_t)b * S * H + h * AT_D;
  const f4v zero4 = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int pass = 0; pass < rpw / 256; ++pass) {
    const int q0 = qt * rpw + wid * (rpw / 8) + pass * 32;
    if (q0 >= S) break;
    bf16x8 qf[2][2], dof[2][2];
    float l_q[2], d_q[2];
    uint32_t rk[2];
    f4v dq[2][4];
    bf16x8 of[2][2];
    uint32_t rw[2][4];
    float lse_q[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {   // every load of the pass first, so that their latencies overlap
      const int qj = q0 + j * 16;
      qf[j][0] = glb_frag(base, ld, qj, 0, lane);
      qf[j][1] = glb_frag(base, ld, qj, 1, lane);
      dof[j][0] = glb_frag(dob, H, qj, 0, lane);
      dof[j][1] = glb_frag(dob, H, qj, 1, lane);
      of[j][0] = glb_frag(ob, H, qj, 0, lane);
      of[j][1] = glb_frag(ob, H, qj, 1, lane);
      if (RES) res_words(at_res_block(ctx_lo, b, A, h, S, qj), lane, rw[j]);
      lse_q[j] = lse[((size_t)b * A + h) * S + qj + li];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int qj = q0 + j * 16;
      const size_t sidx = ((size_t)b * A + h) * S + qj + li;
      l_q[j] = -lse_q[j] / scale;   // accumulator init of the score MFMAs: exp2(scale2 * (q.k + mask/scale - lse/scale))
      float d_part = dot8(dof[j][0], of[j][0]) + dot8(dof[j][1], of[j][1]);
      if (RES)   // the residual O - bf16(O) the forward kept: D to ~12 bits of O (see kbner_attn_bwd)
        d_part += res_dot16(dof[j][0], dof[j][1], rw[j]);
      d_q[j] = group4_sum(d_part);
      if (g == 0) Dv[sidx] = d_q[j];
      // with dropout dS = P (m dP / (1-p) - D) = (1 / (1-p)) P (m dP - (1-p) D): the 1 / (1-p) moves to the final dQ scale and the
      // accumulators of dP start at -(1-p) D as they start at -D without dropout (a dropped element keeps that start value)
      if (DROP) d_q[j] *= 1.0f / dscale;
      rk[j] = DROP ? drop_rowkey(drop_seed, bhS + (uint32_t)(qj + li)) : 0u;
#pragma unroll
      for (int db = 0; db < 4; ++db) dq[j][db] = zero4;
    }
    // Two chunks per trip with compile-time offsets inside the pair: the 12 per-lane LDS addresses are advanced once per
    // pair instead of one v_add per fragment read (30 of the ~110 VALU instructions of a chunk were address adds).
    // Softmax backward arithmetic per score: the score accumulators START at -lse/scale (per query, one value per lane) so
    //   P = exp2(scale2 * acc)                       -- one multiply + one exp, no subtract, no fma
    // and only chunks that hold a masked key (at most the last one for prefix masks) add mask/scale first.
    for (int kc = 0; kc < nkc; kc += 2) {
      const unsigned char* bk0 = pK.kc[0] + kc * 4096;
      const unsigned char* bk1 = pK.kc[1] + kc * 4096;
      const unsigned char* bv0 = pV.kc[0] + kc * 4096;
      const unsigned char* bv1 = pV.kc[1] + kc * 4096;
      const unsigned char* bt[4] = {pK.tr[0] + kc * 4096, pK.tr[1] + kc * 4096, pK.tr[2] + kc * 4096, pK.tr[3] + kc * 4096};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        constexpr int dummy = 0;
        (void)dummy;
        const int co = u * 4096;
        const bool masked = (kc + u) >= nfree;   // wave-uniform
        const bf16x8 k00 = kc_at(bk0, co), k01 = kc_at(bk1, co);
        const bf16x8 k10 = kc_at(bk0, co + 2048), k11 = kc_at(bk1, co + 2048);
        const bf16x8 v00 = kc_at(bv0, co), v01 = kc_at(bv1, co);
        const bf16x8 v10 = kc_at(bv0, co + 2048), v11 = kc_at(bv1, co + 2048);
        uint32_t ck0[4] = {0u, 0u, 0u, 0u}, ck1[4] = {0u, 0u, 0u, 0u};
        if (DROP) {
          const uint4 c0 = *reinterpret_cast<const uint4*>(sCk + (kc + u) * 32 + g * 4);
          const uint4 c1 = *reinterpret_cast<const uint4*>(sCk + (kc + u) * 32 + 16 + g * 4);
          ck0[0] = c0.x; ck0[1] = c0.y; ck0[2] = c0.z; ck0[3] = c0.w;
          ck1[0] = c1.x; ck1[1] = c1.y; ck1[2] = c1.z; ck1[3] = c1.w;
        }
        // S^T and dP^T tiles [key, query] (lane: keys g*4 + r of each 16-key fragment, query li) for both row blocks
        f4v s0[2], s1[2], p0[2], p1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f4v sinit = (f4v){l_q[j], l_q[j], l_q[j], l_q[j]};
          const f4v pinit = (f4v){-d_q[j], -d_q[j], -d_q[j], -d_q[j]};
          s0[j] = MFMA(k00, qf[j][0], sinit);
          s0[j] = MFMA(k01, qf[j][1], s0[j]);
          s1[j] = MFMA(k10, qf[j][0], sinit);
          s1[j] = MFMA(k11, qf[j][1], s1[j]);
          p0[j] = MFMA(v00, dof[j][0], pinit);
          p0[j] = MFMA(v01, dof[j][1], p0[j]);
          p1[j] = MFMA(v10, dof[j][0], pinit);
          p1[j] = MFMA(v11, dof[j][1], p1[j]);
        }
        if (masked) {
          const f4v m0 = *reinterpret_cast<const f4v*>(sMask + (kc + u) * 32 + g * 4);
          const f4v m1 = *reinterpret_cast<const f4v*>(sMask + (kc + u) * 32 + 16 + g * 4);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            s0[j] += m0;
            s1[j] += m1;
          }
        }
        bf16x8 dsb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f4v ds0, ds1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr0 = __builtin_amdgcn_exp2f(s0[j][r] * scale2);
            const float pr1 = __builtin_amdgcn_exp2f(s1[j][r] * scale2);
            float dp0 = p0[j][r], dp1 = p1[j][r];
            if (DROP) {
              dp0 = drop_keep(rk[j], ck0[r], drop_thresh) ? dp0 : -d_q[j];
              dp1 = drop_keep(rk[j], ck1[r], drop_thresh) ? dp1 : -d_q[j];
            }
            ds0[r] = pr0 * dp0;
            ds1[r] = pr1 * dp1;
          }
          dsb[j] = pack_b(ds0, ds1);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 t = tr_at(bt[db], co);
          dq[0][db] = MFMA(t, dsb[0], dq[0][db]);
          dq[1][db] = MFMA(t, dsb[1], dq[1][db]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16_t* orow = dqkv + (size_t)(b * S + q0 + j * 16 + li) * ld + h * AT_D;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        uint2 u;
        u.x = pack2bf(dq[j][db][0] * oscale, dq[j][db][1] * oscale);
        u.y = pack2bf(dq[j][db][2] * oscale, dq[j][db][3] * oscale);
        *reinterpret_cast<uint2*>(orow + db * 16 + g * 4) = u;
        bsum[db] += dq[j][db] * oscale;
      }
    }
  }
  if (dbias != nullptr) flush_colsum<8>(bsum, red, dbias + h * AT_D, wid, lane, tid);
}

template <bool DROP>
__global__ __launch_bounds__(512) void attn_bwd_dkv2_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                           const float* __restrict__ maskbias, const float* __restrict__ lse,
                                                           const float* __restrict__ Dv, bf16_t* __restrict__ dqkv, int S,
                                                           int H, int A, float scale, int rpw, uint32_t drop_seed,
                                                           uint32_t drop_thresh, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[8][64];
  f4v bsk[4], bsv[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) bsk[db] = bsv[db] = (f4v){0.f, 0.f, 0.f, 0.f};
  unsigned char* sQ = smem;
  unsigned char* sO = smem + AT_MAXS * 128;
  float* sL = reinterpret_cast<float*>(smem + 2 * AT_MAXS * 128);
  float* sD = sL + AT_MAXS;
  uint32_t* sRk = reinterpret_cast<uint32_t*>(sD + AT_MAXS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int ld = 3 * H;
  const bf16_t* base = qkv + (size_t)b * S * ld + h * AT_D;
  const bf16_t* dob = dctx + (size_t)b * S * H + h * AT_D;
  stage_panel<8>(base, ld, S, sQ, wid, lane);
  stage_panel<8>(dob, H, S, sO, wid, lane);
  const size_t sbase = ((size_t)b * A + h) * S;
  const uint32_t bhS = (uint32_t)sbase;
  const float dscale = DROP ? drop_scale(drop_thresh) : 1.0f;
  for (int i = tid; i < S; i += 512) {
    sL[i] = -lse[sbase + i] / scale;   // accumulator init of the score MFMAs (see attn_bwd_dq2_kernel)
    sD[i] = -Dv[sbase + i] * (1.0f / dscale);   // -(1-p) D: the 1 / (1-p) of dropout leaves through the final dK / dV scales
    if (DROP) sRk[i] = drop_rowkey(drop_seed, bhS + (uint32_t)i);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int g = lane >> 4, li = lane & 15;
  const int nqc = S / 32;
  const float scale2 = scale * 1.4426950408889634f;
  const PanelBases pQ = panel_bases(sQ, lane), pO = panel_bases(sO, lane);
  const f4v zero4 = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int pass = 0; pass < rpw / 256; ++pass) {
    const int k0 = kt * rpw + wid * (rpw / 8) + pass * 32;
    if (k0 >= S) break;
    bf16x8 kf[2][2], vf[2][2];
    float mb[2];
    uint32_t ck[2];
    f4v dk[2][4], dv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kj = k0 + j * 16;
      kf[j][0] = glb_frag(base + H, ld, kj, 0, lane);
      kf[j][1] = glb_frag(base + H, ld, kj, 1, lane);
      vf[j][0] = glb_frag(base + 2 * H, ld, kj, 0, lane);
      vf[j][1] = glb_frag(base + 2 * H, ld, kj, 1, lane);
      mb[j] = maskbias[(size_t)b * S + kj + li] / scale;
      ck[j] = DROP ? drop_colkey(drop_seed, bhS + (uint32_t)(kj + li)) : 0u;
#pragma unroll
      for (int db = 0; db < 4; ++db) dk[j][db] = dv[j][db] = zero4;
    }
    // a wave whose 32 keys are all masked has P = 0 exactly: dK = dV = 0 for its rows (wave-uniform skip of the loop)
    const bool live = __builtin_amdgcn_readfirstlane(__any((mb[0] > -8.0f) || (mb[1] > -8.0f)) ? 1 : 0) != 0;
    // none of the wave's 32 keys masked (every full-length sentence): the mask add disappears
    const bool masked = __builtin_amdgcn_readfirstlane(__any((mb[0] != 0.0f) || (mb[1] != 0.0f)) ? 1 : 0) != 0;
    // two query chunks per trip, compile-time offsets inside the pair (S % 64 == 0, so nqc is even)
    for (int qc = 0; live && qc < nqc; qc += 2) {
      const unsigned char* bq0 = pQ.kc[0] + qc * 4096;
      const unsigned char* bq1 = pQ.kc[1] + qc * 4096;
      const unsigned char* bo0 = pO.kc[0] + qc * 4096;
      const unsigned char* bo1 = pO.kc[1] + qc * 4096;
      const unsigned char* btq[4] = {pQ.tr[0] + qc * 4096, pQ.tr[1] + qc * 4096, pQ.tr[2] + qc * 4096, pQ.tr[3] + qc * 4096};
      const unsigned char* bto[4] = {pO.tr[0] + qc * 4096, pO.tr[1] + qc * 4096, pO.tr[2] + qc * 4096, pO.tr[3] + qc * 4096};
      const float* pl = sL + qc * 32 + g * 4;
      const float* pd = sD + qc * 32 + g * 4;
      const uint32_t* prk = sRk + qc * 32 + g * 4;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int co = u * 4096;
        const bf16x8 q00 = kc_at(bq0, co), q01 = kc_at(bq1, co);
        const bf16x8 q10 = kc_at(bq0, co + 2048), q11 = kc_at(bq1, co + 2048);
        const bf16x8 o00 = kc_at(bo0, co), o01 = kc_at(bo1, co);
        const bf16x8 o10 = kc_at(bo0, co + 2048), o11 = kc_at(bo1, co + 2048);
        const f4v sl0 = *reinterpret_cast<const f4v*>(pl + u * 32);         // -lse/scale of queries qc*32 + g*4 + r
        const f4v sl1 = *reinterpret_cast<const f4v*>(pl + u * 32 + 16);
        const f4v nd0 = *reinterpret_cast<const f4v*>(pd + u * 32);
        const f4v nd1 = *reinterpret_cast<const f4v*>(pd + u * 32 + 16);
        uint32_t rk0[4] = {0u, 0u, 0u, 0u}, rk1[4] = {0u, 0u, 0u, 0u};
        if (DROP) {
          const uint4 c0 = *reinterpret_cast<const uint4*>(prk + u * 32);
          const uint4 c1 = *reinterpret_cast<const uint4*>(prk + u * 32 + 16);
          rk0[0] = c0.x; rk0[1] = c0.y; rk0[2] = c0.z; rk0[3] = c0.w;
          rk1[0] = c1.x; rk1[1] = c1.y; rk1[2] = c1.z; rk1[3] = c1.w;
        }
        // S and dP tiles [query, key]: lane holds queries (qc+u)*32 + f*16 + g*4 + r, key k0 + j*16 + li
        f4v s0[2], s1[2], p0[2], p1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          s0[j] = MFMA(q00, kf[j][0], sl0);
          s0[j] = MFMA(q01, kf[j][1], s0[j]);
          s1[j] = MFMA(q10, kf[j][0], sl1);
          s1[j] = MFMA(q11, kf[j][1], s1[j]);
          p0[j] = MFMA(o00, vf[j][0], nd0);
          p0[j] = MFMA(o01, vf[j][1], p0[j]);
          p1[j] = MFMA(o10, vf[j][0], nd1);
          p1[j] = MFMA(o11, vf[j][1], p1[j]);
        }
        if (masked) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            s0[j] += mb[j];
            s1[j] += mb[j];
          }
        }
        bf16x8 pb[2], dsb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f4v pr0, pr1, ds0, ds1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e0 = __builtin_amdgcn_exp2f(s0[j][r] * scale2);
            const float e1 = __builtin_amdgcn_exp2f(s1[j][r] * scale2);
            float dp0 = p0[j][r], dp1 = p1[j][r];
            pr0[r] = e0;
            pr1[r] = e1;
            if (DROP) {
              const bool k0_ = drop_keep(rk0[r], ck[j], drop_thresh), k1_ = drop_keep(rk1[r], ck[j], drop_thresh);
              pr0[r] = k0_ ? e0 : 0.0f;          // (kept probabilities unscaled: dV carries the 1 / (1-p))
              pr1[r] = k1_ ? e1 : 0.0f;
              dp0 = k0_ ? dp0 : nd0[r];          // a dropped element keeps the accumulator's start value -(1-p) D
              dp1 = k1_ ? dp1 : nd1[r];
            }
            ds0[r] = e0 * dp0;
            ds1[r] = e1 * dp1;
          }
          pb[j] = pack_b(pr0, pr1);
          dsb[j] = pack_b(ds0, ds1);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const bf16x8 tO = tr_at(bto[db], co);
          const bf16x8 tQ = tr_at(btq[db], co);
          dv[0][db] = MFMA(tO, pb[0], dv[0][db]);
          dv[1][db] = MFMA(tO, pb[1], dv[1][db]);
          dk[0][db] = MFMA(tQ, dsb[0], dk[0][db]);
          dk[1][db] = MFMA(tQ, dsb[1], dk[1][db]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16_t* krow = dqkv + (size_t)(b * S + k0 + j * 16 + li) * ld + H + h * AT_D;
      bf16_t* vrow = krow + H;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        uint2 u;
        const float ks = scale * dscale;
        u.x = pack2bf(dk[j][db][0] * ks, dk[j][db][1] * ks);
        u.y = pack2bf(dk[j][db][2] * ks, dk[j][db][3] * ks);
        *reinterpret_cast<uint2*>(krow + db * 16 + g * 4) = u;
        uint2 w;
        w.x = pack2bf(dv[j][db][0] * dscale, dv[j][db][1] * dscale);
        w.y = pack2bf(dv[j][db][2] * dscale, dv[j][db][3] * dscale);
        *reinterpret_cast<uint2*>(vrow + db * 16 + g * 4) = w;
        bsk[db] += dk[j][db] * ks;
        bsv[db] += dv[j][db] * dscale;
      }
    }
  }
  if (dbias != nullptr) {
    flush_colsum<8>(bsk, red, dbias + H + h * AT_D, wid, lane, tid);
    flush_colsum<8>(bsv, red, dbias + 2 * H + h * AT_D, wid, lane, tid);
  }
}

#ifdef KBNER_ATTN_LAB
#include "attn_bwd3.h"   // tools/experiments/: round 5's software-pipelined backward stream (bit-identical, measured slower)
#endif

#define AT_LDS_BYTES (2 * AT_MAXS * 128 + 3 * AT_MAXS * 4)
#ifndef KBNER_ATTN_PARTS
#define KBNER_ATTN_PARTS 2   // key-axis parts of the 32-row forward kernel (4, for S % 128 == 0, measured slower: 263-268 vs 247-251 us)
#endif

// rows per workgroup: the whole head (one DMA of each panel per head) when the grid still covers the
// chip several times over, otherwise smaller row tiles so small batches spread over more CUs
static inline int pick_rpw(int B, int S, int A) {
  int rpw = 512;
  while (rpw > 128 && (rpw / 2 >= S || (long)B * A * ((S + rpw - 1) / rpw) < 512)) rpw /= 2;
  return rpw;
}

static int at_cu_count() { return kbner_cu_count(); }

template <int NKB, bool DROP>
static int launch_attn_fwd2(const bf16_t* qkv, const float* maskbias, bf16_t* ctx, uint8_t* ctx_lo, float* lse, int B, int H, int A, int rpw,
                            uint32_t seed, uint32_t thresh, hipStream_t stream, bool rows32) {
  static std::atomic<unsigned long long> done0{0}, done1{0};   // one bit per device (common.h)
  int r = kbner_set_max_lds_once(done0, reinterpret_cast<const void*>(attn_fwd_kernel<NKB, DROP, false>), AT_LDS_BYTES);
  if (r) return r;
  r = kbner_set_max_lds_once(done1, reinterpret_cast<const void*>(attn_fwd_kernel<NKB, DROP, true>), AT_LDS_BYTES);
  if (r) return r;
  const int S = NKB * 16;
  const int ncu = at_cu_count();
  // 32 rows per wave and pass (round 3): half the LDS bytes per flop.  Not with dropout: in halves the keep tests push the S = 512
  // instantiation over 256 VGPRs (94-148 spills), in quarters it fits but runs at 322 us against the 16-row kernel's 315
  if (rows32 && !DROP && NKB % 4 == 0 && rpw % 256 == 0) {
    constexpr int NP = (NKB % 8 == 0) ? KBNER_ATTN_PARTS : 2;
    static std::atomic<unsigned long long> done2{0}, done3{0};
    r = kbner_set_max_lds_once(done2, reinterpret_cast<const void*>(attn_fwd32_kernel<NKB, DROP, false, NP>), AT_LDS_BYTES);
    if (r) return r;
    r = kbner_set_max_lds_once(done3, reinterpret_cast<const void*>(attn_fwd32_kernel<NKB, DROP, true, NP>), AT_LDS_BYTES);
    if (r) return r;
    if (rpw == S && B * A >= 2 * ncu)
      hipLaunchKernelGGL((attn_fwd32_kernel<NKB, DROP, true, NP>), dim3(ncu), dim3(512), AT_LDS_BYTES, stream, qkv, maskbias, ctx, ctx_lo,
                         lse, H, A, 0.125f, rpw, seed, thresh, B * A);
    else
      hipLaunchKernelGGL((attn_fwd32_kernel<NKB, DROP, false, NP>), dim3((S + rpw - 1) / rpw, A, B), dim3(512), AT_LDS_BYTES, stream,
                         qkv, maskbias, ctx, ctx_lo, lse, H, A, 0.125f, rpw, seed, thresh, 1);
    hipError_t e32 = hipGetLastError();
    return e32 == hipSuccess ? 0 : -(int)e32;
  }
  if (rpw == S && B * A >= 2 * ncu) {   // whole heads, at least two per CU: walk them persistently, prefetching the next (-4 % at
                                        // full length, -8 % with ragged masks, tools/attn_bench.py at B = 128)
    hipLaunchKernelGGL((attn_fwd_kernel<NKB, DROP, true>), dim3(ncu), dim3(512), AT_LDS_BYTES, stream, qkv, maskbias, ctx, ctx_lo, lse, H,
                       A, 0.125f, rpw, seed, thresh, B * A);
  } else {
    hipLaunchKernelGGL((attn_fwd_kernel<NKB, DROP, false>), dim3((S + rpw - 1) / rpw, A, B), dim3(512), AT_LDS_BYTES, stream, qkv,
                       maskbias, ctx, ctx_lo, lse, H, A, 0.125f, rpw, seed, thresh, 1);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
template <int NKB>
static int launch_attn_fwd(const bf16_t* qkv, const float* maskbias, bf16_t* ctx, uint8_t* ctx_lo, float* lse, int B, int H, int A, int rpw,
                           uint32_t seed, uint32_t thresh, hipStream_t stream, bool rows32) {
  if (thresh) return launch_attn_fwd2<NKB, true>(qkv, maskbias, ctx, ctx_lo, lse, B, H, A, rpw, seed, thresh, stream, rows32);
  return launch_attn_fwd2<NKB, false>(qkv, maskbias, ctx, ctx_lo, lse, B, H, A, rpw, seed, thresh, stream, rows32);
}

#ifdef KBNER_ATTN_LAB
// Lab builds only (-DKBNER_ATTN_LAB, linked with tools/experiments/attention3.hip): the round-3 streaming forward and the A/B
// switch KBNER_ATTN (1 = the 16-row-per-pass forward, 3 = streaming forward where it applies, 4 = forced).  The product
// library has neither: it reads no environment variable.
int kbner_attn_fwd3(const bf16_t* qkv, const float* maskbias, bf16_t* ctx, float* lse, int B, int S, int H, int A,
                    uint32_t drop_seed, uint32_t drop_thresh, hipStream_t stream);
static int attn_variant() {
  static std::atomic<int> v{-1};
  int r = v.load(std::memory_order_relaxed);
  if (r < 0) {
    const char* e = getenv("KBNER_ATTN");
    r = e ? atoi(e) : 2;
    v.store(r, std::memory_order_relaxed);
  }
  return r;
}
#endif

template <bool DROP, bool RES>
static int launch_attn_bwd2(const bf16_t* qkv, const bf16_t* ctx, const uint8_t* ctx_lo, const bf16_t* dctx, const float* maskbias,
                            const float* lse, float* Dws, bf16_t* dqkv, int B, int S, int H, int A, uint32_t seed, uint32_t thresh,
                            float* dbias, hipStream_t s) {
  static std::atomic<unsigned long long> done0{0}, done1{0}, done2{0}, done3{0};   // one bit per device (common.h)
  int r = kbner_set_max_lds_once(done0, reinterpret_cast<const void*>(attn_bwd_dq_kernel<DROP, RES>), AT_LDS_BYTES);
  if (r) return r;
  r = kbner_set_max_lds_once(done1, reinterpret_cast<const void*>(attn_bwd_dkv_kernel<DROP>), AT_LDS_BYTES);
  if (r) return r;
  int rpw = pick_rpw(B, S, A);
  if (rpw < 16 * AT_NWB) rpw = 16 * AT_NWB;  // every wave owns at least one 16-row pass
  const dim3 grid((S + rpw - 1) / rpw, A, B);
  // 32-row-stationary kernels whenever a workgroup's row tile gives each of its 8 waves whole 32-row passes
  if (rpw % 256 == 0 && S % 32 == 0) {
    r = kbner_set_max_lds_once(done2, reinterpret_cast<const void*>(attn_bwd_dq2_kernel<DROP, RES>), AT_LDS_BYTES);
    if (r) return r;
    r = kbner_set_max_lds_once(done3, reinterpret_cast<const void*>(attn_bwd_dkv2_kernel<DROP>), AT_LDS_BYTES);
    if (r) return r;
#ifdef KBNER_ATTN_LAB
    if (!DROP && attn_variant() == 5) {   // KBNER_ATTN=5: the pipelined backward stream (lab builds)
      static std::atomic<unsigned long long> done4{0}, done5{0};
      r = kbner_set_max_lds_once(done4, reinterpret_cast<const void*>(attn_bwd_dq3_kernel<RES>), AT_LDS_BYTES);
      if (r) return r;
      r = kbner_set_max_lds_once(done5, reinterpret_cast<const void*>(attn_bwd_dkv3_kernel), AT_LDS_BYTES);
      if (r) return r;
      hipLaunchKernelGGL((attn_bwd_dq3_kernel<RES>), grid, dim3(512), AT_LDS_BYTES, s, qkv, dctx, ctx, ctx_lo, maskbias, lse, Dws, dqkv, S,
                         H, A, 0.125f, rpw, dbias);
      hipLaunchKernelGGL(attn_bwd_dkv3_kernel, grid, dim3(512), AT_LDS_BYTES, s, qkv, dctx, maskbias, lse, Dws, dqkv, S, H, A, 0.125f, rpw,
                         dbias);
      hipError_t e3 = hipGetLastError();
      return e3 == hipSuccess ? 0 : -(int)e3;
    }
#endif
    hipLaunchKernelGGL((attn_bwd_dq2_kernel<DROP, RES>), grid, dim3(512), AT_LDS_BYTES, s, qkv, dctx, ctx, ctx_lo, maskbias, lse, Dws,
                       dqkv, S, H, A, 0.125f, rpw, seed, thresh, dbias);
    hipLaunchKernelGGL(attn_bwd_dkv2_kernel<DROP>, grid, dim3(512), AT_LDS_BYTES, s, qkv, dctx, maskbias, lse, Dws, dqkv, S, H, A,
                       0.125f, rpw, seed, thresh, dbias);
    hipError_t e2 = hipGetLastError();
    return e2 == hipSuccess ? 0 : -(int)e2;
  }
  hipLaunchKernelGGL((attn_bwd_dq_kernel<DROP, RES>), grid, dim3(AT_NWB * 64), AT_LDS_BYTES, s, qkv, dctx, ctx, ctx_lo, maskbias, lse,
                     Dws, dqkv, S, H, A, 0.125f, rpw, seed, thresh, dbias);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<DROP>, grid, dim3(AT_NWB * 64), AT_LDS_BYTES, s, qkv, dctx, maskbias, lse, Dws, dqkv, S, H, A,
                     0.125f, rpw, seed, thresh, dbias);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
template <bool DROP>
static int launch_attn_bwd(const bf16_t* qkv, const bf16_t* ctx, const uint8_t* ctx_lo, const bf16_t* dctx, const float* maskbias,
                           const float* lse, float* Dws, bf16_t* dqkv, int B, int S, int H, int A, uint32_t seed, uint32_t thresh,
                           float* dbias, hipStream_t s) {
  if (ctx_lo) return launch_attn_bwd2<DROP, true>(qkv, ctx, ctx_lo, dctx, maskbias, lse, Dws, dqkv, B, S, H, A, seed, thresh, dbias, s);
  return launch_attn_bwd2<DROP, false>(qkv, ctx, ctx_lo, dctx, maskbias, lse, Dws, dqkv, B, S, H, A, seed, thresh, dbias, s);
}


extern "C" {

// qkv bf16 [B*S, 3H] ; maskbias f32 [B,S] ; ctx bf16 [B*S, H] ; lse f32 [B, A, S]
// constraints: H = A * 64, S % 64 == 0, 64 <= S <= 512.  drop_thresh = p * 2^32 (0 = no dropout), drop_seed: the site's seed.
// ctx_lo (nullable) B*S*H bytes: the rounding residual O - bf16(O) of the context, one e5m2 byte per element in a layout private
// to these kernels (at_res_block), for kbner_attn_bwd's D (see there).
int kbner_attn_fwd(const bf16_t* qkv, const float* maskbias, bf16_t* ctx, uint8_t* ctx_lo, float* lse, int B, int S, int H, int A,
                   uint32_t drop_seed, uint32_t drop_thresh, void* stream) {
  KBNER_CHECK_ARG(B > 0 && A > 0 && H == A * AT_D && S % 64 == 0 && S >= 64 && S <= AT_MAXS);
  const int rpw = pick_rpw(B, S, A);
  hipStream_t st = (hipStream_t)stream;
#ifdef KBNER_ATTN_LAB
  // (the streaming kernel counts its own stores and does not write ctx_lo: with a residual requested the panel kernels run)
  if (!ctx_lo && ((attn_variant() == 3 && S >= 256 && B * A >= at_cu_count()) || attn_variant() == 4))   // 4: forced (small cases)
    return kbner_attn_fwd3(qkv, maskbias, ctx, lse, B, S, H, A, drop_seed, drop_thresh, st);
  const bool rows32 = attn_variant() != 1;
#else
  const bool rows32 = true;   // 32 query rows per wave and pass wherever launch_attn_fwd can use them (no dropout, >= 256 rows)
#endif
  switch (S / 64) {
    case 1: return launch_attn_fwd<4>(qkv, maskbias, ctx, ctx_lo, lse, B, H, A, rpw, drop_seed, drop_thresh, st, rows32);
    case 2: return launch_attn_fwd<8>(qkv, maskbias, ctx, ctx_lo, lse, B, H, A, rpw, drop_seed, drop_thresh, st, rows32);
    case 3: return launch_attn_fwd<12>(qkv, maskbias, ctx, ctx_lo, lse, B, H, A, rpw, drop_seed, drop_thresh, st, rows32);
    case 4: return launch_attn_fwd<16>(qkv, maskbias, ctx, ctx_lo, lse, B, H, A, rpw, drop_seed, drop_thresh, st, rows32);
    case 5: return launch_attn_fwd<20>(qkv, maskbias, ctx, ctx_lo, lse, B, H, A, rpw, drop_seed, drop_thresh, st, rows32);
    case 6: return launch_attn_fwd<24>(qkv, maskbias, ctx, ctx_lo, lse, B, H, A, rpw, drop_seed, drop_thresh, st, rows32);
    case 7: return launch_attn_fwd<28>(qkv, maskbias, ctx, ctx_lo, lse, B, H, A, rpw, drop_seed, drop_thresh, st, rows32);
    default: return launch_attn_fwd<32>(qkv, maskbias, ctx, ctx_lo, lse, B, H, A, rpw, drop_seed, drop_thresh, st, rows32);
  }
}

// dctx bf16 [B*S,H] (dO) ; ctx (O) ; lse ; Dws f32 [B,A,S] workspace ; dqkv bf16 [B*S,3H] out ;
// dbias_qkv f32 [3H] (nullable): += column sums of dqkv = the QKV projection's bias gradient
// ctx_lo (nullable): the residual kbner_attn_fwd wrote.  The softmax-backward correction D = rowdot(dO, O) stands in for
// sum_j P dP; dS = P (dP - D) is then a difference of nearly equal numbers wherever the rows of V (and of K) of a head are
// nearly parallel -- deep layers of a freshly initialised encoder -- and the 2^-9 rounding of a bf16 O comes back multiplied by
// |O| / |V_j - O| and, in dQ, by |mean K| / |K_j - mean K| (the sum of dS over j is exactly 0 only for the exact D):
// 16 % of layer 23's query.weight gradient at L = 24, std 0.02, against 2.6 % with the residual (tests/selftest.py check_step,
// oracle/encoder.py _AttnCoreFlash reproduces both).  One byte is enough: what matters to the dot product is the absolute error
// of O's elements, and e5m2 of the residual (2 mantissa bits, scaled by 2^14) takes it from 2^-9 |O| to 2^-12 |O|.
// Null = D from the bf16 O alone.
int kbner_attn_bwd(const bf16_t* qkv, const bf16_t* ctx, const uint8_t* ctx_lo, const bf16_t* dctx, const float* maskbias, const float* lse,
                   float* Dws, bf16_t* dqkv, int B, int S, int H, int A, uint32_t drop_seed, uint32_t drop_thresh,
                   float* dbias_qkv, void* stream) {
  KBNER_CHECK_ARG(B > 0 && A > 0 && H == A * AT_D && S % 64 == 0 && S >= 64 && S <= AT_MAXS);
  hipStream_t s = (hipStream_t)stream;
  if (drop_thresh) return launch_attn_bwd<true>(qkv, ctx, ctx_lo, dctx, maskbias, lse, Dws, dqkv, B, S, H, A, drop_seed, drop_thresh, dbias_qkv, s);
  return launch_attn_bwd<false>(qkv, ctx, ctx_lo, dctx, maskbias, lse, Dws, dqkv, B, S, H, A, drop_seed, drop_thresh, dbias_qkv, s);
}

}  // extern "C"
