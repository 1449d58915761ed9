// Attention, third structure (round 3): software-pipelined kernels with 32 stationary rows per wave.
//
// Why a third structure.  rocprofv3 --pmc on the round-2 kernels (profiles/round2_attn_pmc.txt) showed neither pipe
// saturated (MFMA busy 27-33 %, VALU issue 24-26 %): per wave the work ran as SERIAL phases -- all Q.K^T MFMAs of a row
// block, then ~250 exp2 / fma in a row, then all P.V MFMAs -- so the matrix pipe idled through every softmax phase unless
// the second wave of the SIMD happened to be out of phase, and with 16 stationary query rows every 1-KiB LDS fragment read
// fed ONE 16-cycle MFMA (LDS array ~100 % busy at the MFMA rate).  Here
//   * a wave owns 32 query rows (two 16-row blocks): every K / V^T fragment read from LDS feeds two MFMAs;
//   * keys are walked in 64-key blocks with a running maximum (rescale only when a row's maximum grows by more than
//     2^12: P stays <= 2^12, harmless in fp32 accumulators / bf16 operands), so the score tile of block kb+1 is computed by
//     the matrix pipe WHILE the vector pipe exponentiates block kb, and P.V of the first 32 keys runs under the exp2 of the
//     second 32 -- MFMA and VALU instructions alternate in the instruction stream of ONE wave instead of relying on the
//     partner wave's phase;
//   * whole heads per workgroup, persistent over (batch, head) items with the next head's K / mask DMA'd behind the last
//     Q.K^T and its V behind the last P.V (as in round 2); the mask row itself now arrives by LDS-DMA and the per-item
//     mask metadata (last unmasked key, number of mask-free leading blocks) is derived from the LDS copy, so no
//     compiler-visible global load sits between the DMA issue and its counted wait (hipcc would drain vmcnt(0) there).
// Arithmetic is unchanged: P = softmax(Q K^T / sqrt(d) + maskbias[key]), O = P V (transformers 3.0.0 BertSelfAttention,
// reached from flair/embeddings.py:3269); the row sums are fp32 sums of the unrounded probabilities.
#include "attn_common.h"
#include <cstdio>
#include <cstdlib>

#define A3_ROWS 256             // query rows per workgroup pass: 8 waves x 32
#define A3_THR2 12.0f           // lazy-rescale threshold, log2 domain
#define A3_LDS_BYTES (2 * AT_MAXS * 128 + 2 * AT_MAXS * 4 + 2 * AT_MAXS * 4 + 64)

// one 1-KiB LDS-DMA piece of a mask row (S floats): lanes beyond the row are masked off
static __device__ __forceinline__ void stage_mask_row(const float* __restrict__ src, int S, float* dst, int wid, int lane) {
  const int npiece = (S * 4 + 1023) >> 10;
  if (wid < npiece) {
    const int i = wid * 256 + lane * 4;
    if (i < S) glds16(src + i, reinterpret_cast<unsigned char*>(dst) + wid * 1024);
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// VALU budget.  rocprofv3 --pmc on the first version of this kernel: 226 VALU-class instructions per 64-key block and wave at
// an average 4.6 issue cycles (plain fp32 VALU 4, v_exp_f32 8: gfx950 executes a wave64 fp32 instruction over four cycles
// unless it is a packed one) = ~1040 cycles against 512 cycles of MFMA: the kernel was VALU-bound 2:1, exactly like the
// round-2 kernels.  Everything per-score that is not the exponential itself is therefore moved off the vector pipe:
//   * the stationary Q fragments are multiplied by scale * log2(e) ONCE per pass (in registers, re-rounded to bf16), so the
//     score MFMAs produce log2-domain scores;
//   * their accumulators START at -m (the row's reference value), so P = exp2(acc): no subtract, no fma;
//   * no row maximum in the steady state: m is fixed after the first block at (block maximum + 2^A3_MARGIN headroom) and a
//     block is only CHECKED -- bit 14 of a bf16 (the exponent's top bit) is set iff the value is >= 2, so one OR over the
//     packed probabilities + one AND tests "some P >= 2" for the whole tile; if it fires (rare: a later key beats the first
//     block's maximum by more than 2^A3_MARGIN) the block is redone with a raised reference and O / the row sums rescaled;
//   * row sums by one MFMA per 32 keys against an all-ones A fragment (the sum of exactly the bf16 values that multiply V).
// Left on the vector pipe per score: one v_exp_f32 (8 cycles) + half a v_cvt_pk_bf16_f32 (2) -> ~340 cycles per block
// against 36 MFMAs = 576.
#define A3_MARGIN 8.0f

static __device__ __forceinline__ void a3_opaque(const unsigned char*& p) {
  unsigned a = (unsigned)(size_t)(lds_void*)p;
  asm volatile("" : "+v"(a));
  p = (const unsigned char*)(lds_void*)(size_t)a;
}

// Debug instrumentation (KBNER_ATTN_PROF=1): per-wave s_memtime stamps accumulated per code region, printed by the launcher.
struct A3Prof {
  unsigned long long last;
  unsigned long long acc[12];
};
template <bool PROF>
static __device__ __forceinline__ void a3_stamp(A3Prof& pr, int i) {
  if (PROF) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    pr.acc[i] += t - pr.last;
    pr.last = t;
  }
}

struct A3Fwd {
  f4v s[2][4];      // [row block][key fragment]: raw score sums q.k of the current 64-key block (+ mask / scale)
  f4v o[2][4];      // [row block][d block]: O^T accumulators
  f4v osum[2];      // row sums (every register of a lane holds the sum of query lane & 15)
  float m[2];       // row reference (log2 domain), query = lane & 15 of row block j
  bf16x8 ka[2][4];  // [k-step][key fragment]: K fragments of the block whose scores are computed next
};

static __device__ __forceinline__ float a3_rowmax(const f4v (&sc)[4]) {
  float m = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
#pragma unroll
  for (int kf = 1; kf < 4; ++kf) {
    m = fmaxf(fmaxf(m, sc[kf][0]), sc[kf][1]);
    m = fmaxf(fmaxf(m, sc[kf][2]), sc[kf][3]);
  }
  return group4_max(m);
}

// P = exp2(scale2 * score - m) of one 32-key chunk c of the current block, packed to bf16 B fragments (one per row block);
// returns the OR of the packed words (overflow check).  DROP: pbd = dropped probabilities (what multiplies V), pb = undropped
// (row sums).
template <bool DROP>
static __device__ __forceinline__ uint32_t a3_probs(const f4v (&sc)[2][4], int c, bf16x8 (&pb)[2], bf16x8 (&pbd)[2],
                                                    const uint32_t* sCk, int kb, int g, const uint32_t (&rk)[2],
                                                    uint32_t drop_thresh, float scale2, const float (&m)[2]) {
  uint32_t acc = 0u;
  uint32_t ck[2][4];
  if (DROP) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const uint4 t = *reinterpret_cast<const uint4*>(sCk + kb * 64 + (2 * c + f) * 16 + g * 4);
      ck[f][0] = t.x; ck[f][1] = t.y; ck[f][2] = t.z; ck[f][3] = t.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    f4v p[2], pd[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      // log2-domain score minus the row reference: two packed fmas per fragment (v_pk_fma_f32: two scores per issue)
      const f4v x = sc[j][2 * c + f];
      const f2v lo = (f2v){x[0], x[1]} * splat2(scale2) - splat2(m[j]);
      const f2v hi = (f2v){x[2], x[3]} * splat2(scale2) - splat2(m[j]);
      const float t[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(t[r]);
        p[f][r] = e;
        if (DROP) pd[f][r] = drop_keep(rk[j], ck[f][r], drop_thresh) ? e : 0.0f;
      }
    }
    union {
      uint32_t u[4];
      bf16x8 v;
    } w;
    w.u[0] = pack2bf(p[0][0], p[0][1]);
    w.u[1] = pack2bf(p[0][2], p[0][3]);
    w.u[2] = pack2bf(p[1][0], p[1][1]);
    w.u[3] = pack2bf(p[1][2], p[1][3]);
    acc |= (w.u[0] | w.u[1]) | (w.u[2] | w.u[3]);
    pb[j] = w.v;
    if (DROP) pbd[j] = pack_b(pd[0], pd[1]);
  }
  return acc;
}

// One 64-key block kb of a pass, as two phases per wave:
//   A (vector pipe)  the block's V^T fragment reads go out first, then exp2 / convert of the block's scores;
//   B (matrix pipe)  P.V and the row sums of this block, then the scores of block kb+1 (PF; K fragments already in st.ka;
//                    PFMASK: accumulators start from that block's mask values), then the K fragment reads of block kb+2 (KL).
// Every LDS read is thus issued a whole phase (~400 cycles) before the MFMA that consumes it, and the two waves of a SIMD
// settle into opposite phases (one exponentiates while the other owns the matrix pipe); nothing here needs the score tiles of
// two blocks at once, so the accumulators of block kb+1 reuse the registers of block kb.
template <bool PF, bool KL, bool PFMASK, bool DROP, bool PROF>
static __device__ __forceinline__ void a3_fwd_step(A3Prof& pr, A3Fwd& st, const bf16x8 (&qf)[2][2], const unsigned char*& kp0,
                                                   const unsigned char*& kp1, const unsigned char* (&vp)[4],
                                                   const float* sMask, const uint32_t* sCk, int kb, int g,
                                                   const uint32_t (&rk)[2], uint32_t drop_thresh, float scale2) {
  // kp0 / kp1: this lane's K fragment addresses (k-step 0 / 1) of block kb+2; vp[db]: its V^T addresses of block kb.  They are
  // advanced by one block per step so that every read below is base register + immediate offset.
  f4v(&sc)[2][4] = st.s;
  bf16x8 v0[4], v1[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    v0[db] = tr_at(vp[db], 0);
    v1[db] = tr_at(vp[db], 4096);
    vp[db] += 8192;
  }
  __builtin_amdgcn_sched_barrier(0);
  bf16x8 pb[2][2], pbd[2][2];
  uint32_t chk = a3_probs<DROP>(sc, 0, pb[0], pbd[0], sCk, kb, g, rk, drop_thresh, scale2, st.m);
  chk |= a3_probs<DROP>(sc, 1, pb[1], pbd[1], sCk, kb, g, rk, drop_thresh, scale2, st.m);
  if (__any((chk & 0x40004000u) != 0u)) {
    // some probability >= 2: raise the reference of the rows whose block maximum came within the headroom, rescale what has
    // been accumulated and redo the exponentials
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float d = fmaxf(0.0f, a3_rowmax(sc[j]) * scale2 - st.m[j] + A3_MARGIN);
      const float f = __builtin_amdgcn_exp2f(-d);
#pragma unroll
      for (int db = 0; db < 4; ++db) st.o[j][db] *= f;
      st.osum[j] *= f;
      st.m[j] += d;
    }
    a3_probs<DROP>(sc, 0, pb[0], pbd[0], sCk, kb, g, rk, drop_thresh, scale2, st.m);
    a3_probs<DROP>(sc, 1, pb[1], pbd[1], sCk, kb, g, rk, drop_thresh, scale2, st.m);
  }
  a3_stamp<PROF>(pr, 8);
  const s8v ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
  st.osum[0] = MFMA(ones, pb[0][0], st.osum[0]);
  st.osum[1] = MFMA(ones, pb[0][1], st.osum[1]);
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    st.o[0][db] = MFMA(v0[db], DROP ? pbd[0][0] : pb[0][0], st.o[0][db]);
    st.o[1][db] = MFMA(v0[db], DROP ? pbd[0][1] : pb[0][1], st.o[1][db]);
  }
  st.osum[0] = MFMA(ones, pb[1][0], st.osum[0]);
  st.osum[1] = MFMA(ones, pb[1][1], st.osum[1]);
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    st.o[0][db] = MFMA(v1[db], DROP ? pbd[1][0] : pb[1][0], st.o[0][db]);
    st.o[1][db] = MFMA(v1[db], DROP ? pbd[1][1] : pb[1][1], st.o[1][db]);
  }
  if (PF) {
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      f4v i0 = (f4v){0.f, 0.f, 0.f, 0.f};
      if (PFMASK) i0 = *reinterpret_cast<const f4v*>(sMask + (kb + 1) * 64 + kf * 16 + g * 4) * 8.0f;   // mask / scale (scale = 1/8)
      sc[0][kf] = MFMA(st.ka[0][kf], qf[0][0], i0);
      sc[1][kf] = MFMA(st.ka[0][kf], qf[1][0], i0);
    }
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      sc[0][kf] = MFMA(st.ka[1][kf], qf[0][1], sc[0][kf]);
      sc[1][kf] = MFMA(st.ka[1][kf], qf[1][1], sc[1][kf]);
    }
  }
  if (KL) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      st.ka[0][kf] = kc_at(kp0, kf * 2048);
      st.ka[1][kf] = kc_at(kp1, kf * 2048);
    }
    kp0 += 8192;
    kp1 += 8192;
  }
  a3_stamp<PROF>(pr, 9);
}

template <bool DROP, bool PROF>
__global__ __launch_bounds__(512, 2) void attn_fwd3_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ maskbias,
                                                           bf16_t* __restrict__ ctx, float* __restrict__ lse, int S, int H, int A,
                                                           float scale, uint32_t drop_seed, uint32_t drop_thresh, int nitems, int dbg,
                                                           unsigned long long* __restrict__ profout) {
  A3Prof pr;
  if (PROF) {
#pragma unroll
    for (int i = 0; i < 12; ++i) pr.acc[i] = 0;
    pr.last = __builtin_amdgcn_s_memtime();
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = smem + AT_MAXS * 128;
  float* sMaskB = reinterpret_cast<float*>(smem + 2 * AT_MAXS * 128);          // [2][AT_MAXS] raw mask rows (double-buffered)
  uint32_t* sCkB = reinterpret_cast<uint32_t*>(sMaskB + 2 * AT_MAXS);           // [2][AT_MAXS] dropout column keys
  int* sMeta = reinterpret_cast<int*>(sCkB + 2 * AT_MAXS);                      // [2][2]: {1 + last unmasked key, #unmasked keys}
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int ld = 3 * H;
  const float scale2 = scale * 1.4426950408889634f;
  const float dscale = DROP ? drop_scale(drop_thresh) : 1.0f;
  const int npass = (S + A3_ROWS - 1) / A3_ROWS;
  const int npiece = S / 64;   // DMA pieces per wave and panel

  int item = (int)blockIdx.x;
  int slot = 0;
  if (tid < 4) sMeta[tid] = 0;
  {
    const int h = item % A, b = item / A;
    stage_mask_row(maskbias + (size_t)b * S, S, sMaskB, wid, lane);
    stage_panel(qkv + (size_t)b * S * ld + h * AT_D + H, ld, S, sK, wid, lane);
    stage_panel(qkv + (size_t)b * S * ld + h * AT_D + 2 * H, ld, S, sV, wid, lane);
  }
  bf16x8 qf[2][2];   // the pass's stationary Q fragments (loaded for the NEXT pass as soon as the last score MFMA has issued)
  {
    const int h = item % A, b = item / A;
    const bf16_t* base = qkv + (size_t)b * S * ld + h * AT_D;
    const int q0 = wid * 32 < S ? wid * 32 : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      qf[j][0] = glb_frag(base, ld, q0 + j * 16, 0, lane);
      qf[j][1] = glb_frag(base, ld, q0 + j * 16, 1, lane);
    }
  }
  for (;;) {   // items of this workgroup
    const int h = item % A, b = item / A;
    const uint32_t bhS = (uint32_t)((b * A + h) * S);
    const int next = item + (int)gridDim.x;
    const bool has_next = next < nitems;
    const float* sMask = sMaskB + slot * AT_MAXS;
    uint32_t* sCk = sCkB + slot * AT_MAXS;
    // K, the mask row (and this item's first Q fragments) have landed when at most the V pieces are still in flight
    a3_stamp<PROF>(pr, 10);
    wait_vm(npiece);
    __syncthreads();
    a3_stamp<PROF>(pr, 0);
    if (tid < S) {   // mask metadata of this item from the LDS copy of its mask row
      const bool un = sMask[tid] > -1.0f;   // additive bias 0 = attend (anything near -10000 = masked)
      const unsigned long long bal = __ballot(un);
      if (lane == 0 && bal) {
        atomicMax(&sMeta[slot * 2], wid * 64 + 64 - __builtin_clzll(bal));
        atomicAdd(&sMeta[slot * 2 + 1], __builtin_popcountll(bal));
      }
      if (DROP) sCk[tid] = drop_colkey(drop_seed, bhS + (uint32_t)tid);
    }
    if (tid < 2) sMeta[(slot ^ 1) * 2 + tid] = 0;   // the next item's slot (its atomics come after the next barriers)
    for (int pass = 0; pass < npass; ++pass) {
      const int q0 = pass * A3_ROWS + wid * 32;
      const bool active = q0 < S;
      const bool last_pass = pass == npass - 1;
      A3Fwd st;
      uint32_t rk[2] = {0u, 0u};
      // per-lane LDS fragment addresses (see kc_frag / tr_frag), recomputed per pass (kept live across passes they are spilled):
      // K rows f*16 + li -> + f*2048 ; V rows kc*32 + g*4 + (li>>2) -> + kc*4096.  kp / vp are the running copies of a3_fwd_step.
      const unsigned char* kb0 = sK + li * 128 + (((0 * 4 + g) ^ kc_swz(li)) << 4);
      const unsigned char* kb1 = sK + li * 128 + (((1 * 4 + g) ^ kc_swz(li)) << 4);
      const int vrow = g * 4 + (li >> 2);
      const unsigned char* vp[4];
#pragma unroll
      for (int db = 0; db < 4; ++db)
        vp[db] = sV + vrow * 128 + (((db * 2 + ((li & 3) >> 1)) ^ kc_swz(vrow)) << 4) + ((li & 1) << 3);
      const unsigned char* kp0 = kb0 + 2 * 8192;
      const unsigned char* kp1 = kb1 + 2 * 8192;
      // opaque to the optimiser: otherwise it splits "lane part + 0x10000 (sV) + block offset" and re-adds the constant before
      // every read (24 v_add_u32 per block) instead of using the instructions' 16-bit immediate offsets
      a3_opaque(kp0);
      a3_opaque(kp1);
#pragma unroll
      for (int db = 0; db < 4; ++db) a3_opaque(vp[db]);
      if (active) {
        // score tile of block 0 (mask added below, once the metadata is published)
        const f4v zero4 = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
          const bf16x8 a0 = kc_at(kb0, kf * 2048), a1 = kc_at(kb1, kf * 2048);
          st.s[0][kf] = MFMA(a0, qf[0][0], zero4);
          st.s[1][kf] = MFMA(a0, qf[1][0], zero4);
          st.s[0][kf] = MFMA(a1, qf[0][1], st.s[0][kf]);
          st.s[1][kf] = MFMA(a1, qf[1][1], st.s[1][kf]);
        }
      }
      a3_stamp<PROF>(pr, 1);
      if (pass == 0) {   // V (and the metadata atomics) from here on
        wait_vm(0);
        __syncthreads();
      }
      a3_stamp<PROF>(pr, 2);
      int klen = sMeta[slot * 2];
      int nfree = (sMeta[slot * 2 + 1] == klen) ? (klen >> 6) : 0;   // leading 64-key blocks without a masked key (prefix masks)
      if (klen == 0) {   // every key masked: the reference's softmax is then over the masked scores themselves
        klen = S;
        nfree = 0;
      }
      const int nkb = (klen + 63) >> 6;   // blocks behind the last unmasked key contribute exp(-10000 + x) = 0 exactly
      if (active) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (DROP) rk[j] = drop_rowkey(drop_seed, bhS + (uint32_t)(q0 + j * 16 + li));
          st.osum[j] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int db = 0; db < 4; ++db) st.o[j][db] = (f4v){0.f, 0.f, 0.f, 0.f};
        }
        if (nkb > 1) {
#pragma unroll
          for (int kf = 0; kf < 4; ++kf) {
            st.ka[0][kf] = kc_at(kb0, 8192 + kf * 2048);
            st.ka[1][kf] = kc_at(kb1, 8192 + kf * 2048);
          }
        }
        if (nfree == 0) {
#pragma unroll
          for (int kf = 0; kf < 4; ++kf) {
            const f4v mk = *reinterpret_cast<const f4v*>(sMask + kf * 16 + g * 4) * 8.0f;
            st.s[0][kf] += mk;
            st.s[1][kf] += mk;
          }
        }
        // the row reference: first block's maximum + headroom
#pragma unroll
        for (int j = 0; j < 2; ++j) st.m[j] = a3_rowmax(st.s[j]) * scale2 + A3_MARGIN;
#define A3_STEP(PF, KL, MK, KB) a3_fwd_step<PF, KL, MK, DROP, PROF>(pr, st, qf, kp0, kp1, vp, sMask, sCk, (KB), g, rk, drop_thresh, scale2)
        // main loop: every step computes the next block's scores and loads the next-but-one block's K fragments
        int kb = 0;
        for (; kb + 2 < nkb && kb + 1 < nfree; ++kb) A3_STEP(true, true, false, kb);
        for (; kb + 2 < nkb; ++kb) A3_STEP(true, true, true, kb);
      }
      a3_stamp<PROF>(pr, 3);
      // K is dead from here on (its last fragments are in registers): in the last pass, the next item's mask row and K panel
      if (last_pass && has_next) {
        const int nb = next / A, nh = next % A;
        __syncthreads();
        stage_mask_row(maskbias + (size_t)nb * S, S, sMaskB + (slot ^ 1) * AT_MAXS, wid, lane);
        if (!(dbg & 1)) stage_panel(qkv + (size_t)nb * S * ld + nh * AT_D + H, ld, S, sK, wid, lane);
      }
      a3_stamp<PROF>(pr, 4);
      if (active && nkb > 1) {
        if (nkb - 1 < nfree) A3_STEP(true, false, false, nkb - 2); else A3_STEP(true, false, true, nkb - 2);
      }
      // the last score MFMA of the pass has issued: Q fragments of the next pass / the next item's first pass
      {
        int nb = b, nh = h, nq0 = q0 + A3_ROWS;
        if (last_pass) {
          nb = has_next ? next / A : b;
          nh = has_next ? next % A : h;
          nq0 = wid * 32;
        }
        if (nq0 >= S) nq0 = 0;
        bf16x8 qn[2][2];
        const bf16_t* nbase = qkv + (size_t)nb * S * ld + nh * AT_D;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          qn[j][0] = glb_frag(nbase, ld, nq0 + j * 16, 0, lane);
          qn[j][1] = glb_frag(nbase, ld, nq0 + j * 16, 1, lane);
        }
        a3_stamp<PROF>(pr, 5);
        if (active) {
          A3_STEP(false, false, false, nkb - 1);
          a3_stamp<PROF>(pr, 6);
#undef A3_STEP
          // O^T fragment: lane holds O[q0 + j*16 + li][db*16 + g*4 .. +3]
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float sum = st.osum[j][0];
            const float inv = dscale / sum;
            bf16_t* orow = ctx + (size_t)(b * S + q0 + j * 16 + li) * H + h * AT_D;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
              uint2 u;
              u.x = pack2bf(st.o[j][db][0] * inv, st.o[j][db][1] * inv);
              u.y = pack2bf(st.o[j][db][2] * inv, st.o[j][db][3] * inv);
              *reinterpret_cast<uint2*>(orow + db * 16 + g * 4) = u;
            }
            if (g == 0) lse[((size_t)b * A + h) * S + q0 + j * 16 + li] = (st.m[j] + __log2f(sum)) * 0.6931471805599453f;
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          qf[j][0] = qn[j][0];
          qf[j][1] = qn[j][1];
        }
      }
    }
    a3_stamp<PROF>(pr, 7);
    if (!has_next) break;
    __syncthreads();   // every wave is done with sV: the next head's V panel
    item = next;
    slot ^= 1;
    if (!(dbg & 1)) stage_panel(qkv + (size_t)(item / A) * S * ld + (item % A) * AT_D + 2 * H, ld, S, sV, wid, lane);
    a3_stamp<PROF>(pr, 11);
  }
  if (PROF && blockIdx.x < 4 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) profout[(blockIdx.x * 8 + wid) * 12 + i] = pr.acc[i];
  }
}

static int a3_set_lds(const void* f, int bytes) {
  hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e == hipSuccess ? 0 : -(int)e;
}

static int a3_cu_count() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    return prop.multiProcessorCount;
  return 256;
}

template <bool DROP>
static int launch_fwd3(const bf16_t* qkv, const float* maskbias, bf16_t* ctx, float* lse, int B, int S, int H, int A, uint32_t seed,
                       uint32_t thresh, hipStream_t stream) {
  static int ncu = 0, dbg = 0, prof = 0;
  static unsigned long long* profbuf = nullptr;
  if (ncu == 0) {
    int r = a3_set_lds(reinterpret_cast<const void*>(attn_fwd3_kernel<DROP, false>), A3_LDS_BYTES);
    if (r) return r;
    r = a3_set_lds(reinterpret_cast<const void*>(attn_fwd3_kernel<DROP, true>), A3_LDS_BYTES);
    if (r) return r;
    const char* e = getenv("KBNER_ATTN_DBG");
    dbg = e ? atoi(e) : 0;
    e = getenv("KBNER_ATTN_PROF");
    prof = e ? atoi(e) : 0;
    if (prof && hipMalloc(&profbuf, 4 * 8 * 12 * sizeof(unsigned long long)) != hipSuccess) prof = 0;
    ncu = a3_cu_count();
  }
  const int nitems = B * A;
  const int grid = nitems < ncu ? nitems : ncu;
  if (prof) {
    hipLaunchKernelGGL((attn_fwd3_kernel<DROP, true>), dim3(grid), dim3(512), A3_LDS_BYTES, stream, qkv, maskbias, ctx, lse, S, H, A,
                       0.125f, seed, thresh, nitems, dbg, profbuf);
    if (prof == 1) {   // print once: region cycles per wave of workgroups 0..3
      prof = 2;
      unsigned long long h[4 * 8 * 12];
      if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h, profbuf, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
        static const char* nm[12] = {"Kwait+bar", "S0+epi", "Vwait+bar", "loopres", "Kdead+dma", "Qload", "lastres", "epi", "phaseA", "phaseB",
                                     "top", "Vdead+dma"};
        for (int w = 0; w < 32; w += 5) {
          fprintf(stderr, "fwd3 prof wg%d wave%d:", w / 8, w % 8);
          for (int i = 0; i < 12; ++i) fprintf(stderr, " %s=%llu", nm[i], h[w * 12 + i]);
          fprintf(stderr, "\n");
        }
      }
    }
  } else {
    hipLaunchKernelGGL((attn_fwd3_kernel<DROP, false>), dim3(grid), dim3(512), A3_LDS_BYTES, stream, qkv, maskbias, ctx, lse, S, H, A,
                       0.125f, seed, thresh, nitems, dbg, (unsigned long long*)nullptr);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// entry point used by attention.hip's dispatcher (same contract as kbner_attn_fwd)
int kbner_attn_fwd3(const bf16_t* qkv, const float* maskbias, bf16_t* ctx, float* lse, int B, int S, int H, int A,
                    uint32_t drop_seed, uint32_t drop_thresh, hipStream_t stream) {
  if (drop_thresh) return launch_fwd3<true>(qkv, maskbias, ctx, lse, B, S, H, A, drop_seed, drop_thresh, stream);
  return launch_fwd3<false>(qkv, maskbias, ctx, lse, B, S, H, A, drop_seed, drop_thresh, stream);
}
