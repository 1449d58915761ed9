// Attention backward, third structure (round 5): the arithmetic of attn_bwd_dq2 / attn_bwd_dkv2 (attention.hip), bit for bit, on a
// SOFTWARE-PIPELINED, PINNED instruction stream.  Included by attention.hip (needs stage_panel, stage_mask_klen, flush_colsum ...).
//
// Why.  tools/micro/valu_probe.hip (profiles/round3_valu_probe.txt) measured how a gfx950 SIMD issues: an MFMA stream starves every
// plain VALU instruction of BOTH resident waves except about one per 16x16x32 MFMA (two per 32x32x16), which is free; beyond that a
// VALU-class instruction costs its ~4 issue cycles whichever wave it comes from (v_exp 4.4 when interleaved).  hipcc schedules the
// round-2 loops as a pure-MFMA block followed by a pure-VALU block per 32-key chunk (tools/isa_sched.py: 24 M, then ~45 v + 16 e),
// so nothing hides: the measured 672 cycles per chunk and wave of attn_bwd_dq2 ARE 24 x 16.4 + 45 x 4 + 16 x 6.  The interleaved
// GEMM loop of round 4 (gemm256.hip) showed how to get the free slots: one filler per MFMA, pinned with sched_barrier.
//
// How.  The unit of the pipeline is a HALF chunk (i, j): the 32 keys (queries) of chunk i against the wave's j-th 16-row block.
// Step (i, j) issues, interleaved one MFMA : ~2.5 VALU,
//     MFMA   scores S / dP of the NEXT half chunk          (i, 1) in an even step, (i + 1, 0) in an odd step        8 MFMAs
//     MFMA   dQ (dK, dV) of the PREVIOUS half chunk        (i - 1, 1) resp. (i, 0)                                  4 (8) MFMAs
//     VALU   softmax backward of THIS half chunk           P = exp2(scale2 S), dS = P dP, bf16 packs              28 (32) instructions
// The two row blocks j = 0, 1 that the round-2 kernels already hold ARE the two pipeline stages, so the register footprint does not
// grow.  Every LDS fragment register is re-loaded (for chunk i + 1) right behind the MFMA that used it last, in an even step:
// twelve or more MFMA slots before its next use.  Chunk loop bodies are branch-free and identical (the first step's dQ MFMAs add
// zeros, the last step scores one chunk past the end -- LDS inside the allocation, results never read): 12 of 396 MFMAs per pass.
#pragma once

#define A3_SB() __builtin_amdgcn_sched_barrier(0)

// fragment reads from 32-bit LDS addresses (bases kept as opaque VGPR values, offsets as the instructions' immediates)
static __device__ __forceinline__ unsigned a3_lds(const void* p) { return (unsigned)(size_t)(lds_void*)p; }
static __device__ __forceinline__ bf16x8 a3_kc(unsigned base, unsigned off) {
  typedef const s8v __attribute__((address_space(3))) lds_s8v_t;
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<lds_s8v_t*>((size_t)(base + off)));
}
static __device__ __forceinline__ f4v a3_f4(unsigned base, unsigned off) {
  typedef const f4v __attribute__((address_space(3))) lds_f4v_t;
  return *reinterpret_cast<lds_f4v_t*>((size_t)(base + off));
}
static __device__ __forceinline__ bf16x8 a3_tr(unsigned base, unsigned off) {
  typedef s4v __attribute__((address_space(3))) lds_s4v_t;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_t*)(size_t)(base + off));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_t*)(size_t)(base + off + 16u * 128u));
  s8v v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// one softmax-backward unit = two scores (elements 2 RP, 2 RP + 1 of accumulator s<H>[J]) in four stages
// (the empty asm after a stage makes its results opaque AT THAT POINT of the instruction stream: the arithmetic is pure, so without
// it instruction selection is free to gather all of a step's VALU work in front of the step's first MFMA -- which it does --
// whatever sched_barriers stand between the source statements)
#define A3_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b));
#define A3_UA(U, SV, RP) float x0##U = SV[2 * (RP)] * scale2, x1##U = SV[2 * (RP) + 1] * scale2; A3_PIN2(x0##U, x1##U)
#define A3_UB(U) float e0##U = __builtin_amdgcn_exp2f(x0##U), e1##U = __builtin_amdgcn_exp2f(x1##U); A3_PIN2(e0##U, e1##U)
#define A3_UC(U, PV, RP) float d0##U = e0##U * PV[2 * (RP)], d1##U = e1##U * PV[2 * (RP) + 1]; A3_PIN2(d0##U, d1##U)
#define A3_PK(dst, a, b) dst = pack2bf(a, b); asm volatile("" : "+v"(dst));

// ---------------------------------------------------------------------------------------------------------------------------
// dQ.  Same contract, grid and LDS layout as attn_bwd_dq2_kernel<false, RES> (no dropout).
template <bool RES>
__global__ __launch_bounds__(512) void attn_bwd_dq3_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                          const bf16_t* __restrict__ ctx, const uint8_t* __restrict__ ctx_lo,
                                                          const float* __restrict__ maskbias,
                                                          const float* __restrict__ lse, float* __restrict__ Dv,
                                                          bf16_t* __restrict__ dqkv, int S, int H, int A, float scale, int rpw,
                                                          float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[8][64];
  __shared__ int sKlen[2];
  unsigned char* sK = smem;
  unsigned char* sV = smem + AT_MAXS * 128;
  float* sMask = reinterpret_cast<float*>(smem + 2 * AT_MAXS * 128);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int ld = 3 * H;
  const bf16_t* base = qkv + (size_t)b * S * ld + h * AT_D;
  f4v bsum[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) bsum[db] = (f4v){0.f, 0.f, 0.f, 0.f};
  stage_panel<8>(base + H, ld, S, sK, wid, lane);
  stage_panel<8>(base + 2 * H, ld, S, sV, wid, lane);
  int nfree;
  const int klen = stage_mask_klen<512>(maskbias, (size_t)b * S, S, 1.0f / scale, sMask, sKlen, tid, nfree);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int g = lane >> 4, li = lane & 15;
  const int nkc = ((klen + 63) >> 6) << 1;   // key chunks holding an unmasked key, rounded up to the loop's two chunks per trip
  const float scale2 = scale * 1.4426950408889634f;
  const float oscale = scale;
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
      l_q[j] = -lse_q[j] / scale;   // accumulator start of the score MFMAs (attn_bwd_dq2_kernel)
      float d_part = dot8(dof[j][0], of[j][0]) + dot8(dof[j][1], of[j][1]);
      if (RES) d_part += res_dot16(dof[j][0], dof[j][1], rw[j]);
      d_q[j] = group4_sum(d_part);
      if (g == 0) Dv[sidx] = d_q[j];
#pragma unroll
      for (int db = 0; db < 4; ++db) dq[j][db] = zero4;
    }
    const f4v sinit[2] = {(f4v){l_q[0], l_q[0], l_q[0], l_q[0]}, (f4v){l_q[1], l_q[1], l_q[1], l_q[1]}};
    const f4v pinit[2] = {(f4v){-d_q[0], -d_q[0], -d_q[0], -d_q[0]}, (f4v){-d_q[1], -d_q[1], -d_q[1], -d_q[1]}};
    // running per-lane LDS bases (advanced by two chunks = 8 KiB per trip; everything inside a trip is an immediate offset)
    unsigned bk0 = a3_lds(pK.kc[0]), bk1 = a3_lds(pK.kc[1]), bv0 = a3_lds(pV.kc[0]), bv1 = a3_lds(pV.kc[1]);
    unsigned bt0 = a3_lds(pK.tr[0]), bt1 = a3_lds(pK.tr[1]), bt2 = a3_lds(pK.tr[2]), bt3 = a3_lds(pK.tr[3]);
    const float* pm = sMask + g * 4;
    // (opaque: left to itself hipcc keeps ONE base per panel pair and re-derives the others with a v_add in front of every read --
    // the V panel sits 64 KiB behind the K panel, which no ds_read immediate reaches)
    asm volatile("" : "+v"(bk0), "+v"(bk1), "+v"(bv0), "+v"(bv1));
    asm volatile("" : "+v"(bt0), "+v"(bt1), "+v"(bt2), "+v"(bt3));
    // fragments of chunk 0 and the scores of half chunk (0, 0)
    bf16x8 kA0 = a3_kc(bk0, 0), kA1 = a3_kc(bk1, 0), kB0 = a3_kc(bk0, 2048), kB1 = a3_kc(bk1, 2048);
    bf16x8 vA0 = a3_kc(bv0, 0), vA1 = a3_kc(bv1, 0), vB0 = a3_kc(bv0, 2048), vB1 = a3_kc(bv1, 2048);
    bf16x8 t0 = a3_tr(bt0, 0), t1 = a3_tr(bt1, 0), t2 = a3_tr(bt2, 0), t3 = a3_tr(bt3, 0);
    f4v sA[2], sB[2], pA[2], pB[2];
    sA[0] = MFMA(kA0, qf[0][0], sinit[0]);
    sB[0] = MFMA(kB0, qf[0][0], sinit[0]);
    pA[0] = MFMA(vA0, dof[0][0], pinit[0]);
    pB[0] = MFMA(vB0, dof[0][0], pinit[0]);
    sA[0] = MFMA(kA1, qf[0][1], sA[0]);
    sB[0] = MFMA(kB1, qf[0][1], sB[0]);
    pA[0] = MFMA(vA1, dof[0][1], pA[0]);
    pB[0] = MFMA(vB1, dof[0][1], pB[0]);
    sA[1] = sB[1] = pA[1] = pB[1] = zero4;
    uint32_t w0[4] = {0u, 0u, 0u, 0u}, w1[4] = {0u, 0u, 0u, 0u};   // dS of row block 0 / 1 as packed bf16 (the dQ MFMAs' B operand)
    A3_SB();

    // one step.  J: the row block whose softmax backward runs; O = 1 - J: the row block whose scores / dQ MFMAs run.
    // RL (even steps): re-load every fragment behind its last use -- K / V fragments from CN (the next chunk's offset), the
    // transposed K fragments from CT (this chunk's offset: they serve dQ of (i, 0) and (i, 1), one and two steps later).
#define A3_DQ_STEP(J, O, WJ, WO, RL, CN, CT, TAIL)                                                                           \
  {                                                                                                                          \
    bf16x8 dsb_;                                                                                                             \
    {                                                                                                                        \
      union { uint32_t u[4]; bf16x8 v; } c_;                                                                                 \
      c_.u[0] = WO[0]; c_.u[1] = WO[1]; c_.u[2] = WO[2]; c_.u[3] = WO[3];                                                     \
      dsb_ = c_.v;                                                                                                           \
    }                                                                                                                        \
    sA[O] = MFMA(kA0, qf[O][0], sinit[O]); A3_SB();                                                                          \
    A3_UA(a, sA[J], 0) if (RL) kA0 = a3_kc(bk0, (CN)); A3_SB();                                                              \
    sB[O] = MFMA(kB0, qf[O][0], sinit[O]); A3_SB();                                                                          \
    A3_UB(a) if (RL) kB0 = a3_kc(bk0, (CN) + 2048); A3_SB();                                                                 \
    pA[O] = MFMA(vA0, dof[O][0], pinit[O]); A3_SB();                                                                         \
    A3_UC(a, pA[J], 0) A3_UA(b, sA[J], 1) if (RL) vA0 = a3_kc(bv0, (CN)); A3_SB();                                            \
    pB[O] = MFMA(vB0, dof[O][0], pinit[O]); A3_SB();                                                                         \
    A3_PK(WJ[0], d0a, d1a) A3_UB(b) if (RL) vB0 = a3_kc(bv0, (CN) + 2048); A3_SB();                                       \
    sA[O] = MFMA(kA1, qf[O][1], sA[O]); A3_SB();                                                                             \
    A3_UC(b, pA[J], 1) A3_UA(c, sB[J], 0) if (RL) kA1 = a3_kc(bk1, (CN)); A3_SB();                                            \
    sB[O] = MFMA(kB1, qf[O][1], sB[O]); A3_SB();                                                                             \
    A3_PK(WJ[1], d0b, d1b) A3_UB(c) if (RL) kB1 = a3_kc(bk1, (CN) + 2048); A3_SB();                                       \
    pA[O] = MFMA(vA1, dof[O][1], pA[O]); A3_SB();                                                                            \
    A3_UC(c, pB[J], 0) A3_UA(d, sB[J], 1) if (RL) vA1 = a3_kc(bv1, (CN)); A3_SB();                                            \
    pB[O] = MFMA(vB1, dof[O][1], pB[O]); A3_SB();                                                                            \
    A3_PK(WJ[2], d0c, d1c) A3_UB(d) if (RL) vB1 = a3_kc(bv1, (CN) + 2048); A3_SB();                                       \
    dq[O][0] = MFMA(t0, dsb_, dq[O][0]); A3_SB();                                                                            \
    A3_UC(d, pB[J], 1) if (RL) t0 = a3_tr(bt0, (CT)); A3_SB();                                                               \
    dq[O][1] = MFMA(t1, dsb_, dq[O][1]); A3_SB();                                                                            \
    A3_PK(WJ[3], d0d, d1d) if (RL) t1 = a3_tr(bt1, (CT)); A3_SB();                                                       \
    dq[O][2] = MFMA(t2, dsb_, dq[O][2]); A3_SB();                                                                            \
    if (RL) t2 = a3_tr(bt2, (CT)); TAIL A3_SB();                                                                             \
    dq[O][3] = MFMA(t3, dsb_, dq[O][3]); A3_SB();                                                                            \
    if (RL) t3 = a3_tr(bt3, (CT)); A3_SB();                                                                                  \
  }
    // chunks that hold a masked key (at most the last two for prefix masks) add mask / scale to the raw scores first
#define A3_DQ_MASK(J, C)                                                   \
  if ((C) >= nfree) {                                                      \
    const f4v m0_ = *reinterpret_cast<const f4v*>(pm + (C) * 32);          \
    const f4v m1_ = *reinterpret_cast<const f4v*>(pm + (C) * 32 + 16);     \
    sA[J] += m0_;                                                          \
    sB[J] += m1_;                                                          \
    A3_SB();                                                               \
  }
#define A3_NONE
#define A3_DQ_BUMP                                                                                      \
  bk0 += 8192; bk1 += 8192; bv0 += 8192; bv1 += 8192; bt0 += 8192; bt1 += 8192; bt2 += 8192; bt3 += 8192; \
  asm volatile("" : "+v"(bk0), "+v"(bk1), "+v"(bv0), "+v"(bv1));                                         \
  asm volatile("" : "+v"(bt0), "+v"(bt1), "+v"(bt2), "+v"(bt3));
    for (int kc = 0; kc < nkc; kc += 2) {
      A3_DQ_MASK(0, kc)
      A3_DQ_STEP(0, 1, w0, w1, true, 4096, 0, A3_NONE)         // softmax (kc, 0) | scores (kc, 1), dQ (kc - 1, 1)
      A3_DQ_MASK(1, kc)
      A3_DQ_STEP(1, 0, w1, w0, false, 0, 0, A3_NONE)           // softmax (kc, 1) | scores (kc + 1, 0), dQ (kc, 0)
      A3_DQ_MASK(0, kc + 1)
      A3_DQ_STEP(0, 1, w0, w1, true, 8192, 4096, A3_NONE)      // softmax (kc + 1, 0) | scores (kc + 1, 1), dQ (kc, 1)
      A3_DQ_MASK(1, kc + 1)
      A3_DQ_STEP(1, 0, w1, w0, false, 0, 0, A3_DQ_BUMP)        // softmax (kc + 1, 1) | scores (kc + 2, 0), dQ (kc + 1, 0)
    }
    {   // dQ of the last half chunk
      union { uint32_t u[4]; bf16x8 v; } c_;
      c_.u[0] = w1[0]; c_.u[1] = w1[1]; c_.u[2] = w1[2]; c_.u[3] = w1[3];
      dq[1][0] = MFMA(t0, c_.v, dq[1][0]);
      dq[1][1] = MFMA(t1, c_.v, dq[1][1]);
      dq[1][2] = MFMA(t2, c_.v, dq[1][2]);
      dq[1][3] = MFMA(t3, c_.v, dq[1][3]);
    }
#undef A3_DQ_STEP
#undef A3_DQ_MASK
#undef A3_DQ_BUMP
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

// ---------------------------------------------------------------------------------------------------------------------------
// dK, dV.  Same contract, grid and LDS layout as attn_bwd_dkv2_kernel<false> (no dropout).  A step is 16 MFMAs (8 scores of the
// next half chunk, 4 dV + 4 dK of the previous one) around the 32 VALU instructions of the current half chunk's softmax backward.
__global__ __launch_bounds__(512) void attn_bwd_dkv3_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                           const float* __restrict__ maskbias, const float* __restrict__ lse,
                                                           const float* __restrict__ Dv, bf16_t* __restrict__ dqkv, int S,
                                                           int H, int A, float scale, int rpw, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[8][64];
  f4v bsk[4], bsv[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) bsk[db] = bsv[db] = (f4v){0.f, 0.f, 0.f, 0.f};
  unsigned char* sQ = smem;
  unsigned char* sO = smem + AT_MAXS * 128;
  float* sL = reinterpret_cast<float*>(smem + 2 * AT_MAXS * 128);
  float* sD = sL + AT_MAXS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int ld = 3 * H;
  const bf16_t* base = qkv + (size_t)b * S * ld + h * AT_D;
  const bf16_t* dob = dctx + (size_t)b * S * H + h * AT_D;
  stage_panel<8>(base, ld, S, sQ, wid, lane);
  stage_panel<8>(dob, H, S, sO, wid, lane);
  const size_t sbase = ((size_t)b * A + h) * S;
  for (int i = tid; i < AT_MAXS + 32; i += 512) {   // (+ one chunk: the last step's scores read one chunk past the end)
    const bool in = i < S;
    if (i < AT_MAXS) {
      sL[i] = in ? -lse[sbase + i] / scale : 0.0f;   // accumulator start of the score MFMAs (attn_bwd_dq2_kernel)
      sD[i] = in ? -Dv[sbase + i] : 0.0f;
    }
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
    f4v dk[2][4], dv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kj = k0 + j * 16;
      kf[j][0] = glb_frag(base + H, ld, kj, 0, lane);
      kf[j][1] = glb_frag(base + H, ld, kj, 1, lane);
      vf[j][0] = glb_frag(base + 2 * H, ld, kj, 0, lane);
      vf[j][1] = glb_frag(base + 2 * H, ld, kj, 1, lane);
      mb[j] = maskbias[(size_t)b * S + kj + li] / scale;
#pragma unroll
      for (int db = 0; db < 4; ++db) dk[j][db] = dv[j][db] = zero4;
    }
    // a wave whose 32 keys are all masked has P = 0 exactly: dK = dV = 0 for its rows (wave-uniform skip of the loop)
    const bool live = __builtin_amdgcn_readfirstlane(__any((mb[0] > -8.0f) || (mb[1] > -8.0f)) ? 1 : 0) != 0;
    // none of the wave's 32 keys masked (every full-length sentence): the mask add disappears
    const bool masked = __builtin_amdgcn_readfirstlane(__any((mb[0] != 0.0f) || (mb[1] != 0.0f)) ? 1 : 0) != 0;
    if (live) {
      unsigned bq0 = a3_lds(pQ.kc[0]), bq1 = a3_lds(pQ.kc[1]), bo0 = a3_lds(pO.kc[0]), bo1 = a3_lds(pO.kc[1]);
      // the transposed fragments serve dV / dK of the PREVIOUS half chunk: their bases run one chunk behind (see the trips below)
      unsigned bq_t0 = a3_lds(pQ.tr[0]), bq_t1 = a3_lds(pQ.tr[1]), bq_t2 = a3_lds(pQ.tr[2]), bq_t3 = a3_lds(pQ.tr[3]);
      unsigned bo_t0 = a3_lds(pO.tr[0]), bo_t1 = a3_lds(pO.tr[1]), bo_t2 = a3_lds(pO.tr[2]), bo_t3 = a3_lds(pO.tr[3]);
      unsigned pl = a3_lds(sL + g * 4), pd = a3_lds(sD + g * 4);   // byte addresses of this lane group's -lse/scale, -D quads
      asm volatile("" : "+v"(bq0), "+v"(bq1), "+v"(bo0), "+v"(bo1));
      asm volatile("" : "+v"(pl), "+v"(pd));
      asm volatile("" : "+v"(bq_t0), "+v"(bq_t1), "+v"(bq_t2), "+v"(bq_t3));
      asm volatile("" : "+v"(bo_t0), "+v"(bo_t1), "+v"(bo_t2), "+v"(bo_t3));
      // fragments of query chunk 0 and the scores of half chunk (0, 0)
      bf16x8 qA0 = a3_kc(bq0, 0), qA1 = a3_kc(bq1, 0), qB0 = a3_kc(bq0, 2048), qB1 = a3_kc(bq1, 2048);
      bf16x8 oA0 = a3_kc(bo0, 0), oA1 = a3_kc(bo1, 0), oB0 = a3_kc(bo0, 2048), oB1 = a3_kc(bo1, 2048);
      f4v sl0 = a3_f4(pl, 0), sl1 = a3_f4(pl, 64);
      f4v nd0 = a3_f4(pd, 0), nd1 = a3_f4(pd, 64);
      f4v sA[2], sB[2], pA[2], pB[2];
      sA[0] = MFMA(qA0, kf[0][0], sl0);
      sB[0] = MFMA(qB0, kf[0][0], sl1);
      pA[0] = MFMA(oA0, vf[0][0], nd0);
      pB[0] = MFMA(oB0, vf[0][0], nd1);
      sA[0] = MFMA(qA1, kf[0][1], sA[0]);
      sB[0] = MFMA(qB1, kf[0][1], sB[0]);
      pA[0] = MFMA(oA1, vf[0][1], pA[0]);
      pB[0] = MFMA(oB1, vf[0][1], pB[0]);
      sA[1] = sB[1] = pA[1] = pB[1] = zero4;
      // P and dS of key block 0 / 1 as packed bf16 (the dV / dK MFMAs' B operands)
      uint32_t y0[4] = {0u, 0u, 0u, 0u}, y1[4] = {0u, 0u, 0u, 0u}, w0[4] = {0u, 0u, 0u, 0u}, w1[4] = {0u, 0u, 0u, 0u};
      A3_SB();
      // One step = 16 MFMAs.  This kernel has no registers to spare (238 without the pipeline), so what the dQ kernel keeps across
      // steps is fetched just in time here: each transposed fragment five MFMA slots before its dV / dK MFMA (every step: 16 8-byte
      // transpose reads), the next step's accumulator starts (-lse/scale, -D of ITS chunk: CS) behind the last four MFMAs.
      // KV = false (the very first step of a pass): no dV / dK MFMAs -- there is no previous half chunk.
      // RL (even steps): the Q / dO fragments are re-loaded for the next chunk (CN) right behind their last use.
#define A3_KV_STEP(J, O, YJ, WJ, YO, WO, KV, RL, CN, CT, CS, TAIL)                                                           \
  {                                                                                                                          \
    bf16x8 pb_, dsb_, to0, to1, to2, to3, tq0, tq1, tq2, tq3;                                                                \
    {                                                                                                                        \
      union { uint32_t u[4]; bf16x8 v; } c_;                                                                                 \
      c_.u[0] = YO[0]; c_.u[1] = YO[1]; c_.u[2] = YO[2]; c_.u[3] = YO[3];                                                     \
      pb_ = c_.v;                                                                                                            \
      c_.u[0] = WO[0]; c_.u[1] = WO[1]; c_.u[2] = WO[2]; c_.u[3] = WO[3];                                                     \
      dsb_ = c_.v;                                                                                                           \
    }                                                                                                                        \
    sA[O] = MFMA(qA0, kf[O][0], sl0); A3_SB();                                                                               \
    A3_UA(a, sA[J], 0) if (RL) qA0 = a3_kc(bq0, (CN)); A3_SB();                                                              \
    sB[O] = MFMA(qB0, kf[O][0], sl1); A3_SB();                                                                               \
    A3_UB(a) if (RL) qB0 = a3_kc(bq0, (CN) + 2048); A3_SB();                                                                 \
    pA[O] = MFMA(oA0, vf[O][0], nd0); A3_SB();                                                                               \
    A3_UC(a, pA[J], 0) if (RL) oA0 = a3_kc(bo0, (CN)); A3_SB();                                                              \
    pB[O] = MFMA(oB0, vf[O][0], nd1); A3_SB();                                                                               \
    A3_PK(YJ[0], e0a, e1a) A3_PK(WJ[0], d0a, d1a) if (RL) oB0 = a3_kc(bo0, (CN) + 2048);                                      \
    if (KV) to0 = a3_tr(bo_t0, (CT)); A3_SB();                                                                               \
    sA[O] = MFMA(qA1, kf[O][1], sA[O]); A3_SB();                                                                             \
    A3_UA(b, sA[J], 1) if (RL) qA1 = a3_kc(bq1, (CN)); if (KV) to1 = a3_tr(bo_t1, (CT)); A3_SB();                             \
    sB[O] = MFMA(qB1, kf[O][1], sB[O]); A3_SB();                                                                             \
    A3_UB(b) if (RL) qB1 = a3_kc(bq1, (CN) + 2048); if (KV) to2 = a3_tr(bo_t2, (CT)); A3_SB();                                \
    pA[O] = MFMA(oA1, vf[O][1], pA[O]); A3_SB();                                                                             \
    A3_UC(b, pA[J], 1) if (RL) oA1 = a3_kc(bo1, (CN)); if (KV) to3 = a3_tr(bo_t3, (CT)); A3_SB();                             \
    pB[O] = MFMA(oB1, vf[O][1], pB[O]); A3_SB();                                                                             \
    A3_PK(YJ[1], e0b, e1b) A3_PK(WJ[1], d0b, d1b) if (RL) oB1 = a3_kc(bo1, (CN) + 2048);                                      \
    if (KV) tq0 = a3_tr(bq_t0, (CT)); A3_SB();                                                                               \
    if (KV) dv[O][0] = MFMA(to0, pb_, dv[O][0]); A3_SB();                                                                    \
    A3_UA(c, sB[J], 0) if (KV) tq1 = a3_tr(bq_t1, (CT)); A3_SB();                                                            \
    if (KV) dv[O][1] = MFMA(to1, pb_, dv[O][1]); A3_SB();                                                                    \
    A3_UB(c) if (KV) tq2 = a3_tr(bq_t2, (CT)); A3_SB();                                                                      \
    if (KV) dv[O][2] = MFMA(to2, pb_, dv[O][2]); A3_SB();                                                                    \
    A3_UC(c, pB[J], 0) if (KV) tq3 = a3_tr(bq_t3, (CT)); A3_SB();                                                            \
    if (KV) dv[O][3] = MFMA(to3, pb_, dv[O][3]); A3_SB();                                                                    \
    A3_PK(YJ[2], e0c, e1c) A3_PK(WJ[2], d0c, d1c) A3_SB();                                                                   \
    if (KV) dk[O][0] = MFMA(tq0, dsb_, dk[O][0]); A3_SB();                                                                   \
    A3_UA(d, sB[J], 1) sl0 = a3_f4(pl, (CS)); A3_SB();                                                                       \
    if (KV) dk[O][1] = MFMA(tq1, dsb_, dk[O][1]); A3_SB();                                                                   \
    A3_UB(d) sl1 = a3_f4(pl, (CS) + 64); A3_SB();                                                                            \
    if (KV) dk[O][2] = MFMA(tq2, dsb_, dk[O][2]); A3_SB();                                                                   \
    A3_UC(d, pB[J], 1) nd0 = a3_f4(pd, (CS)); A3_SB();                                                                       \
    if (KV) dk[O][3] = MFMA(tq3, dsb_, dk[O][3]); A3_SB();                                                                   \
    A3_PK(YJ[3], e0d, e1d) A3_PK(WJ[3], d0d, d1d) nd1 = a3_f4(pd, (CS) + 64); TAIL A3_SB();                                   \
  }
#define A3_KV_MASK(J)     \
  if (masked) {           \
    sA[J] += mb[J];       \
    sB[J] += mb[J];       \
    A3_SB();              \
  }
      // bases advance by two chunks per trip (8 KiB of panel, 256 B of the -lse / -D rows); the transposed bases, one chunk
      // behind, by ONE chunk after the first trip (whose first step had no previous half chunk) and by two afterwards
#define A3_KV_BUMP(TR)                                                                                                   \
  bq0 += 8192; bq1 += 8192; bo0 += 8192; bo1 += 8192; pl += 256; pd += 256;                                              \
  bq_t0 += (TR); bq_t1 += (TR); bq_t2 += (TR); bq_t3 += (TR); bo_t0 += (TR); bo_t1 += (TR); bo_t2 += (TR); bo_t3 += (TR); \
  asm volatile("" : "+v"(bq0), "+v"(bq1), "+v"(bo0), "+v"(bo1));                                                         \
  asm volatile("" : "+v"(pl), "+v"(pd));                                                                                 \
  asm volatile("" : "+v"(bq_t0), "+v"(bq_t1), "+v"(bq_t2), "+v"(bq_t3));                                                 \
  asm volatile("" : "+v"(bo_t0), "+v"(bo_t1), "+v"(bo_t2), "+v"(bo_t3));
      // first trip (query chunks 0, 1): transposed bases AT chunk 0
      A3_KV_MASK(0)
      A3_KV_STEP(0, 1, y0, w0, y1, w1, false, true, 4096, 0, 128, A3_NONE)       // softmax (0, 0) | scores (0, 1)
      A3_KV_MASK(1)
      A3_KV_STEP(1, 0, y1, w1, y0, w0, true, false, 0, 0, 128, A3_NONE)          // softmax (0, 1) | scores (1, 0), dV dK (0, 0)
      A3_KV_MASK(0)
      A3_KV_STEP(0, 1, y0, w0, y1, w1, true, true, 8192, 0, 256, A3_NONE)        // softmax (1, 0) | scores (1, 1), dV dK (0, 1)
      A3_KV_MASK(1)
      A3_KV_STEP(1, 0, y1, w1, y0, w0, true, false, 0, 4096, 256, A3_KV_BUMP(4096))   // softmax (1, 1) | scores (2, 0), dV dK (1, 0)
      for (int qc = 2; qc < nqc; qc += 2) {   // two query chunks per trip (S % 64 == 0: nqc is even); transposed bases at qc - 1
        A3_KV_MASK(0)
        A3_KV_STEP(0, 1, y0, w0, y1, w1, true, true, 4096, 0, 128, A3_NONE)      // softmax (qc, 0) | scores (qc, 1), dV dK (qc - 1, 1)
        A3_KV_MASK(1)
        A3_KV_STEP(1, 0, y1, w1, y0, w0, true, false, 0, 4096, 128, A3_NONE)     // softmax (qc, 1) | scores (qc + 1, 0), dV dK (qc, 0)
        A3_KV_MASK(0)
        A3_KV_STEP(0, 1, y0, w0, y1, w1, true, true, 8192, 4096, 256, A3_NONE)   // softmax (qc + 1, 0) | scores (qc + 1, 1), dV dK (qc, 1)
        A3_KV_MASK(1)
        A3_KV_STEP(1, 0, y1, w1, y0, w0, true, false, 0, 8192, 256, A3_KV_BUMP(8192))   // softmax (qc + 1, 1) | scores (qc + 2, 0), dV dK (qc + 1, 0)
      }
      {   // dV, dK of the last half chunk (nqc - 1, 1): the transposed bases stand at chunk nqc - 1
        const bf16x8 to0 = a3_tr(bo_t0, 0), to1 = a3_tr(bo_t1, 0), to2 = a3_tr(bo_t2, 0), to3 = a3_tr(bo_t3, 0);
        const bf16x8 tq0 = a3_tr(bq_t0, 0), tq1 = a3_tr(bq_t1, 0), tq2 = a3_tr(bq_t2, 0), tq3 = a3_tr(bq_t3, 0);
        union { uint32_t u[4]; bf16x8 v; } c_;
        c_.u[0] = y1[0]; c_.u[1] = y1[1]; c_.u[2] = y1[2]; c_.u[3] = y1[3];
        dv[1][0] = MFMA(to0, c_.v, dv[1][0]);
        dv[1][1] = MFMA(to1, c_.v, dv[1][1]);
        dv[1][2] = MFMA(to2, c_.v, dv[1][2]);
        dv[1][3] = MFMA(to3, c_.v, dv[1][3]);
        c_.u[0] = w1[0]; c_.u[1] = w1[1]; c_.u[2] = w1[2]; c_.u[3] = w1[3];
        dk[1][0] = MFMA(tq0, c_.v, dk[1][0]);
        dk[1][1] = MFMA(tq1, c_.v, dk[1][1]);
        dk[1][2] = MFMA(tq2, c_.v, dk[1][2]);
        dk[1][3] = MFMA(tq3, c_.v, dk[1][3]);
      }
#undef A3_KV_STEP
#undef A3_KV_MASK
#undef A3_KV_BUMP
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16_t* krow = dqkv + (size_t)(b * S + k0 + j * 16 + li) * ld + H + h * AT_D;
      bf16_t* vrow = krow + H;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        uint2 u;
        const float ks = scale;
        u.x = pack2bf(dk[j][db][0] * ks, dk[j][db][1] * ks);
        u.y = pack2bf(dk[j][db][2] * ks, dk[j][db][3] * ks);
        *reinterpret_cast<uint2*>(krow + db * 16 + g * 4) = u;
        uint2 w;
        w.x = pack2bf(dv[j][db][0], dv[j][db][1]);
        w.y = pack2bf(dv[j][db][2], dv[j][db][3]);
        *reinterpret_cast<uint2*>(vrow + db * 16 + g * 4) = w;
        bsk[db] += dk[j][db] * ks;
        bsv[db] += dv[j][db];
      }
    }
  }
  if (dbias != nullptr) {
    flush_colsum<8>(bsk, red, dbias + H + h * AT_D, wid, lane, tid);
    flush_colsum<8>(bsv, red, dbias + 2 * H + h * AT_D, wid, lane, tid);
  }
}
#undef A3_UA
#undef A3_PIN2
#undef A3_PK
#undef A3_UB
#undef A3_UC
#undef A3_NONE
#undef A3_SB
