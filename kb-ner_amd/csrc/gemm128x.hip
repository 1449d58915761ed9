// bf16 MFMA GEMM, third structure (round 6): 128 x 256 x 64 tiles whose EPILOGUE RUNS UNDER THE NEXT TILE'S K LOOP.
//
// Why.  The encoder's three K = 1024 GEMMs with heavy epilogues -- FFN-up forward (bias + GELU + GELU'; HF BertIntermediate behind
// /root/reference/flair/embeddings.py:3269), FFN-down dgrad (GELU' x dY + column sums), o-projection forward (bias + residual) --
// ran at 0.385 / 0.413 / 0.40 of the dense bf16 MFMA peak on the 256 x 256 ring kernel while every long-K shape ran at 0.50-0.55:
// a 256 x 256 x 1024 tile is 16 K steps (~43.7 K shader cycles) followed by an epilogue during which the matrix pipe idles
// (13.75 K cycles of erf arithmetic for GELU + GELU', two waves per SIMD serialised on the VALU port) and a store burst that
// all 256 CUs issue within the same few microseconds (2 x 128 KiB per CU: the next tile's first K steps wait ~7.8 K cycles
// behind it, vmcnt retiring in order).  21.5 K of a 65 K-cycle tile.  The 256 x 256 tile has no room to do anything about it:
// 128 accumulators + fragments = 252 of 256 VGPRs, 160 of 160 KiB of LDS.
//
// What.  Half-height tiles: a wave owns 64 x 64 (64 accumulator registers), so the PREVIOUS tile's 64 accumulators stay alive
// next to the current ones, and the 16 K steps of a tile each carry one sixteenth of the previous tile's epilogue -- one
// accumulator fragment (16 rows x 4 columns per lane: bias, GELU + GELU', pack, 8-byte LDS transpose writes), and after every
// fourth fragment the row block's full-line stores.  No epilogue phase exists any more: the stores are spread evenly over the
// K loop (4 per wave every 4 steps instead of 32 in a burst), the operand tiles stream without a tile-boundary bubble.
//   Anti-phase VALU.  MFMA and VALU are separate pipes of a SIMD but share its issue port, and the two waves a 512-thread
// workgroup puts on each SIMD run the same code from the same barrier: a VALU block that both execute at the same time
// serialises (that IS the 13.75 K cycles above).  Here waves 0-3 run their epilogue slice at the START of a K step and waves 4-7
// at the END (wave w and w + 4 share a SIMD: MI355X_MICROARCH.md, LDS section), so on every SIMD one wave's ~70 VALU
// instructions issue into the gaps of the other's 32 MFMAs (tools/micro/valu_probe.hip, round 3: a VALU block beside an
// MFMA-only wave takes twice its own time and costs the MFMA wave nothing).
//   Price: 48 KiB of operands per 128 x 256 x 64 step instead of 64 KiB per 256 x 256 x 64 (1.5 x the L2 -> LDS bytes per flop),
// 16 instead of 12 fragment reads per 32 MFMAs -- the loop is bound by the LDS-DMA stream (lab, profiles/round6_gemm128x_lab.txt:
// the DMA skeleton alone 733 us of an 829-us main loop at 256 sentences), so BOTH operands are requested TWO steps ahead: a ring
// of three 48-KiB stages (A 16 KiB + B 32 KiB) = 144 KiB, + 2 KiB of wave-private scratch (one transpose buffer that the two
// outputs of a row block pass through one after the other) = all 160 KiB.  The bias line of a tile lives in ONE VGPR (lane i =
// column i of the wave's 64) and reaches the lanes through ds_bpermute_b32 (no LDS memory).
//   Everything in the K loop that touches memory is issued from inline asm (LDS-DMA, the epilogue's global stores, the bias
// line's DMA) and waited for with hand-counted s_waitcnt vmcnt(N): a compiler-visible VMEM result in flight makes hipcc drain
// vmcnt(0) around the counted waits.  A tile is exactly 16 K steps (K = 1024), written out step by step
// so that every accumulator index is static.
//
// Results are bit-identical to the 256-row kernels' (same MFMA order per output element, same epilogue arithmetic):
// tests/test_gpu_kernels.py test_gemm128x_*.
#include "gemm_tile.h"

#define X_A_BYTES 16384
#define X_B_BASE (3 * X_A_BYTES)
#define X_SCR_BASE (X_B_BASE + 3 * TILE2_BYTES)
#define X_SCR_WAVE 2048
#define X_LDS_BYTES (X_SCR_BASE + 8 * X_SCR_WAVE)   // 163840: everything
#define X_NT 16           // K steps per tile: K = 1024

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

// FL: cache policy of the store (lab): 0 plain, 1 nt, 2 sc1, 3 sc0 sc1
template <int FL = 0>
static __device__ __forceinline__ void gstore16(unsigned voff, u4v d, const void* sbase) {
  if constexpr (FL == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt" : : "v"(voff), "v"(d), "s"(sbase) : "memory");
  else if constexpr (FL == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" : : "v"(voff), "v"(d), "s"(sbase) : "memory");
  else if constexpr (FL == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" : : "v"(voff), "v"(d), "s"(sbase) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(voff), "v"(d), "s"(sbase) : "memory");
}
// one dword per lane, invisible to hipcc's waitcnt pass (the result is not touched before a later counted vmcnt has retired it)
// (the destination is the caller's own long-lived variable, "+v": with a fresh "=v" temporary hipcc copies the temporary out at once
// -- before the load has landed -- and hands its register to the next computation, which the landing load then overwrites)
static __device__ __forceinline__ void gload4_async(float& dst, unsigned voff, const void* sbase) {
  asm volatile("global_load_dword %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// linear id -> origin of a 128 x 256 tile of a single problem: pick_tile<128>'s XCD-aware walk (an XCD's 32 concurrent tiles form
// an 8 x 4 patch: 8 A panels of 128 rows + 4 B panels = 4 MiB at K = 1024, the size of its L2)
static __device__ __forceinline__ void origin128(int id, int total, int M, int N, int& m0, int& n0) {
  const int xcd = id & 7;
  const int q8 = total >> 3, r8 = total & 7;
  const int tile = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int tiles_n = N / T2, tiles_m = M / 128;
  const int group = 8 * tiles_n;
  const int first_m = (tile / group) * 8;
  const int gm = min(tiles_m - first_m, 8);
  const int r = tile % group;
  m0 = (first_m + r % gm) * 128;
  n0 = (r / gm) * T2;
}

#ifdef X128_LAB
// lab trace (ABL bit 512): low 32 bits of s_memtime at 5 points of each of the 16 K steps of every workgroup's 6th tile, per wave
__device__ unsigned x_trace[256 * 8 * 128];
extern "C" int kbner_debug_read_xtrace(unsigned* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(x_trace), sizeof(unsigned) * 256 * 8 * 128);
}
#define XT(S, P)                                                                                             \
  if ((ABL & 512) && tile_no == 5) {                                                                         \
    const unsigned t_ = __builtin_amdgcn_readfirstlane((unsigned)__builtin_readcyclecounter());              \
    if ((S) * 5 + (P) < 64) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(tr0) : "s"(t_), "n"(((S) * 5 + (P)) & 63)); \
    else asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(tr1) : "s"(t_), "n"(((S) * 5 + (P)) & 63));        \
  }
#else
#define XT(S, P)
#endif
// ABL (lab builds, -DX128_LAB: kbner_gemm_set_variant bits 8-11 pick one): 1 = no epilogue slices, 2 = no epilogue stores, 4 = no MFMA,
// 8 = both roles run their slice at the START of a step (lockstep VALU), 16 = no LDS-DMA in the loop; bits 7-8: store cache policy
template <bool B_KS, int EPI, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm128x_kernel(const GroupArgs ga) {
  static_assert(EPI == (EPI_BIAS | EPI_GELU), "specialisations: see kbner_can128x");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  int lane = tid & 63;
  asm volatile("" : "+v"(lane));
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int role = wm;   // 0: epilogue slice at the start of a K step, 1: at its end
  const GemmProblem* gp = problem_ptr(0);
  const int total = ga.total_tiles, gstep = (int)gridDim.x;
  const int M = gp->M, N = gp->N, lda = gp->lda, ldb = gp->ldb, ldc = gp->ldc, ldd = gp->ldout2;
  const float alpha = gp->alpha;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);

  // per-lane source offsets of this wave's LDS-DMA pieces: 2 of the A tile (128 rows x 128 B, row-major image), 4 of the B tile
  unsigned va[2], vb[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wid * 2 + j) * 8 + (lane >> 3);
    va[j] = (unsigned)(row * lda + (((lane & 7) ^ kc_swz(row)) << 3)) * 2u - (unsigned)j * 1024u;
  }
  stage_voff<B_KS, true>(ldb, wid, lane, vb);
  const unsigned a_dst0 = lds0 + wid * 2048, b_dst0 = lds0 + X_B_BASE + wid * 4096;
  const size_t b_kstep = B_KS ? (size_t)BK2 * ldb : (size_t)BK2;

  // fragment read addresses (gemm256f_kernel's maps) of ring slot 0
  unsigned laneA, laneB;
  {
    const int row = wm * 64 + (lane & 15);
    laneA = lds0 + row * 128 + ((((lane >> 4)) ^ kc_swz(row)) << 4);
    if (!B_KS) {
      const int j = lane & 15;
      const int brow = wn * 64 + (j >> 2) * 8 + (j & 3);
      laneB = lds0 + X_B_BASE + brow * 128 + ((((lane >> 4)) ^ kcb_swz(brow)) << 4);
    } else {
      const int p = lane & 15;
      const int r = (lane >> 4) * 8 + (p >> 2);
      laneB = lds0 + X_B_BASE + r * 512 + (((wn * 4 + ((p & 3) >> 1)) ^ ks_swz(r)) << 5) + (((p & 3) & 1) << 4);
    }
  }
  // epilogue lane constants
  const int gq = lane >> 4, r16 = lane & 15;
  const unsigned scr = lds0 + X_SCR_BASE + wid * X_SCR_WAVE;
  unsigned wr0 = scr + r16 * 128 + ((gq ^ (r16 & 7)) << 4);   // column half q = 0 (q = 1: ^ 64); + 8 h
  unsigned wr1 = wr0 ^ 64u;
  unsigned rd0;
  {
    const int rd_row = lane >> 3, rd_chunk = lane & 7;
    rd0 = scr + rd_row * 128 + ((rd_chunk ^ (rd_row & 7)) << 4);
  }
  int bidx = gq * 32;   // ds_bpermute byte index of this lane's first bias value: + 128 q + 16 h + 4 r
  unsigned voffC, voffC8, voffD, voffD8;
  {
    const int rd_row = lane >> 3, rd_chunk = lane & 7;
    voffC = (unsigned)(rd_row * ldc + rd_chunk * 8) * 2u;
    voffC8 = voffC + (unsigned)ldc * 16u;
    voffD = (unsigned)(rd_row * ldd + rd_chunk * 8) * 2u;
    voffD8 = voffD + (unsigned)ldd * 16u;
  }
  const unsigned vlane4 = (unsigned)lane * 4u;

  typedef const s8v __attribute__((address_space(3))) lds_s8v;
  typedef s4v __attribute__((address_space(3))) lds_s4v;
  typedef const u4v __attribute__((address_space(3))) lds_u4v;
  typedef u2v __attribute__((address_space(3))) lds_u2v;
  auto tr2_ = [&](unsigned addr) -> bf16x8 {
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(size_t)addr);
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(size_t)(addr + 4u * 512u));
    s8v v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };
  unsigned pa0 = laneA, pa1 = laneA ^ 64u, pb0 = laneB, pb1 = laneB ^ 64u;
  auto fa_ = [&](int ks, int mi) -> bf16x8 {
    const s8v v = *reinterpret_cast<lds_s8v*>((size_t)((ks ? pa1 : pa0) + (unsigned)mi * 2048u));
    return __builtin_bit_cast(bf16x8, v);
  };
  auto fb_ = [&](int ks, int ni) -> bf16x8 {
    if constexpr (!B_KS) {
      const s8v v = *reinterpret_cast<lds_s8v*>((size_t)((ks ? pb1 : pb0) + (unsigned)((ni >> 1) * 4096 + (ni & 1) * 512)));
      return __builtin_bit_cast(bf16x8, v);
    } else {
      return tr2_(((ni >> 1) ? pb1 : pb0) + (unsigned)ks * 16384u + (unsigned)(ni & 1) * 8u);
    }
  };

  // ---- tile bookkeeping (all wave-uniform)
  int id = blockIdx.x;
  int m0, n0;
  origin128(id, total, M, N, m0, n0);
  const bf16_t* a_cur = uniform_ptr(gp->A + (size_t)m0 * lda);
  const bf16_t* b_cur = uniform_ptr(B_KS ? gp->B + n0 : gp->B + (size_t)n0 * ldb);
  // the tile whose epilogue is running (the first pass finishes the first tile itself with zero accumulators: stores that the real
  // epilogue, issued later by the same wave to the same addresses, overwrites -- this keeps every K step free of branches)
  const bf16_t* c_prev = uniform_ptr(gp->C + (size_t)(m0 + wm * 64) * ldc + n0 + wn * 64);
  const bf16_t* d_prev = uniform_ptr(gp->out2 + (size_t)(m0 + wm * 64) * ldd + n0 + wn * 64);
  const float* bias_ptr = (const float*)uniform_ptr((const bf16_t*)(gp->bias + n0 + wn * 64));
  float bias_cur = 0.0f, bias_nxt = 0.0f;   // lane i: bias of column i of the wave's 64 -- of the tile being finished / computed

  f4v acc[4][4], prev[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) prev[mi][ni] = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
  const f4v zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

  // prologue: stages 0 and 1 of the first tile into ring slots 0 and 1
  unsigned sa_off = 0, sb_off = 0;                                               // ring slot of the stage being consumed
  unsigned a_dst = a_dst0 + 2 * X_A_BYTES, b_dst = b_dst0 + 2 * TILE2_BYTES;     // ring slot of the stage being requested (two ahead)
  glds16_pair<0>(a_cur, va[0], va[1], a_dst0);
  glds16_quad(b_cur, vb[0], vb[1], vb[2], vb[3], b_dst0);
  glds16_pair<0>(a_cur + BK2, va[0], va[1], a_dst0 + X_A_BYTES);
  glds16_quad(b_cur + b_kstep, vb[0], vb[1], vb[2], vb[3], b_dst0 + TILE2_BYTES);
  if (!(ABL & 2)) {
    // four stores of zeros into the first tile's last row block (overwritten by its real epilogue): the first step's wait then
    // counts like every other tile's step 0, which follows the four stores of a step 15
    const u4v z_ = {0u, 0u, 0u, 0u};
    const bf16_t* cp_ = uniform_ptr(c_prev + (size_t)48 * ldc);
    const bf16_t* dp_ = uniform_ptr(d_prev + (size_t)48 * ldd);
    gstore16<(ABL >> 7) & 3>(voffC, z_, cp_);
    gstore16<(ABL >> 7) & 3>(voffC8, z_, cp_);
    gstore16<(ABL >> 7) & 3>(voffD, z_, dp_);
    gstore16<(ABL >> 7) & 3>(voffD8, z_, dp_);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  pp_barrier();

  int tile_no = 0;
  unsigned tr0 = 0, tr1 = 0;
  (void)tile_no; (void)tr0; (void)tr1;
  bf16x8 b0[4], b1[4], a0[2], a1[2];
  u2v dkeep[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};   // the second output's four column quarters of the row block in progress
#define XSB() __builtin_amdgcn_sched_barrier(0)
#define XNOP (void)0
#define XFA(ks, mi) fa_(ks, mi)
#define XFB(ks, ni) fb_(ks, ni)
#define XMF(a, b, mi, ni, Z) \
  if (!(ABL & 4)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[(mi) & 1], (Z) ? zero4 : acc[mi][ni], 0, 0, 0)
#define XGROUP(Z, a, b, pr, f0, f1, f2, f3, f4, f5, f6, f7) \
  XMF(a, b, 2 * (pr), 0, Z); XSB(); f0; XSB();               \
  XMF(a, b, 2 * (pr), 1, Z); XSB(); f1; XSB();               \
  XMF(a, b, 2 * (pr), 2, Z); XSB(); f2; XSB();               \
  XMF(a, b, 2 * (pr), 3, Z); XSB(); f3; XSB();               \
  XMF(a, b, 2 * (pr) + 1, 0, Z); XSB(); f4; XSB();           \
  XMF(a, b, 2 * (pr) + 1, 1, Z); XSB(); f5; XSB();           \
  XMF(a, b, 2 * (pr) + 1, 2, Z); XSB(); f6; XSB();           \
  XMF(a, b, 2 * (pr) + 1, 3, Z); XSB(); f7; XSB();
#define XPB(J) if (!(ABL & 16)) glds16_piece<J>(pb_src, vb[J], b_dst)
#define XPA(J) if (!(ABL & 16)) glds16_piece<J>(pa_src, va[J], a_dst)
  // One sixteenth of the previous tile's epilogue: accumulator fragment (mi, ni) = (s >> 2, s & 3), i.e. 16 rows x this lane's
  // columns q * 32 + gq * 8 + h * 4 .. + 3 of the wave's 64 (q = ni >> 1, h = ni & 1).  The activation's quarter goes into the
  // transpose buffer, the derivative's waits in two registers until the row block is complete (XTAIL).
#define XSLICE(S)                                                                                                  \
  {                                                                                                                \
    constexpr int mi_ = (S) >> 2, ni_ = (S) & 3, q_ = ni_ >> 1, h_ = ni_ & 1;                                       \
    const f4v p_ = prev[mi_][ni_];                                                                                  \
    float v_[4];                                                                                                   \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                \
      const float b_ = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bidx + (q_ * 128 + h_ * 16 + r * 4), \
                                                                               __builtin_bit_cast(int, bias_cur))); \
      v_[r] = p_[r] * alpha;                                                                                       \
      v_[r] += b_;                                                                                                 \
    }                                                                                                              \
    f2v y0_, d0_, y1_, d1_;                                                                                        \
    gelu_both2((f2v){v_[0], v_[1]}, y0_, d0_);                                                                     \
    gelu_both2((f2v){v_[2], v_[3]}, y1_, d1_);                                                                     \
    const u2v co_ = {pack2bf(y0_[0], y0_[1]), pack2bf(y1_[0], y1_[1])};                                            \
    dkeep[ni_] = (u2v){pack2bf(d0_[0], d0_[1]), pack2bf(d1_[0], d1_[1])};                                          \
    *reinterpret_cast<lds_u2v*>((size_t)((q_ ? wr1 : wr0) + (unsigned)(h_ * 8))) = co_;                             \
  }
  // After a row block's fourth fragment its 16 rows x 64 columns leave as full 128-byte lines: the activation rows are read back
  // from the transpose buffer, the derivative's quarters take their place (LDS operations of one wave execute in order) and are
  // read back in turn; four stores.  Step 12: the bias line of the tile being computed (see the wait counts below).
#define XTAIL(S)                                                                                                   \
  {                                                                                                                \
    if (((S) & 3) == 3 && !(ABL & 2)) {                                                                            \
      const u4v h0_ = *reinterpret_cast<lds_u4v*>((size_t)rd0);                                                     \
      const u4v h1_ = *reinterpret_cast<lds_u4v*>((size_t)(rd0 + 1024u));                                           \
      *reinterpret_cast<lds_u2v*>((size_t)(wr0)) = dkeep[0];                                                        \
      *reinterpret_cast<lds_u2v*>((size_t)(wr0 + 8u)) = dkeep[1];                                                   \
      *reinterpret_cast<lds_u2v*>((size_t)(wr1)) = dkeep[2];                                                        \
      *reinterpret_cast<lds_u2v*>((size_t)(wr1 + 8u)) = dkeep[3];                                                   \
      const u4v e0_ = *reinterpret_cast<lds_u4v*>((size_t)rd0);                                                     \
      const u4v e1_ = *reinterpret_cast<lds_u4v*>((size_t)(rd0 + 1024u));                                           \
      const bf16_t* cp_ = uniform_ptr(c_prev + (size_t)(((S) >> 2) * 16) * ldc);                                    \
      const bf16_t* dp_ = uniform_ptr(d_prev + (size_t)(((S) >> 2) * 16) * ldd);                                    \
      gstore16<(ABL >> 7) & 3>(voffC, h0_, cp_);                                                                   \
      gstore16<(ABL >> 7) & 3>(voffC8, h1_, cp_);                                                                  \
      gstore16<(ABL >> 7) & 3>(voffD, e0_, dp_);                                                                   \
      gstore16<(ABL >> 7) & 3>(voffD8, e1_, dp_);                                                                  \
    }                                                                                                              \
    if ((S) == 12) gload4_async(bias_nxt, vlane4, bias_ptr);                                                       \
  }
  // VMEM issue order of a wave in step s: the 6 pieces of stage s + 2, then the step's extras e(s) (4 stores in steps 3, 7, 11, 15;
  // the bias load in step 12).  The barrier at the end of step s needs stage s + 1, i.e. every piece of step s - 1: what may
  // stay in flight is everything issued after those -- e(s - 1) + 6 + e(s).  (vmcnt retires in order, stores included: a store
  // is therefore waited for two steps after it was issued, not one.)
#define XEXTRA(S) ((((S) & 3) == 3 && !(ABL & 2)) ? 4 : (((S) == 12) ? 1 : 0))
#define XWAIT(S)                                                                                                   \
  {                                                                                                                \
    constexpr int n_ = ((ABL & 16) ? 0 : 6) + XEXTRA(((S) + 15) & 15) + XEXTRA(S);                                 \
    static_assert(n_ == 0 || n_ == 1 || n_ == 4 || n_ == 5 || n_ == 6 || n_ == 7 || n_ == 10 || n_ == 11, "wait table");  \
    if (n_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
    else if (n_ == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                             \
    else if (n_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                             \
    else if (n_ == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                             \
    else if (n_ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                             \
    else if (n_ == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                                             \
    else if (n_ == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                           \
    else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");                                                         \
  }
  // One K step: consume the stage in ring slot (sa_off, sb_off), request stage s + 2 (of the next tile in steps 14, 15).
#define XSTEP(S)                                                                                                   \
  {                                                                                                                \
    const bf16_t* pb_src = ((S) < 14) ? b_cur + (size_t)((S) + 2) * b_kstep : b_nxt + (size_t)((S) >= 14 ? (S) - 14 : 0) * b_kstep; \
    const bf16_t* pa_src = ((S) < 14) ? a_cur + (size_t)((S) + 2) * BK2 : a_nxt + (size_t)((S) >= 14 ? (S) - 14 : 0) * BK2; \
    XT(S, 0)                                                                                                       \
    b0[0] = XFB(0, 0);                                                                                             \
    a0[0] = XFA(0, 0);                                                                                             \
    if ((S) == 0) {                                                                                                \
      b0[1] = XFB(0, 1); b0[2] = XFB(0, 2); b0[3] = XFB(0, 3); a0[1] = XFA(0, 1);                                  \
    }                                                                                                              \
    XSB();                                                                                                         \
    if ((role == 0 || (ABL & 8)) && !(ABL & 1)) XSLICE(S)                                                          \
    XSB();                                                                                                         \
    XT(S, 1)                                                                                                       \
    if (ABL & 64) __builtin_amdgcn_s_setprio(1);                                                                   \
    if ((S) == 0) {                                                                                                \
      XPB(0); XPB(1); XPB(2); XPB(3);                                                                              \
      XSB();                                                                                                       \
    } else {                                                                                                       \
      XGROUP(false, a1, b1, 1, b0[1] = XFB(0, 1), XPB(0), b0[2] = XFB(0, 2), XPB(1), b0[3] = XFB(0, 3), XPB(2), a0[1] = XFA(0, 1), XPB(3)) \
    }                                                                                                              \
    XGROUP((S) == 0, a0, b0, 0, a1[0] = XFA(0, 2), XPA(0), a1[1] = XFA(0, 3), XPA(1), XNOP, XNOP, XNOP, XNOP)      \
    XGROUP((S) == 0, a1, b0, 1, b1[0] = XFB(1, 0), a0[0] = XFA(1, 0), b1[1] = XFB(1, 1), b1[2] = XFB(1, 2), b1[3] = XFB(1, 3), \
           a0[1] = XFA(1, 1), XNOP, XNOP)                                                                          \
    XGROUP(false, a0, b1, 0, a1[0] = XFA(1, 2), XNOP, a1[1] = XFA(1, 3), XNOP, XNOP, XNOP, XNOP, XNOP)             \
    if ((S) == 15) {                                                                                               \
      XGROUP(false, a1, b1, 1, XNOP, XNOP, XNOP, XNOP, XNOP, XNOP, XNOP, XNOP)                                     \
    }                                                                                                              \
    XT(S, 2)                                                                                                       \
    if (ABL & 64) __builtin_amdgcn_s_setprio(0);                                                                   \
    if (role == 1 && !(ABL & (1 | 8))) XSLICE(S)                                                                   \
    XSB();                                                                                                         \
    XTAIL(S)                                                                                                       \
    XSB();                                                                                                         \
    if ((S) == 15) {                                                                                               \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                             \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) prev[mi][ni] = acc[mi][ni];                               \
    }                                                                                                              \
    a_dst = (a_dst == a_dst0 + 2 * X_A_BYTES) ? a_dst0 : a_dst + X_A_BYTES;                                        \
    b_dst = (b_dst == b_dst0 + 2 * TILE2_BYTES) ? b_dst0 : b_dst + TILE2_BYTES;                                    \
    sa_off = (sa_off == 2 * X_A_BYTES) ? 0u : sa_off + X_A_BYTES;                                                  \
    sb_off = (sb_off == 2 * TILE2_BYTES) ? 0u : sb_off + TILE2_BYTES;                                              \
    XT(S, 3)                                                                                                       \
    XWAIT(S)                                                                                                       \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
    XT(S, 4)                                                                                                       \
    pp_barrier();                                                                                                  \
    pa0 = laneA + sa_off;                                                                                          \
    asm volatile("" : "+v"(pa0));                                                                                  \
    pa1 = pa0 ^ 64u;                                                                                               \
    asm volatile("" : "+v"(pa1));                                                                                  \
    pb0 = laneB + sb_off;                                                                                          \
    asm volatile("" : "+v"(pb0));                                                                                  \
    pb1 = pb0 ^ 64u;                                                                                               \
    asm volatile("" : "+v"(pb1));                                                                                  \
  }

  for (;;) {
    const int id_next = id + gstep;
    const bool has_next = id_next < total;
    int m0n = m0, n0n = n0;
    if (has_next) origin128(id_next, total, M, N, m0n, n0n);
    // (no next tile: the last two steps re-read this tile's first stages into ring slots nobody reads any more)
    const bf16_t* a_nxt = uniform_ptr(gp->A + (size_t)m0n * lda);
    const bf16_t* b_nxt = uniform_ptr(B_KS ? gp->B + n0n : gp->B + (size_t)n0n * ldb);
    XSTEP(0) XSTEP(1) XSTEP(2) XSTEP(3) XSTEP(4) XSTEP(5) XSTEP(6) XSTEP(7)
    XSTEP(8) XSTEP(9) XSTEP(10) XSTEP(11) XSTEP(12) XSTEP(13) XSTEP(14) XSTEP(15)
    // the tile just computed becomes the one being finished (its bias line, requested in step 12, was retired by step 14's wait)
    c_prev = uniform_ptr(gp->C + (size_t)(m0 + wm * 64) * ldc + n0 + wn * 64);
    d_prev = uniform_ptr(gp->out2 + (size_t)(m0 + wm * 64) * ldd + n0 + wn * 64);
    bias_cur = bias_nxt;
    asm volatile("" : "+v"(bias_cur));
    ++tile_no;
    if (!has_next) break;
    id = id_next;
    m0 = m0n;
    n0 = n0n;
    a_cur = a_nxt;
    b_cur = b_nxt;
    bias_ptr = (const float*)uniform_ptr((const bf16_t*)(gp->bias + n0 + wn * 64));
  }
  // the last tile's epilogue has no K loop to hide under
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define XDRAIN(S) XSLICE(S) XSB(); XTAIL(S) XSB();
  XDRAIN(0) XDRAIN(1) XDRAIN(2) XDRAIN(3) XDRAIN(4) XDRAIN(5) XDRAIN(6) XDRAIN(7)
  XDRAIN(8) XDRAIN(9) XDRAIN(10) XDRAIN(11) XDRAIN(12) XDRAIN(13) XDRAIN(14) XDRAIN(15)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef X128_LAB
  if ((ABL & 512) && blockIdx.x < 256) {
    x_trace[(blockIdx.x * 8 + wid) * 128 + lane] = tr0;
    x_trace[(blockIdx.x * 8 + wid) * 128 + 64 + lane] = tr1;
  }
#endif
#undef XDRAIN
#undef XSTEP
#undef XWAIT
#undef XEXTRA
#undef XTAIL
#undef XSLICE
#undef XPA
#undef XPB
#undef XGROUP
#undef XMF
#undef XFA
#undef XFB
#undef XNOP
#undef XSB
}

template <bool B_KS, int EPI, int ABL = 0>
static int launch128x_t(const GroupArgs& ga, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  const int r = kbner_set_max_lds_once(attr_done, reinterpret_cast<const void*>(gemm128x_kernel<B_KS, EPI, ABL>), X_LDS_BYTES);
  if (r) return r;
  const int grid = ga.total_tiles < ga.ncu ? ga.total_tiles : ga.ncu;
  hipLaunchKernelGGL((gemm128x_kernel<B_KS, EPI, ABL>), dim3(grid), dim3(512), X_LDS_BYTES, stream, ga);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int kbner_gemm_get_variant(void);
bool kbner_can128x(int layout, int M, int N, int K, int epi) {
  if (K != X_NT * BK2 || M % 128 != 0 || N % T2 != 0) return false;
  if (layout == 0 && epi == (EPI_BIAS | EPI_GELU)) return true;
  return false;
}

int kbner_launch128x(int layout, const GroupArgs& ga, hipStream_t stream) {
  const GemmProblem& g = ga.p[0];
  if (ga.nprob != 1 || !kbner_can128x(layout, g.M, g.N, g.K, g.epi)) return 1;
#ifdef X128_LAB
  if (layout == 0 && g.epi == (EPI_BIAS | EPI_GELU)) {
    switch ((kbner_gemm_get_variant() >> 8) & 15) {
      case 1: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 1>(ga, stream);          // no epilogue slices
      case 2: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 2>(ga, stream);          // no epilogue stores
      case 3: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 3>(ga, stream);          // neither: the main loop
      case 4: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 4>(ga, stream);          // no MFMA
      case 5: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 7>(ga, stream);          // LDS-DMA + fragment reads only
      case 6: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 8>(ga, stream);          // lockstep slices
      case 7: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 16 | 3>(ga, stream);     // MFMA + fragment reads only
      case 8: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 8 | 2>(ga, stream);      // lockstep slices, no stores
      case 9: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 64>(ga, stream);         // s_setprio 1 around the MFMA phase
      case 10: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 128>(ga, stream);       // nt stores
      case 11: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 512>(ga, stream);       // cycle trace
      case 12: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 256>(ga, stream);       // sc1 stores
      case 13: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 512 | 2>(ga, stream);   // cycle trace, no stores
      case 14: return launch128x_t<false, (EPI_BIAS | EPI_GELU), 512 | 3>(ga, stream);   // cycle trace, main loop only
      default: break;
    }
  }
#endif
  if (layout == 0 && g.epi == (EPI_BIAS | EPI_GELU)) return launch128x_t<false, (EPI_BIAS | EPI_GELU)>(ga, stream);
  return 1;
}
