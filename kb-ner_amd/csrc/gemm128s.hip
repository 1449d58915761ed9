// bf16 MFMA GEMM, fourth structure (round 6): 128 x 256 x 64 tiles, WAVE-SPECIALISED epilogue.
//
// What round 6 measured on the way here (profiles/round6_gemm128x_lab.txt): an epilogue that shares instruction streams with the
// K loop cannot hide --
//   * beside a wave that issues MFMAs back to back its SIMD sibling gets ~one VALU instruction issued per MFMA, and two waves that
//     both carry MFMAs and erf arithmetic serialise the arithmetic whichever way it is placed;
//   * a global store blocks its wave's issue while the CU's one store path works through the stores of all the waves (~21 cycles
//     per KiB): inside a K loop that is matrix-pipe time.
// So the two kinds of work get their own waves.  Waves 0-3 -- one per SIMD -- are MFMA waves: each owns 128 x 64 of the tile
// (128 accumulators), runs the K loop (LDS-DMA, fragment reads, 64 MFMAs per K step; the interleaved-ring schedule of
// gemm256f_kernel) and NOTHING else: no global store, no epilogue arithmetic.  At the end of a tile they add the bias, round to bf16
// and write the tile into a 64-KiB LDS hand-off buffer (16 ds_write_b128 per wave, ~2 % of a tile), and go on with the next tile.
// Waves 4-7 -- the SIMD siblings -- are epilogue waves: during the 16 K steps of the NEXT tile each reads one sixteenth of its share
// of the handed-off tile per step (one ds_read_b128: 8 pre-activations per lane), applies GELU + GELU' and issues the two full-line
// stores.  Their VALU work issues beside the sibling's MFMAs with the MFMA wave as the older wave (it wins the arbitration), their
// store stalls are their own, and the store stream is spread evenly over the tile instead of a burst at its end.
//   Price: the pre-activation is rounded to bf16 before GELU (as in rounds 1-3; the 256-row kernels evaluate GELU on the fp32
// accumulator since round 4) -- results agree with the 256-row kernel to one bf16 rounding of the pre-activation, not bit for bit;
// a two-stage operand ring (2 x 48 KiB: both operands ONE step ahead) because the hand-off buffer takes 64 of the 160 KiB.
// One barrier per K step + one behind the hand-off, executed by all eight waves.
#include "gemm_tile.h"

#define S_A_BYTES 16384
#define S_STAGE (S_A_BYTES + TILE2_BYTES)     // 48 KiB: A tile (128 rows) + B tile (256 rows)
#define S_HAND_BASE (2 * S_STAGE)            // 96 KiB
#define S_LDS_BYTES (S_HAND_BASE + 65536)    // 160 KiB
#define S_NT 16

typedef unsigned u4v_s __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ void origin128s(int id, int total, int M, int N, int& m0, int& n0) {
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

// 16 bytes per lane from global memory, invisible to hipcc's waitcnt pass (retired by a later hand-written vmcnt(0))
static __device__ __forceinline__ void gload16_async(f4v& dst, unsigned voff, const void* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

#ifdef X128_LAB
__device__ unsigned s_trace[256 * 64];
extern "C" int kbner_debug_read_strace(unsigned* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(s_trace), sizeof(unsigned) * 256 * 64);
}
#define ST(P)                                                                                          \
  if ((ABL & 512) && tno == 5 && wid == 0) {                                                           \
    const unsigned t_ = __builtin_amdgcn_readfirstlane((unsigned)__builtin_readcyclecounter());        \
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(tr0) : "s"(t_), "n"((P) & 63));                   \
  }
#else
#define ST(P)
#endif
// ABL (lab builds, -DX128_LAB): 1 = epilogue waves do no arithmetic / stores (barriers only), 2 = no stores, 4 = no hand-off writes
template <bool B_KS, int EPI, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm128s_kernel(const GroupArgs ga) {
  static_assert(EPI == (EPI_BIAS | EPI_GELU), "specialisations: see kbner_can128x");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  int lane = tid & 63;
  asm volatile("" : "+v"(lane));
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GemmProblem* gp = problem_ptr(0);
  const int total = ga.total_tiles, gstep = (int)gridDim.x;
  const int M = gp->M, N = gp->N;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);
  const int ntiles_mine = (total - (int)blockIdx.x + gstep - 1) / gstep;   // tiles this workgroup walks

  if (wid < 4) {
    // ================================================================== MFMA waves
    const int wn = wid;
    const int lda = gp->lda, ldb = gp->ldb;
    const float alpha = gp->alpha;
    unsigned va[4], vb[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wid * 4 + j) * 8 + (lane >> 3);
      va[j] = (unsigned)(row * lda + (((lane & 7) ^ kc_swz(row)) << 3)) * 2u - (unsigned)(j & 3) * 1024u;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = wid * 8 + j;
      if (!B_KS) {
        const int row = q * 8 + (lane >> 3);
        vb[j] = (unsigned)(row * ldb + (((lane & 7) ^ kcb_swz(row)) << 3)) * 2u - (unsigned)(j & 3) * 1024u;
      } else {
        const int kr = q * 2 + (lane >> 5);
        vb[j] = (unsigned)(kr * ldb + (((lane & 31) ^ (ks_swz(kr) << 1)) << 3)) * 2u - (unsigned)(j & 3) * 1024u;
      }
    }
    const unsigned a_dst0 = lds0 + wid * 4096, b_dst0 = lds0 + S_A_BYTES + wid * 8192;
    const size_t b_kstep = B_KS ? (size_t)BK2 * ldb : (size_t)BK2;
    unsigned laneA, laneB;
    {
      const int row = lane & 15;
      laneA = lds0 + row * 128 + ((((lane >> 4)) ^ kc_swz(row)) << 4);
      if (!B_KS) {
        const int j = lane & 15;
        const int brow = wn * 64 + (j >> 2) * 8 + (j & 3);
        laneB = lds0 + S_A_BYTES + brow * 128 + ((((lane >> 4)) ^ kcb_swz(brow)) << 4);
      } else {
        const int p = lane & 15;
        const int r = (lane >> 4) * 8 + (p >> 2);
        laneB = lds0 + S_A_BYTES + r * 512 + (((wn * 4 + ((p & 3) >> 1)) ^ ks_swz(r)) << 5) + (((p & 3) & 1) << 4);
      }
    }
    typedef const s8v __attribute__((address_space(3))) lds_s8v;
    typedef s4v __attribute__((address_space(3))) lds_s4v;
    typedef u4v_s __attribute__((address_space(3))) lds_u4v;
    auto tr2_ = [&](unsigned addr) -> bf16x8 {
      const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(size_t)addr);
      const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(size_t)(addr + 4u * 512u));
      s8v v;
      v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
      v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
      return __builtin_bit_cast(bf16x8, v);
    };
    unsigned pa0 = laneA, pa1 = laneA ^ 64u, pb0 = laneB, pb1 = laneB ^ 64u;
    auto fa_ = [&](int ks, int mi, unsigned soff) -> bf16x8 {
      const s8v v = *reinterpret_cast<lds_s8v*>((size_t)((ks ? pa1 : pa0) + soff + (unsigned)mi * 2048u));
      return __builtin_bit_cast(bf16x8, v);
    };
    auto fb_ = [&](int ks, int ni, unsigned soff) -> bf16x8 {
      if constexpr (!B_KS) {
        const s8v v = *reinterpret_cast<lds_s8v*>((size_t)((ks ? pb1 : pb0) + soff + (unsigned)((ni >> 1) * 4096 + (ni & 1) * 512)));
        return __builtin_bit_cast(bf16x8, v);
      } else {
        return tr2_(((ni >> 1) ? pb1 : pb0) + soff + (unsigned)ks * 16384u + (unsigned)(ni & 1) * 8u);
      }
    };
    // hand-off write address of this lane: row (mi * 16 + li), 16-byte chunk c = wn * 8 + q * 4 + gq, stored at chunk c ^ li
    // (the 16 rows of a ds_write lane group then fall into 16 different bank groups); q = 1 flips chunk bit 2 = address bit 6
    const int gq = lane >> 4, li = lane & 15;
    unsigned wr0 = lds0 + S_HAND_BASE + li * 512 + (((wn * 8 + gq) ^ li) << 4);
    unsigned wr1 = wr0 ^ 64u;

    int id = blockIdx.x, m0, n0;
    origin128s(id, total, M, N, m0, n0);
    const bf16_t* a_cur = uniform_ptr(gp->A + (size_t)m0 * lda);
    const bf16_t* b_cur = uniform_ptr(B_KS ? gp->B + n0 : gp->B + (size_t)n0 * ldb);

    // prologue: stage 0 of the first tile into slot 0
    glds16_quad(a_cur, va[0], va[1], va[2], va[3], a_dst0);
    glds16_quad(b_cur, vb[0], vb[1], vb[2], vb[3], b_dst0);
    glds16_quad(b_cur, vb[4], vb[5], vb[6], vb[7], b_dst0 + 4096);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pp_barrier();

    unsigned tr0 = 0;
    (void)tr0;
    f4v acc[8][4];
    const f4v zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    bf16x8 b0[4], b1[4], a0[2], a1[2];
#define SSB() __builtin_amdgcn_sched_barrier(0)
#define SNOP (void)0
#define SFA(ks, mi) fa_(ks, mi, soff)
#define SFB(ks, ni) fb_(ks, ni, soff)
#define SFAN(ks, mi) fa_(ks, mi, noff)
#define SFBN(ks, ni) fb_(ks, ni, noff)
#define SMF(a, b, mi, ni, Z) \
  acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[(mi) & 1], (Z) ? zero4 : acc[mi][ni], 0, 0, 0)
#define SGROUP(Z, a, b, pr, f0, f1, f2, f3, f4, f5, f6, f7) \
  SMF(a, b, 2 * (pr), 0, Z); SSB(); f0; SSB();               \
  SMF(a, b, 2 * (pr), 1, Z); SSB(); f1; SSB();               \
  SMF(a, b, 2 * (pr), 2, Z); SSB(); f2; SSB();               \
  SMF(a, b, 2 * (pr), 3, Z); SSB(); f3; SSB();               \
  SMF(a, b, 2 * (pr) + 1, 0, Z); SSB(); f4; SSB();           \
  SMF(a, b, 2 * (pr) + 1, 1, Z); SSB(); f5; SSB();           \
  SMF(a, b, 2 * (pr) + 1, 2, Z); SSB(); f6; SSB();           \
  SMF(a, b, 2 * (pr) + 1, 3, Z); SSB(); f7; SSB();
#define SPA(J) glds16_piece<(J)>(pa_src, va[J], a_dst0 + noff)
#define SPB(J) glds16_piece<((J) & 3)>(pb_src, vb[J], b_dst0 + noff + ((J) >> 2) * 4096u)
    // One K step: consume the stage in slot (S & 1), request stage S + 1 (of the next tile in step 15) into the other slot: its last
    // readers were this step's predecessor's fragment reads, all retired by the barrier in front of this step.  The first fragments
    // of the next stage are read behind the barrier, under the group held back from this step (as in gemm256f_kernel).
#define SSTEP(S)                                                                                                    \
  {                                                                                                                 \
    constexpr unsigned soff = (unsigned)((S) & 1) * S_STAGE, noff = (unsigned)(((S) + 1) & 1) * S_STAGE;            \
    ST(S)                                                                                                           \
    const bf16_t* pa_src = ((S) < 15) ? a_cur + (size_t)((S) + 1) * BK2 : a_nxt;                                    \
    const bf16_t* pb_src = ((S) < 15) ? b_cur + (size_t)((S) + 1) * b_kstep : b_nxt;                                \
    if ((S) == 0) {                                                                                                 \
      b0[0] = SFB(0, 0); b0[1] = SFB(0, 1); b0[2] = SFB(0, 2); b0[3] = SFB(0, 3); a0[0] = SFA(0, 0); a0[1] = SFA(0, 1); \
      SSB();                                                                                                        \
    }                                                                                                               \
    SGROUP((S) == 0, a0, b0, 0, a1[0] = SFA(0, 2), SPB(0), a1[1] = SFA(0, 3), SPB(1), SPB(2), SPB(3), SPB(4), SPB(5))  \
    SGROUP((S) == 0, a1, b0, 1, a0[0] = SFA(0, 4), SPB(6), a0[1] = SFA(0, 5), SPB(7), SPA(0), SPA(1), SPA(2), SPA(3))  \
    SGROUP((S) == 0, a0, b0, 2, a1[0] = SFA(0, 6), SNOP, a1[1] = SFA(0, 7), SNOP, SNOP, SNOP, SNOP, SNOP)           \
    SGROUP((S) == 0, a1, b0, 3, b1[0] = SFB(1, 0), a0[0] = SFA(1, 0), b1[1] = SFB(1, 1), b1[2] = SFB(1, 2), b1[3] = SFB(1, 3), \
           a0[1] = SFA(1, 1), SNOP, SNOP)                                                                           \
    SGROUP(false, a0, b1, 0, a1[0] = SFA(1, 2), SNOP, a1[1] = SFA(1, 3), SNOP, SNOP, SNOP, SNOP, SNOP)              \
    SGROUP(false, a1, b1, 1, a0[0] = SFA(1, 4), SNOP, a0[1] = SFA(1, 5), SNOP, SNOP, SNOP, SNOP, SNOP)              \
    SGROUP(false, a0, b1, 2, a1[0] = SFA(1, 6), SNOP, a1[1] = SFA(1, 7), SNOP, SNOP, SNOP, SNOP, SNOP)              \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
    pp_barrier();                                                                                                   \
    if ((S) < 15) {                                                                                                 \
      b0[0] = SFBN(0, 0);                                                                                           \
      a0[0] = SFAN(0, 0);                                                                                           \
      SSB();                                                                                                        \
      SGROUP(false, a1, b1, 3, b0[1] = SFBN(0, 1), SNOP, b0[2] = SFBN(0, 2), SNOP, b0[3] = SFBN(0, 3), SNOP, a0[1] = SFAN(0, 1), SNOP) \
    } else {                                                                                                        \
      SGROUP(false, a1, b1, 3, SNOP, SNOP, SNOP, SNOP, SNOP, SNOP, SNOP, SNOP)                                      \
    }                                                                                                               \
  }

    for (int tno = 0;; ++tno) {
      const int id_next = id + gstep;
      const bool has_next = id_next < total;
      int m0n = m0, n0n = n0;
      if (has_next) origin128s(id_next, total, M, N, m0n, n0n);
      const bf16_t* a_nxt = uniform_ptr(gp->A + (size_t)m0n * lda);
      const bf16_t* b_nxt = uniform_ptr(B_KS ? gp->B + n0n : gp->B + (size_t)n0n * ldb);
      SSTEP(0) SSTEP(1) SSTEP(2) SSTEP(3) SSTEP(4) SSTEP(5) SSTEP(6) SSTEP(7)
      SSTEP(8) SSTEP(9) SSTEP(10) SSTEP(11) SSTEP(12) SSTEP(13) SSTEP(14) SSTEP(15)
      ST(16)
      // hand-off: bf16(acc * alpha), row-major with the chunk swizzle, 16 bytes (8 consecutive columns) per write; the bias is added
      // by the epilogue waves.  The epilogue waves finished reading the previous tile before the barrier of step 15.
      if (!(ABL & 4)) {
        if (alpha == 1.0f) {   // (wave-uniform; every forward GEMM of the encoder)
#pragma unroll
          for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f4v x = acc[mi][2 * q], y = acc[mi][2 * q + 1];
              const u4v_s o = {pack2bf(x[0], x[1]), pack2bf(x[2], x[3]), pack2bf(y[0], y[1]), pack2bf(y[2], y[3])};
              *reinterpret_cast<lds_u4v*>((size_t)((q ? wr1 : wr0) + (unsigned)mi * 8192u)) = o;
            }
        } else {
#pragma unroll
          for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f4v x = acc[mi][2 * q] * alpha, y = acc[mi][2 * q + 1] * alpha;
              const u4v_s o = {pack2bf(x[0], x[1]), pack2bf(x[2], x[3]), pack2bf(y[0], y[1]), pack2bf(y[2], y[3])};
              *reinterpret_cast<lds_u4v*>((size_t)((q ? wr1 : wr0) + (unsigned)mi * 8192u)) = o;
            }
        }
      }
      ST(17)
      __builtin_amdgcn_s_waitcnt(0xC07F);
      ST(18)
      pp_barrier();   // the tile is in the hand-off buffer
      ST(19)
      if (!has_next) break;
      id = id_next;
      m0 = m0n;
      n0 = n0n;
      a_cur = a_nxt;
      b_cur = b_nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef X128_LAB
    if ((ABL & 512) && wid == 0 && blockIdx.x < 256) s_trace[blockIdx.x * 64 + lane] = tr0;
#endif
#undef SSTEP
#undef SPA
#undef SPB
#undef SGROUP
#undef SMF
#undef SFA
#undef SFB
#undef SFAN
#undef SFBN
#undef SNOP
#undef SSB
  } else {
    // ================================================================== epilogue waves
    const int e = wid - 4;
    const int ldc = gp->ldc, ldd = gp->ldout2;
    typedef const u4v_s __attribute__((address_space(3))) lds_cu4v;
    // chunk k (0..15) of a tile: rows 2 * (4 k + e) + (lane >> 5); LDS position p = lane & 31 of that row holds the 8 columns of
    // chunk (p ^ (row & 15))
    const int half = lane >> 5, p = lane & 31;
    int id = blockIdx.x, m0 = 0, n0 = 0;
    pp_barrier();   // (the MFMA waves' prologue barrier)
    for (int t = 0; t < ntiles_mine; ++t) {
      // the tile being finished during this K loop: the previous one
      const bool has_prev = t > 0;
      bf16_t* cbase = gp->C + (size_t)m0 * ldc + n0;
      bf16_t* dbase = gp->out2 + (size_t)m0 * ldd + n0;
      const float* bias = gp->bias + n0;
#pragma unroll 1
      for (int s = 0; s < S_NT; ++s) {
        if (has_prev && !(ABL & 1)) {
          const int row = 2 * (4 * s + e) + half;
          const u4v_s w = *reinterpret_cast<lds_cu4v*>((size_t)(lds0 + S_HAND_BASE + row * 512 + p * 16));
          const int col = (p ^ (row & 15)) * 8;
          const f4v bl = *reinterpret_cast<const f4v*>(bias + col), bh = *reinterpret_cast<const f4v*>(bias + col + 4);
          const float bb[8] = {bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
          u4v_s yo, dyo;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            f2v y, dy;
            gelu_both2(unpack2bf(w[r]) + (f2v){bb[2 * r], bb[2 * r + 1]}, y, dy);
            yo[r] = pack2bf(y[0], y[1]);
            dyo[r] = pack2bf(dy[0], dy[1]);
          }
          if (!(ABL & 2)) {
            if (ABL & 8) {
              __builtin_nontemporal_store(yo, reinterpret_cast<u4v_s*>(cbase + (size_t)row * ldc + col));
              __builtin_nontemporal_store(dyo, reinterpret_cast<u4v_s*>(dbase + (size_t)row * ldd + col));
            } else {
              *reinterpret_cast<u4v_s*>(cbase + (size_t)row * ldc + col) = yo;
              *reinterpret_cast<u4v_s*>(dbase + (size_t)row * ldd + col) = dyo;
            }
          }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        pp_barrier();
      }
      pp_barrier();   // hand-off of tile t complete
      origin128s(id, total, M, N, m0, n0);   // the tile just handed off becomes the one to finish
      id += gstep;
    }
    // the last tile: no K loop to run beside
    {
      bf16_t* cbase = gp->C + (size_t)m0 * ldc + n0;
      bf16_t* dbase = gp->out2 + (size_t)m0 * ldd + n0;
      const float* bias = gp->bias + n0;
#pragma unroll 1
      for (int s = 0; s < S_NT && !(ABL & 1); ++s) {
        const int row = 2 * (4 * s + e) + half;
        const u4v_s w = *reinterpret_cast<lds_cu4v*>((size_t)(lds0 + S_HAND_BASE + row * 512 + p * 16));
        const int col = (p ^ (row & 15)) * 8;
        const f4v bl = *reinterpret_cast<const f4v*>(bias + col), bh = *reinterpret_cast<const f4v*>(bias + col + 4);
        const float bb[8] = {bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
        u4v_s yo, dyo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f2v y, dy;
          gelu_both2(unpack2bf(w[r]) + (f2v){bb[2 * r], bb[2 * r + 1]}, y, dy);
          yo[r] = pack2bf(y[0], y[1]);
          dyo[r] = pack2bf(dy[0], dy[1]);
        }
        if (!(ABL & 2)) {
          *reinterpret_cast<u4v_s*>(cbase + (size_t)row * ldc + col) = yo;
          *reinterpret_cast<u4v_s*>(dbase + (size_t)row * ldd + col) = dyo;
        }
      }
    }
  }
}

template <bool B_KS, int EPI, int ABL = 0>
static int launch128s_t(const GroupArgs& ga, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  const int r = kbner_set_max_lds_once(attr_done, reinterpret_cast<const void*>(gemm128s_kernel<B_KS, EPI, ABL>), S_LDS_BYTES);
  if (r) return r;
  const int grid = ga.total_tiles < ga.ncu ? ga.total_tiles : ga.ncu;
  hipLaunchKernelGGL((gemm128s_kernel<B_KS, EPI, ABL>), dim3(grid), dim3(512), S_LDS_BYTES, stream, ga);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" int kbner_gemm_get_variant(void);

int kbner_launch128s(int layout, const GroupArgs& ga, hipStream_t stream) {
  const GemmProblem& g = ga.p[0];
  if (ga.nprob != 1 || !kbner_can128x(layout, g.M, g.N, g.K, g.epi)) return 1;
#ifdef X128_LAB
  if (layout == 0 && g.epi == (EPI_BIAS | EPI_GELU)) {
    switch ((kbner_gemm_get_variant() >> 8) & 15) {
      case 1: return launch128s_t<false, (EPI_BIAS | EPI_GELU), 1>(ga, stream);   // epilogue waves idle (barriers only)
      case 2: return launch128s_t<false, (EPI_BIAS | EPI_GELU), 2>(ga, stream);   // no stores
      case 3: return launch128s_t<false, (EPI_BIAS | EPI_GELU), 1 | 4>(ga, stream);   // the K loop alone
      case 4: return launch128s_t<false, (EPI_BIAS | EPI_GELU), 4>(ga, stream);       // no hand-off writes (epilogue on stale LDS)
      case 5: return launch128s_t<false, (EPI_BIAS | EPI_GELU), 4 | 2>(ga, stream);   // ... and no stores
      case 6: return launch128s_t<false, (EPI_BIAS | EPI_GELU), 8>(ga, stream);       // nt stores
      case 7: return launch128s_t<false, (EPI_BIAS | EPI_GELU), 512 | 1>(ga, stream);   // trace, idle epilogue waves
      case 8: return launch128s_t<false, (EPI_BIAS | EPI_GELU), 512>(ga, stream);       // trace, full
      default: break;
    }
  }
#endif
  if (layout == 0 && g.epi == (EPI_BIAS | EPI_GELU)) return launch128s_t<false, (EPI_BIAS | EPI_GELU)>(ga, stream);
  return 1;
}
