// Attention, third structure (round 3): streaming kernels with 32 stationary rows per wave.
//
// What round 2's counters missed and tools/micro/valu_probe.hip measured on MI355X:
//   * a wave64 fp32 VALU instruction issues over 4 cycles (v_exp_f32 8, v_pk_fma_f32 8 -- no gain from packing), and
//   * MFMA and VALU work on one SIMD SERIALISE, whichever wave they come from: 36 mfma_16x16x32 (592 cycles) next to the
//     84-instruction softmax phase of a 64-key block (564 cycles alone) take 1050-1150 cycles in every arrangement (one wave or
//     two per SIMD, phases or interleaved); only ~1 VALU instruction hides behind each 16-cycle MFMA.
// So the round-2 kernels (226 VALU-class instructions per 64-key block and wave, rocprofv3 --pmc) were bound by the SUM of both
// pipes, and the lever is the instruction count of the vector pipe, not occupancy or overlap:
//   * a wave owns 32 query rows (two 16-row blocks): every K / V^T fragment read from LDS feeds two MFMAs;
//   * per score only scale-and-shift (half a v_pk_fma_f32), v_exp_f32 and half a v_cvt_pk_bf16_f32 remain.  No row maximum in
//     the steady state: the reference m of a row is fixed after the first 64-key block at (block maximum + 2^A3_MARGIN of
//     headroom) and later blocks are only CHECKED -- bit 14 of a bf16 (the top exponent bit) is set iff the value is >= 2, so one
//     OR over the packed probabilities plus one AND answers "is some P >= 2" for a whole tile; if it fires (a later key beats
//     the first block's maximum by more than 2^A3_MARGIN: rare) the reference is raised, what has been accumulated is rescaled
//     and the tile redone.  Row sums come from one MFMA per 32 keys against an all-ones A fragment;
//   * the MFMAs of a step sit between the vector instructions of the same wave (P.V of the previous 32 keys next to the exp2
//     of the next 32, the next block's scores next to the second half), which is where the one-VALU-per-MFMA overlap is.
// And at B = 128 the forward moves 537 MB per call (Q, K, V in, O out): ~100 us at the HBM rate the chip sustains, as much as
// its arithmetic.  The panels of the streamed side therefore ROLL: a workgroup item is one (head, 256-query tile); as soon as
// every wave is done with a 64-row block of the current item's K / V, the same LDS stage is refilled (LDS-DMA) with the next
// item's block, one barrier per step, so the DMA runs under the whole computation instead of in a window at the end of a head.
// NOTHING the compiler can see touches global memory inside the item loop: the mask row arrives by DMA (its metadata is derived
// from the LDS copy), the next item's stationary fragments are loaded and the outputs stored by inline asm, and every wait is a
// counted s_waitcnt written here -- a compiler-visible load would make hipcc drain vmcnt(0), i.e. the whole DMA queue, at its
// first use, and __syncthreads() behind a visible store does the same (raw s_barrier + lgkmcnt(0) instead).
// Arithmetic: P = softmax(Q K^T / sqrt(d) + maskbias[key]), O = P V (transformers 3.0.0 BertSelfAttention, reached from
// flair/embeddings.py:3269), products exact in fp32 as before (the scale is applied to the fp32 sums).
#include "attn_common.h"
#include <cstdio>
#include <cstdlib>

#define A3_ROWS 256      // stationary rows per item: 8 waves x 32
#define A3_MARGIN 8.0f   // headroom of the row reference above the first block's maximum, log2 domain
// LDS: two 64-KiB rings (8 stages of 64 rows x 128 B), 2 x 8 KiB mask staging (one 1-KiB piece per wave), 2 x 2 KiB column keys
#define A3_RING 65536
#define A3_LDS_BYTES (2 * A3_RING + 2 * 8192 + 2 * AT_MAXS * 4)

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

// workgroup barrier that orders LDS traffic only (see the header: no vmcnt drain)
static __device__ __forceinline__ void a3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- global memory, all by inline asm (counted by hand) -------------------------------------------------------------------
// piece `wid` (8 rows) of 64-row block `blk` of a [S,64] bf16 panel -> ring stage `stage` (swizzled 128-B rows, see stage_panel)
static __device__ __forceinline__ void a3_dma_block(const bf16_t* __restrict__ panel, int ld, int blk, unsigned char* ring, int stage,
                                                    int wid, int lane) {
  const int row = blk * 64 + wid * 8 + (lane >> 3);
  const int pos = lane & 7;
  glds16(panel + (size_t)row * ld + ((pos ^ kc_swz(row)) << 3), ring + stage * 8192 + wid * 1024);
}
// every wave stages one 1-KiB piece of the mask row (S floats): waves beyond the row re-read its first piece into their own
// (unused) KiB so that all waves issue the same number of DMA instructions
static __device__ __forceinline__ void a3_dma_mask(const float* __restrict__ src, int S, float* dst, int wid, int lane) {
  const int npiece = (S * 4 + 1023) >> 10;
  const int pc = wid < npiece ? wid : 0;
  int i = pc * 256 + lane * 4;
  if (i >= S) i = 0;
  glds16(src + i, reinterpret_cast<unsigned char*>(dst) + wid * 1024);
}
// stationary fragment (16 rows x 32 k) straight from global memory; waited for by a3_wait_frags
static __device__ __forceinline__ void a3_load_frag(bf16x8& dst, const bf16_t* __restrict__ base, int ld, int r0, int ks, int lane) {
  const bf16_t* p = base + (size_t)(r0 + (lane & 15)) * ld + ks * 32 + (lane >> 4) * 8;
  s8v v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  dst = __builtin_bit_cast(bf16x8, v);
}
// N = VMEM instructions this wave has issued since the fragment loads (all younger ones may stay in flight)
template <int N>
static __device__ __forceinline__ void a3_wait_frags(bf16x8 (&f)[2][2]) {
  s8v a = __builtin_bit_cast(s8v, f[0][0]), b = __builtin_bit_cast(s8v, f[0][1]), c = __builtin_bit_cast(s8v, f[1][0]),
      d = __builtin_bit_cast(s8v, f[1][1]);
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N) : "memory");
  f[0][0] = __builtin_bit_cast(bf16x8, a);
  f[0][1] = __builtin_bit_cast(bf16x8, b);
  f[1][0] = __builtin_bit_cast(bf16x8, c);
  f[1][1] = __builtin_bit_cast(bf16x8, d);
}
// Anchor after a counted wait (a3_wait_vm_n): the fragments are "redefined" here, so every consumer is scheduled behind the wait.
// (One statement with tied operands: a switch over per-count statements made hipcc copy the registers BEFORE the wait.)
static __device__ __forceinline__ void a3_frags_landed(bf16x8 (&f)[2][2]) {
  s8v a = __builtin_bit_cast(s8v, f[0][0]), b = __builtin_bit_cast(s8v, f[0][1]), c = __builtin_bit_cast(s8v, f[1][0]),
      d = __builtin_bit_cast(s8v, f[1][1]);
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
  f[0][0] = __builtin_bit_cast(bf16x8, a);
  f[0][1] = __builtin_bit_cast(bf16x8, b);
  f[1][0] = __builtin_bit_cast(bf16x8, c);
  f[1][1] = __builtin_bit_cast(bf16x8, d);
}
static __device__ __forceinline__ void a3_store8(void* p, uint2 v) {
  asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
static __device__ __forceinline__ void a3_store4(void* p, float v) {
  asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
static __device__ __forceinline__ void a3_wait_vm_n(int n) {   // at most n (<= 31) VMEM instructions of this wave outstanding
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
  }
}
// opaque to the optimiser: otherwise it splits "lane part + ring base + block offset" and re-adds the constant before every
// read (24 v_add_u32 per block) instead of using the instructions' 16-bit immediate offsets
static __device__ __forceinline__ unsigned a3_opaque(unsigned a) {
  asm volatile("" : "+v"(a));
  return a;
}
static __device__ __forceinline__ const unsigned char* a3_lds(unsigned off) { return (const unsigned char*)(lds_void*)(size_t)off; }

// mask metadata of an item from the LDS copy of its mask row, computed redundantly by every wave (no barrier):
// klen = 1 + last unmasked key, nfree = leading 64-key blocks without a masked key (prefix masks), else 0
static __device__ __forceinline__ void a3_mask_meta(const float* sMask, int S, int lane, int& klen, int& nfree) {
  int last = 0, cnt = 0;
  for (int i = 0; i < S / 64; ++i) {
    const unsigned long long bal = __ballot(sMask[i * 64 + lane] > -1.0f);   // additive bias 0 = attend (near -10000 = masked)
    if (bal) last = i * 64 + 64 - __builtin_clzll(bal);
    cnt += __builtin_popcountll(bal);
  }
  klen = last;
  nfree = (cnt == last) ? (last >> 6) : 0;
  if (last == 0) {   // every key masked: the reference's softmax is then over the masked scores themselves
    klen = S;
    nfree = 0;
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
struct A3Fwd {
  f4v s[2][4];      // [row block][key fragment]: raw score sums q.k of the current 64-key block (+ mask / scale)
  f4v o[2][4];      // [row block][d block]: O^T accumulators
  f4v osum[2];      // row sums (every register of a lane holds the sum of query lane & 15)
  float m[2];       // row reference (log2 domain), query = lane & 15 of row block j
};

static __device__ __forceinline__ float a3_rowmax2(const f4v a, const f4v b) {
  float m = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
  m = fmaxf(fmaxf(m, b[0]), b[1]);
  m = fmaxf(fmaxf(m, b[2]), b[3]);
  return group4_max(m);
}

// P = exp2(scale2 * score - m) of one 32-key chunk c of the current block, packed to bf16 B fragments (one per row block);
// returns the OR of the packed words (overflow check).  DROP: pbd = dropped probabilities (what multiplies V), pb = undropped
// (row sums).
template <bool DROP>
static __device__ __forceinline__ uint32_t a3_probs(const f4v (&sc)[2][4], int c, bf16x8 (&pb)[2], bf16x8 (&pbd)[2],
                                                    const uint32_t* sCk, int key0, int g, const uint32_t (&rk)[2],
                                                    uint32_t drop_thresh, float scale2, const float (&m)[2]) {
  uint32_t acc = 0u;
  uint32_t ck[2][4];
  if (DROP) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const uint4 t = *reinterpret_cast<const uint4*>(sCk + key0 + (2 * c + f) * 16 + g * 4);
      ck[f][0] = t.x; ck[f][1] = t.y; ck[f][2] = t.z; ck[f][3] = t.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    f4v p[2], pd[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
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
static __device__ __forceinline__ bf16x8 a3_scale_packed(const bf16x8 v, float c) {
  union {
    uint32_t u[4];
    bf16x8 v;
  } x;
  x.v = v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f2v t = unpack2bf(x.u[i]);
    x.u[i] = pack2bf(t[0] * c, t[1] * c);
  }
  return x.v;
}

#define A3_ONES() __builtin_bit_cast(bf16x8, (s8v){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80})

// P.V (+ row sums) of one 32-key chunk: 10 MFMAs
template <bool DROP>
static __device__ __forceinline__ void a3_pv(A3Fwd& st, const bf16x8 (&pb)[2], const bf16x8 (&pbd)[2], const bf16x8 (&v)[4]) {
  const bf16x8 ones = A3_ONES();
  st.osum[0] = MFMA(ones, pb[0], st.osum[0]);
  st.osum[1] = MFMA(ones, pb[1], st.osum[1]);
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    st.o[0][db] = MFMA(v[db], DROP ? pbd[0] : pb[0], st.o[0][db]);
    st.o[1][db] = MFMA(v[db], DROP ? pbd[1] : pb[1], st.o[1][db]);
  }
}

// One 64-key block kb, three instruction groups per wave (LDS reads are issued one group before the MFMAs that consume them):
//   G1  reads K(kb+1) fragments 0-1                     | exp2 of the block's first 32 keys |
//   G2  reads V^T(kb) first half, K(kb+1) fragments 2-3 | exp2 of the last 32 keys          | scores of block kb+1, fragments 0-1 (PF)
//   G3  reads V^T(kb) second half                       |                                   | P.V of all 64 keys, scores of kb+1 fragments 2-3
// The score accumulators of block kb+1 overwrite those of block kb fragment by fragment as they are consumed (one tile live).
// vp0: this lane's V^T fragment address of block kb, d block 0; kn0: its K fragment address (k-step 0) of block kb+1; the others
// differ from them by an XOR (the swizzle is an XOR of the 16-byte column index, stage bases are multiples of 8 KiB): d block
// db -> ^ (db << 5), k-step 1 -> ^ 64.  pfm: block kb+1 holds masked keys (sMask + key0 + 64: its mask values, added as
// mask / scale to the raw sums).
template <bool PF, bool DROP, bool PROF, int SGB>
static __device__ __forceinline__ void a3_fwd_step(A3Prof& pr, A3Fwd& st, const bf16x8 (&qf)[2][2], unsigned vp0, unsigned kn0,
                                                   bool pfm, const float* sMask, const uint32_t* sCk, int key0, int g,
                                                   const uint32_t (&rk)[2], uint32_t drop_thresh, float scale2) {
  const unsigned vp[4] = {vp0, a3_opaque(vp0 ^ 32u), a3_opaque(vp0 ^ 64u), a3_opaque(vp0 ^ 96u)};
  const unsigned kn1 = a3_opaque(kn0 ^ 64u);
  f4v(&sc)[2][4] = st.s;
  const f4v zero4 = (f4v){0.f, 0.f, 0.f, 0.f};
  bf16x8 v0[4], v1[4], pb0[2], pb0d[2], pb1[2], pb1d[2], ka[2][2], kc[2][2];
  // ---- G1
  if (PF) {
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
      ka[0][kf] = kc_at(a3_lds(kn0), kf * 2048);
      ka[1][kf] = kc_at(a3_lds(kn1), kf * 2048);
    }
  }
  uint32_t chk = a3_probs<DROP>(sc, 0, pb0, pb0d, sCk, key0, g, rk, drop_thresh, scale2, st.m);
  if (__any((chk & 0x40004000u) != 0u)) {
    // some probability >= 2: raise the reference of the rows whose chunk maximum came within the headroom, rescale what has been
    // accumulated and redo the exponentials
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float d = fmaxf(0.0f, a3_rowmax2(sc[j][0], sc[j][1]) * scale2 - st.m[j] + A3_MARGIN);
      const float f = __builtin_amdgcn_exp2f(-d);
#pragma unroll
      for (int db = 0; db < 4; ++db) st.o[j][db] *= f;
      st.osum[j] *= f;
      st.m[j] += d;
    }
    a3_probs<DROP>(sc, 0, pb0, pb0d, sCk, key0, g, rk, drop_thresh, scale2, st.m);
  }
  a3_stamp<PROF>(pr, 8);
  __builtin_amdgcn_sched_barrier(0);
  // ---- G2
#pragma unroll
  for (int db = 0; db < 4; ++db) v0[db] = tr_at(a3_lds(vp[db]), 0);
  if (PF) {
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
      kc[0][kf] = kc_at(a3_lds(kn0), (kf + 2) * 2048);
      kc[1][kf] = kc_at(a3_lds(kn1), (kf + 2) * 2048);
    }
  }
  // the second chunk's probabilities read sc[.][2..3]; the score MFMAs of this group overwrite sc[.][0..1] only
  chk = a3_probs<DROP>(sc, 1, pb1, pb1d, sCk, key0, g, rk, drop_thresh, scale2, st.m);
  if (PF) {
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
      sc[0][kf] = MFMA(ka[0][kf], qf[0][0], zero4);
      sc[1][kf] = MFMA(ka[0][kf], qf[1][0], zero4);
    }
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
      sc[0][kf] = MFMA(ka[1][kf], qf[0][1], sc[0][kf]);
      sc[1][kf] = MFMA(ka[1][kf], qf[1][1], sc[1][kf]);
    }
    if (SGB) {
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }
  }
  if (__any((chk & 0x40004000u) != 0u)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float d = fmaxf(0.0f, a3_rowmax2(sc[j][2], sc[j][3]) * scale2 - st.m[j] + A3_MARGIN);
      const float f = __builtin_amdgcn_exp2f(-d);
#pragma unroll
      for (int db = 0; db < 4; ++db) st.o[j][db] *= f;
      st.osum[j] *= f;
      pb0[j] = a3_scale_packed(pb0[j], f);   // the first chunk's probabilities have not been multiplied into O yet
      if (DROP) pb0d[j] = a3_scale_packed(pb0d[j], f);
      st.m[j] += d;
    }
    a3_probs<DROP>(sc, 1, pb1, pb1d, sCk, key0, g, rk, drop_thresh, scale2, st.m);
  }
  a3_stamp<PROF>(pr, 9);
  __builtin_amdgcn_sched_barrier(0);
  // ---- G3
#pragma unroll
  for (int db = 0; db < 4; ++db) v1[db] = tr_at(a3_lds(vp[db]), 4096);
  a3_pv<DROP>(st, pb0, pb0d, v0);
  if (PF) {
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
      sc[0][kf + 2] = MFMA(kc[0][kf], qf[0][0], zero4);
      sc[1][kf + 2] = MFMA(kc[0][kf], qf[1][0], zero4);
    }
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
      sc[0][kf + 2] = MFMA(kc[1][kf], qf[0][1], sc[0][kf + 2]);
      sc[1][kf + 2] = MFMA(kc[1][kf], qf[1][1], sc[1][kf + 2]);
    }
  }
  a3_pv<DROP>(st, pb1, pb1d, v1);
  if (PF && pfm) {   // (wave-uniform, rare: at most the last block of a prefix-masked sentence)
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      const f4v mk = *reinterpret_cast<const f4v*>(sMask + key0 + 64 + kf * 16 + g * 4) * 8.0f;   // mask / scale (scale = 1/8)
      sc[0][kf] += mk;
      sc[1][kf] += mk;
    }
  }
  a3_stamp<PROF>(pr, 10);
}

template <bool DROP, bool PROF, int SGB>
__global__ __launch_bounds__(512, 2) void attn_fwd3_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ maskbias,
                                                           bf16_t* __restrict__ ctx, float* __restrict__ lse, int S, int H, int A,
                                                           float scale, uint32_t drop_seed, uint32_t drop_thresh, int nitems, int dbg,
                                                           unsigned long long* __restrict__ profout, int rolln) {
  A3Prof pr;
  if (PROF) {
#pragma unroll
    for (int i = 0; i < 12; ++i) pr.acc[i] = 0;
    pr.last = __builtin_amdgcn_s_memtime();
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;                                                      // ring of 8 stages x (64 keys x 128 B)
  unsigned char* sV = smem + A3_RING;
  float* sMaskB = reinterpret_cast<float*>(smem + 2 * A3_RING);                  // [2][2048] mask rows (slot = item parity)
  uint32_t* sCkB = reinterpret_cast<uint32_t*>(smem + 2 * A3_RING + 2 * 8192);   // [2][AT_MAXS] dropout column keys
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int ld = 3 * H;
  const float scale2 = scale * 1.4426950408889634f;
  const float dscale = DROP ? drop_scale(drop_thresh) : 1.0f;
  const int npass = (S + A3_ROWS - 1) / A3_ROWS;
  const int nb = S / 64;   // 64-row blocks per panel

  // item = (batch b, head h, query tile t): items of this workgroup are blockIdx.x, + gridDim.x, ...
  int item = (int)blockIdx.x;
  int seq = 0;   // items done by this workgroup: ring phase and mask slot
  {
    const int t0 = item / npass;
    const int h = t0 % A, b = t0 / A;
    a3_dma_mask(maskbias + (size_t)b * S, S, sMaskB, wid, lane);
    for (int blk = 0; blk < nb; ++blk) {
      a3_dma_block(qkv + (size_t)b * S * ld + h * AT_D + H, ld, blk, sK, blk, wid, lane);
      a3_dma_block(qkv + (size_t)b * S * ld + h * AT_D + 2 * H, ld, blk, sV, blk, wid, lane);
    }
  }
  // per-lane fragment address parts (see kc_frag / tr_frag): K rows f*16 + li -> + f*2048 ; V rows kc*32 + g*4 + (li>>2) -> + kc*4096
  const unsigned kl0 = li * 128 + (((0 * 4 + g) ^ kc_swz(li)) << 4);
  const int vrow = g * 4 + (li >> 2);
  const unsigned vl0 = vrow * 128 + (((0 * 2 + ((li & 3) >> 1)) ^ kc_swz(vrow)) << 4) + ((li & 1) << 3);
  const unsigned sKo = (unsigned)(size_t)(lds_void*)sK, sVo = (unsigned)(size_t)(lds_void*)sV;

  bf16x8 qf[2][2];   // the item's stationary Q fragments; reloaded for the NEXT item before the last step (which has no score MFMAs)
  {
    const int t0 = item / npass, pass = item % npass;
    const int h = t0 % A, b = t0 / A;
    const bf16_t* base = qkv + (size_t)b * S * ld + h * AT_D;
    int q0 = pass * A3_ROWS + wid * 32;
    if (q0 >= S) q0 = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      a3_load_frag(qf[j][0], base, ld, q0 + j * 16, 0, lane);
      a3_load_frag(qf[j][1], base, ld, q0 + j * 16, 1, lane);
    }
  }
  int top_wait = 0;   // how many of this wave's youngest VMEM instructions may still be in flight at the top of the next item
  for (;;) {
    const int t0 = item / npass, pass = item % npass;
    const int h = t0 % A, b = t0 / A;
    const uint32_t bhS = (uint32_t)((b * A + h) * S);
    const int next = item + (int)gridDim.x;
    const bool has_next = next < nitems;
    const int nt0 = next / npass, npi = next % npass;
    const int nh = nt0 % A, nbh = nt0 / A;
    const int slot = seq & 1;
    const float* sMask = sMaskB + slot * 2048;
    uint32_t* sCk = sCkB + slot * AT_MAXS;
    const int ring0 = (seq * nb) & 7;     // stage of this item's block 0
    const int ringn = (ring0 + nb) & 7;   // stage of the next item's block 0
    const int q0 = pass * A3_ROWS + wid * 32;
    const bool active = q0 < S;
    a3_stamp<PROF>(pr, 11);
    // Everything this item reads in its first two steps (its Q fragments, K blocks 0..2, V blocks 0..1, the mask row) is older
    // than the youngest `top_wait` instructions when the previous item rolled its blocks in over >= 6 steps (end of the loop)
    if (!(dbg & 2)) a3_wait_vm_n(top_wait);
    a3_frags_landed(qf);
    a3_barrier();
    a3_stamp<PROF>(pr, 0);
    int klen, nfree;
    a3_mask_meta(sMask, S, lane, klen, nfree);
    const int nkb = (klen + 63) >> 6;   // blocks behind the last unmasked key contribute exp(-10000 + x) = 0 exactly
    if (DROP) {
      if (tid < S) sCk[tid] = drop_colkey(drop_seed, bhS + (uint32_t)tid);
      a3_barrier();
    }
    // (dbg bit 16: timing experiment, always re-fetch the first head's panels -- L2-resident sources)
    const bf16_t* nK = (dbg & 16) ? qkv + H : qkv + (size_t)nbh * S * ld + nh * AT_D + H;
    const bf16_t* nV = nK + H;
    A3Fwd st;
    uint32_t rk[2] = {0u, 0u};
    const unsigned kb_off = (unsigned)ring0 * 8192u;
    if (active) {
      const f4v zero4 = (f4v){0.f, 0.f, 0.f, 0.f};
      const unsigned k0 = a3_opaque(sKo + kb_off + kl0), k1 = a3_opaque(k0 ^ 64u);
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) {
        const bf16x8 a0 = kc_at(a3_lds(k0), kf * 2048), a1 = kc_at(a3_lds(k1), kf * 2048);
        st.s[0][kf] = MFMA(a0, qf[0][0], zero4);
        st.s[1][kf] = MFMA(a0, qf[1][0], zero4);
        st.s[0][kf] = MFMA(a1, qf[0][1], st.s[0][kf]);
        st.s[1][kf] = MFMA(a1, qf[1][1], st.s[1][kf]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (DROP) rk[j] = drop_rowkey(drop_seed, bhS + (uint32_t)(q0 + j * 16 + li));
        st.osum[j] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int db = 0; db < 4; ++db) st.o[j][db] = (f4v){0.f, 0.f, 0.f, 0.f};
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
      for (int j = 0; j < 2; ++j)
        st.m[j] = fmaxf(a3_rowmax2(st.s[j][0], st.s[j][1]), a3_rowmax2(st.s[j][2], st.s[j][3])) * scale2 + A3_MARGIN;
    }
    a3_stamp<PROF>(pr, 1);
    // ---- steps
#define A3_STEP(PF, KB)                                                                                                   \
  if (active) {                                                                                                                  \
    const unsigned ov = (kb_off + (unsigned)(KB) * 8192u) & (A3_RING - 1);                                                       \
    const unsigned ok = (kb_off + (unsigned)((KB) + 1) * 8192u) & (A3_RING - 1);                                                 \
    const unsigned vp0 = a3_opaque(sVo + ov + vl0), kn0 = a3_opaque(sKo + ok + kl0);                                             \
    a3_fwd_step<PF, DROP, PROF, SGB>(pr, st, qf, vp0, kn0, (KB) + 1 >= nfree, sMask, sCk, (KB) * 64, g, rk, drop_thresh,         \
                                            scale2);                                                                             \
  }
    // After every `rolln`-th step (not the last one: its blocks go out with the final batch): every wave is done with K blocks
    // <= kb and V blocks <= kb -- roll in the next item's.  After step 0 everything the previous item sent has landed.
#define A3_ROLL(KB)                                                                                          \
  {                                                                                                          \
    const bool roll = has_next && !(dbg & 1) && (((KB) + 1) % rolln == 0);                                   \
    if ((KB) == 0 && !(dbg & 2)) a3_wait_vm_n(0);                                                            \
    if (((KB) == 0 || roll) && !(dbg & 4)) a3_barrier();                                                     \
    if (roll) {                                                                                              \
      if ((KB) + 1 == rolln) a3_dma_mask(maskbias + (size_t)nbh * S, S, sMaskB + (slot ^ 1) * 2048, wid, lane); \
      for (int blk = (KB) + 1 - rolln; blk <= (KB); ++blk) a3_dma_block(nK, ld, blk, sK, (ringn + blk) & 7, wid, lane); \
      for (int blk = (KB)-rolln; blk < (KB); ++blk)                                                          \
        if (blk >= 0 && !(dbg & 8)) a3_dma_block(nV, ld, blk, sV, (ringn + blk) & 7, wid, lane);             \
    }                                                                                                        \
  }                                                                                                          \
  a3_stamp<PROF>(pr, 2);
    // the next item's stationary fragments, requested before the last step (qf is dead: no score MFMAs in it)
#define A3_NEXTQ()                                                                  \
  if (has_next) {                                                                   \
    const bf16_t* nbase = qkv + (size_t)nbh * S * ld + nh * AT_D;                   \
    int nq0 = npi * A3_ROWS + wid * 32;                                             \
    if (nq0 >= S) nq0 = 0;                                                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                 \
      a3_load_frag(qf[j][0], nbase, ld, nq0 + j * 16, 0, lane);                     \
      a3_load_frag(qf[j][1], nbase, ld, nq0 + j * 16, 1, lane);                     \
    }                                                                               \
  }
    for (int kb = 0; kb + 1 < nkb; ++kb) {
      A3_STEP(true, kb)
      A3_ROLL(kb)
    }
    A3_NEXTQ()
    A3_STEP(false, nkb - 1)
#undef A3_STEP
#undef A3_ROLL
#undef A3_NEXTQ
    a3_stamp<PROF>(pr, 3);
    int nfinal = 0;
    if (has_next && !(dbg & 1)) {
      // every wave has issued its last LDS read of this item: the remaining
      // blocks of the next item -- K blocks 0 .. nkb-2 and V blocks 0 .. nkb-3 went out behind the steps
      a3_barrier();
      const int nrolled = ((nkb - 1) / rolln) * rolln;   // K blocks 0 .. nrolled-1 and V blocks 0 .. nrolled-2 went out behind steps
      const int kfirst = nrolled;
      const int vfirst = nrolled > 0 ? nrolled - 1 : 0;
      if (nrolled == 0) a3_dma_mask(maskbias + (size_t)nbh * S, S, sMaskB + (slot ^ 1) * 2048, wid, lane);
      for (int blk = kfirst; blk < nb; ++blk) a3_dma_block(nK, ld, blk, sK, (ringn + blk) & 7, wid, lane);
      if (!(dbg & 8))
        for (int blk = vfirst; blk < nb; ++blk) a3_dma_block(nV, ld, blk, sV, (ringn + blk) & 7, wid, lane);
      nfinal = nrolled >= 2 ? (nb - kfirst) + (nb - vfirst) : -1;
    }
    a3_stamp<PROF>(pr, 5);
    if (active) {
      // O^T fragment: lane holds O[q0 + j*16 + li][db*16 + g*4 .. +3]; 10 store instructions per wave (counted below)
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
          a3_store8(orow + db * 16 + g * 4, u);
        }
        float* lp = lse + ((size_t)b * A + h) * S + q0 + j * 16 + li;
        if (g != 0) lp = lse + ((size_t)b * A + h) * S + q0 + j * 16 + li;   // (same address for every g: one store instruction,
        a3_store4(lp, (st.m[j] + __log2f(sum)) * 0.6931471805599453f);        //  identical values -- keeps EXEC full and the count fixed)
      }
    }
    a3_stamp<PROF>(pr, 4);
    if (!has_next) break;
    // at the top of the next item its Q fragments, mask row, K blocks 0..1 and V block 0 (what its step 0 reads) must have
    // landed: they are older than the final batch when at least two blocks were rolled in behind steps; younger than the final
    // batch are only this wave's 10 output stores.  Otherwise drain.  (After its step 0 the next item waits for everything.)
    top_wait = (nfinal >= 0 && nfinal + 10 <= 31) ? nfinal + (active ? 10 : 0) : 0;
    item = next;
    ++seq;
  }
  if (PROF && blockIdx.x < 4 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) profout[(blockIdx.x * 8 + wid) * 12 + i] = pr.acc[i];
  }
}

template <bool DROP>
static int launch_fwd3(const bf16_t* qkv, const float* maskbias, bf16_t* ctx, float* lse, int B, int S, int H, int A, uint32_t seed,
                       uint32_t thresh, hipStream_t stream) {
#ifdef A3_LAB
  // lab build only (hipcc -DA3_LAB ...; tools/micro/attn_lab): debug switches from the environment, a profile buffer that is
  // allocated, synchronised on and printed here.  The product library compiles none of this: its entry point allocates nothing,
  // never synchronises and reads no environment.
  static int dbg = -1, prof = 0, sgb = 0, rolln = 1;
  static unsigned long long* profbuf = nullptr;
  static std::atomic<unsigned long long> done0{0}, done1{0}, done2{0};
  int r = kbner_set_max_lds_once(done0, reinterpret_cast<const void*>(attn_fwd3_kernel<DROP, false, 0>), A3_LDS_BYTES);
  if (r) return r;
  r = kbner_set_max_lds_once(done1, reinterpret_cast<const void*>(attn_fwd3_kernel<DROP, false, 1>), A3_LDS_BYTES);
  if (r) return r;
  r = kbner_set_max_lds_once(done2, reinterpret_cast<const void*>(attn_fwd3_kernel<DROP, true, 0>), A3_LDS_BYTES);
  if (r) return r;
  const int ncu = kbner_cu_count();
  if (dbg < 0) {
    const char* e = getenv("KBNER_ATTN_DBG");
    dbg = e ? atoi(e) : 0;
    e = getenv("KBNER_ATTN_PROF");
    prof = e ? atoi(e) : 0;
    e = getenv("KBNER_ATTN_SGB");
    sgb = e ? atoi(e) : 0;
    e = getenv("KBNER_ATTN_ROLL");
    rolln = e ? atoi(e) : 1;
    if (rolln < 1) rolln = 1;
    if (prof && hipMalloc(&profbuf, 4 * 8 * 12 * sizeof(unsigned long long)) != hipSuccess) prof = 0;
  }
  const int nitems = B * A * ((S + A3_ROWS - 1) / A3_ROWS);
  const int grid = nitems < ncu ? nitems : ncu;
  if (prof) {
    hipLaunchKernelGGL((attn_fwd3_kernel<DROP, true, 0>), dim3(grid), dim3(512), A3_LDS_BYTES, stream, qkv, maskbias, ctx, lse, S, H, A,
                       0.125f, seed, thresh, nitems, dbg, profbuf, rolln);
    if (prof == 1) {   // print once: region cycles per wave of workgroups 0..3
      prof = 2;
      unsigned long long h[4 * 8 * 12];
      if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h, profbuf, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
        static const char* nm[12] = {"wait+bar", "S0", "roll", "tail", "flush+epi", "final-dma", "-", "-", "G1", "G2", "G3", "top"};
        for (int w = 0; w < 32; w += 5) {
          fprintf(stderr, "fwd3 prof wg%d wave%d:", w / 8, w % 8);
          for (int i = 0; i < 12; ++i) fprintf(stderr, " %s=%llu", nm[i], h[w * 12 + i]);
          fprintf(stderr, "\n");
        }
      }
    }
  } else if (sgb) {
    hipLaunchKernelGGL((attn_fwd3_kernel<DROP, false, 1>), dim3(grid), dim3(512), A3_LDS_BYTES, stream, qkv, maskbias, ctx, lse, S, H, A,
                       0.125f, seed, thresh, nitems, dbg, (unsigned long long*)nullptr, rolln);
  } else {
    hipLaunchKernelGGL((attn_fwd3_kernel<DROP, false, 0>), dim3(grid), dim3(512), A3_LDS_BYTES, stream, qkv, maskbias, ctx, lse, S, H, A,
                       0.125f, seed, thresh, nitems, dbg, (unsigned long long*)nullptr, rolln);
  }
#else
  static std::atomic<unsigned long long> done0{0};
  const int r = kbner_set_max_lds_once(done0, reinterpret_cast<const void*>(attn_fwd3_kernel<DROP, false, 0>), A3_LDS_BYTES);
  if (r) return r;
  const int ncu = kbner_cu_count();
  const int nitems = B * A * ((S + A3_ROWS - 1) / A3_ROWS);
  const int grid = nitems < ncu ? nitems : ncu;
  hipLaunchKernelGGL((attn_fwd3_kernel<DROP, false, 0>), dim3(grid), dim3(512), A3_LDS_BYTES, stream, qkv, maskbias, ctx, lse, S, H, A,
                     0.125f, seed, thresh, nitems, 0, (unsigned long long*)nullptr, 1);
#endif
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// entry point used by attention.hip's dispatcher (same contract as kbner_attn_fwd)
int kbner_attn_fwd3(const bf16_t* qkv, const float* maskbias, bf16_t* ctx, float* lse, int B, int S, int H, int A,
                    uint32_t drop_seed, uint32_t drop_thresh, hipStream_t stream) {
  if (drop_thresh) return launch_fwd3<true>(qkv, maskbias, ctx, lse, B, S, H, A, drop_seed, drop_thresh, stream);
  return launch_fwd3<false>(qkv, maskbias, ctx, lse, B, S, H, A, drop_seed, drop_thresh, stream);
}
