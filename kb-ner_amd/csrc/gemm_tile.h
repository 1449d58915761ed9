// Shared pieces of the 256-column-tile bf16 MFMA GEMM kernels (gemm256.hip: 256 x 256 tiles; gemm128x.hip: 128 x 256 tiles with
// the previous tile's epilogue running under the current tile's K loop): problem descriptors, LDS-DMA issue helpers, LDS image
// swizzles, the XCD-aware tile walk.  gfx950 only.
#pragma once
#include "common.h"

#define EPI_BIAS 1
#define EPI_GELU 2   // C = gelu(pre), out2 = gelu'(pre) (bf16): what the backward EPI_DGELU multiplies by
#define EPI_ADD 4
#define EPI_DGELU 8
#define EPI_ATOMIC32 16
#define EPI_RMW32 32  // C32[m,n] += result, non-atomic 16-byte RMW (each output element owned by one lane)
#define EPI_COLSUM 64  // colsum[n] += sum_m out[m,n] (bias gradient of the producing layer), fp32 atomics, 2 per column per tile
#define EPI_STORE32 256  // C32[m,n] = result (fp32, plain stores): one split-K slab, summed by kbner_splitk_finish
#define EPI_COLSUM_WS 512  // with EPI_COLSUM: colsum is a workspace f32 [2 * M/256, N]; row 2*tile_row + wave_row receives this
                           // tile's column sums by plain stores (no atomics); kbner_colsum_rows_f32 folds the rows afterwards
#define EPI_GELU_FWD 1024  // C = gelu(pre), NO derivative output: the forward of inference (evaluate, frozen stack encoders)
#define EPI_DROP 128   // dropout on (acc*alpha + bias) BEFORE the residual add (BertSelfOutput / BertOutput); not with COLSUM

#define G2_MAXP 16
#ifndef KBNER_GEMM_VARIANT_DEFAULT
#define KBNER_GEMM_VARIANT_DEFAULT 3
#endif

struct GemmProblem {
  const bf16_t* A;
  const bf16_t* B;
  bf16_t* C;
  float* C32;
  const float* bias;
  const bf16_t* addend;
  const bf16_t* aux;
  bf16_t* out2;
  float* colsum;
  int M, N, K;
  int lda, ldb, ldc, ldc32, ldadd, ldaux, ldout2;
  int epi;
  float alpha;
  int tile_begin;
  uint32_t drop_seed;
  uint32_t drop_thresh;
  int pad_;
};

struct GroupArgs {
  int nprob;
  int total_tiles;
  int ncu;
  int pad_;
  int* sched;  // dynamic tile scheduling (DYN kernels): 8 per-XCD counters, zeroed by the caller before the launch
  int tile_begin[G2_MAXP];  // compact copy of p[i].tile_begin: one s_load_dwordx16 picks the problem
  GemmProblem p[G2_MAXP];
};

#define T2 256
#define BK2 64
#define TILE2_BYTES 32768
#define STAGE2_BYTES 65536
#define G2_LDS_BYTES (2 * STAGE2_BYTES + 8 * 4096)  // two operand stages + two 2-KiB epilogue transpose buffers per wave

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_cvoid;

// LDS-DMA issued from inline asm so the compiler does not see it: with a compiler-visible
// global_load_lds in flight hipcc (ROCm 7.2) degrades every ds_read wait in the loop to lgkmcnt(0)
// (mixed pending LGKM event types), which serialises the fragment stream behind full LDS latency.
// The DMA's completion is ordered by hand: s_waitcnt vmcnt(0) + barrier before the stage is read.
// Address form: wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset.  The per-lane part is
// loop-invariant, so only 8 VGPRs (not 8 x 64-bit pointers) stay live across the MFMA loop.
static __device__ __forceinline__ void glds16(const void* sbase, unsigned voff, void* l) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)l);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(dst)
               : "memory");
}
// A wave's 2 or 4 consecutive 1-KiB pieces under ONE M0 write: the instruction's immediate offset advances BOTH the LDS and the
// global address by j KiB, so the j-th per-lane offset is passed as voff_j - j * 1024 (never negative: a piece step is >= 1 KiB
// of source bytes for every image, ld >= 64).  Saves three of the four (s_mov m0 / s_nop / restore) sequences per operand tile.
static __device__ __forceinline__ void glds16x4(const void* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned v3, void* l) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)l);
  // (m0 is declared clobbered instead of being saved and restored: nothing else in these kernels keeps a value in it)
  asm volatile(
      "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %4\n\t"
      "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
      "global_load_lds_dwordx4 %2, %4 offset:2048\n\t"
      "global_load_lds_dwordx4 %3, %4 offset:3072"
      :
      : "v"(v0), "v"(v1 - 1024u), "v"(v2 - 2048u), "v"(v3 - 3072u), "s"(sbase), "s"(dst)
      : "memory", "m0");
}
static __device__ __forceinline__ void glds16x2(const void* sbase, unsigned v0, unsigned v1, void* l) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)l);
  asm volatile(
      "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %2\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:1024"
      :
      : "v"(v0), "v"(v1 - 1024u), "s"(sbase), "s"(dst)
      : "memory", "m0");
}
static __device__ __forceinline__ int kc_swz(int row) { return (row >> 1) & 7; }
static __device__ __forceinline__ int ks_swz(int krow) { return (krow & 3) | ((krow >> 1) & 4); }
// B operand, KC image: a fragment's 16 lanes read rows {8a + 4h + c : a,c in 0..3} of a 32-row block (the column
// permutation below), so the conflict-free chunk swizzle keys on row bits 4,3 and 1
static __device__ __forceinline__ int kcb_swz(int row) { return (((row >> 3) & 3) << 1) | ((row >> 1) & 1); }

// Output-column permutation: MFMA fragment ni (0..3) of a wave's 64 columns takes operand row j (= lane & 15) from
// column (ni>>1)*32 + (j>>2)*8 + (ni&1)*4 + (j&3).  With the operand-swapped MFMA a lane then owns 8 CONTIGUOUS
// output columns per fragment pair (g*8 .. g*8+7 of each 32-column block): every epilogue access is 16 bytes
// (dwordx4) instead of 8 -- the row-per-lane store tail is issue-bound, halving the instruction count halves it.

template <bool KS, bool ISB, int ROWS = 256>
static __device__ __forceinline__ void stage256(const bf16_t* __restrict__ P, int ld, int row0, int k0, unsigned char* s, int wid,
                                                int lane) {
  static_assert(ROWS == 256 || (ROWS == 128 && !KS), "half-height tiles exist for the row-major (KC) A image only");
  // uniform tile origin in SGPRs
  const bf16_t* sbase = KS ? P + (size_t)k0 * ld + row0 : P + (size_t)row0 * ld + k0;
  unsigned voff[ROWS / 64];
#pragma unroll
  for (int j = 0; j < ROWS / 64; ++j) {
    const int q = wid * (ROWS / 64) + j;  // ROWS / 8 wave-instructions x 1 KiB (32 KiB for a 256-row tile)
    if (!KS) {
      const int row = q * 8 + (lane >> 3);
      const int pos = lane & 7;
      voff[j] = (unsigned)(row * ld + ((pos ^ (ISB ? kcb_swz(row) : kc_swz(row))) << 3)) * 2u;
    } else {
      const int kr = q * 2 + (lane >> 5);
      const int pos = lane & 31;
      voff[j] = (unsigned)(kr * ld + ((pos ^ (ks_swz(kr) << 1)) << 3)) * 2u;
    }
  }
  unsigned char* dst = s + wid * (ROWS / 64) * 1024;   // this wave's pieces are consecutive in the image
  if constexpr (ROWS == 256) glds16x4(sbase, voff[0], voff[1], voff[2], voff[3], dst);
  else glds16x2(sbase, voff[0], voff[1], dst);
}

template <bool KS>
static __device__ __forceinline__ bf16x8 frag256(const unsigned char* s, int r0, int ks, int lane) {
  if (!KS) {
    const int row = r0 + (lane & 15);
    const int c = ks * 4 + (lane >> 4);
    const s8v v = *reinterpret_cast<const s8v*>(s + row * 128 + ((c ^ kc_swz(row)) << 4));
    return __builtin_bit_cast(bf16x8, v);
  } else {
    const int p = lane & 15;
    const int r = ks * 32 + (lane >> 4) * 8 + (p >> 2);
    const unsigned char* a = s + r * 512 + ((((r0 >> 4) ^ ks_swz(r))) << 5) + ((p & 3) << 3);
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a + 4 * 512));
    s8v v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  }
}

// B-operand fragment ni of the wave whose columns start at c0 (see the permutation note above)
template <bool KS>
static __device__ __forceinline__ bf16x8 fragB256(const unsigned char* s, int c0, int ni, int ks, int lane) {
  if (!KS) {
    const int j = lane & 15;
    const int row = c0 + (ni >> 1) * 32 + (j >> 2) * 8 + (ni & 1) * 4 + (j & 3);
    const int c = ks * 4 + (lane >> 4);
    const s8v v = *reinterpret_cast<const s8v*>(s + row * 128 + ((c ^ kcb_swz(row)) << 4));
    return __builtin_bit_cast(bf16x8, v);
  } else {
    const int p = lane & 15;
    const int r = ks * 32 + (lane >> 4) * 8 + (p >> 2);
    const int col = c0 + (ni >> 1) * 32 + (p & 3) * 8 + (ni & 1) * 4;  // this lane's 4-column piece
    const unsigned char* a = s + r * 512 + ((((col >> 4) ^ ks_swz(r))) << 5) + ((col & 15) << 1);
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a + 4 * 512));
    s8v v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  }
}

// The two halves of pick_tile (below) for kernels that pick a tile ONCE and hand the record (problem, origin) on: which problem
// a linear id falls into (no memory access), and the tile origin inside it.
static __device__ __forceinline__ void pick_problem(const GroupArgs& ga, int id, int total, int& pi, int& wg) {
  const int xcd = id & 7;
  const int q8 = total >> 3, r8 = total & 7;
  wg = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  pi = 0;
#pragma unroll
  for (int i = 1; i < G2_MAXP; ++i)
    if (wg >= ga.tile_begin[i]) pi = i;
}
static __device__ __forceinline__ const GemmProblem* problem_ptr(int pi) {
  return &((const GroupArgs*)__builtin_amdgcn_kernarg_segment_ptr())->p[pi];
}
// a wave-uniform pointer that the compiler computed with vector instructions (64-bit multiply-add), back in scalar registers:
// the "s" operands of the LDS-DMA inline asm are not legalised by hipcc
static __device__ __forceinline__ const bf16_t* uniform_ptr(const bf16_t* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const bf16_t*)(((unsigned long long)hi << 32) | lo);
}
static __device__ __forceinline__ void tile_origin(int tile, int M, int N, int& m0, int& n0) {
  const int tiles_n = N / T2, tiles_m = M / T2;
  const int group = 8 * tiles_n;
  const int first_m = (tile / group) * 8;
  const int gm = min(tiles_m - first_m, 8);
  const int r = tile % group;
  m0 = (first_m + r % gm) * T2;
  n0 = (r / gm) * T2;
}

// linear id -> (problem, tile origin).  XCD-aware bijective remap (block b runs on XCD b % 8; persistent ids keep
// id % 8): each XCD's private L2 sees a contiguous run of tiles, n fastest, so neighbours share the A panel.
// The problem is picked with static indices only (a runtime-indexed kernarg array would go to scratch).
template <int TM = 256>
static __device__ __forceinline__ void pick_tile(const GroupArgs& ga, int id, int total, GemmProblem& g, int& m0, int& n0) {
  const int xcd = id & 7;
  const int q8 = total >> 3, r8 = total & 7;
  const int wg = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < G2_MAXP; ++i)
    if (wg >= ga.tile_begin[i]) pi = i;  // unused slots hold INT_MAX
  // the descriptor itself is read from the kernarg segment with a RUNTIME index (scalar loads); selecting it
  // from by-value kernargs makes hipcc preload all 16 descriptors into ~500 SGPRs and spill them
#if defined(__HIP_DEVICE_COMPILE__)
  const GroupArgs* kp = (const GroupArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  g = kp->p[pi];
#else
  g = ga.p[pi];
#endif
  // Grouped (8 tile-rows at a time, column-major inside the group) ordering: the ~32 tiles an XCD works on concurrently
  // then form an 8 x 4 patch that shares 8 A panels and 4 B panels through that XCD's L2, instead of a 1 x 32 strip that
  // shares one A panel and streams 32 different B panels from MALL/HBM (measured: the strip order is fabric-bound).
  const int tile = wg - g.tile_begin;
  const int tiles_n = g.N / T2, tiles_m = g.M / TM;
  const int group = 8 * tiles_n;
  const int first_m = (tile / group) * 8;
  const int gm = min(tiles_m - first_m, 8);
  const int r = tile % group;
  m0 = (first_m + r % gm) * TM;
  n0 = (r / gm) * T2;
}

// two consecutive 1-KiB LDS-DMA pieces under one M0 write (v1 carries the - 1 KiB of its immediate offset, see glds16x4)
template <int J0>
static __device__ __forceinline__ void glds16_pair(const void* sbase, unsigned v0, unsigned v1, unsigned dst) {
  asm volatile(
      "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %2 offset:%4\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:%5"
      :
      : "v"(v0), "v"(v1), "s"(sbase), "s"(dst), "n"(J0 * 1024), "n"(J0 * 1024 + 1024)
      : "memory", "m0");
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: the ring main loop (gemm256f_kernel below).  LDS = 3 A slots + 2 B slots of 32 KiB (one K = 64 operand tile each).
#define PP_B_BASE (3 * TILE2_BYTES)
#define PP_LDS_BYTES (5 * TILE2_BYTES)

static __device__ __forceinline__ void pp_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
}

// the per-lane source offsets of a wave's four 1-KiB pieces of an operand tile (stage256's arithmetic, kept apart from the issue;
// piece j's offset already carries the - j KiB of its instruction's immediate offset, see glds16x4)
template <bool KS, bool ISB>
static __device__ __forceinline__ void stage_voff(int ld, int wid, int lane, unsigned (&voff)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wid * 4 + j;
    if (!KS) {
      const int row = q * 8 + (lane >> 3);
      const int pos = lane & 7;
      voff[j] = (unsigned)(row * ld + ((pos ^ (ISB ? kcb_swz(row) : kc_swz(row))) << 3)) * 2u - (unsigned)j * 1024u;
    } else {
      const int kr = q * 2 + (lane >> 5);
      const int pos = lane & 31;
      voff[j] = (unsigned)(kr * ld + ((pos ^ (ks_swz(kr) << 1)) << 3)) * 2u - (unsigned)j * 1024u;
    }
  }
}
// piece J of the four, with its own M0 write (two scalar instructions: free between two MFMAs)
template <int J>
static __device__ __forceinline__ void glds16_piece(const void* sbase, unsigned voff, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
               :
               : "v"(voff), "s"(sbase), "s"(dst), "n"(J * 1024)
               : "memory", "m0");
}
// all four pieces under one M0 write (prologue)
static __device__ __forceinline__ void glds16_quad(const void* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned dst) {
  asm volatile(
      "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %4\n\t"
      "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
      "global_load_lds_dwordx4 %2, %4 offset:2048\n\t"
      "global_load_lds_dwordx4 %3, %4 offset:3072"
      :
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(dst)
      : "memory", "m0");
}

// gemm128x.hip: 128 x 256 tiles, epilogue of tile i-1 under the K loop of tile i (K = 1024 problems only); returns 1 when the
// launch is not one of its specialisations (the caller then takes the 256-row path), 0 / -hipError otherwise
int kbner_launch128x(int layout, const GroupArgs& ga, hipStream_t stream);
bool kbner_can128x(int layout, int M, int N, int K, int epi);
// gemm128s.hip: the same tiles with wave-specialised epilogues (4 MFMA waves hand the tile to 4 epilogue waves through LDS)
int kbner_launch128s(int layout, const GroupArgs& ga, hipStream_t stream);
