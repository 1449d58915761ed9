// bf16 MFMA GEMM, second structure: 256 x 256 x 64 block tile, 512 threads = 8 wavefronts (2 x 4),
// each wave 128 x 64 (8 x 4 MFMA 16x16x32 fragments, 128 accumulator registers), one workgroup per CU,
// two 64 KiB LDS stages filled by global_load_lds.  Versus the 128^2 kernel (gemm.hip) this halves
// the L2->LDS bytes per flop (the 128^2 tile needs ~34 TB/s of L2 at MFMA peak, the whole chip's L2
// bandwidth) and the LDS->register bytes per flop.  Grouped: one launch can carry up to 8 independent
// problems (same layout) so the four weight-gradient GEMMs of a layer fill the chip WITHOUT split-K:
// their fp32 accumulate-into-gradient epilogue is a plain 16-byte read-modify-write, no atomics.
//
// Operand images and fragment maps are those of gemm.hip; the k-strided (KS) image here carries a
// 32-byte-block XOR swizzle so ds_read_b64_tr_b16 is bank-conflict free (8 k-rows x 32 B per half-wave
// land on 64 distinct banks).
#include "gemm_tile.h"

// Epilogue of one 256x256 tile, instantiated per flag set (EPI_CT; -1 = generic runtime flags for uncommon
// combinations).  With runtime flags every (row-fragment, column-half) step is a chain of ~10 scalar branches and its
// own basic block: ~160 branches per tile cost more than the arithmetic they guard (3.6 us of a 32-us tile measured
// with s_memtime stamps), and the residual / GELU' operand loads cannot be hoisted out of their block, so each one
// exposes a full memory latency.  Here the flags fold at compile time, the operand loads of row fragment mi+1 are
// issued before fragment mi is processed, and sched_barriers keep the compiler from interleaving all eight fragments
// (which spills).
// IN_DMA (the ring kernels, which have a second wave-private 4 KiB of LDS -- scr2 -- at tile ends): the residual / GELU' operand
// tile reaches its lanes through LDS instead of 16-rows-x-64-B global loads in the MFMA layout, which the memory pipe retires at
// a third of the full-line rate (tools/micro/store_pattern.hip; per-tile trace, round 4: bias 3.7 K cycles, bias + residual
// 13.9 K).  Per 16-row fragment two LDS-DMA pieces fetch 16 x 128 B as full lines into one of two 2-KiB buffers, one fragment
// ahead (the DMA needs no registers), chunk-swizzled like the operand tiles so that the MFMA-layout ds_read_b128 is conflict-free.
template <int EPI_CT, int MI = 8, bool DRAIN = false, bool IN_DMA = false>
static __device__ __forceinline__ void epilogue256(const GemmProblem& g, f4v (&acc)[MI][4], int m0, int n0, int wm, int wn,
                                                   int lane, unsigned char* scr, unsigned char* scr2 = nullptr) {
  const int epi = EPI_CT >= 0 ? EPI_CT : g.epi;
  const float alpha = g.alpha;
  const int gq = lane >> 4;
  const int ncol = n0 + wn * 64 + gq * 8;  // + q * 32: the 8 contiguous columns this lane owns in column-half q
  const int mrow = m0 + wm * (MI * 16) + (lane & 15);
  float csum[2][8];
  float bq[2][8];  // this lane's 16 bias values, loaded once per tile
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      csum[q][r] = 0.0f;
      bq[q][r] = 0.0f;
    }
  const bool drop = (epi & EPI_DROP) != 0;
  const float dscale = drop_scale(g.drop_thresh);
  if (drop) {
    // the 16 column keys of this lane live in csum's registers (EPI_DROP excludes EPI_COLSUM): no extra VGPRs
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 8; ++r) csum[q][r] = __uint_as_float(drop_colkey(g.drop_seed, (uint32_t)(ncol + q * 32 + r)));
  }
  if (epi & EPI_BIAS) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float* bp = g.bias + ncol + q * 32;
      const float4 b0 = *reinterpret_cast<const float4*>(bp);
      const float4 b1 = *reinterpret_cast<const float4*>(bp + 4);
      bq[q][0] = b0.x; bq[q][1] = b0.y; bq[q][2] = b0.z; bq[q][3] = b0.w;
      bq[q][4] = b1.x; bq[q][5] = b1.y; bq[q][6] = b1.z; bq[q][7] = b1.w;
    }
  }
  // bf16 outputs leave through a 2-KiB per-wave LDS transpose (two buffers: C and out2): in the MFMA layout a store
  // instruction covers 16 rows x 64 B, which the memory pipe retires at ~34 GB/s per CU; re-read as 8 rows x 128 B (8
  // consecutive lanes per cache line) it retires at ~98 GB/s (tools/micro/store_pattern.hip)
  // (specialised forward epilogues only: on the dgrad layout the direct stores measured marginally better in situ)
  const bool lds_out = EPI_CT >= 0 && !(EPI_CT & (EPI_RMW32 | EPI_ATOMIC32 | EPI_STORE32));
  const int wr_row = lane & 15;
  const int wr_off[2] = {wr_row * 128 + (((0 * 4 + gq) ^ (wr_row & 7)) << 4), wr_row * 128 + (((1 * 4 + gq) ^ (wr_row & 7)) << 4)};
  const int rd_row = lane >> 3, rd_chunk = lane & 7;
  const int rd_lds = rd_row * 128 + ((rd_chunk ^ (rd_row & 7)) << 4);
  const size_t rd_off = (size_t)(m0 + wm * (MI * 16) + rd_row) * 1;  // row index; scaled by the leading dimension at the store
  // operand tile (residual addend or saved pre-activation) of the row fragment about to be processed, one fragment ahead
  const bool has_in = (epi & (EPI_ADD | EPI_DGELU)) != 0;
  const bf16_t* inp = (epi & EPI_ADD) ? g.addend : g.aux;
  const int ldin = (epi & EPI_ADD) ? g.ldadd : g.ldaux;
  uint4 nxt_in[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  // IN_DMA state: the per-lane source offsets of a fragment's two pieces (row (lane >> 3) [+ 8], 16-byte chunk (lane & 7) ^
  // swizzle), the running source pointer of the wave's 64-column strip, the lane's two read addresses in a buffer
  unsigned dv0 = 0, dv1 = 0, drd0 = 0, drd1 = 0, dbuf = 0, dbuf2 = 0;
  const bf16_t* dsrc = nullptr;
  size_t dstep = 0;
  // Four 2-KiB buffers (scr: 0, 1; scr2: 2, 3), fragment mi in buffer mi & 3: its operand rows land there, are read into
  // registers, and the same buffer then stages the fragment's output transpose; fragment mi + 4's pieces are issued into it at the
  // end of iteration mi, i.e. three iterations (~1600 cycles: an HBM round trip) before they are needed.
  if (IN_DMA && has_in) {
    static_assert(!IN_DMA || MI == 8, "the counted DMA waits below are written for 8 fragments");
    const int drow = lane >> 3;
    dv0 = (unsigned)(drow * ldin + (((lane & 7) ^ kc_swz(drow)) << 3)) * 2u;
    dv1 = (unsigned)((drow + 8) * ldin + (((lane & 7) ^ kc_swz(drow + 8)) << 3)) * 2u - 1024u;
    dsrc = uniform_ptr(inp + (size_t)(m0 + wm * (MI * 16)) * ldin + n0 + wn * 64);
    dstep = (size_t)16 * ldin;
    dbuf = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)scr);
    dbuf2 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)scr2);
    const int r = lane & 15;
    drd0 = (unsigned)(r * 128 + (((0 * 4 + gq) ^ kc_swz(r)) << 4));
    drd1 = (unsigned)(r * 128 + (((1 * 4 + gq) ^ kc_swz(r)) << 4));
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      glds16_pair<0>(dsrc, dv0, dv1, ((f & 2) ? dbuf2 : dbuf) + (unsigned)(f & 1) * 2048u);
      dsrc += dstep;
    }
  } else if (has_in) {
#pragma unroll
    for (int q = 0; q < 2; ++q) nxt_in[q] = *reinterpret_cast<const uint4*>(inp + (size_t)mrow * ldin + ncol + q * 32);
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mrow + mi * 16;
    const uint32_t rk = drop ? drop_rowkey(g.drop_seed, (uint32_t)m) : 0u;
    uint4 cur_in[2] = {nxt_in[0], nxt_in[1]};
    // (re-defined per fragment behind an opaque asm: hipcc otherwise hoists the alpha multiplies of ALL fragments to the top of the
    // epilogue and keeps scaled and unscaled accumulators alive -- the GELU' x + column-sum variant spilled 270 bytes that way)
    float alpha_f = alpha;
    asm volatile("" : "+v"(alpha_f));
    unsigned char* fbuf = (IN_DMA && has_in) ? ((mi & 2) ? scr2 : scr) + (mi & 1) * 2048 : scr;   // this fragment's buffer (IN_DMA)
    unsigned char* stage = fbuf;   // ... which also stages the output transpose
    if (IN_DMA && has_in) {
      // pieces / stores queued behind fragment mi's two pieces (in order): fragments mi+1.. of the first four, then per finished
      // iteration its two stores and the two pieces it issued
      constexpr int kYounger[8] = {6, 8, 10, 12, 12, 10, 8, 6};
      if (kYounger[mi] == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (kYounger[mi] == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (kYounger[mi] == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      cur_in[0] = *reinterpret_cast<const uint4*>(fbuf + drd0);
      cur_in[1] = *reinterpret_cast<const uint4*>(fbuf + drd1);
    } else if (has_in && mi + 1 < MI) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
        nxt_in[q] = *reinterpret_cast<const uint4*>(inp + (size_t)(m + 16) * ldin + ncol + q * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = ncol + q * 32;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[mi][2 * q][r] * alpha_f;
        v[4 + r] = acc[mi][2 * q + 1][r] * alpha_f;
      }
      if (epi & EPI_STORE32) {
        float4* c = reinterpret_cast<float4*>(g.C32 + (size_t)m * g.ldc32 + n);
        c[0] = make_float4(v[0], v[1], v[2], v[3]);
        c[1] = make_float4(v[4], v[5], v[6], v[7]);
        continue;
      }
      if (epi & EPI_RMW32) {
        float4* c = reinterpret_cast<float4*>(g.C32 + (size_t)m * g.ldc32 + n);
        float4 o0 = c[0], o1 = c[1];
        o0.x += v[0]; o0.y += v[1]; o0.z += v[2]; o0.w += v[3];
        o1.x += v[4]; o1.y += v[5]; o1.z += v[6]; o1.w += v[7];
        c[0] = o0;
        c[1] = o1;
        continue;
      }
      if (epi & EPI_ATOMIC32) {
        float* c = g.C32 + (size_t)m * g.ldc32 + n;
#pragma unroll
        for (int r = 0; r < 8; ++r) atomicAdd(c + r, v[r]);
        continue;
      }
      if (epi & EPI_BIAS) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += bq[q][r];
      }
      if (drop) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = drop_keep(rk, __float_as_uint(csum[q][r]), g.drop_thresh) ? v[r] * dscale : 0.0f;
      }
      const uint32_t w_in[4] = {cur_in[q].x, cur_in[q].y, cur_in[q].z, cur_in[q].w};
      if (epi & EPI_ADD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[2 * r] += __uint_as_float(w_in[r] << 16);
          v[2 * r + 1] += __uint_as_float(w_in[r] & 0xffff0000u);
        }
      }
      if (epi & EPI_DGELU) {  // aux holds gelu'(pre), stored by the forward EPI_GELU epilogue
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f2v gg = unpack2bf(w_in[r]);
          v[2 * r] *= gg[0];
          v[2 * r + 1] *= gg[1];
        }
      }
      if (epi & EPI_GELU_FWD) {   // inference: the activation alone (no erf derivative, no second output)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f2v y = gelu2((f2v){v[2 * r], v[2 * r + 1]});
          v[2 * r] = y[0];
          v[2 * r + 1] = y[1];
        }
      }
      if (epi & EPI_GELU) {
        // C = gelu(pre), out2 = gelu'(pre), both evaluated at the fp32 pre-activation (rounds 1-3 rounded it to bf16 first, as if it
        // had been stored: 3 of the ~30 vector instructions per element pair of this issue-bound epilogue, and a rounding the
        // reference's fp32 path does not have)
        uint4 du;
        uint32_t dw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f2v y, dy;
          gelu_both2((f2v){v[2 * r], v[2 * r + 1]}, y, dy);
          v[2 * r] = y[0];
          v[2 * r + 1] = y[1];
          dw[r] = pack2bf(dy[0], dy[1]);
        }
        du.x = dw[0]; du.y = dw[1]; du.z = dw[2]; du.w = dw[3];
        if (lds_out) *reinterpret_cast<uint4*>(stage + wr_off[q] + 2048) = du;
        else *reinterpret_cast<uint4*>(g.out2 + (size_t)m * g.ldout2 + n) = du;
      }
      uint4 o;
      o.x = pack2bf(v[0], v[1]);
      o.y = pack2bf(v[2], v[3]);
      o.z = pack2bf(v[4], v[5]);
      o.w = pack2bf(v[6], v[7]);
      if (lds_out) *reinterpret_cast<uint4*>(stage + wr_off[q]) = o;
      else *reinterpret_cast<uint4*>(g.C + (size_t)m * g.ldc + n) = o;
      if (epi & EPI_COLSUM) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma clang fp contract(off)   // no FMA contraction with the multiply that produced v: the same sums in every variant
          csum[q][r] = csum[q][r] + v[r];
        }
      }
    }
    if (lds_out) {
      const uint4 h0 = *reinterpret_cast<const uint4*>(stage + rd_lds);
      const uint4 h1 = *reinterpret_cast<const uint4*>(stage + rd_lds + 8 * 128);
      bf16_t* d0 = g.C + (rd_off + mi * 16) * g.ldc + n0 + wn * 64 + rd_chunk * 8;
      *reinterpret_cast<uint4*>(d0) = h0;
      *reinterpret_cast<uint4*>(d0 + (size_t)8 * g.ldc) = h1;
      if (epi & EPI_GELU) {
        const uint4 e0 = *reinterpret_cast<const uint4*>(stage + rd_lds + 2048);
        const uint4 e1 = *reinterpret_cast<const uint4*>(stage + rd_lds + 2048 + 8 * 128);
        bf16_t* d2 = g.out2 + (rd_off + mi * 16) * g.ldout2 + n0 + wn * 64 + rd_chunk * 8;
        *reinterpret_cast<uint4*>(d2) = e0;
        *reinterpret_cast<uint4*>(d2 + (size_t)8 * g.ldout2) = e1;
      }
    }
    if (IN_DMA && has_in && mi + 4 < MI) {
      // this fragment's buffer is done with (operand rows read, output rows staged and read back: the stores above carry them):
      // fragment mi + 4's operand rows go there
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      glds16_pair<0>(dsrc, dv0, dv1, ((mi & 2) ? dbuf2 : dbuf) + (unsigned)(mi & 1) * 2048u);
      dsrc += dstep;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (epi & EPI_COLSUM) {
    // this wave's 128 rows: 8 in registers (mi), 16 across the lanes of a group (xor-shuffle); then either one atomic per
    // column, or -- EPI_COLSUM_WS -- two 16-byte stores into this (tile row, wave row)'s line of a workspace that a reduce
    // kernel folds afterwards.  The atomics were the epilogue: 512 device-scope fp32 atomics per tile onto 4096 addresses that
    // every CU hits made the FFN-down dgrad (GELU' + bias gradient) the slowest GEMM of the step, 752 TFLOP/s against 1040-1250
    // for the other dgrad shapes.
    const bool to_ws = (epi & EPI_COLSUM_WS) != 0;
    float* wrow = g.colsum + (to_ws ? (size_t)((m0 / (MI * 32)) * 2 + wm) * g.N : 0);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float red[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float x = csum[q][r];
        x += __shfl_xor(x, 1, 64);
        x += __shfl_xor(x, 2, 64);
        x += __shfl_xor(x, 4, 64);
        x += __shfl_xor(x, 8, 64);
        red[r] = x;
      }
      if ((lane & 15) == 0) {
        if (to_ws) {
          float4* d = reinterpret_cast<float4*>(wrow + ncol + q * 32);
          d[0] = make_float4(red[0], red[1], red[2], red[3]);
          d[1] = make_float4(red[4], red[5], red[6], red[7]);
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) atomicAdd(g.colsum + ncol + q * 32 + r, red[r]);
        }
      }
    }
  }
  // The generic variant issues its operand / bias loads and consumes them under runtime flags: hipcc's waitcnt pass (path-
  // insensitive) must assume a path on which a load is issued and never consumed, carries that "pending" load into the next
  // tile's main loop and drains vmcnt(0) in front of the first ds_read that reuses its destination register -- i.e. it waits for
  // the LDS-DMA in flight there (once per tile in gemm256_kernel's peeled first K step, in EVERY K step of the ping-pong loop).
  // A compiler-visible vmcnt(0) here (DRAIN: the ping-pong kernel's generic variant; it also drains this tile's stores) keeps the state clean;
  // the specialised variants consume every load on straight-line code.
  if constexpr (EPI_CT < 0 && DRAIN) __builtin_amdgcn_s_waitcnt(0x0F70);
}

// -DG2_TRACE (tools/gemm_trace.sh, never the product build): wave 0 stamps s_memtime at tile start / main-loop end /
// epilogue end so tools/gemm_trace.py can split a persistent workgroup's time per tile.
#ifdef G2_TRACE
// effective shader clock of a launch: s_memtime (shader cycles) against s_memrealtime (constant 100 MHz) at workgroup start / end
__device__ unsigned long long g2_clk[256 * 4];
#define G2_CLK(slot)                                                                   \
  if (threadIdx.x == 0 && blockIdx.x < 256) {                                          \
    g2_clk[blockIdx.x * 4 + (slot) * 2] = __builtin_readcyclecounter();                \
    g2_clk[blockIdx.x * 4 + (slot) * 2 + 1] = __builtin_amdgcn_s_memrealtime();        \
  }
__device__ unsigned long long g2_trace[256 * 32 * 4];
#define G2_T(slot)                                                                        \
  if (threadIdx.x == 0 && tile_no < 32) g2_trace[(blockIdx.x * 32 + tile_no) * 4 + (slot)] = __builtin_readcyclecounter();
#else
#define G2_T(slot)
#define G2_CLK(slot)
#endif

// Dynamic tile scheduling (DYN = true; data-parallel training, where RCCL's kernels take CUs away while a bucket is in flight):
// a workgroup DRAWS its tiles from per-XCD counters instead of walking id, id + grid, ...  A workgroup that the dispatcher could
// only place late (its CU was busy with a collective) then simply finds fewer tiles left; with the static walk it would run its
// full share after everybody else had finished and double the launch time.  XCD locality is kept: the workgroup on XCD x draws
// from x's share {x, x+8, ...} first (same tile->XCD map as the static walk) and steals from the other shares when x's is empty.
// Returns an id >= total when nothing is left.  Called by thread 0 only; the id travels to the other waves through LDS.
static __device__ __forceinline__ int draw_tile(int* sched, int xcd, int total) {
#pragma unroll 1
  for (int s = 0; s < 8; ++s) {
    const int y = (xcd + s) & 7;
    const int share = (total - y + 7) >> 3;
    if (share <= 0) continue;
    const int n = __hip_atomic_fetch_add(sched + y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n < share) return n * 8 + y;
  }
  return total;
}

// TM = 128: half-height tiles (128 x 256, each wave 64 x 64) for forward / dgrad problems whose 256 x 256 tiling would leave half
// of the CUs without a tile (small micro-batches: M = 2048 tokens x N = 4096 is 128 big tiles on 256 CUs).  Same B tile, LDS
// images, K loop and epilogues; the A tile is 16 KiB, a K step 4 MFMA groups instead of 8.
template <bool A_KS, bool B_KS, bool DYN = false, int TM = 256>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GroupArgs ga) {
  constexpr int MI = TM / 32;   // 16-row accumulator fragments per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  // DYN: the drawn tile id is handed to the other waves through the last word of wave 0's epilogue scratch -- written by
  // thread 0 only between its own epilogue and the next tile's first barrier, read by everybody after that barrier
  volatile int* s_next = reinterpret_cast<volatile int*>(smem + 2 * STAGE2_BYTES + 4092);

  // Persistent: this workgroup walks tiles id, id + gridDim.x, ... (grid = min(tiles, #CUs)).  The first K tile
  // of the NEXT output tile is DMA'd during the last K iteration of the current one, so the epilogue's loads and
  // stores run under that flight and the next main loop starts without a cold prologue.
  const int total = ga.total_tiles;
  int id = blockIdx.x;
  G2_CLK(0)
  if (DYN) {
    if (tid == 0) *s_next = draw_tile(ga.sched, blockIdx.x & 7, total);
    __syncthreads();
    id = __builtin_amdgcn_readfirstlane(*s_next);
    __syncthreads();
    if (id >= total) return;   // a late workgroup: every tile has been taken
  }
  int it = 0;    // running K-iteration counter: LDS stage parity
  int pend = 0;  // epilogue stores issued after the DMA that is in flight at a tile boundary
  {
    GemmProblem g0;
    int m00, n00;
    pick_tile<TM>(ga, id, total, g0, m00, n00);
    stage256<A_KS, false, TM>(g0.A, g0.lda, m00, 0, smem, wid, lane);
    stage256<B_KS, true>(g0.B, g0.ldb, n00, 0, smem + TILE2_BYTES, wid, lane);
  }

  int tile_no = 0;
  (void)tile_no;
  for (;;) {
  G2_T(0)
  // (the problem descriptor is re-read from kernarg where it is needed -- main loop, next-tile prefetch, epilogue --
  // instead of being carried in ~30 SGPRs across the MFMA loop, which spilled)
  int m0, n0, nt, lda, ldb;
  const bf16_t* __restrict__ Ap;
  const bf16_t* __restrict__ Bp;
  {
    GemmProblem gm;
    pick_tile<TM>(ga, id, total, gm, m0, n0);
    nt = gm.K / BK2;
    Ap = gm.A;
    Bp = gm.B;
    lda = gm.lda;
    ldb = gm.ldb;
  }
  int id_next = id + (int)gridDim.x;
  bool has_next = id_next < total;
  // DYN: the next tile is drawn one tile ahead; thread 0 waits for its atomic here, before the tile's first barrier, and all
  // waves read the id after that barrier.  (Measured on MI355X, whole training step: static walk 880 sentences/s, this 863;
  // issuing the atomic here but consuming it two K steps later -- to hide its round trip under the DMA wait -- ran at 739: a
  // compiler-visible VMEM result in flight across the K loop makes hipcc drain vmcnt(0) around the hand-counted waits.)
  if (DYN && tid == 0) *s_next = draw_tile(ga.sched, blockIdx.x & 7, total);

  f4v acc[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f4v){0.0f, 0.0f, 0.0f, 0.0f};

  for (int t = 0; t < nt; ++t) {
    // the DMA of this K tile has landed: every wave drains its own pieces (counted: the previous tile's epilogue
    // stores, younger than that DMA, may still be in flight), then the barrier publishes all of them and fences
    // the previous iteration's reads of the stage about to be refilled.  Raw s_barrier: __syncthreads() would add a
    // vmcnt(0) release for the epilogue's global stores.
    if (t == 0 && pend == 8) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (t == 0 && pend == 16) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else if (t == 0 && pend == 32) {
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (DYN && t == 0) {
      if (nt == 1) __builtin_amdgcn_s_barrier();   // (K = 64: the draw was written after this tile's only barrier was passed)
      id_next = __builtin_amdgcn_readfirstlane(*s_next);
      has_next = id_next < total;
      // K = 64: no later barrier of this tile orders the read before wave 0's epilogue, whose scratch holds the word
      if (nt == 1) __builtin_amdgcn_s_barrier();
    }
    unsigned char* cur = smem + (it & 1) * STAGE2_BYTES;
    unsigned char* nxt = smem + ((it + 1) & 1) * STAGE2_BYTES;
    if (t + 1 < nt) {
      stage256<A_KS, false, TM>(Ap, lda, m0, (t + 1) * BK2, nxt, wid, lane);
      stage256<B_KS, true>(Bp, ldb, n0, (t + 1) * BK2, nxt + TILE2_BYTES, wid, lane);
    } else if (has_next) {
      GemmProblem gn;
      int m0n, n0n;
      pick_tile<TM>(ga, id_next, total, gn, m0n, n0n);
      stage256<A_KS, false, TM>(gn.A, gn.lda, m0n, 0, nxt, wid, lane);
      stage256<B_KS, true>(gn.B, gn.ldb, n0n, 0, nxt + TILE2_BYTES, wid, lane);
    }
    ++it;
    // Software-pipelined fragment stream: the ds_reads of step s+1 are issued BEFORE the 8 MFMAs of step s
    // (8 steps per K tile = 2 k-steps x 4 row pairs), pinned with sched_barrier so the compiler's counted
    // lgkmcnt waits only for the fragments the current MFMAs consume.
#define G2_SB() __builtin_amdgcn_sched_barrier(0)
#define G2_LOADB(dst, ks)                                                                                     \
  _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) dst[ni] = fragB256<B_KS>(cur + TILE2_BYTES, wn * 64, ni, ks, lane)
#define G2_LOADA(dst, ks, pr)                                         \
  dst[0] = frag256<A_KS>(cur, wm * (TM / 2) + (2 * (pr)) * 16, ks, lane);  \
  dst[1] = frag256<A_KS>(cur, wm * (TM / 2) + (2 * (pr) + 1) * 16, ks, lane)
#define G2_MM(a, b, pr)                                                                                              \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) acc[2 * (pr) + j][ni] = \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[j], acc[2 * (pr) + j][ni], 0, 0, 0)
    bf16x8 b0[4], b1[4], a0[2], a1[2];
    G2_LOADB(b0, 0);
    G2_LOADA(a0, 0, 0);
    G2_SB();
    if constexpr (TM == 256) {
      G2_LOADA(a1, 0, 1); G2_SB(); G2_MM(a0, b0, 0); G2_SB();
      G2_LOADA(a0, 0, 2); G2_SB(); G2_MM(a1, b0, 1); G2_SB();
      G2_LOADA(a1, 0, 3); G2_SB(); G2_MM(a0, b0, 2); G2_SB();
      G2_LOADB(b1, 1);
      G2_LOADA(a0, 1, 0); G2_SB(); G2_MM(a1, b0, 3); G2_SB();
      G2_LOADA(a1, 1, 1); G2_SB(); G2_MM(a0, b1, 0); G2_SB();
      G2_LOADA(a0, 1, 2); G2_SB(); G2_MM(a1, b1, 1); G2_SB();
      G2_LOADA(a1, 1, 3); G2_SB(); G2_MM(a0, b1, 2); G2_SB();
      G2_MM(a1, b1, 3);
    } else {
      G2_LOADA(a1, 0, 1); G2_SB(); G2_MM(a0, b0, 0); G2_SB();
      G2_LOADB(b1, 1);
      G2_LOADA(a0, 1, 0); G2_SB(); G2_MM(a1, b0, 1); G2_SB();
      G2_LOADA(a1, 1, 1); G2_SB(); G2_MM(a0, b1, 0); G2_SB();
      G2_MM(a1, b1, 1);
    }
  }

  G2_T(1)
  GemmProblem g;
  {
    int mm, nn;
    pick_tile<TM>(ga, id, total, g, mm, nn);
  }
  const int epi = g.epi;
  unsigned char* scr = smem + 2 * STAGE2_BYTES + wid * 4096;
  // forward (NT) tiles take the specialised epilogues; the dgrad layout (whose transpose-read B operand leaves fewer free
  // registers) only the plain / residual-add ones -- its specialised GELU' + column-sum variant spills and measured
  // slower in situ than the generic epilogue
  if (!B_KS) {
    switch (epi) {
      case 0: epilogue256<0, MI>(g, acc, m0, n0, wm, wn, lane, scr); break;
      case EPI_BIAS: epilogue256<EPI_BIAS, MI>(g, acc, m0, n0, wm, wn, lane, scr); break;
      case EPI_BIAS | EPI_GELU: epilogue256<(EPI_BIAS | EPI_GELU), MI>(g, acc, m0, n0, wm, wn, lane, scr); break;
      case EPI_BIAS | EPI_GELU_FWD: epilogue256<(EPI_BIAS | EPI_GELU_FWD), MI>(g, acc, m0, n0, wm, wn, lane, scr); break;
      case EPI_BIAS | EPI_ADD: epilogue256<(EPI_BIAS | EPI_ADD), MI>(g, acc, m0, n0, wm, wn, lane, scr); break;
      case EPI_BIAS | EPI_ADD | EPI_DROP: epilogue256<(EPI_BIAS | EPI_ADD | EPI_DROP), MI>(g, acc, m0, n0, wm, wn, lane, scr); break;
      default: epilogue256<-1, MI>(g, acc, m0, n0, wm, wn, lane, scr); break;
    }
  } else if (A_KS && epi == EPI_RMW32) {
    epilogue256<EPI_RMW32, MI>(g, acc, m0, n0, wm, wn, lane, scr);
  } else if (A_KS && epi == EPI_STORE32) {
    epilogue256<EPI_STORE32, MI>(g, acc, m0, n0, wm, wn, lane, scr);
  } else if (!A_KS && epi == EPI_ADD) {
    epilogue256<EPI_ADD, MI>(g, acc, m0, n0, wm, wn, lane, scr);
  } else if (!A_KS && epi == 0) {
    epilogue256<0, MI>(g, acc, m0, n0, wm, wn, lane, scr);
  } else {
    epilogue256<-1, MI>(g, acc, m0, n0, wm, wn, lane, scr);
  }
  G2_T(2)
  ++tile_no;
  if (!has_next) { G2_CLK(1) break; }
  // stores this wave issued after the in-flight DMA: 2 per 16-row fragment (4 with a second output / fp32 outputs)
  pend = (epi & EPI_ATOMIC32) ? 0 : ((epi & (EPI_GELU | EPI_RMW32 | EPI_STORE32)) ? 4 * MI : 2 * MI);
  id = id_next;
  }  // persistent tile loop
}



// Round 5: tile-boundary re-synchronisation of the workgroups that share an XCD's L2 (long-K launches of the ring kernel only).
// The 8 x 4 patch of tiles an XCD works on shares 8 A panels and 4 B panels through its 4-MiB L2 -- as long as the 32 workgroups
// stay within the ~10 K steps of history that L2 holds.  The two-stage loop of rounds 1-3 waited for every stage's landing, so a
// workgroup that ran ahead paid the misses and the others caught up: its miss count on the weight-gradient launch (K = 65536:
// 3 x 1024 K steps per workgroup) is the same to four digits in every launch (TCC_MISS 8.760e7 lines = 11.2 GB, tools/wgrad_pmc.sh).
// The ring loop hides the landing time (issue-bound), nothing pulls a straggler back, and the drift accumulates over a launch:
// 2.31e7 misses at K = 16384 (= the two-stage loop's), but 1.28-1.44e8 at K = 65536 (16.4-18.5 GB, +46...+65 %, different in every
// launch).  So: after each of a workgroup's tiles but the last, thread 0 counts itself in on its XCD's counter and waits (bounded)
// for the other 31; the workgroup's other waves are held by the first K step's barrier.  Monotonic counters (one round = +32),
// never reset; a timed-out wait only costs the sharing.  Grid = 256 launches only (32 workgroups per counter).
__device__ int g2_xcd_round[8 * 32];   // one counter per XCD, 128 B apart
// -> false when the round did not complete within the bound.  (Round 6 tried what ADVICE round 5 suggested -- a workgroup whose meeting
// timed out stops meeting for the rest of its walk, so that a launch whose 32 workgroups per counter are NOT co-resident does not
// spin out the bound at every tile boundary -- and measured the opposite of an improvement on the launch the meeting exists for:
// after a 1024-step tile the arrivals of an XCD's workgroups are spread over more than the bound, the early ones time out ROUTINELY,
// and with them gone from the later meetings the L2-miss traffic of the weight-gradient launch went back from 30.2 to 34.4 GB per
// launch, the unsynchronised level.  So every tile boundary meets again; the price in the non-co-resident case stays bounded at
// ~0.2 ms per boundary, and dynamic launches -- the only ones that share CUs with a collective -- never meet: pad_ = 0 there.)
static __device__ __forceinline__ bool xcd_tile_sync(int xcd) {
  int* c = g2_xcd_round + xcd * 32;
  const int v = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int target = (v | 31) + 1;
#pragma unroll 1
  for (int spin = 0; spin < 256; ++spin) {   // <= 256 x (load round trip + 64 cycles) ~ 0.3 ms at the very most; the workgroups of
                                              // an undisturbed launch arrive within ~20 us of each other
    if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target >= 0) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The INTERLEAVED ring loop (round 4, final form).  tools/micro/issue_probe.hip is the cost model behind it: next to 64 MFMAs
// per wave and two waves per SIMD in lockstep (identical code from the same barrier -- the GEMM's situation), an instruction
// that sits BETWEEN two MFMAs is free (40 scalar instructions +5 cycles, 24 fragment reads +44, 8 LDS-DMA pieces +64), while the
// same instructions in chunks between 8-MFMA groups are not (+212 / +70 / +137): both waves are in their chunks at the same time
// and the matrix pipe has nothing to issue.  The two-stage loop's shape (barrier, 8-piece DMA burst, 8 x (3 reads + scalar work,
// 8 MFMAs)) costs 2861 cycles per K step in the probe and 2950 in the kernel; the same ingredients spread out cost 2150.
// So: the ring of gemm256r_kernel (operand slots, running cursors, DMA wait in front of the pre-epilogue barrier, last group
// held across the barrier), its 8-MFMA groups and their one-group-ahead fragment prefetch -- but every fragment read, every DMA
// piece (own M0 write: two scalar instructions) and the cursor arithmetic is a statement placed after ONE MFMA, pinned with
// sched_barrier; no branch inside a step.
template <bool A_KS, bool B_KS, int ABL = 0, bool MIDSYNC = false>
__global__ __launch_bounds__(512, 2) void gemm256f_kernel(const GroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int total = ga.total_tiles;
  const int gstep = (int)gridDim.x;
  // (ga.pad_: tile-boundary sync requested by the launcher -- long K, grid of 256; see xcd_tile_sync)
  const int sync_rounds = ((ga.pad_ & 1) && gstep == 256) ? total / 256 - 1 : 0;
  // ... and, when every problem of the launch has the same K, also every (sync_mask + 1) K steps inside each of the total / 256
  // tiles that every workgroup has (ga.pad_ >> 8 = the period; 0 = tile boundaries only)
  // (MIDSYNC: its own instantiation -- the test in the K loop costs the other layouts scalar registers they do not have)
  const int sync_mask = (MIDSYNC && (ga.pad_ & 1) && gstep == 256) ? (ga.pad_ >> 8) - 1 : -1;
  const int sync_tiles = total / 256;
  G2_CLK(0)

  int lane_m = lane;   // opaque copy for the main loop's address arithmetic (see gemm256pp_kernel)
  asm volatile("" : "+v"(lane_m));
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);
  // A tile is picked ONCE, by the A cursor (the first to need it); the record (problem, origin) travels on to the B cursor and
  // to the consumer through two one-deep stashes (a cursor crosses into its next tile exactly one K step after the cursor ahead
  // of it; at the end of a step B advances before A, so B reads the stash before A overwrites it; pi < 0 = no more tiles)
  int rab_pi = -1, rab_m = 0, rab_n = 0;   // A -> B
  int rbc_pi = -1, rbc_m = 0, rbc_n = 0;   // B -> consumer
  int a_id = blockIdx.x, a_left = 0;
  const bf16_t* a_ptr = nullptr;
  size_t a_stride = 0;
  unsigned a_dst = lds0 + wid * 4096;
  unsigned va[4];
  int b_left = 0;
  const bf16_t* b_ptr = nullptr;
  size_t b_stride = 0;
  unsigned b_dst = lds0 + PP_B_BASE + wid * 4096;
  unsigned vb[4];
#define RF_ALOAD()                                                                        \
  {                                                                                       \
    int pi_, wg_, m_, n_;                                                                 \
    pick_problem(ga, a_id, total, pi_, wg_);                                              \
    const GemmProblem* gp_ = problem_ptr(pi_);                                            \
    tile_origin(wg_ - gp_->tile_begin, gp_->M, gp_->N, m_, n_);                           \
    const int lda_ = gp_->lda;                                                            \
    a_left = gp_->K / BK2;                                                                \
    a_ptr = uniform_ptr(A_KS ? gp_->A + m_ : gp_->A + (size_t)m_ * lda_);                \
    a_stride = A_KS ? (size_t)BK2 * lda_ : (size_t)BK2;                                   \
    stage_voff<A_KS, false>(lda_, wid, lane_m, va);                                       \
    rab_pi = pi_;                                                                         \
    rab_m = m_;                                                                           \
    rab_n = n_;                                                                           \
  }
#define RF_BLOAD()                                                                        \
  {                                                                                       \
    const GemmProblem* gp_ = problem_ptr(rab_pi);                                         \
    const int ldb_ = gp_->ldb;                                                            \
    b_left = gp_->K / BK2;                                                                \
    b_ptr = uniform_ptr(B_KS ? gp_->B + rab_n : gp_->B + (size_t)rab_n * ldb_);          \
    b_stride = B_KS ? (size_t)BK2 * ldb_ : (size_t)BK2;                                   \
    stage_voff<B_KS, true>(ldb_, wid, lane_m, vb);                                        \
    rbc_pi = rab_pi;                                                                      \
    rbc_m = rab_m;                                                                        \
    rbc_n = rab_n;                                                                        \
  }
  // the cheap half of a cursor step (fillers between MFMAs) and its rare tile crossing (behind the step's last MFMA)
#define RF_AADV_PTR()                                                   \
  {                                                                     \
    a_ptr += a_stride;                                                  \
    a_dst = (a_dst + TILE2_BYTES >= lds0 + PP_B_BASE) ? a_dst - 2 * TILE2_BYTES : a_dst + TILE2_BYTES; \
  }
#define RF_BADV_PTR()                                                   \
  {                                                                     \
    b_ptr += b_stride;                                                  \
    b_dst = (2 * (lds0 + PP_B_BASE + wid * 4096) + TILE2_BYTES) - b_dst; \
  }
#define RF_AADV_TILE()                                                  \
  {                                                                     \
    if (--a_left == 0) {                                                \
      a_id += gstep;                                                    \
      if (a_id < total) RF_ALOAD()                                      \
      else {                                                            \
        a_ptr -= a_stride; /* exhausted: keep re-reading the last stage (see RF_BODY) */ \
        a_stride = 0;                                                   \
        a_left = 0x7fffffff;                                            \
        rab_pi = -1;                                                    \
      }                                                                 \
    }                                                                   \
  }
#define RF_BADV_TILE()                                                  \
  {                                                                     \
    if (--b_left == 0) {                                                \
      if (rab_pi >= 0) RF_BLOAD()                                       \
      else {                                                            \
        b_ptr -= b_stride;                                              \
        b_stride = 0;                                                   \
        b_left = 0x7fffffff;                                            \
        rbc_pi = -1;                                                    \
      }                                                                 \
    }                                                                   \
  }

  RF_ALOAD();
  RF_BLOAD();
  int c_pi = rab_pi, m0 = rab_m, n0 = rab_n;   // the consumer's tile
  glds16_quad(a_ptr, va[0], va[1], va[2], va[3], a_dst);
  RF_AADV_PTR();
  RF_AADV_TILE();
  glds16_quad(b_ptr, vb[0], vb[1], vb[2], vb[3], b_dst);
  RF_BADV_PTR();
  RF_BADV_TILE();
  glds16_quad(a_ptr, va[0], va[1], va[2], va[3], a_dst);
  RF_AADV_PTR();
  RF_AADV_TILE();
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  pp_barrier();   // stage 0 of the first tile is visible

  unsigned sa_off = 0, sb_off = PP_B_BASE;
  int tile_no = 0;
  (void)tile_no;
#define RF_SB() __builtin_amdgcn_sched_barrier(0)
#define RF_NOP (void)0
  // Fragment addresses.  Row-major (KC) images: ONE address register per operand and k half for the whole step -- the row block
  // (A: mi * 2 KiB, B: (ni >> 1) * 4 KiB + (ni & 1) * 512 B) is the ds_read's immediate offset and k half 1 is k half 0 with
  // bit 6 flipped (the chunk swizzles only look at row bits the row block does not touch) -- made opaque so that hipcc keeps this
  // form (left alone it hoists eight per-row-block registers out of the loop and adds the slot offset to each: two VALU
  // instructions per read, each of which takes the matrix pipe's issue port)
  typedef const s8v __attribute__((address_space(3))) lds_s8v;
  unsigned laneA = 0, laneB = 0, pa0 = 0, pa1 = 0, pb0 = 0, pb1 = 0;
  (void)pa1;
#define RF_LANE_BASES()                                                                                          \
  {                                                                                                              \
    if (!A_KS) {                                                                                                 \
      const int row = wm * 128 + (lane_m & 15);                                                                  \
      laneA = lds0 + row * 128 + ((((lane_m >> 4)) ^ kc_swz(row)) << 4);                                         \
    } else { /* k-strided image: k row r, 32-byte block (row block ^ swizzle): the row block mi enters by XOR, see fa_ */ \
      const int p = lane_m & 15;                                                                                 \
      const int r = (lane_m >> 4) * 8 + (p >> 2);                                                                \
      laneA = lds0 + r * 512 + ((p & 3) << 3) + wm * 256 + (ks_swz(r) << 5);                                     \
    }                                                                                                            \
    if (!B_KS) {                                                                                                 \
      const int j = lane_m & 15;                                                                                 \
      const int row = wn * 64 + (j >> 2) * 8 + (j & 3);                                                          \
      laneB = lds0 + row * 128 + ((((lane_m >> 4)) ^ kcb_swz(row)) << 4);                                        \
    } else {                                                                                                     \
      const int p = lane_m & 15;                                                                                 \
      const int r = (lane_m >> 4) * 8 + (p >> 2);                                                                \
      laneB = lds0 + r * 512 + (((wn * 4 + ((p & 3) >> 1)) ^ ks_swz(r)) << 5) + (((p & 3) & 1) << 4);            \
    }                                                                                                            \
  }
#define RF_STEP_BASES()                                          \
  {                                                              \
    pa0 = laneA + sa_off;                                        \
    asm volatile("" : "+v"(pa0));                                \
    if (!A_KS) {                                                 \
      pa1 = pa0 ^ 64u;                                           \
      asm volatile("" : "+v"(pa1));                              \
    }                                                            \
    pb0 = laneB + sb_off;                                        \
    asm volatile("" : "+v"(pb0));                                \
    pb1 = pb0 ^ 64u;                                             \
    asm volatile("" : "+v"(pb1));                                \
  }
  typedef s4v __attribute__((address_space(3))) lds_s4v;
  auto tr2_ = [&](unsigned addr) -> bf16x8 {   // the two transposed 8-byte reads of a k-strided fragment (k rows r and r + 4)
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(size_t)addr);
#if defined(RF_EXP_HALFTR)     /* timing only (wrong operands): one transposed read per fragment instead of two */
    const s4v hi = lo;
#else
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(size_t)(addr + 4u * 512u));
#endif
    s8v v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  };
  auto fa_ = [&](int ks, int mi) -> bf16x8 {
    if constexpr (!A_KS) {
      const s8v v = *reinterpret_cast<lds_s8v*>((size_t)((ks ? pa1 : pa0) + (unsigned)mi * 2048u));
      return __builtin_bit_cast(bf16x8, v);
    } else {
      // (block ^ swizzle) << 5 with block = 8 wm + mi: the low three bits of the block are mi, so the row block is an XOR on
      // address bits 5-7 (one VALU instruction per fragment; the k half is an immediate: 32 k rows x 512 B)
#if defined(RF_EXP_NOXOR)      /* timing only (wrong operands): what do the 16 address XORs of a TN K step cost? */
      return tr2_(pa0 + (unsigned)mi * 32u + (unsigned)ks * 16384u);
#else
      return tr2_((pa0 ^ ((unsigned)mi << 5)) + (unsigned)ks * 16384u);
#endif
    }
  };
  auto fb_ = [&](int ks, int ni) -> bf16x8 {
    if constexpr (!B_KS) {
      const s8v v = *reinterpret_cast<lds_s8v*>((size_t)((ks ? pb1 : pb0) + (unsigned)((ni >> 1) * 4096 + (ni & 1) * 512)));
      return __builtin_bit_cast(bf16x8, v);
    } else {
      // column block (4 wn + 2 (ni >> 1) + ...) ^ swizzle: ni >> 1 flips address bit 6 (pb1), ni & 1 adds 8 bytes
      return tr2_(((ni >> 1) ? pb1 : pb0) + (unsigned)ks * 16384u + (unsigned)(ni & 1) * 8u);
    }
  };
#define RF_FA(ks, mi) fa_(ks, mi)
#define RF_FB(ks, ni) fb_(ks, ni)
  // (Z: the MFMA starts its accumulator -- the k half 0 groups of a tile's first step; saves zeroing 128 registers per tile)
#define RF_M(a, b, pr, j, ni, Z)                                                                                        \
  if (!(ABL & 4)) acc[2 * (pr) + (j)][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                    \
      b[ni], a[j], (Z) ? (f4v){0.0f, 0.0f, 0.0f, 0.0f} : acc[2 * (pr) + (j)][ni], 0, 0, 0)
  // one 8-MFMA group (A fragments a[0..1] x B fragments b[0..3] -> accumulator rows 2 pr, 2 pr + 1), one filler statement per MFMA
#define RF_GROUPZ(Z, a, b, pr, f0, f1, f2, f3, f4, f5, f6, f7)             \
  RF_M(a, b, pr, 0, 0, Z); RF_SB(); f0; RF_SB();                              \
  RF_M(a, b, pr, 0, 1, Z); RF_SB(); f1; RF_SB();                              \
  RF_M(a, b, pr, 0, 2, Z); RF_SB(); f2; RF_SB();                              \
  RF_M(a, b, pr, 0, 3, Z); RF_SB(); f3; RF_SB();                              \
  RF_M(a, b, pr, 1, 0, Z); RF_SB(); f4; RF_SB();                              \
  RF_M(a, b, pr, 1, 1, Z); RF_SB(); f5; RF_SB();                              \
  RF_M(a, b, pr, 1, 2, Z); RF_SB(); f6; RF_SB();                              \
  RF_M(a, b, pr, 1, 3, Z); RF_SB(); f7; RF_SB();
#define RF_GROUP(a, b, pr, f0, f1, f2, f3, f4, f5, f6, f7) RF_GROUPZ(false, a, b, pr, f0, f1, f2, f3, f4, f5, f6, f7)
#define RF_PB(J) if (!(ABL & (8 | 16))) glds16_piece<J>(b_ptr, vb[J], b_dst)
// RF_PBF: B(t+1)'s pieces in a tile's FIRST step (no held group in front of it); in the other steps they ride on the held group
#define RF_PBF(FIRST, J) if (FIRST) RF_PB(J)
#define RF_PA(J) if (!(ABL & (8 | 32))) glds16_piece<J>(a_ptr, va[J], a_dst)
  // groups 0..6 of a K step (group 7 is held across the barrier), the DMA of the step -- B(t+1), which has to land within this
  // step, right behind the barrier (on the held group; in a tile's first step at the head of group 0), A(t+2) in groups 0-3,
  // one piece behind an MFMA --, the cursor steps, the DMA wait, the barrier, the slot rotation.  No branch: a cursor that
  // has run out of tiles (the last two K steps of a workgroup's walk) keeps re-reading its last stage into the ring slot that
  // would be next -- free by construction, and never the slot the epilogue uses as scratch -- so every step issues 4 + 4 pieces
  // and the DMA wait is always vmcnt(4).
#define RF_BODY(first_step)                                                                                                       \
  {                                                                                                                              \
    RF_GROUPZ(first_step, a0, b0, 0, a1[0] = RF_FA(0, 2), RF_PBF(first_step, 0), a1[1] = RF_FA(0, 3), RF_PBF(first_step, 1), RF_PBF(first_step, 2), RF_PBF(first_step, 3), RF_PA(0), RF_NOP)   \
    RF_GROUPZ(first_step, a1, b0, 1, a0[0] = RF_FA(0, 4), RF_NOP, a0[1] = RF_FA(0, 5), RF_NOP, RF_PA(1), RF_NOP, RF_NOP, RF_NOP)              \
    RF_GROUPZ(first_step, a0, b0, 2, a1[0] = RF_FA(0, 6), RF_NOP, a1[1] = RF_FA(0, 7), RF_NOP, RF_PA(2), RF_NOP, RF_NOP, RF_NOP)              \
    RF_GROUPZ(first_step, a1, b0, 3, b1[0] = RF_FB(1, 0), a0[0] = RF_FA(1, 0), b1[1] = RF_FB(1, 1), b1[2] = RF_FB(1, 2), b1[3] = RF_FB(1, 3),   \
             a0[1] = RF_FA(1, 1), RF_PA(3), RF_NOP)                                                                              \
    RF_GROUP(a0, b1, 0, a1[0] = RF_FA(1, 2), RF_NOP, a1[1] = RF_FA(1, 3), RF_NOP, RF_NOP, RF_NOP, RF_NOP, RF_NOP)                \
    RF_GROUP(a1, b1, 1, a0[0] = RF_FA(1, 4), RF_NOP, a0[1] = RF_FA(1, 5), RF_NOP, RF_NOP, RF_NOP, RF_NOP, RF_NOP)                \
    RF_GROUP(a0, b1, 2, a1[0] = RF_FA(1, 6), RF_NOP, a1[1] = RF_FA(1, 7), RF_NOP, RF_BADV_PTR(), RF_AADV_PTR(), RF_NOP, RF_NOP)  \
    RF_BADV_TILE();                                                                                                              \
    RF_AADV_TILE();                                                                                                              \
    /* B(t+1) (and the older A(t+1)) have landed: the only younger pieces in this wave's queue are A(t+2)'s four */               \
    if (ABL & 64) {} /* ablation: no DMA wait at all */                                                                          \
    else if (!(ABL & (8 | 16 | 32))) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                            \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                        \
    /* (the builtin, not inline asm: hipcc must KNOW the LGKM queue is empty here -- the scalar loads of a cursor's tile crossing */ \
    /* would otherwise leave it assuming mixed SMEM / LDS events and turn every later counted lgkmcnt into lgkmcnt(0)) */         \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                                                          \
    if (!(ABL & 1)) pp_barrier();                                                                                                \
    sa_off = (sa_off == 2 * TILE2_BYTES) ? 0u : sa_off + TILE2_BYTES;                                                            \
    sb_off = (2 * PP_B_BASE + TILE2_BYTES) - sb_off;                                                                             \
    RF_STEP_BASES();                                                                                                             \
  }
  RF_LANE_BASES();
  RF_STEP_BASES();
  for (;;) {
    G2_T(0)
    const int nt = problem_ptr(c_pi)->K / BK2;
    int x_pi = -1, x_m = 0, x_n = 0;   // the consumer's next tile: read from the stash at the start of this tile's last K step
    bf16x8 b0[4], b1[4], a0[2], a1[2];
    // first fragments of the tile's first stage (visible since the barrier in front of the previous tile's epilogue)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b0[ni] = RF_FB(0, ni);
    a0[0] = RF_FA(0, 0);
    a0[1] = RF_FA(0, 1);
    f4v acc[8][4];
    RF_SB();
#define RF_TAKE_NEXT(last) \
  if (last) {              \
    x_pi = rbc_pi;         \
    x_m = rbc_m;           \
    x_n = rbc_n;           \
  }
    RF_TAKE_NEXT(nt == 1)
    RF_BODY(true)
    for (int t = 1; t < nt; ++t) {
      if (MIDSYNC && sync_mask > 0 && (t & sync_mask) == 0 && tile_no < sync_tiles && tid == 0) xcd_tile_sync(blockIdx.x & 7);
      // behind the barrier: request the first fragments of the new stage, then the group held back across the barrier
      // (group 7 of the previous step), which covers their latency
      b0[0] = RF_FB(0, 0);
      a0[0] = RF_FA(0, 0);
      RF_SB();
      // (B pieces alternate with the fragment reads: the same cycles per step on the NT / NN layouts, -1.7 % on TN, whose 48
      // transpose reads per step leave no MFMA without a filler -- tools/gemm_clk_ab.sh)
      RF_GROUP(a1, b1, 3, b0[1] = RF_FB(0, 1), RF_PB(0), b0[2] = RF_FB(0, 2), RF_PB(1), b0[3] = RF_FB(0, 3), RF_PB(2), a0[1] = RF_FA(0, 1), RF_PB(3))
      RF_TAKE_NEXT(t + 1 == nt)
      RF_BODY(false)
    }
    RF_GROUP(a1, b1, 3, RF_NOP, RF_NOP, RF_NOP, RF_NOP, RF_NOP, RF_NOP, RF_NOP, RF_NOP)   // the last step's group 7
    G2_T(1)
    const GemmProblem g = *problem_ptr(c_pi);
    const int epi = g.epi;
    // opaque lane id: the epilogue variants' lane constants are recomputed per tile -- hoisted out of the tile loop they are
    // spilled across the main loop (one set per variant)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    // the A slot consumed last (sa_off already points at the next one)
    unsigned char* scr = smem + ((sa_off == 0) ? 2 * TILE2_BYTES : sa_off - TILE2_BYTES) + wid * 4096;
    // ... and the B slot consumed last (free until this wave's own pieces of the next tile's B(1) go there, in its first step):
    // a second 4 KiB per wave, for the epilogue's operand tile (epilogue256, IN_DMA)
    unsigned char* scr2 = smem + ((2 * PP_B_BASE + TILE2_BYTES) - sb_off) + wid * 4096;
    if (!B_KS) {
      switch (epi) {
        case 0: epilogue256<(0), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2); break;
        case EPI_BIAS: epilogue256<(EPI_BIAS), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2); break;
        case EPI_BIAS | EPI_GELU: epilogue256<(EPI_BIAS | EPI_GELU), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2); break;
        case EPI_BIAS | EPI_GELU_FWD: epilogue256<(EPI_BIAS | EPI_GELU_FWD), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2); break;
        case EPI_BIAS | EPI_ADD: epilogue256<(EPI_BIAS | EPI_ADD), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2); break;
        case EPI_BIAS | EPI_ADD | EPI_DROP: epilogue256<(EPI_BIAS | EPI_ADD | EPI_DROP), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2); break;
        default: epilogue256<-1, 8, true, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2); break;
      }
    } else if (A_KS && epi == EPI_RMW32) {
      epilogue256<(EPI_RMW32), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2);
    } else if (A_KS && epi == EPI_STORE32) {   // weight gradients of the first backward pass after an optimizer step: overwrite
      epilogue256<(EPI_STORE32), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2);
    } else if (!A_KS && epi == EPI_ADD) {
      epilogue256<(EPI_ADD), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2);
    } else if (!A_KS && epi == 0) {
      epilogue256<(0), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2);
    } else if (!A_KS && epi == (EPI_DGELU | EPI_COLSUM | EPI_COLSUM_WS)) {
      // FFN-down dgrad: x GELU' + column sums (the two-stage kernel keeps this one generic: specialised there it spilled)
      epilogue256<(EPI_DGELU | EPI_COLSUM | EPI_COLSUM_WS), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2);
    } else if (!A_KS && epi == EPI_DGELU) {
      epilogue256<(EPI_DGELU), 8, false, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2);
    } else {
      epilogue256<-1, 8, true, true>(g, acc, m0, n0, wm, wn, lane_e, scr, scr2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // scratch reads done before this wave's next DMA lands there
    G2_T(2)
    ++tile_no;
    if (x_pi < 0) break;
    // long-K launches: meet the XCD's other workgroups before the next tile (every workgroup has at least sync_rounds + 1 tiles)
    if (tile_no <= sync_rounds && tid == 0) xcd_tile_sync(blockIdx.x & 7);
    c_pi = x_pi;
    m0 = x_m;
    n0 = x_n;
    lane_m = lane;
    asm volatile("" : "+v"(lane_m));
    RF_LANE_BASES();
    RF_STEP_BASES();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the re-read pieces of an exhausted cursor may still be in flight)
  G2_CLK(1)
#undef RF_ALOAD
#undef RF_BLOAD
#undef RF_AADV_PTR
#undef RF_BADV_PTR
#undef RF_AADV_TILE
#undef RF_BADV_TILE
#undef RF_SB
#undef RF_NOP
#undef RF_FA
#undef RF_FB
#undef RF_M
#undef RF_GROUP
#undef RF_GROUPZ
#undef RF_PB
#undef RF_PA
#undef RF_PBF
#undef RF_BODY
#undef RF_TAKE_NEXT
#undef RF_LANE_BASES
#undef RF_STEP_BASES
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: 128-row tiles on a DEEP ring -- single forward / dgrad problems whose 256 x 256 tiling would occupy at most half of the
// CUs (small micro-batches: the YAMLs' 4 sentences per optimizer step are M = 2048 tokens).  Such a GEMM is a chain of K / 64
// steps on every CU that has a tile, and with the two-stage loop (gemm256_kernel<..., 128>) each step is one exposed LDS-DMA round
// trip: 26-36 us per K = 1024 GEMM whatever M (profiles/round5_mb4x1_kernel_stats.md).  A 128 x 256 x 64 step is only 32 MFMAs per
// wave (~0.5 us per SIMD), so the landing time has to be hidden by depth, not by work: the A tile is 16 KiB here, which lets the
// 160 KiB of LDS hold FOUR A slots and THREE B slots -- A is requested three steps ahead, B two.  One workgroup per tile, no
// persistence (the launch has at most ~2 tiles per CU), one counted wait + one barrier per step:
//   issue order  A0 B0 A1 B1 A2 | step t: B(t+2) A(t+3)   ->  behind B(t) there are always A(t+1) + B(t+1) + A(t+2) = 8 pieces of
//   this wave, so the wait is a constant vmcnt(8); requests past the last stage re-read it into a slot nobody reads any more.
#define R128_A_BYTES 16384
#define R128_A_SLOTS 4
#define R128_B_SLOTS 3
#define R128_B_BASE (R128_A_SLOTS * R128_A_BYTES)
#define R128_LDS_BYTES (R128_B_BASE + R128_B_SLOTS * TILE2_BYTES)

template <bool B_KS>
__global__ __launch_bounds__(512, 2) void gemm128r_kernel(const GroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  GemmProblem g;
  int m0, n0;
  pick_tile<128>(ga, blockIdx.x, ga.total_tiles, g, m0, n0);
  const int nt = g.K / BK2;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);
  // per-lane source offsets of this wave's pieces: 2 of the A tile (128 rows x 128 B, row-major image), 4 of the B tile
  unsigned va[2], vb[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wid * 2 + j) * 8 + (lane >> 3);
    va[j] = (unsigned)(row * g.lda + (((lane & 7) ^ kc_swz(row)) << 3)) * 2u - (unsigned)j * 1024u;
  }
  stage_voff<B_KS, true>(g.ldb, wid, lane, vb);
  const bf16_t* a_base = g.A + (size_t)m0 * g.lda;
  const bf16_t* b_base = B_KS ? g.B + n0 : g.B + (size_t)n0 * g.ldb;
  const size_t b_stride = B_KS ? (size_t)BK2 * g.ldb : (size_t)BK2;
  const unsigned a_dst0 = lds0 + wid * 2048, b_dst0 = lds0 + R128_B_BASE + wid * 4096;
#define R1_ISSUE_A(T, SLOT)                                                                                   \
  {                                                                                                           \
    const int tt_ = (T) < nt ? (T) : nt - 1;                                                                  \
    glds16_pair<0>(uniform_ptr(a_base + (size_t)tt_ * BK2), va[0], va[1], a_dst0 + (unsigned)(SLOT) * R128_A_BYTES); \
  }
#define R1_ISSUE_B(T, SLOT)                                                                                   \
  {                                                                                                           \
    const int tt_ = (T) < nt ? (T) : nt - 1;                                                                  \
    glds16_quad(uniform_ptr(b_base + (size_t)tt_ * b_stride), vb[0], vb[1], vb[2], vb[3], b_dst0 + (unsigned)(SLOT) * TILE2_BYTES); \
  }
  R1_ISSUE_A(0, 0)
  R1_ISSUE_B(0, 0)
  R1_ISSUE_A(1, 1)
  R1_ISSUE_B(1, 1)
  R1_ISSUE_A(2, 2)

  f4v acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f4v){0.0f, 0.0f, 0.0f, 0.0f};

  int sa = 0, sb = 0;   // ring slots of the stage this step consumes
#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // the slots consumed one step ago are free: B(t+2) and A(t+3) go there
    R1_ISSUE_B(t + 2, sb == 0 ? 2 : sb - 1)
    R1_ISSUE_A(t + 3, (sa + 3) & 3)
    const unsigned char* as_ = smem + sa * R128_A_BYTES;
    const unsigned char* bs_ = smem + R128_B_BASE + sb * TILE2_BYTES;
#define R1_SB() __builtin_amdgcn_sched_barrier(0)
#define R1_LOADB(dst, ks) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) dst[ni] = fragB256<B_KS>(bs_, wn * 64, ni, ks, lane)
#define R1_LOADA(dst, ks, pr)                                          \
  dst[0] = frag256<false>(as_, wm * 64 + (2 * (pr)) * 16, ks, lane);   \
  dst[1] = frag256<false>(as_, wm * 64 + (2 * (pr) + 1) * 16, ks, lane)
#define R1_MM(a, b, pr)                                                                                                  \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) acc[2 * (pr) + j][ni] = \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[j], acc[2 * (pr) + j][ni], 0, 0, 0)
    bf16x8 b0[4], b1[4], a0[2], a1[2];
    R1_LOADB(b0, 0);
    R1_LOADA(a0, 0, 0);
    R1_SB();
    R1_LOADA(a1, 0, 1); R1_SB(); R1_MM(a0, b0, 0); R1_SB();
    R1_LOADB(b1, 1);
    R1_LOADA(a0, 1, 0); R1_SB(); R1_MM(a1, b0, 1); R1_SB();
    R1_LOADA(a1, 1, 1); R1_SB(); R1_MM(a0, b1, 0); R1_SB();
    R1_MM(a1, b1, 1);
    sa = (sa + 1) & 3;
    sb = sb == 2 ? 0 : sb + 1;
  }
  // the re-read pieces of the last steps may still be landing in the slots the epilogue uses as scratch
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const int epi = g.epi;
  unsigned char* scr = smem + wid * 4096;
  if (!B_KS) {
    switch (epi) {
      case 0: epilogue256<0, 4>(g, acc, m0, n0, wm, wn, lane, scr); break;
      case EPI_BIAS: epilogue256<EPI_BIAS, 4>(g, acc, m0, n0, wm, wn, lane, scr); break;
      case EPI_BIAS | EPI_GELU: epilogue256<(EPI_BIAS | EPI_GELU), 4>(g, acc, m0, n0, wm, wn, lane, scr); break;
      case EPI_BIAS | EPI_GELU_FWD: epilogue256<(EPI_BIAS | EPI_GELU_FWD), 4>(g, acc, m0, n0, wm, wn, lane, scr); break;
      case EPI_BIAS | EPI_ADD: epilogue256<(EPI_BIAS | EPI_ADD), 4>(g, acc, m0, n0, wm, wn, lane, scr); break;
      case EPI_BIAS | EPI_ADD | EPI_DROP: epilogue256<(EPI_BIAS | EPI_ADD | EPI_DROP), 4>(g, acc, m0, n0, wm, wn, lane, scr); break;
      default: epilogue256<-1, 4>(g, acc, m0, n0, wm, wn, lane, scr); break;
    }
  } else if (epi == EPI_ADD) {
    epilogue256<EPI_ADD, 4>(g, acc, m0, n0, wm, wn, lane, scr);
  } else if (epi == 0) {
    epilogue256<0, 4>(g, acc, m0, n0, wm, wn, lane, scr);
  } else {
    epilogue256<-1, 4>(g, acc, m0, n0, wm, wn, lane, scr);
  }
#undef R1_ISSUE_A
#undef R1_ISSUE_B
#undef R1_SB
#undef R1_LOADB
#undef R1_LOADA
#undef R1_MM
}

template <bool B_KS>
static int launch128r(const GroupArgs& ga, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  const int r = kbner_set_max_lds_once(attr_done, reinterpret_cast<const void*>(gemm128r_kernel<B_KS>), R128_LDS_BYTES);
  if (r) return r;
  hipLaunchKernelGGL((gemm128r_kernel<B_KS>), dim3(ga.total_tiles), dim3(512), R128_LDS_BYTES, stream, ga);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 6: the 128-row tiles on the INTERLEAVED ring.  gemm128r_kernel above hides the landing time of a stage by depth but still
// runs the plain wait - barrier - DMA burst - compute loop; this kernel is the K loop round 6 built for csrc/gemm128x.hip -- three
// 48-KiB stages (A 16 KiB + B 32 KiB), BOTH operands requested two K steps ahead (the constant wait is vmcnt(6): this step's own
// pieces), every fragment read and LDS-DMA piece alone between two MFMAs, the step's last MFMA group held across the barrier --
// as a loop over a run-time number of K steps with the ordinary epilogues behind it: one workgroup per tile (small micro-batches:
// the YAMLs' 4 sentences per optimizer step, where every forward / dgrad GEMM is a chain of 16 K steps on at most 256 tiles).
// Same fragment maps, MFMA order and epilogue arithmetic as the other kernels: bit-identical results.
#define R2_A_BYTES 16384
#define R2_B_BASE (3 * R2_A_BYTES)
#define R2_LDS_BYTES (R2_B_BASE + 3 * TILE2_BYTES)

template <bool B_KS>
__global__ __launch_bounds__(512, 2) void gemm128i_kernel(const GroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  int lane = tid & 63;
  asm volatile("" : "+v"(lane));
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  GemmProblem g;
  int m0, n0;
  pick_tile<128>(ga, blockIdx.x, ga.total_tiles, g, m0, n0);
  const int nt = g.K / BK2;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)smem);
  unsigned va[2], vb[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wid * 2 + j) * 8 + (lane >> 3);
    va[j] = (unsigned)(row * g.lda + (((lane & 7) ^ kc_swz(row)) << 3)) * 2u - (unsigned)j * 1024u;
  }
  stage_voff<B_KS, true>(g.ldb, wid, lane, vb);
  const bf16_t* a_base = uniform_ptr(g.A + (size_t)m0 * g.lda);
  const bf16_t* b_base = uniform_ptr(B_KS ? g.B + n0 : g.B + (size_t)n0 * g.ldb);
  const size_t b_kstep = B_KS ? (size_t)BK2 * g.ldb : (size_t)BK2;
  const unsigned a_dst0 = lds0 + wid * 2048, b_dst0 = lds0 + R2_B_BASE + wid * 4096;
  unsigned laneA, laneB;
  {
    const int row = wm * 64 + (lane & 15);
    laneA = lds0 + row * 128 + ((((lane >> 4)) ^ kc_swz(row)) << 4);
    if (!B_KS) {
      const int j = lane & 15;
      const int brow = wn * 64 + (j >> 2) * 8 + (j & 3);
      laneB = lds0 + R2_B_BASE + brow * 128 + ((((lane >> 4)) ^ kcb_swz(brow)) << 4);
    } else {
      const int p = lane & 15;
      const int r = (lane >> 4) * 8 + (p >> 2);
      laneB = lds0 + R2_B_BASE + r * 512 + (((wn * 4 + ((p & 3) >> 1)) ^ ks_swz(r)) << 5) + (((p & 3) & 1) << 4);
    }
  }
  typedef const s8v __attribute__((address_space(3))) lds_s8v;
  typedef s4v __attribute__((address_space(3))) lds_s4v;
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
  // prologue: stages 0 and 1 (a one-step tile re-reads stage 0 into slot 1: nobody reads it)
  {
    const int t1 = nt > 1 ? 1 : 0;
    glds16_pair<0>(a_base, va[0], va[1], a_dst0);
    glds16_quad(b_base, vb[0], vb[1], vb[2], vb[3], b_dst0);
    glds16_pair<0>(a_base + (size_t)t1 * BK2, va[0], va[1], a_dst0 + R2_A_BYTES);
    glds16_quad(b_base + (size_t)t1 * b_kstep, vb[0], vb[1], vb[2], vb[3], b_dst0 + TILE2_BYTES);
  }
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  pp_barrier();
  unsigned sa_off = 0, sb_off = 0;
  unsigned a_dst = a_dst0 + 2 * R2_A_BYTES, b_dst = b_dst0 + 2 * TILE2_BYTES;
  f4v acc[4][4];
  const f4v zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  bf16x8 b0[4], b1[4], a0[2], a1[2];
#define ISB() __builtin_amdgcn_sched_barrier(0)
#define INOP (void)0
#define IMF(a, b, mi, ni, Z) \
  acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[(mi) & 1], (Z) ? zero4 : acc[mi][ni], 0, 0, 0)
#define IGROUP(Z, a, b, pr, f0, f1, f2, f3, f4, f5, f6, f7) \
  IMF(a, b, 2 * (pr), 0, Z); ISB(); f0; ISB();               \
  IMF(a, b, 2 * (pr), 1, Z); ISB(); f1; ISB();               \
  IMF(a, b, 2 * (pr), 2, Z); ISB(); f2; ISB();               \
  IMF(a, b, 2 * (pr), 3, Z); ISB(); f3; ISB();               \
  IMF(a, b, 2 * (pr) + 1, 0, Z); ISB(); f4; ISB();           \
  IMF(a, b, 2 * (pr) + 1, 1, Z); ISB(); f5; ISB();           \
  IMF(a, b, 2 * (pr) + 1, 2, Z); ISB(); f6; ISB();           \
  IMF(a, b, 2 * (pr) + 1, 3, Z); ISB(); f7; ISB();
#define IPB(J) glds16_piece<J>(pb_src, vb[J], b_dst)
#define IPA(J) glds16_piece<J>(pa_src, va[J], a_dst)
  // groups 0-2 of a step (group 3 is held across the barrier), the wait, the barrier, the slot rotation
#define IBODY(Z)                                                                                                    \
  {                                                                                                                 \
    IGROUP(Z, a0, b0, 0, a1[0] = fa_(0, 2), IPA(0), a1[1] = fa_(0, 3), IPA(1), INOP, INOP, INOP, INOP)              \
    IGROUP(Z, a1, b0, 1, b1[0] = fb_(1, 0), a0[0] = fa_(1, 0), b1[1] = fb_(1, 1), b1[2] = fb_(1, 2), b1[3] = fb_(1, 3), \
           a0[1] = fa_(1, 1), INOP, INOP)                                                                           \
    IGROUP(false, a0, b1, 0, a1[0] = fa_(1, 2), INOP, a1[1] = fa_(1, 3), INOP, INOP, INOP, INOP, INOP)              \
    a_dst = (a_dst == a_dst0 + 2 * R2_A_BYTES) ? a_dst0 : a_dst + R2_A_BYTES;                                       \
    b_dst = (b_dst == b_dst0 + 2 * TILE2_BYTES) ? b_dst0 : b_dst + TILE2_BYTES;                                     \
    sa_off = (sa_off == 2 * R2_A_BYTES) ? 0u : sa_off + R2_A_BYTES;                                                 \
    sb_off = (sb_off == 2 * TILE2_BYTES) ? 0u : sb_off + TILE2_BYTES;                                               \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                                \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
    pp_barrier();                                                                                                   \
    pa0 = laneA + sa_off;                                                                                           \
    asm volatile("" : "+v"(pa0));                                                                                   \
    pa1 = pa0 ^ 64u;                                                                                                \
    asm volatile("" : "+v"(pa1));                                                                                   \
    pb0 = laneB + sb_off;                                                                                           \
    asm volatile("" : "+v"(pb0));                                                                                   \
    pb1 = pb0 ^ 64u;                                                                                                \
    asm volatile("" : "+v"(pb1));                                                                                   \
  }
  // step 0: first fragments, the B pieces of stage 2 in a block, groups 0-2
  {
    const int tt = nt > 2 ? 2 : nt - 1;
    const bf16_t* pa_src = a_base + (size_t)tt * BK2;
    const bf16_t* pb_src = b_base + (size_t)tt * b_kstep;
    b0[0] = fb_(0, 0); b0[1] = fb_(0, 1); b0[2] = fb_(0, 2); b0[3] = fb_(0, 3);
    a0[0] = fa_(0, 0); a0[1] = fa_(0, 1);
    ISB();
    IPB(0); IPB(1); IPB(2); IPB(3);
    ISB();
    IBODY(true)
  }
#pragma unroll 1
  for (int t = 1; t < nt; ++t) {
    const int tt = t + 2 < nt ? t + 2 : nt - 1;
    const bf16_t* pa_src = a_base + (size_t)tt * BK2;
    const bf16_t* pb_src = b_base + (size_t)tt * b_kstep;
    b0[0] = fb_(0, 0);
    a0[0] = fa_(0, 0);
    ISB();
    IGROUP(false, a1, b1, 1, b0[1] = fb_(0, 1), IPB(0), b0[2] = fb_(0, 2), IPB(1), b0[3] = fb_(0, 3), IPB(2), a0[1] = fa_(0, 1), IPB(3))
    IBODY(false)
  }
  IGROUP(false, a1, b1, 1, INOP, INOP, INOP, INOP, INOP, INOP, INOP, INOP)   // the last step's held group
#undef IBODY
#undef IPA
#undef IPB
#undef IGROUP
#undef IMF
#undef INOP
#undef ISB
  // the re-read pieces of the last steps may still be landing in the slots the epilogue uses as scratch
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const int epi = g.epi;
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  unsigned char* scr = smem + wid * 4096;
  if (!B_KS) {
    switch (epi) {
      case 0: epilogue256<0, 4>(g, acc, m0, n0, wm, wn, lane_e, scr); break;
      case EPI_BIAS: epilogue256<EPI_BIAS, 4>(g, acc, m0, n0, wm, wn, lane_e, scr); break;
      case EPI_BIAS | EPI_GELU: epilogue256<(EPI_BIAS | EPI_GELU), 4>(g, acc, m0, n0, wm, wn, lane_e, scr); break;
      case EPI_BIAS | EPI_GELU_FWD: epilogue256<(EPI_BIAS | EPI_GELU_FWD), 4>(g, acc, m0, n0, wm, wn, lane_e, scr); break;
      case EPI_BIAS | EPI_ADD: epilogue256<(EPI_BIAS | EPI_ADD), 4>(g, acc, m0, n0, wm, wn, lane_e, scr); break;
      case EPI_BIAS | EPI_ADD | EPI_DROP: epilogue256<(EPI_BIAS | EPI_ADD | EPI_DROP), 4>(g, acc, m0, n0, wm, wn, lane_e, scr); break;
      default: epilogue256<-1, 4>(g, acc, m0, n0, wm, wn, lane_e, scr); break;
    }
  } else if (epi == EPI_ADD) {
    epilogue256<EPI_ADD, 4>(g, acc, m0, n0, wm, wn, lane_e, scr);
  } else if (epi == 0) {
    epilogue256<0, 4>(g, acc, m0, n0, wm, wn, lane_e, scr);
  } else {
    epilogue256<-1, 4>(g, acc, m0, n0, wm, wn, lane_e, scr);
  }
}

template <bool B_KS>
static int launch128i(const GroupArgs& ga, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  const int r = kbner_set_max_lds_once(attr_done, reinterpret_cast<const void*>(gemm128i_kernel<B_KS>), R2_LDS_BYTES);
  if (r) return r;
  hipLaunchKernelGGL((gemm128i_kernel<B_KS>), dim3(ga.total_tiles), dim3(512), R2_LDS_BYTES, stream, ga);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// one_tile_each: grid = number of tiles, every workgroup computes exactly ONE (its cursors find no next tile): the hardware
// dispatcher then hands tiles to CUs as they become free -- the dynamic schedule of the long-K launches at N > 1 (see
// gemm_grouped_impl), at the price of a cold prologue and an exposed epilogue per tile, which a K >= 3072 tile does not notice.
template <bool A_KS, bool B_KS, int ABL = 0, bool MIDSYNC = false>
static int launch256f(const GroupArgs& ga, hipStream_t stream, bool one_tile_each = false) {
  static std::atomic<unsigned long long> attr_done{0};
  const int r = kbner_set_max_lds_once(attr_done, reinterpret_cast<const void*>(gemm256f_kernel<A_KS, B_KS, ABL, MIDSYNC>), PP_LDS_BYTES);
  if (r) return r;
  const int grid = (one_tile_each || ga.total_tiles < ga.ncu) ? ga.total_tiles : ga.ncu;
  hipLaunchKernelGGL((gemm256f_kernel<A_KS, B_KS, ABL, MIDSYNC>), dim3(grid), dim3(512), PP_LDS_BYTES, stream, ga);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// which main loop the 256-row static launches use: 1 = the interleaved ring loop (gemm256f_kernel, default), 0 = the two-stage loop
// of rounds 1-3 (gemm256_kernel; also what the 128-row and the dynamic-tile launches run).  Process-wide, atomic; set through
// kbner_gemm_set_variant (include/kbner.h).  Trace builds (-DG2_TRACE) read compile-time ablations of the ring loop from bits 12-15.
static std::atomic<int> g_gemm_variant{KBNER_GEMM_VARIANT_DEFAULT};

template <bool A_KS, bool B_KS, bool DYN, int TM = 256>
static int launch256(const GroupArgs& ga, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};   // per template instantiation, one bit per device
  const int r = kbner_set_max_lds_once(attr_done, reinterpret_cast<const void*>(gemm256_kernel<A_KS, B_KS, DYN, TM>), G2_LDS_BYTES);
  if (r) return r;
  const int grid = ga.total_tiles < ga.ncu ? ga.total_tiles : ga.ncu;
  hipLaunchKernelGGL((gemm256_kernel<A_KS, B_KS, DYN, TM>), dim3(grid), dim3(512), G2_LDS_BYTES, stream, ga);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

static int device_cu_count() { return kbner_cu_count(); }

// public mirror of GemmProblem (include/kbner.h: kbner_gemm_problem)
struct kbner_gemm_problem {
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
  uint32_t drop_seed;
  uint32_t drop_thresh;
};

#ifdef G2_TRACE
extern "C" int kbner_debug_read_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g2_trace), sizeof(unsigned long long) * 256 * 32 * 4);
}
extern "C" int kbner_debug_read_clk(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g2_clk), sizeof(unsigned long long) * 256 * 4);
}
#endif

static int pick_tile_rows(int layout, int nprob, const kbner_gemm_problem* probs, bool dyn, int variant);

extern "C" {

// rows of an output tile kbner_gemm_bf16_grouped uses for a SINGLE problem of this layout and shape (256, or 128 when the
// 256 x 256 tiling would occupy at most half of the CUs): callers of KBNER_EPI_COLSUM_WS size and fold 2 * M / rows lines
int kbner_gemm_tile_rows(int layout, int M, int N) {
  kbner_gemm_problem p = {};
  p.M = M;
  p.N = N;
  return pick_tile_rows(layout, 1, &p, false, g_gemm_variant.load(std::memory_order_relaxed));
}

// Grouped GEMM: nprob (1..16) problems of the SAME layout in one launch.
// Constraints per problem: M % 256 == 0, N % 256 == 0, K % 64 == 0, lda/ldb % 8 == 0.
static int gemm_grouped_impl(int layout, int nprob, const kbner_gemm_problem* probs, int* sched, void* stream);
static int pick_tile_rows(int layout, int nprob, const kbner_gemm_problem* probs, bool dyn, int variant);

int kbner_gemm_set_variant(int variant) {
  KBNER_CHECK_ARG(variant >= 0 && variant < 65536);   // (bits 12-15: trace-build ablations)
  g_gemm_variant.store(variant, std::memory_order_relaxed);
  return 0;
}
int kbner_gemm_get_variant(void) { return g_gemm_variant.load(std::memory_order_relaxed); }

int kbner_gemm_bf16_grouped(int layout, int nprob, const kbner_gemm_problem* probs, void* stream) {
  return gemm_grouped_impl(layout, nprob, probs, nullptr, stream);
}

// Same launch with DYNAMIC tile scheduling (see draw_tile): sched = 8 device ints the caller zeroed (on this stream) since their
// last use.  Meant for steps whose GEMMs share the GPU with RCCL collectives; identical results, static tile->XCD affinity kept.
int kbner_gemm_bf16_grouped_dyn(int layout, int nprob, const kbner_gemm_problem* probs, int* sched, void* stream) {
  KBNER_CHECK_ARG(sched != nullptr);
  return gemm_grouped_impl(layout, nprob, probs, sched, stream);
}

}  // extern "C"

// tile height of a launch: 128-row tiles when the problem is a single forward / dgrad GEMM whose 256 x 256 tiling would give at
// most half of the CUs a tile (and the static walk is used); 256 otherwise
// (variant bit 4: a single K = 1024 problem with one of gemm128x.hip's epilogues runs on 128 x 256 tiles whose epilogue overlaps the
// next tile's K loop -- when the launch has at least two such tiles per CU, otherwise the exposed last epilogue is all there is)
static bool wants_128x(int layout, int nprob, const kbner_gemm_problem* probs, bool dyn, int variant) {
  if (dyn || nprob != 1 || !(variant & (16 | 32))) return false;
  const kbner_gemm_problem& s = probs[0];
  return kbner_can128x(layout, s.M, s.N, s.K, s.epi) && (long)(s.M / 128) * (s.N / T2) >= 2L * device_cu_count();
}
// (variant: ONE load of the process-wide switch per launch, passed down -- a concurrent kbner_gemm_set_variant cannot make the tile
// height and the kernel choice of one launch disagree.  Grouped NT / NN launches -- the K slices of a split-K GEMM -- take 128-row
// tiles with the ring kernels (variant bit 0 set, bit 3 clear); kbner_gemm_tile_rows answers for nprob == 1 only, which is what
// KBNER_EPI_COLSUM_WS callers launch.)
static int pick_tile_rows(int layout, int nprob, const kbner_gemm_problem* probs, bool dyn, int variant) {
  if (wants_128x(layout, nprob, probs, dyn, variant)) return 128;
  if (layout == 2 || dyn) return T2;
  // (a grouped launch too -- the K slices of a split-K GEMM -- when the ring kernels run it: round 5)
  if (nprob != 1 && (!(variant & 1) || (variant & 8))) return T2;
  long tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const kbner_gemm_problem& s = probs[i];
    if (s.M % T2 != 0 || s.N % T2 != 0) return T2;
    tiles += (long)(s.M / T2) * (s.N / T2);
  }
  return 2 * tiles <= device_cu_count() ? 128 : T2;
}

static int gemm_grouped_impl(int layout, int nprob, const kbner_gemm_problem* probs, int* sched, void* stream) {
  KBNER_CHECK_ARG(layout >= 0 && layout <= 2 && nprob >= 1 && nprob <= G2_MAXP && probs != nullptr);
  GroupArgs ga;
  ga.nprob = nprob;
  const int variant = g_gemm_variant.load(std::memory_order_relaxed);   // the ONE read of the switch for this launch
  const int TM = pick_tile_rows(layout, nprob, probs, sched != nullptr, variant);
  int tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const kbner_gemm_problem& s = probs[i];
    KBNER_CHECK_ARG(s.M > 0 && s.N > 0 && s.K > 0 && s.M % T2 == 0 && s.N % T2 == 0 && s.K % BK2 == 0);
    KBNER_CHECK_ARG(s.A != nullptr && s.B != nullptr && s.lda % 8 == 0 && s.ldb % 8 == 0);
    if (s.epi & (EPI_ATOMIC32 | EPI_RMW32 | EPI_STORE32)) {
      KBNER_CHECK_ARG(s.C32 != nullptr && s.ldc32 >= s.N && s.ldc32 % 4 == 0);
    } else {
      KBNER_CHECK_ARG(s.C != nullptr && s.ldc >= s.N && s.ldc % 8 == 0);
    }
    if (s.epi & EPI_BIAS) KBNER_CHECK_ARG(s.bias != nullptr);
    if (s.epi & EPI_ADD) KBNER_CHECK_ARG(s.addend != nullptr && s.ldadd % 8 == 0);
    if (s.epi & EPI_DGELU) KBNER_CHECK_ARG(s.aux != nullptr && s.ldaux % 8 == 0);
    if (s.epi & EPI_GELU) KBNER_CHECK_ARG(s.out2 != nullptr && s.ldout2 % 8 == 0);
    if (s.epi & EPI_COLSUM) KBNER_CHECK_ARG(s.colsum != nullptr && !(s.epi & (EPI_ATOMIC32 | EPI_RMW32 | EPI_STORE32)));
    if (s.epi & EPI_COLSUM_WS) KBNER_CHECK_ARG((s.epi & EPI_COLSUM) != 0 && s.N % 4 == 0);
    if (s.epi & EPI_STORE32) KBNER_CHECK_ARG(s.epi == EPI_STORE32);
    if (s.epi & EPI_DROP) KBNER_CHECK_ARG(!(s.epi & (EPI_ATOMIC32 | EPI_RMW32 | EPI_COLSUM | EPI_GELU | EPI_DGELU | EPI_GELU_FWD)));
    if (s.epi & EPI_GELU_FWD) KBNER_CHECK_ARG(!(s.epi & (EPI_GELU | EPI_DGELU | EPI_ATOMIC32 | EPI_RMW32 | EPI_STORE32 | EPI_COLSUM)));
    GemmProblem& d = ga.p[i];
    d.A = s.A; d.B = s.B; d.C = s.C; d.C32 = s.C32; d.bias = s.bias; d.addend = s.addend; d.aux = s.aux; d.out2 = s.out2; d.colsum = s.colsum;
    d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc; d.ldc32 = s.ldc32; d.ldadd = s.ldadd;
    d.ldaux = s.ldaux; d.ldout2 = s.ldout2; d.epi = s.epi; d.alpha = s.alpha; d.tile_begin = tiles; d.pad_ = 0;
    d.drop_seed = s.drop_seed; d.drop_thresh = (s.epi & EPI_DROP) ? s.drop_thresh : 0u;
    ga.tile_begin[i] = tiles;
    tiles += (s.M / TM) * (s.N / T2);
  }
  for (int i = nprob; i < G2_MAXP; ++i) {
    ga.p[i] = ga.p[0];
    ga.p[i].tile_begin = 0x7fffffff;
    ga.tile_begin[i] = 0x7fffffff;
  }
  ga.total_tiles = tiles;
  ga.ncu = device_cu_count();
  // tile-boundary sync of the ring kernel (variant bit 1): every problem's K loop is long enough for the drift to matter
  int min_k = 0x7fffffff;
  for (int i = 0; i < nprob; ++i) min_k = probs[i].K < min_k ? probs[i].K : min_k;
  int max_k = 0;
  for (int i = 0; i < nprob; ++i) max_k = probs[i].K > max_k ? probs[i].K : max_k;
  const int gv = variant;
  ga.pad_ = ((gv & 2) && min_k >= 256 * BK2 && tiles >= 2 * ga.ncu) ? 1 : 0;
  if (ga.pad_ && (gv & 4) && layout == 2 && min_k == max_k && min_k >= 512 * BK2) ga.pad_ |= 256 << 8;   // + every 256 K steps inside a tile
  ga.sched = sched;
  hipStream_t st = (hipStream_t)stream;
  if (sched != nullptr && (gv & 1) && min_k >= 1024) {
    // Dynamic scheduling (round 5): the ring kernel, ONE workgroup per tile.  No draw, no counters: the dispatcher places the
    // 768 (weight gradients) ... 4096 workgroups on whatever CUs are free, which is exactly what a step that shares the GPU with a
    // collective needs (tools/contention_lab.py: 8 CUs held -> static walk +45 %, this +13 %, the tile draw on the two-stage loop
    // +10 % of a 7 % slower loop).  A tile pays a cold prologue and an exposed epilogue: undisturbed, the whole step with EVERY
    // launch dynamic runs at 133.6 ms against 130.8 static (the draw on the two-stage loop: 140.6).  Launches with a K loop
    // shorter than 16 steps (none in the encoder) keep the draw below.  `sched` is left untouched.
    ga.pad_ = 0;
    switch (layout) {
      case 0: return launch256f<false, false>(ga, st, true);
      case 1: return launch256f<false, true>(ga, st, true);
      default: return launch256f<true, true>(ga, st, true);
    }
  }
  if (sched != nullptr) {
    switch (layout) {
      case 0: return launch256<false, false, true>(ga, st);
      case 1: return launch256<false, true, true>(ga, st);
      default: return launch256<true, true, true>(ga, st);
    }
  }
  if (TM == 128 && wants_128x(layout, nprob, probs, sched != nullptr, variant))
    return (variant & 32) ? kbner_launch128s(layout, ga, st) : kbner_launch128x(layout, ga, st);
  if (TM == 128) {
    // (variant bit 3 clear = default: the deep-ring kernel; set: the two-stage loop's 128-row tiles of rounds 2-4, for the A/B)
    // (variant bit 6 set: round 5's deep-ring kernel with the plain loop, for the A/B; clear = default: the interleaved ring, round 6)
    if ((variant & 1) && !(variant & 8) && !(variant & 64)) return layout == 0 ? launch128i<false>(ga, st) : launch128i<true>(ga, st);
    if ((variant & 1) && !(variant & 8)) return layout == 0 ? launch128r<false>(ga, st) : launch128r<true>(ga, st);
    return layout == 0 ? launch256<false, false, false, 128>(ga, st) : launch256<false, true, false, 128>(ga, st);
  }
#ifdef G2_TRACE
  if ((variant & 1) && layout == 2 && (variant & 0xF000)) {   // cycle-accounting ablations, TN: no DMA wait / no DMA
    switch ((variant >> 12) & 15) {
      case 4: return launch256f<true, true, 64>(ga, st);
      default: return launch256f<true, true, 8>(ga, st);
    }
  }
  if ((variant & 1) && layout == 0 && (variant & 0xF000)) {   // cycle-accounting ablations, NT
    switch ((variant >> 12) & 15) {
      case 1: return launch256f<false, false, 1>(ga, st);
      case 2: return launch256f<false, false, 16>(ga, st);
      case 3: return launch256f<false, false, 32>(ga, st);
      case 4: return launch256f<false, false, 64>(ga, st);
      case 8: return launch256f<false, false, 8>(ga, st);
      default: return launch256f<false, false, 9>(ga, st);
    }
  }
#endif
  if (variant & 1) {
    switch (layout) {
      case 0: return launch256f<false, false>(ga, st);
      case 1: return launch256f<false, true>(ga, st);
      default: return (ga.pad_ >> 8) ? launch256f<true, true, 0, true>(ga, st) : launch256f<true, true>(ga, st);
    }
  }
  switch (layout) {
    case 0: return launch256<false, false, false>(ga, st);
    case 1: return launch256<false, true, false>(ga, st);
    default: return launch256<true, true, false>(ga, st);
  }
}
