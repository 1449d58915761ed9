// bf16 MFMA GEMM for gfx950 (CDNA4): C[M,N] = sum_k A[m,k] * B[n,k], fp32 accumulate.
//
// One kernel template covers the three layouts the encoder needs, by operand "image" kind:
//   KC (k-contiguous): memory [rows][K]   -> fragments by ds_read_b128
//   KS (k-strided)   : memory [K][rows]   -> fragments by ds_read_b64_tr_b16 (LDS transpose read)
//   forward   Y  = X W^T        : A = X  (KC), B = W  (KC)      (W is [out,in] as HF stores it)
//   dgrad     dX = dY W         : A = dY (KC), B = W  (KS)
//   wgrad     dW += dY^T X      : A = dY (KS), B = X  (KS), split-K over tokens, fp32 atomics
// so no transposed copies of weights, activations or gradients ever touch HBM.
//
// Tiling: 128 x 128 x 64 block tile, 256 threads = 4 wavefronts (2 x 2), each wave 64 x 64 =
// 4 x 4 MFMA 16x16x32 fragments; global->LDS by global_load_lds (16 B/lane, LDS image lane-linear,
// XOR swizzle applied on the per-lane SOURCE address and again on the read: both-or-neither);
// two LDS stages (64 KiB) so the DMA of tile t+1 flies under the MFMAs of tile t, one barrier
// per K tile.  MFMAs are issued operand-swapped (C^T fragments) so each lane owns 4 consecutive
// output columns: 8-byte bf16 / 16-byte fp32 epilogue accesses.  blockIdx is remapped so each
// XCD's private L2 sees a contiguous run of tiles.
//
// Replaces the cuBLAS calls behind torch.nn.Linear / autograd in transformers' BertSelfAttention,
// BertSelfOutput, BertIntermediate, BertOutput (reached from flair/embeddings.py:3269).
#include "common.h"

#define EPI_BIAS 1     // + bias[n]
#define EPI_GELU 2     // C = gelu(pre), out2 = gelu'(pre) (bf16)
#define EPI_ADD 4      // + addend[m,n] (bf16)
#define EPI_GELU_FWD 1024  // C = gelu(pre) alone (inference)
#define EPI_DGELU 8    // * aux[m,n]  (aux = the gelu'(pre) the forward epilogue stored)
#define EPI_ATOMIC32 16  // atomicAdd into C32 (fp32), no bf16 output
#define EPI_DROP 128     // dropout on (acc*alpha + bias) before the residual add

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  int M, N, K;
  int lda, ldb;
  bf16_t* C;
  int ldc;
  float* C32;
  int ldc32;
  const float* bias;
  const bf16_t* addend;
  int ldadd;
  const bf16_t* aux;
  int ldaux;
  bf16_t* out2;
  int ldout2;
  int splitk;
  int epi;
  float alpha;
  uint32_t drop_seed;
  uint32_t drop_thresh;
};

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES 32768
#define TILE_BYTES 16384

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_cvoid;

static __device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((glb_cvoid*)g, (lds_void*)l, 16, 0, 0);
}

// KC image: [128 rows][64 k] bf16, 128-B rows = 8 chunks of 16 B; logical chunk c of row r lives
// at chunk (c ^ swz(r)), swz(r) = (r >> 1) & 7  (conflict-free ds_read_b128 for 16 rows x 1 chunk)
static __device__ __forceinline__ int kc_swz(int row) { return (row >> 1) & 7; }

template <bool KS>
static __device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ P, int ld, int row0, int k0, unsigned char* s,
                                                  int wid, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wid * 4 + j;  // 16 wave-instructions of 1 KiB cover the 16 KiB tile
    const bf16_t* src;
    if (!KS) {
      const int row = q * 8 + (lane >> 3);
      const int pos = lane & 7;
      src = P + (size_t)(row0 + row) * ld + k0 + ((pos ^ kc_swz(row)) << 3);
    } else {
      const int kr = q * 4 + (lane >> 4);
      const int pos = lane & 15;
      src = P + (size_t)(k0 + kr) * ld + row0 + (pos << 3);
    }
    glds16(src, s + q * 1024);
  }
}

// fragment of 16 rows (r0..r0+15) x 32 k (k-step ks) for mfma_f32_16x16x32_bf16:
// lane (g = lane>>4, i = lane&15) holds row r0+i, k = ks*32 + g*8 + 0..7
template <bool KS>
static __device__ __forceinline__ bf16x8 load_frag(const unsigned char* s, int r0, int ks, int lane) {
  if (!KS) {
    const int row = r0 + (lane & 15);
    const int c = ks * 4 + (lane >> 4);
    const s8v v = *reinterpret_cast<const s8v*>(s + row * 128 + ((c ^ kc_swz(row)) << 4));
    return __builtin_bit_cast(bf16x8, v);
  } else {
    // KS image: [64 k][128 cols], 256-B rows.  ds_read_b64_tr_b16: within a 16-lane group, lane p
    // supplies the address of (k-row p/4, 4 columns (p%4)*4..) and receives column p of k-rows 0..3.
    const int p = lane & 15;
    const int kb = ks * 32 + (lane >> 4) * 8;
    const unsigned char* a = s + (kb + (p >> 2)) * 256 + (r0 + (p & 3) * 4) * 2;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a + 4 * 256));
    s8v v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
  }
}

template <bool A_KS, bool B_KS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

  // XCD-aware bijective remap: hardware places block b on XCD b % 8; give each XCD a contiguous run
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);

  const int tiles_n = g.N / BN;
  const int z = wg % g.splitk;
  const int tile = wg / g.splitk;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kper = g.K / g.splitk;
  const int kbeg = z * kper;
  const int nt = kper / BK;

  f4v acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f4v){0.0f, 0.0f, 0.0f, 0.0f};

  // prologue: tile 0 -> stage 0
  stage_tile<A_KS>(g.A, g.lda, m0, kbeg, smem, wid, lane);
  stage_tile<B_KS>(g.B, g.ldb, n0, kbeg, smem + TILE_BYTES, wid, lane);

  for (int t = 0; t < nt; ++t) {
    // tile t has landed (every wave drains its own DMA, then the barrier publishes all of it);
    // the barrier also fences the previous iteration's reads of the stage we are about to refill
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned char* cur = smem + (t & 1) * STAGE_BYTES;
    if (t + 1 < nt) {
      unsigned char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
      stage_tile<A_KS>(g.A, g.lda, m0, kbeg + (t + 1) * BK, nxt, wid, lane);
      stage_tile<B_KS>(g.B, g.ldb, n0, kbeg + (t + 1) * BK, nxt + TILE_BYTES, wid, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[mi] = load_frag<A_KS>(cur, wm * 64 + mi * 16, ks, lane);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bfr[ni] = load_frag<B_KS>(cur + TILE_BYTES, wn * 64 + ni * 16, ks, lane);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          // operand-swapped: D = Bfrag x Afrag^T = C^T fragment -> lane holds C[m = lane&15][n = g*4 + r]
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
  }

  // epilogue
  const int epi = g.epi;
  const int gq = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + (lane & 15);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + gq * 4;
      float v[4] = {acc[mi][ni][0] * g.alpha, acc[mi][ni][1] * g.alpha, acc[mi][ni][2] * g.alpha, acc[mi][ni][3] * g.alpha};
      if (epi & EPI_ATOMIC32) {
        float* c = g.C32 + (size_t)m * g.ldc32 + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(c + r, v[r]);
        continue;
      }
      if (epi & EPI_BIAS) {
        const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (epi & EPI_DROP) {  // same counter-based mask as gemm256.hip (common.h)
        const uint32_t rk = drop_rowkey(g.drop_seed, (uint32_t)m);
        const float ds = drop_scale(g.drop_thresh);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = drop_keep(rk, drop_colkey(g.drop_seed, (uint32_t)(n + r)), g.drop_thresh) ? v[r] * ds : 0.0f;
      }
      if (epi & EPI_ADD) {
        const uint2 u = *reinterpret_cast<const uint2*>(g.addend + (size_t)m * g.ldadd + n);
        v[0] += __uint_as_float(u.x << 16);
        v[1] += __uint_as_float(u.x & 0xffff0000u);
        v[2] += __uint_as_float(u.y << 16);
        v[3] += __uint_as_float(u.y & 0xffff0000u);
      }
      if (epi & EPI_DGELU) {
        const uint2 u = *reinterpret_cast<const uint2*>(g.aux + (size_t)m * g.ldaux + n);
        // aux holds gelu'(pre), stored by the forward EPI_GELU epilogue
        v[0] *= __uint_as_float(u.x << 16);
        v[1] *= __uint_as_float(u.x & 0xffff0000u);
        v[2] *= __uint_as_float(u.y << 16);
        v[3] *= __uint_as_float(u.y & 0xffff0000u);
      }
      if (epi & EPI_GELU_FWD) {
        const f2v y0 = gelu2((f2v){v[0], v[1]}), y1 = gelu2((f2v){v[2], v[3]});
        v[0] = y0[0]; v[1] = y0[1]; v[2] = y1[0]; v[3] = y1[1];
      }
      if (epi & EPI_GELU) {
        // C = gelu(pre), out2 = gelu'(pre), both evaluated at the fp32 pre-activation (rounds 1-3 rounded it to bf16 first, as if it
        // had been stored: 3 of the ~30 vector instructions per element pair of this issue-bound epilogue, and a rounding the
        // reference's fp32 path does not have) (same as gemm256.hip)
        f2v y0, d0, y1, d1;
        gelu_both2((f2v){v[0], v[1]}, y0, d0);
        gelu_both2((f2v){v[2], v[3]}, y1, d1);
        uint2 du;
        du.x = pack2bf(d0[0], d0[1]);
        du.y = pack2bf(d1[0], d1[1]);
        *reinterpret_cast<uint2*>(g.out2 + (size_t)m * g.ldout2 + n) = du;
        v[0] = y0[0]; v[1] = y0[1]; v[2] = y1[0]; v[3] = y1[1];
      }
      uint2 o;
      o.x = pack2bf(v[0], v[1]);
      o.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<uint2*>(g.C + (size_t)m * g.ldc + n) = o;
    }
  }
}

template <bool A_KS, bool B_KS>
static int launch_gemm(const GemmArgs& g, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};   // per template instantiation, one bit per device
  const int r = kbner_set_max_lds_once(attr_done, reinterpret_cast<const void*>(gemm_kernel<A_KS, B_KS>), 2 * STAGE_BYTES);
  if (r) return r;
  const int grid = (g.M / BM) * (g.N / BN) * g.splitk;
  hipLaunchKernelGGL((gemm_kernel<A_KS, B_KS>), dim3(grid), dim3(256), 2 * STAGE_BYTES, stream, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

extern "C" {

// layout: 0 = NT (A [M,K], B [N,K]); 1 = NN dgrad (A [M,K], B memory [K,N]); 2 = TN wgrad (A memory
// [K,M], B memory [K,N]).  C (bf16, ldc) unless epi has EPI_ATOMIC32 (then C32 += result).
// Constraints: M % 128 == 0, N % 128 == 0, K % (64 * splitk) == 0, all ld % 8 == 0.
int kbner_gemm_bf16(int layout, const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, bf16_t* C, int ldc,
                    float* C32, int ldc32, const float* bias, const bf16_t* addend, int ldadd, const bf16_t* aux, int ldaux,
                    bf16_t* out2, int ldout2, int epi, int splitk, float alpha, uint32_t drop_seed, uint32_t drop_thresh,
                    void* stream) {
  KBNER_CHECK_ARG(layout >= 0 && layout <= 2);
  KBNER_CHECK_ARG(M > 0 && N > 0 && K > 0 && splitk >= 1);
  KBNER_CHECK_ARG(M % BM == 0 && N % BN == 0 && K % (BK * splitk) == 0);
  KBNER_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0);
  KBNER_CHECK_ARG(A != nullptr && B != nullptr);
  if (epi & EPI_ATOMIC32) {
    KBNER_CHECK_ARG(C32 != nullptr && ldc32 >= N);
  } else {
    KBNER_CHECK_ARG(C != nullptr && ldc >= N && ldc % 4 == 0 && splitk == 1);
  }
  if (epi & EPI_BIAS) KBNER_CHECK_ARG(bias != nullptr);
  if (epi & EPI_ADD) KBNER_CHECK_ARG(addend != nullptr && ldadd % 4 == 0);
  if (epi & EPI_DGELU) KBNER_CHECK_ARG(aux != nullptr && ldaux % 4 == 0);
  if (epi & EPI_GELU) KBNER_CHECK_ARG(out2 != nullptr && ldout2 % 4 == 0);
  GemmArgs g;
  g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
  g.C = C; g.ldc = ldc; g.C32 = C32; g.ldc32 = ldc32; g.bias = bias;
  g.addend = addend; g.ldadd = ldadd; g.aux = aux; g.ldaux = ldaux; g.out2 = out2; g.ldout2 = ldout2;
  g.splitk = splitk; g.epi = epi; g.alpha = alpha;
  g.drop_seed = drop_seed; g.drop_thresh = (epi & EPI_DROP) ? drop_thresh : 0u;
  if (epi & EPI_DROP) KBNER_CHECK_ARG(!(epi & (EPI_ATOMIC32 | EPI_GELU | EPI_DGELU)));
  hipStream_t s = (hipStream_t)stream;
  switch (layout) {
    case 0: return launch_gemm<false, false>(g, s);
    case 1: return launch_gemm<false, true>(g, s);
    default: return launch_gemm<true, true>(g, s);
  }
}

}  // extern "C"
