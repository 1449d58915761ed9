// Shared device/host helpers for the kbner HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KBNER_OK 0
#define KBNER_EINVAL (-22)

#define KBNER_CHECK_ARG(cond) \
  do {                        \
    if (!(cond)) return KBNER_EINVAL; \
  } while (0)

// Every launcher returns 0 or -(hipError).  hipGetLastError also clears the sticky flag.
#define KBNER_LAUNCH_RET()                     \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    return e__ == hipSuccess ? 0 : -(int)e__;  \
  } while (0)

typedef uint16_t bf16_t;  // raw bfloat16 storage

// One-off per-DEVICE host-side initialisation (a kernel's dynamic-LDS limit, a device's CU count): a bit per device ordinal in an
// atomic mask, so a second device gets its own call and two host threads racing here both make the same idempotent call.  This
// is, with the two A/B switches listed there, the mutable process state of the library (include/kbner.h, conventions).
#include <atomic>
static inline int kbner_device_ordinal() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev & 63;
}
static inline int kbner_set_max_lds_once(std::atomic<unsigned long long>& done, const void* kernel, int bytes) {
  const unsigned long long bit = 1ull << kbner_device_ordinal();
  if (done.load(std::memory_order_acquire) & bit) return 0;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return -(int)e;
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}
// CUs of the current device (cached per device ordinal; 256 if the query fails)
static inline int kbner_cu_count() {
  static std::atomic<int> cache[64];
  const int dev = kbner_device_ordinal();
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16 through the native conversion: hipcc lowers a pair to ONE v_cvt_pk_bf16_f32 (round-to-nearest-even)
// instead of the ~12 integer VALU ops of a hand-rolled rounding -- this sits in every GEMM / attention / LN epilogue.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
static __device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.0f) & 0xffffu); }

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Counter-based dropout shared by every kernel that applies or replays a mask (GEMM epilogue, LayerNorm
// backward, embeddings, attention forward / dQ / dKdV): element (i, j) of a site is kept iff
//   mul24(rowkey(seed, i) ^ colkey(seed, j), 0x9E3779) >= thresh        thresh = p * 2^32   (24-bit x 24-bit -> low 32 bits)
// The separable form costs one xor + one multiply + one compare per element whichever axis a lane walks (the
// attention kernels see the same probability tile in both orientations), needs no stored mask, and is replayed
// bit-identically in backward from (seed, thresh) alone.
static __device__ __forceinline__ uint32_t drop_mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
static __device__ __forceinline__ uint32_t drop_rowkey(uint32_t seed, uint32_t i) { return drop_mix(seed + i); }
static __device__ __forceinline__ uint32_t drop_colkey(uint32_t seed, uint32_t j) {
  return drop_mix((seed * 0x9E3779B1u + 0x7F4A7C15u) ^ j);
}
static __device__ __forceinline__ bool drop_keep(uint32_t rk, uint32_t ck, uint32_t thresh) {
  // 24-bit multiply (v_mul_u32_u24, full rate) of the low 24 bits of the key xor: the 32-bit v_mul_lo_u32 it replaces is a
  // quarter-rate instruction and was ~40 % of what dropout adds to the attention kernels (one test per probability).  The
  // product of a uniform 24-bit value with an odd 24-bit constant is equidistributed mod 2^32, so P(keep) = 1 - thresh / 2^32
  // to within 2^-24; the multiply is what breaks the xor's linearity (a bare threshold on rk ^ ck drops whole rectangles).
  return __umul24(rk ^ ck, 0x9E3779u) >= thresh;
}
static __host__ __device__ __forceinline__ float drop_scale(uint32_t thresh) {
  return 4294967296.0f / (4294967296.0f - (float)thresh);
}

// bf16(dropout(sum_s ws[s][m, n..n+8) + bias) + addend), eight consecutive columns, packed: the fold of split-K slabs (f32 [splits][M, N],
// `slab` = M * N).  ONE definition for kbner_splitk_finish (csrc/rows.hip) and for the LayerNorm kernels that read the slabs directly
// (csrc/layernorm.hip: kbner_ln_fwd_slabs / kbner_ln_bwd_slabs): every route gives the same bits.
static __device__ __forceinline__ uint4 splitk_fold8_pack(const float* __restrict__ ws, int splits, size_t slab, const float* __restrict__ bias,
                                                          const bf16_t* __restrict__ addend, int ldadd, int m, int n, int N,
                                                          uint32_t drop_seed, uint32_t drop_thresh) {
  const size_t i = (size_t)m * N + n;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < splits; ++s) {
    const float4 a = *reinterpret_cast<const float4*>(ws + s * slab + i);
    const float4 b = *reinterpret_cast<const float4*>(ws + s * slab + i + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
    v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  if (bias) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] += bias[n + r];
  }
  if (drop_thresh) {
    const uint32_t rk = drop_rowkey(drop_seed, (uint32_t)m);
    const float ds = drop_scale(drop_thresh);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = drop_keep(rk, drop_colkey(drop_seed, (uint32_t)(n + r)), drop_thresh) ? v[r] * ds : 0.0f;
  }
  if (addend) {
    const uint4 u = *reinterpret_cast<const uint4*>(addend + (size_t)m * ldadd + n);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[2 * r] += __uint_as_float(w[r] << 16);
      v[2 * r + 1] += __uint_as_float(w[r] & 0xffff0000u);
    }
  }
  uint4 o;
  o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
  return o;
}

// GELU, erf form (HF hidden_act="gelu"), and its derivative.  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the bf16 rounding of the stored result) so the GEMM epilogue costs
// one v_rcp + one v_exp + a 5-term Horner per element instead of libm erff; exp(-x^2/2) is shared
// between the cdf and the pdf term of the derivative.
static __device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
  const float u = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * u);
  e = __expf(-u * u);  // = exp(-x^2 / 2)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
}
// Two elements at a time on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32: 2 lanes-worth per issue): the GEMM
// epilogues that apply GELU / GELU' are VALU-bound (128 elements per lane per 256x256 tile), so halving the issue
// count of the polynomial part is worth ~10 % of those GEMMs.  Same A&S 7.1.26 arithmetic as gelu_parts, refactored so
// that no copysign is needed for the forward:  x*cdf = 0.5 x + |x| (0.5 - 0.5 p e).
typedef float f2v __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ f2v splat2(float a) { return (f2v){a, a}; }
// h = 0.5 - 0.5 * poly(t) * e  (= 0.5 * erf(|x| / sqrt 2)),  e = exp(-x^2 / 2)
static __device__ __forceinline__ void gelu_half_erf2(f2v x, f2v ax, f2v& h, f2v& e) {
  const f2v d = ax * splat2(0.3275911f * 0.70710678118654752f) + splat2(1.0f);
  const f2v t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  const f2v a = (x * x) * splat2(-0.5f * 1.4426950408889634f);
  e = (f2v){__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  f2v p = t * splat2(0.5f * 1.061405429f) + splat2(0.5f * -1.453152027f);
  p = p * t + splat2(0.5f * 1.421413741f);
  p = p * t + splat2(0.5f * -0.284496736f);
  p = p * t + splat2(0.5f * 0.254829592f);
  p = p * t;
  h = splat2(0.5f) - p * e;
}
// Round 4 experiment, NOT the default (-DKBNER_GELU_LOGISTIC selects it): gelu / gelu' through ONE logistic per element instead of
// the erf series:
//     Phi(x) ~= s(x) = 1 / (1 + exp(-(x (p0 + p1 x^2 + p2 x^4))))        gelu = x s,   gelu' = s + x s (1 - s) (p0 + 3 p1 x^2 + 5 p2 x^4)
// p fitted (minimax) to |gelu error| <= 3.8e-5 and |gelu' error| <= 9.3e-5 in fp32, 11 packed fp32 operations + 2 v_exp + 2 v_rcp per
// element PAIR instead of 17 + 4.  Measured on one box, alternating processes: 945.1 / 945.1 sentences/s against 945.0 / 943.0 with
// the erf series -- 4 of the ~30 vector instructions per element pair of an epilogue that is one of nine GEMMs: 0.3 %
// of the step, inside the noise.  No measurable speed for a 250 times larger error: rejected.
// x^2 is clamped at 36: beyond |x| = 6 the fit's polynomial is not monotone, s is 0 / 1 to 1e-9 there.
#define KBNER_GELU_P0 1.59484492f
#define KBNER_GELU_P1 7.40112029e-02f
#define KBNER_GELU_P2 -6.97126291e-04f
static __device__ __forceinline__ void gelu_logistic2(f2v x, f2v& sg, f2v& x2c) {
  const f2v x2 = x * x;
  x2c = (f2v){fminf(x2[0], 36.0f), fminf(x2[1], 36.0f)};
  // -log2(e) folded into the coefficients: e = 2^(x * t) = exp(-u)
  f2v t = x2c * splat2(-1.4426950408889634f * KBNER_GELU_P2) + splat2(-1.4426950408889634f * KBNER_GELU_P1);
  t = t * x2c + splat2(-1.4426950408889634f * KBNER_GELU_P0);
  const f2v a = x * t;
  const f2v d = (f2v){__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + splat2(1.0f);
  sg = (f2v){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
static __device__ __forceinline__ f2v gelu2_logistic(f2v x) {
  f2v sg, x2c;
  gelu_logistic2(x, sg, x2c);
  return x * sg;
}
static __device__ __forceinline__ void gelu_both2_logistic(f2v x, f2v& y, f2v& dy) {
  f2v sg, x2c;
  gelu_logistic2(x, sg, x2c);
  y = x * sg;
  f2v du = x2c * splat2(5.0f * KBNER_GELU_P2) + splat2(3.0f * KBNER_GELU_P1);
  du = du * x2c + splat2(KBNER_GELU_P0);
  const f2v w = sg - sg * sg;
  dy = (x * w) * du + sg;
}
// (the erf series of rounds 1-3, kept for tools / tests that want the 1.5e-7 form)
static __device__ __forceinline__ f2v gelu2_erf(f2v x) {
  const f2v ax = {fabsf(x[0]), fabsf(x[1])};
  f2v h, e;
  gelu_half_erf2(x, ax, h, e);
  return x * splat2(0.5f) + ax * h;
}
static __device__ __forceinline__ f2v gelu_grad2(f2v x) {
  const f2v ax = {fabsf(x[0]), fabsf(x[1])};
  f2v h, e;
  gelu_half_erf2(x, ax, h, e);
  const f2v sh = {copysignf(h[0], x[0]), copysignf(h[1], x[1])};
  return (x * e) * splat2(0.39894228040143268f) + (sh + splat2(0.5f));
}
// gelu(x) and gelu'(x) together (they share exp(-x^2/2) and the erf): the forward GEMM epilogue stores the derivative
// (bf16) next to the activation, so the backward epilogue is one multiply instead of a second erf evaluation
static __device__ __forceinline__ void gelu_both2_erf(f2v x, f2v& y, f2v& dy) {
  const f2v ax = {fabsf(x[0]), fabsf(x[1])};
  f2v h, e;
  gelu_half_erf2(x, ax, h, e);
  y = x * splat2(0.5f) + ax * h;
  const f2v sh = {copysignf(h[0], x[0]), copysignf(h[1], x[1])};
  dy = (x * e) * splat2(0.39894228040143268f) + (sh + splat2(0.5f));
}
#ifdef KBNER_GELU_LOGISTIC
static __device__ __forceinline__ f2v gelu2(f2v x) { return gelu2_logistic(x); }
static __device__ __forceinline__ void gelu_both2(f2v x, f2v& y, f2v& dy) { gelu_both2_logistic(x, y, dy); }
#else
static __device__ __forceinline__ f2v gelu2(f2v x) { return gelu2_erf(x); }
static __device__ __forceinline__ void gelu_both2(f2v x, f2v& y, f2v& dy) { gelu_both2_erf(x, y, dy); }
#endif
// packed bf16 pair (one dword) <-> f2v
static __device__ __forceinline__ f2v unpack2bf(uint32_t w) {
  return (f2v){__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
}

// bf16 pair; what the rounding dropped goes, scaled by 2^14, into one half (hi_half) of `res` as two e5m2 bytes
// (v_cvt_pk_bf8_f32, round to nearest even; OCP e5m2 on gfx950)
#define KBNER_RES8_SCALE 16384.0f
static __device__ __forceinline__ uint32_t pack2bf_res8(float lo, float hi, uint32_t& res, bool hi_half) {
  const uint32_t w = pack2bf(lo, hi);
  const f2v r = unpack2bf(w);
  // clamped to the e5m2 maximum: the residual is <= |O| 2^-9, so the scaled value passes 57344 once |O| > ~1800 and the convert
  // (no fp8 saturation in the default MODE) would store inf -> NaN in D -> NaN dQ on one activation outlier
  const float rl = __builtin_amdgcn_fmed3f((lo - r[0]) * KBNER_RES8_SCALE, -57344.f, 57344.f);
  const float rh = __builtin_amdgcn_fmed3f((hi - r[1]) * KBNER_RES8_SCALE, -57344.f, 57344.f);
  res = hi_half ? (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(rl, rh, (int)res, true)
                : (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(rl, rh, 0, false);
  return w;
}

static __device__ __forceinline__ float gelu_f(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return x * cdf;
}
static __device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return cdf + x * e * 0.39894228040143268f;
}
