// Shared device helpers of the attention kernels (attention.hip, attention3.hip): LDS-DMA staging of [S,64] bf16 panels into
// XOR-swizzled 128-byte rows, k-contiguous and transposed MFMA fragment reads, bf16 packing, 4-lane-group reductions.
#pragma once
#include "common.h"

#define AT_D 64
#define AT_MAXS 512
#define AT_NW 8  // wavefronts per workgroup (2 per SIMD: one wave's softmax VALU overlaps the other's MFMAs)
#ifndef AT_NWB
#define AT_NWB 8  // wavefronts per workgroup of the two backward kernels (-DAT_NWB=16, 4 waves per SIMD at 128 VGPRs, measured no faster)
#endif

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_cvoid;

// LDS-DMA from inline asm (hidden from hipcc's waitcnt bookkeeping, see gemm256.hip): completion is
// ordered by hand with counted s_waitcnt vmcnt(N) + a barrier before the panel is read.
static __device__ __forceinline__ void glds16(const void* g, void* l) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)l);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(dst)
               : "memory");
}
// wait until at most n of this wave's DMA instructions are outstanding (n = S/64 per panel: every wave
// issues exactly S/64 one-KiB pieces of each [S,64] panel)
static __device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  }
}
static __device__ __forceinline__ int kc_swz(int row) { return (row >> 1) & 7; }

// DMA a [S rows][64] bf16 panel (row stride ld elements) into a swizzled LDS image (128-B rows)
template <int NW = AT_NW>
static __device__ __forceinline__ void stage_panel(const bf16_t* __restrict__ src, int ld, int S, unsigned char* s, int wid,
                                                   int lane) {
  const int ninstr = S / 8;  // 1 KiB = 8 rows per wave-instruction
  for (int q = wid; q < ninstr; q += NW) {
    const int row = q * 8 + (lane >> 3);
    const int pos = lane & 7;
    glds16(src + (size_t)row * ld + ((pos ^ kc_swz(row)) << 3), s + q * 1024);
  }
}

// 16 rows (r0 + lane&15) x 32 k (ks): k-contiguous fragment, ds_read_b128
static __device__ __forceinline__ bf16x8 kc_frag(const unsigned char* s, int r0, int ks, int lane) {
  const int row = r0 + (lane & 15);
  const int c = ks * 4 + (lane >> 4);
  const s8v v = *reinterpret_cast<const s8v*>(s + row * 128 + ((c ^ kc_swz(row)) << 4));
  return __builtin_bit_cast(bf16x8, v);
}

// transposed fragment: 16 "rows" = panel columns db*16 + (lane&15); k = panel rows of the 32-row
// chunk kc in the split order {g*4+j (j<4), 16+g*4+(j-4)} that matches two stacked 16x16 C tiles
static __device__ __forceinline__ bf16x8 tr_frag(const unsigned char* s, int kc, int db, int lane) {
  const int p = lane & 15;
  const int row = kc * 32 + (lane >> 4) * 4 + (p >> 2);
  const int c = db * 2 + ((p & 3) >> 1);
  const unsigned char* a = s + row * 128 + ((c ^ kc_swz(row)) << 4) + ((p & 1) << 3);
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a + 16 * 128));
  s8v v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// stationary fragment straight from global: row (r0 + lane&15), 8 consecutive d at ks*32 + g*8
static __device__ __forceinline__ bf16x8 glb_frag(const bf16_t* __restrict__ base, int ld, int r0, int ks, int lane) {
  const s8v v = *reinterpret_cast<const s8v*>(base + (size_t)(r0 + (lane & 15)) * ld + ks * 32 + (lane >> 4) * 8);
  return __builtin_bit_cast(bf16x8, v);
}

static __device__ __forceinline__ bf16x8 pack_b(const f4v lo, const f4v hi) {
  union {
    uint32_t u[4];
    bf16x8 v;
  } r;
  r.u[0] = pack2bf(lo[0], lo[1]);
  r.u[1] = pack2bf(lo[2], lo[3]);
  r.u[2] = pack2bf(hi[0], hi[1]);
  r.u[3] = pack2bf(hi[2], hi[3]);
  return r.v;
}

static __device__ __forceinline__ float group4_sum(float v) {  // across the 4 lane groups g (same lane&15)
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
static __device__ __forceinline__ float group4_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

static __device__ __forceinline__ float dot8(const bf16x8 a, const bf16x8 b) {
  const s8v x = __builtin_bit_cast(s8v, a), y = __builtin_bit_cast(s8v, b);
  float acc = 0.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    acc += __uint_as_float(((uint32_t)(uint16_t)x[j]) << 16) * __uint_as_float(((uint32_t)(uint16_t)y[j]) << 16);
  return acc;
}

// ---- the residual of the context, O - bf16(O) (kbner_attn_fwd's ctx_lo; read by the dQ kernels for D = rowdot(dO, O)) ----------
// One e5m2 byte per element (of residual * 2^14).  Layout private to the attention kernels: [b][head][16-row block][lane][16 B],
// where `lane` is the lane of the FORWARD wave that held the values -- its O^T accumulator fragments: lane (g', li) has row li,
// columns db*16 + g'*4 + r of the head's 64 (db, r = 0..3) -> byte db*4 + r of its 16 -- so the forward stores one 16-B vector
// per lane (1 KiB contiguous per wave and block).
static __device__ __forceinline__ uint8_t* at_res_block(uint8_t* base, int b, int A, int h, int S, int q) {
  return base + (((size_t)(b * A + h) * (S >> 4) + (q >> 4)) << 10);
}
static __device__ __forceinline__ const uint8_t* at_res_block(const uint8_t* base, int b, int A, int h, int S, int q) {
  return base + (((size_t)(b * A + h) * (S >> 4) + (q >> 4)) << 10);
}
// this lane's residual words for the two glb_frag fragments of a row (d = ks*32 + g*8 + e): element e of fragment ks is byte e%4 of
// word db = ks*2 + g/2 of forward lane (g' = (g%2)*2 + e/4, li) -> w[0], w[1] = fragment 0 (e 0..3, 4..7), w[2], w[3] = fragment 1
static __device__ __forceinline__ void res_words(const uint8_t* __restrict__ blk, int lane, uint32_t (&w)[4]) {
  const int g = lane >> 4, li = lane & 15;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(blk + ((((g & 1) * 2) * 16 + li) << 4)) + (g >> 1);
  w[0] = p[0];
  w[1] = p[64];   // + 16 lanes
  w[2] = p[2];
  w[3] = p[66];
}
// sum_d dO[row][d] * residual[row][d] over this lane's 16 d
static __device__ __forceinline__ float res_dot16(const bf16x8 do0, const bf16x8 do1, const uint32_t (&w)[4]) {
  const s8v x0 = __builtin_bit_cast(s8v, do0), x1 = __builtin_bit_cast(s8v, do1);
  auto f = [](const s8v& x, int e) { return __uint_as_float(((uint32_t)(uint16_t)x[e]) << 16); };
  float acc = 0.0f;
  f2v r;
  r = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[0], false); acc += f(x0, 0) * r[0] + f(x0, 1) * r[1];
  r = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[0], true);  acc += f(x0, 2) * r[0] + f(x0, 3) * r[1];
  r = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[1], false); acc += f(x0, 4) * r[0] + f(x0, 5) * r[1];
  r = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[1], true);  acc += f(x0, 6) * r[0] + f(x0, 7) * r[1];
  r = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[2], false); acc += f(x1, 0) * r[0] + f(x1, 1) * r[1];
  r = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[2], true);  acc += f(x1, 2) * r[0] + f(x1, 3) * r[1];
  r = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[3], false); acc += f(x1, 4) * r[0] + f(x1, 5) * r[1];
  r = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[3], true);  acc += f(x1, 6) * r[0] + f(x1, 7) * r[1];
  return acc * (1.0f / KBNER_RES8_SCALE);
}

// Hoisted addressing: the XOR swizzle of an LDS row depends only on (row & 15), so every fragment address is a
// per-lane base (computed once per kernel) plus a tile offset that is a multiple of 2048 bytes.
struct PanelBases {
  const unsigned char* kc[2];  // k-contiguous fragment bases for k-step 0 / 1 (row = lane & 15)
  const unsigned char* tr[4];  // transpose-read bases for d-block 0..3 (row = g*4 + (lane&15)>>2)
};
static __device__ __forceinline__ PanelBases panel_bases(const unsigned char* s, int lane) {
  PanelBases b;
  const int g = lane >> 4, li = lane & 15;
  b.kc[0] = s + li * 128 + (((0 * 4 + g) ^ kc_swz(li)) << 4);
  b.kc[1] = s + li * 128 + (((1 * 4 + g) ^ kc_swz(li)) << 4);
  const int vrow = g * 4 + (li >> 2);
#pragma unroll
  for (int db = 0; db < 4; ++db)
    b.tr[db] = s + vrow * 128 + (((db * 2 + ((li & 3) >> 1)) ^ kc_swz(vrow)) << 4) + ((li & 1) << 3);
  return b;
}
static __device__ __forceinline__ bf16x8 kc_at(const unsigned char* base, int off) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const s8v*>(base + off));
}
static __device__ __forceinline__ bf16x8 tr_at(const unsigned char* base, int off) {
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(base + off));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(base + off + 16 * 128));
  s8v v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

