// Attention lab: a torch-free executable that checks the attention kernels of libkbner_hip.so against naive fp32 GPU
// references and times them (HIP events), so one short gpurun call can A/B several kernel variants.
//   build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -I kb-ner_amd/csrc -I include tools/micro/attn_lab.hip \
//            -Lkb-ner_amd/kbner -lkbner_hip -Wl,-rpath,'$ORIGIN/../../kb-ner_amd/kbner' -o tools/micro/attn_lab
//   run:   tools/micro/attn_lab [Bcheck] [Btime] [S] [reps]         (env KBNER_ATTN selects the library's variant)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint16_t bf16_t;
extern "C" {
int kbner_attn_fwd(const bf16_t* qkv, const float* maskbias, bf16_t* ctx, bf16_t* ctx_lo, float* lse, int B, int S, int H, int A, uint32_t drop_seed,
                   uint32_t drop_thresh, void* stream);
int kbner_attn_bwd(const bf16_t* qkv, const bf16_t* ctx, const bf16_t* ctx_lo, const bf16_t* dctx, const float* maskbias, const float* lse, float* Dws,
                   bf16_t* dqkv, int B, int S, int H, int A, uint32_t drop_seed, uint32_t drop_thresh, float* dbias_qkv, void* stream);
int kbner_dropout_mask(float* out, int Z, int M, int N, uint32_t seed, uint32_t thresh, void* stream);
}

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                             \
    }                                                                                      \
  } while (0)

static inline float bf2f_h(bf16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline bf16_t f2bf_h(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ inline float bf2f_d(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// ---- naive references (fp32, one thread per output element / row) -------------------------------------------------------
// P[bh][q][k] (softmax, undropped), lse[bh][q]
__global__ void ref_probs(const bf16_t* qkv, const float* mb, float* P, float* lse, int S, int H, int A) {
  const int q = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  extern __shared__ float sh[];
  float* sc = sh;
  const int ld = 3 * H;
  const bf16_t* qr = qkv + (size_t)(b * S + q) * ld + h * 64;
  float mx = -INFINITY;
  for (int k = threadIdx.x; k < S; k += blockDim.x) {
    const bf16_t* kr = qkv + (size_t)(b * S + k) * ld + H + h * 64;
    float a = 0.f;
    for (int d = 0; d < 64; ++d) a += bf2f_d(qr[d]) * bf2f_d(kr[d]);
    a = a * 0.125f + mb[b * S + k];
    sc[k] = a;
    mx = fmaxf(mx, a);
  }
  __shared__ float red[256];
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float s = 0.f;
  for (int k = threadIdx.x; k < S; k += blockDim.x) {
    const float e = expf(sc[k] - mx);
    sc[k] = e;
    s += e;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  s = red[0];
  float* pr = P + ((size_t)(b * A + h) * S + q) * S;
  for (int k = threadIdx.x; k < S; k += blockDim.x) pr[k] = sc[k] / s;
  if (threadIdx.x == 0) lse[(size_t)(b * A + h) * S + q] = mx + logf(s);
}
// O[b,q,h,d] = sum_k (P * DM)[q,k] V[k,d]   (DM: dropout multiplier or null)
__global__ void ref_pv(const bf16_t* qkv, const float* P, const float* DM, float* O, int S, int H, int A) {
  const int q = blockIdx.x, h = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
  const int ld = 3 * H;
  const size_t pr = ((size_t)(b * A + h) * S + q) * S;
  float a = 0.f;
  for (int k = 0; k < S; ++k) a += P[pr + k] * (DM ? DM[pr + k] : 1.0f) * bf2f_d(qkv[(size_t)(b * S + k) * ld + 2 * H + h * 64 + d]);
  O[(size_t)(b * S + q) * H + h * 64 + d] = a;
}
// dS[q,k] = P * (DM * dP - D) where dP[q,k] = dO[q].V[k], D[q] = sum_k P DM dP   (written over a copy of P) ; PD = P * DM
__global__ void ref_ds(const bf16_t* qkv, const bf16_t* dO, const float* P, const float* DM, float* dS, float* PD, int S, int H, int A) {
  const int q = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int ld = 3 * H;
  const size_t pr = ((size_t)(b * A + h) * S + q) * S;
  extern __shared__ float sh[];
  __shared__ float red[256];
  float part = 0.f;
  for (int k = threadIdx.x; k < S; k += blockDim.x) {
    float a = 0.f;
    for (int d = 0; d < 64; ++d) a += bf2f_d(dO[(size_t)(b * S + q) * H + h * 64 + d]) * bf2f_d(qkv[(size_t)(b * S + k) * ld + 2 * H + h * 64 + d]);
    const float dm = DM ? DM[pr + k] : 1.0f;
    sh[k] = a * dm;
    part += P[pr + k] * a * dm;
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float D = red[0];
  for (int k = threadIdx.x; k < S; k += blockDim.x) {
    dS[pr + k] = P[pr + k] * (sh[k] - D);
    PD[pr + k] = P[pr + k] * (DM ? DM[pr + k] : 1.0f);
  }
}
// dQ[q,d] = scale sum_k dS[q,k] K[k,d] ; dK[k,d] = scale sum_q dS[q,k] Q[q,d] ; dV[k,d] = sum_q PD[q,k] dO[q,d]
__global__ void ref_dqkv(const bf16_t* qkv, const bf16_t* dO, const float* dS, const float* PD, float* dqkv, int S, int H, int A) {
  const int r = blockIdx.x, h = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
  const int ld = 3 * H;
  const size_t hb = (size_t)(b * A + h) * S * S;
  float aq = 0.f, ak = 0.f, av = 0.f;
  for (int t = 0; t < S; ++t) {
    aq += dS[hb + (size_t)r * S + t] * bf2f_d(qkv[(size_t)(b * S + t) * ld + H + h * 64 + d]);
    ak += dS[hb + (size_t)t * S + r] * bf2f_d(qkv[(size_t)(b * S + t) * ld + h * 64 + d]);
    av += PD[hb + (size_t)t * S + r] * bf2f_d(dO[(size_t)(b * S + t) * H + h * 64 + d]);
  }
  float* o = dqkv + (size_t)(b * S + r) * ld + h * 64 + d;
  o[0] = aq * 0.125f;
  o[H] = ak * 0.125f;
  o[2 * H] = av;
}

static double rel_l2(const std::vector<float>& a, const std::vector<float>& b, size_t off, size_t n, size_t stride, size_t rows) {
  double num = 0, den = 0;
  for (size_t r = 0; r < rows; ++r)
    for (size_t i = 0; i < n; ++i) {
      const double x = a[r * stride + off + i], y = b[r * stride + off + i];
      num += (x - y) * (x - y);
      den += y * y;
    }
  return std::sqrt(num / (den + 1e-300));
}

static uint32_t rng_state = 12345u;
static float frand() {   // uniform in [-1, 1)
  rng_state = rng_state * 1664525u + 1013904223u;
  return (float)((rng_state >> 8) & 0xffffff) / 8388608.0f - 1.0f;
}
static float grand() {   // ~N(0,1): sum of 4 uniforms
  return (frand() + frand() + frand() + frand()) * 0.8660254f;
}

struct Bufs {
  bf16_t *qkv, *dctx, *ctx, *ctx_lo, *dqkv;   // ctx_lo: LAB_RESIDUAL=1 allocates it (null = D from the bf16 O alone)
  float *mb, *lse, *dws, *dbias;
};

static void fill(int B, int S, int H, std::vector<bf16_t>& qkv, std::vector<bf16_t>& dctx, std::vector<float>& mb, int ragged, int spike) {
  qkv.resize((size_t)B * S * 3 * H);
  dctx.resize((size_t)B * S * H);
  mb.assign((size_t)B * S, 0.0f);
  for (auto& x : qkv) x = f2bf_h(grand());
  for (auto& x : dctx) x = f2bf_h(grand());
  if (ragged)
    for (int b = 0; b < B; ++b) {
      int len = S;
      if (ragged == 1) len = S - ((b * 37) % (S / 2));      // prefix masks of assorted lengths
      if (ragged == 2) len = (b % 3 == 0) ? S : 1 + (b * 53) % S;
      for (int i = len; i < S; ++i) mb[(size_t)b * S + i] = -10000.0f;
      if (ragged == 3 && b % 2) {   // non-prefix mask: holes
        for (int i = 0; i < S; ++i) mb[(size_t)b * S + i] = (i % 7 == 3) ? -10000.0f : 0.0f;
      }
    }
  if (spike) {
    // forces the lazy-rescale branch: in every head of sentence 0, key `spike` is 6x key-like for queries 5 and 40
    // (raw dot ~ 6 * |q|^2 ~ 6 * 64 -> scaled ~ 48 above the other scores)
    for (int h = 0; h < H / 64; ++h)
      for (int d = 0; d < 64; ++d) {
        const float q5 = bf2f_h(qkv[(size_t)5 * 3 * H + h * 64 + d]);
        qkv[(size_t)spike * 3 * H + H + h * 64 + d] = f2bf_h(6.0f * q5);
      }
  }
}

int main(int argc, char** argv) {
  const int Bc = argc > 1 ? atoi(argv[1]) : 4;
  const int Bt = argc > 2 ? atoi(argv[2]) : 128;
  const int S = argc > 3 ? atoi(argv[3]) : 512;
  const int reps = argc > 4 ? atoi(argv[4]) : 10;
  const int A = 16, H = A * 64;
  const int do_bwd = getenv("LAB_NO_BWD") ? 0 : 1;
  int rc = 0;
  // ---------------- correctness ----------------
  for (int cas = 0; cas < 6 && Bc > 0; ++cas) {
    const int ragged = cas == 0 ? 0 : (cas == 1 ? 1 : (cas == 2 ? 2 : (cas == 3 ? 3 : (cas == 4 ? 1 : 0))));
    const int spike = cas == 4 ? 300 % S : (cas == 5 ? (S - 1) : 0);
    const uint32_t dthr = (cas == 2 || cas == 5) ? 429496730u : 0u;   // p = 0.1
    const uint32_t dseed = 424242u + cas;
    const int B = Bc;
    std::vector<bf16_t> qkv, dctx;
    std::vector<float> mb;
    rng_state = 777u + cas;
    fill(B, S, H, qkv, dctx, mb, ragged, spike);
    Bufs d;
    const size_t nq = (size_t)B * S * 3 * H, nc = (size_t)B * S * H, nl = (size_t)B * A * S, np = nl * S;
    CK(hipMalloc(&d.qkv, nq * 2)); CK(hipMalloc(&d.dctx, nc * 2)); CK(hipMalloc(&d.ctx, nc * 2)); CK(hipMalloc(&d.dqkv, nq * 2));
    d.ctx_lo = nullptr;
    if (getenv("LAB_RESIDUAL")) CK(hipMalloc(&d.ctx_lo, nc * 2));
    CK(hipMalloc(&d.mb, (size_t)B * S * 4)); CK(hipMalloc(&d.lse, nl * 4)); CK(hipMalloc(&d.dws, nl * 4)); CK(hipMalloc(&d.dbias, 3 * H * 4));
    CK(hipMemcpy(d.qkv, qkv.data(), nq * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.dctx, dctx.data(), nc * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.mb, mb.data(), (size_t)B * S * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d.ctx, 0xff, nc * 2)); CK(hipMemset(d.dqkv, 0xff, nq * 2)); CK(hipMemset(d.lse, 0xff, nl * 4));
    CK(hipMemset(d.dbias, 0, 3 * H * 4));
    float *P, *DM = nullptr, *Oref, *lref, *dSr, *PD, *dref;
    CK(hipMalloc(&P, np * 4)); CK(hipMalloc(&Oref, nc * 4)); CK(hipMalloc(&lref, nl * 4)); CK(hipMalloc(&dSr, np * 4));
    CK(hipMalloc(&PD, np * 4)); CK(hipMalloc(&dref, nq * 4));
    if (dthr) {
      CK(hipMalloc(&DM, np * 4));
      int r = kbner_dropout_mask(DM, B * A, S, S, dseed, dthr, nullptr);
      if (r) { fprintf(stderr, "dropout_mask rc=%d\n", r); return 3; }
    }
    hipLaunchKernelGGL(ref_probs, dim3(S, A, B), dim3(256), S * 4, 0, d.qkv, d.mb, P, lref, S, H, A);
    hipLaunchKernelGGL(ref_pv, dim3(S, A, B), dim3(64), 0, 0, d.qkv, P, DM, Oref, S, H, A);
    int r1 = kbner_attn_fwd(d.qkv, d.mb, d.ctx, d.ctx_lo, d.lse, B, S, H, A, dseed, dthr, nullptr);
    CK(hipDeviceSynchronize());
    std::vector<float> o_ref(nc), l_ref(nl), l_got(nl), o_got(nc);
    std::vector<bf16_t> o_bf(nc);
    CK(hipMemcpy(o_ref.data(), Oref, nc * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(l_ref.data(), lref, nl * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(l_got.data(), d.lse, nl * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o_bf.data(), d.ctx, nc * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nc; ++i) o_got[i] = bf2f_h(o_bf[i]);
    const double e_ctx = rel_l2(o_got, o_ref, 0, nc, 0, 1);
    double e_lse = 0, e_max = 0;
    for (size_t i = 0; i < nl; ++i) e_lse = std::fmax(e_lse, std::fabs((double)l_got[i] - l_ref[i]));
    for (size_t i = 0; i < nc; ++i) e_max = std::fmax(e_max, std::fabs((double)o_got[i] - o_ref[i]));
    int nan_ctx = 0;
    for (size_t i = 0; i < nc; ++i) nan_ctx += !(o_got[i] == o_got[i]);
    printf("case %d (ragged %d spike %d drop %d) fwd rc=%d: ctx rel_l2 %.3e max_abs %.3e lse max_abs %.3e nan %d", cas, ragged, spike,
           dthr ? 1 : 0, r1, e_ctx, e_max, e_lse, nan_ctx);
    const bool ok_f = r1 == 0 && e_ctx < 1.0e-2 && e_lse < 5e-3 && nan_ctx == 0;
    if (!ok_f) rc = 1;
    if (do_bwd) {
      hipLaunchKernelGGL(ref_ds, dim3(S, A, B), dim3(256), S * 4, 0, d.qkv, d.dctx, P, DM, dSr, PD, S, H, A);
      hipLaunchKernelGGL(ref_dqkv, dim3(S, A, B), dim3(64), 0, 0, d.qkv, d.dctx, dSr, PD, dref, S, H, A);
      int r2 = kbner_attn_bwd(d.qkv, d.ctx, d.ctx_lo, d.dctx, d.mb, d.lse, d.dws, d.dqkv, B, S, H, A, dseed, dthr, d.dbias, nullptr);
      CK(hipDeviceSynchronize());
      std::vector<float> g_ref(nq), g_got(nq), db(3 * H);
      std::vector<bf16_t> g_bf(nq);
      CK(hipMemcpy(g_ref.data(), dref, nq * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(g_bf.data(), d.dqkv, nq * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(db.data(), d.dbias, 3 * H * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < nq; ++i) g_got[i] = bf2f_h(g_bf[i]);
      const size_t rows = (size_t)B * S;
      const double eq = rel_l2(g_got, g_ref, 0, H, 3 * H, rows), ek = rel_l2(g_got, g_ref, H, H, 3 * H, rows),
                   ev = rel_l2(g_got, g_ref, 2 * H, H, 3 * H, rows);
      double eb = 0, bmax = 0;
      for (int c = 0; c < 3 * H; ++c) {
        double s = 0;
        for (size_t r = 0; r < rows; ++r) s += g_ref[r * 3 * H + c];
        eb = std::fmax(eb, std::fabs(s - db[c]));
        bmax = std::fmax(bmax, std::fabs(s));
      }
      printf(" | bwd rc=%d: dq %.3e dk %.3e dv %.3e dbias %.3e", r2, eq, ek, ev, eb / (bmax + 1e-30));
      if (!(r2 == 0 && eq < 3e-2 && ek < 3e-2 && ev < 3e-2 && eb / (bmax + 1e-30) < 2e-2)) rc = 1;
    }
    printf(" %s\n", rc ? "FAIL-SO-FAR" : "ok");
    hipFree(P); hipFree(Oref); hipFree(lref); hipFree(dSr); hipFree(PD); hipFree(dref);
    if (DM) hipFree(DM);
    hipFree(d.qkv); hipFree(d.dctx); hipFree(d.ctx); hipFree(d.ctx_lo); hipFree(d.dqkv); hipFree(d.mb); hipFree(d.lse); hipFree(d.dws); hipFree(d.dbias);
  }
  // ---------------- timing ----------------
  if (Bt > 0) {
    const int B = Bt;
    std::vector<bf16_t> qkv, dctx;
    std::vector<float> mb;
    rng_state = 999u;
    const char* rl = getenv("LAB_REAL_LEN");
    fill(B, S, H, qkv, dctx, mb, 0, 0);
    if (rl)
      for (int b = 0; b < B; ++b)
        for (int i = atoi(rl); i < S; ++i) mb[(size_t)b * S + i] = -10000.0f;
    Bufs d;
    const size_t nq = (size_t)B * S * 3 * H, nc = (size_t)B * S * H, nl = (size_t)B * A * S;
    CK(hipMalloc(&d.qkv, nq * 2)); CK(hipMalloc(&d.dctx, nc * 2)); CK(hipMalloc(&d.ctx, nc * 2)); CK(hipMalloc(&d.dqkv, nq * 2));
    d.ctx_lo = nullptr;
    if (getenv("LAB_RESIDUAL")) CK(hipMalloc(&d.ctx_lo, nc * 2));
    CK(hipMalloc(&d.mb, (size_t)B * S * 4)); CK(hipMalloc(&d.lse, nl * 4)); CK(hipMalloc(&d.dws, nl * 4)); CK(hipMalloc(&d.dbias, 3 * H * 4));
    CK(hipMemcpy(d.qkv, qkv.data(), nq * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.dctx, dctx.data(), nc * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.mb, mb.data(), (size_t)B * S * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d.dbias, 0, 3 * H * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double f = 4.0 * S * S * 64 * A * B;
    for (int drop = 0; drop < 2; ++drop) {
      const uint32_t thr = drop ? 429496730u : 0u;
      for (int i = 0; i < 3; ++i) kbner_attn_fwd(d.qkv, d.mb, d.ctx, d.ctx_lo, d.lse, B, S, H, A, 1u, thr, nullptr);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) kbner_attn_fwd(d.qkv, d.mb, d.ctx, d.ctx_lo, d.lse, B, S, H, A, 1u, thr, nullptr);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      printf("time B=%d S=%d drop=%d  fwd %8.1f us  %7.1f TFLOP/s", B, S, drop, ms * 1e3, f / ms / 1e9);
      if (do_bwd) {
        for (int i = 0; i < 3; ++i) kbner_attn_bwd(d.qkv, d.ctx, d.ctx_lo, d.dctx, d.mb, d.lse, d.dws, d.dqkv, B, S, H, A, 1u, thr, d.dbias, nullptr);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) kbner_attn_bwd(d.qkv, d.ctx, d.ctx_lo, d.dctx, d.mb, d.lse, d.dws, d.dqkv, B, S, H, A, 1u, thr, d.dbias, nullptr);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("  | bwd %8.1f us  %7.1f TFLOP/s (5 algorithmic matmuls)", ms * 1e3, 2.5 * f / ms / 1e9);
      }
      printf("\n");
    }
  }
  printf("LAB %s\n", rc ? "FAIL" : "PASS");
  return rc;
}
