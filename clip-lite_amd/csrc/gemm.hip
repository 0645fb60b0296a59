// Host launchers (C ABI, include/clite.h) for the implicit-GEMM engine in igemm.h.
#include "igemm.h"

using namespace clite;

namespace {

FastDiv fastdiv_make(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) { f.mul = 0; f.shift = 0; f.d = 1; return f; }
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.shift = l;
  f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  return f;
}

ConvGeom geom_dense(int rows, int K) {  // [rows][K] row-major seen as a 1x1 window over a rows x 1 x 1 image
  ConvGeom g;
  g.H = 1; g.W = 1; g.C = K;
  g.sN = K; g.sH = 0; g.sW = 0;
  g.RH = 1; g.RW = 1; g.R = 1; g.S = 1; g.stride = 1; g.pad = 0;
  g.rows = rows;
  g.div_hw = fastdiv_make(1);
  g.div_w = fastdiv_make(1);
  return g;
}

bool fits32(size_t elems, size_t esize) { return elems * esize < 0xF0000000ull; }

typedef TileCfg<128, 128, 32, 64, 64> Cfg128;
typedef TileCfg<256, 64, 32, 64, 64> Cfg256x64;

template <class CFG, class LA, class LB>
int launch(const LA& la, const LB& lb, const clite_epilogue& ep, int M, int N, int Ktot, int splits, hipStream_t st) {
  int ktiles = (Ktot + CFG::BK - 1) / CFG::BK;
  if (splits < 1) splits = 1;
  if (splits > ktiles) splits = ktiles;
  int per = (ktiles + splits - 1) / splits;
  splits = (ktiles + per - 1) / per;
  int tiles = ((M + CFG::BM - 1) / CFG::BM) * ((N + CFG::BN - 1) / CFG::BN);
  hipLaunchKernelGGL((igemm_kernel<CFG, LA, LB>), dim3(tiles, 1, splits), dim3(256), 0, st, la, lb, ep, M, N, ktiles, per);
  return (int)hipGetLastError();
}

int pick_splits(int M, int N, int BM, int BN, int ktiles) {
  long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  long want = 1024 / tiles;      // ~4 workgroups per CU
  if (want < 1) want = 1;
  if (want > ktiles) want = ktiles;
  return (int)want;
}

int check_ep(const clite_epilogue* ep, int N) {
  if (!ep || !ep->out) return -1;
  if (!ep->atomic && (N % 8 != 0 || ep->ldc % 8 != 0)) return -1;
  if (ep->atomic && !ep->out_f32) return -1;
  return 0;
}

}  // namespace

extern "C" int clite_abi_version(void) { return CLITE_ABI_VERSION; }

extern "C" int clite_gemm_nt(const void* A, const void* B, int M, int N, int K, const clite_epilogue* ep, void* stream) {
  if (check_ep(ep, N) || M <= 0 || N <= 0 || K <= 0 || K % 8) return -1;
  if (!fits32((size_t)M * K, 2) || !fits32((size_t)N * K, 2)) return -1;
  hipStream_t st = (hipStream_t)stream;
  int splits = ep->atomic ? pick_splits(M, N, 128, 128, (K + 31) / 32) : 1;
  if (N <= 64) {
    GatherKC<256, 32, false> la{A, (uint32_t)((size_t)M * K * 2), geom_dense(M, K)};
    GatherKC<64, 32, false> lb{B, (uint32_t)((size_t)N * K * 2), geom_dense(N, K)};
    return launch<Cfg256x64>(la, lb, *ep, M, N, K, splits, st);
  }
  GatherKC<128, 32, false> la{A, (uint32_t)((size_t)M * K * 2), geom_dense(M, K)};
  GatherKC<128, 32, false> lb{B, (uint32_t)((size_t)N * K * 2), geom_dense(N, K)};
  return launch<Cfg128>(la, lb, *ep, M, N, K, splits, st);
}

extern "C" int clite_gemm_nn(const void* A, const void* B, int M, int N, int K, const clite_epilogue* ep, void* stream) {
  if (check_ep(ep, N) || M <= 0 || N <= 0 || K <= 0 || K % 8) return -1;
  if (!fits32((size_t)M * K, 2) || !fits32((size_t)N * K, 2)) return -1;
  hipStream_t st = (hipStream_t)stream;
  int splits = ep->atomic ? pick_splits(M, N, 128, 128, (K + 31) / 32) : 1;
  if (N <= 64) {
    GatherKC<256, 32, false> la{A, (uint32_t)((size_t)M * K * 2), geom_dense(M, K)};
    StridedXC<64, 32> lb{B, (uint32_t)((size_t)N * K * 2), N, N, K, 1};
    return launch<Cfg256x64>(la, lb, *ep, M, N, K, splits, st);
  }
  GatherKC<128, 32, false> la{A, (uint32_t)((size_t)M * K * 2), geom_dense(M, K)};
  StridedXC<128, 32> lb{B, (uint32_t)((size_t)N * K * 2), N, N, K, 1};
  return launch<Cfg128>(la, lb, *ep, M, N, K, splits, st);
}

extern "C" int clite_gemm_tn(const void* A, const void* B, int M, int N, int K, const clite_epilogue* ep, void* stream) {
  if (check_ep(ep, N) || M <= 0 || N <= 0 || K <= 0 || M % 8 || N % 8) return -1;
  if (!fits32((size_t)M * K, 2) || !fits32((size_t)N * K, 2)) return -1;
  hipStream_t st = (hipStream_t)stream;
  int splits = ep->atomic ? pick_splits(M, N, 128, 128, (K + 31) / 32) : 1;
  StridedXC<128, 32> la{A, (uint32_t)((size_t)M * K * 2), M, M, K, 1};
  StridedXC<128, 32> lb{B, (uint32_t)((size_t)N * K * 2), N, N, K, 1};
  return launch<Cfg128>(la, lb, *ep, M, N, K, splits, st);
}

// ------------------------------------------------------------------------------------------------ conv
namespace {
int check_conv(const clite_conv* c) {
  if (!c) return -1;
  if (c->C % 8 || c->K % 8) return -1;
  if (c->R * c->S > 1 && (c->C % 32 || c->K % 32)) return -1;   // a K tile must stay inside one (r,s)
  if (c->Ho != (c->H + 2 * c->pad - c->R) / c->stride + 1) return -1;
  if (c->Wo != (c->W + 2 * c->pad - c->S) / c->stride + 1) return -1;
  if (!fits32((size_t)c->N * c->H * c->W * c->C, 2) || !fits32((size_t)c->N * c->Ho * c->Wo * c->K, 2)) return -1;
  return 0;
}
ConvGeom geom_fwd(const clite_conv& c) {   // rows = output pixels, gather x
  ConvGeom g;
  g.H = c.H; g.W = c.W; g.C = c.C;
  g.sN = c.H * c.W * c.C; g.sH = c.W * c.C; g.sW = c.C;
  g.RH = c.Ho; g.RW = c.Wo; g.R = c.R; g.S = c.S; g.stride = c.stride; g.pad = c.pad;
  g.rows = c.N * c.Ho * c.Wo;
  g.div_hw = fastdiv_make(c.Ho * c.Wo);
  g.div_w = fastdiv_make(c.Wo);
  return g;
}
ConvGeom geom_dgrad(const clite_conv& c) {  // rows = input pixels, gather dy
  ConvGeom g;
  g.H = c.Ho; g.W = c.Wo; g.C = c.K;
  g.sN = c.Ho * c.Wo * c.K; g.sH = c.Wo * c.K; g.sW = c.K;
  g.RH = c.H; g.RW = c.W; g.R = c.R; g.S = c.S; g.stride = c.stride; g.pad = c.pad;
  g.rows = c.N * c.H * c.W;
  g.div_hw = fastdiv_make(c.H * c.W);
  g.div_w = fastdiv_make(c.W);
  return g;
}
}  // namespace

extern "C" int clite_conv_fwd(const void* x, const void* w, const clite_conv* cv, const clite_epilogue* ep, void* stream) {
  if (check_conv(cv) || check_ep(ep, cv->K)) return -1;
  const clite_conv& c = *cv;
  hipStream_t st = (hipStream_t)stream;
  int M = c.N * c.Ho * c.Wo, Ktot = c.R * c.S * c.C;
  uint32_t xb = (uint32_t)((size_t)c.N * c.H * c.W * c.C * 2), wb = (uint32_t)((size_t)c.K * Ktot * 2);
  if (c.K <= 64) {
    GatherKC<256, 32, false> la{x, xb, geom_fwd(c)};
    GatherKC<64, 32, false> lb{w, wb, geom_dense(c.K, Ktot)};
    return launch<Cfg256x64>(la, lb, *ep, M, c.K, Ktot, 1, st);
  }
  GatherKC<128, 32, false> la{x, xb, geom_fwd(c)};
  GatherKC<128, 32, false> lb{w, wb, geom_dense(c.K, Ktot)};
  return launch<Cfg128>(la, lb, *ep, M, c.K, Ktot, 1, st);
}

extern "C" int clite_conv_dgrad(const void* dy, const void* w, const clite_conv* cv, const clite_epilogue* ep, void* stream) {
  if (check_conv(cv) || check_ep(ep, cv->C)) return -1;
  const clite_conv& c = *cv;
  hipStream_t st = (hipStream_t)stream;
  int M = c.N * c.H * c.W, Ktot = c.R * c.S * c.K;
  uint32_t yb = (uint32_t)((size_t)c.N * c.Ho * c.Wo * c.K * 2), wb = (uint32_t)((size_t)c.K * c.R * c.S * c.C * 2);
  if (c.C <= 64) {
    GatherKC<256, 32, true> la{dy, yb, geom_dgrad(c)};
    StridedXC<64, 32> lb{w, wb, c.R * c.S * c.C, c.C, c.K, c.R * c.S};
    return launch<Cfg256x64>(la, lb, *ep, M, c.C, Ktot, 1, st);
  }
  GatherKC<128, 32, true> la{dy, yb, geom_dgrad(c)};
  StridedXC<128, 32> lb{w, wb, c.R * c.S * c.C, c.C, c.K, c.R * c.S};
  return launch<Cfg128>(la, lb, *ep, M, c.C, Ktot, 1, st);
}

extern "C" int clite_conv_wgrad(const void* dy, const void* x, const clite_conv* cv, float* dw, void* stream) {
  if (check_conv(cv) || !dw) return -1;
  const clite_conv& c = *cv;
  hipStream_t st = (hipStream_t)stream;
  int P = c.N * c.Ho * c.Wo, Ncols = c.R * c.S * c.C;
  uint32_t yb = (uint32_t)((size_t)P * c.K * 2), xb = (uint32_t)((size_t)c.N * c.H * c.W * c.C * 2);
  clite_epilogue ep = {};
  ep.out = dw; ep.ldc = Ncols; ep.out_f32 = 1; ep.atomic = 1; ep.alpha = 1.f;
  StridedXC<128, 32> la{dy, yb, c.K, c.K, P, 1};
  GatherXC<128, 32> lb{x, xb, geom_fwd(c)};
  int splits = pick_splits(c.K, Ncols, 128, 128, (P + 31) / 32);
  return launch<Cfg128>(la, lb, ep, c.K, Ncols, P, splits, st);
}
