// Host side of the wide-K 8-wave kernels (igemm_wide.h): tile-shape selection and the instantiation table.
#include "igemm_wide.h"
#include "wide_api.h"
#include <atomic>

using namespace clite;

static std::atomic<int> g_tile_policy{0};

namespace {

typedef WideCfg<128, 128, 64, 64, 2> W128;        // 2 x 2 waves x 2 k-groups, 3-stage ring (96 KB)
typedef WideCfg<256, 128, 64, 64, 1> W256x128;    // 4 x 2 waves, 3-stage ring (144 KB)
typedef WideCfg<256, 256, 128, 64, 1> W256;       // 2 x 4 waves of 128 x 64, 2-stage ring (128 KB)

template <class CFG> struct Stages { static constexpr int n = 3; };
template <> struct Stages<W256> { static constexpr int n = 2; };
#ifdef CLITE_W128_STAGES          // A/B builds only: the 128 x 128 wide tile's ring depth (3 stages = 96 KB of LDS, 2 = 64 KB: what fits beside another kernel's workgroup on a CU)
template <> struct Stages<W128> { static constexpr int n = CLITE_W128_STAGES; };
#endif

template <int ROWS> WideKC<ROWS, false> mk_kc(const WideOperand& o) { return WideKC<ROWS, false>{o.ptr, o.bytes, o.g}; }
template <int ROWS> WideKC<ROWS, true> mk_kcd(const WideOperand& o) { return WideKC<ROWS, true>{o.ptr, o.bytes, o.g}; }
template <int COLS> DmaXCStrided<bf16, COLS, WIDE_BK, WIDE_NW> mk_xs(const WideOperand& o) {
  return DmaXCStrided<bf16, COLS, WIDE_BK, WIDE_NW>{o.ptr, o.bytes, o.ld, o.Cx, o.Ck, o.RS};
}

struct Grid { dim3 grid; int ktiles, per, xsplits; };

template <class CFG>
Grid plan(int M, int N, int Ktot, int splits) {
  Grid p;
  p.ktiles = (Ktot + WIDE_BK - 1) / WIDE_BK;
  if (splits < 1) splits = 1;
  if (splits > p.ktiles) splits = p.ktiles;
  p.per = (p.ktiles + splits - 1) / splits;
  splits = (p.ktiles + p.per - 1) / p.per;
  const int tiles = ((M + CFG::BM - 1) / CFG::BM) * ((N + CFG::BN - 1) / CFG::BN);
  p.xsplits = splits >= 8 ? splits : 0;          // >= 8 K splits: each XCD owns whole splits (igemm_dma.h)
  p.grid = p.xsplits ? dim3(8 * ((splits + 7) / 8) * tiles, 1, 1) : dim3(tiles, 1, splits);
  return p;
}

template <class CFG, int EPI, class LA, class LB>
int go(const LA& la, const LB& lb, const clite_epilogue& ep, const RowMap& rm, int M, int N, int Ktot, int splits, hipStream_t st) {
  const Grid p = plan<CFG>(M, N, Ktot, splits);
  hipLaunchKernelGGL((igemm_wide_kernel<CFG, LA, LB, Stages<CFG>::n, EPI>), p.grid, dim3(512), 0, st, la, lb, ep, rm, M, N, p.ktiles, p.per, p.xsplits);
  return (int)hipGetLastError();
}

bool is_plain(const clite_epilogue& ep) {
  return !ep.atomic && !ep.out_f32 && !ep.preact && ep.act == CLITE_ACT_NONE && !ep.dact_aux && ep.drop_p <= 0.f && !ep.residual && !ep.bn_y && !ep.mask_after_residual;
}

// The compile-time epilogue forms of BERT's linears (igemm_wide.h: WideEpiForm 3 - 6), recognised from the run-time description; 0 = none of them.
int bert_form(const clite_epilogue& ep) {
#ifdef CLITE_NO_BERT_FORMS          // A/B builds only (make variant VAR_EXTRA=-DCLITE_NO_BERT_FORMS): everything through the run-time-flag form
  return 0;
#endif
  if (ep.atomic || ep.out_f32 || ep.bn_y || ep.mask_after_residual || ep.relu_bits || ep.splitk_ws) return 0;
  const bool drop = ep.drop_p > 0.f, aux = ep.dact_aux != nullptr;
  if (ep.bias && ep.preact && ep.act == CLITE_ACT_GELU && !aux && !drop && !ep.residual && !ep.colsum) return 3;
  if (!ep.bias && !ep.preact && ep.act == CLITE_ACT_NONE && aux && ep.dact == 2 && !drop && !ep.residual && (!ep.colsum || ep.colsum_rows == 1)) return 4;
  if (ep.bias && !ep.preact && ep.act == CLITE_ACT_NONE && !aux && drop && ep.residual && !ep.colsum) return 5;
  if (!ep.bias && !ep.preact && ep.act == CLITE_ACT_NONE && !aux && !drop && ep.residual && !ep.colsum) return 6;
  return 0;
}

// forward-type operand pairs (generic / plain epilogues)
template <class CFG, class LA, class LB>
int go_fwd(const LA& la, const LB& lb, const clite_epilogue& ep, const RowMap& rm, int M, int N, int Ktot, int splits, hipStream_t st) {
  if (is_plain(ep)) return go<CFG, 2>(la, lb, ep, rm, M, N, Ktot, splits, st);
  return go<CFG, 0>(la, lb, ep, rm, M, N, Ktot, splits, st);
}
// k-contiguous x k-contiguous operands (every BERT linear, forward and — on the transposed weight copies — input gradient): + the four
// specialised forms
template <class CFG, class LA, class LB>
int go_linear(const LA& la, const LB& lb, const clite_epilogue& ep, const RowMap& rm, int M, int N, int Ktot, int splits, hipStream_t st) {
  if (splits == 1 && !rm.on) {
    switch (bert_form(ep)) {
      case 3: return go<CFG, 3>(la, lb, ep, rm, M, N, Ktot, 1, st);
      case 4: return go<CFG, 4>(la, lb, ep, rm, M, N, Ktot, 1, st);
      case 5: return go<CFG, 5>(la, lb, ep, rm, M, N, Ktot, 1, st);
      case 6: return go<CFG, 6>(la, lb, ep, rm, M, N, Ktot, 1, st);
      default: break;
    }
  }
  return go_fwd<CFG>(la, lb, ep, rm, M, N, Ktot, splits, st);
}

template <class CFG>
int dispatch(const WideOperand& a, const WideOperand& b, const clite_epilogue& ep, const RowMap& rm, int M, int N, int Ktot, int splits, hipStream_t st) {
  constexpr int BM = CFG::BM, BN = CFG::BN;
  const bool bn = ep.bn_y || ep.mask_after_residual;
  if (bn) {
    // BatchNorm-backward epilogue: out = [(alpha*acc) (* relu'(aux))] (+ residual) [(* relu'(aux))], statistics; conv dgrad operands only
    // (ep.bias: the constant row of the folded BatchNorm backward, added before the mask — clite_conv_dgrad_bnfold)
    if (ep.atomic || ep.act || ep.preact || ep.drop_p > 0.f || (ep.dact_aux && ep.dact != 1) || splits != 1) return -1;
    if (a.kind == WOP_KC_DGRAD && b.kind == WOP_XC_STRIDED) return go<CFG, 1>(mk_kcd<BM>(a), mk_xs<BN>(b), ep, rm, M, N, Ktot, 1, st);
    if (a.kind == WOP_KC_DGRAD && b.kind == WOP_KC) return go<CFG, 1>(mk_kcd<BM>(a), mk_kc<BN>(b), ep, rm, M, N, Ktot, 1, st);      // transposed weights
    return -1;
  }
  if (a.kind == WOP_KC && b.kind == WOP_KC) return go_linear<CFG>(mk_kc<BM>(a), mk_kc<BN>(b), ep, rm, M, N, Ktot, splits, st);
  if (a.kind == WOP_KC && b.kind == WOP_XC_STRIDED) return go_fwd<CFG>(mk_kc<BM>(a), mk_xs<BN>(b), ep, rm, M, N, Ktot, splits, st);
  if (a.kind == WOP_KC_DGRAD && b.kind == WOP_XC_STRIDED) return go_fwd<CFG>(mk_kcd<BM>(a), mk_xs<BN>(b), ep, rm, M, N, Ktot, splits, st);
  if (a.kind == WOP_KC_DGRAD && b.kind == WOP_KC) return go_fwd<CFG>(mk_kcd<BM>(a), mk_kc<BN>(b), ep, rm, M, N, Ktot, splits, st);
  return WIDE_NOT_TAKEN;
}

// a K tile of 64 must stay inside one window position (r, s) / one k-plane
bool operand_ok(const WideOperand& o) {
  switch (o.kind) {
    case WOP_KC: case WOP_KC_DGRAD: return o.g.R * o.g.S == 1 || o.g.C % WIDE_BK == 0;
    case WOP_XC_STRIDED: return o.RS == 1 || o.Ck % WIDE_BK == 0;
    default: return true;
  }
}

}  // namespace

// policy: 0 automatic; 1 / 2 / 3 force the 128 x 128 / 256 x 128 / 256 x 256 wide tile wherever an instantiation exists; 4 keeps every
// launch on the 4-wave 128 x 128 x 32 kernels
extern "C" int clite_set_tile_policy(int policy) {
  if (policy < 0 || policy > 4) return -1;
  g_tile_policy.store(policy, std::memory_order_relaxed);
  return 0;
}

int clite::tile_policy_value() { return g_tile_policy.load(std::memory_order_relaxed); }

bool clite_group_wide_enabled() { return g_tile_policy.load(std::memory_order_relaxed) != 4; }

int clite::launch_wide(const WideOperand& a, const WideOperand& b, const clite_epilogue& ep, const RowMap& rm, int M, int N, int Ktot, int splits, hipStream_t st) {
  const int policy = g_tile_policy.load(std::memory_order_relaxed);
  if (policy == 4 || !operand_ok(a) || !operand_ok(b) || rm.on == 2) return WIDE_NOT_TAKEN;          // (the subsampled-residual form: 4-wave kernels only)
  // weight gradients (XC x XC operands, f32 atomic split-K accumulation) have no wide instantiation: measured slower (61 vs 49 us on the
  // BERT 2304 x 768 x 3840 gradient) — many small workgroups hide the atomic epilogue better
  if (a.kind == WOP_XC_STRIDED || ep.atomic) return WIDE_NOT_TAKEN;
  const long tm128 = (M + 127) / 128, tm256 = (M + 255) / 256, tn128 = (N + 127) / 128, tn256 = (N + 255) / 256;
  int pick = policy;
  if (pick == 0) {
    // Measured per launch on MI355X against the 4-wave kernels (tools/probe_tiles.py, profiles/r2_tile_policy_probe.txt):
    //  * weight gradients (f32 atomic accumulation, split K) and K < 512 stay on the 4-wave kernels: a short K loop is all prologue +
    //    epilogue, and three small workgroups per CU overlap those where one 8-wave workgroup cannot (1x1 64 -> 256 @56: 60 vs 79-147 us)
    //  * at most one round of 128 x 128 tiles (the BERT N = 768 GEMMs, the 7 x 7-resolution convs): 128 x 128 wide — whole-line K slabs,
    //    two k-groups per tile (FFN2 36.3 -> 30.8 us, 512 -> 512 3x3 dgrad 66.4 -> 58.7)
    //  * N >= 1024 with >= 128 tiles of 256 x 256 (FFN1 / QKV, the 512 <-> 2048 1x1 convs): 256 x 256 (28.8 -> 24.8, 29.9 -> 24.2)
    //  * windowed convs and K >= 1024 on larger grids: 256 x 128 (256 -> 256 3x3 44.4 -> 40.8, 1024 -> 256 1x1 27.7 -> 23.3)
    const bool window = (a.kind == WOP_KC || a.kind == WOP_KC_DGRAD) && a.g.R * a.g.S > 1 && !a.g.concat;
    if (Ktot < 512) return WIDE_NOT_TAKEN;
    //  * inside the step (tools/layer_profile.py, profiles/r2_layers_*.txt) one 8-wave workgroup per CU exposes its epilogue — nothing else
    //    on the CU computes meanwhile — so launches whose epilogue is VALU-heavy (erf-GELU forward / derivative over a 3840 x 3072 tile set:
    //    ~14 us of VALU) or reads three extra tensors (the BatchNorm-backward form) go wide only when the K loop is long enough to pay for it
    //    (round 4: the two GELU launches of BERT's FFN have compile-time epilogue forms now — WideEpiForm 3 / 4 — and run the 256 x 256 tile faster
    //    than the 4-wave kernel: forward 41.1 vs 49.0 us, input gradient with the bias sums 44.2 vs 59.4; other GELU combinations keep the rule)
    const bool gelu = (ep.act == CLITE_ACT_GELU || (ep.dact_aux && ep.dact == 2)) && !(a.kind == WOP_KC && b.kind == WOP_KC && splits == 1 && bert_form(ep) != 0);
    const bool bn = ep.bn_y || ep.mask_after_residual;
    if (tm128 * tn128 <= 256) pick = 1;
    //    (round 3, against the row-range persistent 4-wave kernel with its specialised epilogue, cold caches: 1x1 256 <- 1024 @14 45.2 narrow / 41.5
    //    wide; windowed 128 <- 128 @28, K = 1152: 75 / 83; 256 <- 256 @14, K = 2304: 58 / 60 — the windowed forms gain nothing from 128-byte K rows
    //    below K = 2048, the 1x1 forms do from K = 1024)
    else if (gelu || (bn && Ktot < (window ? 2048 : 1024))) return WIDE_NOT_TAKEN;
    else if (N >= 1024 && tm256 * tn256 >= 128) pick = 3;
    else if (window || Ktot >= 1024) pick = 2;
    else return WIDE_NOT_TAKEN;
  }
  switch (pick) {
    case 3: return dispatch<W256>(a, b, ep, rm, M, N, Ktot, splits, st);
    case 2: return dispatch<W256x128>(a, b, ep, rm, M, N, Ktot, splits, st);
    default: return dispatch<W128>(a, b, ep, rm, M, N, Ktot, splits, st);
  }
}
