// aot_exec.h -- step bodies of the ahead-of-time kernels (aot.h), written once and compiled twice like vm_exec.h: into aot_kernel.hip (gfx950) and into the
// test-only host simulator (vm_sim.cpp), which executes translated programs without a GPU.
//
// Arithmetic: the interpreter's (vm_exec.h: signed 28-bit limbs, mac28 into 28 lazy 64-bit columns, one Montgomery reduction per lane-op).  What differs is
// everything AROUND the multiply-adds, which a body that serves one signature only can afford to specialise:
//   * operand addresses are absolute (no address arithmetic), there is no lane predicate (idle lanes compute on the zero constant into a junk slot);
//   * K_DOT finish in the 64-bit columns (aot_dot_finish): post-added terms enter the columns as ONE multiply-add per limb with a signed per-lane coefficient,
//     the bias offs * p and the weak reduction's - q p as ONE multiply-add pass with the coefficient offs - q (q from the top two columns: no table, no global
//     load), then ONE carry pass.  A result can therefore differ from the interpreter's by a multiple of p (the two weak reductions estimate q differently);
//     both satisfy the bounds the host compiler books (value >= 0, below 3.02 p after a weak reduction), and every canonical output is the same.
#pragma once
#include <type_traits>
#include "aot.h"
#include "vm_exec.h"

namespace nbls {

// rows of the Montgomery reduction on the signed columns WITHOUT the closing carry pass: afterwards the value is sum_k acc[NL + k] 2^(28 k) (signed columns)
NBLS_HD void redc28_rows(u64* acc) {
  const u32 P[NL] = NBLS_P28;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const u32 m = ((u32)acc[i] * NBLS_N0_28) & LMASK;
#pragma unroll
    for (int j = 0; j < NL; j++) acc[i + j] += (u64)m * P[j];
    acc[i + 1] = (u64)((i64)acc[i + 1] + ((i64)acc[i] >> 28));
  }
}
// closing carry pass: columns c[0..13] (signed 64-bit) -> normalised limbs, the top limb keeps the rest
NBLS_HD void carry_cols(u32* r, const u64* c) {
  i64 cy = 0;
#pragma unroll
  for (int k = 0; k < NL - 1; k++) { const i64 v = (i64)c[k] + cy; r[k] = (u32)v & LMASK; cy = v >> 28; }
  r[NL - 1] = (u32)((i64)c[NL - 1] + cy);
}
static const i32 AOT_P_TOP = 106513;      // floor(p / 2^364)
static const i32 AOT_Q_MARGIN = 330;      // the estimate below is off by less than 130 (columns 11 and lower: |c| < 2^63 -> |c / 2^56| < 128); twice that and a bit
// Finish of a K_DOT lane-op in the columns.  acc: the 28 columns after the product rounds (no bias inside).  Semantics as vm_exec.h dot_finish:
//   dst = m * (REDC(sum) + offs p) + sum_t coef_t X_t  [weakly reduced]  [halved]
// REDUCED (round 6, the lane-split forms): the rows of the reduction have run already -- on every sub-lane, before the cross-lane sum -- and acc[NL ..] holds the summed upper columns.
template <u32 FLAGS, u32 T, bool REDUCED = false, typename LDSP = char*>
NBLS_HD void aot_dot_finish(u32* r, u64* acc, const u32 w0, const u32* post, LDSP lds) {
  const u32 P[NL] = NBLS_P28;
  if (!REDUCED) redc28_rows(acc);
  u64* c = acc + NL;
  i32 boffs = (FLAGS & AF_OFFS) ? (i32)((w0 >> 20) & 0xfu) : 0;
  if (FLAGS & (AF_MULTSH | AF_MULT3)) {
    // a multiplier on the reduced sum: the columns may be as large as 2^62.9, so they are carried to limbs first (the interpreter's order of operations)
    if (FLAGS & AF_OFFS) {
#pragma unroll
      for (int k = 0; k < NL; k++) c[k] = (u64)((i64)c[k] + (i64)boffs * (i64)P[k]);
    }
    u32 t[NL];
    carry_cols(t, c);
    const u32 sh = (w0 >> 16) & 3u, m3 = (FLAGS & AF_MULT3) ? (0u - ((w0 >> 18) & 1u)) : 0u;
#pragma unroll
    for (int k = 0; k < NL; k++) c[k] = (u64)((t[k] << sh) + (t[k] & m3));   // m = 1 .. 4 as (r << (m >> 1)) + (m == 3 ? r : 0)   // limbs below 2^31: m <= 4, top limb below 2^29 (values below 2047 p)
    boffs = 0;
  }
#pragma unroll
  for (u32 t = 0; t < T; t++) {
    u32 X[NL];
    ld14(X, lds, post[t] & 0xffffu);
    const i32 coef = (i32)post[t] >> 16;
#pragma unroll
    for (int k = 0; k < NL; k++) c[k] = (u64)((i64)c[k] + (i64)(i32)X[k] * (i64)coef);
  }
  i32 f = boffs;
  if (FLAGS & AF_WRED) {
    // q <= V / p from V / 2^364 = c[13] + c[12] / 2^28 + (less than 130), with the pending bias counted in
    const i64 est = (i64)c[NL - 1] + ((i64)c[NL - 2] >> 28);
    i32 e = (i32)est + boffs * AOT_P_TOP - AOT_Q_MARGIN;
    e = e < 0 ? 0 : e;
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 q = __umulhi((u32)e, 2642610142u) >> 16;     // floor(2^48 / 106514)
#else
    const u32 q = (u32)(((u64)(u32)e * 2642610142u) >> 48);
#endif
    f = boffs - (i32)q;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // one multiply-add per limb with an opaque coefficient: with the weak reduction the optimiser otherwise splits the pass into offs * P[k] and - q * P[k]; without it (round 5) it knows
  // offs < 16, computes offs * P[k] with a 32-bit multiply and adds it in 64 bits -- two instructions per limb where one v_mad_i64_i32 does (-14 per lane-op with a bias)
  asm volatile("" : "+v"(f));
#endif
  if (((FLAGS & AF_OFFS) && !(FLAGS & (AF_MULTSH | AF_MULT3))) || (FLAGS & AF_WRED)) {
#pragma unroll
    for (int k = 0; k < NL; k++) c[k] = (u64)((i64)c[k] + (i64)f * (i64)(i32)P[k]);
  }
  carry_cols(r, c);
  if (FLAGS & AF_HALVE) { if (w0 & (1u << 19)) halve28(r); }
}

// K_LIN with absolute addresses: dst = sum of the added slots - sum of the subtracted ones [weakly reduced] [halved]; fields: 16 bits each, from word 1
template <u32 NADD, u32 NSUB, u32 FLAGS, typename LDSP, typename W>
NBLS_HD u32 aot_lin(u32* r, LDSP lds, W&& word, const u32* __restrict__ qp_table) {
  const u32 w0 = word(0);
  auto field = [&](u32 t) { return (word(1 + t / 2) >> (16 * (t & 1))) & 0xffffu; };
  ld14(r, lds, field(0));
#pragma unroll
  for (u32 t = 1; t < NADD; t++) {
    u32 X[NL];
    ld14(X, lds, field(t));
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] += X[i];
  }
#pragma unroll
  for (u32 t = NADD; t < NADD + NSUB; t++) {
    u32 X[NL];
    ld14(X, lds, field(t));
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] -= X[i];
  }
  if (FLAGS & AF_WRED) weak_reduce(r, qp_table);
  carry_norm(r);
  if (FLAGS & AF_HALVE) { if (w0 & (1u << 16)) halve28(r); }
  return w0 & 0xffffu;
}

// buffer steps: w0 = slot | buffer << 16 | active << 31, w1 = byte offset
template <u32 KIND, u32 P0, typename LDSP>
NBLS_HD void aot_io(LDSP lds, const u32 w0, const u32 off, const u32 item, const bool live, const IOBuf* bufs) {
  const u32 slot = w0 & 0xffffu;
  const bool act = live && (w0 >> 31);
  const IOBuf& b = bufs[(w0 >> 16) & 7u];
  u32* g = (u32*)(b.ptr + (u64)item * b.stride + off);
  if (KIND == K_LOADW) {
    u32 x[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] = act ? g[i] : 0u;
    st14(lds, slot, x);   // inactive lanes: the junk slot
  } else if (KIND == K_LOAD) {
    const int nw = P0 ? (int)P0 / 4 : 12;
    u32 w[12], x[NL];
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = (act && i < nw) ? bswap32(g[nw - 1 - i]) : 0u;
    words_to_limbs(x, w);
    st14(lds, slot, x);
  } else if (KIND == K_STOREW) {
    u32 x[NL];
    ld14(x, lds, slot);
    if (act) {
#pragma unroll
      for (int i = 0; i < NL; i++) g[i] = x[i];
      g[14] = 0; g[15] = 0;
    }
  } else {   // K_STORE
    u32 x[NL], w[12];
    ld14(x, lds, slot);
    if (P0 == 0) csub_p(x);
    limbs_to_words(w, x);
    if (act) {
#pragma unroll
      for (int i = 0; i < 12; i++) g[11 - i] = bswap32(w[i]);
    }
  }
}

// Column budget instead of operand normalisation.  The host compiler keeps a lane-op's 64-bit columns from overflowing by normalising sum operands first
// (a carry pass of 39 instructions per operand, paid by the whole wavefront as soon as one lane needs it: 312 of the 2,199 instructions of an Fp12 squaring step).
// A specialised body knows every round's operand shapes, so it can bound every COLUMN instead: a round with operand magnitudes ca, cb (1 for a slot or a
// difference: limbs inside (-2^28, 2^28); 2 for a sum or a mixed-sign operand) adds at most terms(k) * ca * cb units of 2^56 to column k, terms(k) = the
// number of limb products that fall into it (k + 1 up to 14, then down).  A column that would pass 113 units (2^63 = 128; 14 for the reduction's own m * p rows,
// one of slack) is compressed first -- its part above 2^28 moves into the next column, four instructions -- and operands are multiplied un-normalised.  Only the
// middle columns ever need it: ~21 column compressions per Fp12 squaring step.
struct AotCompressPlan { u32 before[MAX_DOT_PRODUCTS + 1]; };
constexpr u32 aot_shape_units(u32 mode) { return (mode == 1u || mode == 3u) ? 2u : 1u; }
constexpr AotCompressPlan aot_compress_plan(u32 P0, u32 SH0, u32 SH1, u32 S = 1) {   // S: lane split -- the columns of S sub-lanes are summed before the reduction, so a round counts S times
  AotCompressPlan pl = {};
  u32 B[2 * NL] = {};
  const u32 LIMIT = 113;
  for (u32 r = 0; r < P0 && r < (u32)MAX_DOT_PRODUCTS; r++) {
    const u32 shape = ((r < 4 ? SH0 : SH1) >> (8 * (r & 3))) & 0xffu;
    const u32 add = S * aot_shape_units(shape & 3u) * aot_shape_units((shape >> SH_B_SHIFT) & 3u);
    u32 mask = 0;
    for (u32 k = 0; k < 2 * NL - 1; k++) { const u32 terms = k < (u32)NL ? k + 1 : 2 * NL - 1 - k; if (B[k] + terms * add > LIMIT) mask |= 1u << k; }
    for (u32 k = 0; k < 2 * NL - 1; k++) if (mask & (1u << k)) { B[k + 1] += 1; B[k] = 1; }
    for (u32 k = 0; k < 2 * NL - 1; k++) { const u32 terms = k < (u32)NL ? k + 1 : 2 * NL - 1 - k; B[k] += terms * add; }
    pl.before[r] = mask;
  }
  return pl;
}
// does any round of the signature carry a "normalise first" flag (the only signatures whose bodies change)
constexpr bool aot_has_norm(u32 SH0, u32 SH1) { return ((SH0 | SH1) & 0x24242424u) != 0; }
template <u32 MASK>
NBLS_HD void aot_compress_columns(u64* acc) {
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
    if (MASK & (1u << k)) {
      acc[k + 1] = (u64)((i64)acc[k + 1] + ((i64)acc[k] >> 28));
      acc[k] &= (u64)LMASK;
    }
  }
}

// host only: what aot_acc_sum calls in place of the DPP stages (set by the simulator; null elsewhere): (columns, how many)
typedef void (*AotSimLsHook)(u64*, int);
inline AotSimLsHook& aot_sim_ls_hook() { static AotSimLsHook h = nullptr; return h; }
// Lane split: sum of NC 64-bit columns over the LS = 4 (2) adjacent lanes of a lane-op, result in the first of them.  Two DPP stages (lane i += lane i + 1, then
// lane i += lane i + 2, inside rows of 16 lanes; groups of four never straddle a row) -- one stage for LS = 2.  On the host the simulator's hook does the same sum.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NBLS_LS_SUM_ASM)
#define NBLS_LS_SUM_ASM 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// 64-bit add of the same column of a neighbouring lane as TWO instructions (v_add_co_u32_dpp + v_addc_co_u32_dpp): the compiler's form of `v += dpp(v)` is two
// v_mov_b32_dpp and two adds (it folds a DPP move into a 32-bit add, not into a carry chain).  One block for all fourteen columns: a DPP operand must not have been
// written by the two preceding VALU instructions, which only holds for certain inside a block (s_nop 1 covers whatever the compiler placed in front of it; inside, the
// second stage reads a register written 27 instructions earlier).
#define NBLS_DPP_ADD64(L, H, CTRL) "v_add_co_u32_dpp " L ", vcc, " L ", " L " " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32_dpp " H ", vcc, " H ", " H ", vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define NBLS_DPP_STAGE(CTRL) NBLS_DPP_ADD64("%0", "%1", CTRL) NBLS_DPP_ADD64("%2", "%3", CTRL) NBLS_DPP_ADD64("%4", "%5", CTRL) NBLS_DPP_ADD64("%6", "%7", CTRL) NBLS_DPP_ADD64("%8", "%9", CTRL) \
  NBLS_DPP_ADD64("%10", "%11", CTRL) NBLS_DPP_ADD64("%12", "%13", CTRL) NBLS_DPP_ADD64("%14", "%15", CTRL) NBLS_DPP_ADD64("%16", "%17", CTRL) NBLS_DPP_ADD64("%18", "%19", CTRL) \
  NBLS_DPP_ADD64("%20", "%21", CTRL) NBLS_DPP_ADD64("%22", "%23", CTRL) NBLS_DPP_ADD64("%24", "%25", CTRL) NBLS_DPP_ADD64("%26", "%27", CTRL)
template <u32 LS>
__device__ __forceinline__ void aot_acc_sum14_asm(u64* c) {
  u32 l[NL], h[NL];
#pragma unroll
  for (int k = 0; k < NL; k++) { l[k] = (u32)c[k]; h[k] = (u32)(c[k] >> 32); }
#define NBLS_DPP_OPS "+v"(l[0]), "+v"(h[0]), "+v"(l[1]), "+v"(h[1]), "+v"(l[2]), "+v"(h[2]), "+v"(l[3]), "+v"(h[3]), "+v"(l[4]), "+v"(h[4]), "+v"(l[5]), "+v"(h[5]), "+v"(l[6]), "+v"(h[6]), \
  "+v"(l[7]), "+v"(h[7]), "+v"(l[8]), "+v"(h[8]), "+v"(l[9]), "+v"(h[9]), "+v"(l[10]), "+v"(h[10]), "+v"(l[11]), "+v"(h[11]), "+v"(l[12]), "+v"(h[12]), "+v"(l[13]), "+v"(h[13])
  if (LS == 4) asm volatile("s_nop 1\n\t" NBLS_DPP_STAGE("row_shl:1") NBLS_DPP_STAGE("row_shl:2") : NBLS_DPP_OPS : : "vcc");
  else asm volatile("s_nop 1\n\t" NBLS_DPP_STAGE("row_shl:1") : NBLS_DPP_OPS : : "vcc");
#pragma unroll
  for (int k = 0; k < NL; k++) c[k] = ((u64)h[k] << 32) | l[k];
}
#endif
template <u32 LS, int NC>
NBLS_HD void aot_acc_sum(u64* acc) {
#if defined(__HIP_DEVICE_COMPILE__)
#if NBLS_LS_SUM_ASM
  if (NC == NL) { aot_acc_sum14_asm<LS>(acc); return; }
#endif
#pragma unroll
  for (int c = 0; c < NC; c++) {
    u64 v = acc[c];
    v += ((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0x101, 0xf, 0xf, true) << 32) | (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0x101, 0xf, 0xf, true);   // row_shl:1
    if (LS == 4) v += ((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0x102, 0xf, 0xf, true) << 32) | (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0x102, 0xf, 0xf, true);   // row_shl:2
    acc[c] = v;
  }
#else
  if (aot_sim_ls_hook()) aot_sim_ls_hook()(acc, NC);   // the simulator supplies the cross-lane sum (vm_sim.cpp: lanes are visited from 63 down, so the partners' columns are there)
#endif
}
// Reduce first, then sum (round 6).  Until round 5 the 27 product columns of the S sub-lanes were summed (27 x 4 instructions per DPP stage: 216 for S = 4, 108 for S = 2) and the first sub-lane
// reduced the sum while the others computed junk.  But every lane of a wavefront executes the reduction anyway, and the reduction is linear modulo p: each sub-lane reduces ITS OWN columns
// (REDC(a) + REDC(b) = (a + b) / R mod p, each term adding less than p -- the host compiler books S p instead of p for these programs, trace.cpp emit_dot) and only the 14 upper columns
// cross the lanes: 112 / 56 instructions.  The sum of S columns must stay inside 63 bits: a column that could pass 127 units of 2^56 over the S sub-lanes is compressed first
// (aot_compress_columns), as the product rounds do under the column budget.  Returns the mask (bit k: upper column k) for a signature; S = 1: nothing to do.
constexpr u32 aot_ls_presum_mask(u32 P0, u32 SH0, u32 SH1, u32 S) {
  if (S <= 1) return 0;
  u32 B[2 * NL + 1] = {};
  const bool budget = aot_has_norm(SH0, SH1);
  const u32 LIMIT = 113;
  for (u32 r = 0; r < P0 && r < (u32)MAX_DOT_PRODUCTS; r++) {
    const u32 shape = ((r < 4 ? SH0 : SH1) >> (8 * (r & 3))) & 0xffu;
    const u32 add = aot_shape_units(shape & 3u) * aot_shape_units((shape >> SH_B_SHIFT) & 3u);
    if (budget) {   // the sub-lane's own compression plan (aot_compress_plan with S = 1)
      u32 mask = 0;
      for (u32 k = 0; k < 2 * NL - 1; k++) { const u32 terms = k < (u32)NL ? k + 1 : 2 * NL - 1 - k; if (B[k] + terms * add > LIMIT) mask |= 1u << k; }
      for (u32 k = 0; k < 2 * NL - 1; k++) if (mask & (1u << k)) { B[k + 1] += 1; B[k] = 1; }
    }
    for (u32 k = 0; k < 2 * NL - 1; k++) { const u32 terms = k < (u32)NL ? k + 1 : 2 * NL - 1 - k; B[k] += terms * add; }
  }
  // the reduction's rows: row i adds m P[j] (one unit each) to column i + j and the carry of column i (less than a unit) to column i + 1
  u32 mask = 0;
  for (u32 k = NL; k < 2 * NL; k++) {
    const u32 rows = k < 2 * NL - 1 ? 2 * NL - 1 - k : 0;
    const u32 U = B[k] + rows + 1;
    if (S * U > 127 && k < 2 * NL - 1) { mask |= 1u << (k - NL); B[k + 1] += 1; }
  }
  return mask;
}
// The columns are made opaque between product rounds: the optimiser would otherwise re-associate every column's sum over ALL rounds of an unrolled body
// (every round's operands live at once: 330-480 registers); the scheduling barrier keeps the LDS reads of later rounds from being hoisted to the top.
NBLS_HD void aot_round_fence(u64* acc) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int c = 0; c < 2 * NL - 1; c++) asm volatile("" : "+v"(acc[c]));
  __builtin_amdgcn_sched_barrier(0);
#else
  (void)acc;
#endif
}
NBLS_HD u32 aot_word(const V4& q, u32 i) { return i == 0 ? q.x : i == 1 ? q.y : i == 2 ? q.z : q.w; }

// One step of a translated program for one lane.  `D::quad(i)` returns 16-byte word i of the lane's descriptor (device: the first three were fetched a step
// ahead, the others are loaded here, a round ahead of their use); `commit(dst, limbs)` stores a result (device: at once -- LDS operations of a wavefront execute in
// order, so every read of the step precedes it; simulator: after all lanes of the step).
template <u32 KIND, u32 P0, u32 FLAGS, u32 T, u32 SH0, u32 SH1, u32 LS = 1, typename D = void, typename LDSP = char*, typename Commit = void>
NBLS_HD void aot_step(const D& desc, LDSP lds, const u32 item, const bool live, const IOBuf* bufs, const u32* __restrict__ qp_table, Commit&& commit) {
  if constexpr (KIND == K_DOT) {
    const u32 HQ = (AOT_DOT_HDR + T + 3) / 4;
    const V4 h0 = desc.quad(0);
    u32 post[T > 0 ? T : 1];
#pragma unroll
    for (u32 t = 0; t < T; t++) post[t] = aot_word(desc.quad((AOT_DOT_HDR + t) / 4), (AOT_DOT_HDR + t) & 3u);
    u64 acc[2 * NL];
#pragma unroll
    for (int i = 0; i < 2 * NL; i++) acc[i] = 0;
    V4 cur = desc.quad(HQ);
    // rounds with compile-time index (the compression plan is a constant expression of the signature)
    auto do_round = [&](auto RC) __attribute__((always_inline)) {
      constexpr u32 r = decltype(RC)::value;
      constexpr bool budget = aot_has_norm(SH0, SH1);   // signatures without normalised operands keep the host compiler's own budget (sum of ca * cb <= 8)
      constexpr u32 cmask = budget ? aot_compress_plan(P0, SH0, SH1, 1).before[r] : 0u;   // (lane split: every sub-lane reduces its own columns, so a round counts once)
      V4 nx = cur;
      if (r + 1 < P0) nx = desc.quad(HQ + r + 1);
      constexpr u32 shape = ((r < 4 ? SH0 : SH1) >> (8 * (r & 3))) & 0xffu;
      const u32 neg = (h0.y >> (4 * r)) & 15u;
      if (cmask) aot_compress_columns<cmask>(acc);
      {
        constexpr u32 sa = shape & (budget ? 3u : 7u), sb = (shape >> SH_B_SHIFT) & (budget ? 3u : 7u);   // under the column budget operands are never normalised
        u32 A[NL], B[NL], X[NL], Y[NL];
        ld14(A, lds, cur.x);
        if (sa & 3u) ld14(X, lds, cur.y);
        ld14(B, lds, cur.z);
        if (sb & 3u) ld14(Y, lds, cur.w);
        dot_combine(A, X, sa, neg & 3u);
        dot_combine(B, Y, sb, (neg >> 2) & 3u);
        mac28(acc, A, B);
      }
      aot_round_fence(acc);
      cur = nx;
    };
    if constexpr (P0 > 0) do_round(std::integral_constant<u32, 0>{});
    if constexpr (P0 > 1) do_round(std::integral_constant<u32, 1>{});
    if constexpr (P0 > 2) do_round(std::integral_constant<u32, 2>{});
    if constexpr (P0 > 3) do_round(std::integral_constant<u32, 3>{});
    if constexpr (P0 > 4) do_round(std::integral_constant<u32, 4>{});
    if constexpr (P0 > 5) do_round(std::integral_constant<u32, 5>{});
    if constexpr (P0 > 6) do_round(std::integral_constant<u32, 6>{});
    if constexpr (P0 > 7) do_round(std::integral_constant<u32, 7>{});
    u32 res[NL];
    if constexpr (LS > 1) {
      // lane split: every sub-lane reduces its own columns, the 14 upper columns are summed across the sub-lanes and land in the first one (the others finish into the junk slot)
      redc28_rows(acc);
      constexpr u32 smask = aot_ls_presum_mask(P0, SH0, SH1, LS);
      if (smask) aot_compress_columns<(smask << NL)>(acc);
      aot_acc_sum<LS, NL>(acc + NL);
      aot_dot_finish<FLAGS, T, true>(res, acc, h0.x, post, lds);
    } else aot_dot_finish<FLAGS, T>(res, acc, h0.x, post, lds);
    commit(h0.x & 0xffffu, res);
  } else if constexpr (KIND == K_LIN) {
    u32 res[NL];
    const u32 dst = aot_lin<P0, T, FLAGS>(res, lds, [&](u32 i) { return aot_word(desc.quad(i / 4), i & 3u); }, qp_table);
    commit(dst, res);
  } else if constexpr (KIND == K_LOAD || KIND == K_LOADW || KIND == K_STORE || KIND == K_STOREW) {
    const V4 h0 = desc.quad(0);
    aot_io<KIND, P0>(lds, h0.x, h0.y, item, live, bufs);
  } else {
    // flags, selects, comparisons, status: the interpreter's lane code on absolute addresses (instance base 0)
    Step st;
    st.kind = (uint8_t)KIND; st.nlanes = 64; st.p0 = (uint8_t)P0; st.p1 = 0; st.desc_off = 0; st.stride = KIND == K_STATUS ? 8 : 4; st.lin = 0;
    st.shape[0] = st.shape[1] = 0; st.rsv[0] = st.rsv[1] = 0;
    const V4 h0 = desc.quad(0);
    u32 d[8] = {h0.x, h0.y, h0.z, h0.w, 0, 0, 0, 0};
    if (KIND == K_STATUS) { const V4 h1 = desc.quad(1); d[4] = h1.x; d[5] = h1.y; d[6] = h1.z; d[7] = h1.w; }
    LaneCtx cx;
    cx.inst = 0; cx.item = item; cx.shared = false;
    cx.live = KIND == K_STATUS ? (live && (h0.x >> 31)) : live;
    u32 res[NL];
    const u32 dst = exec_lane(st, d, lds, cx, bufs, res, qp_table);
    if (dst != 0xffffffffu) commit(dst, res);
  }
}

}  // namespace nbls
