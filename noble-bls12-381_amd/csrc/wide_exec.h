// wide_exec.h -- step programs with ONE LIMB PER LANE: the engine's multi-wavefront item form (round 6).
//
// The lane-split forms (aot.h) shorten a lone item's instruction stream by sharing a lane-op's PRODUCTS among two or four lanes; the Montgomery reduction and the finish --
// more than half of a cyclotomic-squaring step by then -- stay on one lane, so a single final exponentiation still takes 0.85 ms, and every verifyBatch ends in one.  This form
// divides the reduction too.  A lane-op runs on a ROW of sixteen lanes, limb j of every operand and of the result in lane j (pow_wide.h has the idea in its simplest setting);
// a K_DOT lane-op is fourteen rows of
//     acc += A^r_j * b^r_i  for every product r       b^r_i: limb i of the round's second operand, broadcast inside the row (ds_swizzle)
//     m = (acc_0 * n0) mod 2^28 ; acc += p_j * m      lane 0's m reaches the row through v_readlane
//     acc_j <- (acc_j >> 28) + (acc_(j+1) mod 2^28)   one DPP row shift
// -- 2 r + 14 instructions per row where a lane of the lane-split form spends 28 (r / 4) + 28 + the cross-lane sums.  A wavefront holds four rows, an item of W lane-ops takes
// ceil(W / 4) wavefronts of ONE workgroup: the slots live in the workgroup's LDS exactly as the other forms lay them out (limb l at byte 4 l of its slot), and since a step's
// lane-ops now sit in different wavefronts, a step is bracketed by two barriers (all reads of the step | its writes | the next step's reads) -- the engine's first.  Limbs are
// kept LAZILY normalised and SIGNED between steps (|limb| below 2^28 + 2^6: a carry moves one lane per step at most, never ripples); values keep the bounds the host compiler
// books (a result is REDC(sum) in [V / R, V / R + p) as in every other form), so the same compiled programs run unchanged and every canonical output is the same.  Scratch in HBM
// (K_STOREW) is written with exact limbs, as the other forms expect to find it.
//
// Written once, compiled twice: the policy L supplies the per-lane types and the cross-lane moves (device: vm_wide_kernel.hip; host: the test-only simulator, where a value is
// the array of a row's sixteen lanes).
//   L::I (signed 32-bit per lane), L::W (signed 64-bit per lane)
//   I add(I, I), sub(I, I), and_(I, u32), sar(I, int), shl(I, int), mul_lo(I, u32), mul_small(I, u32), lo(W), I zero()
//   W wzero(), mad(I a, I b, W acc), mad_p(I ml, W acc)   acc + p_j * (lane 0 of the row's ml),   mad_pq(I q, W acc) acc + p_j * q_j (q small, non-negative),   sar28(W), addw(W, I), addww(W, W)
//   I low13(I) lane < 13: the low 28 bits, lane 13: the word as is ; I carry13(I) lane < 13: the word as is, lane 13 and above: 0
//   I konst(u32) the same value on every lane ; I wred_q(I top) the weak reduction's multiplier from the top limb
//   I bcast(I v, int i), shl1(I), shr1(I)        row moves: broadcast of lane i ; lane j <- lane j + 1 ; lane j <- lane j - 1 (zero filled)
//   fence()                                      device: a scheduling barrier (keeps the requests above it) ; host: nothing
//   I ld(u32 byte_offset)                        limb j of the slot at that LDS offset (lanes 14, 15: the slot's padding words, kept zero)
#pragma once
#include "vm_exec.h"

namespace nbls {

template <class L>
struct WideOps {
  typedef typename L::I I;
  typedef typename L::W W;
  L& l;
  explicit NBLS_HD WideOps(L& l_) : l(l_) {}
  // one lazy carry pass: limbs back into (-2^6, 2^28 + 2^6) for |limb| below 2^33.  The TOP limb (lane 13) keeps whatever is above it -- as carry_norm does in the one-lane
  // forms: a reduction's raw result may be negative before the bias offs p is added, and its sign must stay in the top limb, where the weak reduction's estimate reads it
  // (low13 / carry13: the mask and the carry of every lane below the top one; the top lane keeps its word and sends nothing on)
  NBLS_HD I norm1(const I& r) { return l.add(l.low13(r), l.shr1(l.carry13(l.sar(r, 28)))); }
  NBLS_HD I norm1w(const W& a) { return l.add(l.low13(l.lo(a)), l.shr1(l.carry13(l.lo(l.sar28(a))))); }
  // exact limbs of a non-negative value below 2^392 (for scratch in HBM): a carry ripples at most thirteen lanes
  NBLS_HD I exact(I r) {
#pragma unroll
    for (int k = 0; k < NL - 1; k++) r = norm1(r);
    return r;
  }
  // REDC(sum_r A[r] * B[r]): signed operand limbs below 2^30 in magnitude, up to eight products; result as a 64-bit column per lane (magnitude below 2^34), not yet normalised
  template <int P0>
  NBLS_HD W dot_rows(const I* A, const I* B) {
    // A row's reduction is ONE dependent chain (m from the column, its broadcast, the multiply-add, the shift: every instruction waits for the one before, ~11 clocks each for a
    // lone wavefront), and the products of the NEXT row do not depend on it: they are summed into a column of their own (q) while the chain runs and join the accumulator at
    // the start of their row.  The broadcasts run two rows ahead of their use for the same reason.
    W acc = l.wzero(), q = l.wzero();
    I b[P0 > 0 ? P0 : 1], nb[P0 > 0 ? P0 : 1];
#pragma unroll
    for (int r = 0; r < P0; r++) b[r] = l.bcast(B[r], 0);
#pragma unroll
    for (int r = 0; r < P0; r++) nb[r] = l.bcast(B[r], 1);
#pragma unroll
    for (int r = 0; r < P0; r++) q = l.mad(A[r], b[r], q);
#pragma unroll
    for (int i = 0; i < NL; i++) {
#pragma unroll
      for (int r = 0; r < P0; r++) b[r] = i + 2 < NL ? l.bcast(B[r], i + 2) : nb[r];      // row i + 2's limbs requested
      l.fence();
      acc = l.addww(acc, q);
      W qn = l.wzero();
      if (i + 1 < NL) {
#pragma unroll
        for (int r = 0; r < P0; r++) qn = l.mad(A[r], nb[r], qn);                         // row i + 1's products: independent of the chain below
      }
      const I ml = l.and_(l.mul_lo(l.lo(acc), NBLS_N0_28), LMASK);
      acc = l.mad_p(ml, acc);
      const I lo28 = l.and_(l.lo(acc), LMASK);          // lane 0 of the row: zero
      acc = l.addw(l.sar28(acc), l.shl1(lo28));
      q = qn;
#pragma unroll
      for (int r = 0; r < P0; r++) nb[r] = b[r];
    }
    return acc;
  }
  // r +- k p with a small per-row multiplier k (the bias offs <= 15, the weak reduction's q <= 127): k p_j is a 35-bit number, so its part above 2^28 goes to the next lane
  NBLS_HD I addmul_p(const I& r, const I& k, bool subtract) {
    const W t = l.mad_pq(k, l.wzero());
    const I lo = l.and_(l.lo(t), LMASK), hi = l.shr1(l.lo(l.sar28(t)));
    return subtract ? l.sub(l.sub(r, lo), hi) : l.add(l.add(r, lo), hi);
  }
  // weak reduction (vm_exec.h weak_reduce): r -= q p, q = floor((r_13 - 9) * floor(2^48 / 106514) / 2^48) <= V / p for limbs below 2^31 - 2^28 in magnitude; afterwards below 2.01 p
  NBLS_HD I weak(const I& r) { return addmul_p(r, l.wred_q(l.bcast(r, NL - 1)), true); }
  // the finish of a K_DOT lane-op (vm_exec.h dot_finish): m * (REDC + offs p) +- post-added slots [weakly reduced], lazily normalised.  post: nadd added then nsub subtracted limbs
  NBLS_HD I dot_finish(const W& acc, u32 mult, u32 offs, const I* post, int nadd, int nsub, bool wred) {
    I r = norm1w(acc);
    if (offs) r = addmul_p(r, l.konst(offs), false);
    if (mult > 1) r = l.mul_small(norm1(r), mult);
#pragma unroll
    for (int t = 0; t < MAX_DOT_LINEAR; t++) if (t < nadd) r = l.add(r, post[t]);
#pragma unroll
    for (int t = 0; t < 2 * MAX_DOT_LINEAR; t++) if (t >= nadd && t < nadd + nsub) r = l.sub(r, post[t]);
    if (wred) r = weak(r);
    return norm1(r);
  }
  // K_LIN: sum of the added minus the subtracted slots [weakly reduced], lazily normalised
  NBLS_HD I lin(const I* terms, int nadd, int nsub, bool wred) {
    I r = terms[0];
#pragma unroll
    for (int t = 1; t < MAX_LIN_TERMS; t++) if (t < nadd) r = l.add(r, terms[t]);
#pragma unroll
    for (int t = 1; t < 2 * MAX_LIN_TERMS; t++) if (t >= nadd && t < nadd + nsub) r = l.sub(r, terms[t]);
    if (wred) r = weak(r);
    return norm1(r);
  }
  // operand of a product round from its one or two slots (vm_exec.h dot_combine; the "normalise first" flag of a shape has no meaning here: the columns never hold more than one row)
  NBLS_HD I combine(const I& x, const I& y, u32 mode, u32 neg) {
    if (mode == 0) return x;
    if (mode == 1) return l.add(x, y);
    if (mode == 2) return l.sub(x, y);
    const I a = (neg & 1u) ? l.sub(l.zero(), x) : x;
    return (neg & 2u) ? l.sub(a, y) : l.add(a, y);
  }
};

// One step for one lane-op (a row): every LDS read of the step happens in here; the result (if the kind has one) is handed back and written by the caller AFTER all rows of
// the workgroup have read (device: a barrier; host: the simulator commits after the last row).  d(k): word k of the lane-op's descriptor (vm.h).  Global buffers: l.gload / l.gstore.
template <class L, class D>
NBLS_HD bool wide_step(WideOps<L>& o, const Step& st, const D& d, const IOBuf* bufs, u32 item, bool live, u32& dst, typename L::I& out) {
  typedef typename L::I I;
  typedef typename L::W W;
  L& l = o.l;
  const u32 w0 = d(0);
  if (st.kind == K_DOT) {
    const int nadd = (int)(st.lin & 7u), nsub = (int)((st.lin >> 4) & 7u);
    I post[2 * MAX_DOT_LINEAR];
#pragma unroll
    for (int t = 0; t < 2 * MAX_DOT_LINEAR; t++) if (t < nadd + nsub) post[t] = l.ld((d(4 + t / 2) >> (16 * (t & 1))) & 0xffffu);
    const u32 w1 = d(1);
    I A[MAX_DOT_PRODUCTS], B[MAX_DOT_PRODUCTS];
#pragma unroll
    for (int r = 0; r < MAX_DOT_PRODUCTS; r++) {
      if (r < (int)st.p0) {
        const u32 shape = round_shape(st, (u32)r), sa = shape & 3u, sb = (shape >> SH_B_SHIFT) & 3u, neg = (w1 >> (4 * r)) & 15u;
        const I x = l.ld(d(DOT_HDR_WORDS + DOT_ROUND_WORDS * r)), y = sa ? l.ld(d(DOT_HDR_WORDS + DOT_ROUND_WORDS * r + 1)) : l.zero();
        const I u = l.ld(d(DOT_HDR_WORDS + DOT_ROUND_WORDS * r + 2)), v = sb ? l.ld(d(DOT_HDR_WORDS + DOT_ROUND_WORDS * r + 3)) : l.zero();
        A[r] = o.combine(x, y, sa, neg & 3u);
        B[r] = o.combine(u, v, sb, (neg >> 2) & 3u);
      }
    }
    W acc = l.wzero();
    if (!l.skip_rows()) switch (st.p0) {
      case 1: acc = o.template dot_rows<1>(A, B); break;
      case 2: acc = o.template dot_rows<2>(A, B); break;
      case 3: acc = o.template dot_rows<3>(A, B); break;
      case 4: acc = o.template dot_rows<4>(A, B); break;
      case 5: acc = o.template dot_rows<5>(A, B); break;
      case 6: acc = o.template dot_rows<6>(A, B); break;
      case 7: acc = o.template dot_rows<7>(A, B); break;
      case 8: acc = o.template dot_rows<8>(A, B); break;
      default: break;
    }
    out = o.dot_finish(acc, (st.p1 & DOTF_MULT) ? ((w0 >> 16) & 7u) : 1u, (st.p1 & DOTF_OFFS) ? ((w0 >> 20) & 0xfu) : 0u, post, nadd, nsub, (st.p1 & DOTF_WRED) != 0);
    dst = w0 & 0xffffu;
    return true;
  }
  if (st.kind == K_LIN) {
    const int nadd = st.p0, nsub = st.p1;
    I terms[2 * MAX_LIN_TERMS];
#pragma unroll
    for (int t = 0; t < 2 * MAX_LIN_TERMS; t++) if (t < nadd + nsub) terms[t] = l.ld((d(1 + t / 2) >> (16 * (t & 1))) & 0xffffu);
    out = o.lin(terms, nadd, nsub, (st.lin & 1u) != 0);
    dst = w0 & 0xffffu;
    return true;
  }
  const IOBuf& b = bufs[(w0 >> 16) & 7u];
  u32* g = (u32*)(b.ptr + (u64)item * b.stride + d(1));
  if (st.kind == K_LOADW) { out = l.gload(g, live); dst = w0 & 0xffffu; return true; }
  l.gstore(g, o.exact(l.ld(w0 & 0xffffu)), live);      // K_STOREW
  return false;
}

// what the form implements: K_DOT and K_LIN without halving, raw loads and stores (the programs of the final exponentiation's middle); per-lane halving bits: w0 bit 19 / 16
static inline bool wide_step_supported(const Step& st, const u32* descs) {
  if (st.kind == K_DOT) { if ((st.p1 & DOTF_HALVE) || st.p0 > MAX_DOT_PRODUCTS) return false; for (u32 k = 0; k < st.nlanes; k++) if (descs[st.desc_off + k * st.stride] & (1u << 19)) return false; return true; }
  if (st.kind == K_LIN) { for (u32 k = 0; k < st.nlanes; k++) if (descs[st.desc_off + k * st.stride] & (1u << 16)) return false; return true; }
  return st.kind == K_LOADW || st.kind == K_STOREW;
}

}  // namespace nbls
