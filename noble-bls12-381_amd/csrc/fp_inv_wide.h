// fp_inv_wide.h -- the modular inverse (fp_inv.h: Pornin's binary GCD, 28 rounds of 28 inner iterations) with ONE LIMB PER LANE, four elements per wavefront (round 6).
//
// Every single call ends in at least one Fp inversion on ONE lane: the norm of the Miller value in the final exponentiation's easy part (math.ts:856-861 through Fp12.invert),
// the Z of a hash point, of a signature, of a multi-scalar product -- 0.10-0.12 ms each, ~8,700 clocks per round, of which the 28 inner iterations on the 58-bit
// approximations are a third; the rest is what a lane does alone on fourteen limbs: picking the top limbs (14 x 14 masked moves), applying the factors to a, b, u, v (4 x 14
// multiply-adds with carries), negating, reducing u and v.  With limb j of a, b, u, v in lane j of a row of sixteen lanes those are a handful of instructions: the top limbs
// come through a ballot and six gathers, the factors are two multiply-adds per lane and one row shift, u and v stay lazily normalised and signed (they are only ever multiplied
// and added until the end: |u| grows by at most p per round), and only a and b -- whose top bits the next round reads -- get an exact carry pass (thirteen shifts).  The inner
// iterations are the same code on row-uniform values.  Same algorithm, same factors, the same (a, b) after every round as fp_mont_inverse; the result is y^-1 R^2 as a
// representative below 2.1 p with exact limbs (the one-lane routine returns one below 2 p: consumers load it as a scratch element below 8 p and multiply).
//
// Written once, compiled twice (device: vm_wide_kernel.hip nbls_fp_inv_wide_kernel; host: the simulator, tests/test_vm_sim.py).  L: the policy of wide_exec.h plus
//   I or_(I, I), I spread(i32) (a row-uniform scalar on every lane), i32 first(I) (a row-uniform value as a scalar), I plimbs() (p_j),
//   u32 nonzero_mask(I) (bit j set when lane j of the row is non-zero, lanes 0..13), I gather(I, u32 j) (lane j of the row on every lane), I pick(u32 lanebit, I a, I b)
//   (a where bit j of lanebit is set, else b), I lane_eq(u32 k, I a, I b) (a on lane k, b elsewhere), I muls(I, int)
#pragma once
#include "wide_exec.h"
#include "fp_inv.h"

namespace nbls {

// the 28 inner iterations of one round (fp_inv.h): update factors from the approximations
NBLS_HD void fp_inv_inner(u64 abar, u64 bbar, i32& f0, i32& g0, i32& f1, i32& g1) {
  f0 = 1; g0 = 0; f1 = 0; g1 = 1;
  for (int j = 0; j < 28; j++) {
    const u32 odd = 0u - (u32)(abar & 1);
    const u32 sw = odd & ((abar < bbar) ? ~0u : 0u);
    const u64 sw64 = (u64)(i64)(i32)sw, odd64 = (u64)(i64)(i32)odd;
    const u64 t = (abar ^ bbar) & sw64; abar ^= t; bbar ^= t;
    const u32 tf = ((u32)f0 ^ (u32)f1) & sw; f0 = (i32)((u32)f0 ^ tf); f1 = (i32)((u32)f1 ^ tf);
    const u32 tg = ((u32)g0 ^ (u32)g1) & sw; g0 = (i32)((u32)g0 ^ tg); g1 = (i32)((u32)g1 ^ tg);
    abar -= bbar & odd64;
    f0 -= (i32)((u32)f1 & odd); g0 -= (i32)((u32)g1 & odd);
    abar >>= 1;
    f1 = (i32)((u32)f1 << 1); g1 = (i32)((u32)g1 << 1);
  }
}

template <class L>
struct WideInv {
  typedef typename L::I I;
  typedef typename L::W W;
  WideOps<L>& o;
  explicit NBLS_HD WideInv(WideOps<L>& o_) : o(o_) {}
  // (x f + y g) / 2^28 for an exact division, limbs lazily placed: lane j takes the high part of its column and the low 28 bits of the next
  NBLS_HD I combine_shift(const I& x, i32 f, const I& y, i32 g) {
    L& l = o.l;
    const W c = l.mad(x, l.spread(f), l.mad(y, l.spread(g), l.wzero()));
    return l.add(l.lo(l.sar28(c)), l.shl1(l.and_(l.lo(c), LMASK)));
  }
  // exact limbs 0..12 in [0, 2^28), the signed rest in lane 13 -> the same for the negated value.  The +1 of the two's complement ripples through the low limbs that
  // are zero: limb j below the first non-zero one stays 0, that one becomes 2^28 - e, those above 2^28 - 1 - e, the top -top - 1 (or -top when every low limb is zero)
  NBLS_HD I negate_exact(const I& e) {
    L& l = o.l;
    const u32 nz = l.nonzero_mask(e) & 0x1fffu;                                      // limbs 0..12
    const u32 lowbit = nz & (0u - nz);                                                // the first non-zero limb (0: none)
    const u32 above = lowbit ? ~(lowbit | (lowbit - 1u)) : 0u;                        // lanes above it
    const I comp = l.sub(l.konst(LMASK), e);                                          // 2^28 - 1 - e
    I r = l.pick(above & 0x1fffu, comp, l.zero());
    r = l.pick(lowbit, l.add(comp, l.konst(1u)), r);
    const I top = l.sub(l.sub(l.zero(), e), l.konst(lowbit ? 1u : 0u));
    return l.lane_eq(13u, top, r);
  }
  // y: exact limbs of a value below 2^392 (lanes 14, 15 zero); r3: the limbs of R^3 mod p.  Returns y^-1 R^2 mod p (0 for y = 0) with exact limbs, below 2.1 p
  NBLS_HD I invert(const I& y, const I& r3) {
    L& l = o.l;
    I a = y, b = l.plimbs(), u = l.lane_eq(0u, l.konst(1u), l.zero()), v = l.zero();
    for (int round = 0; round < 28; round++) {
      // approximations (fp_inv.h): the low limb exactly + the top 30 bits of the 64-bit window that starts at the highest limb where a | b is non-zero
      const u32 nz = l.nonzero_mask(l.or_(a, b)) & 0x3fffu;
      const u32 h = nz ? (u32)(63 - clz64((u64)nz)) : 0u;                              // the highest limb where a | b is non-zero
      const u32 hh = h < 2u ? 2u : h;
      const u32 a0 = (u32)l.first(l.gather(a, 0u)), b0 = (u32)l.first(l.gather(b, 0u));
      const u32 at = (u32)l.first(l.gather(a, hh)), at1 = (u32)l.first(l.gather(a, hh - 1u)), at2 = (u32)l.first(l.gather(a, hh - 2u));
      const u32 bt = (u32)l.first(l.gather(b, hh)), bt1 = (u32)l.first(l.gather(b, hh - 1u)), bt2 = (u32)l.first(l.gather(b, hh - 2u));
      const bool small = h <= 2u && ((at | bt) >> 2) == 0;
      const u64 A = ((u64)at << 36) | ((u64)at1 << 8) | (at2 >> 20), B = ((u64)bt << 36) | ((u64)bt1 << 8) | (bt2 >> 20);
      int sh = clz64(A | B); if (sh > 63) sh = 0;
      u64 abar = (((A << sh) >> 34) << 28) | a0;
      u64 bbar = (((B << sh) >> 34) << 28) | b0;
      if (small) { abar = ((u64)at << 56) | ((u64)at1 << 28) | at2; bbar = ((u64)bt << 56) | ((u64)bt1 << 28) | bt2; }      // (hh = 2: at2 = a[0])
      i32 f0, g0, f1, g1;
      fp_inv_inner(abar, bbar, f0, g0, f1, g1);
      // (a, b) <- (a f0 + b g0, a f1 + b g1) / 2^28 with exact limbs, made non-negative (the factors follow the sign)
      I na = o.exact(combine_shift(a, f0, b, g0)), nb = o.exact(combine_shift(a, f1, b, g1));
      const bool sa = l.first(l.gather(na, 13u)) < 0, sb = l.first(l.gather(nb, 13u)) < 0;
      const I nna = negate_exact(na), nnb = negate_exact(nb);
      a = l.pick(sa ? 0xffffu : 0u, nna, na); b = l.pick(sb ? 0xffffu : 0u, nnb, nb);
      if (sa) { f0 = -f0; g0 = -g0; }
      if (sb) { f1 = -f1; g1 = -g1; }
      // (u, v) <- (u f0 + v g0, u f1 + v g1) / 2^28 mod p: one Montgomery step each; signed, lazily normalised, never folded (|u|, |v| grow by at most p per round)
      const u32 u0 = (u32)l.first(l.gather(u, 0u)), v0 = (u32)l.first(l.gather(v, 0u));
      const u32 qu = ((u0 * (u32)f0 + v0 * (u32)g0) * NBLS_N0_28) & LMASK, qv = ((u0 * (u32)f1 + v0 * (u32)g1) * NBLS_N0_28) & LMASK;
      const W cu = l.mad(u, l.spread(f0), l.mad(v, l.spread(g0), l.mad_pq(l.spread((i32)qu), l.wzero())));
      const W cv = l.mad(u, l.spread(f1), l.mad(v, l.spread(g1), l.mad_pq(l.spread((i32)qv), l.wzero())));
      u = o.norm1(l.add(l.lo(l.sar28(cu)), l.shl1(l.and_(l.lo(cu), LMASK))));
      v = o.norm1(l.add(l.lo(l.sar28(cv)), l.shl1(l.and_(l.lo(cv), LMASK))));
    }
    // b = gcd = 1 and v = y^-1 mod p (any sign, below 29 p in magnitude): y^-1 R^2 = REDC(v R^3), in (-0.02 p, 1.02 p); + p and exact limbs
    const I A[1] = {v}, B[1] = {r3};
    const W acc = o.template dot_rows<1>(A, B);
    return o.exact(o.addmul_p(o.norm1w(acc), l.konst(1u), false));
  }
};

}  // namespace nbls
