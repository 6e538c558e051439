// pow_wide.h -- the fixed-exponent powers (pow_exec.h: Fp.sqrt, Fp2.sqrt, sqrt_div_fp2; reference math.ts:251-264, 521-538, 1196-1198) with ONE LIMB PER LANE: the latency
// form for launches of at most a wavefront or two per SIMD (round 6).
//
// A single verify / sign spends 0.7 ms in nbls_fp2_pow_kernel: 377 Fp2 squarings on ONE lane pair, each squaring a 196-multiply-add product plus a 196-multiply-add
// Montgomery reduction per lane (~480 instructions at the issue cadence of a lone wavefront), while 62 lanes idle.  The lane split of the step programs cannot help here:
// it divides a lane-op's PRODUCTS over lanes, and the reduction -- half of a squaring -- stays serial.  This form divides the reduction itself: an Fp value lives on a row
// of 16 lanes, limb j in lane j (14 used), and a Montgomery product is 14 rows of
//     acc += A_j * b_i                     one multiply-add per lane (b_i: limb i of B, broadcast inside the row by ds_swizzle -- no LDS memory involved)
//     m = (acc_0 * n0) mod 2^28            on lane 0 of the row, broadcast through v_readlane (an SGPR)
//     acc += p_j * m                       lane 0's column is now divisible by 2^28
//     acc_j <- (acc_j >> 28) + (acc_(j+1) mod 2^28)      one DPP row shift: every column moves one lane down, its carry stays where it is
// -- ten instructions per row where the one-lane form spends 28, and the carries never ripple: limbs stay "lazily" normalised (below 2^28 + 16), which the 64-bit columns
// absorb.  An Fp2 value takes two rows (component r on row r); the components of a product are computed side by side, the partner's component arrives by one ds_swizzle.
// One wavefront per element; the window table of the sliding-window chain lives in LDS.
//
// Written once and compiled twice (like pow_exec.h): the policy L supplies the per-lane types and the cross-lane moves.  On the device a value of type U is the lane's
// own 32-bit word; on the host (vm_sim.cpp, test-only) it is an array over the 32 lanes of the two rows and every cross-lane move is a loop -- the CPU suite runs
// the same sequences against Python's pow (tests/test_vm_sim.py).
#pragma once
#include "pow_exec.h"

namespace nbls {

// 16 p with limbs that dominate any lazily normalised limb.  NBLS_BIAS16_28 is 16 p with every limb at least 2^28 - 1 (the one-lane kernels' bias); adding 2^28 to limb j < 13 and
// taking 1 from limb j + 1 leaves the sum unchanged: every limb below the top is at least 2^29 - 2 (and below 2^29 + 2^28), the top one is 1,704,207 -- above the top limb of any value
// below 2 p (213,027)
struct WideConsts { u32 p[16], bias[16], r1[16]; };
inline WideConsts wide_consts() {
  WideConsts c = {};
  const u32 P[NL] = NBLS_P28, B16[NL] = NBLS_BIAS16_28;
  for (int j = 0; j < NL; j++) { c.p[j] = P[j]; c.r1[j] = NBLS_R1[j]; c.bias[j] = B16[j] + (j < NL - 1 ? (1u << 28) : 0u) - (j > 0 ? 1u : 0u); }
  return c;
}

// L::U (32-bit per lane), L::W (64-bit per lane) and
//   U konst(const u32* t16)            per-lane constant: t16[lane mod 16]
//   bool-like row selection:  U sel(U a, U b)   -> row 1 ? b : a
//   U add(U, U), sub(U, U), and_(U, u32), shr(U, int), mul_lo(U, u32), U lo(W)
//   W zero(), mad(U a, U b, W acc), mad_s(U a, u32 s, W acc), shr28(W), add_lo(W w, U x)   (low word += x; the caller guarantees no carry)
//   U bcast(U v, int i)                lane i of the own row to the whole row        U xchg(U v)   the same lane of the other row
//   U shl1(U v) / shr1(U v)            row shift towards lane 0 / away from it, zero filled
//   u32 lane_of(U v, int k)            wavefront-uniform copy of lane k
//   tab_put(int e, U v) / U tab_get(int e)
template <class L, bool FP2>
struct WideField {
  typedef typename L::U U;
  typedef typename L::W W;
  L& l;
  U P, P0, P1, BIAS, R1;      // p_j ; p_j on row 0 / row 1 only ; the subtraction bias ; the Montgomery one
  explicit NBLS_HD WideField(L& l_, const WideConsts& c) : l(l_) {
    u32 z[16] = {0};
    P = l.konst(c.p); BIAS = l.konst(c.bias); R1 = l.konst(c.r1);
    const U Z = l.konst(z);
    P0 = l.sel(P, Z); P1 = l.sel(Z, P);
  }
  // Montgomery product(s) on every row at once: A1 * B1 (+ A2 * B2) / R, limbs lazily normalised (below 2^28 + 16), value below (sum of the products) / R + p
  template <bool TWO>
  NBLS_HD U mont(const U& A1, const U& B1, const U& A2, const U& B2) {
    U b1[NL], b2[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) { b1[i] = l.bcast(B1, i); if (TWO) b2[i] = l.bcast(B2, i); }
    W acc = l.zero();
#pragma unroll
    for (int i = 0; i < NL; i++) {
      acc = l.mad(A1, b1[i], acc);
      if (TWO) acc = l.mad(A2, b2[i], acc);
      const U ml = l.and_(l.mul_lo(l.lo(acc), NBLS_N0_28), LMASK);
      acc = l.mad_s(P0, l.lane_of(ml, 0), acc);
      if (FP2) acc = l.mad_s(P1, l.lane_of(ml, 16), acc);
      const U lo28 = l.and_(l.lo(acc), LMASK);          // lane 0 of a row: zero
      acc = l.add_lo(l.shr28(acc), l.shl1(lo28));       // below 2^32: the columns stay below 2^60 (operand limbs below 2^30, two products per row at most)
    }
    const U r = l.lo(acc);
    return l.add(l.and_(r, LMASK), l.shr1(l.shr(r, 28)));
  }
  NBLS_HD U contract(const U& a) { return mont<false>(a, R1, a, a); }      // a * 1: any value with limbs below 2^30 -> below 2 p
  // exact limbs (below 2^28) for the store: the carries ripple at most thirteen lanes
  NBLS_HD U normalise(U r) {
#pragma unroll
    for (int k = 0; k < NL - 1; k++) r = l.add(l.and_(r, LMASK), l.shr1(l.shr(r, 28)));
    return r;
  }
  // ---- the policy of pow_exec.h's chains (fp_pow_seq / fp2_pow_seq)
  typedef U V;
  NBLS_HD void copy(V& r, const V& a) { r = a; }
  NBLS_HD void sqr(V& res, const V& a) {
    if (!FP2) { res = mont<false>(a, a, a, a); return; }
    const U pa = l.xchg(a);
    // row 0: (a0 + a1)(a0 - a1 + 16 p) ; row 1: (2 a1) a0      (math.ts:477-484)
    const U A = l.add(a, l.sel(pa, a));
    const U B = l.sel(l.sub(l.add(a, BIAS), pa), pa);
    res = mont<false>(A, B, A, B);
  }
  NBLS_HD void mul(V& res, const V& a, const V& e) {
    if (!FP2) { res = mont<false>(a, e, a, e); return; }
    const U pa = l.xchg(a), pe = l.xchg(e);
    // row 0: a0 e0 + (16 p - a1) e1 ; row 1: a1 e0 + a0 e1
    res = mont<true>(a, l.sel(e, pe), l.sel(l.sub(BIAS, pa), pa), l.sel(pe, e));
  }
  NBLS_HD void conj(V& res, const V& a) { res = contract(l.sel(a, l.sub(BIAS, a))); }      // (a0, -a1), contracted: every value that enters sqr / mul is below 2 p
  NBLS_HD void load(V& r) { r = contract(l.load()); }
  NBLS_HD void store(const V& a) { l.store(normalise(a)); }
  NBLS_HD void tab_put(int j, const V& a) { l.tab_put(j, a); }
  NBLS_HD void tab_get(V& r, unsigned j) { r = l.tab_get((int)j); }
};

}  // namespace nbls
