// g1_wide.h -- the window combination of the G1 multi-scalar multiplication with ONE LIMB PER LANE (round 6).
//
// dev_msm ends in  acc <- 2^12 acc + S_w  from the top window down: with the scalars split along the endomorphism, ten rounds of twelve doublings and one complete addition on
// ONE point -- 120 dependent doublings, 0.62 of the 2.18 ms of a 65,536-point call, on a four-lane step program whose lane-ops carry one product each (nothing for a lane split
// to divide).  What shortens such a chain is dividing the product itself (pow_wide.h): an Fp value on a row of sixteen lanes, four rows per wavefront, so the FOUR products of a
// doubling level run side by side, each as fourteen rows of [multiply-add | m | multiply-add with p | shift] (wide_exec.h WideOps::dot_rows).  The whole combination is one
// wavefront and one launch; the state lives in LDS slots of 64 bytes, read by whichever row needs them (a row reads any slot: the operands of its product are small signed
// combinations c1 * slot1 + c2 * slot2 -- the curve constants 3 b = 12 and the factors 2, 3, 8 of the formulas ride on the operands).
//
// The formulas are curve.h's (complete doubling and addition of Renes-Costello-Batina, a = 0, regrouped): a doubling is two rounds of four products --
//   t0 = y y, t1 = y z, t2 = (4 z)(3 z), xy = x y          with y = U - V kept as two slots (pt_dbl_n<SFp>: never summed, every use is a product operand)
//   x' = (t0 - 3 t2)(2 xy), U' = (t0 + 3 t2)^2, V' = (4 t2)(3 t2), z' = (4 t0)(2 t1)
// an addition three rounds of at most two products per row --
//   x0 = (3 x1) x2, t1 = y1 y2, b2 = (4 z1)(3 z2), t3 = x1 y2 + x2 y1 | t4 = y1 z2 + y2 z1, y3 = (4 x1)(3 z2) + (4 x2)(3 z1) |
//   x' = t3 (t1 - b2) - t4 y3, y' = (t1 - b2)(t1 + b2) + y3 x0, z' = (t1 + b2) t4 + x0 t3
// Values are SIGNED: a result is REDC(sum) in (V / R, V / R + p), |V| / R a small fraction of p for operands of a few p, so everything stays inside (-0.1 p, 1.1 p) and no bias
// is ever added; limbs are lazily normalised (WideOps::norm1), operand limbs stay below 2^30 in magnitude (one product per row) resp. products below 2^60 (two).
// Written once, compiled twice (device: vm_wide_kernel.hip; host: the simulator, tests/test_wide_sim.py).
#pragma once
#include "wide_exec.h"

namespace nbls {

enum { GW_ZERO, GW_X, GW_U, GW_V, GW_Z, GW_T0, GW_T1, GW_T2, GW_XY, GW_X2, GW_Y2, GW_Z2, GW_X0, GW_T1A, GW_B2, GW_T3, GW_T4, GW_Y3, GW_JUNK, GW_SLOTS };
// one operand: c1 * slot s1 + c2 * slot s2, packed s1 | s2 << 8 | (c1 & 255) << 16 | (c2 & 255) << 24
#define GW_T(s1, c1, s2, c2) ((u32)(s1) | ((u32)(s2) << 8) | (((u32)(c1) & 255u) << 16) | (((u32)(c2) & 255u) << 24))
#define GW_1(s) GW_T(s, 1, GW_ZERO, 0)
#define GW_Y GW_T(GW_U, 1, GW_V, -1)
#define GW_NONE GW_T(GW_ZERO, 0, GW_ZERO, 0)
// a row of a round: operands A0, B0, A1, B1 (the second product only in the rounds of an addition), the slot written
static const int GW_ROW_WORDS = 5;
static const int GW_DBL_ROUNDS = 2, GW_ADD_ROUNDS = 3;
#define GW_TABLE_INIT { \
  /* doubling, round 1 */ \
  GW_Y, GW_Y, GW_NONE, GW_NONE, GW_T0,   GW_Y, GW_1(GW_Z), GW_NONE, GW_NONE, GW_T1,   GW_T(GW_Z, 4, GW_ZERO, 0), GW_T(GW_Z, 3, GW_ZERO, 0), GW_NONE, GW_NONE, GW_T2,   GW_1(GW_X), GW_Y, GW_NONE, GW_NONE, GW_XY, \
  /* doubling, round 2 */ \
  GW_T(GW_T0, 1, GW_T2, -3), GW_T(GW_XY, 2, GW_ZERO, 0), GW_NONE, GW_NONE, GW_X,   GW_T(GW_T0, 1, GW_T2, 3), GW_T(GW_T0, 1, GW_T2, 3), GW_NONE, GW_NONE, GW_U, \
  GW_T(GW_T2, 4, GW_ZERO, 0), GW_T(GW_T2, 3, GW_ZERO, 0), GW_NONE, GW_NONE, GW_V,   GW_T(GW_T0, 4, GW_ZERO, 0), GW_T(GW_T1, 2, GW_ZERO, 0), GW_NONE, GW_NONE, GW_Z, \
  /* addition, round 1 */ \
  GW_T(GW_X, 3, GW_ZERO, 0), GW_1(GW_X2), GW_NONE, GW_NONE, GW_X0,   GW_Y, GW_1(GW_Y2), GW_NONE, GW_NONE, GW_T1A, \
  GW_T(GW_Z, 4, GW_ZERO, 0), GW_T(GW_Z2, 3, GW_ZERO, 0), GW_NONE, GW_NONE, GW_B2,   GW_1(GW_X), GW_1(GW_Y2), GW_1(GW_X2), GW_Y, GW_T3, \
  /* addition, round 2 */ \
  GW_Y, GW_1(GW_Z2), GW_1(GW_Y2), GW_1(GW_Z), GW_T4,   GW_T(GW_X, 4, GW_ZERO, 0), GW_T(GW_Z2, 3, GW_ZERO, 0), GW_T(GW_X2, 4, GW_ZERO, 0), GW_T(GW_Z, 3, GW_ZERO, 0), GW_Y3, \
  GW_NONE, GW_NONE, GW_NONE, GW_NONE, GW_JUNK,   GW_NONE, GW_NONE, GW_NONE, GW_NONE, GW_JUNK, \
  /* addition, round 3: the sum becomes the state (x, U = y, V = 0, z) */ \
  GW_1(GW_T3), GW_T(GW_T1A, 1, GW_B2, -1), GW_T(GW_T4, -1, GW_ZERO, 0), GW_1(GW_Y3), GW_X,   GW_T(GW_T1A, 1, GW_B2, -1), GW_T(GW_T1A, 1, GW_B2, 1), GW_1(GW_Y3), GW_1(GW_X0), GW_U, \
  GW_NONE, GW_NONE, GW_NONE, GW_NONE, GW_V,   GW_T(GW_T1A, 1, GW_B2, 1), GW_1(GW_T4), GW_1(GW_X0), GW_1(GW_T3), GW_Z }
// word k of row r of round q (q: 0, 1 the doubling; 2, 3, 4 the addition) at [(4 q + r) * GW_ROW_WORDS + k]
static const int GW_TABLE_WORDS = (GW_DBL_ROUNDS + GW_ADD_ROUNDS) * 4 * GW_ROW_WORDS;

// L: the policy of wide_exec.h plus   I muls(I, int)  (a limb times a small signed constant)
template <class L>
struct WideG1 {
  typedef typename L::I I;
  typedef typename L::W W;
  WideOps<L>& o;
  explicit NBLS_HD WideG1(WideOps<L>& o_) : o(o_) {}
  NBLS_HD I term(u32 w) {
    L& l = o.l;
    const int c1 = (int)(signed char)((w >> 16) & 255u), c2 = (int)(signed char)(w >> 24);
    return l.add(l.muls(l.ld((w & 255u) * 64u), c1), l.muls(l.ld(((w >> 8) & 255u) * 64u), c2));
  }
  // the result of this row for the round described by d (GW_ROW_WORDS words); the caller writes it to slot d[4] once every row of the round has read
  template <int P0>
  NBLS_HD I round(const u32* d) {
    I A[P0], B[P0];
#pragma unroll
    for (int r = 0; r < P0; r++) { A[r] = term(d[2 * r]); B[r] = term(d[2 * r + 1]); }
    const W acc = o.template dot_rows<P0>(A, B);
    return o.norm1w(acc);
  }
  // the coordinate as it leaves: +k p (non-negative), exact limbs -- what the step programs expect to find in HBM scratch (below 8 p)
  NBLS_HD I leave(const I& v, u32 k) { return o.exact(o.addmul_p(v, o.l.konst(k), false)); }
};

}  // namespace nbls
