// curve.h -- G1 / G2 point arithmetic over symbolic field values, for the validity checks, point sums, cofactor clearing.
// Points that cross the C ABI are affine, so any exact group law reproduces the reference's results; we use the COMPLETE
// projective formulas of Renes-Costello-Batina (2016, algorithms 7 and 9 for a = 0) because they need no data-dependent
// branches: both curves (E/Fp and E'/Fp2) have odd order, so the formulas are valid for every pair of inputs including
// P = Q, P = -Q and the identity (0 : 1 : 0).  The reference's ProjectivePoint.add/double (math.ts:974-1025) branch on those
// cases instead; the group elements produced are the same.
#pragma once
#include <cassert>
#include <vector>
#include "config.h"
#include "tower.h"

namespace nbls {

static inline SFp mat(const SFp& a) { return SFp(materialize(a)); }
// b3 = 3b: G1 b = 4, G2 b = 4(1+u)
static inline SFp mul_b3(const SFp& a) { return scale(a, 12); }
static inline SFp2 mul_b3(const SFp2& a) { return scale(mulnr(a), 12); }
static inline SFp f_zero(const SFp*) { return SFp(); }
static inline SFp2 f_zero(const SFp2*) { return fp2_zero(); }
static inline SFp f_one(const SFp*) { return fp_one(); }
static inline SFp2 f_one(const SFp2*) { return fp2_one(); }

template <class F> struct Pt { F x, y, z; };
template <class F> static inline Pt<F> pt_identity() { return {f_zero((F*)0), f_one((F*)0), f_zero((F*)0)}; }
template <class F> static inline Pt<F> pt_affine(const F& x, const F& y) { return {x, y, f_one((F*)0)}; }
template <class F> static inline Pt<F> pt_neg(const Pt<F>& p) { return {p.x, -p.y, p.z}; }
template <class F> static inline Pt<F> pt_mat(const Pt<F>& p) { return {mat(p.x), mat(p.y), mat(p.z)}; }

// RCB16 algorithm 7 (complete addition, a = 0), regrouped into sums of products
template <class F> static inline Pt<F> pt_add(const Pt<F>& p, const Pt<F>& q) {
  F t0 = mat(mul(p.x, q.x)), t1 = mat(mul(p.y, q.y)), t2 = mat(mul(p.z, q.z));
  F t3 = mat(mul(p.x, q.y) + mul(q.x, p.y));
  F t4 = mat(mul(p.y, q.z) + mul(q.y, p.z));
  F ty = mat(mul(p.x, q.z) + mul(q.x, p.z));
  F b2 = mat(mul_b3(t2));
  F z3 = mat(t1 + b2), s1 = mat(t1 - b2), y3 = mat(mul_b3(ty)), x0 = mat(scale(t0, 3));
  return pt_mat<F>({mul(t3, s1) - mul(t4, y3), mul(s1, z3) + mul(y3, x0), mul(z3, t4) + mul(x0, t3)});
}
// add-1998-cmo-2 exactly as the reference's ProjectivePoint.add generic branch (math.ts:1008-1024): valid on ANY short
// Weierstrass curve (no curve coefficient appears), used for the one addition on the isogenous curve E' in hash-to-G2.
// The equal / opposite / zero branches of the reference (math.ts:1000-1013) are not reproduced: for two independent
// SWU outputs they are cryptographically unreachable.
template <class F> static inline Pt<F> pt_add_generic(const Pt<F>& p1, const Pt<F>& p2) {
  F U1 = mat(mul(p2.y, p1.z)), U2 = mat(mul(p1.y, p2.z)), V1 = mat(mul(p2.x, p1.z)), V2 = mat(mul(p1.x, p2.z));
  F U = mat(U1 - U2), V = mat(V1 - V2);
  F VV = mat(sqr(V)), W = mat(mul(p1.z, p2.z));
  F VVV = mat(mul(VV, V)), V2VV = mat(mul(V2, VV));
  F A = mat(mul(mat(sqr(U)), W) - VVV - scale(V2VV, 2));
  return pt_mat<F>({mul(V, A), mul(U, V2VV - A) - mul(VVV, U2), mul(VVV, W)});
}
// RCB16 algorithm 9 (complete doubling, a = 0)
template <class F> static inline Pt<F> pt_dbl(const Pt<F>& p) {
  F t0 = mat(sqr(p.y)), t1 = mat(mul(p.y, p.z)), t2 = mat(mul_b3(sqr(p.z))), xy = mat(mul(p.x, p.y));
  F a = mat(t0 - scale(t2, 3)), b = mat(t0 + t2);
  return pt_mat<F>({scale(mul(a, xy), 2), mul(a, b) + scale(mul(t0, t2), 8), scale(mul(t0, t1), 8)});
}
// The same doubling over Fp regrouped into two levels of products around one level of sums: 12 z^2 is ONE lane-op (3 * (2z)(2z)),
// the factor 8 is carried by e = 2 t0 (a sum) and 4 e t = (e + e)(t + t), so no product has to be evaluated on its own
// before it can be scaled.  2 DOT + 1 LIN steps per doubling instead of 3 + 1 (doubling chains: subgroup checks, cofactor
// clearing, the ladders, the window shifts of the MSM).  Same projective point up to nothing: identical coordinates.
template <> inline Pt<SFp> pt_dbl<SFp>(const Pt<SFp>& p) {
  SFp t0 = mat(sqr(p.y)), t1 = mat(mul(p.y, p.z)), t2 = mat(mul_b3(sqr(p.z))), xy = mat(mul(p.x, p.y));
  SFp a = mat(t0 - scale(t2, 3)), b = mat(t0 + t2), e = mat(scale(t0, 2));
  return pt_mat<SFp>({scale(mul(a, xy), 2), mul(a, b) + scale(mul(e, t2), 4), scale(mul(e, t1), 4)});
}
// n successive doublings.  Over Fp2 the constant 3b = 12(1 + u) does not fit one lane-op (coefficients 12, 12, 24 on z0^2, z1^2,
// z0 z1), but 3(1 + u) d^2 with d = 2z does (3, 3, 6 = multiplier 3, one operand doubled): a run of doublings therefore keeps
// (x, y, d = 2z), on which a doubling is again two levels of products around one level of sums; t1 = y d = 2 y z, and the new
// d is 2 * 8 t0 (y z) = 2 e t1 with e = 4 t0.  Entering costs one sum (d = z + z), leaving one (x, y doubled: (2x : 2y : d) is
// the same projective point as (x : y : d / 2)).  A single doubling stays with pt_dbl.
template <class F> static inline Pt<F> pt_dbl_n(const Pt<F>& p, int n) { Pt<F> r = p; for (int i = 0; i < n; i++) r = pt_dbl(r); return r; }
// Over Fp the second level of a doubling has three results, x' (one product), y' = a b + 8 t0 t2 (two) and z' (one), on four lanes: the wavefront walks a
// two-product round for the sake of one lane.  With y' = t0^2 + 6 t0 t2 - 3 t2^2 = S^2 - 12 t2^2 (S = t0 + 3 t2; 3 is a non-residue, so it stays two
// products) the two squares go to two lanes, U = S^2 and V = 12 t2^2 = 3 (2 t2)(2 t2), and y' = U - V is never formed: every use of y in the next doubling
// is a product operand (y^2, y z, x y), where a two-term form is a pre-addition.  Four lanes x one product in both levels; y is summed once, when the run ends.
template <> inline Pt<SFp> pt_dbl_n<SFp>(const Pt<SFp>& p, int n) {
  static const bool lazy_y = !env_set("NBLS_DBL_PLAIN");
  if (n < 4 || !lazy_y) { Pt<SFp> r = p; for (int i = 0; i < n; i++) r = pt_dbl(r); return r; }   // short runs (the 3-bit windows of the ladders): the closing sum costs a step, measured slower
  SFp x = p.x, y = p.y, z = p.z;
  for (int i = 0; i < n; i++) {
    SFp t0 = mat(sqr(y)), t1 = mat(mul(y, z)), t2 = mat(mul_b3(sqr(z))), xy = mat(mul(x, y));
    SFp a = mat(t0 - scale(t2, 3)), S = mat(t0 + scale(t2, 3));
    SFp nx = mat(scale(mul(a, xy), 2)), U = mat(sqr(S)), V = mat(scale(sqr(t2), 12)), nz = mat(scale(mul(t0, t1), 8));
    x = nx; y = U - V; z = nz;
  }
  return {x, mat(y), z};
}
template <> inline Pt<SFp2> pt_dbl_n<SFp2>(const Pt<SFp2>& p, int n) {
  if (n < 2) return n ? pt_dbl(p) : p;
  SFp2 x = p.x, y = p.y, d = mat(scale(p.z, 2));
  for (int i = 0; i < n; i++) {
    // n2 = -t2 = -3(1 + u) d^2: as POSITIVE terms of the sums a = t0 + 3 n2 and b' = -b = n2 - t0 it needs no bound contraction
    // (a subtracted term must stay below 6p, and this lane-op's result is bounded by ~6.1p)
    SFp2 t0 = mat(sqr(y)), t1 = mat(mul(y, d)), n2 = mat(-scale(mulnr(sqr(d)), 3)), xy = mat(mul(x, y));
    static const bool two_squares = !env_set("NBLS_DBL_PLAIN");
    if (two_squares) {
      // y' = -a nb - 8 t0 n2 = t0^2 - 6 t0 n2 - 3 n2^2 = S^2 - 12 n2^2 with S = t0 - 3 n2: a difference of two Fp2 SQUARES costs two limb products per
      // coefficient where the sum of two Fp2 products costs four -- the second level of a doubling is then ONE product round instead of two.  The
      // factor 12 rides on sums: 12 (q0^2 - q1^2) = 4 (q0 + q1)(3 q0 - 3 q1), 24 q0 q1 = 8 q0 (3 q1) (x 4 = both single-slot operands doubled).
      SFp2 a = mat(t0 + scale(n2, 3)), S = mat(t0 - scale(n2, 3));
      SFp P = SFp(materialize(n2.c0 + n2.c1)), M3 = SFp(materialize(scale(n2.c0 - n2.c1, 3))), K = SFp(materialize(scale(n2.c1, 3)));
      SFp2 nx = mat(scale(mul(a, xy), 2));
      SFp2 ny = mat(SFp2{mul(S.c0 + S.c1, S.c0 - S.c1) - scale(mul(P, M3), 4), scale(mul(S.c0, S.c1), 2) - scale(mul(n2.c0, K), 8)});
      SFp2 nd = mat(scale(mul(t0, t1), 8));
      x = nx; y = ny; d = nd;
      continue;
    }
    SFp2 a = mat(t0 + scale(n2, 3)), nb = mat(n2 - t0), e = mat(scale(t0, 4));
    SFp2 nx = mat(scale(mul(a, xy), 2)), ny = mat(-mul(a, nb) - scale(mul(e, n2), 2)), nd = mat(scale(mul(e, t1), 2));
    x = nx; y = ny; d = nd;
  }
  return {mat(scale(x, 2)), mat(scale(y, 2)), d};
}
// [k]P for a public 64-bit constant k, MSB-first double-and-add (the reference's multiplyUnsafe, math.ts:1048-1058, is
// LSB-first; the group element is the same)
template <class F> static inline Pt<F> pt_mul_u64(const Pt<F>& p, uint64_t k) {
  Pt<F> r = p; int top = 63; while (top > 0 && !((k >> top) & 1)) top--;
  int run = 0;
  for (int i = top - 1; i >= 0; i--) { run++; if ((k >> i) & 1) { r = pt_add(pt_dbl_n(r, run), p); run = 0; } }
  return pt_dbl_n(r, run);
}
// [k]P for a per-item scalar held in a raw integer slot: fixed WIN-bit windows, MSB first.  Per window: WIN doublings, a
// table entry [0 .. 2^WIN - 1]P picked by a binary tree of masked selects on the scalar bits (every entry is read, the K_SEL
// lane-op merges both candidates under a mask), one complete addition (the entry may be the identity).  Instruction stream
// and LDS access pattern are independent of k (the reference's constant-time path is wNAF with precomputes,
// math.ts:1116-1157; the group element is the same).  The complete formulas make the identity start value and every
// intermediate case valid.  Round 4: WIN = 2 in both groups -- the table lives in LDS for the whole ladder, and with eight entries G1_MUL held 42 slots x 16 items
// (three workgroups per CU) and G2_MUL 72 x 8 (four): a quarter to a third of the wavefronts a SIMD can hold.  Four entries: 22 / 39 slots, seven workgroups per CU, at the
// price of 128 instead of 85 additions (+11 % instructions).
template <class F> static inline F sel(const SFp& f, const F& a, const F& b);
template <> inline SFp sel<SFp>(const SFp& f, const SFp& a, const SFp& b) { return select(f, a, b); }
template <> inline SFp2 sel<SFp2>(const SFp& f, const SFp2& a, const SFp2& b) { return {select(f, a.c0, b.c0), select(f, a.c1, b.c1)}; }
template <class F> static inline Pt<F> pt_sel(const SFp& f, const Pt<F>& a, const Pt<F>& b) { return {sel<F>(f, a.x, b.x), sel<F>(f, a.y, b.y), sel<F>(f, a.z, b.z)}; }
template <class F> static inline Pt<F> pt_mul_ladder(const Pt<F>& p, const SFp& k_raw, int nbits, const int WIN = 3) {
  assert(WIN == 2 || WIN == 3);
  Pt<F> T[8];
  T[0] = pt_mat(pt_identity<F>()); T[1] = pt_mat(p); T[2] = pt_dbl(p);
  for (int j = 3; j < (1 << WIN); j++) T[j] = pt_add(T[j - 1], T[1]);
  Pt<F> r = T[0];
  int hi = nbits;
  const int top = nbits % WIN;                       // a short leading window (256 = 1 + 85 * 3)
  if (top) {
    assert(top == 1);
    r = pt_sel<F>(bit_flag(k_raw, nbits - 1), T[1], T[0]);
    hi = nbits - 1;
  }
  for (int lo = hi - WIN; lo >= 0; lo -= WIN) {
    r = pt_dbl_n(r, WIN);
    SFp b0 = bit_flag(k_raw, lo), b1 = bit_flag(k_raw, lo + 1);
    Pt<F> u0 = pt_sel<F>(b0, T[1], T[0]), u1 = pt_sel<F>(b0, T[3], T[2]);
    Pt<F> v0 = pt_sel<F>(b1, u1, u0);
    if (WIN == 3) {
      SFp b2 = bit_flag(k_raw, lo + 2);
      Pt<F> u2 = pt_sel<F>(b0, T[5], T[4]), u3 = pt_sel<F>(b0, T[7], T[6]);
      Pt<F> v1 = pt_sel<F>(b1, u3, u2);
      v0 = pt_sel<F>(b2, v1, v0);
    }
    r = pt_add(r, v0);
  }
  return r;
}
// [k]G for the FIXED base point G1.BASE (getPublicKey, index.ts:738-740; PointG1.fromPrivateKey 350-353) with no doubling at all: k = sum_w d_w 2^(WIN w) over WIN-bit
// digits, [k]G = sum_w [d_w 2^(WIN w)]G with the multiples [d 2^(WIN w)]G, d = 1 .. 2^WIN - 1, read from a table in HBM that every item shares (buffer `buf`, entry (2^WIN - 1) w + d - 1:
// raw x, y, z = 1; built once per context on the device by the ladder above, pipelines_codec.cpp ensure_g1_fixed).  Per window: ALL entries are loaded and a binary tree of
// masked selects on the four scalar bits picks one (digit 0: the identity (0 : 1 : 0)), then one complete addition -- the instruction stream, the LDS accesses and the global
// addresses are the same for every key, as in the ladder.  With 3-bit windows (the default: seven entries per window keep the LDS image small enough for seven workgroups per CU; 4-bit windows hold 35+ slots per item) 86 additions where the 2-bit-window ladder spends 256 doublings and 128 additions.
static const int G1_FIXED_WIN = (int)env_long("NBLS_G1FIXED_WIN", 3);   // bits per window of the fixed-base table (pipelines_codec.cpp builds the table for the same value)
static inline int g1_fixed_windows() { return (256 + G1_FIXED_WIN - 1) / G1_FIXED_WIN; }
static inline int g1_fixed_entries() { return (1 << G1_FIXED_WIN) - 1; }
static inline Pt<SFp> pt_mul_fixed_g1(const SFp& k_raw, int buf) {
  const int WIN = G1_FIXED_WIN, NW = g1_fixed_windows(), NE = g1_fixed_entries();
  Pt<SFp> r = pt_mat(pt_identity<SFp>());
  const SFp one = mat(fp_one()), zero = mat(SFp());
  for (int w = 0; w < NW; w++) {
    const int nb = std::min(WIN, 256 - WIN * w);           // the top window may be short
    std::vector<SFp> b(nb); for (int i = 0; i < nb; i++) b[i] = bit_flag(k_raw, WIN * w + i);
    const int m0 = 1 << nb;
    std::vector<SFp> X(m0), Y(m0);
    X[0] = zero; Y[0] = one;
    for (int d = 1; d < m0; d++) { const int o = 144 * (NE * w + d - 1); X[d] = inputw(buf, o); Y[d] = inputw(buf, o + 48); }
    for (int lvl = 0, m = m0; lvl < nb; lvl++, m /= 2)
      for (int j = 0; j < m / 2; j++) { X[j] = select(b[lvl], X[2 * j + 1], X[2 * j]); Y[j] = select(b[lvl], Y[2 * j + 1], Y[2 * j]); }
    SFp z = zero; for (int i = 0; i < nb; i++) z = select(b[i], one, z);   // digit 0: the identity (0 : 1 : 0)
    r = pt_add(r, Pt<SFp>{X[0], Y[0], z});
  }
  return r;
}
// projective equality flags (math.ts:915-927): X1 Z2 == X2 Z1 and Y1 Z2 == Y2 Z1
static inline SFp eq_zero(const SFp& d) { return is_zero(d); }
static inline SFp eq_zero(const SFp2& d) { return f_and(is_zero(d.c0), is_zero(d.c1)); }
template <class F> static inline SFp pt_equal(const Pt<F>& a, const Pt<F>& b) {
  return f_and(eq_zero(mul(a.x, b.z) - mul(b.x, a.z)), eq_zero(mul(a.y, b.z) - mul(b.y, a.z)));
}
static inline SFp pt_is_identity(const Pt<SFp>& a) { return is_zero(a.z); }
static inline SFp pt_is_identity(const Pt<SFp2>& a) { return eq_zero(a.z); }

// ---- reference validity checks on affine inputs
// PointG1.isOnCurve / isTorsionFree (index.ts:408-448): y^2 = x^3 + 4 ;  -[x^2]P == phi(P), phi(x,y) = (beta x, y)
static inline void g1_validity_flags(const SFp& x, const SFp& y, SFp& on_curve, SFp& in_subgroup) {
  SFp x2 = mat(sqr(x));
  on_curve = is_zero(sqr(y) - mul(x2, x) - scale(fp_one(), 4));
  Pt<SFp> P = pt_affine(x, y);
  Pt<SFp> xP = pt_neg(pt_mul_u64(P, NBLS_X));           // mulCurveX
  Pt<SFp> u2P = pt_mul_u64(xP, NBLS_X);                 // mulCurveMinusX
  Pt<SFp> phi = pt_affine(mat(mul(x, fp_const(NBLS_BETA))), y);
  in_subgroup = pt_equal(u2P, phi);
}
// psi on an affine point (math.ts:1398-1403), reduced to conj(x) * PSI_X, conj(y) * PSI_Y (tools/gen_consts.py)
static inline void psi_affine(const SFp2& x, const SFp2& y, SFp2& px, SFp2& py) {
  px = mat(mul(conj(x), fp2_const(NBLS_PSI_X))); py = mat(mul(conj(y), fp2_const(NBLS_PSI_Y)));
}
// PointG2.isOnCurve / isTorsionFree (index.ts:675-690): y^2 = x^3 + 4(1+u) ;  [-x]P == psi(P)
static inline void g2_validity_flags(const SFp2& x, const SFp2& y, SFp& on_curve, SFp& in_subgroup) {
  SFp2 x2 = mat(sqr(x));
  SFp2 four = {scale(fp_one(), 4), scale(fp_one(), 4)};
  on_curve = eq_zero(sqr(y) - mul(x2, x) - four);
  Pt<SFp2> P = pt_affine(x, y);
  Pt<SFp2> xP = pt_neg(pt_mul_u64(P, NBLS_X));
  SFp2 px, py; psi_affine(x, y, px, py);
  in_subgroup = pt_equal(xP, pt_affine(px, py));
}

}  // namespace nbls
