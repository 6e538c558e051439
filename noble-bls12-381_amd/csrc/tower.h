// tower.h -- Fp2 / Fp6 / Fp12 / curve arithmetic over symbolic Fp values (trace.h).
// Each routine states which reference routine it is algebraically identical to (math.ts line numbers); the way
// an individual product is evaluated is free (field elements are unique), but wherever the reference's result
// depends on a projective representative (Miller-loop point R and its line coefficients) the same polynomial
// formulas are used so that pairing(P, Q, false) is bit-exact too.
#pragma once
#include "trace.h"
#include "config.h"
#include "consts_gen.h"

namespace nbls {

struct SFp2 { SFp c0, c1; };
struct SFp6 { SFp2 c0, c1, c2; };
struct SFp12 { SFp6 c0, c1; };

static inline SFp fp_const(const u32* m) { return constant(m); }
static inline SFp2 fp2_const(const u32 m[2][NLIMBS]) { return {constant(m[0]), constant(m[1])}; }
static inline SFp fp_one() { return constant(NBLS_R1); }
static inline SFp2 fp2_one() { return {fp_one(), SFp()}; }
static inline SFp2 fp2_zero() { return {SFp(), SFp()}; }


// ---- Fp2 (math.ts:403-550).  Products stay pending (lazy); an output coefficient is evaluated by one DOT lane-op.
static inline SFp2 operator+(const SFp2& a, const SFp2& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
static inline SFp2 operator-(const SFp2& a, const SFp2& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
static inline SFp2 operator-(const SFp2& a) { return {-a.c0, -a.c1}; }
static inline SFp2 scale(const SFp2& a, int k) { return {scale(a.c0, k), scale(a.c1, k)}; }
static inline SFp2 mul(const SFp2& a, const SFp2& b) {            // = math.ts:451-462 (schoolbook; same element as Karatsuba)
  return {mul(a.c0, b.c0) - mul(a.c1, b.c1), mul(a.c0, b.c1) + mul(a.c1, b.c0)};
}
static inline SFp2 sqr(const SFp2& a) {                           // math.ts:477-484
  return {mul(a.c0 + a.c1, a.c0 - a.c1), mul(scale(a.c0, 2), a.c1)};
}
static inline SFp2 mul_fp(const SFp2& a, const SFp& k) { return {mul(a.c0, k), mul(a.c1, k)}; }
static inline SFp2 mulnr(const SFp2& a) { return {a.c0 - a.c1, a.c0 + a.c1}; }              // * (u+1), math.ts:471-475
static inline SFp2 mul_by_b(const SFp2& a) { return scale(mulnr(a), 4); }                   // * 4(1+u), math.ts:532-539
static inline SFp2 conj(const SFp2& a) { return {a.c0, -a.c1}; }
static inline SFp2 halve(const SFp2& a) { return {halve(a.c0), halve(a.c1)}; }
static inline SFp2 frob(const SFp2& a, int power) { return (power & 1) ? conj(a) : a; }    // math.ts:529-531
static inline SFp2 mat(const SFp2& a) { return {SFp(materialize(a.c0)), SFp(materialize(a.c1))}; }

// ---- Fp6 (math.ts:554-700).  Schoolbook over Fp2 with the non-residue folded into an operand: xi*(x*y) = (xi*x)*y and
// xi*x = (x.c0 - x.c1, x.c0 + x.c1) is a fused pre-addition, so every output coefficient is a sum of 3 Fp2 products.
static inline SFp6 operator+(const SFp6& a, const SFp6& b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
static inline SFp6 operator-(const SFp6& a, const SFp6& b) { return {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; }
static inline SFp6 operator-(const SFp6& a) { return {-a.c0, -a.c1, -a.c2}; }
static inline SFp6 scale(const SFp6& a, int k) { return {scale(a.c0, k), scale(a.c1, k), scale(a.c2, k)}; }
static inline SFp6 mul(const SFp6& a, const SFp6& b) {            // = math.ts:601-618
  SFp2 xa1 = mulnr(a.c1), xa2 = mulnr(a.c2);
  return {mul(a.c0, b.c0) + mul(xa1, b.c2) + mul(xa2, b.c1),
          mul(a.c0, b.c1) + mul(a.c1, b.c0) + mul(xa2, b.c2),
          mul(a.c0, b.c2) + mul(a.c1, b.c1) + mul(a.c2, b.c0)};
}
static inline SFp6 mulnr(const SFp6& a) { return {mulnr(a.c2), a.c0, a.c1}; }               // * v, math.ts:627-629
static inline SFp6 mul_by_1(const SFp6& a, const SFp2& b1) {      // = math.ts:631-637
  return {mul(mulnr(a.c2), b1), mul(a.c0, b1), mul(a.c1, b1)};
}
static inline SFp6 mul_by_01(const SFp6& a, const SFp2& b0, const SFp2& b1) {   // = math.ts:639-651
  return {mul(a.c0, b0) + mul(mulnr(a.c2), b1), mul(a.c0, b1) + mul(a.c1, b0), mul(a.c1, b1) + mul(a.c2, b0)};
}
static inline SFp6 mul_by_fp2(const SFp6& a, const SFp2& k) { return {mul(a.c0, k), mul(a.c1, k), mul(a.c2, k)}; }
static inline SFp6 sqr(const SFp6& a) {                           // = math.ts:658-670
  return {sqr(a.c0) + scale(mul(mulnr(a.c1), a.c2), 2), scale(mul(a.c0, a.c1), 2) + mulnr(sqr(a.c2)), sqr(a.c1) + scale(mul(a.c0, a.c2), 2)};
}
static inline SFp6 frob(const SFp6& a, int power) {               // math.ts:682-688
  return {frob(a.c0, power), mul(frob(a.c1, power), fp2_const(NBLS_FROB6_1[power % 6])), mul(frob(a.c2, power), fp2_const(NBLS_FROB6_2[power % 6]))};
}
static inline SFp6 mat(const SFp6& a) { return {mat(a.c0), mat(a.c1), mat(a.c2)}; }

// ---- Fp12 (math.ts:705-885)
static inline SFp12 fp12_one() { return {{fp2_one(), fp2_zero(), fp2_zero()}, {fp2_zero(), fp2_zero(), fp2_zero()}}; }
static inline SFp12 mat(const SFp12& a) { return {mat(a.c0), mat(a.c1)}; }
static inline SFp12 mul(const SFp12& a, const SFp12& b) {         // math.ts:748-759 (Karatsuba over Fp6, three 18-lane products)
  SFp6 t1 = mat(mul(a.c0, b.c0)), t2 = mat(mul(a.c1, b.c1));
  SFp6 sa = mat(a.c0 + a.c1), sb = mat(b.c0 + b.c1);
  SFp6 v = mul(sa, sb);
  // The middle product alone would be six lanes of six limb products (three rounds) beside six idle lanes: every coefficient is cut into two lane-ops of
  // three products on all twelve lanes (a round and a half), and the halves meet in the sum that forms c1 anyway (v - t1 - t2 becomes vl + vh - t1 - t2).
  static const bool split_mid = !env_set("NBLS_MUL12_PLAIN");
  if (split_mid) {
    // each half takes one of the subtracted terms as a post-subtraction of its lane-op (offset inside the accumulator), so the closing sum has two positive terms and needs no k p constant
    auto halves = [](const SFp& x, const SFp& s1, const SFp& s2) {
      const size_t h = (x.f.size() + 1) / 2;
      SFp lo, hi;
      lo.f.assign(x.f.begin(), x.f.begin() + h); hi.f.assign(x.f.begin() + h, x.f.end());
      return SFp(materialize(lo - s1)) + SFp(materialize(hi - s2));
    };
    auto halves2 = [&](const SFp2& x, const SFp2& s1, const SFp2& s2) { return SFp2{halves(x.c0, s1.c0, s2.c0), halves(x.c1, s1.c1, s2.c1)}; };
    return {t1 + mulnr(t2), {halves2(v.c0, t1.c0, t2.c0), halves2(v.c1, t1.c1, t2.c1), halves2(v.c2, t1.c2, t2.c2)}};
  }
  return {t1 + mulnr(t2), v - (t1 + t2)};
}
// f * (o0 + o1 v + o4 v w): every output coefficient is 3 Fp2 products of f with the line  (= math.ts:768-777)
static inline SFp12 mul_by_014(const SFp12& a, const SFp2& o0, const SFp2& o1, const SFp2& o4) {
  const SFp6 &x = a.c0, &y = a.c1;
  SFp2 xx2 = mulnr(x.c2), xy1 = mulnr(y.c1), xy2 = mulnr(y.c2);
  return {{mul(x.c0, o0) + mul(xx2, o1) + mul(xy1, o4), mul(x.c0, o1) + mul(x.c1, o0) + mul(xy2, o4), mul(x.c1, o1) + mul(x.c2, o0) + mul(y.c0, o4)},
          {mul(y.c0, o0) + mul(xy2, o1) + mul(xx2, o4), mul(y.c0, o1) + mul(y.c1, o0) + mul(x.c0, o4), mul(y.c1, o1) + mul(y.c2, o0) + mul(x.c1, o4)}};
}
static inline SFp12 sqr(const SFp12& a) {                         // = math.ts:783-791: (c0^2 + v c1^2, 2 c0 c1)
  // v * c1^2 = (xi * w.c2, w.c0, w.c1) with w = c1^2; xi * w.c2 is written with the non-residue folded into operands
  // (xi * (2 x y) = 2 (xi x) y: two products per coefficient where xi applied to the pending form would take four)
  const SFp6& b = a.c1;
  SFp6 s = sqr(a.c0), w = sqr(b);
  SFp2 xw2 = mulnr(sqr(b.c1)) + scale(mul(mulnr(b.c0), b.c2), 2);
  return {{s.c0 + xw2, s.c1 + w.c0, s.c2 + w.c1}, scale(mul(a.c0, a.c1), 2)};
}
static inline SFp12 conj(const SFp12& a) { return {a.c0, -a.c1}; }                             // math.ts:799-801
static inline SFp12 frob(const SFp12& a, int power) {             // math.ts:804-809
  return {frob(a.c0, power), mul_by_fp2(frob(a.c1, power), fp2_const(NBLS_FROB12[power % 12]))};
}
static inline void fp4_square(const SFp2& a, const SFp2& b, SFp2& first, SFp2& second) {   // = math.ts:811-818
  first = mulnr(sqr(b)) + sqr(a);
  second = scale(mul(a, b), 2);       // (a + b)^2 - a^2 - b^2
}
static inline SFp12 cyclotomic_sqr(const SFp12& x) {              // math.ts:824-843
  SFp2 t3, t4, t5, t6, t7, t8;
  fp4_square(x.c0.c0, x.c1.c1, t3, t4);
  fp4_square(x.c1.c0, x.c0.c2, t5, t6);
  fp4_square(x.c0.c1, x.c1.c2, t7, t8);
  // (1 + u) * t8 with the non-residue folded into an operand ((xi a) b: two products per coefficient; xi applied to the pending
  // product form would expand to four) -- the same field element
  SFp2 t9 = scale(mul(mulnr(x.c0.c1), x.c1.c2), 2);
  (void)t8;
  return {{scale(t3 - x.c0.c0, 2) + t3, scale(t5 - x.c0.c1, 2) + t5, scale(t7 - x.c0.c2, 2) + t7},
          {scale(t9 + x.c1.c0, 2) + t9, scale(t4 + x.c1.c1, 2) + t4, scale(t6 + x.c1.c2, 2) + t6}};
}
// The same squaring on the TRIPLED element (round 3).  Every output of the Granger-Scott squaring is 3 t -+ 2 g with t quadratic in the input; the
// factor 3 is a multiplier on the reduced sum (28 instructions per lane-op) and forces a second carry pass.  With G = 3 g carried instead of g:
//   3 (3 t(g) -+ 2 g) = 9 t(g) -+ 6 g = t(G) -+ 2 G            (t is quadratic: t(3 g) = 9 t(g))
// so the tripled state squares WITHOUT a multiplier, and a product with an untripled base keeps the factor ((3 g) b = 3 (g b)).  cyclotomic_exp_x
// triples its input once and multiplies by 1/3 at the end; the element computed is the same.
static inline SFp12 cyclotomic_sqr_tripled(const SFp12& x) {
  SFp2 t3, t4, t5, t6, t7, t8;
  fp4_square(x.c0.c0, x.c1.c1, t3, t4);
  fp4_square(x.c1.c0, x.c0.c2, t5, t6);
  fp4_square(x.c0.c1, x.c1.c2, t7, t8);
  SFp2 t9 = scale(mul(mulnr(x.c0.c1), x.c1.c2), 2);
  (void)t8;
  return {{t3 - scale(x.c0.c0, 2), t5 - scale(x.c0.c1, 2), t7 - scale(x.c0.c2, 2)},
          {t9 + scale(x.c1.c0, 2), t4 + scale(x.c1.c1, 2), t6 + scale(x.c1.c2, 2)}};
}
// ---- Karabina's compressed squaring (round 3; "Squaring in cyclotomic subgroups", Math. Comp. 82 (2013)).  In the Fp4 view of the tower
// (x = A + B w + C w^2 over Fp4 = Fp2[s], s = w^3: A = (c0.c0, c1.c1), B = (c1.c0, c0.c2), C = (c0.c1, c1.c2)) the Granger-Scott squaring reads
//   A' = 3 A^2 - 2 conj(A),   B' = 3 s C^2 + 2 conj(B),   C' = 3 B^2 - 2 conj(C)
// so the four Fp2 coordinates of (B, C) square among themselves, and A can be recovered from them at the end (decompression, one Fp2 inversion):
//   g1 = (xi g5^2 + 3 g4^2 - 2 g3) / (4 g2),   g0 = (2 g1^2 + g2 g5 - 3 g3 g4) xi + 1      with g0 = c0.c0, g1 = c1.c1, g2 = c1.c0, g3 = c0.c2, g4 = c0.c1, g5 = c1.c2
// (valid while g2 != 0; items where some g2 vanishes are flagged and recomputed by the plain program).  A run of squarings on (g2, g3, g4, g5) is
// 8 lane-ops per item instead of 12: eight items per wavefront instead of five at the same instructions per step.  The same result as math.ts:824-852.
struct SCyc4 { SFp2 g2, g3, g4, g5; };
static inline SCyc4 mat(const SCyc4& a) { return {mat(a.g2), mat(a.g3), mat(a.g4), mat(a.g5)}; }
// on the tripled state (see cyclotomic_sqr_tripled): no multiplier
static inline SCyc4 compressed_sqr_tripled(const SCyc4& s) {
  SFp2 t5, t6, t7, t8;
  fp4_square(s.g2, s.g3, t5, t6);      // B^2
  fp4_square(s.g4, s.g5, t7, t8);      // C^2
  SFp2 t9 = scale(mul(mulnr(s.g4), s.g5), 2);   // xi * t8 with the non-residue folded into an operand
  (void)t8;
  return {t9 + scale(s.g2, 2), t7 - scale(s.g3, 2), t5 - scale(s.g4, 2), t6 + scale(s.g5, 2)};
}
static inline SFp12 mul_fp(const SFp12& a, const SFp& k) {
  auto m2 = [&](const SFp2& v) { return SFp2{mul(v.c0, k), mul(v.c1, k)}; };
  return {{m2(a.c0.c0), m2(a.c0.c1), m2(a.c0.c2)}, {m2(a.c1.c0), m2(a.c1.c1), m2(a.c1.c2)}};
}
// z^|x| for unitary z (math.ts:845-852).  The reference starts from ONE and squares through all 64 bits; the
// leading squarings of ONE are identities, so starting at the top set bit gives the same element.
// `reload` (optional): fetches a again where it is multiplied in (five times): the base then does not occupy twelve LDS slots throughout the 63
// squarings (EXPX: 42 -> 30 slots, twelve instead of ten wavefronts per CU); the loads cost five cheap steps
template <class Reload>
static inline SFp12 cyclotomic_exp_x(const SFp12& a, Reload reload) {
  static const bool tripled = !env_set("NBLS_CYCSQR_PLAIN");   // A/B switch, read once per process (the old formulas: multiplier 3 in every squaring lane-op)
  if (!tripled) {
    SFp12 z = a;   // after bit 63
    for (int i = 62; i >= 0; i--) {
      z = mat(cyclotomic_sqr(z));
      if ((NBLS_X >> i) & 1) z = mat(mul(z, reload()));
    }
    return z;
  }
  Builder* B = Builder::cur();
  SFp12 z = mat(mul_fp(a, SFp(B->small_const(3))));   // 3 a as one product per coefficient (bound ~1; a sum a + a + a would carry three times the input's bound into the first squaring)
  for (int i = 62; i >= 0; i--) {
    z = mat(cyclotomic_sqr_tripled(z));
    if ((NBLS_X >> i) & 1) z = mat(mul(z, reload()));
  }
  return mul_fp(z, SFp(B->frac_const(1, 3)));
}
static inline SFp12 cyclotomic_exp_x(const SFp12& a) { return cyclotomic_exp_x(a, [&]() { return a; }); }

// I/O helpers: Fp12 in the reference's toBytes order (math.ts:875-884)
static inline SFp2 input_fp2(int buf, int off) { return {input(buf, off), input(buf, off + 48)}; }
static inline SFp12 input_fp12(int buf, int off) {
  SFp2 c[6]; for (int i = 0; i < 6; i++) c[i] = input_fp2(buf, off + 96 * i);
  return {{c[0], c[1], c[2]}, {c[3], c[4], c[5]}};
}
static inline void output_fp2(const SFp2& a, int buf, int off) { output(a.c0, buf, off); output(a.c1, buf, off + 48); }
static inline void output_fp12(const SFp12& a, int buf, int off) {
  const SFp2* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  for (int i = 0; i < 6; i++) output_fp2(*c[i], buf, off + 96 * i);
}
static inline SFp12 inputw_fp12(int buf, int off) {
  SFp2 c[6]; for (int i = 0; i < 6; i++) c[i] = {inputw(buf, off + 96 * i), inputw(buf, off + 96 * i + 48)};
  return {{c[0], c[1], c[2]}, {c[3], c[4], c[5]}};
}
static inline void outputw_fp12(const SFp12& a, int buf, int off) {
  const SFp2* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  for (int i = 0; i < 6; i++) { outputw(c[i]->c0, buf, off + 96 * i); outputw(c[i]->c1, buf, off + 96 * i + 48); }
}

}  // namespace nbls
