// codec.h -- symbolic restatements of the reference's point decoders and hash-to-G2 map (everything between the
// wire bytes and an affine curve point), for the wave VM.  Data-dependent choices are expressed with flag / select lane-ops.
#pragma once
#include "curve.h"

namespace nbls {

static inline SFp2 select2(const SFp& f, const SFp2& a, const SFp2& b) { return {select(f, a.c0, b.c0), select(f, a.c1, b.c1)}; }
// x (Montgomery form) -> canonical standard integer x/R mod p, for comparisons on the true value
static inline SFp std_canon(const SFp& x) {
  Builder* B = Builder::cur();
  Operand a; a.s0 = materialize(x); Operand b; b.s0 = B->rawone_atom;
  SFp r; r.f.push_back({PROD_BASE + B->product(a, b), 1});
  return canon(SFp(materialize(r)));
}
// floor(2 v / p) == 1  <=>  v > (p-1)/2      ("(y.value * 2n) / P", index.ts:314, 524-525)
static inline SFp gt_half(const SFp& v_std) { return cmp_gt(v_std, raw_const(NBLS_HALF_P_RAW)); }
static inline SFp2 fp2_b() { return {scale(fp_one(), 4), scale(fp_one(), 4)}; }
static inline SFp2 mul_i(const SFp2& a) { return {-a.c1, a.c0}; }

// ---------------------------------------------------------------- PointG1.toHex(true) / PointG2.toSignature of non-zero affine points
// (index.ts:359-371, 586-602): x plus the compression flag 2^383 and the sign flag 2^381 * floor(2y / p)
static inline void g1_compress(int in_buf, int out_buf) {
  SFp x = input_raw(in_buf, 0), y = input_raw(in_buf, 48);
  SFp flag = gt_half(y);
  output_raw(x + select(flag, raw_const(NBLS_POW2_381_RAW), SFp()) + raw_const(NBLS_POW2_383_RAW), out_buf, 0);
}
static inline void g2_compress(int in_buf, int out_buf) {
  SFp x0 = input_raw(in_buf, 0), x1 = input_raw(in_buf, 48), y0 = input_raw(in_buf, 96), y1 = input_raw(in_buf, 144);
  SFp flag = select(is_zero(y1), gt_half(y0), gt_half(y1));      // tmp = y1 > 0 ? y1 * 2 : y0 * 2
  output_raw(x1 + select(flag, raw_const(NBLS_POW2_381_RAW), SFp()) + raw_const(NBLS_POW2_383_RAW), out_buf, 0);
  output_raw(x0, out_buf, 48);
}

// ---------------------------------------------------------------- PointG1.fromHex, 48-byte compressed (index.ts:301-315, 325)
// phase A: x and x^3 + 4;  [kernel: (x^3+4)^((p+1)/4)];  phase B: check, sign, validity
static inline void g1_decompress_A(int in_buf, int x_buf, int rhs_buf) {
  SFp z = input_raw(in_buf, 0);
  SFp x = to_mont(bit_and(z, raw_const(NBLS_MASK381_RAW)));           // x = value mod 2^381 (reduced mod p by the Fp constructor)
  outputw(x, x_buf, 0);
  outputw(mul(mat(sqr(x)), x) + scale(fp_one(), 4), rhs_buf, 0);
}
static inline void g1_decompress_B(int in_buf, int x_buf, int rhs_buf, int cand_buf, int out_buf, int status_buf) {
  SFp z = input_raw(in_buf, 0);
  SFp inf = bit_flag(z, 382), aflag = bit_flag(z, 381);               // bflag / aflag (index.ts:305, 313)
  SFp x = inputw(x_buf, 0), rhs = inputw(rhs_buf, 0), y = inputw(cand_buf, 0);
  SFp ok = is_zero(sqr(y) - rhs);                                     // Fp.sqrt: root^2 == a (math.ts:262)
  SFp flip = f_xor(gt_half(std_canon(y)), aflag);
  SFp ysel = select(flip, -y, y);
  SFp oc, sg; g1_validity_flags(x, ysel, oc, sg);
  SFp not_inf = f_not(inf);
  status_out({{not_inf, 1}, {ok, 4}, {sg, 3}}, status_buf);
  SFp good = f_and(f_and(not_inf, ok), sg);
  output(select(good, x, SFp()), out_buf, 0);
  output(select(good, ysel, SFp()), out_buf, 48);
}

// ---------------------------------------------------------------- PointG2.fromSignature, 96-byte compressed (index.ts:500-530)
// sig192: the 192-byte branch of fromSignature (index.ts:500-515) reads z1 and z2 as 96-byte big-endian integers: x.c1 = z1 mod 2^381 (the low
// 381 bits: the second 48 bytes) and x.c0 = z2 mod p (a 768-bit value, reduced by the Fp constructor)
static inline void g2_decompress_A(int in_buf, int x_buf, int rhs_buf, bool sig192 = false) {
  Builder* B = Builder::cur();
  SFp z1 = input_raw(in_buf, sig192 ? 48 : 0);
  SFp x0;
  if (!sig192) x0 = to_mont(input_raw(in_buf, 48));
  else {   // (hi 2^384 + lo) R = REDC(hi * (2^384 R^2) + lo * R^2)
    Operand t; t.s0 = materialize(input_raw(in_buf, 96)); Operand r3; r3.s0 = B->const_atom(NBLS_TOP384);
    Operand l; l.s0 = materialize(input_raw(in_buf, 144)); Operand r2; r2.s0 = B->r2_atom;
    SFp f; f.f = form_add({{PROD_BASE + B->product(t, r3), 1}}, {{PROD_BASE + B->product(l, r2), 1}}, 1);
    x0 = SFp(materialize(f));
  }
  SFp2 x = {x0, to_mont(bit_and(z1, raw_const(NBLS_MASK381_RAW)))};   // x = Fp2(z2, z1 mod 2^381)
  outputw(x.c0, x_buf, 0); outputw(x.c1, x_buf, 48);
  SFp2 rhs = mul(mat(sqr(x)), x) + fp2_b();
  outputw(rhs.c0, rhs_buf, 0); outputw(rhs.c1, rhs_buf, 48);
}
// Fp2.sqrt (math.ts:486-507) given cand = a^((p^2+7)/16): returns the chosen root and the `found` flag
static inline SFp2 fp2_sqrt_finish(const SFp2& a, const SFp2& cand, SFp& found) {
  SFp2 c2 = mat(sqr(cand));
  SFp2 ia = mul_i(a);
  SFp f0 = eq_zero(c2 - a), f1 = eq_zero(c2 - ia), f2 = eq_zero(c2 + a), f3 = eq_zero(c2 + ia);   // cand^2 / a in {1, i, -1, -i}
  found = f_or(f_or(f0, f1), f_or(f2, f3));
  SFp2 x1_1 = mat(mul(cand, fp2_const(NBLS_ROOTS8_INV[1]))), x1_2 = mat(mul(cand, fp2_const(NBLS_ROOTS8_INV[2]))), x1_3 = mat(mul(cand, fp2_const(NBLS_ROOTS8_INV[3])));
  SFp2 x1 = select2(f0, cand, select2(f1, x1_1, select2(f2, x1_2, x1_3)));
  SFp2 x2 = mat(-x1);
  SFp re1 = std_canon(x1.c0), im1 = std_canon(x1.c1), re2 = std_canon(x2.c0), im2 = std_canon(x2.c1);
  SFp choose1 = f_or(cmp_gt(im1, im2), f_and(is_zero(x1.c1), cmp_gt(re1, re2)));     // im1 > im2 || (im1 == im2 && re1 > re2)
  return select2(choose1, x1, x2);
}
// mode 0: PointG2.fromSignature, 96 bytes (index.ts:500-530); 1: its 192-byte branch (flags in the second 48 bytes);
// 2: PointG2.fromHex on 96 compressed bytes (index.ts:532-562): flag rules first, the root by the S bit, NO subgroup check.
//    Fp2.sqrt returns the root with the larger imaginary part (real part on ties), whose Y_bit is 1, so "bitS && Y_bit ? y : -y" keeps
//    that root when S is set and takes the other one when it is clear.
//    status: 6 invalid encoding flag (0x20, 0x60, 0xe0), 8 compression bit clear, 7 infinity bit with other bits set, 1 zero point, 4 no square root
static inline void g2_decompress_B(int in_buf, int x_buf, int rhs_buf, int cand_buf, int out_buf, int status_buf, int mode = 0) {
  SFp z1 = input_raw(in_buf, mode == 1 ? 48 : 0);
  SFp inf = bit_flag(z1, 382), aflag = bit_flag(z1, 381);
  SFp2 x = {inputw(x_buf, 0), inputw(x_buf, 48)}, rhs = {inputw(rhs_buf, 0), inputw(rhs_buf, 48)}, cand = {inputw(cand_buf, 0), inputw(cand_buf, 48)};
  SFp found; SFp2 y = fp2_sqrt_finish(rhs, cand, found);
  SFp not_inf = f_not(inf);
  SFp2 zero = fp2_zero();
  if (mode == 2) {
    SFp cbit = bit_flag(z1, 383);
    SFp enc_ok = f_or(f_not(aflag), f_and(cbit, not_inf));                      // S set requires C set and I clear
    SFp rest = f_or(cmp_gt(bit_and(z1, raw_const(NBLS_MASK381_RAW)), SFp()), cmp_gt(input_raw(in_buf, 48), SFp()));   // any bit besides the flags
    SFp inf_ok = f_not(f_and(inf, rest));
    SFp2 ysel = select2(aflag, y, mat(-y));
    status_out({{enc_ok, 6}, {cbit, 8}, {inf_ok, 7}, {not_inf, 1}, {found, 4}}, status_buf);
    SFp good = f_and(f_and(f_and(enc_ok, cbit), not_inf), found);
    output_fp2(select2(good, x, zero), out_buf, 0);
    output_fp2(select2(good, ysel, zero), out_buf, 96);
    return;
  }
  SFp y1nz = f_not(is_zero(y.c1));
  SFp big1 = gt_half(std_canon(y.c1)), big0 = gt_half(std_canon(y.c0));
  SFp neg = f_or(f_and(y1nz, f_xor(big1, aflag)), f_and(f_not(y1nz), f_xor(big0, aflag)));   // isGreater || isZero (index.ts:524-526)
  SFp2 ysel = select2(neg, mat(-y), y);
  SFp oc, sg; g2_validity_flags(x, ysel, oc, sg);
  status_out({{not_inf, 1}, {found, 4}, {sg, 3}}, status_buf);
  SFp good = f_and(f_and(not_inf, found), sg);
  output_fp2(select2(good, x, zero), out_buf, 0);
  output_fp2(select2(good, ysel, zero), out_buf, 96);
}
// ---------------------------------------------------------------- uncompressed forms (index.ts:317-321, 563-575): coordinates as big-endian
// integers (reduced by the Fp constructor), infinity flag = bit 6 of the first byte, then assertValidity; G2 in the order x.c1 x.c0 y.c1 y.c0
static inline void g1_from_raw(int in_buf, int out_buf, int status_buf) {
  SFp zx = input_raw(in_buf, 0);
  SFp not_inf = f_not(bit_flag(zx, 382));
  SFp x = to_mont(zx), y = input(in_buf, 48), oc, sg;
  g1_validity_flags(x, y, oc, sg);
  status_out({{not_inf, 1}, {oc, 2}, {sg, 3}}, status_buf);
  SFp good = f_and(f_and(not_inf, oc), sg);
  output(select(good, x, SFp()), out_buf, 0); output(select(good, y, SFp()), out_buf, 48);
}
static inline void g2_from_raw(int in_buf, int out_buf, int status_buf) {
  // PointG2.fromHex applies its flag rules to 192-byte input as well (index.ts:534-537, 563): sign bit without the compression bit, or with
  // the infinity bit, is 'Invalid encoding flag' (6); the compression bit on 192 bytes falls through to 'Invalid point G2, expected 96/192
  // bytes' (8); only then the infinity bit (1).  So a coordinate x.c1 + k p whose VALUE reaches bit 381 is rejected, not reduced.
  SFp zx1 = input_raw(in_buf, 0);
  SFp sbit = bit_flag(zx1, 381), ibit = bit_flag(zx1, 382), cbit = bit_flag(zx1, 383);
  SFp enc_ok = f_or(f_not(sbit), f_and(cbit, f_not(ibit)));
  SFp not_c = f_not(cbit);
  SFp not_inf = f_not(ibit);
  SFp2 x = {input(in_buf, 48), to_mont(zx1)}, y = {input(in_buf, 144), input(in_buf, 96)};
  SFp oc, sg; g2_validity_flags(x, y, oc, sg);
  status_out({{enc_ok, 6}, {not_c, 8}, {not_inf, 1}, {oc, 2}, {sg, 3}}, status_buf);
  SFp good = f_and(f_and(f_and(enc_ok, not_c), f_and(not_inf, oc)), sg);
  SFp2 zero = fp2_zero();
  output_fp2(select2(good, x, zero), out_buf, 0); output_fp2(select2(good, y, zero), out_buf, 96);
}
// affine wire order (c0 || c1) <-> the order of PointG2.toHex(false) (c1 || c0), index.ts:622-629: its own inverse
static inline void g2_swap_halves(int in_buf, int out_buf) {
  for (int k = 0; k < 2; k++) { output_raw(input_raw(in_buf, 96 * k + 48), out_buf, 96 * k); output_raw(input_raw(in_buf, 96 * k), out_buf, 96 * k + 48); }
}

// ---------------------------------------------------------------- hash_to_field tail + SWU + isogeny + cofactor (index.ts:256-263, 481-490)
// os2ip(64 bytes) mod p: v = top16 * 2^384 + low48  ->  Montgomery form  v R = REDC(top16 * (2^384 R^2) + low48 * R^2)
static inline SFp field_elem_from_64(int buf, int off) {
  Builder* B = Builder::cur();
  Operand t; t.s0 = materialize(input_raw(buf, off, 16)); Operand r3; r3.s0 = B->const_atom(NBLS_TOP384);
  Operand l; l.s0 = materialize(input_raw(buf, off + 16, 48)); Operand r2; r2.s0 = B->r2_atom;
  SFp f; f.f = form_add({{PROD_BASE + B->product(t, r3), 1}}, {{PROD_BASE + B->product(l, r2), 1}}, 1);
  return SFp(materialize(f));
}
struct SwuState { SFp2 t, zt2, num, den, v, u, uv7, uv15; };
// map_to_curve_simple_swu_9mod16 up to the exponentiation input (math.ts:1220-1241, 1196-1198)
static inline SwuState swu_prepare(const SFp2& t) {
  SwuState s; s.t = t;
  SFp2 Z = fp2_const(NBLS_SWU_Z), A = fp2_const(NBLS_SWU_A), Bc = fp2_const(NBLS_SWU_B);
  SFp2 t2 = mat(sqr(t));
  s.zt2 = mat(mul(Z, t2));
  SFp2 ztzt = mat(s.zt2 + sqr(s.zt2));
  SFp2 den0 = mat(-mul(A, ztzt));
  s.num = mat(mul(Bc, ztzt + fp2_one()));
  SFp dz = eq_zero(den0);
  s.den = select2(dz, mat(mul(Z, A)), den0);                                  // exceptional case (math.ts:1233)
  SFp2 den2 = mat(sqr(s.den));
  s.v = mat(mul(den2, s.den));
  SFp2 num2 = mat(sqr(s.num));
  s.u = mat(mul(num2, s.num) + mul(mat(mul(A, s.num)), den2) + mul(Bc, s.v));
  SFp2 v2 = mat(sqr(s.v)), v4 = mat(sqr(v2)), v3 = mat(mul(v2, s.v)), v7 = mat(mul(v4, v3));
  s.uv7 = mat(mul(s.u, v7));
  s.uv15 = mat(mul(s.uv7, mat(mul(v7, s.v))));
  return s;
}
// the values swu_finish needs beside t, as twelve raw elements in HBM scratch between P_H2C_A and P_H2C_B1 (the state is cheaper to reload than to recompute:
// 76 of the 274 products of the round-3 H2C_B were swu_prepare run a second time)
static inline void swu_state_store(const SwuState& s, int buf, int off) {
  const SFp2* f[6] = {&s.zt2, &s.num, &s.den, &s.v, &s.u, &s.uv7};
  for (int k = 0; k < 6; k++) { outputw(f[k]->c0, buf, off + 96 * k); outputw(f[k]->c1, buf, off + 96 * k + 48); }
}
static inline SwuState swu_state_load(const SFp2& t, int buf, int off) {
  SwuState s; s.t = t;
  SFp2* f[6] = {&s.zt2, &s.num, &s.den, &s.v, &s.u, &s.uv7};
  for (int k = 0; k < 6; k++) *f[k] = {inputw(buf, off + 96 * k), inputw(buf, off + 96 * k + 48)};
  return s;
}
// sgn0_fp2 (math.ts:1179-1185) on Montgomery values
static inline SFp sgn0(const SFp2& x) {
  SFp s0 = is_odd(std_canon(x.c0)), z0 = is_zero(x.c0), s1 = is_odd(std_canon(x.c1));
  return f_or(s0, f_and(z0, s1));
}
// rest of map_to_curve_simple_swu_9mod16 (math.ts:1199-1266) given gp = uv15^((p^2-9)/16); returns the point projectively (X : Y : Z) = (num : y den : den)
static inline Pt<SFp2> swu_finish(const SwuState& s, const SFp2& gp) {
  SFp2 gamma = mat(mul(gp, s.uv7));
  SFp ok[4]; SFp2 cand[4];
  for (int k = 0; k < 4; k++) { cand[k] = k == 0 ? gamma : mat(mul(fp2_const(NBLS_ROOTS8[k]), gamma)); ok[k] = eq_zero(mul(mat(sqr(cand[k])), s.v) - s.u); }
  SFp success = f_or(f_or(ok[0], ok[1]), f_or(ok[2], ok[3]));
  SFp2 res = select2(ok[0], cand[0], select2(ok[1], cand[1], select2(ok[2], cand[2], select2(ok[3], cand[3], gamma))));
  SFp2 t3 = mat(mul(mat(sqr(s.t)), s.t));
  SFp2 x1c = mat(mul(res, t3));                                               // sqrt_candidate(x1) = sqrt_candidate(x0) * t^3
  SFp2 zt2_3 = mat(mul(mat(sqr(s.zt2)), s.zt2));
  SFp2 u2 = mat(mul(zt2_3, s.u));                                             // u(x1) = Z^3 t^6 u(x0)
  SFp ok2[4]; SFp2 ec[4];
  for (int k = 0; k < 4; k++) { ec[k] = mat(mul(fp2_const(NBLS_ETAS[k]), x1c)); ok2[k] = eq_zero(mul(mat(sqr(ec[k])), s.v) - u2); }
  SFp2 y2 = select2(ok2[0], ec[0], select2(ok2[1], ec[1], select2(ok2[2], ec[2], ec[3])));
  SFp2 y = select2(success, res, y2);
  SFp2 num = select2(success, s.num, mat(mul(s.num, s.zt2)));                   // success2: numerator *= Z t^2
  SFp flip = f_xor(sgn0(s.t), sgn0(y));
  y = select2(flip, mat(-y), y);
  return pt_mat<SFp2>({num, mul(y, s.den), s.den});
}
// ---- the SWU square root by the NORM method (round 6; the reference's sqrt_div_fp2, math.ts:1195-1214, is one 758-bit Fp2 exponentiation -- here 377 Fp2 squarings on a lane
// pair, pow_kernels.hip).  map_to_curve_simple_swu_9mod16 needs ANY y with y^2 v = u (or Z^3 t^6 u when u / v is not a square): it fixes the sign by sgn0 afterwards
// (math.ts:1264), so every method that finds a root returns the reference's point.  With a = u conj(v) and d = N(v) (u / v = a / d, d in Fp):
//   n = N(a)^((p+1)/4)                    first Fp exponentiation; n^2 = N(a) exactly when u / v is a square of Fp2.  Otherwise n^2 = -N(a), and the value whose root is
//                                         wanted is a' = a (Z t^2)^3 with N(a') = 125 N(t)^6 N(a) = (sqrt(-125) N(t)^3 n)^2 (N(Z) = 5 is a non-residue): no second test
//   delta = (a0 + n) / 2                  y0^2 = delta / d and y1 = a1 / (2 d y0) solve (y0 + y1 i)^2 = a / d; delta = 0 only when a1 = 0 and n = -a0: take n = a0 then
//   e = (delta d^3)^((p-3)/4)             second Fp exponentiation: r = delta d e has r^2 = +-delta / d, and 1 / (d r) = delta d^4 e^3 -- no inversion
//   r^2 d == delta ? y = (r, a1 / (2 d r)) : y = (a1 / (2 d r), r)       (in the second case -delta / d = (n' - a0 / d) / 2 for the other root n' = -n / d of the norm)
// Two 379-bit Fp exponentiations with the dedicated squaring (1.96 against 2.53 ms of kernel time per 65,536 roots) and none of the candidate tests of swu_finish.
struct SwuNormState { SFp2 t, zt2, num, den, a; SFp d, na; };
static inline SwuNormState swu_norm_prepare(const SFp2& t) {
  SwuNormState s; s.t = t;
  SFp2 Z = fp2_const(NBLS_SWU_Z), A = fp2_const(NBLS_SWU_A), Bc = fp2_const(NBLS_SWU_B);
  SFp2 t2 = mat(sqr(t));
  s.zt2 = mat(mul(Z, t2));
  SFp2 ztzt = mat(s.zt2 + sqr(s.zt2));
  SFp2 den0 = mat(-mul(A, ztzt));
  s.num = mat(mul(Bc, ztzt + fp2_one()));
  SFp dz = eq_zero(den0);
  s.den = select2(dz, mat(mul(Z, A)), den0);                                  // exceptional case (math.ts:1233)
  SFp2 den2 = mat(sqr(s.den));
  SFp2 v = mat(mul(den2, s.den));
  SFp2 num2 = mat(sqr(s.num));
  SFp2 u = mat(mul(num2, s.num) + mul(mat(mul(A, s.num)), den2) + mul(Bc, v));
  s.a = mat(mul(u, conj(v)));
  s.d = mat(sqr(v.c0) + sqr(v.c1));
  s.na = mat(sqr(s.a.c0) + sqr(s.a.c1));
  return s;
}
// between the exponentiations: n = N(a)^((p+1)/4) -> the numerator of the chosen x, a1 / 2 and delta of the value whose root is taken, and the second exponentiation's input
struct SwuNormMid { SFp2 num; SFp a1h, delta, g; };
static inline SwuNormMid swu_norm_mid(const SFp2& t, const SFp2& zt2, const SFp2& num, const SFp2& a, const SFp& d, const SFp& n) {
  SFp na = mat(sqr(a.c0) + sqr(a.c1));
  SFp success = is_zero(sqr(n) - na);
  SFp2 zt2_3 = mat(mul(mat(sqr(zt2)), zt2));
  SFp2 a2 = mat(mul(a, zt2_3));                                               // u(x1) conj(v) = Z^3 t^6 u(x0) conj(v)   (math.ts:1246)
  SFp nt = mat(sqr(t.c0) + sqr(t.c1));
  SFp nt3 = mat(mul(mat(sqr(nt)), nt));
  SFp n2 = mat(mul(mat(mul(n, fp_const(NBLS_SWU_SQRT_M125))), nt3));
  SFp2 as = select2(success, a, a2); SFp ns = select(success, n, n2);
  SFp d0 = halve(as.c0 + ns);
  SwuNormMid m;
  m.delta = select(is_zero(d0), mat(as.c0), d0);
  m.a1h = halve(as.c1);
  m.g = mat(mul(m.delta, mat(mul(mat(sqr(d)), d))));
  m.num = select2(success, num, mat(mul(num, zt2)));                          // success2: numerator *= Z t^2   (math.ts:1261)
  return m;
}
// after the second exponentiation e = g^((p-3)/4): the point on E2' projectively, (X : Y : Z) = (num : y den : den), as swu_finish returns it
static inline Pt<SFp2> swu_norm_finish(const SFp2& t, const SFp2& num, const SFp2& den, const SFp& a1h, const SFp& delta, const SFp& d, const SFp& e) {
  SFp q = mat(mul(d, e)), r = mat(mul(delta, q)), q2 = mat(sqr(q));
  SFp sd = mat(mul(mat(mul(r, q2)), d));                                      // 1 / (d r)
  SFp other = mat(mul(a1h, sd));
  SFp pos = is_zero(mul(mat(sqr(r)), d) - delta);
  SFp2 y = {select(pos, r, other), select(pos, other, r)};
  SFp flip = f_xor(sgn0(t), sgn0(y));                                         // math.ts:1264
  y = select2(flip, mat(-y), y);
  return pt_mat<SFp2>({num, mul(y, den), den});
}
// isogenyMapG2 (math.ts:1315-1325) on a projective point (X : Y : Z): homogenised Horner evaluation, no inversion
static inline Pt<SFp2> isogeny_g2_proj(const Pt<SFp2>& p) {
  SFp2 Z2 = mat(sqr(p.z)), Z3 = mat(mul(Z2, p.z));
  auto horner = [&](const u32 c[4][2][NLIMBS]) {
    // sum_i c[i] X^(3-i) Z^i  (coefficient lists are highest degree first)
    SFp2 X2 = mat(sqr(p.x)), X3 = mat(mul(X2, p.x));
    return mat(mul(fp2_const(c[0]), X3) + mul(fp2_const(c[1]), mat(mul(X2, p.z))) + mul(fp2_const(c[2]), mat(mul(p.x, Z2))) + mul(fp2_const(c[3]), Z3));
  };
  SFp2 xn = horner(NBLS_ISO_XNUM), xd = horner(NBLS_ISO_XDEN), yn = horner(NBLS_ISO_YNUM), yd = horner(NBLS_ISO_YDEN);
  // x' = xn/xd, y' = (Y/Z) yn/yd  (all four polynomials carry the same factor Z^3)
  SFp2 zxd = mat(mul(p.z, xd));
  return pt_mat<SFp2>({mul(mat(mul(xn, yd)), p.z), mul(mat(mul(p.y, yn)), xd), mul(zxd, yd)});
}
// ---------------------------------------------------------------- G1 hash-to-curve (index.ts:331-350)
// map_to_curve_simple_swu_3mod4 (math.ts:1272-1313) split around its exponentiation tv4^((p-3)/4)
struct Swu1State { SFp u, tv1, xNum1, xNum2, xDen, gxd, gx1, tv2, tv4; };
static inline Swu1State swu1_prepare(const SFp& u) {
  Swu1State s; s.u = u;
  SFp A = fp_const(NBLS_G1_SWU_A), Bc = fp_const(NBLS_G1_SWU_B), Z = fp_const(NBLS_G1_SWU_Z);
  s.tv1 = mat(sqr(u));
  SFp tv3 = mat(mul(Z, s.tv1));
  SFp xd0 = mat(sqr(tv3) + tv3);
  s.xNum1 = mat(mul(xd0 + fp_one(), Bc));
  s.xNum2 = mat(mul(tv3, s.xNum1));
  SFp xd1 = mat(-mul(A, xd0));
  s.xDen = select(is_zero(xd1), mat(mul(A, Z)), xd1);                          // exceptional case (math.ts:1288)
  SFp xd2 = mat(sqr(s.xDen));
  s.gxd = mat(mul(xd2, s.xDen));
  s.gx1 = mat(mul(mat(sqr(s.xNum1) + mul(A, xd2)), s.xNum1) + mul(Bc, s.gxd));  // x1n^3 + A x1n xd^2 + B xd^3
  s.tv2 = mat(mul(s.gx1, s.gxd));
  s.tv4 = mat(mul(mat(sqr(s.gxd)), s.tv2));
  return s;
}
// given pw = tv4^((p-3)/4): the point on E1' projectively, (X : Y : Z) = (xNum : y xDen : xDen)
static inline Pt<SFp> swu1_finish(const Swu1State& s, const SFp& pw) {
  SFp y1 = mat(mul(pw, s.tv2));
  SFp y2 = mat(mul(mat(mul(mat(mul(y1, fp_const(NBLS_G1_SWU_C2))), s.tv1)), s.u));
  SFp ok = is_zero(mul(mat(sqr(y1)), s.gxd) - s.gx1);                           // y1^2 gxd == gx1
  SFp xNum = select(ok, s.xNum1, s.xNum2), yPos = select(ok, y1, y2);
  SFp flip = f_xor(is_odd(std_canon(s.u)), is_odd(std_canon(yPos)));            // sgn0_m_eq_1 (math.ts:1187-1189)
  SFp y = select(flip, mat(-yPos), yPos);
  return pt_mat<SFp>({xNum, mul(y, s.xDen), s.xDen});
}
// isogenyMapG1 (11-isogeny, math.ts:1315-1327) on a projective point: homogenised Horner, h_j = h_(j-1) X + c_j Z^j
static inline Pt<SFp> isogeny_g1_proj(const Pt<SFp>& p) {
  std::vector<SFp> zp(16); zp[1] = p.z;
  for (int j = 2; j <= 15; j++) zp[j] = mat(mul(zp[j - 1], p.z));
  auto horner = [&](const u32 (*c)[NLIMBS], int d) {
    SFp h = fp_const(c[0]);
    for (int j = 1; j <= d; j++) h = mat(mul(h, p.x) + mul(fp_const(c[j]), zp[j]));
    return h;
  };
  SFp xn = horner(NBLS_G1_ISO_XNUM, 11), xd = horner(NBLS_G1_ISO_XDEN, 10), yn = horner(NBLS_G1_ISO_YNUM, 15), yd = horner(NBLS_G1_ISO_YDEN, 15);
  // x' = XN / (XD Z), y' = (Y / Z) YN / YD   (XN carries Z^11, XD Z^10, YN and YD Z^15)
  return pt_mat<SFp>({mul(xn, yd), mul(mat(mul(p.y, yn)), xd), mul(mat(mul(xd, p.z)), yd)});
}
// PointG1.clearCofactor (index.ts:401-405): [|x|]P + P
static inline Pt<SFp> clear_cofactor_g1(const Pt<SFp>& P) { return pt_add(pt_mul_u64(P, NBLS_X), P); }

static inline Pt<SFp2> psi_proj(const Pt<SFp2>& p) { return pt_mat<SFp2>({mul(conj(p.x), fp2_const(NBLS_PSI_X)), mul(conj(p.y), fp2_const(NBLS_PSI_Y)), conj(p.z)}); }
static inline Pt<SFp2> psi2_proj(const Pt<SFp2>& p) { return pt_mat<SFp2>({mul_fp(p.x, fp_const(NBLS_PSI2_C1)), -p.y, p.z}); }
// [k]Q for Q IN THE SUBGROUP G2 and a per-item scalar, with the scalar split along the endomorphism psi (round 5; sign, index.ts:744-752: Q = H(m) lies in G2 by construction):
// psi acts on G2 as [z] = [-|z|] (the eigenvalue the reference's own subgroup check uses, index.ts:640-657), so with k = a0 + a1 |z| + a2 |z|^2 + a3 |z|^3 (four digits of at most
// 65 bits, msm_kernels.hip msm_decompose_kernel)   [k]Q = [a0]Q - psi([a1]Q) + psi^2([a2]Q) - psi^3([a3]Q).
// One accumulator, 2-bit windows over the FOUR digits at once: per window two doublings and four additions of psi^i(+-T[d_i]) with T = {0, Q, 2Q, 3Q} -- 33 windows, 66 doublings and
// 132 additions where the plain ladder spends 256 and 128.  Constant time as the ladder: every table entry is read, selects are masked, psi is applied to whatever was selected.
static inline Pt<SFp2> pt_mul_gls_g2(const Pt<SFp2>& q, const SFp a_raw[4]) {
  Pt<SFp2> T[4];
  T[0] = pt_mat(pt_identity<SFp2>()); T[1] = pt_mat(q); T[2] = pt_dbl(q); T[3] = pt_add(T[2], T[1]);
  Pt<SFp2> r = T[0];
  for (int w = 32; w >= 0; w--) {
    if (w != 32) r = pt_dbl_n(r, 2);
    for (int i = 0; i < 4; i++) {
      const SFp b0 = bit_flag(a_raw[i], 2 * w), b1 = bit_flag(a_raw[i], 2 * w + 1);
      Pt<SFp2> v = pt_sel<SFp2>(b1, pt_sel<SFp2>(b0, T[3], T[2]), pt_sel<SFp2>(b0, T[1], T[0]));
      if (i == 1) v = pt_neg(psi_proj(v));
      else if (i == 2) v = psi2_proj(v);
      else if (i == 3) v = pt_neg(psi_proj(psi2_proj(v)));
      r = pt_add(r, v);
    }
  }
  return r;
}
// The same multiplication with ONE addition per bit (round 5, launches of at most one wavefront per SIMD, where the length of the instruction stream is the time): the four digits are
// recoded SIGN-ALIGNED (Faz-Hernandez, Longa, Sanchez 2013, "GLV-SAC"; msm_kernels.hip sac_recode_kernel): with a0 odd, a0 = sum_i s_i 2^i over 66 digits s_i = +-1 (s_65 = +1), and
// a_j = sum_i s_i e_ji 2^i with e_ji in {0, 1} for j = 1 .. 3, so   [k]Q = sum_i 2^i s_i (Q0 + e_1i Q1 + e_2i Q2 + e_3i Q3),  Q0 = Q, Q1 = -psi(Q), Q2 = psi^2(Q), Q3 = -psi^3(Q):
// a table of the eight sums Q0 + ..., per bit one doubling, one entry picked by masked selects on (e_3i, e_2i, e_1i), its y negated under the mask of s_i, one complete addition --
// 65 doublings + 66 additions + 7 for the table where the windowed form spends 66 + 132.  An even a0 is recoded as a0 + 1 and Q0 is subtracted at the end (both results are computed,
// one is selected).  rc[0]: bits 0 .. 65 = (s_i == +1), bit 66 = "a0 was even"; rc[j]: bit i = e_ji.  The table costs 48 slots (101 with the working set: three workgroups per CU): the form for launches whose wavefronts are all resident at once (6144 keys), the wrong one for dense launches.
static inline Pt<SFp2> pt_mul_sac_g2(const Pt<SFp2>& q, const SFp rc[4]) {
  const Pt<SFp2> Q0 = pt_mat(q), Q1 = pt_mat(pt_neg(psi_proj(Q0))), Q2 = pt_mat(psi2_proj(Q0)), Q3 = pt_mat(pt_neg(psi_proj(Q2)));
  Pt<SFp2> T[8];
  T[0] = Q0; T[1] = pt_add(Q0, Q1); T[2] = pt_add(Q0, Q2); T[3] = pt_add(T[1], Q2);
  for (int j = 0; j < 4; j++) T[4 + j] = pt_add(T[j], Q3);
  auto pick = [&](int i) {
    const SFp e1 = bit_flag(rc[1], i), e2 = bit_flag(rc[2], i), e3 = bit_flag(rc[3], i);
    Pt<SFp2> lo = pt_sel<SFp2>(e2, pt_sel<SFp2>(e1, T[3], T[2]), pt_sel<SFp2>(e1, T[1], T[0]));
    Pt<SFp2> hi = pt_sel<SFp2>(e2, pt_sel<SFp2>(e1, T[7], T[6]), pt_sel<SFp2>(e1, T[5], T[4]));
    return pt_sel<SFp2>(e3, hi, lo);
  };
  Pt<SFp2> r = pick(65);
  for (int i = 64; i >= 0; i--) {
    Pt<SFp2> v = pick(i);
    v.y = sel<SFp2>(bit_flag(rc[0], i), v.y, -v.y);
    r = pt_add(pt_dbl(r), v);
  }
  return pt_sel<SFp2>(bit_flag(rc[0], 66), pt_add(r, pt_neg(Q0)), r);
}
// PointG2.clearCofactor (index.ts:659-672)
// The same in two halves around the second multiplication by x, chained through HBM so that neither program keeps more than the ladder's base,
// its running point and their temporaries live (26 slots instead of 38: twelve wavefronts per CU instead of six -- the one-program form ran at
// 1.5 wavefronts per SIMD and half the issue rate).  clear_cofactor_g2(P) = S + (-[x]base) with
//   t1 = -[x]P, base = t1 + psi(P), S = psi^2(2P) - psi(P) - t1 - P.
// Round 4: the two points that do not depend on t1 -- v = psi(P) and u = psi^2(2P) - psi(P) - P -- are computed by a program of their own (P_H2C_C0); the first ladder
// program reads v back AFTER its ladder (base = t1 + v) and hands t1 on, the second reads t1 and u after ITS ladder: result = u + (-[x]base - t1).  Every final addition then
// sees two points and the temporaries of one addition (26 slots, eleven workgroups per CU) where the round-3 form held t1, t2, t3 and P at the end of the first ladder
// (39 slots, seven workgroups per CU).  Same group elements, other projective representatives.
static inline void clear_cofactor_g2_pre(const Pt<SFp2>& P, Pt<SFp2>& v, Pt<SFp2>& u) {
  Pt<SFp2> w = pt_add(psi2_proj(pt_dbl(P)), pt_neg(P));
  v = psi_proj(P);
  u = pt_add(w, pt_neg(v));
}
static inline void clear_cofactor_g2_first(const Pt<SFp2>& P, const Pt<SFp2>& v, Pt<SFp2>& base, Pt<SFp2>& t1) {
  t1 = pt_mat(pt_neg(pt_mul_u64(P, NBLS_X)));            // [-x]P
  base = pt_add(t1, v);
}
static inline Pt<SFp2> clear_cofactor_g2_second(const Pt<SFp2>& base, const Pt<SFp2>& t1, const Pt<SFp2>& u) {
  return pt_add(u, pt_add(pt_neg(pt_mul_u64(base, NBLS_X)), pt_neg(t1)));
}
static inline Pt<SFp2> clear_cofactor_g2(const Pt<SFp2>& P) {
  Pt<SFp2> t1 = pt_neg(pt_mul_u64(P, NBLS_X));           // [-x]P
  Pt<SFp2> t2 = psi_proj(P);
  Pt<SFp2> t3 = psi2_proj(pt_dbl(P));
  t3 = pt_add(t3, pt_neg(t2));
  t2 = pt_add(t1, t2);
  t2 = pt_neg(pt_mul_u64(t2, NBLS_X));
  t3 = pt_add(t3, t2);
  t3 = pt_add(t3, pt_neg(t1));
  return pt_add(t3, pt_neg(P));
}

}  // namespace nbls
