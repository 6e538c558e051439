// programs.cpp -- the step programs of the pairing path, traced from the tower formulas (tower.h).
//
// Buffer conventions (KernelArgs.bufs index):
//   0  G1 affine points, 96 B/item  (x||y, big-endian)                    reference wire format
//   1  G2 affine points, 192 B/item (x.c0||x.c1||y.c0||y.c1)
//   2  Fp12 wire bytes, 576 B/item (Fp12.toBytes order, math.ts:875-884)  outputs / final_exp input
//   3  F  : raw Montgomery Fp12 scratch, 576 B/item (12 x 12 words)
//   4  N  : raw Fp scratch, 48 B/item (norm to invert / its inverse)
//   5  F' : second raw Fp12 scratch (product reduction output)
#include "programs.h"
#include "config.h"
#include <mutex>
#include "aot.h"
#include "tower.h"
#include "curve.h"
#include "codec.h"
#include "vm_exec.h"

namespace nbls {

// ---------------------------------------------------------------- Miller loop (math.ts:1331-1388, fused)
// calcPairingPrecomputes and millerLoop are fused: line coefficients are produced and consumed on the fly.  The
// R-point update and the line coefficients are the reference's polynomials in (Rx, Ry, Rz, Qx, Qy); `.div(2n)` is
// realised by the halve LIN post-op (x/2 mod p is the same field element as x * 2^-1).
static SFp12 trace_miller(const SFp& Px, const SFp& Py, const SFp2& Qx, const SFp2& Qy) {
  SFp2 Rx = Qx, Ry = Qy, Rz = fp2_one();
  SFp12 f = fp12_one();
  for (int i = 62; i >= 0; i--) {
    // doubling step, math.ts:1339-1351
    SFp2 t0 = sqr(Ry), t1 = sqr(Rz);
    SFp2 t2 = mat(mul_by_b(scale(t1, 3)));
    SFp2 t3 = scale(t2, 3);
    SFp2 t4 = mat(sqr(Ry + Rz) - t1 - t0);
    SFp2 e0 = t2 - t0, e1 = scale(sqr(Rx), 3), e2 = -t4;
    SFp2 nRx = mul(halve(t0 - t3), mul(Rx, Ry));       // ((T0 - T3) * Rx * Ry) / 2
    SFp2 nRy = sqr(halve(t0 + t3)) - scale(sqr(t2), 3);  // ((T0 + T3)/2)^2 - 3 T2^2
    SFp2 nRz = mul(t0, t4);
    Rx = mat(nRx); Ry = mat(nRy); Rz = mat(nRz);
    f = mat(mul_by_014(f, e0, mul_fp(e1, Px), mul_fp(e2, Py)));   // math.ts:1379
    if ((NBLS_X >> i) & 1) {
      // addition step, math.ts:1353-1367
      SFp2 a0 = mat(Ry - mul(Qy, Rz)), a1 = mat(Rx - mul(Qx, Rz));
      SFp2 g0 = mul(a0, Qx) - mul(a1, Qy), g1 = -a0, g2 = a1;
      SFp2 a2 = mat(sqr(a1)), a3 = mat(mul(a2, a1)), a4 = mat(mul(a2, Rx));
      SFp2 a5 = mat(a3 - scale(a4, 2) + mul(sqr(a0), Rz));
      nRx = mul(a1, a5);
      nRy = mul(a4 - a5, a0) - mul(a3, Ry);
      nRz = mul(Rz, a3);
      Rx = mat(nRx); Ry = mat(nRy); Rz = mat(nRz);
      f = mat(mul_by_014(f, g0, mul_fp(g1, Px), mul_fp(g2, Py)));   // math.ts:1383
    }
    if (i != 0) f = mat(sqr(f));
  }
  return conj(f);
}

// Product of m Miller loops with ONE shared accumulator: f <- (f * prod_j line_j)^2 per bit instead of m separate
// accumulators, which saves m-1 of the m Fp12 squarings per bit.  prod_j millerLoop(P_j, Q_j) is the same field element
// either way ((prod a_j)^2 = prod a_j^2), so Miller products (verify / verifyBatch, index.ts:756-821) stay bit-exact.
static SFp12 trace_miller_shared(const std::vector<SFp>& Px, const std::vector<SFp>& Py, const std::vector<SFp2>& Qx, const std::vector<SFp2>& Qy) {
  const size_t m = Px.size();
  std::vector<SFp2> Rx = Qx, Ry = Qy, Rz(m, fp2_one());
  SFp12 f = fp12_one();
  for (int i = 62; i >= 0; i--) {
    for (size_t j = 0; j < m; j++) {
      // doubling step, math.ts:1339-1351
      SFp2 t0 = sqr(Ry[j]), t1 = sqr(Rz[j]);
      SFp2 t2 = mat(mul_by_b(scale(t1, 3)));
      SFp2 t3 = scale(t2, 3);
      SFp2 t4 = mat(sqr(Ry[j] + Rz[j]) - t1 - t0);
      SFp2 e0 = t2 - t0, e1 = scale(sqr(Rx[j]), 3), e2 = -t4;
      SFp2 nRx = mul(halve(t0 - t3), mul(Rx[j], Ry[j]));
      SFp2 nRy = sqr(halve(t0 + t3)) - scale(sqr(t2), 3);
      SFp2 nRz = mul(t0, t4);
      Rx[j] = mat(nRx); Ry[j] = mat(nRy); Rz[j] = mat(nRz);
      f = mat(mul_by_014(f, e0, mul_fp(e1, Px[j]), mul_fp(e2, Py[j])));
    }
    if ((NBLS_X >> i) & 1) {
      for (size_t j = 0; j < m; j++) {
        // addition step, math.ts:1353-1367
        SFp2 a0 = mat(Ry[j] - mul(Qy[j], Rz[j])), a1 = mat(Rx[j] - mul(Qx[j], Rz[j]));
        SFp2 g0 = mul(a0, Qx[j]) - mul(a1, Qy[j]), g1 = -a0, g2 = a1;
        SFp2 a2 = mat(sqr(a1)), a3 = mat(mul(a2, a1)), a4 = mat(mul(a2, Rx[j]));
        SFp2 a5 = mat(a3 - scale(a4, 2) + mul(sqr(a0), Rz[j]));
        SFp2 nRx = mul(a1, a5), nRy = mul(a4 - a5, a0) - mul(a3, Ry[j]), nRz = mul(Rz[j], a3);
        Rx[j] = mat(nRx); Ry[j] = mat(nRy); Rz[j] = mat(nRz);
        f = mat(mul_by_014(f, g0, mul_fp(g1, Px[j]), mul_fp(g2, Py[j])));
      }
    }
    if (i != 0) f = mat(sqr(f));
  }
  return conj(f);
}


// ---------------------------------------------------------------- Miller loop in two programs (math.ts:1331-1388)
// LINES: the point chain R <- 2R (+ Q) of calcPairingPrecomputes with the reference's polynomials (so that every line coefficient is the
// reference's field element and pairing(P, Q, false) stays bit-exact), regrouped for the lane-op model: the state is (Rx, Ry, D = 2 Rz), on
// which a doubling is two levels of ten lane-ops around one level of sums --
//   level 1: t0 = Ry^2, t2 = 3 xi D^2 (= 3 b Rz^2, multiplier 3 and one doubled operand), t4 = Ry D (= (Ry + Rz)^2 - Rz^2 - Ry^2), c1 = 3 Rx^2, Rx Ry
//   sums:    A = (t0 - 3 t2) / 2, B = (t0 + 3 t2) / 2, t3 = 3 t2, c0 = t2 - t0, c2 = -t4
//   level 2: Rx' = A (Rx Ry), Ry' = B^2 - t2 t3, D' = 2 t0 t4 (= 2 Rz'), and with P known c1 Px, c2 Py
// -- 20 lane-ops per bit where the fused trace of round 1 spent 37.  Entry j of a table: c0.c0 c0.c1 c1.c0 c1.c1 c2.c0 c2.c1 (raw elements).
struct LineSink { int buf; int base; const SFp* Px; const SFp* Py; };
static void emit_line(const LineSink& o, int j, const SFp2& c0, const SFp2& c1, const SFp2& c2) {
  const int off = o.base + 48 * 6 * j;
  SFp2 e1 = c1, e2 = c2;
  if (o.Px) { e1 = mul_fp(c1, *o.Px); e2 = mul_fp(c2, *o.Py); }   // E[1].multiply(Px), E[2].multiply(Py), math.ts:1379
  outputw(c0.c0, o.buf, off); outputw(c0.c1, o.buf, off + 48);
  outputw(e1.c0, o.buf, off + 96); outputw(e1.c1, o.buf, off + 144);
  outputw(e2.c0, o.buf, off + 192); outputw(e2.c1, o.buf, off + 240);
}
static void trace_lines(const SFp2& Qx, const SFp2& Qy, const LineSink& out) {
  SFp2 Rx = Qx, Ry = Qy, D = mat(scale(fp2_one(), 2));
  int j = 0;
  for (int i = 62; i >= 0; i--) {
    // doubling step, math.ts:1339-1351
    SFp2 t0 = mat(sqr(Ry)), t2 = mat(scale(mulnr(sqr(D)), 3)), t4 = mat(mul(Ry, D)), c1 = mat(scale(sqr(Rx), 3)), rxry = mat(mul(Rx, Ry));
    SFp2 A = halve(t0 - scale(t2, 3)), Bh = halve(t0 + scale(t2, 3));
    SFp2 nRy;
    static const bool two_squares = !env_set("NBLS_DBL_PLAIN");
    if (out.Px && two_squares) {
      // Ry' = B^2 - 3 t2^2 is a difference of two Fp2 squares: two limb products per coefficient (B^2: (b0 + b1)(b0 - b1), 2 b0 b1; 3 t2^2: (t0' + t1')(3 t0' - 3 t1'),
      // 2 t0' (3 t1') with the factor 3 carried by sums) where B^2 - t2 t3 costs three, so the second level is one full product round.  The three sums take the lanes that
      // -t4 occupied: with P known, c2 Py = -(t4 Py) negates inside the product (LINES_Q stores c2 itself and keeps the other form).
      SFp Ps = SFp(materialize(t2.c0 + t2.c1)), M = SFp(materialize(scale(t2.c0 - t2.c1, 3))), K = SFp(materialize(scale(t2.c1, 3)));
      emit_line(out, j++, mat(t2 - t0), c1, -t4);
      nRy = mat(SFp2{mul(Bh.c0 + Bh.c1, Bh.c0 - Bh.c1) - mul(Ps, M), scale(mul(Bh.c0, Bh.c1), 2) - scale(mul(t2.c0, K), 2)});
    } else {
      SFp2 t3 = mat(scale(t2, 3));
      emit_line(out, j++, mat(t2 - t0), c1, mat(-t4));
      nRy = mat(sqr(Bh) - mul(t2, t3));
    }
    SFp2 nRx = mat(mul(A, rxry)), nD = mat(scale(mul(t0, t4), 2));
    Rx = nRx; Ry = nRy; D = nD;
    if ((NBLS_X >> i) & 1) {
      // addition step, math.ts:1353-1367, on Rz = D / 2
      SFp2 Rz = halve(D);
      SFp2 a0 = mat(Ry - mul(Qy, Rz)), a1 = mat(Rx - mul(Qx, Rz));
      emit_line(out, j++, mat(mul(a0, Qx) - mul(a1, Qy)), mat(-a0), a1);
      SFp2 a2 = mat(sqr(a1)), a3 = mat(mul(a2, a1)), a4 = mat(mul(a2, Rx));
      SFp2 a5 = mat(a3 - scale(a4, 2) + mul(sqr(a0), Rz));
      nRx = mat(mul(a1, a5)); nRy = mat(mul(a4 - a5, a0) - mul(a3, Ry));
      SFp2 nRz = mat(mul(Rz, a3));
      Rx = nRx; Ry = nRy; D = mat(scale(nRz, 2));
    }
  }
}
static SFp2 load_line_coef(int buf, int off) { return {inputw(buf, off), inputw(buf, off + 48)}; }
// ACC: millerLoop (math.ts:1373-1388) over m line tables per item with ONE accumulator: f <- (f * prod_t line_t)^2 per bit.  m = 1 is the
// reference's loop; for m > 1 the product of the m Miller values is the same field element ((prod a_t)^2 = prod a_t^2), so Miller products
// (verify / verifyBatch, index.ts:756-821) stay bit-exact while m - 1 of the m Fp12 squarings per bit are saved.  With Px / Py given the
// tables hold prepared lines (calcPairingPrecomputes output) and the G1 coordinates are folded in here.
static SFp12 trace_acc(int m, int buf, const std::vector<SFp>* Px = nullptr, const std::vector<SFp>* Py = nullptr) {
  SFp12 f = fp12_one();
  int j = 0;
  auto step = [&](int jj) {
    for (int t = 0; t < m; t++) {
      const int off = 48 * LINE_ELEMS * t + 48 * 6 * jj;
      SFp2 c0 = load_line_coef(buf, off), c1 = load_line_coef(buf, off + 96), c2 = load_line_coef(buf, off + 192);
      if (Px) { c1 = mat(mul_fp(c1, (*Px)[t])); c2 = mat(mul_fp(c2, (*Py)[t])); }
      f = mat(mul_by_014(f, c0, c1, c2));
    }
  };
  for (int i = 62; i >= 0; i--) {
    step(j++);
    if ((NBLS_X >> i) & 1) step(j++);
    if (i != 0) f = mat(sqr(f));
  }
  return conj(f);
}

// ---------------------------------------------------------------- Fp12 inversion split around the one Fp inversion
// Fp12.invert (math.ts:793-797) -> Fp6.invert (672-680) -> Fp2.invert (522-526) -> Fp.invert.  Everything except
// the Fp inversion is recomputed on both sides of the inversion kernel (cheap: ~90 Fp products).
struct InvChain { SFp6 t; SFp2 T0, T1, T2, d; SFp n; };
static InvChain inv_chain(const SFp12& f) {
  InvChain c;
  c.t = mat(sqr(f.c0) - mulnr(sqr(f.c1)));                      // c0^2 - c1^2 * v
  c.T0 = mat(sqr(c.t.c0) - mulnr(mul(c.t.c2, c.t.c1)));
  c.T1 = mat(mulnr(sqr(c.t.c2)) - mul(c.t.c0, c.t.c1));
  c.T2 = mat(sqr(c.t.c1) - mul(c.t.c0, c.t.c2));
  c.d = mat(mulnr(mul(c.t.c2, c.T1) + mul(c.t.c1, c.T2)) + mul(c.t.c0, c.T0));
  c.n = sqr(c.d.c0) + sqr(c.d.c1);
  return c;
}
static SFp12 inv_finish(const SFp12& f, const InvChain& c, const SFp& ninv) {
  SFp2 dinv = {mul(c.d.c0, ninv), -mul(c.d.c1, ninv)};
  SFp6 tinv = mat(SFp6{mul(dinv, c.T0), mul(dinv, c.T1), mul(dinv, c.T2)});
  return {mul(f.c0, tinv), -mul(f.c1, tinv)};
}

// ---------------------------------------------------------------- final exponentiation (math.ts:856-874)
// Split into phase programs chained through raw Fp12 scratch buffers in HBM (576 B per value per item): the long-lived
// intermediates t1..t7 would otherwise pin ~300 LDS slots per instance and starve occupancy.  One EXPX program is reused
// for all five cyclotomic exponentiations.
//   t0 = f^(p^6) / f ; t1 = t0^(p^2) * t0                      FE_EASY
//   t2 = conj(t1^x)                                             EXPX
//   t3 = conj(cycsqr(t1)) * t2                                  FE_MID1
//   t4 = conj(t3^x) ; t5 = conj(t4^x) ; t6' = conj(t5^x)        EXPX x3
//   t6 = t6' * cycsqr(t2)                                       FE_MID2
//   t7 = conj(t6^x)                                             EXPX
//   (t2 t5)^(p^2) * (t4 t1)^(p^3) * (t6 conj(t1))^p * t7 conj(t3) t1     FE_FINAL
static SFp12 trace_fe_easy(const SFp12& f, const SFp12& finv) {
  SFp12 t0 = mat(mul(frob(f, 6), finv));
  return mul(frob(t0, 2), t0);
}
static SFp12 trace_fe_final(const SFp12& t1, const SFp12& t2, const SFp12& t3, const SFp12& t4, const SFp12& t5, const SFp12& t6, const SFp12& t7) {
  SFp12 a = mat(frob(mul(t2, t5), 2));
  SFp12 b = mat(frob(mul(t4, t1), 3));
  SFp12 c = mat(frob(mul(t6, conj(t1)), 1));
  SFp12 d = mat(mul(mat(mul(t7, conj(t3))), t1));
  return mul(mat(mul(mat(mul(a, b)), c)), d);
}

static void load_points(SFp& Px, SFp& Py, SFp2& Qx, SFp2& Qy) {
  Px = input(0, 0); Py = input(0, 48);
  Qx = input_fp2(1, 0); Qy = input_fp2(1, 96);
}

// lanes per work item (instances per wave = 64 / W): an Fp12 lane-op step has 12 heavy lanes
static int env_int(const char* name, int dflt) { return (int)env_long(name, dflt); }
static const int MILLER_W = env_int("NBLS_MILLER_W", 16);
// lanes per item of the point programs.  A G1 operation has at most ~4 independent field products per dependency level, so 4 lanes
// per item take the same number of steps as 8 with twice the items per wavefront (G1 decode + subgroup check 5.6 -> 3.9 ms at
// 65,536 keys); the G2 ladder at 8 lanes instead of 16: +9 % steps, twice the items (5.9 -> 4.5 ms at 8192 signatures).  The other
// G2 programs stay at 8: at 4 lanes their LDS footprint (50 KB per wavefront) leaves less than one wavefront per SIMD.
static const int G1_W = env_int("NBLS_G1_W", 4), G2_W = env_int("NBLS_G2_W", 8), G2MUL_W = env_int("NBLS_G2MUL_W", 8), G1MUL_W = env_int("NBLS_G1MUL_W", 4);
static const int EXPX_W = env_int("NBLS_EXPX_W", 12);
// the two Miller programs: ten lane-ops per level of the point chain (6 items per wavefront); an Fp12 op has 12 lane-ops (5 items, no idle lane).
// ACC_WINDOW keeps the line loads about one and a half iterations ahead of their use (they would otherwise all be hoisted to the front and pin 408 slots)
static const int LINES_W = env_int("NBLS_LINES_W", 10), ACC_W = env_int("NBLS_ACC_W", 12), ACC_WINDOW = env_int("NBLS_ACC_WINDOW", 330);   // an Fp12 op has exactly 12 lane-ops: 5 items per wave, no idle lane (+5..9 % over 16 lanes once >= 2 waves share a SIMD)

// programs that run on an ahead-of-time kernel (aot.h): their descriptors hold absolute addresses, so ONE shared copy of the constants costs nothing
// (the interpreter pays three instructions per operand for it) and frees the replicated copies' LDS: FE_MID1 13,440 -> 11,904 B, which keeps twelve
// workgroups per CU for the chained final exponentiation
static bool aot_listed(ProgId id) {
#define AOT_HAS(PART, NAME, P0, P1, P2, P3) if (id == P0 || id == P1 || id == P2 || id == P3) return true;
  NBLS_AOT_KERNELS(AOT_HAS)
#undef AOT_HAS
  return false;
}
static Program build(ProgId id) {
  Builder B;
  if (aot_listed(id) && env_int("NBLS_AOT_SHARED_CONSTS", 1)) B.shared_consts = 1;
  // the Fp12 squaring has 6 coefficients of 7 products next to 6 of 6: capping lane-ops at 6 products moves the seventh into a light
  // step that has free lanes (-4 % instructions); with two point chains per item (MILLER_RAW2) those steps are full, so no cap there
  if (id == P_MILLER_BYTES || id == P_MILLER_RAW || id == P_MILLER_FE) B.max_dot = env_int("NBLS_MILLER_MAXDOT", 6);
  if (id == P_EXPX) B.max_dot = env_int("NBLS_EXPX_MAXDOT", 8);
  const bool ls2 = id == P_MILLER_BYTES_LS2 || id == P_MILLER_RAW_LS2 || id == P_MILLER_FE_LS2 || id == P_EXPX_LS2 || id == P_H2C_C1_LS2 || id == P_H2C_C2_LS2 || id == P_G2_MUL_SAC_LS2;
  const bool ls = ls2 || id == P_MILLER_BYTES_LS || id == P_MILLER_RAW_LS || id == P_MILLER_FE_LS || id == P_EXPX_LS;
  if (ls) { B.lane_split = ls2 ? 2 : 4; B.max_dot = env_int("NBLS_LS_MAXDOT", 8); }   // four (two) sub-lanes share a lane-op's products: eight of them are two (four) rounds
  switch (id) {
    case P_MILLER_BYTES: case P_MILLER_BYTES_LS: case P_MILLER_BYTES_LS2: {
      SFp Px, Py; SFp2 Qx, Qy; load_points(Px, Py, Qx, Qy);
      output_fp12(trace_miller(Px, Py, Qx, Qy), 2, 0);
      return B.compile(ls2 ? "miller_bytes_ls2" : ls ? "miller_bytes_ls" : "miller_bytes", ls ? 16 : MILLER_W);
    }
    case P_MILLER_RAW: case P_MILLER_RAW_LS: case P_MILLER_RAW_LS2: {
      SFp Px, Py; SFp2 Qx, Qy; load_points(Px, Py, Qx, Qy);
      outputw_fp12(trace_miller(Px, Py, Qx, Qy), 3, 0);
      return B.compile(ls2 ? "miller_raw_ls2" : ls ? "miller_raw_ls" : "miller_raw", ls ? 16 : MILLER_W);
    }
    case P_MILLER_RAW2: {   // two pairs per item: g1 (buf 0, 2 x 96 B), g2 (buf 1, 2 x 192 B) -> raw Fp12 of the product (buf 3)
      std::vector<SFp> Px, Py; std::vector<SFp2> Qx, Qy;
      for (int j = 0; j < 2; j++) { Px.push_back(input(0, 96 * j)); Py.push_back(input(0, 96 * j + 48)); Qx.push_back(input_fp2(1, 192 * j)); Qy.push_back(input_fp2(1, 192 * j + 96)); }
      outputw_fp12(trace_miller_shared(Px, Py, Qx, Qy), 3, 0);
      B.sched_window = env_int("NBLS_MILLER2_WINDOW", 450);   // keeps the two point-update chains within ~1.5 iterations of the accumulator chain
      return B.compile("miller_raw2", MILLER_W);
    }
    case P_MILLER_FE: case P_MILLER_FE_LS: case P_MILLER_FE_LS2: {
      SFp Px, Py; SFp2 Qx, Qy; load_points(Px, Py, Qx, Qy);
      SFp12 f = mat(trace_miller(Px, Py, Qx, Qy));
      outputw_fp12(f, 3, 0);
      outputw(inv_chain(f).n, 4, 0);
      return B.compile(ls2 ? "miller_fe_ls2" : ls ? "miller_fe_ls" : "miller_fe", ls ? 16 : MILLER_W);
    }
    case P_NORM_RAW: {
      SFp12 f = inputw_fp12(3, 0);
      outputw(inv_chain(f).n, 4, 0);
      return B.compile("norm_raw", 16);
    }
    case P_NORM_BYTES: {
      SFp12 f = mat(input_fp12(2, 0));
      outputw_fp12(f, 3, 0);
      outputw(inv_chain(f).n, 4, 0);
      return B.compile("norm_bytes", 16);
    }
    case P_FE_EASY: {
      SFp12 f = inputw_fp12(3, 0);
      SFp ninv = inputw(4, 0);
      InvChain c = inv_chain(f);
      SFp12 finv = mat(inv_finish(f, c, ninv));
      outputw_fp12(trace_fe_easy(f, finv), 5, 0);
      return B.compile("fe_easy", env_int("NBLS_FE_EASY_W", 16));
    }
    case P_EXPX: case P_EXPX_LS: case P_EXPX_LS2: {
      if (env_int("NBLS_EXPX_RELOAD", 1)) {
        B.sched_window = env_int("NBLS_EXPX_WINDOW", 150);   // the reloads of the base are scheduled about one squaring ahead of the multiplication that needs them
        outputw_fp12(conj(cyclotomic_exp_x(inputw_fp12(3, 0), [&]() { return inputw_fp12(3, 0); })), 5, 0);
      } else outputw_fp12(conj(cyclotomic_exp_x(inputw_fp12(3, 0))), 5, 0);
      return B.compile(ls2 ? "expx_ls2" : ls ? "expx_ls" : "expx", ls ? 12 : EXPX_W);
    }
    case P_EXPC_SQ: {
      // raw Fp12 element order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 (two raw elements each)
      auto ld = [&](int i) { return SFp2{inputw(3, 96 * i), inputw(3, 96 * i + 48)}; };
      SCyc4 a = {ld(3), ld(2), ld(1), ld(5)};
      const SFp three = SFp(B.small_const(3));
      auto m3 = [&](const SFp2& v) { return SFp2{mul(v.c0, three), mul(v.c1, three)}; };
      SCyc4 z = mat(SCyc4{m3(a.g2), m3(a.g3), m3(a.g4), m3(a.g5)});
      int slot = 0;
      for (int k = 1; k <= EXPC_TOP; k++) {
        z = mat(compressed_sqr_tripled(z));
        if ((NBLS_X >> k) & 1) {
          const SFp2* c[4] = {&z.g2, &z.g3, &z.g4, &z.g5};
          for (int e = 0; e < 4; e++) { outputw(c[e]->c0, 5, (8 * slot + 2 * e) * 48); outputw(c[e]->c1, 5, (8 * slot + 2 * e + 1) * 48); }
          slot++;
        }
      }
      B.store_batch = 8;
      return B.compile("expc_sq", env_int("NBLS_EXPC_SQ_W", 8));
    }
    case P_EXPC_DEC_A: {
      // scratch layout (buf 5, EXPC_DEC_ELEMS raw elements): w_j = numerator * conj(g2) (2 x 3) | all-but-one products (3) | the same / 3 (3) | v_j (2 x 3) | zero flag
      auto ld = [&](int j, int e) { return SFp2{inputw(3, (8 * j + 2 * e) * 48), inputw(3, (8 * j + 2 * e + 1) * 48)}; };
      const SFp third = SFp(B.frac_const(1, 3));
      SFp n[EXPC_POWERS], z[EXPC_POWERS];
      for (int j = 0; j < EXPC_POWERS; j++) {
        SFp2 g2 = ld(j, 0), g3 = ld(j, 1), g4 = ld(j, 2), g5 = ld(j, 3);
        SFp d0 = scale(g2.c0, 2), d1 = scale(g2.c1, 2);
        SFp nj = SFp(materialize(mul(d0, d0) + mul(d1, d1)));                 // |2 g2|^2 = 4 |g2|^2: the 4 of the denominator rides in the doubled operands
        z[j] = f_and(is_zero(g2.c0), is_zero(g2.c1));
        n[j] = select(z[j], fp_one(), nj);                                    // a vanishing g2 must not wipe out the other inverses; the item is flagged and redone
        SFp2 num = mat(mulnr(sqr(g5)) + scale(sqr(g4), 3) - scale(g3, 6));    // on the tripled state: 3 g1 = (xi G5^2 + 3 G4^2 - 6 G3) / (4 G2)
        SFp2 w = mul(num, conj(g2));
        outputw(w.c0, 5, (2 * j) * 48); outputw(w.c1, 5, (2 * j + 1) * 48);
        // the part of the tripled g0 = xi (2 G1 G1/3 + G2/3 G5 - G3 G4) + 3 that does not need g1
        SFp2 u2 = mat(mul_fp(g2, third));
        SFp2 v = mul(mulnr(u2), g5) - mul(mulnr(g3), g4);
        outputw(v.c0, 5, (4 * EXPC_POWERS + 2 * j) * 48); outputw(v.c1, 5, (4 * EXPC_POWERS + 2 * j + 1) * 48);
      }
      // inverse of n_j = (n_0 n_1 n_2)^-1 * (product of the other two)
      static_assert(EXPC_POWERS == 3, "the product tree below is written for three denominators");
      SFp abo[3] = {SFp(materialize(mul(n[1], n[2]))), SFp(materialize(mul(n[0], n[2]))), SFp(materialize(mul(n[0], n[1])))};
      outputw(mul(abo[2], n[2]), 4, 0);
      for (int j = 0; j < EXPC_POWERS; j++) {
        outputw(abo[j], 5, (2 * EXPC_POWERS + j) * 48);
        outputw(mul(abo[j], third), 5, (3 * EXPC_POWERS + j) * 48);
      }
      outputw(f_or(f_or(z[0], z[1]), z[2]), 5, (6 * EXPC_POWERS) * 48);
      return B.compile("expc_dec_a", env_int("NBLS_EXPC_DECA_W", 12));
    }
    case P_EXPC_DEC_B: {
      auto ld = [&](int j, int e) { return SFp2{inputw(3, (8 * j + 2 * e) * 48), inputw(3, (8 * j + 2 * e + 1) * 48)}; };
      const SFp ninv = inputw(4, 0);
      SFp12 d[EXPC_POWERS];   // the tripled elements (3 A)^(2^16), (3 A)^(2^48), (3 A)^(2^57), decompressed
      for (int j = 0; j < EXPC_POWERS; j++) {
        SFp2 w = {inputw(6, (2 * j) * 48), inputw(6, (2 * j + 1) * 48)};
        SFp inv = SFp(materialize(mul(ninv, inputw(6, (2 * EXPC_POWERS + j) * 48))));    // 1 / |2 g2|^2
        SFp inv3 = SFp(materialize(mul(ninv, inputw(6, (3 * EXPC_POWERS + j) * 48))));   // a third of it
        SFp2 g1 = mat(mul_fp(w, inv)), u1 = mat(mul_fp(w, inv3));                       // tripled g1, and g1 itself
        SFp2 v = {inputw(6, (4 * EXPC_POWERS + 2 * j) * 48), inputw(6, (4 * EXPC_POWERS + 2 * j + 1) * 48)};
        SFp2 g0 = mul(mulnr(scale(g1, 2)), u1) + v;
        g0.c0 = g0.c0 + SFp(B.small_const(3));
        d[j] = {{mat(g0), ld(j, 2), ld(j, 1)}, {ld(j, 0), g1, ld(j, 3)}};
      }
      // A^|x| = A^(2^16) A^(2^48) A^(2^57) A^(2^60) A^(2^62) A^(2^63): the three top powers by plain squarings of the decompressed 2^57 power (3 + 2 + 1)
      SFp12 acc = mat(mul(d[0], d[1])), t = d[2];
      acc = mat(mul(acc, t));
      const int runs[3] = {3, 2, 1};
      for (int r = 0; r < 3; r++) {
        for (int k = 0; k < runs[r]; k++) t = mat(cyclotomic_sqr_tripled(t));
        acc = mat(mul(acc, t));
      }
      // six tripled factors: divide by 3^6
      u32 c[NLIMBS]; memcpy(c, NBLS_INV3, NLIMBS * 4);
      for (int k = 1; k < 6; k++) { u32 t2[NLIMBS]; mont_mul28(t2, c, NBLS_INV3); csub_p(t2); memcpy(c, t2, NLIMBS * 4); }
      outputw_fp12(conj(mul_fp(acc, constant(c))), 5, 0);
      status_out({{f_not(inputw(6, (6 * EXPC_POWERS) * 48)), 1}}, 7);
      B.sched_window = env_int("NBLS_EXPC_DEC_WINDOW", 150);
      return B.compile("expc_dec_b", env_int("NBLS_EXPC_DECB_W", 12));
    }
    case P_FE_MID1: {
      SFp12 a = inputw_fp12(3, 0), b = inputw_fp12(5, 0);
      outputw_fp12(mul(conj(mat(cyclotomic_sqr(a))), b), 6, 0);
      B.sched_window = env_int("NBLS_FE_MID_WINDOW", 0);
      return B.compile("fe_mid1", env_int("NBLS_FE_MID_W", 12));
    }
    case P_FE_MID2: {
      SFp12 a = inputw_fp12(3, 0), b = inputw_fp12(5, 0);
      outputw_fp12(mul(a, mat(cyclotomic_sqr(b))), 6, 0);
      B.sched_window = env_int("NBLS_FE_MID_WINDOW", 0);
      return B.compile("fe_mid2", env_int("NBLS_FE_MID_W", 12));
    }
    case P_FE_FINAL: {
      SFp12 t[7]; for (int i = 0; i < 7; i++) t[i] = inputw_fp12(i, 0);
      output_fp12(trace_fe_final(t[0], t[1], t[2], t[3], t[4], t[5], t[6]), 7, 0);
      return B.compile("fe_final", env_int("NBLS_FE_FINAL_W", 32));
    }
    case P_MUL2: {
      SFp12 a = inputw_fp12(3, 0), b = inputw_fp12(3, 576);
      outputw_fp12(mul(a, b), 5, 0);
      return B.compile("fp12_mul2", 16);
    }
    case P_MUL2S: {
      SFp12 a = inputw_fp12(3, 0), b = inputw_fp12(4, 0);
      outputw_fp12(mul(a, b), 5, 0);
      Program P = B.compile("fp12_mul2s", 16);
      P.aliases.push_back({5, 3});      // reduce_product (pipelines_pairing.cpp) runs the tree in place: the output buffer IS the first input
      return P;
    }
    case P_RAW_TO_BYTES: {
      output_fp12(inputw_fp12(3, 0), 2, 0);
      return B.compile("raw_to_bytes", 16);
    }
    case P_G1_VALIDATE: {
      SFp x = input(0, 0), y = input(0, 48), oc, sg;
      g1_validity_flags(x, y, oc, sg);
      status_out({{oc, 2}, {sg, 3}}, 7);
      return B.compile("g1_validate", G1_W);
    }
    case P_G2_VALIDATE: {
      SFp2 x = input_fp2(1, 0), y = input_fp2(1, 96); SFp oc, sg;
      g2_validity_flags(x, y, oc, sg);
      status_out({{oc, 2}, {sg, 3}}, 7);
      return B.compile("g2_validate", G2_W);
    }
    case P_G1_DEC_A: g1_decompress_A(0, 3, 4); return B.compile("g1_dec_a", 4);
    case P_G1_DEC_B: g1_decompress_B(0, 3, 4, 5, 6, 7); B.sched_window = env_int("NBLS_G1DEC_WINDOW", 0); return B.compile("g1_dec_b", G1_W);
    case P_G2_DEC_A: g2_decompress_A(0, 3, 4); return B.compile("g2_dec_a", 4);
    case P_G2_DEC_B: g2_decompress_B(0, 3, 4, 5, 6, 7); return B.compile("g2_dec_b", G2_W);
    case P_H2C_A: {
      for (int k = 0; k < 2; k++) {
        SFp2 t = {field_elem_from_64(0, 128 * k), field_elem_from_64(0, 128 * k + 64)};
        outputw(t.c0, 3, 96 * k); outputw(t.c1, 3, 96 * k + 48);
        SwuState s = swu_prepare(t);
        outputw(s.uv15.c0, 4, 96 * k); outputw(s.uv15.c1, 4, 96 * k + 48);
        swu_state_store(s, 5, 576 * k);
      }
      B.store_batch = 8;   // the state goes out as it is produced (28 slots otherwise 40)
      return B.compile("h2c_a", 8);
    }
    case P_H2C_B: {
      Pt<SFp2> pts[2];
      for (int k = 0; k < 2; k++) {
        SFp2 t = {inputw(3, 96 * k), inputw(3, 96 * k + 48)};
        SFp2 gp = {inputw(5, 96 * k), inputw(5, 96 * k + 48)};
        pts[k] = swu_finish(swu_prepare(t), gp);
      }
      Pt<SFp2> q = isogeny_g2_proj(pt_add_generic(pts[0], pts[1]));   // index.ts:487-488
      outputw(q.x.c0, 6, 0); outputw(q.x.c1, 6, 48); outputw(q.y.c0, 6, 96); outputw(q.y.c1, 6, 144); outputw(q.z.c0, 6, 192); outputw(q.z.c1, 6, 240);
      return B.compile("h2c_b", 8);
    }
    case P_H2C_B1: {
      SFp2 t = {inputw(3, 0), inputw(3, 48)}, gp = {inputw(5, 0), inputw(5, 48)};
      Pt<SFp2> q = swu_finish(swu_state_load(t, 4, 0), gp);
      outputw(q.x.c0, 6, 0); outputw(q.x.c1, 6, 48); outputw(q.y.c0, 6, 96); outputw(q.y.c1, 6, 144); outputw(q.z.c0, 6, 192); outputw(q.z.c1, 6, 240);
      B.sched_window = env_int("NBLS_H2C_B1_WINDOW", 40);
      return B.compile("h2c_b1", env_int("NBLS_H2C_B1_W", 8));
    }
    case P_H2C_B2: {
      auto ld = [&](int off) { return Pt<SFp2>{{inputw(3, off), inputw(3, off + 48)}, {inputw(3, off + 96), inputw(3, off + 144)}, {inputw(3, off + 192), inputw(3, off + 240)}}; };
      Pt<SFp2> q = isogeny_g2_proj(pt_add_generic(ld(0), ld(288)));   // index.ts:487-488
      outputw(q.x.c0, 6, 0); outputw(q.x.c1, 6, 48); outputw(q.y.c0, 6, 96); outputw(q.y.c1, 6, 144); outputw(q.z.c0, 6, 192); outputw(q.z.c1, 6, 240);
      return B.compile("h2c_b2", env_int("NBLS_H2C_B2_W", 8));
    }
    case P_H2C_NA: {
      for (int k = 0; k < 2; k++) {
        SFp2 t = {field_elem_from_64(0, 128 * k), field_elem_from_64(0, 128 * k + 64)};
        outputw(t.c0, 3, 96 * k); outputw(t.c1, 3, 96 * k + 48);
        SwuNormState s = swu_norm_prepare(t);
        outputw(s.na, 4, 48 * k);
        const SFp2* f[4] = {&s.zt2, &s.num, &s.den, &s.a};
        for (int j = 0; j < 4; j++) { outputw(f[j]->c0, 5, 768 * k + 96 * j); outputw(f[j]->c1, 5, 768 * k + 96 * j + 48); }
        outputw(s.d, 5, 768 * k + 384);
      }
      B.store_batch = 8;
      return B.compile("h2c_na", 8);
    }
    case P_H2C_NM: {
      SFp2 t = {inputw(3, 0), inputw(3, 48)}, zt2 = {inputw(4, 0), inputw(4, 48)}, num = {inputw(4, 96), inputw(4, 144)}, a = {inputw(4, 288), inputw(4, 336)};
      SwuNormMid m = swu_norm_mid(t, zt2, num, a, inputw(4, 384), inputw(5, 0));
      outputw(m.num.c0, 6, 0); outputw(m.num.c1, 6, 48); outputw(m.a1h, 6, 96); outputw(m.delta, 6, 144);
      outputw(m.g, 7, 0);
      return B.compile("h2c_nm", env_int("NBLS_H2C_NM_W", 4));
    }
    case P_H2C_NB: {
      // the state's slots 9..12 are P_H2C_NM's output (buffer 6 there): num, a1 / 2, delta
      SFp2 t = {inputw(3, 0), inputw(3, 48)}, den = {inputw(4, 192), inputw(4, 240)}, num = {inputw(4, 432), inputw(4, 480)};
      Pt<SFp2> q = swu_norm_finish(t, num, den, inputw(4, 528), inputw(4, 576), inputw(4, 384), inputw(5, 0));
      outputw(q.x.c0, 6, 0); outputw(q.x.c1, 6, 48); outputw(q.y.c0, 6, 96); outputw(q.y.c1, 6, 144); outputw(q.z.c0, 6, 192); outputw(q.z.c1, 6, 240);
      return B.compile("h2c_nb", env_int("NBLS_H2C_NB_W", 4));
    }
    case P_H2C_C: {   // its own program: chained to H2C_B through HBM so that neither keeps more than ~40 slots live (LDS-limited occupancy)
      Pt<SFp2> p = {{inputw(3, 0), inputw(3, 48)}, {inputw(3, 96), inputw(3, 144)}, {inputw(3, 192), inputw(3, 240)}};
      Pt<SFp2> q = clear_cofactor_g2(p);                               // index.ts:489, 659-672
      outputw(q.x.c0, 6, 0); outputw(q.x.c1, 6, 48); outputw(q.y.c0, 6, 96); outputw(q.y.c1, 6, 144); outputw(q.z.c0, 6, 192); outputw(q.z.c1, 6, 240);
      outputw(sqr(q.z.c0) + sqr(q.z.c1), 7, 0);
      return B.compile("h2c_c", G2_W);
    }
    case P_G1_TO_PROJ: {
      outputw(input(0, 0), 3, 0); outputw(input(0, 48), 3, 48); outputw(fp_one(), 3, 96);
      return B.compile("g1_to_proj", 4);
    }
    case P_G1_ADD2: {
      Pt<SFp> a = {inputw(3, 0), inputw(3, 48), inputw(3, 96)}, b = {inputw(3, 144), inputw(3, 192), inputw(3, 240)};
      Pt<SFp> r = pt_add(a, b);
      outputw(r.x, 5, 0); outputw(r.y, 5, 48); outputw(r.z, 5, 96);
      return B.compile("g1_add2", 4);
    }
    case P_G1_NORM: { outputw(inputw(3, 96), 4, 0); return B.compile("g1_norm", 1); }
    case P_G1_TO_AFFINE: {
      SFp X = inputw(3, 0), Y = inputw(3, 48), Z = inputw(3, 96), zi = inputw(4, 0);
      status_out({{f_not(is_zero(Z)), 1}}, 7);
      output(mul(X, zi), 2, 0); output(mul(Y, zi), 2, 48);
      return B.compile("g1_to_affine", 4);
    }
    case P_G2_TO_PROJ: {
      for (int k = 0; k < 4; k++) outputw(input(1, 48 * k), 3, 48 * k);
      outputw(fp_one(), 3, 192); outputw(SFp(), 3, 240);
      return B.compile("g2_to_proj", 8);
    }
    case P_G2_ADD2: {
      auto ld = [&](int off) { return Pt<SFp2>{{inputw(3, off), inputw(3, off + 48)}, {inputw(3, off + 96), inputw(3, off + 144)}, {inputw(3, off + 192), inputw(3, off + 240)}}; };
      Pt<SFp2> r = pt_add(ld(0), ld(288));
      outputw(r.x.c0, 5, 0); outputw(r.x.c1, 5, 48); outputw(r.y.c0, 5, 96); outputw(r.y.c1, 5, 144); outputw(r.z.c0, 5, 192); outputw(r.z.c1, 5, 240);
      return B.compile("g2_add2", 8);
    }
    case P_G2_NORM: { SFp z0 = inputw(3, 192), z1 = inputw(3, 240); outputw(sqr(z0) + sqr(z1), 4, 0); return B.compile("g2_norm", 2); }
    case P_G2_TO_AFFINE: {
      SFp2 X = {inputw(3, 0), inputw(3, 48)}, Y = {inputw(3, 96), inputw(3, 144)}, Z = {inputw(3, 192), inputw(3, 240)};
      SFp ni = inputw(4, 0);
      SFp2 zi = mat(SFp2{mul(Z.c0, ni), -mul(Z.c1, ni)});          // Fp2.invert (math.ts:522-526)
      status_out({{f_not(eq_zero(Z)), 1}}, 7);
      output_fp2(mul(X, zi), 2, 0); output_fp2(mul(Y, zi), 2, 96);
      return B.compile("g2_to_affine", 8);
    }
    case P_T_SWU: {
      SFp2 t = {inputw(3, 0), inputw(3, 48)}, gp = {inputw(5, 0), inputw(5, 48)};
      Pt<SFp2> q = swu_finish(swu_prepare(t), gp);
      outputw(q.x.c0, 6, 0); outputw(q.x.c1, 6, 48); outputw(q.y.c0, 6, 96); outputw(q.y.c1, 6, 144); outputw(q.z.c0, 6, 192); outputw(q.z.c1, 6, 240);
      return B.compile("t_swu", 8);
    }
    case P_T_ISO: case P_T_CLEAR: {
      Pt<SFp2> p = {{inputw(3, 0), inputw(3, 48)}, {inputw(3, 96), inputw(3, 144)}, {inputw(3, 192), inputw(3, 240)}};
      Pt<SFp2> q = id == P_T_ISO ? isogeny_g2_proj(p) : clear_cofactor_g2(p);
      outputw(q.x.c0, 6, 0); outputw(q.x.c1, 6, 48); outputw(q.y.c0, 6, 96); outputw(q.y.c1, 6, 144); outputw(q.z.c0, 6, 192); outputw(q.z.c1, 6, 240);
      return B.compile(id == P_T_ISO ? "t_iso" : "t_clear", 8);
    }
    case P_H2C1_A: case P_ENC1_A: {
      const int count = id == P_H2C1_A ? 2 : 1;
      for (int k = 0; k < count; k++) {
        SFp u = field_elem_from_64(0, 64 * k);
        outputw(u, 3, 48 * k);
        outputw(swu1_prepare(u).tv4, 4, 48 * k);
      }
      return B.compile(id == P_H2C1_A ? "h2c1_a" : "enc1_a", 4);
    }
    case P_H2C1_B: case P_ENC1_B: {
      const int count = id == P_H2C1_B ? 2 : 1;
      Pt<SFp> pts[2];
      for (int k = 0; k < count; k++) pts[k] = swu1_finish(swu1_prepare(inputw(3, 48 * k)), inputw(5, 48 * k));
      Pt<SFp> q = isogeny_g1_proj(count == 2 ? pt_add_generic(pts[0], pts[1]) : pts[0]);   // index.ts:336-337 / 348
      outputw(q.x, 6, 0); outputw(q.y, 6, 48); outputw(q.z, 6, 96);
      return B.compile(id == P_H2C1_B ? "h2c1_b" : "enc1_b", 8);
    }
    case P_G1_CLEAR: {
      Pt<SFp> q = clear_cofactor_g1({inputw(3, 0), inputw(3, 48), inputw(3, 96)});
      outputw(q.x, 6, 0); outputw(q.y, 6, 48); outputw(q.z, 6, 96);
      outputw(q.z, 7, 0);
      return B.compile("g1_clear", G1_W);
    }
    case P_ENC2_A: {
      SFp2 t = {field_elem_from_64(0, 0), field_elem_from_64(0, 64)};
      outputw(t.c0, 3, 0); outputw(t.c1, 3, 48);
      SwuState st = swu_prepare(t);
      outputw(st.uv15.c0, 4, 0); outputw(st.uv15.c1, 4, 48);
      return B.compile("enc2_a", 8);
    }
    case P_ENC2_B: {
      SFp2 t = {inputw(3, 0), inputw(3, 48)}, gp = {inputw(5, 0), inputw(5, 48)};
      Pt<SFp2> q = isogeny_g2_proj(swu_finish(swu_prepare(t), gp));              // index.ts:494-495
      outputw(q.x.c0, 6, 0); outputw(q.x.c1, 6, 48); outputw(q.y.c0, 6, 96); outputw(q.y.c1, 6, 144); outputw(q.z.c0, 6, 192); outputw(q.z.c1, 6, 240);
      return B.compile("enc2_b", 8);
    }
    case P_G1_COMPRESS: g1_compress(0, 2); return B.compile("g1_compress", 2);
    case P_G2_COMPRESS: g2_compress(0, 2); return B.compile("g2_compress", 4);
    case P_G1_MUL: case P_G1_MUL_W3: {
      SFp x = input(0, 0), y = input(0, 48);
      SFp k = input_raw(2, 0, 32);
      const bool w3 = id == P_G1_MUL_W3;
      Pt<SFp> r = pt_mul_ladder(pt_affine(x, y), k, 256, w3 ? 3 : env_int("NBLS_G1MUL_WIN", 2));
      outputw(r.x, 3, 0); outputw(r.y, 3, 48); outputw(r.z, 3, 96);
      outputw(r.z, 4, 0);
      B.sched_window = w3 ? 300 : env_int("NBLS_G1MUL_WINDOW", 200);   // scalar bits are extracted just in time instead of all 256 up front (they would pin 256 LDS slots)
      return B.compile(w3 ? "g1_mul_w3" : "g1_mul", G1MUL_W);
    }
    case P_G2_MUL_GLS: {
      SFp2 x = input_fp2(1, 0), y = input_fp2(1, 96);
      SFp a[4]; for (int i = 0; i < 4; i++) a[i] = input_raw(2, 32 * i, 32);
      Pt<SFp2> r = pt_mul_gls_g2(pt_affine(x, y), a);
      outputw(r.x.c0, 3, 0); outputw(r.x.c1, 3, 48); outputw(r.y.c0, 3, 96); outputw(r.y.c1, 3, 144); outputw(r.z.c0, 3, 192); outputw(r.z.c1, 3, 240);
      outputw(sqr(r.z.c0) + sqr(r.z.c1), 4, 0);
      B.sched_window = env_int("NBLS_G2GLS_WINDOW", 300);
      return B.compile("g2_mul_gls", G2MUL_W);
    }
    case P_G2_MUL_SAC: case P_G2_MUL_SAC_LS2: {
      // the base point is a raw PROJECTIVE point (buf 1, six raw elements): sign hands over the hash point as cofactor clearing leaves it, without the inversion and the affine
      // program in between (round 5: -0.12 ms per call).  The windowed form above keeps its affine input: with a projective one it holds 56 instead of 50 slots (five
      // workgroups per CU instead of six) and a 65,536-key call takes 10 % longer
      Pt<SFp2> q = {{inputw(1, 0), inputw(1, 48)}, {inputw(1, 96), inputw(1, 144)}, {inputw(1, 192), inputw(1, 240)}};
      SFp rc[4]; for (int i = 0; i < 4; i++) rc[i] = input_raw(2, 32 * i, 32);
      Pt<SFp2> r = pt_mul_sac_g2(q, rc);
      outputw(r.x.c0, 3, 0); outputw(r.x.c1, 3, 48); outputw(r.y.c0, 3, 96); outputw(r.y.c1, 3, 144); outputw(r.z.c0, 3, 192); outputw(r.z.c1, 3, 240);
      outputw(sqr(r.z.c0) + sqr(r.z.c1), 4, 0);
      B.sched_window = env_int("NBLS_G2SAC_WINDOW", 150);   // with 300 the seven additions of the table run side by side: 108 slots (two workgroups per CU); 150: 87 slots (three) at the same instruction count
      return B.compile(ls2 ? "g2_mul_sac_ls2" : "g2_mul_sac", G2MUL_W);
    }
    case P_G1_MUL_FIXED: {
      SFp k = input_raw(2, 0, 32);
      Pt<SFp> r = pt_mul_fixed_g1(k, 5);
      outputw(r.x, 3, 0); outputw(r.y, 3, 48); outputw(r.z, 3, 96);
      outputw(r.z, 4, 0);
      B.sched_window = env_int("NBLS_G1FIXED_WINDOW", 25);   // the table loads and bit extractions of a window are scheduled just ahead of its selects (they would otherwise pin 30 slots per window from the start)
      return B.compile("g1_mul_fixed", G1MUL_W);
    }
    case P_G2_MUL: case P_G2_MUL_W3: {
      SFp2 x = input_fp2(1, 0), y = input_fp2(1, 96);
      SFp k = input_raw(2, 0, 32);
      const bool w3 = id == P_G2_MUL_W3;
      Pt<SFp2> r = pt_mul_ladder(pt_affine(x, y), k, 256, w3 ? 3 : env_int("NBLS_G2MUL_WIN", 2));
      outputw(r.x.c0, 3, 0); outputw(r.x.c1, 3, 48); outputw(r.y.c0, 3, 96); outputw(r.y.c1, 3, 144); outputw(r.z.c0, 3, 192); outputw(r.z.c1, 3, 240);
      outputw(sqr(r.z.c0) + sqr(r.z.c1), 4, 0);      // Fp2 norm, inverted by the inversion kernel (Fp2.invert, math.ts:522-526)
      B.sched_window = env_int("NBLS_MUL_WINDOW", 300);
      return B.compile(w3 ? "g2_mul_w3" : "g2_mul", G2MUL_W);
    }
    case P_G1_MSM_PREP: {
      // phi(x, y) = (beta x, y) acts on G1 as [-z^2] (the reference's subgroup check compares [-z^2]P with phi(P), index.ts:424-448)
      SFp x = input(0, 0), y = input(0, 48);
      outputw(x, 3, 0); outputw(y, 3, 48); outputw(fp_one(), 3, 96);
      outputw(mul(x, fp_const(NBLS_BETA)), 3, 144); outputw(-y, 3, 192); outputw(fp_one(), 3, 240);
      return B.compile("g1_msm_prep", 4);
    }
    case P_G2_MSM_PREP: {
      // psi acts on G2 as [z] = [-|z|] (PointG2 subgroup check, index.ts:640-657): [|z|^i]Q = (-1)^i psi^i(Q)
      Pt<SFp2> q = pt_affine(input_fp2(1, 0), input_fp2(1, 96));
      Pt<SFp2> q1 = psi_proj(q), q2 = psi2_proj(q), q3 = psi_proj(q2);
      Pt<SFp2> outs[4] = {q, pt_neg(q1), q2, pt_neg(q3)};
      for (int k = 0; k < 4; k++) {
        const Pt<SFp2>& r = outs[k]; const int o = 288 * k;
        outputw(r.x.c0, 3, o); outputw(r.x.c1, 3, o + 48); outputw(r.y.c0, 3, o + 96); outputw(r.y.c1, 3, o + 144); outputw(r.z.c0, 3, o + 192); outputw(r.z.c1, 3, o + 240);
      }
      return B.compile("g2_msm_prep", 8);
    }
    case P_G1_ADD_AB: case P_G1_HORNER: case P_G1_SHIFTADD: {
      auto ld = [&](int buf, int off) { return Pt<SFp>{inputw(buf, off), inputw(buf, off + 48), inputw(buf, off + 96)}; };
      Pt<SFp> r;
      if (id == P_G1_ADD_AB) r = pt_add(ld(3, 0), ld(4, 0));
      else if (id == P_G1_HORNER) { r = ld(3, 144 * (MSM_WINDOW_BITS - 1)); for (int t = MSM_WINDOW_BITS - 2; t >= 0; t--) r = pt_add(pt_dbl(r), ld(3, 144 * t)); }
      else r = pt_add(pt_dbl_n(ld(3, 0), MSM_WINDOW_BITS), ld(4, 0));
      outputw(r.x, 5, 0); outputw(r.y, 5, 48); outputw(r.z, 5, 96);
      return B.compile(id == P_G1_ADD_AB ? "g1_add_ab" : id == P_G1_HORNER ? "g1_horner" : "g1_shiftadd", 4);
    }
    case P_G2_ADD_AB: case P_G2_HORNER: case P_G2_SHIFTADD: {
      auto ld = [&](int buf, int off) { return Pt<SFp2>{{inputw(buf, off), inputw(buf, off + 48)}, {inputw(buf, off + 96), inputw(buf, off + 144)}, {inputw(buf, off + 192), inputw(buf, off + 240)}}; };
      Pt<SFp2> r;
      if (id == P_G2_ADD_AB) r = pt_add(ld(3, 0), ld(4, 0));
      else if (id == P_G2_HORNER) { r = ld(3, 288 * (MSM_WINDOW_BITS - 1)); for (int t = MSM_WINDOW_BITS - 2; t >= 0; t--) r = pt_add(pt_dbl(r), ld(3, 288 * t)); }
      else r = pt_add(pt_dbl_n(ld(3, 0), MSM_WINDOW_BITS), ld(4, 0));
      outputw(r.x.c0, 5, 0); outputw(r.x.c1, 5, 48); outputw(r.y.c0, 5, 96); outputw(r.y.c1, 5, 144); outputw(r.z.c0, 5, 192); outputw(r.z.c1, 5, 240);
      return B.compile(id == P_G2_ADD_AB ? "g2_add_ab" : id == P_G2_HORNER ? "g2_horner" : "g2_shiftadd", 8);
    }
    case P_H2C_C0: {
      Pt<SFp2> p = {{inputw(3, 0), inputw(3, 48)}, {inputw(3, 96), inputw(3, 144)}, {inputw(3, 192), inputw(3, 240)}}, v, u;
      clear_cofactor_g2_pre(p, v, u);
      const Pt<SFp2>* o[2] = {&v, &u};
      for (int k = 0; k < 2; k++) { const Pt<SFp2>& q = *o[k]; const int b = k ? 5 : 6; outputw(q.x.c0, b, 0); outputw(q.x.c1, b, 48); outputw(q.y.c0, b, 96); outputw(q.y.c1, b, 144); outputw(q.z.c0, b, 192); outputw(q.z.c1, b, 240); }
      return B.compile("h2c_c0", G2_W);
    }
    case P_H2C_C1: case P_H2C_C1_LS2: {
      auto ld = [&](int buf) { return Pt<SFp2>{{inputw(buf, 0), inputw(buf, 48)}, {inputw(buf, 96), inputw(buf, 144)}, {inputw(buf, 192), inputw(buf, 240)}}; };
      Pt<SFp2> p = ld(3), base, t1;
      clear_cofactor_g2_first(p, ld(6), base, t1);
      const Pt<SFp2>* o[2] = {&base, &t1};
      for (int k = 0; k < 2; k++) { const Pt<SFp2>& q = *o[k]; const int b = k ? 3 : 6; outputw(q.x.c0, b, 0); outputw(q.x.c1, b, 48); outputw(q.y.c0, b, 96); outputw(q.y.c1, b, 144); outputw(q.z.c0, b, 192); outputw(q.z.c1, b, 240); }
      B.sched_window = env_int("NBLS_CLEAR_WINDOW", 100);   // psi(P) is only needed after the ladder: its load must not occupy slots throughout
      return B.compile(ls2 ? "h2c_c1_ls2" : "h2c_c1", G2_W);
    }
    case P_H2C_C2: case P_H2C_C2_LS2: {
      auto ld = [&](int buf) { return Pt<SFp2>{{inputw(buf, 0), inputw(buf, 48)}, {inputw(buf, 96), inputw(buf, 144)}, {inputw(buf, 192), inputw(buf, 240)}}; };
      Pt<SFp2> q = clear_cofactor_g2_second(ld(3), ld(4), ld(5));
      outputw(q.x.c0, 6, 0); outputw(q.x.c1, 6, 48); outputw(q.y.c0, 6, 96); outputw(q.y.c1, 6, 144); outputw(q.z.c0, 6, 192); outputw(q.z.c1, 6, 240);
      outputw(sqr(q.z.c0) + sqr(q.z.c1), 7, 0);
      B.sched_window = env_int("NBLS_CLEAR_WINDOW", 100);   // t1 and u likewise
      return B.compile(ls2 ? "h2c_c2_ls2" : "h2c_c2", G2_W);
    }
    case P_G2_DEC_A192: g2_decompress_A(0, 3, 4, true); return B.compile("g2_dec_a192", 4);
    case P_G2_DEC_B192: g2_decompress_B(0, 3, 4, 5, 6, 7, 1); return B.compile("g2_dec_b192", G2_W);
    case P_G2_DEC_B_HEX: g2_decompress_B(0, 3, 4, 5, 6, 7, 2); return B.compile("g2_dec_b_hex", G2_W);
    case P_G1_FROM_RAW: g1_from_raw(0, 6, 7); return B.compile("g1_from_raw", G1_W);
    case P_G2_FROM_RAW: g2_from_raw(0, 6, 7); return B.compile("g2_from_raw", G2_W);
    case P_G2_SWAP: g2_swap_halves(0, 2); return B.compile("g2_swap", 4);
    case P_LINES_PQ: {
      SFp Px, Py; SFp2 Qx, Qy; load_points(Px, Py, Qx, Qy);
      LineSink o{3, 0, &Px, &Py};
      B.light_max = MAX_DOT_PRODUCTS;   // the ten lane-ops of a level (1..3 products) belong in ONE step
      B.neg_cap = 7.0; B.store_batch = 6;   // t2 = 3 xi D^2 is bounded by 6.2 p: subtracting it as it is saves a contraction level per bit
      trace_lines(Qx, Qy, o);
      return B.compile("lines_pq", LINES_W);
    }
    case P_LINES_Q: {
      SFp2 Qx = input_fp2(1, 0), Qy = input_fp2(1, 96);
      LineSink o{3, 0, nullptr, nullptr};
      B.light_max = MAX_DOT_PRODUCTS; B.neg_cap = 7.0; B.store_batch = 6;
      trace_lines(Qx, Qy, o);
      return B.compile("lines_q", LINES_W);
    }
    case P_LINES_BYTES: {   // one ITEM per line triple (launched over 68 n items): 6 raw elements (buf 3) -> c0 || c1 || c2 as Fp2.toBytes (buf 2)
      for (int e = 0; e < 6; e++) output(inputw(3, 48 * e), 2, 48 * e);
      return B.compile("lines_bytes", 6);
    }
    case P_LINES_FROM_BYTES: {   // the inverse: 288 wire bytes (buf 2) -> 6 raw elements (buf 3)
      for (int e = 0; e < 6; e++) outputw(input(2, 48 * e), 3, 48 * e);
      return B.compile("lines_from_bytes", 6);
    }
    case P_ACC_BYTES: { B.sched_window = ACC_WINDOW; output_fp12(trace_acc(1, 3), 2, 0); return B.compile("acc_bytes", ACC_W); }
    case P_ACC_RAW: { B.sched_window = ACC_WINDOW; outputw_fp12(trace_acc(1, 3), 5, 0); return B.compile("acc_raw", ACC_W); }
    case P_ACC_FE: {
      B.sched_window = ACC_WINDOW;
      SFp12 f = mat(trace_acc(1, 3));
      outputw_fp12(f, 5, 0);
      outputw(inv_chain(f).n, 4, 0);
      return B.compile("acc_fe", ACC_W);
    }
    case P_ACC2_RAW: { B.sched_window = ACC_WINDOW; outputw_fp12(trace_acc(2, 3), 5, 0); return B.compile("acc2_raw", ACC_W); }
    case P_ACC4_RAW: { B.sched_window = ACC_WINDOW; outputw_fp12(trace_acc(4, 3), 5, 0); return B.compile("acc4_raw", ACC_W); }
    case P_ACC8_RAW: {   // window 250: 24 slots, the 80-byte slot stride fits twelve workgroups per CU (330: 30 slots)
      B.sched_window = env_int("NBLS_ACC8_WINDOW", 250); outputw_fp12(trace_acc(8, 3), 5, 0); return B.compile("acc8_raw", ACC_W);
    }
    case P_ACC_Q: {
      B.sched_window = ACC_WINDOW;
      std::vector<SFp> Px{input(0, 0)}, Py{input(0, 48)};
      outputw_fp12(trace_acc(1, 3, &Px, &Py), 5, 0);
      return B.compile("acc_q", ACC_W);
    }
    default: break;
  }
  return Program();
}


// ---------------------------------------------------------------- single tower operations (nbls_tower_op_batch)
// op codes = include/nbls.h NBLS_TOP_*
namespace {
enum { T_ADD = 0, T_SUB, T_NEG, T_MUL, T_SQR, T_INV, T_FROB, T_CONJ, T_MULNR, T_MUL_BY_B, T_MUL_BY_1, T_MUL_BY_01, T_MUL_BY_014, T_CYC_SQR, T_CYC_EXP, T_COUNT };
SFp6 in_fp6(int buf, int off) { return {input_fp2(buf, off), input_fp2(buf, off + 96), input_fp2(buf, off + 192)}; }
void out_fp6(const SFp6& a, int buf, int off) { output_fp2(a.c0, buf, off); output_fp2(a.c1, buf, off + 96); output_fp2(a.c2, buf, off + 192); }
// inverses: Fp2.invert math.ts:522-526, Fp6.invert 672-680, Fp12.invert 793-797 -- everything around the one Fp inversion (the inversion kernel)
struct Inv6 { SFp2 T0, T1, T2, d; };
Inv6 inv6_chain(const SFp6& t) {
  Inv6 c;
  c.T0 = mat(sqr(t.c0) - mulnr(mul(t.c2, t.c1)));
  c.T1 = mat(mulnr(sqr(t.c2)) - mul(t.c0, t.c1));
  c.T2 = mat(sqr(t.c1) - mul(t.c0, t.c2));
  c.d = mat(mulnr(mul(t.c2, c.T1) + mul(t.c1, c.T2)) + mul(t.c0, c.T0));
  return c;
}
SFp6 inv6_finish(const Inv6& c, const SFp& ninv) {
  SFp2 dinv = {mul(c.d.c0, ninv), -mul(c.d.c1, ninv)};
  return {mul(dinv, c.T0), mul(dinv, c.T1), mul(dinv, c.T2)};
}
Program build_tower(int field, int op, int param, int part) {
  Builder B;
  char name[64]; snprintf(name, sizeof name, "tower_f%d_op%d_%d_%d", field, op, param, part);
  const int W = field == 12 ? 12 : field == 6 ? 6 : field == 2 ? 2 : 1;
  if (field == 1) {
    SFp a = input(0, 0);
    if (op == T_INV) { if (part == 0) outputw(a, 4, 0); else output(inputw(5, 0), 7, 0); return B.compile(name, 4); }
    SFp r;
    switch (op) {
      case T_ADD: r = a + input(1, 0); break;
      case T_SUB: r = a - input(1, 0); break;
      case T_NEG: r = -a; break;
      case T_MUL: r = mul(a, input(1, 0)); break;
      case T_SQR: r = sqr(a); break;
      default: return Program();
    }
    output(r, 7, 0);
    return B.compile(name, 4);
  }
  if (field == 2) {
    SFp2 a = input_fp2(0, 0);
    if (op == T_INV) {
      if (part == 0) outputw(sqr(a.c0) + sqr(a.c1), 4, 0);
      else { SFp ni = inputw(5, 0); output_fp2({mul(a.c0, ni), -mul(a.c1, ni)}, 7, 0); }
      return B.compile(name, 4);
    }
    SFp2 r;
    switch (op) {
      case T_ADD: r = a + input_fp2(1, 0); break;
      case T_SUB: r = a - input_fp2(1, 0); break;
      case T_NEG: r = -a; break;
      case T_MUL: r = mul(a, input_fp2(1, 0)); break;
      case T_SQR: r = sqr(a); break;
      case T_FROB: r = frob(a, param); break;
      case T_CONJ: r = conj(a); break;
      case T_MULNR: r = mulnr(a); break;
      case T_MUL_BY_B: r = mul_by_b(a); break;
      default: return Program();
    }
    output_fp2(r, 7, 0);
    return B.compile(name, 4);
  }
  if (field == 6) {
    SFp6 a = in_fp6(0, 0);
    if (op == T_INV) {
      Inv6 c = inv6_chain(a);
      if (part == 0) outputw(sqr(c.d.c0) + sqr(c.d.c1), 4, 0);
      else out_fp6(inv6_finish(c, inputw(5, 0)), 7, 0);
      return B.compile(name, 8);
    }
    SFp6 r;
    switch (op) {
      case T_ADD: r = a + in_fp6(1, 0); break;
      case T_SUB: r = a - in_fp6(1, 0); break;
      case T_NEG: r = -a; break;
      case T_MUL: r = mul(a, in_fp6(1, 0)); break;
      case T_SQR: r = sqr(a); break;
      case T_FROB: r = frob(a, param); break;
      case T_MULNR: r = mulnr(a); break;
      case T_MUL_BY_1: r = mul_by_1(a, input_fp2(1, 0)); break;
      case T_MUL_BY_01: r = mul_by_01(a, input_fp2(1, 0), input_fp2(2, 0)); break;
      default: return Program();
    }
    out_fp6(r, 7, 0);
    return B.compile(name, 8);
  }
  if (field == 12) {
    SFp12 a = mat(input_fp12(0, 0));
    if (op == T_INV) {
      InvChain c = inv_chain(a);
      if (part == 0) outputw(c.n, 4, 0);
      else output_fp12(inv_finish(a, c, inputw(5, 0)), 7, 0);
      return B.compile(name, 16);
    }
    SFp12 r;
    switch (op) {
      case T_ADD: { SFp12 b = input_fp12(1, 0); r = {a.c0 + b.c0, a.c1 + b.c1}; break; }
      case T_SUB: { SFp12 b = input_fp12(1, 0); r = {a.c0 - b.c0, a.c1 - b.c1}; break; }
      case T_NEG: r = {-a.c0, -a.c1}; break;
      case T_MUL: r = mul(a, mat(input_fp12(1, 0))); break;
      case T_SQR: r = sqr(a); break;
      case T_FROB: r = frob(a, param); break;
      case T_CONJ: r = conj(a); break;
      case T_MUL_BY_014: r = mul_by_014(a, input_fp2(1, 0), input_fp2(2, 0), input_fp2(3, 0)); break;
      case T_CYC_SQR: r = cyclotomic_sqr(a); break;
      case T_CYC_EXP: r = cyclotomic_exp_x(a); break;
      default: return Program();
    }
    output_fp12(r, 7, 0);
    return B.compile(name, op == T_CYC_EXP ? 12 : 16);
  }
  (void)W;
  return Program();
}
}  // namespace
const Program* get_tower_program(int field, int op, int param, int part) {
  static std::map<std::tuple<int, int, int, int>, Program> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> g(mu);
  if (!(field == 1 || field == 2 || field == 6 || field == 12) || op < 0 || op >= T_COUNT || part < 0 || part > 1 || (part == 1 && op != T_INV)) return nullptr;
  if (op != T_FROB) param = 0; else if (param < 0 || param > 11) return nullptr;
  auto key = std::make_tuple(field, op, param, part);
  auto it = cache.find(key);
  if (it == cache.end()) it = cache.emplace(key, build_tower(field, op, param, part)).first;
  return it->second.steps.empty() ? nullptr : &it->second;
}

const Program& get_program(ProgId id) {
  static Program cache[P_COUNT];
  static bool built[P_COUNT];
  static std::mutex mu;
  std::lock_guard<std::mutex> g(mu);
  if (!built[id]) { cache[id] = build(id); built[id] = true; }
  return cache[id];
}

void print_stats(const Program& p) {
  printf("%-14s W=%2u G=%u steps=%5zu (dot %4u, lin %4u, other %3u)  dot_ops=%6u products=%6u (product-slot fill %.2f)  lin_ops=%6u terms=%6u  slots=%4u  lds=%6u B  descs=%zu KB  operands: combined %u normalised %u; round operands %u: single %u sum %u diff %u mixed %u norm %u  est_valu=%.0fk\n",
         p.name.c_str(), p.W, p.G, p.steps.size(), p.n_dot_steps, p.n_lin_steps, p.n_other_steps, p.n_dot_ops, p.n_products,
         p.n_prod_slots ? (double)p.n_products / p.n_prod_slots : 0.0, p.n_lin_ops, p.n_lin_terms, p.slots, p.lds_bytes(), p.descs.size() * 4 / 1024, p.n_comb_operands, p.n_norm_operands, p.n_round_ops, p.n_op_mode[0], p.n_op_mode[1], p.n_op_mode[2], p.n_op_mode[3], p.n_op_norm, p.est_valu / 1e3);
}

}  // namespace nbls
