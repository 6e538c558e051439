// pow_exec.h -- fixed-exponent powers with one element per lane (Fp) or per lane pair (Fp2): the serial chains behind Fp.sqrt / Fp2.sqrt / sqrt_div_fp2
// (reference math.ts:251-264, 521-538, 1196-1198).  Written once and compiled twice like fp_inv.h: into pow_kernels.hip (gfx950) and into the test-only
// simulator, which runs the same sequences on the host (tests/test_vm_sim.py checks them against a plain square-and-multiply).
//
// Round 5:
//   * Fp squarings are squarings: 14 squares + 91 doubled cross products into the 28 columns (105 multiply-adds) instead of the general product's 196.
//   * Fp2 operands are SIGNED limb vectors (the VM's form, vm_exec.h): a0 - a1 and -a1 are plain limb-wise subtractions, the bias p * R in the upper columns
//     keeps the reduced value non-negative; the 16p bias + carry pass per operand of round 4 is gone (two of them per squaring).
//   * The exponents are public constants, so the chains are SLIDING windows of up to five bits over a table of the sixteen odd powers (377 squarings + 16 + ~64
//     multiplications where 4-bit fixed windows took 376 + 14 + ~88).  An op list = pairs of bytes (squarings to do, table index of the odd power to multiply
//     by afterwards or 0xff); the first op only names the table entry the chain starts from.
#pragma once
#include <vector>
#include "vm_exec.h"

namespace nbls {

static const int POW_TAB = 17;          // table entries per lane in the scratch buffer: sixteen odd powers + one spare (the Fp2 kernel parks a^tail there)
static const int POW_WINDOW = 5;

// acc += a * a for a vector of limbs in (-2^29, 2^29): 14 squares + 91 products with the doubled operand
NBLS_HD void sqr28(u64* acc, const u32* a) {
  u32 d[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) d[i] = a[i] << 1;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    acc[2 * i] = (u64)((i64)acc[2 * i] + (i64)(i32)a[i] * (i64)(i32)a[i]);
#pragma unroll
    for (int j = i + 1; j < NL; j++) acc[i + j] = (u64)((i64)acc[i + j] + (i64)(i32)a[i] * (i64)(i32)d[j]);
  }
}
NBLS_HD void mont_sqr28(u32* r, const u32* a) {
  u64 acc[2 * NL];
#pragma unroll
  for (int i = 0; i < 2 * NL; i++) acc[i] = 0;
  sqr28(acc, a);
  redc28(r, acc);
}

// Fp2 with one component per lane: lane parity r owns component c_r of every value and computes component r of every product; the partner's component (Y, V)
// comes from the neighbouring lane (device: a DPP lane swap).  Results are normalised and below 2p + (|V| / R); operands may be any non-negative normalised values
// below 16p.
// component r of a^2 (math.ts:477-484): r = 0: (a0 + a1)(a0 - a1) ; r = 1: (2 a0) a1
NBLS_HD void fp2_sqr_c(u32* res, const u32* X, const u32* Y, bool r) {
  u32 o1[NL], o2[NL];
#pragma unroll
  for (int k = 0; k < NL; k++) { o1[k] = (r ? Y[k] : X[k]) + Y[k]; o2[k] = r ? X[k] : X[k] - Y[k]; }
  u64 acc[2 * NL];
  acc_init(acc, 1);
  mac28(acc, o1, o2);
  redc28(res, acc);
}
// component r of a * b; X, U: own components of a, b; Y, V: the partner's.  r = 0: a0 b0 - a1 b1 ; r = 1: a1 b0 + a0 b1
NBLS_HD void fp2_mul_c(u32* res, const u32* X, const u32* Y, const u32* U, const u32* V, bool r) {
  u32 B1[NL], B2[NL], A2[NL];
#pragma unroll
  for (int k = 0; k < NL; k++) { B1[k] = r ? V[k] : U[k]; B2[k] = r ? U[k] : V[k]; A2[k] = r ? Y[k] : 0u - Y[k]; }
  u64 acc[2 * NL];
  acc_init(acc, 1);
  mac28(acc, X, B1); mac28(acc, A2, B2);
  redc28(res, acc);
}

// The chains, over a policy O that says what a value is and how the lane(s) holding it multiply:
//   O::V                      a value (device: the lane's 14 limbs; host: both components)
//   sqr(r, a), mul(r, a, b)   r may alias a
//   conj(r, a)                Fp2 only
//   one(r), load(r), store(a), tab_put(j, a), tab_get(r, j)
// a^e for the op list of e
template <class O>
NBLS_HD void pow_chain(O& o, typename O::V& acc, const unsigned char* __restrict__ ops, int nops) {
  o.tab_get(acc, ops[1]);
  for (int k = 1; k < nops; k++) {
    const unsigned nsq = ops[2 * k], idx = ops[2 * k + 1];   // uniform over the wavefront
    for (unsigned s = 0; s < nsq; s++) o.sqr(acc, acc);
    if (idx != 0xffu) { typename O::V e; o.tab_get(e, idx); o.mul(acc, acc, e); }
  }
}
// table of the odd powers b, b^3, .., b^31
template <class O>
NBLS_HD void pow_table(O& o, const typename O::V& b) {
  typename O::V b2, t;
  o.sqr(b2, b);
  o.tab_put(0, b);
  o.copy(t, b);
  for (int j = 1; j < 16; j++) { o.mul(t, t, b2); o.tab_put(j, t); }
}
template <class O>
NBLS_HD void fp_pow_seq(O& o, const unsigned char* __restrict__ ops, int nops) {
  typename O::V x, acc;
  o.load(x);
  pow_table(o, x);
  pow_chain(o, acc, ops, nops);
  o.store(acc);
}
// a^e in Fp2 for the two exponents of the square roots, e = (p^2 + 7) / 16 (decompression, math.ts:547-561) and (p^2 - 9) / 16 (SWU, math.ts:1196-1198).  With
// p = 16 K + 11:   (p^2 + 7) / 16 = K p + 11 K + 8   and   (p^2 - 9) / 16 = K p + 11 K + 7,   and a^p = conj(a), so
//        a^e = (conj(a) a^11)^K  a^tail ,   tail = 8 or 7:
// ONE 377-bit exponent on the base b = conj(a) a^11 and a handful of products for a^2 .. a^11 (round 1's joint double exponentiation a^c0 conj(a)^c1 took 382
// squarings + ~193 multiplications).  ops = the op list of K.
template <class O>
NBLS_HD void fp2_pow_seq(O& o, const unsigned char* __restrict__ ops, int nops, int tail) {
  typename O::V acc;
  {
    typename O::V a1, a2, a3, a4, a8, x;
    o.load(a1);                                    // contracted below 2p by the policy
    o.sqr(a2, a1); o.mul(a3, a2, a1); o.sqr(a4, a2); o.sqr(a8, a4);
    if (tail == 7) o.mul(x, a4, a3); else o.copy(x, a8);
    o.tab_put(16, x);                              // a^tail
    o.mul(x, a8, a3);                              // a^11
    o.conj(a2, a1);
    o.mul(x, a2, x);                               // b = conj(a) a^11
    pow_table(o, x);
  }
  pow_chain(o, acc, ops, nops);
  typename O::V e;
  o.tab_get(e, 16);
  o.mul(acc, acc, e);
  o.store(acc);
}

// host side: op list of an exponent given as 64-bit words (least significant first): left-to-right sliding windows of up to POW_WINDOW bits
inline std::vector<unsigned char> pow_make_ops(const uint64_t* e, int nbits) {
  auto bit = [&](int i) { return (int)((e[i >> 6] >> (i & 63)) & 1); };
  std::vector<unsigned char> ops;
  int i = nbits - 1, pending = 0;
  while (i >= 0 && !bit(i)) i--;
  bool first = true;
  while (i >= 0) {
    if (!bit(i)) { pending++; i--; continue; }
    int j = i - POW_WINDOW + 1; if (j < 0) j = 0;
    while (!bit(j)) j++;
    int val = 0; for (int k = i; k >= j; k--) val = 2 * val + bit(k);
    int nsq = pending + (i - j + 1);
    if (first) nsq = 0;
    while (nsq > 255) { ops.push_back(255); ops.push_back(0xff); nsq -= 255; }
    ops.push_back((unsigned char)nsq); ops.push_back((unsigned char)((val - 1) / 2));
    first = false; pending = 0; i = j - 1;
  }
  while (pending > 0) { const int c = pending > 255 ? 255 : pending; ops.push_back((unsigned char)c); ops.push_back(0xff); pending -= c; }
  return ops;
}

}  // namespace nbls
