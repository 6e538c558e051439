// fp_inv.h -- modular inverse of one Fp element per lane.  Stands for Fp.invert (reference math.ts:134-156, 239-241:
// extended Euclid on bigints); the canonical result is the same field element.
//
// Algorithm: Pornin's optimised binary GCD (eprint 2020/972, alg. 2) on the 28-bit limb representation, k = 28:
// 28 outer rounds, each running 28 branch-free inner iterations on 58-bit approximations of (a, b) -- the top 30 and
// the low 28 bits -- that accumulate update factors f0,g0,f1,g1 (|f|+|g| <= 2^28); the factors are then applied to the
// full-size (a, b) exactly (a' = (a f0 + b g0) / 2^28) and to (u, v) with one Montgomery step (u' = (u f0 + v g0) / 2^28
// mod p), which keeps the invariants a = u y, b = v y (mod p).  28 * 28 = 784 >= 2 * 392 - 1 iterations: enough for any
// y < 2^392.  Every lane executes the same instruction stream (no data-dependent branches), which is what a wavefront
// wants; the instruction count is ~5x below the word-serial Kaliski loop it replaces.
// Shared by the HIP kernel (pow_kernels.hip) and the test-only simulator.
#pragma once
#include "vm_exec.h"

namespace nbls {

NBLS_HD int clz64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __clzll((long long)x);
#else
  return x ? __builtin_clzll(x) : 64;
#endif
}

// in : y = a * 2^392 mod p as 14 normalised limbs, any representative below 2^392 ; a == 0 returns 0
// out: a^-1 * 2^392 mod p, 14 normalised limbs, < 2p
NBLS_HD void fp_mont_inverse(u32* out, const u32* y) {
  const u32 P[NL] = NBLS_P28;
  const u32 R3[NL] = NBLS_R3_INIT;
  u32 a[NL], b[NL], u[NL], v[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) { a[i] = y[i]; b[i] = P[i]; u[i] = 0; v[i] = 0; }
  u[0] = 1;
  for (int round = 0; round < 28; round++) {
    // approximations: low limb exactly + the top 30 bits of a 64-bit window that starts at the highest limb where
    // a | b is non-zero (three limbs: 28 + 28 + 8 bits); numbers below 2^58 are taken exactly
    u32 c0 = ~0u, c1 = ~0u, c2 = ~0u, a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
#pragma unroll
    for (int j = NL - 1; j >= 0; j--) {
      const u32 aw = a[j], bw = b[j];
      a0 ^= (a0 ^ aw) & c0; a1 ^= (a1 ^ aw) & c1; a2 ^= (a2 ^ aw) & c2;
      b0 ^= (b0 ^ bw) & c0; b1 ^= (b1 ^ bw) & c1; b2 ^= (b2 ^ bw) & c2;
      c2 = c1; c1 = c0;
      c0 &= ((aw | bw) == 0) ? ~0u : 0u;
    }
    u32 hi = 0;
#pragma unroll
    for (int j = 3; j < NL; j++) hi |= a[j] | b[j];
    const bool small = hi == 0 && ((a[2] | b[2]) >> 2) == 0;
    const u64 A = ((u64)a0 << 36) | ((u64)a1 << 8) | (a2 >> 20), B = ((u64)b0 << 36) | ((u64)b1 << 8) | (b2 >> 20);
    int sh = clz64(A | B); if (sh > 63) sh = 0;
    u64 abar = (((A << sh) >> 34) << 28) | a[0];
    u64 bbar = (((B << sh) >> 34) << 28) | b[0];
    if (small) { abar = ((u64)a[2] << 56) | ((u64)a[1] << 28) | a[0]; bbar = ((u64)b[2] << 56) | ((u64)b[1] << 28) | b[0]; }
    i32 f0 = 1, g0 = 0, f1 = 0, g1 = 1;
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
    // (a, b) <- (a f0 + b g0, a f1 + b g1) / 2^28, made non-negative (the factors follow the sign)
    u32 na[NL], nb[NL];
    i64 ca = 0, cb = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      ca += (i64)(i32)a[i] * f0 + (i64)(i32)b[i] * g0;
      cb += (i64)(i32)a[i] * f1 + (i64)(i32)b[i] * g1;
      if (i > 0) { na[i - 1] = (u32)ca & LMASK; nb[i - 1] = (u32)cb & LMASK; }
      ca >>= 28; cb >>= 28;
    }
    na[NL - 1] = (u32)ca; nb[NL - 1] = (u32)cb;
    const u32 sa = ca < 0 ? ~0u : 0u, sb = cb < 0 ? ~0u : 0u;
    u32 cya = sa & 1, cyb = sb & 1;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
      u32 ta = (na[i] ^ (sa & LMASK)) + cya; a[i] = ta & LMASK; cya = ta >> 28;
      u32 tb = (nb[i] ^ (sb & LMASK)) + cyb; b[i] = tb & LMASK; cyb = tb >> 28;
    }
    a[NL - 1] = (na[NL - 1] ^ sa) + cya; b[NL - 1] = (nb[NL - 1] ^ sb) + cyb;
    f0 = (i32)(((u32)f0 ^ sa) - sa); g0 = (i32)(((u32)g0 ^ sa) - sa);
    f1 = (i32)(((u32)f1 ^ sb) - sb); g1 = (i32)(((u32)g1 ^ sb) - sb);
    // (u, v) <- (u f0 + v g0, u f1 + v g1) / 2^28 mod p : one Montgomery step each, result folded back into [0, p)
    const u32 qu = ((u[0] * (u32)f0 + v[0] * (u32)g0) * NBLS_N0_28) & LMASK;
    const u32 qv = ((u[0] * (u32)f1 + v[0] * (u32)g1) * NBLS_N0_28) & LMASK;
    u32 nu[NL], nv[NL];
    i64 cu = 0, cv = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      cu += (i64)(i32)u[i] * f0 + (i64)(i32)v[i] * g0 + (i64)((u64)qu * P[i]);
      cv += (i64)(i32)u[i] * f1 + (i64)(i32)v[i] * g1 + (i64)((u64)qv * P[i]);
      if (i > 0) { nu[i - 1] = (u32)cu & LMASK; nv[i - 1] = (u32)cv & LMASK; }
      cu >>= 28; cv >>= 28;
    }
    nu[NL - 1] = (u32)cu; nv[NL - 1] = (u32)cv;
    const u32 su = cu < 0 ? ~0u : 0u, sv = cv < 0 ? ~0u : 0u;   // in (-p, 2p): add p if negative, then subtract p if >= p
#pragma unroll
    for (int i = 0; i < NL; i++) { nu[i] += P[i] & su; nv[i] += P[i] & sv; }
    carry_norm(nu); carry_norm(nv);
    csub_p(nu); csub_p(nv);
#pragma unroll
    for (int i = 0; i < NL; i++) { u[i] = nu[i]; v[i] = nv[i]; }
  }
  // b = gcd = 1 and v = y^-1 mod p ;  a^-1 R = y^-1 R^2 = v * R^3 / R
  mont_mul28(out, v, R3);
}

}  // namespace nbls
