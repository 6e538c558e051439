// fp_inv.h -- Montgomery inverse of one Fp element per lane (Kaliski's almost-inverse on 32-bit words + one table-driven
// Montgomery product).  Stands for Fp.invert (reference math.ts:134-156, 239-241: extended Euclid on bigints); the
// canonical result is the same field element.  Shared by the HIP kernel (pow_kernels.hip) and the test-only simulator.
#pragma once
#include "vm_exec.h"

namespace nbls {

NBLS_HD u32 addc32(u32 a, u32 b, u32 cin, u32* cout) { u64 s = (u64)a + b + cin; *cout = (u32)(s >> 32); return (u32)s; }
NBLS_HD u32 subb32(u32 a, u32 b, u32 bin, u32* bout) { u64 d = (u64)a - b - bin; *bout = (u32)(d >> 63); return (u32)d; }
NBLS_HD bool csub12(u32* x, const u32* m) {   // if (x >= m) x -= m ; returns whether it subtracted
  u32 d[12], br = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) d[i] = subb32(x[i], m[i], br, &br);
#pragma unroll
  for (int i = 0; i < 12; i++) x[i] = br ? x[i] : d[i];
  return br == 0;
}

// in : x = a * 2^392 mod p as 14 normalised limbs, any representative below 2^384 ; a != 0 (a == 0 returns 0)
// out: a^-1 * 2^392 mod p, 14 normalised limbs, < 2p
// pow2_table[j] = 2^(414 + j) mod p as 14 limbs, j = 0..381  (see make_inv_table)
NBLS_HD void fp_mont_inverse(u32* out, const u32* x, const u32* __restrict__ pow2_table) {
  const u32 P[12] = NBLS_P_WORDS_INIT;
  u32 u[12], v[12], r[12], s[12];
  limbs_to_words(v, x);
  for (int it = 0; it < 10; it++) if (!csub12(v, P)) break;   // canonical
#pragma unroll
  for (int i = 0; i < 12; i++) { u[i] = P[i]; r[i] = 0; s[i] = 0; }
  s[0] = 1;
  int k = 0;
  for (int it = 0; it < 768; it++) {
    u32 vz = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) vz |= v[i];
    if (vz == 0) break;
    if ((u[0] & 1) == 0) {
#pragma unroll
      for (int i = 0; i < 11; i++) u[i] = (u[i] >> 1) | (u[i + 1] << 31);
      u[11] >>= 1;
#pragma unroll
      for (int i = 11; i > 0; i--) s[i] = (s[i] << 1) | (s[i - 1] >> 31);
      s[0] <<= 1;
    } else if ((v[0] & 1) == 0) {
#pragma unroll
      for (int i = 0; i < 11; i++) v[i] = (v[i] >> 1) | (v[i + 1] << 31);
      v[11] >>= 1;
#pragma unroll
      for (int i = 11; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 31);
      r[0] <<= 1;
    } else {
      u32 d[12], br = 0;
#pragma unroll
      for (int i = 0; i < 12; i++) d[i] = subb32(v[i], u[i], br, &br);
      if (br) {    // u > v: u = (u - v)/2, r += s, s *= 2
        u32 b2 = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) d[i] = subb32(u[i], v[i], b2, &b2);
#pragma unroll
        for (int i = 0; i < 11; i++) u[i] = (d[i] >> 1) | (d[i + 1] << 31);
        u[11] = d[11] >> 1;
        u32 c = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) r[i] = addc32(r[i], s[i], c, &c);
#pragma unroll
        for (int i = 11; i > 0; i--) s[i] = (s[i] << 1) | (s[i - 1] >> 31);
        s[0] <<= 1;
      } else {     // v >= u: v = (v - u)/2, s += r, r *= 2
#pragma unroll
        for (int i = 0; i < 11; i++) v[i] = (d[i] >> 1) | (d[i + 1] << 31);
        v[11] = d[11] >> 1;
        u32 c = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = addc32(s[i], r[i], c, &c);
#pragma unroll
        for (int i = 11; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 31);
        r[0] <<= 1;
      }
    }
    k++;
  }
  // almost inverse: r in [0, 2p);  x^-1 * 2^k = p - r
  csub12(r, P);
  u32 t[12], br = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) t[i] = subb32(P[i], r[i], br, &br);   // in (0, p]
  // a^-1 R = t * 2^(3*392 - k) / R  (Montgomery product with the table entry);  k in [381, 762] for a != 0
  int j = 762 - k; if (j < 0) j = 0; if (j > 381) j = 381;
  u32 tl[NL], c[NL];
  words_to_limbs(tl, t);
#pragma unroll
  for (int i = 0; i < NL; i++) c[i] = pow2_table[NL * j + i];
  mont_mul28(out, tl, c);
  if (k == 0) {
#pragma unroll
    for (int i = 0; i < NL; i++) out[i] = 0;
  }
}

// host: table[j] = 2^(414 + j) mod p, j = 0..381, as 14 limbs each
static inline void make_inv_table(u32* table) {
  const u32 P[12] = NBLS_P_WORDS_INIT;
  u32 x[12] = {0}; x[0] = 1;
  for (int e = 0; e < 414 + 382; e++) {
    if (e >= 414) words_to_limbs(table + NL * (e - 414), x);
    u32 c = 0;
    for (int i = 0; i < 12; i++) { u32 n = (x[i] << 1) | c; c = x[i] >> 31; x[i] = n; }   // < 2p < 2^384
    csub12(x, P);
  }
}

}  // namespace nbls
