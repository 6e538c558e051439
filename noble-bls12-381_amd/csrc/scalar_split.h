// scalar_split.h -- the scalar side of the endomorphism splits: base-|z| digits of a 256-bit scalar and their sign-aligned recoding.  Written once and compiled twice like
// fp_inv.h / pow_exec.h: into msm_kernels.hip (one thread per scalar) and into the test-only simulator (tests/test_vm_sim.py checks both against Python integers).
// Everything here is branch-free in the scalar: sign feeds SECRET keys through it.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define NBLS_SS_HD __host__ __device__ inline
#else
#define NBLS_SS_HD inline
#endif

namespace nbls {

// limbs (little-endian 64-bit) /= |z|, returns the remainder; bitwise long division, |z| = 0xd201000000010000 has its top bit set
NBLS_SS_HD uint64_t ss_div_step(uint64_t* limbs, int nl) {
  const uint64_t Z = 0xd201000000010000ull;
  uint64_t rem = 0;
  for (int i = nl - 1; i >= 0; i--) {
    uint64_t q = 0; const uint64_t v = limbs[i];
    for (int b = 63; b >= 0; b--) {
      const uint64_t carry = rem >> 63;
      rem = (rem << 1) | ((v >> b) & 1);
      const uint64_t ge = (uint64_t)(carry | (uint64_t)(rem >= Z));
      rem -= Z & (0 - ge);
      q = (q << 1) | ge;
    }
    limbs[i] = q;
  }
  return rem;
}
NBLS_SS_HD uint32_t ss_bswap32(uint32_t x) { return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24); }
// 32-byte big-endian of l0 + l1 2^64 + l2 2^128
NBLS_SS_HD void ss_store_be(uint8_t* out, uint64_t l0, uint64_t l1, uint64_t l2) {
  uint32_t* o = (uint32_t*)out;
  o[0] = 0; o[1] = 0; o[2] = ss_bswap32((uint32_t)(l2 >> 32)); o[3] = ss_bswap32((uint32_t)l2);
  o[4] = ss_bswap32((uint32_t)(l1 >> 32)); o[5] = ss_bswap32((uint32_t)l1); o[6] = ss_bswap32((uint32_t)(l0 >> 32)); o[7] = ss_bswap32((uint32_t)l0);
}
NBLS_SS_HD void ss_load_be(const uint8_t* k32, uint64_t* l) {
  const uint32_t* k = (const uint32_t*)k32;
  for (int j = 0; j < 4; j++) l[j] = ((uint64_t)ss_bswap32(k[6 - 2 * j]) << 32) | ss_bswap32(k[7 - 2 * j]);
}
// k = a0 + a1 |z| + a2 |z|^2 + a3 |z|^3 (a0..a2 < 2^64, a3 < 2^65 for k < 2^256).  dims = 4 (G2, [|z|^i]Q = (-1)^i psi^i(Q)): the four digits; dims = 2 (G1, [z^2]P = -phi(P)):
// k mod z^2 = a0 + a1 |z| and k div z^2.  Output: dims scalars of 32 bytes big-endian each (the format msm_keys_kernel and the ladders read).
NBLS_SS_HD void scalar_decompose(const uint8_t* k32, unsigned dims, uint8_t* o) {
  uint64_t l[4];
  ss_load_be(k32, l);
  const uint64_t a0 = ss_div_step(l, 4);
  const uint64_t a1 = ss_div_step(l, 4);          // l = k div z^2 (< 2^129)
  if (dims == 2) {
    const uint64_t Z = 0xd201000000010000ull;
    const unsigned __int128 s = (unsigned __int128)a1 * Z + a0;
    ss_store_be(o, (uint64_t)s, (uint64_t)(s >> 64), 0);
    ss_store_be(o + 32, l[0], l[1], l[2]);
  } else {
    const uint64_t a2 = ss_div_step(l, 3);          // l = a3 (< 2^65)
    ss_store_be(o, a0, 0, 0); ss_store_be(o + 32, a1, 0, 0); ss_store_be(o + 64, a2, 0, 0); ss_store_be(o + 96, l[0], l[1], 0);
  }
}
// The four digits recoded SIGN-ALIGNED for the one-addition-per-bit ladder of sign (codec.h pt_mul_sac_g2; Faz-Hernandez, Longa, Sanchez 2013): with a0 made odd (a0 + 1 when even:
// the ladder subtracts Q again), a0 = sum_i s_i 2^i over 66 digits s_i = +-1 with s_i = 2 bit_(i+1)(a0) - 1 and s_65 = +1; every other digit is rewritten over the same signs,
// a_j = sum_i s_i e_ji 2^i with e_ji = a_j mod 2 and a_j <- (a_j >> 1) + (e_ji and s_i = -1).  Output, 4 x 32 bytes big-endian: [bits 0..65: s_i = +1, bit 66: a0 was even],
// then the bits e_1i, e_2i, e_3i.
NBLS_SS_HD void scalar_sac_recode(const uint8_t* k32, uint8_t* o) {
  uint64_t l[4];
  ss_load_be(k32, l);
  uint64_t alo[4], ahi[4];
  alo[0] = ss_div_step(l, 4); alo[1] = ss_div_step(l, 4); alo[2] = ss_div_step(l, 3); alo[3] = l[0];
  ahi[0] = ahi[1] = ahi[2] = 0; ahi[3] = l[1];
  const uint64_t even = (alo[0] & 1) ^ 1;
  alo[0] |= 1;
  const uint64_t slo = alo[0] >> 1, shi = 2;          // bit i set <=> s_i = +1: bits 63, 64 clear (a0 < 2^64), bit 65 set
  ss_store_be(o, slo, shi | (even << 2), 0);
  for (int j = 1; j < 4; j++) {
    uint64_t lo = alo[j], hi = ahi[j], elo = 0, ehi = 0;
    for (int b = 0; b < 66; b++) {
      const uint64_t e = lo & 1;
      const uint64_t sp = b < 64 ? (slo >> b) & 1 : (shi >> (b - 64)) & 1;
      if (b < 64) elo |= e << b; else ehi |= e << (b - 64);
      lo = (lo >> 1) | (hi << 63); hi >>= 1;
      const uint64_t inc = e & (sp ^ 1);
      lo += inc; hi += (uint64_t)(lo < inc);
    }
    ss_store_be(o + 32 * j, elo, ehi, 0);
  }
}

}  // namespace nbls
