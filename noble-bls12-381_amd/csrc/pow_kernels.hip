// pow_kernels.hip -- serial per-element routines with one work item per LANE (the wave VM would leave 63 lanes idle on
// a single dependent chain): the Fp inversion of the final exponentiation (see fp_inv.h).
#include <hip/hip_runtime.h>
#include "vm_exec.h"
#include "consts_gen.h"
#include "fp_inv.h"

namespace nbls {

// Fixed-exponent powers, one element per lane, 4-bit fixed windows (exponent given as nibbles, most significant first).
// Fp:  Fp.pow / Fp.sqrt's a^((p+1)/4) (math.ts:251-264).   Fp2: Fp2.pow for Fp2.sqrt and sqrt_div_fp2 (math.ts:463-465, 493, 1200).
// Elements are raw scratch elements (16 words: 14 limbs + padding), Montgomery form; every product contracts the value
// to below ~1.1 p, sums stay far below the 16p subtraction bias.
struct Fp2r { u32 c0[NL], c1[NL]; };
__device__ __forceinline__ void mm(u32* r, const u32* a, const u32* b) { mont_mul28(r, a, b); }
__device__ __forceinline__ void add28(u32* r, const u32* a, const u32* b) {
#pragma unroll
  for (int i = 0; i < NL; i++) r[i] = a[i] + b[i];
  carry_norm(r);
}
__device__ __forceinline__ void sub28(u32* r, const u32* a, const u32* b) {   // a - b + 16p
  const u32 BIAS[NL] = NBLS_BIAS16_28;
#pragma unroll
  for (int i = 0; i < NL; i++) r[i] = a[i] + BIAS[i] - b[i];
  carry_norm(r);
}
__device__ __forceinline__ void fp2_mul_r(Fp2r& r, const Fp2r& a, const Fp2r& b) {   // schoolbook with one reduction per component
  const u32 BIAS[NL] = NBLS_BIAS16_28;
  u32 nb[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) nb[i] = BIAS[i] - a.c1[i];
  carry_norm(nb);
  u64 acc[2 * NL];
#pragma unroll
  for (int i = 0; i < 2 * NL; i++) acc[i] = 0;
  mac28(acc, a.c0, b.c0); mac28(acc, nb, b.c1);          // a0 b0 - a1 b1
  u32 r0[NL]; redc28(r0, acc);
#pragma unroll
  for (int i = 0; i < 2 * NL; i++) acc[i] = 0;
  mac28(acc, a.c0, b.c1); mac28(acc, a.c1, b.c0);        // a0 b1 + a1 b0
  redc28(r.c1, acc);
#pragma unroll
  for (int i = 0; i < NL; i++) r.c0[i] = r0[i];
}
__device__ __forceinline__ void fp2_sqr_r(Fp2r& r, const Fp2r& a) {                  // math.ts:477-484
  u32 s[NL], d[NL], e[NL], r0[NL];
  add28(s, a.c0, a.c1); sub28(d, a.c0, a.c1); add28(e, a.c0, a.c0);
  mm(r0, s, d); mm(r.c1, e, a.c1);
#pragma unroll
  for (int i = 0; i < NL; i++) r.c0[i] = r0[i];
}

extern "C" __global__ void __launch_bounds__(64) nbls_fp_inv_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 x[NL], y[NL];
#pragma unroll
  for (int k = 0; k < NL; k++) x[k] = in[SLOT_WORDS * i + k];
  fp_mont_inverse(y, x);
#pragma unroll
  for (int k = 0; k < NL; k++) out[SLOT_WORDS * i + k] = y[k];
  out[SLOT_WORDS * i + 14] = 0; out[SLOT_WORDS * i + 15] = 0;
}

extern "C" __global__ void __launch_bounds__(64) nbls_fp_pow_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out, const unsigned char* __restrict__ nib, int nnib, u32* __restrict__ scratch) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32* tab = scratch + (size_t)i * 16 * 16;   // table[j] = x^j, j = 0..15, in global scratch
  u32 x[NL], acc[NL], t[NL];
#pragma unroll
  for (int k = 0; k < NL; k++) { x[k] = in[16 * i + k]; acc[k] = NBLS_R1[k]; tab[k] = NBLS_R1[k]; tab[16 + k] = x[k]; t[k] = x[k]; }
  for (int j = 2; j < 16; j++) {
    u32 u[NL]; mm(u, t, x);
#pragma unroll
    for (int k = 0; k < NL; k++) { t[k] = u[k]; tab[16 * j + k] = u[k]; }
  }
  for (int w = 0; w < nnib; w++) {
    if (w) { mm(t, acc, acc); mm(acc, t, t); mm(t, acc, acc); mm(acc, t, t); }
    unsigned d = nib[w];   // uniform
    if (d) {
      u32 e[NL];
#pragma unroll
      for (int k = 0; k < NL; k++) e[k] = tab[16 * d + k];
      mm(t, acc, e);
#pragma unroll
      for (int k = 0; k < NL; k++) acc[k] = t[k];
    }
  }
#pragma unroll
  for (int k = 0; k < NL; k++) out[16 * i + k] = acc[k];
  out[16 * i + 14] = 0; out[16 * i + 15] = 0;
}

// a^e in Fp2 for e up to ~2^762, as a^c0 * conj(a)^c1 with e = c0 + c1 p (a^p = conj(a): the Frobenius is free), joint 2-bit windows:
// 2 squarings + at most one multiplication by a^i conj(a)^j per window, 191 windows -- 382 squarings instead of the 757 of a plain
// left-to-right exponentiation (same field element).  digits[w] = c1 bits << 2 | c0 bits, most significant window first.
extern "C" __global__ void __launch_bounds__(64) nbls_fp2_pow_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out, const unsigned char* __restrict__ digits, int nwin, u32* __restrict__ scratch) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 BIAS[NL] = NBLS_BIAS16_28;
  u32* tab = scratch + (size_t)i * 16 * 32;     // tab[j << 2 | i] = a^i conj(a)^j
  Fp2r pw[4], acc, t;
#pragma unroll
  for (int k = 0; k < NL; k++) { pw[1].c0[k] = in[32 * i + k]; pw[1].c1[k] = in[32 * i + 16 + k]; pw[0].c0[k] = NBLS_R1[k]; pw[0].c1[k] = 0; }
  fp2_sqr_r(pw[2], pw[1]); fp2_mul_r(pw[3], pw[2], pw[1]);
  for (int jj = 0; jj < 4; jj++) {
    Fp2r cj = pw[jj];                            // conj(a^j) = conj(a)^j: (c0, 16p - c1), normalised
    if (jj) {
#pragma unroll
      for (int k = 0; k < NL; k++) cj.c1[k] = BIAS[k] - pw[jj].c1[k];
      carry_norm(cj.c1);
    }
    for (int ii = 0; ii < 4; ii++) {
      Fp2r u;
      if (ii == 0) u = cj; else if (jj == 0) u = pw[ii]; else fp2_mul_r(u, pw[ii], cj);
      const int d = (jj << 2) | ii;
#pragma unroll
      for (int k = 0; k < NL; k++) { tab[32 * d + k] = u.c0[k]; tab[32 * d + 16 + k] = u.c1[k]; }
    }
  }
  acc = pw[0];
  for (int w = 0; w < nwin; w++) {
    if (w) { fp2_sqr_r(t, acc); fp2_sqr_r(acc, t); }
    const unsigned d = digits[w];
    if (d) {
      Fp2r e;
#pragma unroll
      for (int k = 0; k < NL; k++) { e.c0[k] = tab[32 * d + k]; e.c1[k] = tab[32 * d + 16 + k]; }
      fp2_mul_r(t, acc, e); acc = t;
    }
  }
#pragma unroll
  for (int k = 0; k < NL; k++) { out[32 * i + k] = acc.c0[k]; out[32 * i + 16 + k] = acc.c1[k]; }
  out[32 * i + 14] = out[32 * i + 15] = out[32 * i + 30] = out[32 * i + 31] = 0;
}

}  // namespace nbls

extern "C" int nbls_fp_pow_launch(unsigned n, const void* in, void* out, const void* nibbles, int nnib, void* scratch, int is_fp2, void* stream) {
  if (n == 0) return 0;
  if (is_fp2) hipLaunchKernelGGL(nbls::nbls_fp2_pow_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out, (const unsigned char*)nibbles, nnib, (nbls::u32*)scratch);
  else hipLaunchKernelGGL(nbls::nbls_fp_pow_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out, (const unsigned char*)nibbles, nnib, (nbls::u32*)scratch);
  return (int)hipGetLastError();
}

extern "C" int nbls_fp_inv_launch(unsigned n, const void* in, void* out, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(nbls::nbls_fp_inv_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out);
  return (int)hipGetLastError();
}
