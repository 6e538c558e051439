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
// TWO LANES PER ELEMENT: lane parity r owns component c_r of every Fp2 value and computes component r of every product (both
// components cost the same: two limb products + one reduction for a multiplication, one + one for a squaring); the partner's
// component arrives by a DPP lane swap.  Half the instructions per lane and twice the wavefronts of the one-lane-per-element
// form, which at 131,072 elements filled the chip only two wavefronts deep and was latency-bound (5.4 ms).
__device__ __forceinline__ void swap_pair(u32* p, const u32* x) {
#pragma unroll
  for (int k = 0; k < NL; k++) p[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)x[k], 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
}
// component r of a * b; X, U: own components of a, b
__device__ __forceinline__ void fp2_mul_2l(u32* res, const u32* X, const u32* U, bool r) {
  const u32 BIAS[NL] = NBLS_BIAS16_28;
  u32 Y[NL], V[NL], B1[NL], B2[NL], A2[NL];
  swap_pair(Y, X); swap_pair(V, U);
#pragma unroll
  for (int k = 0; k < NL; k++) { B1[k] = r ? V[k] : U[k]; B2[k] = r ? U[k] : V[k]; A2[k] = r ? Y[k] : BIAS[k] - Y[k]; }   // r = 0: a0 b0 - a1 b1 ; r = 1: a1 b0 + a0 b1
  carry_norm(A2);
  u64 acc[2 * NL];
#pragma unroll
  for (int k = 0; k < 2 * NL; k++) acc[k] = 0;
  mac28(acc, X, B1); mac28(acc, A2, B2);
  redc28(res, acc);
}
// component r of a^2 (math.ts:477-484): r = 0: (a0 + a1)(a0 - a1) ; r = 1: (2 a0) a1
__device__ __forceinline__ void fp2_sqr_2l(u32* res, const u32* X, bool r) {
  const u32 BIAS[NL] = NBLS_BIAS16_28;
  u32 Y[NL], o1[NL], o2[NL];
  swap_pair(Y, X);
#pragma unroll
  for (int k = 0; k < NL; k++) { o1[k] = (r ? Y[k] : X[k]) + Y[k]; o2[k] = r ? X[k] : X[k] + BIAS[k] - Y[k]; }
  carry_norm(o1); carry_norm(o2);
  mont_mul28(res, o1, o2);
}
// a^e in Fp2 for the two exponents of the square roots, e = (p^2 + 7) / 16 (decompression, math.ts:547-561) and (p^2 - 9) / 16 (SWU, math.ts:1196-1198).  With
// p = 16 K + 11:   (p^2 + 7) / 16 = K p + 11 K + 8   and   (p^2 - 9) / 16 = K p + 11 K + 7,   and a^p = conj(a), so
//        a^e = (conj(a) a^11)^K  a^tail ,   tail = 8 or 7:
// ONE 377-bit exponent on the base b = conj(a) a^11 (4-bit fixed windows: 376 squarings, ~88 multiplications, a 14-multiplication table) and 9 more
// multiplications / squarings for a^2 .. a^11 -- 377 squarings + ~112 multiplications where the joint 2-bit double exponentiation a^c0 conj(a)^c1 that this
// replaces took 382 + ~193 (same field element: -17 % instructions).  digits = the nibbles of K, most significant first.
// TWO LANES PER ELEMENT (see above).
extern "C" __global__ void __launch_bounds__(64) nbls_fp2_pow_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out, const unsigned char* __restrict__ digits, int nwin, u32* __restrict__ scratch, int tail) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 1;
  const bool r = t & 1;
  if (i >= n) return;                              // both lanes of a pair leave together
  const u32 BIAS[NL] = NBLS_BIAS16_28;
  u32* tab = scratch + (size_t)t * 16 * 16;        // tab[d] = own component of b^d for d = 1 .. 15; tab[0] = a^tail
  u32 acc[NL], tt[NL];
  {
    u32 a1[NL], a2[NL], a3[NL], a4[NL], a8[NL], x[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) a1[k] = in[32 * i + 16 * r + k];
    fp2_sqr_2l(a2, a1, r); fp2_mul_2l(a3, a2, a1, r); fp2_sqr_2l(a4, a2, r); fp2_sqr_2l(a8, a4, r);
    if (tail == 7) fp2_mul_2l(x, a4, a3, r);       // a^7
    else {
#pragma unroll
      for (int k = 0; k < NL; k++) x[k] = a8[k];   // a^8
    }
#pragma unroll
    for (int k = 0; k < NL; k++) tab[k] = x[k];
    fp2_mul_2l(x, a8, a3, r);                      // a^11
    u32 cj[NL];                                    // own component of conj(a): (c0, 16p - c1)
#pragma unroll
    for (int k = 0; k < NL; k++) cj[k] = r ? BIAS[k] - a1[k] : a1[k];
    carry_norm(cj);
    fp2_mul_2l(acc, cj, x, r);                     // b = conj(a) a^11
#pragma unroll
    for (int k = 0; k < NL; k++) tab[16 + k] = acc[k];
  }
  for (int d = 2; d < 16; d++) {                   // b^d = b^(d-1) b
    u32 prev[NL], b1[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) { prev[k] = tab[16 * (d - 1) + k]; b1[k] = tab[16 + k]; }
    fp2_mul_2l(tt, prev, b1, r);
#pragma unroll
    for (int k = 0; k < NL; k++) tab[16 * d + k] = tt[k];
  }
#pragma unroll
  for (int k = 0; k < NL; k++) acc[k] = r ? 0u : NBLS_R1[k];
  for (int w = 0; w < nwin; w++) {
    if (w) { fp2_sqr_2l(tt, acc, r); fp2_sqr_2l(acc, tt, r); fp2_sqr_2l(tt, acc, r); fp2_sqr_2l(acc, tt, r); }
    const unsigned d = digits[w];                   // uniform
    if (d) {
      u32 e[NL];
#pragma unroll
      for (int k = 0; k < NL; k++) e[k] = tab[16 * d + k];
      fp2_mul_2l(tt, acc, e, r);
#pragma unroll
      for (int k = 0; k < NL; k++) acc[k] = tt[k];
    }
  }
  {
    u32 e[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) e[k] = tab[k];
    fp2_mul_2l(tt, acc, e, r);                     // times a^tail
  }
#pragma unroll
  for (int k = 0; k < NL; k++) out[32 * i + 16 * r + k] = tt[k];
  out[32 * i + 16 * r + 14] = 0; out[32 * i + 16 * r + 15] = 0;
}

}  // namespace nbls

extern "C" int nbls_fp_pow_launch(unsigned n, const void* in, void* out, const void* nibbles, int nnib, void* scratch, int is_fp2, void* stream) {   // is_fp2: 0 Fp, 7 / 8: Fp2 with that tail (nbls_fp2_pow_kernel)
  if (n == 0) return 0;
  if (is_fp2) hipLaunchKernelGGL(nbls::nbls_fp2_pow_kernel, dim3((2 * n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out, (const unsigned char*)nibbles, nnib, (nbls::u32*)scratch, is_fp2);
  else hipLaunchKernelGGL(nbls::nbls_fp_pow_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out, (const unsigned char*)nibbles, nnib, (nbls::u32*)scratch);
  return (int)hipGetLastError();
}

// list[k] = index of the k-th item whose int8 flag is set, *count = their number (order irrelevant): the items of a compressed cyclotomic exponentiation that
// met a zero denominator and are recomputed by the plain program over this index list (nbls_api.cpp expx)
namespace nbls {
__global__ void nbls_flag_compact_kernel(unsigned n, const signed char* __restrict__ flags, u32* __restrict__ list, u32* __restrict__ count) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) list[atomicAdd(count, 1u)] = i;
}
}
extern "C" int nbls_flag_compact_launch(unsigned n, const void* flags, void* list, void* count, void* stream) {
  if (n == 0) return 0;
  if (hipMemsetAsync(count, 0, 4, (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
  hipLaunchKernelGGL(nbls::nbls_flag_compact_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, (const signed char*)flags, (nbls::u32*)list, (nbls::u32*)count);
  return (int)hipGetLastError();
}

extern "C" int nbls_fp_inv_launch(unsigned n, const void* in, void* out, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(nbls::nbls_fp_inv_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out);
  return (int)hipGetLastError();
}
