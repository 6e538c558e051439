// pow_kernels.hip -- serial per-element routines with one work item per LANE (the wave VM would leave 63 lanes idle on
// a single dependent chain): the Fp inversion of the final exponentiation (see fp_inv.h).
#include <hip/hip_runtime.h>
#include "vm_exec.h"
#include "consts_gen.h"
#include "fp_inv.h"

namespace nbls {

// out[i] = in[i]^-1 on raw Montgomery limbs (12 words per element), values in [0,2p) in and out.
extern "C" __global__ void __launch_bounds__(64) nbls_fp_inv_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out, const u32* __restrict__ table) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 x[12], y[12];
#pragma unroll
  for (int k = 0; k < 12; k++) x[k] = in[12 * i + k];
  fp_mont_inverse(y, x, table);
#pragma unroll
  for (int k = 0; k < 12; k++) out[12 * i + k] = y[k];
}

}  // namespace nbls

namespace nbls {

// Fixed-exponent powers, one element per lane, 4-bit fixed windows (exponent given as nibbles, most significant first).
// Fp:  Fp.pow / Fp.sqrt's a^((p+1)/4) (math.ts:251-264).   Fp2: Fp2.pow for Fp2.sqrt and sqrt_div_fp2 (math.ts:463-465, 493, 1200).
// Values are raw Montgomery limbs in [0,2p) in and out.
struct Fp2r { u32 c0[12], c1[12]; };
__device__ __forceinline__ void mm(u32* r, const u32* a, const u32* b) { const u32 P2[12] = NBLS_2P32; mont_mul12(r, a, b); csub<12>(r, P2); }
__device__ __forceinline__ void add2p(u32* r, const u32* a, const u32* b) {   // a + b reduced to [0,2p)
  const u32 P2[12] = NBLS_2P32; u32 c = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) r[i] = addc(a[i], b[i], c, &c);
  csub<12>(r, P2);
}
__device__ __forceinline__ void sub2p(u32* r, const u32* a, const u32* b) {   // a - b + 2p reduced to [0,2p)
  const u32 P2[12] = NBLS_2P32; u32 br = 0, t[12];
#pragma unroll
  for (int i = 0; i < 12; i++) t[i] = subb(a[i], b[i], br, &br);
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) r[i] = addc(t[i], P2[i], c, &c);
  csub<12>(r, P2);
}
__device__ __forceinline__ void fp2_mul_r(Fp2r& r, const Fp2r& a, const Fp2r& b) {   // Karatsuba, math.ts:451-462
  u32 t1[12], t2[12], s1[12], s2[12], m[12];
  mm(t1, a.c0, b.c0); mm(t2, a.c1, b.c1);
  add2p(s1, a.c0, a.c1); add2p(s2, b.c0, b.c1); mm(m, s1, s2);
  sub2p(r.c0, t1, t2);
  sub2p(m, m, t1); sub2p(r.c1, m, t2);
}
__device__ __forceinline__ void fp2_sqr_r(Fp2r& r, const Fp2r& a) {                  // math.ts:477-484
  u32 s[12], d[12], e[12];
  add2p(s, a.c0, a.c1); sub2p(d, a.c0, a.c1); add2p(e, a.c0, a.c0);
  u32 r0[12]; mm(r0, s, d); mm(r.c1, e, a.c1);
#pragma unroll
  for (int i = 0; i < 12; i++) r.c0[i] = r0[i];
}

extern "C" __global__ void __launch_bounds__(64) nbls_fp_pow_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out, const unsigned char* __restrict__ nib, int nnib, u32* __restrict__ scratch) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // table[j] = x^j, j = 0..15, kept in global scratch (16 * 12 words per element, lane-interleaved by element)
  u32* tab = scratch + (size_t)i * 16 * 12;
  u32 x[12], acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) { x[k] = in[12 * i + k]; acc[k] = NBLS_R1[k]; tab[k] = NBLS_R1[k]; tab[12 + k] = x[k]; }
  {
    u32 t[12];
#pragma unroll
    for (int k = 0; k < 12; k++) t[k] = x[k];
    for (int j = 2; j < 16; j++) {
      u32 u[12]; mm(u, t, x);
#pragma unroll
      for (int k = 0; k < 12; k++) { t[k] = u[k]; tab[12 * j + k] = u[k]; }
    }
  }
  for (int w = 0; w < nnib; w++) {
    u32 t[12];
    if (w) { mm(t, acc, acc); mm(acc, t, t); mm(t, acc, acc); mm(acc, t, t); }
    unsigned d = nib[w];   // uniform
    if (d) {
      u32 e[12];
#pragma unroll
      for (int k = 0; k < 12; k++) e[k] = tab[12 * d + k];
      mm(t, acc, e);
#pragma unroll
      for (int k = 0; k < 12; k++) acc[k] = t[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 12; k++) out[12 * i + k] = acc[k];
}

extern "C" __global__ void __launch_bounds__(64) nbls_fp2_pow_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out, const unsigned char* __restrict__ nib, int nnib, u32* __restrict__ scratch) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32* tab = scratch + (size_t)i * 16 * 24;
  Fp2r x, acc, t;
#pragma unroll
  for (int k = 0; k < 12; k++) { x.c0[k] = in[24 * i + k]; x.c1[k] = in[24 * i + 12 + k]; acc.c0[k] = NBLS_R1[k]; acc.c1[k] = 0; }
#pragma unroll
  for (int k = 0; k < 12; k++) { tab[k] = acc.c0[k]; tab[12 + k] = 0; tab[24 + k] = x.c0[k]; tab[36 + k] = x.c1[k]; }
  t = x;
  for (int j = 2; j < 16; j++) {
    Fp2r u; fp2_mul_r(u, t, x); t = u;
#pragma unroll
    for (int k = 0; k < 12; k++) { tab[24 * j + k] = u.c0[k]; tab[24 * j + 12 + k] = u.c1[k]; }
  }
  for (int w = 0; w < nnib; w++) {
    if (w) { fp2_sqr_r(t, acc); fp2_sqr_r(acc, t); fp2_sqr_r(t, acc); fp2_sqr_r(acc, t); }
    unsigned d = nib[w];
    if (d) {
      Fp2r e;
#pragma unroll
      for (int k = 0; k < 12; k++) { e.c0[k] = tab[24 * d + k]; e.c1[k] = tab[24 * d + 12 + k]; }
      fp2_mul_r(t, acc, e); acc = t;
    }
  }
#pragma unroll
  for (int k = 0; k < 12; k++) { out[24 * i + k] = acc.c0[k]; out[24 * i + 12 + k] = acc.c1[k]; }
}

}  // namespace nbls

extern "C" int nbls_fp_pow_launch(unsigned n, const void* in, void* out, const void* nibbles, int nnib, void* scratch, int is_fp2, void* stream) {
  if (n == 0) return 0;
  if (is_fp2) hipLaunchKernelGGL(nbls::nbls_fp2_pow_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out, (const unsigned char*)nibbles, nnib, (nbls::u32*)scratch);
  else hipLaunchKernelGGL(nbls::nbls_fp_pow_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out, (const unsigned char*)nibbles, nnib, (nbls::u32*)scratch);
  return (int)hipGetLastError();
}

extern "C" int nbls_fp_inv_launch(unsigned n, const void* in, void* out, const void* table, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(nbls::nbls_fp_inv_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out, (const nbls::u32*)table);
  return (int)hipGetLastError();
}
