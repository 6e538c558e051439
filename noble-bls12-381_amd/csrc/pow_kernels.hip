// pow_kernels.hip -- serial per-element routines with one work item per LANE (the wave VM would leave 63 lanes idle on
// a single dependent chain): the Fp inversion of the final exponentiation (see fp_inv.h).
#include <hip/hip_runtime.h>
#include "vm_exec.h"
#include "consts_gen.h"
#include "fp_inv.h"
#include "pow_exec.h"
#include "pow_wide.h"

namespace nbls {

// Fixed-exponent powers (pow_exec.h): one element per lane (Fp) or per pair of lanes (Fp2), sliding windows over a table of odd powers kept in global scratch.
// Fp:  Fp.pow / Fp.sqrt's a^((p+1)/4) (math.ts:251-264).   Fp2: Fp2.pow for Fp2.sqrt and sqrt_div_fp2 (math.ts:463-465, 493, 1200).
// Elements are raw scratch elements (16 words: 14 limbs + padding), Montgomery form.
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

// policy of pow_exec.h's chains for one Fp element per lane
struct FpLane {
  typedef struct { u32 v[NL]; } V;
  const u32* in; u32* out; u32* tab;
  __device__ __forceinline__ void sqr(V& r, const V& a) { mont_sqr28(r.v, a.v); }
  __device__ __forceinline__ void mul(V& r, const V& a, const V& b) { mont_mul28(r.v, a.v, b.v); }
  __device__ __forceinline__ void copy(V& r, const V& a) {
#pragma unroll
    for (int k = 0; k < NL; k++) r.v[k] = a.v[k];
  }
  __device__ __forceinline__ void load(V& r) {
#pragma unroll
    for (int k = 0; k < NL; k++) r.v[k] = in[k];
  }
  __device__ __forceinline__ void store(const V& a) {
#pragma unroll
    for (int k = 0; k < NL; k++) out[k] = a.v[k];
    out[14] = 0; out[15] = 0;
  }
  __device__ __forceinline__ void tab_put(int j, const V& a) {
#pragma unroll
    for (int k = 0; k < NL; k++) tab[16 * j + k] = a.v[k];
  }
  __device__ __forceinline__ void tab_get(V& r, unsigned j) {
#pragma unroll
    for (int k = 0; k < NL; k++) r.v[k] = tab[16 * j + k];
  }
};
extern "C" __global__ void __launch_bounds__(64) nbls_fp_pow_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out, const unsigned char* __restrict__ ops, int nops, u32* __restrict__ scratch) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  FpLane o{in + 16 * (size_t)i, out + 16 * (size_t)i, scratch + (size_t)i * 16 * POW_TAB};
  fp_pow_seq(o, ops, nops);
}

// TWO LANES PER ELEMENT: lane parity r owns component c_r of every Fp2 value and computes component r of every product (both components cost the same: two limb
// products + one reduction for a multiplication, one + one for a squaring); the partner's component arrives by a DPP lane swap.  Half the instructions per lane and
// twice the wavefronts of the one-lane-per-element form, which at 131,072 elements filled the chip only two wavefronts deep and was latency-bound (5.4 ms).
__device__ __forceinline__ void swap_pair(u32* p, const u32* x) {
#pragma unroll
  for (int k = 0; k < NL; k++) p[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)x[k], 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
}
struct Fp2Lane {
  typedef struct { u32 v[NL]; } V;
  const u32* in; u32* out; u32* tab; bool r;
  __device__ __forceinline__ void sqr(V& res, const V& a) { u32 Y[NL]; swap_pair(Y, a.v); fp2_sqr_c(res.v, a.v, Y, r); }
  __device__ __forceinline__ void mul(V& res, const V& a, const V& b) { u32 Y[NL], W[NL]; swap_pair(Y, a.v); swap_pair(W, b.v); fp2_mul_c(res.v, a.v, Y, b.v, W, r); }
  __device__ __forceinline__ void conj(V& res, const V& a) {   // own component of conj(a): (c0, 16p - c1)
    const u32 BIAS[NL] = NBLS_BIAS16_28;
#pragma unroll
    for (int k = 0; k < NL; k++) res.v[k] = r ? BIAS[k] - a.v[k] : a.v[k];
    carry_norm(res.v);
  }
  __device__ __forceinline__ void copy(V& res, const V& a) {
#pragma unroll
    for (int k = 0; k < NL; k++) res.v[k] = a.v[k];
  }
  __device__ __forceinline__ void load(V& res) {   // times one: any stored representative -> below 2p
    u32 x[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) x[k] = in[k];
    mont_mul28(res.v, x, NBLS_R1);
  }
  __device__ __forceinline__ void store(const V& a) {
#pragma unroll
    for (int k = 0; k < NL; k++) out[k] = a.v[k];
    out[14] = 0; out[15] = 0;
  }
  __device__ __forceinline__ void tab_put(int j, const V& a) {
#pragma unroll
    for (int k = 0; k < NL; k++) tab[16 * j + k] = a.v[k];
  }
  __device__ __forceinline__ void tab_get(V& res, unsigned j) {
#pragma unroll
    for (int k = 0; k < NL; k++) res.v[k] = tab[16 * j + k];
  }
};
extern "C" __global__ void __launch_bounds__(64) nbls_fp2_pow_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out, const unsigned char* __restrict__ ops, int nops, u32* __restrict__ scratch, int tail) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 1;
  const bool r = t & 1;
  if (i >= n) return;                              // both lanes of a pair leave together
  Fp2Lane o{in + 32 * (size_t)i + 16 * r, out + 32 * (size_t)i + 16 * r, scratch + (size_t)t * 16 * POW_TAB, r};
  fp2_pow_seq(o, ops, nops, tail);
}

// ---- one limb per lane (pow_wide.h): one wavefront per element, rows 0 / 1 of the wavefront = the components of an Fp2 value (an Fp value: row 0)
struct WideDev {
  typedef u32 U; typedef u64 W;
  u32 lane; const u32* in; u32* out; u32* tab; bool live;      // in / out: the lane's row of the element (16 words per component); tab: the wavefront's LDS table, [entry][64 lanes]
  __device__ __forceinline__ bool row() const { return (lane & 16u) != 0; }
  __device__ __forceinline__ U konst(const u32* t16) const { return t16[lane & 15u]; }
  __device__ __forceinline__ U sel(U a, U b) const { return row() ? b : a; }
  __device__ __forceinline__ U add(U a, U b) const { return a + b; }
  __device__ __forceinline__ U sub(U a, U b) const { return a - b; }
  __device__ __forceinline__ U and_(U a, u32 m) const { return a & m; }
  __device__ __forceinline__ U shr(U a, int k) const { return a >> k; }
  __device__ __forceinline__ U mul_lo(U a, u32 k) const { return a * k; }
  __device__ __forceinline__ U lo(W w) const { return (u32)w; }
  __device__ __forceinline__ W zero() const { return 0; }
  __device__ __forceinline__ W mad(U a, U b, W acc) const { return acc + (u64)a * b; }
  __device__ __forceinline__ W mad_s(U a, u32 s, W acc) const { return acc + (u64)a * s; }
  __device__ __forceinline__ W shr28(W w) const { return w >> 28; }
  __device__ __forceinline__ W add_lo(W w, U x) const { return (w & 0xffffffff00000000ull) | (u32)((u32)w + x); }
  __device__ __forceinline__ U bcast(U v, int i) const {      // ds_swizzle, bit-mask mode: source lane = (lane & 0x10) | i inside every group of 32 lanes; the pattern is an immediate
    switch (i) {
#define NBLS_SWZ(I) case I: return (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x10 | (I << 5));
      NBLS_SWZ(0) NBLS_SWZ(1) NBLS_SWZ(2) NBLS_SWZ(3) NBLS_SWZ(4) NBLS_SWZ(5) NBLS_SWZ(6) NBLS_SWZ(7) NBLS_SWZ(8) NBLS_SWZ(9) NBLS_SWZ(10) NBLS_SWZ(11) NBLS_SWZ(12) default: NBLS_SWZ(13)
#undef NBLS_SWZ
    }
  }
  __device__ __forceinline__ U xchg(U v) const { return (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x1f | (0x10 << 10)); }      // lane ^ 16
  __device__ __forceinline__ U shl1(U v) const { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }      // row_shl:1
  __device__ __forceinline__ U shr1(U v) const { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }      // row_shr:1
  __device__ __forceinline__ u32 lane_of(U v, int k) const { return (u32)__builtin_amdgcn_readlane((int)v, k); }
  __device__ __forceinline__ U load() const { return (live && (lane & 15u) < (u32)NL) ? in[lane & 15u] : 0u; }
  __device__ __forceinline__ void store(U v) const { if (live) out[lane & 15u] = (lane & 15u) < (u32)NL ? v : 0u; }
  __device__ __forceinline__ void tab_put(int e, U v) const { tab[64 * e + lane] = v; }      // every lane its own word (the upper half of the wavefront idles, but it executes the stores)
  __device__ __forceinline__ U tab_get(int e) const { return tab[64 * e + lane]; }
};
// tail: 0 = Fp (op list of the exponent itself), 7 / 8 = Fp2 with that tail (fp2_pow_seq)
extern "C" __global__ void __launch_bounds__(64) nbls_pow_wide_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out, const unsigned char* __restrict__ ops, int nops, int tail, WideConsts consts) {
  __shared__ u32 tab[POW_TAB * 64];
  const unsigned i = blockIdx.x, lane = threadIdx.x;
  const bool fp2 = tail != 0;
  const bool live = i < n && lane < (fp2 ? 32u : 16u);
  const size_t off = (size_t)i * (fp2 ? 32u : 16u) + (fp2 ? 16u * ((lane >> 4) & 1u) : 0u);
  WideDev l{lane, in + off, out + off, tab, live};
  if (fp2) { WideField<WideDev, true> f(l, consts); fp2_pow_seq(f, ops, nops, tail); }
  else { WideField<WideDev, false> f(l, consts); fp_pow_seq(f, ops, nops); }
}

}  // namespace nbls

extern "C" int nbls_pow_wide_launch(unsigned n, const void* in, void* out, const void* ops, int nops, int is_fp2, void* stream) {
  if (n == 0) return 0;
  static const nbls::WideConsts consts = nbls::wide_consts();
  hipLaunchKernelGGL(nbls::nbls_pow_wide_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out, (const unsigned char*)ops, nops, is_fp2, consts);
  return (int)hipGetLastError();
}
extern "C" int nbls_fp_pow_launch(unsigned n, const void* in, void* out, const void* ops, int nops, void* scratch, int is_fp2, void* stream) {   // ops: pow_exec.h op list (device memory); is_fp2: 0 Fp, 7 / 8: Fp2 with that tail (nbls_fp2_pow_kernel); scratch: POW_TAB raw elements per lane
  if (n == 0) return 0;
  if (is_fp2) hipLaunchKernelGGL(nbls::nbls_fp2_pow_kernel, dim3((2 * n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out, (const unsigned char*)ops, nops, (nbls::u32*)scratch, is_fp2);
  else hipLaunchKernelGGL(nbls::nbls_fp_pow_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out, (const unsigned char*)ops, nops, (nbls::u32*)scratch);
  return (int)hipGetLastError();
}

// list[k] = index of the k-th item whose int8 flag is set, *count = their number (order irrelevant): the items of a compressed cyclotomic exponentiation that
// met a zero denominator and are recomputed by the plain program over this index list (pipelines_pairing.cpp expx)
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
