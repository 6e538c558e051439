// vm_exec.h -- per-lane semantics of every step kind, written once and compiled twice:
//   * vm_kernel.hip : the gfx950 kernel (LDS = __shared__ memory, one wavefront per workgroup)
//   * vm_sim.cpp    : a host-side simulator used ONLY by the CPU test-suite to check compiled programs
//                     without a GPU (tests/); it is not part of libnbls.so.
//
// Field representation: 14 limbs of 28 bits (in 32-bit words, little-endian limb order), Montgomery form with
// R = 2^392.  Why 28 bits: on gfx950 v_mad_u64_u32 AND every carry-consuming add (v_addc_co) issue at half rate, and a
// lone wavefront issues one VALU instruction per ~5.5 clocks whatever it is (tools/ubench/carry_rates.hip), so the cost of
// a big-integer product is its instruction COUNT.  With 28-bit limbs a 64-bit column accumulator absorbs 112 limb
// products without overflow, so a limb product is ONE in-place multiply-add (196 per product, no carry handling at all).
// Product operands are SIGNED limb vectors: x - y and -x are plain limb-wise subtractions (limbs in (-2^28, 2^28)), x + y
// has limbs below 2^29, nothing is normalised before the multiplication, and the products go through v_mad_i64_i32 into
// signed column accumulators (budget: sum over the products of c_a * c_b <= 8, c = 2 for an un-normalised sum, else 1;
// the tracer asks for a normalisation pass where a lane-op would exceed it).  The Montgomery reduction works on the
// signed columns with arithmetic shifts; a multiple of p baked into the upper columns (offs * p * R) keeps the reduced
// value non-negative.  Because p < 2^381 there are 11 bits of headroom (values up to 2047 p fit), so results never need
// conditional subtractions except where a canonical representative is required (store to wire format, zero tests,
// comparisons).  Slots always hold non-negative values with normalised limbs.
#pragma once
#include "vm.h"
#include "consts_gen.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NBLS_HD __host__ __device__ __forceinline__
#else
#define NBLS_HD inline
#endif

namespace nbls {
typedef uint32_t u32;
typedef uint64_t u64;

#define NL 14
#define LMASK 0x0fffffffu
// p, -p^-1 mod 2^28
#define NBLS_P28 NBLS_P_INIT
#define NBLS_N0_28 NBLS_N0_LIMB
#define NBLS_BIAS16_28 NBLS_BIAS16_INIT   // 16p with every limb >= 2^28 - 1 (per-lane pow kernels only)
typedef int32_t i32;
typedef int64_t i64;

// x (signed limbs in (-2^31, 2^31), value in [0, 2^392)) -> normalised limbs (< 2^28; the top limb keeps the rest)
NBLS_HD void carry_norm(u32* x) {
  i32 c = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; i++) { i32 v = (i32)x[i] + c; x[i] = (u32)v & LMASK; c = v >> 28; }
  x[NL - 1] = (u32)((i32)x[NL - 1] + c);
}

template <typename LDSP>
NBLS_HD void ld14(u32* x, LDSP lds, u32 off) {
#if defined(NBLS_EXP_NOLDS)     // timing experiment only (tools/exp_variants.sh): operands made up from the address, no LDS traffic; results are garbage
#pragma unroll
  for (int i = 0; i < NL; i++) x[i] = (off * 2654435761u + i * 40503u) & LMASK;
#else
#pragma unroll
  for (int i = 0; i < NL; i++) x[i] = lds[off + i];
#endif
}
// word offset of an operand's slot: constants live at the start of the LDS image, everything else in the instance region
NBLS_HD u32 slot_addr(u32 op, u32 inst) {
  const u32 is_const = (u32)((i32)(op << 18) >> 31);   // OP_CONST (bit 13) -> all ones
  return (inst & ~is_const) + ((op & OP_SLOT_MASK) << 4);
}

// operand of a DOT product, as signed limbs: x, x + y or x - y, optionally normalised (sums only), optionally negated
template <typename LDSP>
NBLS_HD void dot_operand(u32* A, u32 enc, LDSP lds, u32 inst) {
  const u32 e0 = enc & 0xffff, e1 = enc >> 16;
  ld14(A, lds, slot_addr(e0, inst));
  if ((enc & (OP_NEG | OP_NORM | (OP_PRESENT << 16))) == 0) return;   // plain slot: the common case
  if (e1 & OP_PRESENT) {
    u32 X[NL];
    ld14(X, lds, slot_addr(e1, inst));
    if (e1 & OP_NEG) {
#pragma unroll
      for (int i = 0; i < NL; i++) A[i] -= X[i];
    } else {
#pragma unroll
      for (int i = 0; i < NL; i++) A[i] += X[i];
    }
  }
  if (e0 & OP_NORM) carry_norm(A);
  if (e0 & OP_NEG) {
#pragma unroll
    for (int i = 0; i < NL; i++) A[i] = 0u - A[i];
  }
}

// acc[i+j] += a[j] * b[i] on signed limbs: 196 in-place v_mad_i64_i32, no carries
NBLS_HD void mac28(u64* acc, const u32* a, const u32* b) {
#pragma unroll
  for (int i = 0; i < NL; i++) {
#pragma unroll
    for (int j = 0; j < NL; j++) acc[i + j] = (u64)((i64)acc[i + j] + (i64)(i32)a[j] * (i64)(i32)b[i]);
  }
}
// acc = offs * p * R (the bias that keeps a reduction with negative products non-negative)
NBLS_HD void acc_init(u64* acc, u32 offs) {
  const u32 P[NL] = NBLS_P28;
#pragma unroll
  for (int i = 0; i < NL; i++) { acc[i] = 0; acc[NL + i] = (u64)offs * P[i]; }
}
// Montgomery reduction of the signed lazy column accumulators (value V >= 0): r (normalised) in [V / 2^392, V / 2^392 + p)
NBLS_HD void redc28(u32* r, u64* acc) {
  const u32 P[NL] = NBLS_P28;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const u32 m = ((u32)acc[i] * NBLS_N0_28) & LMASK;
#pragma unroll
    for (int j = 0; j < NL; j++) acc[i + j] += (u64)m * P[j];
    acc[i + 1] = (u64)((i64)acc[i + 1] + ((i64)acc[i] >> 28));
  }
  i64 c = 0;
#pragma unroll
  for (int k = 0; k < NL - 1; k++) { i64 v = (i64)acc[NL + k] + c; r[k] = (u32)v & LMASK; c = v >> 28; }
  r[NL - 1] = (u32)((i64)acc[2 * NL - 1] + c);
}
// r = a * b / R, normalised, < a*b/R + p   (used by the per-lane kernels in pow_kernels.hip / fp_inv.h)
NBLS_HD void mont_mul28(u32* r, const u32* a, const u32* b) {
  u64 acc[2 * NL];
#pragma unroll
  for (int i = 0; i < 2 * NL; i++) acc[i] = 0;
  mac28(acc, a, b);
  redc28(r, acc);
}
// x (normalised, < 2p) -> canonical [0, p)
NBLS_HD void csub_p(u32* x) {
  const u32 P[NL] = NBLS_P28;
  u32 d[NL], br = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) { u32 t = x[i] - P[i] - br; br = t >> 31; d[i] = (i < NL - 1) ? (t & LMASK) : t; }
#pragma unroll
  for (int i = 0; i < NL; i++) x[i] = br ? x[i] : d[i];
}
// 28-bit limbs <-> twelve 32-bit words of the same integer (< 2^384)
NBLS_HD void limbs_to_words(u32* w, const u32* x) {
#pragma unroll
  for (int k = 0; k < 12; k++) {
    const int bit = 32 * k, l = bit / 28, off = bit % 28;
    u64 v = x[l];
    if (l + 1 < NL) v |= (u64)x[l + 1] << 28;
    if (l + 2 < NL) v |= (u64)x[l + 2] << 56;
    w[k] = (u32)(v >> off);
  }
}
NBLS_HD void words_to_limbs(u32* x, const u32* w) {
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int bit = 28 * i, k = bit / 32, off = bit % 32;
    u64 v = w[k];
    if (k + 1 < 12) v |= (u64)w[k + 1] << 32;
    x[i] = (u32)(v >> off) & LMASK;
  }
}
NBLS_HD u32 bswap32(u32 x) { return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24); }
NBLS_HD bool is_zero_mod_p(const u32* x) {   // x normalised, < 2p
  const u32 P[NL] = NBLS_P28;
  u32 z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) { z |= x[i]; e |= x[i] ^ P[i]; }
  return z == 0 || e == 0;
}
// (x + (x odd ? p : 0)) / 2 on normalised limbs
NBLS_HD void halve28(u32* r) {
  const u32 P[NL] = NBLS_P28;
  const u32 mask = (r[0] & 1) ? 0xffffffffu : 0u;
#pragma unroll
  for (int i = 0; i < NL; i++) r[i] += P[i] & mask;
  carry_norm(r);
#pragma unroll
  for (int i = 0; i < NL - 1; i++) r[i] = (r[i] >> 1) | ((r[i + 1] & 1) << 27);
  r[NL - 1] >>= 1;
}

// ------------------------------------------------------------------------------------------------
// Per-lane step execution.  `lds` is the workgroup's LDS image (device) or a plain array (simulator):
//   words [0, nconst*16)            program constants (shared by all instances)
//   words [inst, inst + slots*16)   this instance's slots
// The function reads operands, computes, and returns the result in `res` (14 limbs) together with the destination
// word offset (or 0xffffffff when the step has no LDS destination); the caller commits the limbs afterwards, so that
// every read of a step precedes every write of that step (in-order LDS within a wavefront; explicit two-phase loop in
// the simulator).
struct LaneCtx {
  u32 inst;       // word offset of the instance region
  u32 item;       // global work-item index
  bool live;      // item < n_items (dead instances compute on zeros but never touch global memory)
};

// K_DOT in two pieces so that the two-wave kernel (vm_kernel.hip, small batches) can split the products of a lane-op
// between two wavefronts: dot_products accumulates products [lo, hi) of this lane's descriptor into the 28 signed
// columns; dot_result reduces the columns and applies multiplier, post-added slots, normalisation and halving.
template <typename LDSP>
NBLS_HD void dot_products(u64* acc, const Step& st, const u32* d, const u32* __restrict__ gd, LDSP lds, const LaneCtx& cx, u32 lo, u32 hi) {
  const u32 k = (d[0] >> 16) & 0xf;
  u32 na, nb;
  if (lo == 0) { na = d[4]; nb = d[5]; } else if (lo < hi) { na = gd[4 + 2 * lo]; nb = gd[5 + 2 * lo]; } else { na = 0; nb = 0; }
  for (u32 i = lo; i < hi; i++) {   // uniform trip count; next product's operand words are fetched ahead
    const u32 ea = na, eb = nb;
    if (i + 1 < hi) { na = gd[6 + 2 * i]; nb = gd[7 + 2 * i]; }
    if (i < k) {
      u32 A[NL], B[NL];
      dot_operand(A, ea, lds, cx.inst);
      dot_operand(B, eb, lds, cx.inst);
      mac28(acc, A, B);
    }
  }
}
template <typename LDSP>
NBLS_HD u32 dot_result(u32* res, u64* acc, bool have_products, const Step& st, const u32* d, LDSP lds, const LaneCtx& cx) {
  const u32 w0 = d[0];
  const u32 L = (w0 >> 20) & 0xf, mult = (w0 >> 24) & 0x7;
  u32 r[NL];
  if (have_products) redc28(r, acc);
  else {
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = 0;
  }
  if (mult > 1) {
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] *= mult;     // m <= 4; with <= 4 linear terms the limb sums stay inside (-2^31, 2^31)
  }
#pragma unroll
  for (int t = 0; t < MAX_DOT_LINEAR; t++) {
    if (t < (int)st.pad) {   // uniform
      u32 term = (d[2 + t / 2] >> (16 * (t & 1))) & 0xffff;
      if ((u32)t < L) {
        u32 X[NL];
        ld14(X, lds, slot_addr(term, cx.inst));
        if (term & OP_NEG) {
#pragma unroll
          for (int i = 0; i < NL; i++) r[i] -= X[i];
        } else {
#pragma unroll
          for (int i = 0; i < NL; i++) r[i] += X[i];
        }
      }
    }
  }
  if (mult > 1 || st.pad > 0) carry_norm(r);
  if (w0 & (1u << 27)) halve28(r);
#pragma unroll
  for (int i = 0; i < NL; i++) res[i] = r[i];
  return slot_addr(w0 & 0xffff, cx.inst);
}

template <typename LDSP>
NBLS_HD u32 exec_lane(const Step& st, const u32* d /* first 8 descriptor words, already loaded */, const u32* __restrict__ gd /* this lane's descriptor in global memory */,
                      LDSP lds, const LaneCtx& cx, const IOBuf* bufs, u32* res) {
  switch (st.kind) {
    case K_DOT: {
      u64 acc[2 * NL];
      if (st.p0 > 0) {   // uniform
        acc_init(acc, d[0] >> 28);
        dot_products(acc, st, d, gd, lds, cx, 0, st.p0);
      }
      return dot_result(res, acc, st.p0 > 0, st, d, lds, cx);
    }
    case K_LIN: {
      const u32 w0 = d[0];
      const u32 nt = (w0 >> 16) & 0xff;
      u32 r[NL];
#pragma unroll
      for (int i = 0; i < NL; i++) r[i] = 0;
#pragma unroll
      for (int t = 0; t < MAX_LIN_TERMS; t++) {
        if (t < (int)st.p0) {   // uniform
          u32 term = (d[1 + t / 2] >> (16 * (t & 1))) & 0xffff;
          if ((u32)t < nt) {
            u32 X[NL];
            ld14(X, lds, slot_addr(term, cx.inst));
            if (term & OP_NEG) {
#pragma unroll
              for (int i = 0; i < NL; i++) r[i] -= X[i];
            } else {
#pragma unroll
              for (int i = 0; i < NL; i++) r[i] += X[i];
            }
          }
        }
      }
      carry_norm(r);
      if (w0 & (1u << 24)) halve28(r);
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = r[i];
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_LOAD: {
      u32 w0 = d[0], off = d[1];
      const IOBuf& b = bufs[(w0 >> 16) & 7];
      const u32* src = (const u32*)(b.ptr + (u64)cx.item * b.stride + off);
      const int nw = st.p0 ? (int)st.p0 / 4 : 12;   // number of 32-bit words (big-endian integer of p0 bytes)
      u32 w[12];
#pragma unroll
      for (int i = 0; i < 12; i++) w[i] = (cx.live && i < nw) ? bswap32(src[nw - 1 - i]) : 0u;
      words_to_limbs(res, w);
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_LOADW: {
      u32 w0 = d[0], off = d[1];
      const IOBuf& b = bufs[(w0 >> 16) & 7];
      const u32* src = (const u32*)(b.ptr + (u64)cx.item * b.stride + off);
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = cx.live ? src[i] : 0u;
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_STORE: {
      u32 w0 = d[0], off = d[1];
      u32 X[NL], w[12];
      ld14(X, lds, slot_addr(w0 & 0xffff, cx.inst));
      if (st.p0 == 0) csub_p(X);     // p0 = 1: raw 384-bit integer (compressed encodings carry flag bits above bit 380)
      limbs_to_words(w, X);
      if (cx.live) {
        const IOBuf& b = bufs[(w0 >> 16) & 7];
        u32* dst = (u32*)(b.ptr + (u64)cx.item * b.stride + off);
#pragma unroll
        for (int i = 0; i < 12; i++) dst[11 - i] = bswap32(w[i]);
      }
      return 0xffffffffu;
    }
    case K_STOREW: {
      u32 w0 = d[0], off = d[1];
      u32 X[NL];
      ld14(X, lds, slot_addr(w0 & 0xffff, cx.inst));
      if (cx.live) {
        const IOBuf& b = bufs[(w0 >> 16) & 7];
        u32* dst = (u32*)(b.ptr + (u64)cx.item * b.stride + off);
#pragma unroll
        for (int i = 0; i < NL; i++) dst[i] = X[i];
        dst[14] = 0; dst[15] = 0;
      }
      return 0xffffffffu;
    }
    case K_ISZ: {
      u32 w0 = d[0];
      u32 X[NL];
      ld14(X, lds, slot_addr(w0 >> 16, cx.inst));
      bool z = is_zero_mod_p(X);
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = 0;
      res[0] = z ? 1u : 0u;
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_SEL: {
      u32 w0 = d[0], w1 = d[1];
      // both sources are read and merged with a mask: the LDS access pattern does not depend on the flag (the flag
      // can be a secret scalar bit in the sign / getPublicKey ladders)
      const u32 m = 0u - (lds[slot_addr(w0 >> 16, cx.inst)] != 0 ? 1u : 0u);
      u32 Xa[NL], Xb[NL];
      ld14(Xa, lds, slot_addr(w1 & 0xffff, cx.inst));
      ld14(Xb, lds, slot_addr(w1 >> 16, cx.inst));
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = (Xa[i] & m) | (Xb[i] & ~m);
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_CANON: {
      u32 w0 = d[0];
      ld14(res, lds, slot_addr(w0 >> 16, cx.inst));
      csub_p(res);
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_CMP: {
      u32 w0 = d[0], w1 = d[1];
      u32 X[NL], Y[NL];
      ld14(X, lds, slot_addr(w1 & 0xffff, cx.inst));
      u32 f;
      if (st.p0 == 0) {   // X > Y  <=>  Y - X borrows (normalised limbs)
        ld14(Y, lds, slot_addr(w1 >> 16, cx.inst));
        u32 br = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) { u32 t = Y[i] - X[i] - br; br = t >> 31; }
        f = br;
      } else {
        f = X[0] & 1;
      }
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = 0;
      res[0] = f;
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_BIT: {
      u32 w0 = d[0], bit = d[1];
      u32 w = lds[slot_addr(w0 >> 16, cx.inst) + bit / 28];
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = 0;
      res[0] = (w >> (bit % 28)) & 1;
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_BITAND: {
      u32 w0 = d[0], w1 = d[1];
      u32 X[NL], Y[NL];
      ld14(X, lds, slot_addr(w1 & 0xffff, cx.inst));
      ld14(Y, lds, slot_addr(w1 >> 16, cx.inst));
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = X[i] & Y[i];
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_FLAG: {
      u32 w0 = d[0], w1 = d[1];
      u32 a = lds[slot_addr(w1 & 0xffff, cx.inst)] & 1, b = lds[slot_addr(w1 >> 16, cx.inst)] & 1;
      u32 f = st.p0 == 0 ? (a & b) : st.p0 == 1 ? (a | b) : st.p0 == 2 ? (a ^ b) : (a & (b ^ 1));
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = 0;
      res[0] = f;
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_STATUS: {
      u32 w0 = d[0];
      u32 n = w0 & 0xff, code = 0;
      for (int k = (int)n - 1; k >= 0; k--) {
        u32 e = d[1 + k];
        if ((lds[slot_addr(e & 0xffff, cx.inst)] & 1) == 0) code = e >> 16;
      }
      if (cx.live) { const IOBuf& b = bufs[(w0 >> 16) & 7]; ((int8_t*)b.ptr)[(u64)cx.item * b.stride] = (int8_t)code; }
      return 0xffffffffu;
    }
  }
  return 0xffffffffu;
}

}  // namespace nbls
