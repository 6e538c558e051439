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

// LDS access: `lds` is a byte pointer to the workgroup's LDS image (device) or to a plain array (simulator); slots are
// 16-byte aligned, so a slot is read with three 16-byte accesses and one 8-byte access (ds_read_b128 x 3 + ds_read_b64).
struct alignas(16) V4 { u32 x, y, z, w; };
struct alignas(8) V2 { u32 x, y; };
template <typename LDSP>
NBLS_HD void ld14(u32* x, LDSP lds, u32 addr) {
  const V4 a = *(const V4*)(lds + addr), b = *(const V4*)(lds + addr + 16), c = *(const V4*)(lds + addr + 32);
  const V2 d = *(const V2*)(lds + addr + 48);
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
  x[8] = c.x; x[9] = c.y; x[10] = c.z; x[11] = c.w; x[12] = d.x; x[13] = d.y;
}
template <typename LDSP>
NBLS_HD void st14(LDSP lds, u32 addr, const u32* x) {
  V4 a = {x[0], x[1], x[2], x[3]}, b = {x[4], x[5], x[6], x[7]}, c = {x[8], x[9], x[10], x[11]};
  V2 d = {x[12], x[13]};
  *(V4*)(lds + addr) = a; *(V4*)(lds + addr + 16) = b; *(V4*)(lds + addr + 32) = c; *(V2*)(lds + addr + 48) = d;
}
template <typename LDSP>
NBLS_HD u32 ld1(LDSP lds, u32 addr) { return *(const u32*)(lds + addr); }

struct LaneCtx {
  u32 inst;       // byte offset of the instance region (shared constants: minus 2, see term_addr)
  u32 item;       // global work-item index
  bool live;      // item < n_items (dead instances compute on zeros but never touch global memory)
  bool shared;    // uniform: the program keeps ONE copy of its constants at the start of the LDS image; a compile-time constant in the kernels
};
// LDS byte address of an operand field.  Replicated constants (the pairing programs): base + offset, one addition.  Shared constants (programs
// that run 8 or 16 instances per wavefront, where replication would cost occupancy): bit 1 of the offset marks a SLOT (relative to the instance
// region), constants are absolute; with inst = base - 2 that is (inst AND sign-extended bit 1) + offset: v_bfe_i32, v_and, v_add.
NBLS_HD u32 term_addr(u32 f, const LaneCtx& cx) {
  if (cx.shared) return (cx.inst & (u32)((i32)(f << 30) >> 31)) + f;
  return cx.inst + f;
}

// Operand of a DOT product round, as signed limbs: x, x + y, x - y or (mode 3) +-x +- y with the signs per lane (`neg`: bit 0 first term, bit 1
// second term), optionally normalised.  The terms are in registers already (dot_round loads all four of a round first); `shape` (3 bits: mode,
// normalise) is uniform for the wavefront, so these are scalar branches; a lane that has no second term where another lane has one adds the zero
// constant.  The normalisation sits inside every two-term branch (a single slot is normalised already): as a separate stage after the point
// where the modes merge it cost ~15 register copies per operand (the compiler kept the operand in two places, one per successor).
NBLS_HD void dot_combine(u32* A, const u32* X, u32 shape, u32 neg) {
  const u32 mode = shape & 3;
  const bool norm = (shape & 4) != 0;
  if (mode == 1) {
#pragma unroll
    for (int i = 0; i < NL; i++) A[i] += X[i];
    if (norm) carry_norm(A);
  } else if (mode == 2) {
#pragma unroll
    for (int i = 0; i < NL; i++) A[i] -= X[i];
    if (norm) carry_norm(A);
  } else if (mode == 3) {
    const u32 n0 = neg & 1u, n1 = (neg >> 1) & 1u, m0 = 0u - n0, m1 = 0u - n1, c = n0 + n1;
#pragma unroll
    for (int i = 0; i < NL; i++) A[i] = (A[i] ^ m0) + (X[i] ^ m1) + c;
    if (norm) carry_norm(A);
  }
}

// v_mad_i64_i32 / v_mad_u64_u32 are VOP3B instructions: besides the 64-bit result they write a carry-out SGPR pair (SDST), and the compiler hands every
// multiply-add the same dead pair.  Round-3 experiment (tools/ubench/mad_sdst.hip, tools/exp_sdst.sh; profiles/round3_sdst_ab.txt): in a pure stream of
// inline-assembly multiply-adds a rotation over four pairs issues 18-28 % faster than one pair -- but in the kernel, with the product block and the
// reduction written as inline assembly with rotating pairs (NBLS_SDST_PAIRS = 2 / 4 / 8), a lone wavefront is 6-16 % SLOWER (3.27-3.58 ms against
// 3.08 ms per 4096-pairing call) and a saturated launch unchanged: the hazard recogniser pads every inline-assembly group with s_nop, and the
// compiler's own schedule of the block is already better than what the micro-benchmark's single-pair stream suggested.  Kept as a build-time switch
// (default 0 = plain C, the compiler's schedule); the simulator always uses the plain C form.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NBLS_SDST_PAIRS)
#define NBLS_SDST_PAIRS 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && NBLS_SDST_PAIRS > 0
#define NBLS_MAD_ROT 1
#define NBLS_ASM_MADI(ACC, A, B, SD, C0, C1) asm volatile("v_mad_i64_i32 %0, " SD ", %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B) : C0, C1)
#define NBLS_ASM_MADU(ACC, A, B, SD, C0, C1) asm volatile("v_mad_u64_u32 %0, " SD ", %1, %2, %0" : "+v"(ACC) : "v"(A), "s"(B) : C0, C1)
#define NBLS_ROT_CASES(M, ACC, A, B, K) \
  switch ((K) % NBLS_SDST_PAIRS) { \
    case 0: M(ACC, A, B, "s[84:85]", "s84", "s85"); break; \
    case 1: M(ACC, A, B, "s[86:87]", "s86", "s87"); break; \
    case 2: M(ACC, A, B, "s[88:89]", "s88", "s89"); break; \
    case 3: M(ACC, A, B, "s[90:91]", "s90", "s91"); break; \
    case 4: M(ACC, A, B, "s[92:93]", "s92", "s93"); break; \
    case 5: M(ACC, A, B, "s[94:95]", "s94", "s95"); break; \
    case 6: M(ACC, A, B, "s[96:97]", "s96", "s97"); break; \
    default: M(ACC, A, B, "s[98:99]", "s98", "s99"); break; \
  }
__device__ __forceinline__ void madi_rot(u64& acc, u32 a, u32 b, int k) { NBLS_ROT_CASES(NBLS_ASM_MADI, acc, a, b, k) }
__device__ __forceinline__ void madu_rot(u64& acc, u32 m, u32 p, int k) { NBLS_ROT_CASES(NBLS_ASM_MADU, acc, m, p, k) }
#endif
// acc[i+j] += a[j] * b[i] on signed limbs: 196 in-place v_mad_i64_i32, no carries
NBLS_HD void mac28(u64* acc, const u32* a, const u32* b) {
#pragma unroll
  for (int i = 0; i < NL; i++) {
#pragma unroll
    for (int j = 0; j < NL; j++) {
#if defined(NBLS_MAD_ROT)
      madi_rot(acc[i + j], a[j], b[i], i * NL + j);
#else
      acc[i + j] = (u64)((i64)acc[i + j] + (i64)(i32)a[j] * (i64)(i32)b[i]);
#endif
    }
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
    for (int j = 0; j < NL; j++) {
#if defined(NBLS_MAD_ROT)
      madu_rot(acc[i + j], m, P[j], i * NL + j);
#else
      acc[i + j] += (u64)m * P[j];
#endif
    }
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
// Weak reduction of a value V = sum r[k] 2^(28 k) >= 0 given as signed un-normalised limbs (|r[k]| < 2^31 - 2^28, V < 120 p): subtracts q p with
// q = floor((r[13] - 9) * floor(2^48 / 106514) / 2^48) <= V / p  (106514 > p / 2^364; the lower limbs move V / 2^364 by less than 8.01, so q p <= V),
// read from a table of multiples of p.  Afterwards 0 <= V < 2.01 p (1.002 p measured over 2 * 10^5 random values; the tracer books 3.02 p); the caller normalises.  Replaces folding post-added terms into the dot
// product as extra limb products (a 196-multiply-add round) or a contraction lane-op by ~20 instructions.
NBLS_HD void weak_reduce(u32* r, const u32* __restrict__ qp_table) {
  i32 t = (i32)r[NL - 1] - 9;
  t = t < 0 ? 0 : t;
#if defined(__HIP_DEVICE_COMPILE__)
  u32 q = __umulhi((u32)t, 2642610142u) >> 16;     // floor(2^48 / 106514)
#else
  u32 q = (u32)(((u64)(u32)t * 2642610142u) >> 48);
#endif
  q = q > (u32)(QP_TABLE_ENTRIES - 1) ? (u32)(QP_TABLE_ENTRIES - 1) : q;
  const u32* T = qp_table + q * RAW_WORDS;
#pragma unroll
  for (int i = 0; i < NL; i++) r[i] -= T[i];
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
// Per-lane step execution.  `lds` is the workgroup's LDS image (device) or a plain array (simulator), addressed in bytes:
//   [inst, inst + inst_bytes)   this instance's region: the program's constants first, then its slots
// The functions read operands, compute, and return the result in `res` (14 limbs) together with the destination byte
// address (or 0xffffffff when the step has no LDS destination); the caller commits the limbs afterwards, so that
// every read of a step precedes every write of that step (in-order LDS within a wavefront; explicit two-phase loop in
// the simulator).
NBLS_HD u32 slot_addr(u32 field, const LaneCtx& cx) { return term_addr(field & 0xffffu, cx); }
// 16-bit offset number t of a packed list that starts at word `first` of the descriptor
NBLS_HD u32 field16(const u32* d, int first, int t) { return (d[first + t / 2] >> (16 * (t & 1))) & 0xffffu; }

// K_DOT in three pieces (the lane-split kernel sums the columns of four lanes between the second and the third): dot_init loads the bias,
// dot_round accumulates one product of this lane into the 28 signed columns, dot_finish reduces the columns and applies multiplier,
// post-added slots, normalisation and halving.
NBLS_HD void dot_init(u64* acc, const Step& st, u32 w0) {
  if (st.p1 & DOTF_OFFS) acc_init(acc, (w0 >> 20) & 0xf);
  else {
#pragma unroll
    for (int i = 0; i < 2 * NL; i++) acc[i] = 0;
  }
}
NBLS_HD u32 round_shape(const Step& st, u32 r) { return ((r < 4 ? st.shape[0] : st.shape[1]) >> (8 * (r & 3))) & 0xffu; }
// One product round: acc += A * B.  All four terms are requested before the first is used (one LDS round trip per round instead of two: -4 % for
// a lone wavefront); `neg`: the four per-lane sign bits of the round (a0, a1, b0, b1), read in mode 3 only.
template <typename LDSP>
NBLS_HD void dot_round(u64* acc, u32 shape, u32 neg, u32 a0, u32 a1, u32 b0, u32 b1, LDSP lds, const LaneCtx& cx) {
  const u32 sa = shape & 7, sb = (shape >> SH_B_SHIFT) & 7;
  u32 A[NL], B[NL], X[NL], Y[NL];
  ld14(A, lds, term_addr(a0, cx));
  if (sa & 3) ld14(X, lds, term_addr(a1, cx));
  ld14(B, lds, term_addr(b0, cx));
  if (sb & 3) ld14(Y, lds, term_addr(b1, cx));
  dot_combine(A, X, sa, neg & 3);
  dot_combine(B, Y, sb, (neg >> 2) & 3);
  mac28(acc, A, B);
}
// sign bits of round r in word 1 of the lane descriptor
NBLS_HD u32 round_signs(u32 w1, u32 r) { return (w1 >> (4 * r)) & 15u; }
template <typename LDSP>
NBLS_HD u32 dot_finish(u32* res, u64* acc, const Step& st, const u32* d /* the 8 header words */, LDSP lds, const LaneCtx& cx, const u32* __restrict__ qp_table) {
  const u32 w0 = d[0];
  u32 r[NL];
  if (st.p0 > 0) redc28(r, acc);
  else {
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = 0;
  }
  if (st.p1 & DOTF_MULT) {   // m in 1..4 as shift and add; with <= 4 post-added terms the limb sums stay inside (-2^31, 2^31)
    const u32 mult = (w0 >> 16) & 7, sh = mult >> 1, m3 = mult == 3 ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = (r[i] << sh) + (r[i] & m3);
  }
  const int nadd = (int)(st.lin & 7), nsub = (int)((st.lin >> 4) & 7);
#pragma unroll
  for (int t = 0; t < MAX_DOT_LINEAR; t++) {
    if (t < nadd) {   // uniform
      u32 X[NL];
      ld14(X, lds, slot_addr(field16(d, 4, t), cx));
#pragma unroll
      for (int i = 0; i < NL; i++) r[i] += X[i];
    }
  }
#pragma unroll
  for (int t = 0; t < 2 * MAX_DOT_LINEAR; t++) {
    if (t >= nadd && t < nadd + nsub) {   // uniform
      u32 X[NL];
      ld14(X, lds, slot_addr(field16(d, 4, t), cx));
#pragma unroll
      for (int i = 0; i < NL; i++) r[i] -= X[i];
    }
  }
  if (st.p1 & DOTF_WRED) weak_reduce(r, qp_table);
  if ((st.p1 & (DOTF_MULT | DOTF_WRED)) || st.lin) carry_norm(r);
  if (st.p1 & DOTF_HALVE) { if (w0 & (1u << 19)) halve28(r); }
#pragma unroll
  for (int i = 0; i < NL; i++) res[i] = r[i];
  return slot_addr(w0, cx);
}

// every step kind except K_DOT
template <typename LDSP>
NBLS_HD u32 exec_lane(const Step& st, const u32* d /* first 8 descriptor words */, LDSP lds, const LaneCtx& cx, const IOBuf* bufs, u32* res, const u32* __restrict__ qp_table) {
  switch (st.kind) {
    case K_LIN: {
      const u32 w0 = d[0];
      const int nadd = st.p0, nsub = st.p1;
      u32 r[NL];
      // the first field is always an added term (the host pads with the zero constant); added terms first, then the subtracted ones, as two
      // separate runs of uniform branches: one run whose body chooses between adding and subtracting cost 14 register copies per term
      ld14(r, lds, slot_addr(field16(d, 1, 0), cx));
#pragma unroll
      for (int t = 1; t < MAX_LIN_TERMS; t++) {
        if (t < nadd) {   // uniform
          u32 X[NL];
          ld14(X, lds, slot_addr(field16(d, 1, t), cx));
#pragma unroll
          for (int i = 0; i < NL; i++) r[i] += X[i];
        }
      }
#pragma unroll
      for (int t = 1; t < 2 * MAX_LIN_TERMS; t++) {
        if (t >= nadd && t < nadd + nsub) {   // uniform
          u32 X[NL];
          ld14(X, lds, slot_addr(field16(d, 1, t), cx));
#pragma unroll
          for (int i = 0; i < NL; i++) r[i] -= X[i];
        }
      }
      if (st.lin & 1) weak_reduce(r, qp_table);     // LIN steps carry the weak-reduction flag in the header's `lin` field
      carry_norm(r);
      if (w0 & (1u << 16)) halve28(r);
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = r[i];
      return slot_addr(w0, cx);
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
      return slot_addr(w0, cx);
    }
    case K_LOADW: {
      u32 w0 = d[0], off = d[1];
      const IOBuf& b = bufs[(w0 >> 16) & 7];
      const u32* src = (const u32*)(b.ptr + (u64)cx.item * b.stride + off);
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = cx.live ? src[i] : 0u;
      return slot_addr(w0, cx);
    }
    case K_STORE: {
      u32 w0 = d[0], off = d[1];
      u32 X[NL], w[12];
      ld14(X, lds, slot_addr(w0, cx));
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
      ld14(X, lds, slot_addr(w0, cx));
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
      ld14(X, lds, slot_addr(w0 >> 16, cx));
      bool z = is_zero_mod_p(X);
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = 0;
      res[0] = z ? 1u : 0u;
      return slot_addr(w0, cx);
    }
    case K_SEL: {
      u32 w0 = d[0], w1 = d[1];
      // both sources are read and merged with a mask: the LDS access pattern does not depend on the flag (the flag
      // can be a secret scalar bit in the sign / getPublicKey ladders)
      const u32 m = 0u - (ld1(lds, slot_addr(w0 >> 16, cx)) != 0 ? 1u : 0u);
      u32 Xa[NL], Xb[NL];
      ld14(Xa, lds, slot_addr(w1, cx));
      ld14(Xb, lds, slot_addr(w1 >> 16, cx));
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = (Xa[i] & m) | (Xb[i] & ~m);
      return slot_addr(w0, cx);
    }
    case K_CANON: {
      u32 w0 = d[0];
      ld14(res, lds, slot_addr(w0 >> 16, cx));
      csub_p(res);
      return slot_addr(w0, cx);
    }
    case K_CMP: {
      u32 w0 = d[0], w1 = d[1];
      u32 X[NL], Y[NL];
      ld14(X, lds, slot_addr(w1, cx));
      u32 f;
      if (st.p0 == 0) {   // X > Y  <=>  Y - X borrows (normalised limbs)
        ld14(Y, lds, slot_addr(w1 >> 16, cx));
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
      return slot_addr(w0, cx);
    }
    case K_BIT: {
      u32 w0 = d[0], bit = d[1];
      u32 w = ld1(lds, slot_addr(w0 >> 16, cx) + 4 * (bit / 28));
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = 0;
      res[0] = (w >> (bit % 28)) & 1;
      return slot_addr(w0, cx);
    }
    case K_BITAND: {
      u32 w0 = d[0], w1 = d[1];
      u32 X[NL], Y[NL];
      ld14(X, lds, slot_addr(w1, cx));
      ld14(Y, lds, slot_addr(w1 >> 16, cx));
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = X[i] & Y[i];
      return slot_addr(w0, cx);
    }
    case K_FLAG: {
      u32 w0 = d[0], w1 = d[1];
      u32 a = ld1(lds, slot_addr(w1, cx)) & 1, b = ld1(lds, slot_addr(w1 >> 16, cx)) & 1;
      u32 f = st.p0 == 0 ? (a & b) : st.p0 == 1 ? (a | b) : st.p0 == 2 ? (a ^ b) : (a & (b ^ 1));
#pragma unroll
      for (int i = 0; i < NL; i++) res[i] = 0;
      res[0] = f;
      return slot_addr(w0, cx);
    }
    case K_STATUS: {
      u32 w0 = d[0];
      u32 n = w0 & 0xff, code = 0;
      for (int k = (int)n - 1; k >= 0; k--) {
        u32 e = d[1 + k];
        if ((ld1(lds, slot_addr(e, cx)) & 1) == 0) code = e >> 16;
      }
      if (cx.live) { const IOBuf& b = bufs[(w0 >> 16) & 7]; ((int8_t*)b.ptr)[(u64)cx.item * b.stride] = (int8_t)code; }
      return 0xffffffffu;
    }
  }
  return 0xffffffffu;
}

}  // namespace nbls
