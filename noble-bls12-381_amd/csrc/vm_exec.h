// vm_exec.h -- per-lane semantics of every step kind, written once and compiled twice:
//   * vm_kernel.hip : the gfx950 kernel (LDS = __shared__ memory, one wavefront per workgroup)
//   * vm_sim.cpp    : a host-side simulator used ONLY by the CPU test-suite to check compiled programs
//                     without a GPU (tests/); it is not part of libnbls.so.
// Field representation: 12 x 32-bit little-endian limbs, Montgomery form (R = 2^384), redundant range [0,2p).
#pragma once
#include "vm.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NBLS_HD __host__ __device__ __forceinline__
#else
#define NBLS_HD inline
#endif

namespace nbls {
typedef uint32_t u32;
typedef uint64_t u64;

#define NBLS_P32 {0xffffaaabu,0xb9feffffu,0xb153ffffu,0x1eabfffeu,0xf6b0f624u,0x6730d2a0u,0xf38512bfu,0x64774b84u,0x434bacd7u,0x4b1ba7b6u,0x397fe69au,0x1a0111eau}
#define NBLS_2P32 {0xffff5556u,0x73fdffffu,0x62a7ffffu,0x3d57fffdu,0xed61ec48u,0xce61a541u,0xe70a257eu,0xc8ee9709u,0x869759aeu,0x96374f6cu,0x72ffcd34u,0x340223d4u}
#define NBLS_N0INV 0xfffcfffdu

NBLS_HD u32 addc(u32 a, u32 b, u32 cin, u32* cout) {
#if defined(__has_builtin) && __has_builtin(__builtin_addc)
  unsigned co; u32 r = __builtin_addc(a, b, cin, &co); *cout = co; return r;
#else
  u64 s = (u64)a + b + cin; *cout = (u32)(s >> 32); return (u32)s;
#endif
}
NBLS_HD u32 subb(u32 a, u32 b, u32 bin, u32* bout) {
#if defined(__has_builtin) && __has_builtin(__builtin_subc)
  unsigned bo; u32 r = __builtin_subc(a, b, bin, &bo); *bout = bo; return r;
#else
  u64 d = (u64)a - b - bin; *bout = (u32)(d >> 63); return (u32)d;
#endif
}

// r = a*b/R (mod p), r < a*b/R + p.  Row-wise CIOS: 12 MADs (v_mad_u64_u32) + one 32-bit carry chain per row.
NBLS_HD void mont_mul12(u32* __restrict__ r, const u32* a, const u32* b) {
  const u32 P[12] = NBLS_P32;
  u32 t[13];
#pragma unroll
  for (int i = 0; i < 13; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    u32 lo[12], hi[12], s[13];
#pragma unroll
    for (int j = 0; j < 12; j++) { u64 x = (u64)a[j] * b[i] + t[j]; lo[j] = (u32)x; hi[j] = (u32)(x >> 32); }
    u32 c = 0;
    s[0] = lo[0];
#pragma unroll
    for (int j = 1; j < 12; j++) s[j] = addc(lo[j], hi[j - 1], c, &c);
    s[12] = addc(t[12], hi[11], c, &c);
    u32 top = c;
    u32 m = s[0] * NBLS_N0INV;
#pragma unroll
    for (int j = 0; j < 12; j++) { u64 x = (u64)m * P[j] + s[j]; lo[j] = (u32)x; hi[j] = (u32)(x >> 32); }
    c = 0;
#pragma unroll
    for (int j = 1; j < 12; j++) t[j - 1] = addc(lo[j], hi[j - 1], c, &c);
    t[11] = addc(s[12], hi[11], c, &c);
    t[12] = top + c;
  }
#pragma unroll
  for (int i = 0; i < 12; i++) r[i] = t[i];
  // t[12] == 0 whenever a*b < 2^384 * p * 8 (all callers: operands < 4p)
}

// if (x >= m) x -= m, N words
template <int N>
NBLS_HD void csub(u32* x, const u32* m) {
  u32 d[N], br = 0;
#pragma unroll
  for (int i = 0; i < N; i++) d[i] = subb(x[i], m[i], br, &br);
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = br ? x[i] : d[i];
}

// A = a0 (+|-) a1 for a MUL operand; mode 1 add, 2 sub (adds 2p so the value stays non-negative).  Result < 4p.
NBLS_HD void pre_add(u32* A, const u32* a1, u32 mode) {
  const u32 P2[12] = NBLS_2P32;
  u32 mask = (mode == 2) ? 0xffffffffu : 0u, c = (mode == 2) ? 1u : 0u;
#pragma unroll
  for (int i = 0; i < 12; i++) A[i] = addc(A[i], a1[i] ^ mask, c, &c);
  c = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) A[i] = addc(A[i], P2[i] & mask, c, &c);
}

// operand of a DOT product: (+-x) or (+-x +- y), plus 2p per negated term; result in (0, 4p)
template <typename LDSP>
NBLS_HD void dot_operand(u32* A, u32 enc, LDSP lds, u32 inst) {
  const u32 P2[12] = NBLS_2P32;
  const u32 e0 = enc & 0xffff, e1 = enc >> 16;
  const u32 o0 = ((e0 & OP_CONST) ? 0u : inst) + (e0 & OP_SLOT_MASK) * 12u;
#pragma unroll
  for (int i = 0; i < 12; i++) A[i] = lds[o0 + i];
  if ((enc & (OP_NEG | (OP_PRESENT << 16))) == 0) return;   // plain slot: the common case
  if (e0 & OP_NEG) {   // 2p - x
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) A[i] = subb(P2[i], A[i], br, &br);
  }
  if (e1 & OP_PRESENT) {
    const u32 o1 = ((e1 & OP_CONST) ? 0u : inst) + (e1 & OP_SLOT_MASK) * 12u;
    u32 m1 = (e1 & OP_NEG) ? 0xffffffffu : 0u, c = m1 & 1u;
#pragma unroll
    for (int i = 0; i < 12; i++) A[i] = addc(A[i], lds[o1 + i] ^ m1, c, &c);
    if (m1) {
      c = 0;
#pragma unroll
      for (int i = 0; i < 12; i++) A[i] = addc(A[i], P2[i], c, &c);
    }
  }
}

// acc (25 words) += a * b  (full 24-word product, then one 25-word addition)
NBLS_HD void wide_mac(u32* acc, const u32* a, const u32* b) {
  u32 t[24];
#pragma unroll
  for (int i = 0; i < 24; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    u32 lo[12], hi[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { u64 x = (u64)a[j] * b[i] + t[i + j]; lo[j] = (u32)x; hi[j] = (u32)(x >> 32); }
    u32 c = 0;
    t[i] = lo[0];
#pragma unroll
    for (int j = 1; j < 12; j++) t[i + j] = addc(lo[j], hi[j - 1], c, &c);
    t[i + 12] = hi[11] + c;   // fresh word: the partial product of rows 0..i is < 2^(32(i+13))
  }
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 24; i++) acc[i] = addc(acc[i], t[i], c, &c);
  acc[24] += c;
}

// ---- lazy-carry accumulation --------------------------------------------------------------------------------
// The sum of products of a DOT lane-op is accumulated WITHOUT carry propagation: word positions (2k, 2k+1) share the 64-bit
// accumulator e[k], positions (2k+1, 2k+2) share o[k]; a limb product a_j*b_i (64 bits, at position i+j) is added to the
// accumulator aligned with it by ONE v_mad_u64_u32, whose hardware carry-out is counted in a third word (ec[k] at position
// 2k+2, oc[k] at 2k+3) by ONE v_addc.  No per-row carry chain, no register shuffling; carries are resolved once per lane-op
// (lazy_normalize) before the Montgomery reduction.
struct LazyAcc { u64 e[12]; u64 o[12]; u32 ec[12]; u32 oc[12]; };

NBLS_HD void mac3(u64& acc, u32& cw, u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  u64 cy;
  asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2" : "+v"(acc), "+v"(cw), "=&s"(cy) : "v"(a), "v"(b));
#else
  u64 s = acc + (u64)a * b; cw += (s < acc) ? 1u : 0u; acc = s;
#endif
}
NBLS_HD void lazy_zero(LazyAcc& L) {
#pragma unroll
  for (int i = 0; i < 12; i++) { L.e[i] = 0; L.o[i] = 0; L.ec[i] = 0; L.oc[i] = 0; }
}
NBLS_HD void lazy_mac(LazyAcc& L, const u32* a, const u32* b) {
#pragma unroll
  for (int i = 0; i < 12; i++) {
#pragma unroll
    for (int j = 0; j < 12; j++) {
      const int w = i + j;
      if ((w & 1) == 0) mac3(L.e[w / 2], L.ec[w / 2], a[j], b[i]); else mac3(L.o[(w - 1) / 2], L.oc[(w - 1) / 2], a[j], b[i]);
    }
  }
}
// resolve the carries: t[0..24] = the accumulated integer
NBLS_HD void lazy_normalize(u32* t, const LazyAcc& L) {
  u64 c = 0;
#pragma unroll
  for (int w = 0; w < 25; w++) {
    u64 s = c;
    if ((w & 1) == 0) { if (w / 2 < 12) s += (u32)L.e[w / 2]; if (w >= 2) { s += (u32)(L.o[(w - 2) / 2] >> 32); s += L.ec[(w - 2) / 2]; } }
    else { s += (u32)(L.e[(w - 1) / 2] >> 32); s += (u32)L.o[(w - 1) / 2]; if (w >= 3) s += L.oc[(w - 3) / 2]; }
    t[w] = (u32)s; c = s >> 32;
  }
}

// Montgomery reduction of a 25-word accumulator V: r (13 words) = V / R mod-ish, r < V/R + p
NBLS_HD void wide_redc(u32* r, u32* acc) {
  const u32 P[12] = NBLS_P32;
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    u32 m = acc[i] * NBLS_N0INV;
    u32 lo[12], hi[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { u64 x = (u64)m * P[j] + acc[i + j]; lo[j] = (u32)x; hi[j] = (u32)(x >> 32); }
    u32 c = 0;
#pragma unroll
    for (int j = 1; j < 12; j++) acc[i + j] = addc(lo[j], hi[j - 1], c, &c);
    u32 c1, c2;
    u32 w = addc(acc[i + 12], hi[11], c, &c1);
    acc[i + 12] = addc(w, carry, 0, &c2);
    carry = c1 + c2;
  }
#pragma unroll
  for (int i = 0; i < 12; i++) r[i] = acc[12 + i];
  r[12] = acc[24] + carry;
}

NBLS_HD u32 bswap32(u32 x) { return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24); }

NBLS_HD bool is_zero_mod_p(const u32* x) {   // x in [0,2p)
  const u32 P[12] = NBLS_P32;
  u32 z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) { z |= x[i]; e |= x[i] ^ P[i]; }
  return z == 0 || e == 0;
}

}  // namespace nbls

// ------------------------------------------------------------------------------------------------
// Per-lane step execution.  `lds` is the workgroup's LDS image (device) or a plain array (simulator):
//   words [0, nconst*12)            program constants (shared by all instances)
//   words [pm2, pm2 + 17*16)        k * 2p for k = 0..16, 13 significant words each (LIN offsets / reduction moduli)
//   words [inst, inst + slots*12)   this instance's slots
// The function reads operands, computes, and returns the result in `res` together with the destination word
// offset (or 0xffffffff when the step has no LDS destination); the caller commits the 12 words afterwards, so
// that every read of a step precedes every write of that step (in-order LDS within a wavefront; explicit
// two-phase loop in the simulator).
namespace nbls {

struct LaneCtx {
  u32 inst;       // word offset of the instance region
  u32 pm2;        // word offset of the PM2 table
  u32 item;       // global work-item index
  bool live;      // item < n_items (dead instances compute on zeros but never touch global memory)
};

NBLS_HD u32 slot_addr(u32 op, u32 inst) { return ((op & OP_CONST) ? 0u : inst) + (op & OP_SLOT_MASK) * 12u; }

template <typename LDSP>
NBLS_HD void ld12(u32* x, LDSP lds, u32 off) {
#pragma unroll
  for (int i = 0; i < 12; i++) x[i] = lds[off + i];
}

template <typename LDSP>
NBLS_HD u32 exec_lane(const Step& st, const u32* d /* first 8 descriptor words, already loaded */, const u32* __restrict__ gd /* this lane's descriptor in global memory */,
                      LDSP lds, const LaneCtx& cx, const IOBuf* bufs, u32* res) {
  const u32 P[12] = NBLS_P32;
  const u32 P2[12] = NBLS_2P32;
  switch (st.kind) {
    case K_MUL: {
      u32 w0 = d[0], w1 = d[1], w2 = d[2];
      u32 A[12], B[12], X[12];
      ld12(A, lds, slot_addr(w0 & 0xffff, cx.inst));
      ld12(B, lds, slot_addr(w1 & 0xffff, cx.inst));
      if (st.p0 & 1) { u32 m = w0 >> (16 + OP_MODE_SHIFT); if (m) { ld12(X, lds, slot_addr(w0 >> 16, cx.inst)); pre_add(A, X, m); } }
      if (st.p0 & 2) { u32 m = w1 >> (16 + OP_MODE_SHIFT); if (m) { ld12(X, lds, slot_addr(w1 >> 16, cx.inst)); pre_add(B, X, m); } }
      mont_mul12(res, A, B);
      csub<12>(res, P2);
      return slot_addr(w2 & 0xffff, cx.inst);
    }
    case K_DOT: {
      const u32 w0 = d[0];
      const u32 k = (w0 >> 16) & 0xf, L = (w0 >> 20) & 0xf, mult = (w0 >> 24) & 0x7;
      u32 r[13];
      if (st.p0 > 0) {   // uniform
        u32 acc[25];
#pragma unroll
        for (int i = 0; i < 25; i++) acc[i] = 0;
        u32 na = d[4], nb = d[5];
        for (u32 i = 0; i < st.p0; i++) {   // uniform trip count; next product's operand words are fetched ahead
          const u32 ea = na, eb = nb;
          if (i + 1 < st.p0) { na = gd[6 + 2 * i]; nb = gd[7 + 2 * i]; }
          if (i < k) {
            u32 A[12], B[12];
            dot_operand(A, ea, lds, cx.inst);
            dot_operand(B, eb, lds, cx.inst);
            wide_mac(acc, A, B);
          }
        }
        wide_redc(r, acc);
      } else {
#pragma unroll
        for (int i = 0; i < 13; i++) r[i] = 0;
      }
      if (mult > 1) {   // m * dot, m <= 4
        u32 t[13];
#pragma unroll
        for (int i = 0; i < 13; i++) t[i] = r[i];
        for (u32 j = 1; j < mult; j++) {
          u32 c = 0;
#pragma unroll
          for (int i = 0; i < 13; i++) r[i] = addc(r[i], t[i], c, &c);
        }
      }
      u32 nneg = 0;
#pragma unroll
      for (int t = 0; t < MAX_DOT_LINEAR; t++) {
        if (t < (int)st.pad) {   // uniform
          u32 term = (d[2 + t / 2] >> (16 * (t & 1))) & 0xffff;
          if ((u32)t < L) {
            u32 neg = (term >> OP_MODE_SHIFT) & 1, mask = neg ? 0xffffffffu : 0u, c = neg;
            u32 X[12];
            ld12(X, lds, slot_addr(term, cx.inst));
#pragma unroll
            for (int i = 0; i < 12; i++) r[i] = addc(r[i], X[i] ^ mask, c, &c);
            r[12] = r[12] + mask + c;
            nneg += neg;
          }
        }
      }
      if (st.pad > 0) {
        u32 c = 0, off = cx.pm2 + nneg * 16;
#pragma unroll
        for (int i = 0; i < 13; i++) r[i] = addc(r[i], lds[off + i], c, &c);
      }
      for (int s = (int)st.p1 - 1; s >= 0; s--) {
        u32 M[13], off = cx.pm2 + (16u << s);
#pragma unroll
        for (int i = 0; i < 13; i++) M[i] = lds[off + i];
        csub<13>(r, M);
      }
      if (w0 & (1u << 27)) {   // halve
        u32 mask = (r[0] & 1) ? 0xffffffffu : 0u, c = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) r[i] = addc(r[i], P[i] & mask, c, &c);
#pragma unroll
        for (int i = 0; i < 11; i++) r[i] = (r[i] >> 1) | (r[i + 1] << 31);
        r[11] >>= 1;
      }
#pragma unroll
      for (int i = 0; i < 12; i++) res[i] = r[i];
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_LIN: {
      const u32* w = d;
      u32 nt = (w[0] >> 16) & 0xff, nneg = 0;
      u32 acc[13];
#pragma unroll
      for (int i = 0; i < 13; i++) acc[i] = 0;
#pragma unroll
      for (int t = 0; t < 14; t++) {
        if (t < (int)st.p0) {   // uniform
          u32 term = (w[1 + t / 2] >> (16 * (t & 1))) & 0xffff;
          if ((u32)t < nt) {
            u32 neg = (term >> OP_MODE_SHIFT) & 1, mask = neg ? 0xffffffffu : 0u, c = neg;
            u32 X[12];
            ld12(X, lds, slot_addr(term, cx.inst));
#pragma unroll
            for (int i = 0; i < 12; i++) acc[i] = addc(acc[i], X[i] ^ mask, c, &c);
            acc[12] = acc[12] + mask + c;
            nneg += neg;
          }
        }
      }
      {
        u32 c = 0, off = cx.pm2 + nneg * 16;
#pragma unroll
        for (int i = 0; i < 13; i++) acc[i] = addc(acc[i], lds[off + i], c, &c);
      }
      for (int s = (int)st.p1 - 1; s >= 0; s--) {
        u32 M[13], off = cx.pm2 + (16u << s);
#pragma unroll
        for (int i = 0; i < 13; i++) M[i] = lds[off + i];
        csub<13>(acc, M);
      }
      if (w[0] & (1u << 24)) {   // halve: (x + (x odd ? p : 0)) >> 1, result < 1.5p
        u32 mask = (acc[0] & 1) ? 0xffffffffu : 0u, c = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) acc[i] = addc(acc[i], P[i] & mask, c, &c);
#pragma unroll
        for (int i = 0; i < 11; i++) acc[i] = (acc[i] >> 1) | (acc[i + 1] << 31);
        acc[11] >>= 1;
      }
#pragma unroll
      for (int i = 0; i < 12; i++) res[i] = acc[i];
      return slot_addr(w[0] & 0xffff, cx.inst);
    }
    case K_LOAD: {
      u32 w0 = d[0], off = d[1];
      const IOBuf& b = bufs[(w0 >> 16) & 7];
      const u32* src = (const u32*)(b.ptr + (u64)cx.item * b.stride + off);
      const int nw = st.p0 ? (int)st.p0 / 4 : 12;   // number of 32-bit words (big-endian integer of p0 bytes)
#pragma unroll
      for (int i = 0; i < 12; i++) res[i] = (cx.live && i < nw) ? bswap32(src[nw - 1 - i]) : 0u;
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_LOADW: {
      u32 w0 = d[0], off = d[1];
      const IOBuf& b = bufs[(w0 >> 16) & 7];
      const u32* src = (const u32*)(b.ptr + (u64)cx.item * b.stride + off);
#pragma unroll
      for (int i = 0; i < 12; i++) res[i] = cx.live ? src[i] : 0u;
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_STORE: {
      u32 w0 = d[0], off = d[1];
      u32 X[12];
      ld12(X, lds, slot_addr(w0 & 0xffff, cx.inst));
      csub<12>(X, P);
      if (cx.live) {
        const IOBuf& b = bufs[(w0 >> 16) & 7];
        u32* dst = (u32*)(b.ptr + (u64)cx.item * b.stride + off);
#pragma unroll
        for (int i = 0; i < 12; i++) dst[11 - i] = bswap32(X[i]);
      }
      return 0xffffffffu;
    }
    case K_STOREW: {
      u32 w0 = d[0], off = d[1];
      u32 X[12];
      ld12(X, lds, slot_addr(w0 & 0xffff, cx.inst));
      if (cx.live) {
        const IOBuf& b = bufs[(w0 >> 16) & 7];
        u32* dst = (u32*)(b.ptr + (u64)cx.item * b.stride + off);
#pragma unroll
        for (int i = 0; i < 12; i++) dst[i] = X[i];
      }
      return 0xffffffffu;
    }
    case K_ISZ: {
      u32 w0 = d[0];
      u32 X[12];
      ld12(X, lds, slot_addr(w0 >> 16, cx.inst));
      bool z = is_zero_mod_p(X);
#pragma unroll
      for (int i = 0; i < 12; i++) res[i] = 0;
      res[0] = z ? 1u : 0u;
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_SEL: {
      u32 w0 = d[0], w1 = d[1];
      u32 f = lds[slot_addr(w0 >> 16, cx.inst)];
      u32 src = f ? (w1 & 0xffff) : (w1 >> 16);
      ld12(res, lds, slot_addr(src, cx.inst));
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_CANON: {
      u32 w0 = d[0];
      ld12(res, lds, slot_addr(w0 >> 16, cx.inst));
      csub<12>(res, P);
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_CMP: {
      u32 w0 = d[0], w1 = d[1];
      u32 X[12], Y[12];
      ld12(X, lds, slot_addr(w1 & 0xffff, cx.inst));
      u32 f;
      if (st.p0 == 0) {   // X > Y  <=>  Y - X borrows
        ld12(Y, lds, slot_addr(w1 >> 16, cx.inst));
        u32 br = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) (void)subb(Y[i], X[i], br, &br);
        f = br;
      } else {
        f = X[0] & 1;
      }
#pragma unroll
      for (int i = 0; i < 12; i++) res[i] = 0;
      res[0] = f;
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_BIT: {
      u32 w0 = d[0], bit = d[1];
      u32 w = lds[slot_addr(w0 >> 16, cx.inst) + (bit >> 5)];
#pragma unroll
      for (int i = 0; i < 12; i++) res[i] = 0;
      res[0] = (w >> (bit & 31)) & 1;
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_BITAND: {
      u32 w0 = d[0], w1 = d[1];
      u32 X[12], Y[12];
      ld12(X, lds, slot_addr(w1 & 0xffff, cx.inst));
      ld12(Y, lds, slot_addr(w1 >> 16, cx.inst));
#pragma unroll
      for (int i = 0; i < 12; i++) res[i] = X[i] & Y[i];
      return slot_addr(w0 & 0xffff, cx.inst);
    }
    case K_FLAG: {
      u32 w0 = d[0], w1 = d[1];
      u32 a = lds[slot_addr(w1 & 0xffff, cx.inst)] & 1, b = lds[slot_addr(w1 >> 16, cx.inst)] & 1;
      u32 f = st.p0 == 0 ? (a & b) : st.p0 == 1 ? (a | b) : st.p0 == 2 ? (a ^ b) : (a & (b ^ 1));
#pragma unroll
      for (int i = 0; i < 12; i++) res[i] = 0;
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
