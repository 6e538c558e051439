// vm_wide_kernel.hip -- the one-limb-per-lane interpreter (wide_exec.h): ONE item per workgroup, a lane-op per row of sixteen lanes, four rows per wavefront, ceil(W / 4)
// wavefronts per workgroup; the compiled step program, its descriptors and the LDS slot layout are the plain interpreter's (vm.h).  Two barriers per step.
// For launches of a few hundred items at most (nbls_internal.h wide_max): the final exponentiation of a single verify / sign, the one-element tail of every verifyBatch.
#include <hip/hip_runtime.h>
#include "wide_exec.h"
#include "g1_wide.h"
#include "fp_inv_wide.h"

namespace nbls {

struct WideLane {
  typedef i32 I;
  typedef i64 W;
  char* lds; u32 j4, j, base;      // base: start of the item's slot region (behind the shared constants, vm_exec.h term_addr)
  i32 pr0, pr1, pr2, pr3, pj;      // p_j on row k of the wavefront (0 on the other rows) ; p_j
  __device__ __forceinline__ bool skip_rows() const { return false; }
  u32 lmask13, cmask13;            // lane < 13: 2^28 - 1 / all ones ; lane 13 and above: all ones / zero (the top limb keeps the rest)
  u32 rowshift = 0;                // 16 * (row inside the wavefront): the first lane of this lane's row (fp_inv_wide.h)
  __device__ __forceinline__ I low13(I a) const { return (i32)((u32)a & lmask13); }
  __device__ __forceinline__ I carry13(I a) const { return (i32)((u32)a & cmask13); }
  __device__ __forceinline__ I zero() const { return 0; }
  __device__ __forceinline__ void fence() const { __builtin_amdgcn_sched_barrier(0); }
  __device__ __forceinline__ I konst(u32 k) const { return (i32)k; }
  __device__ __forceinline__ I add(I a, I b) const { return a + b; }
  __device__ __forceinline__ I sub(I a, I b) const { return a - b; }
  __device__ __forceinline__ I and_(I a, u32 m) const { return (i32)((u32)a & m); }
  __device__ __forceinline__ I sar(I a, int k) const { return a >> k; }
  __device__ __forceinline__ I mul_lo(I a, u32 k) const { return (i32)((u32)a * k); }
  __device__ __forceinline__ I mul_small(I a, u32 k) const { return a * (i32)k; }
  __device__ __forceinline__ I muls(I a, int k) const { return a * k; }
  // fp_inv_wide.h: row-uniform scalars and lane picks
  __device__ __forceinline__ I or_(I a, I b) const { return a | b; }
  __device__ __forceinline__ I spread(i32 s) const { return s; }
  __device__ __forceinline__ i32 first(I v) const { return v; }
  __device__ __forceinline__ I plimbs() const { return pj; }
  __device__ __forceinline__ u32 nonzero_mask(I v) const { const u64 m = __builtin_amdgcn_ballot_w64(v != 0); return (u32)(m >> (rowshift)) & 0xffffu; }
  __device__ __forceinline__ I gather(I v, u32 k) const { return __builtin_amdgcn_ds_bpermute((int)((rowshift + k) << 2), v); }
  __device__ __forceinline__ I pick(u32 lanebit, I a, I b) const { return ((lanebit >> j) & 1u) ? a : b; }
  __device__ __forceinline__ I lane_eq(u32 k, I a, I b) const { return j == k ? a : b; }
  __device__ __forceinline__ I lo(W w) const { return (i32)w; }
  __device__ __forceinline__ W wzero() const { return 0; }
  __device__ __forceinline__ W mad(I a, I b, W acc) const { return acc + (i64)a * (i64)b; }
  // lane 0 of every row to its row by three DPP moves (quad broadcast, then the first quad to the second, then the first half to the second: bank_mask picks the quads written) --
  // all four rows of the wavefront at once and no SGPR in the dependency chain (four v_readlane + four masked multiply-adds before)
  __device__ __forceinline__ W mad_p(I ml, W acc) const {
#if !defined(NBLS_WIDE_READLANE)
    i32 m = __builtin_amdgcn_update_dpp(0, ml, 0x00 /* quad_perm [0,0,0,0] */, 0xf, 0xf, false);
    m = __builtin_amdgcn_update_dpp(m, m, 0x114 /* row_shr:4 */, 0xf, 0x2, false);
    m = __builtin_amdgcn_update_dpp(m, m, 0x118 /* row_shr:8 */, 0xf, 0xc, false);
    return (i64)((u64)acc + (u64)(u32)pj * (u32)m);
#else
    const i32 m0 = __builtin_amdgcn_readlane(ml, 0), m1 = __builtin_amdgcn_readlane(ml, 16), m2 = __builtin_amdgcn_readlane(ml, 32), m3 = __builtin_amdgcn_readlane(ml, 48);
    u64 a = (u64)acc;
    a += (u64)(u32)pr0 * (u32)m0; a += (u64)(u32)pr1 * (u32)m1; a += (u64)(u32)pr2 * (u32)m2; a += (u64)(u32)pr3 * (u32)m3;
    return (i64)a;
#endif
  }
  __device__ __forceinline__ W mad_pq(I q, W acc) const { return (i64)((u64)acc + (u64)(u32)pj * (u32)q); }      // q: a small non-negative multiplier
  __device__ __forceinline__ W sar28(W w) const { return w >> 28; }
  __device__ __forceinline__ W addww(W a, W b) const { return a + b; }
  __device__ __forceinline__ W addw(W w, I x) const { return (i64)((u64)w + (u32)x); }      // x: a non-negative 28-bit low part
  __device__ __forceinline__ I wred_q(I top) const { i32 t = top - 9; t = t < 0 ? 0 : t; const u32 q = __umulhi((u32)t, 2642610142u) >> 16; return (i32)(q > (u32)(QP_TABLE_ENTRIES - 1) ? (u32)(QP_TABLE_ENTRIES - 1) : q); }
  __device__ __forceinline__ I bcast(I v, int i) const {      // ds_swizzle, bit-mask mode: source lane = (lane & 0x10) | i inside every group of 32 lanes
    switch (i) {
#define NBLS_SWZ(K) case K: return __builtin_amdgcn_ds_swizzle(v, 0x10 | (K << 5));
      NBLS_SWZ(0) NBLS_SWZ(1) NBLS_SWZ(2) NBLS_SWZ(3) NBLS_SWZ(4) NBLS_SWZ(5) NBLS_SWZ(6) NBLS_SWZ(7) NBLS_SWZ(8) NBLS_SWZ(9) NBLS_SWZ(10) NBLS_SWZ(11) NBLS_SWZ(12) default: NBLS_SWZ(13)
#undef NBLS_SWZ
    }
  }
  __device__ __forceinline__ I shl1(I v) const { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, true); }      // row_shl:1: lane j <- lane j + 1
  __device__ __forceinline__ I shr1(I v) const { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }      // row_shr:1: lane j <- lane j - 1
  __device__ __forceinline__ u32 addr(u32 f) const { return (f & 2u) ? base + f - 2u : f; }      // bit 1 marks a slot of the instance region, without it the offset is a (shared) constant's
  __device__ __forceinline__ I ld(u32 f) const { return *(const i32*)(lds + addr(f) + j4); }
  __device__ __forceinline__ I gload(const u32* g, bool live) const { i32 x = (live && j < (u32)NL) ? (i32)g[j] : 0; asm volatile("" : "+v"(x)); return x; }      // (settled here: see `settle` in the kernel)
  __device__ __forceinline__ void gstore(u32* g, I v, bool live) const { if (live) g[j] = (u32)v; }      // lanes 14, 15 hold zero: the element's padding words
};
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the lane-op's descriptor: ten 16-byte words in registers (8 header words + 4 per product round)
struct WideDesc {
  uint4 q[10];
  __device__ __forceinline__ u32 operator()(int k) const { const uint4& v = q[k >> 2]; return (k & 3) == 0 ? v.x : (k & 3) == 1 ? v.y : (k & 3) == 2 ? v.z : v.w; }
};

extern "C" __global__ void __launch_bounds__(256) nbls_vm_kernel_wide(KernelArgs ka) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds = smem;
  const u32 tid = threadIdx.x, nthreads = blockDim.x, row = tid >> 4, j = tid & 15u;
  u32 n_items = ka.n_items;
  if (ka.n_items_dev) { const u32 v = *ka.n_items_dev; n_items = v < n_items ? v : n_items; }
  u32 item = blockIdx.x;
  const bool live = item < n_items;
  if (!live) return;                                      // the whole workgroup (one item) leaves together
  if (ka.item_index) item = ka.item_index[item];
  // LDS image of the one instance: zeroed (the padding words of every slot stay zero: lanes 14, 15 read them), then the constants
  const u32 base = ka.shared_consts ? ka.nconst * ka.slot_bytes : 0u, image = base + ka.inst_bytes;
  for (u32 i = tid; i < image / 4; i += nthreads) ((u32*)lds)[i] = 0;
  __syncthreads();
  for (u32 i = tid; i < ka.nconst * 16; i += nthreads) { const u32 c = i >> 4, l = i & 15u; if (l < (u32)NL) *(u32*)(lds + c * ka.slot_bytes + 4 * l) = ka.consts[c * RAW_WORDS + l]; }
  const u32 P[NL] = NBLS_P28;
  u32 pj = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) pj = j == (u32)k ? P[k] : pj;
  const u32 rw = (tid >> 4) & 3u;                        // row inside the wavefront
  WideLane l{lds, 4 * j, j, base, rw == 0 ? (i32)pj : 0, rw == 1 ? (i32)pj : 0, rw == 2 ? (i32)pj : 0, rw == 3 ? (i32)pj : 0, (i32)pj, j < 13u ? LMASK : 0xffffffffu, j < 13u ? 0xffffffffu : 0u};
  WideOps<WideLane> o(l);
  __syncthreads();
  const uint4* descs4 = (const uint4*)ka.descs;
  // the step headers through the constant address space: scalar loads.  As plain global loads the compiler issues them on the vector memory path (the kernel stores to global
  // memory and its barriers clobber memory, so nothing proves the headers unchanged) and -- the fields being wavefront-uniform -- reads them back into SGPRs at once: a global
  // round trip in the middle of every step
  typedef const Step __attribute__((address_space(4))) * StepPtr;
  const StepPtr steps = (StepPtr)ka.steps;
  auto step_at = [&](u32 k) __attribute__((always_inline)) { Step x; const u32 __attribute__((address_space(4)))* w = (const u32 __attribute__((address_space(4)))*)(steps + k); u32 t[8];
#pragma unroll
    for (int q = 0; q < 8; q++) t[q] = w[q];
    __builtin_memcpy(&x, t, 32); return x; };
  Step st = step_at(0), nst = step_at(ka.nsteps > 1 ? 1 : 0);
  WideDesc d, nd;
  auto fetch = [&](WideDesc& x, const Step& s) __attribute__((always_inline)) {
    const u32 r = row < s.nlanes ? row : 0u, o4 = (s.desc_off + r * s.stride) >> 2, nq = s.stride >> 2;
#pragma unroll
    for (u32 k = 0; k < 10; k++) if (k < nq) x.q[k] = descs4[o4 + k];
  };
  fetch(d, st);
  // The compiler's wait-count model is not path sensitive: registers that some load targets anywhere around the loop count as "pending" at every merge point, and the first use of
  // a descriptor word in each basic block then waits for vmcnt(0) -- i.e. for the NEXT step's prefetch, just issued.  Passing the words through an empty asm makes them plain
  // VALU values: the one wait sits here (end of the step), where the prefetch has had the whole step to arrive.
  auto settle = [&](WideDesc& x) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 10; k++) { asm volatile("" : "+v"(x.q[k].x), "+v"(x.q[k].y), "+v"(x.q[k].z), "+v"(x.q[k].w)); }
  };
  settle(d);
  const u32 wave_row0 = (tid >> 6) << 2;                 // first row of this wavefront
  for (u32 s = 0; s < ka.nsteps; s++) {
    const Step nnst = step_at((s + 2 < ka.nsteps) ? s + 2 : ka.nsteps - 1);
    fetch(nd, nst);                                        // the next step's descriptor travels while this step computes
    u32 dst = 0; i32 out = 0; bool has = false;
    const bool active = row < st.nlanes;
    if (wave_row0 < st.nlanes) has = wide_step(o, st, d, ka.bufs, item, active, dst, out);      // (a wavefront without an active row only keeps the barriers)
    // the barriers order LDS traffic only: __syncthreads() would also wait for the descriptor prefetch above (a fence over global memory: s_waitcnt vmcnt(0)) and expose a
    // global-memory round trip in every step -- 2.0 us per step where the arithmetic takes 0.8
    lds_barrier();                                         // every read of the step precedes every write of the step
    if (has && active) *(i32*)(lds + l.addr(dst) + 4 * j) = out;
    lds_barrier();
    st = nst; nst = nnst; d = nd; settle(d);
  }
}

// ---- the window combination of the G1 MSM (g1_wide.h): ONE wavefront, the four products of a doubling level on its four rows, the state in LDS slots
__constant__ u32 GW_TABLE_DEV[GW_TABLE_WORDS] = GW_TABLE_INIT;
// S: nwin projective points (3 raw elements each: the per-window sums, window 0 first); out: sum_w 2^(12 w) S_w as a projective point (3 raw elements, exact limbs, below 8 p)
// One workgroup (= one wavefront) per sum: workgroup b reads its nwin points at S + b * nwin * 3 raw elements and writes out + b * 3 raw elements (dev_msm: first the Horner sum
// over the twelve bit-slices of every window, shift 1, a workgroup per window; then the combination of the windows, shift 12, one workgroup)
extern "C" __global__ void __launch_bounds__(64) nbls_g1_wide_combine_kernel(const u32* __restrict__ S, int nwin, int shift, u32* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) char lds[GW_SLOTS * 64];
  S += (size_t)blockIdx.x * (size_t)nwin * 3 * RAW_WORDS; out += (size_t)blockIdx.x * 3 * RAW_WORDS;
  const u32 tid = threadIdx.x, row = tid >> 4, j = tid & 15u;
  for (u32 i = tid; i < (u32)GW_SLOTS * 16u; i += 64u) ((u32*)lds)[i] = 0;
  const u32 P[NL] = NBLS_P28;
  u32 pj = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) pj = j == (u32)k ? P[k] : pj;
  WideLane l{lds, 4 * j, j, 0u, row == 0 ? (i32)pj : 0, row == 1 ? (i32)pj : 0, row == 2 ? (i32)pj : 0, row == 3 ? (i32)pj : 0, (i32)pj, j < 13u ? LMASK : 0xffffffffu, j < 13u ? 0xffffffffu : 0u};
  WideOps<WideLane> o(l);
  WideG1<WideLane> g(o);
  u32 d[GW_DBL_ROUNDS + GW_ADD_ROUNDS][GW_ROW_WORDS];      // this row's part of every round: loop invariant, in registers
#pragma unroll
  for (int q = 0; q < GW_DBL_ROUNDS + GW_ADD_ROUNDS; q++) {
#pragma unroll
    for (int k = 0; k < GW_ROW_WORDS; k++) d[q][k] = GW_TABLE_DEV[(4 * q + row) * GW_ROW_WORDS + k];
  }
  auto put = [&](u32 slot, i32 v) __attribute__((always_inline)) { *(i32*)(lds + slot * 64u + 4u * j) = v; };
  // a point from HBM: row r < 3 reads coordinate r into slot s0 / s1 / s2
  auto fetch = [&](const u32* pt, u32 s0, u32 s1, u32 s2) __attribute__((always_inline)) {
    const i32 v = (row < 3u && j < (u32)NL) ? (i32)pt[row * RAW_WORDS + j] : 0;
    put(row == 0 ? s0 : row == 1 ? s1 : row == 2 ? s2 : (u32)GW_JUNK, v);
  };
  fetch(S + (size_t)(nwin - 1) * 3 * RAW_WORDS, GW_X, GW_U, GW_Z);
  for (int w = nwin - 2; w >= 0; w--) {
    fetch(S + (size_t)w * 3 * RAW_WORDS, GW_X2, GW_Y2, GW_Z2);
#pragma unroll 1
    for (int i = 0; i < shift; i++) {
      i32 r = g.round<1>(d[0]); put(d[0][4], r);
      r = g.round<1>(d[1]); put(d[1][4], r);
    }
#pragma unroll
    for (int q = 0; q < GW_ADD_ROUNDS; q++) { const i32 r = g.round<2>(d[GW_DBL_ROUNDS + q]); put(d[GW_DBL_ROUNDS + q][4], r); }
  }
  // leave: x + p, (U - V) + 2 p, z + p with exact limbs
  const i32 v = row == 0 ? l.ld(GW_X * 64u) : row == 1 ? l.sub(l.ld(GW_U * 64u), l.ld(GW_V * 64u)) : l.ld(GW_Z * 64u);
  const i32 e = g.leave(v, row == 1 ? 2u : 1u);
  if (row < 3u) out[row * RAW_WORDS + j] = (u32)e;
}

// ---- the modular inverse with one limb per lane (fp_inv_wide.h): four elements per wavefront, one per row
extern "C" __global__ void __launch_bounds__(64) nbls_fp_inv_wide_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out) {
  const u32 tid = threadIdx.x, row = tid >> 4, j = tid & 15u;
  const u32 P[NL] = NBLS_P28, R3[NL] = NBLS_R3_INIT;
  u32 pj = 0, r3 = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) { pj = j == (u32)k ? P[k] : pj; r3 = j == (u32)k ? R3[k] : r3; }
  WideLane l{nullptr, 4 * j, j, 0u, row == 0 ? (i32)pj : 0, row == 1 ? (i32)pj : 0, row == 2 ? (i32)pj : 0, row == 3 ? (i32)pj : 0, (i32)pj, j < 13u ? LMASK : 0xffffffffu, j < 13u ? 0xffffffffu : 0u};
  l.rowshift = 16u * row;
  WideOps<WideLane> o(l);
  WideInv<WideLane> w(o);
  const u32 e = blockIdx.x * 4u + row;
  const bool live = e < n;
  const i32 y = (live && j < (u32)NL) ? (i32)in[(size_t)e * SLOT_WORDS + j] : 0;
  const i32 r = w.invert(y, (i32)r3);
  if (live) out[(size_t)e * SLOT_WORDS + j] = j < (u32)NL ? (u32)r : 0u;
}
}  // namespace nbls

extern "C" int nbls_vm_wide_launch(const nbls::KernelArgs* ka, unsigned lds_bytes, void* stream) {
  using namespace nbls;
  if (ka->n_items == 0) return 0;
  if (ka->lsplit != 1 || ka->W > 16 || lds_bytes > 64 * 1024) return -1;
  const unsigned waves = (ka->W + 3) / 4;
  hipLaunchKernelGGL(nbls_vm_kernel_wide, dim3(ka->n_items), dim3(64 * waves), lds_bytes, (hipStream_t)stream, *ka);
  return (int)hipGetLastError();
}
extern "C" int nbls_g1_wide_combine_launch(const void* S, int nwin, int shift, void* out, unsigned sums, void* stream) {
  if (nwin < 1) return -1;
  if (sums == 0) return 0;
  hipLaunchKernelGGL(nbls::nbls_g1_wide_combine_kernel, dim3(sums), dim3(64), 0, (hipStream_t)stream, (const nbls::u32*)S, nwin, shift, (nbls::u32*)out);
  return (int)hipGetLastError();
}
extern "C" int nbls_fp_inv_wide_launch(unsigned n, const void* in, void* out, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(nbls::nbls_fp_inv_wide_kernel, dim3((n + 3) / 4), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out);
  return (int)hipGetLastError();
}
