// vm_kernel.hip -- the gfx950 wave-VM kernel: one wavefront (= one workgroup of 64 lanes) interprets a compiled
// step list for G work items at once; see vm.h / vm_exec.h.  Integer VALU work (v_mad_i64_i32 into lazy 64-bit columns):
// no MFMA by construction (independent 381-bit products are not a dense contraction).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "vm_exec.h"

namespace nbls {

template <bool FAIR> __device__ __forceinline__ void vm_kernel_body(const KernelArgs& ka) {
  extern __shared__ __attribute__((aligned(16))) u32 smem[];
  const u32 lane = threadIdx.x;
  u32 n_items = ka.n_items;
  if (ka.n_items_dev) { const u32 v = *ka.n_items_dev; n_items = v < n_items ? v : n_items; if (blockIdx.x * ka.G >= n_items) return; }
  const u32 shared_words = ka.nconst * SLOT_WORDS;
  for (u32 i = lane; i < shared_words; i += 64) smem[i] = ka.consts[i];
  const u32 W = ka.W;
  const u32 inst_id = lane / W;
  const u32 lane_in = (inst_id < ka.G) ? (lane - inst_id * W) : 0xffffu;
  LaneCtx cx;
  cx.inst = shared_words + inst_id * ka.slots * SLOT_WORDS;
  cx.item = blockIdx.x * ka.G + inst_id;
  cx.live = inst_id < ka.G && cx.item < n_items;
  if (ka.item_index && cx.live) cx.item = ka.item_index[cx.item];
  if (ka.hwid_out && lane == 0) { ka.hwid_out[5 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); ka.hwid_out[5 * blockIdx.x + 1] = __builtin_readcyclecounter(); ka.hwid_out[5 * blockIdx.x + 3] = wall_clock64(); }   // HW_ID, XCC_ID, start tick (s_memtime), start time (s_memrealtime, 100 MHz): placement study
  __syncthreads();   // single wave: orders the constant fill before first use
  // Software-pipelined interpreter loop: this lane's descriptor words for step s+1 and the header of step s+2 are
  // requested before step s executes, so their L2 latency overlaps the arithmetic.
  const uint4* descs4 = (const uint4*)ka.descs;
  Step st = ka.steps[0];
  Step nst = ka.steps[ka.nsteps > 1 ? 1 : 0];
  uint4 d0 = make_uint4(0, 0, 0, 0), d1 = make_uint4(0, 0, 0, 0);
  if (lane_in < st.nlanes) {
    const u32 o = (st.desc_off + lane_in * st.stride) >> 2;
    d0 = descs4[o];
    if (st.stride > 4) d1 = descs4[o + 1];
  }
  // Fairness between the wavefronts that share a SIMD: the issue arbiter prefers the oldest wavefront, which then runs at ~94 % of
  // its lone speed while a second one gets ~55 % and a third ~23 % (tools/placement.py), so the youngest finishes long after the
  // others and runs the tail alone.  Lowering the own priority with progress lets the wavefront that is behind catch up: +8..10 %
  // for launches that put 2-4 wavefronts on every SIMD in a single round (8192..16384 pairings).  The launcher picks the FAIR instantiation only
  // for those: with one wavefront per SIMD there is nothing to balance, and in steady state (many rounds, or several batches in
  // flight) the priorities cost ~2 %.
  // Two instantiations: even a never-taken priority test in this loop costs a lone wavefront 5 % (measured), so the variant without
  // the priority code is a kernel of its own.
  const u32 quarter = (ka.nsteps >> 2) + 1;
  if (FAIR) __builtin_amdgcn_s_setprio(3);
  for (u32 s = 0; s < ka.nsteps; s++) {
    if (FAIR) { if (s == quarter) __builtin_amdgcn_s_setprio(2); else if (s == 3 * quarter) __builtin_amdgcn_s_setprio(1); }
    // header of step s+2 is requested now and first looked at one iteration later; the header of step s+1 arrived during
    // the previous step, so the descriptor prefetch below does not wait on global memory (a lone wavefront has nobody to
    // hide a ~2 us header round trip per step behind)
    const u32 sn = (s + 2 < ka.nsteps) ? s + 2 : ka.nsteps - 1;
    const Step nnst = ka.steps[sn];
    uint4 n0 = make_uint4(0, 0, 0, 0), n1 = make_uint4(0, 0, 0, 0);
    if (lane_in < nst.nlanes) {
      const u32 o = (nst.desc_off + lane_in * nst.stride) >> 2;
      n0 = descs4[o];
      if (nst.stride > 4) n1 = descs4[o + 1];
    }
    if (lane_in < st.nlanes) {
      u32 d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
      u32 res[NL];
      u32 dst = exec_lane(st, d, ka.descs + st.desc_off + lane_in * st.stride, smem, cx, ka.bufs, res);
      if (dst != 0xffffffffu) {
#pragma unroll
        for (int i = 0; i < NL; i++) smem[dst + i] = res[i];
      }
    }
    st = nst; nst = nnst; d0 = n0; d1 = n1;
  }
  if (ka.hwid_out && lane == 0) { ka.hwid_out[5 * blockIdx.x + 2] = __builtin_readcyclecounter(); ka.hwid_out[5 * blockIdx.x + 4] = wall_clock64(); }
}
extern "C" __global__ void __launch_bounds__(64) nbls_vm_kernel(KernelArgs ka) { vm_kernel_body<false>(ka); }
extern "C" __global__ void __launch_bounds__(64) nbls_vm_kernel_fair(KernelArgs ka) { vm_kernel_body<true>(ka); }

// Two-wave variant for small batches.  A lone wavefront per SIMD issues at ~1/3 of the VALU rate (every instruction waits
// for the previous one; tools/ubench/lone_wave.hip) and a second wavefront on the same SIMD runs at full speed beside it,
// so when a launch has no more workgroups than the chip has SIMDs each workgroup gets TWO wavefronts that share the work
// of every K_DOT lane-op: wave 0 accumulates the first half of the products, wave 1 the second half; wave 1 hands its 28
// column accumulators over through LDS, wave 0 adds them, reduces once and finishes the lane-op.  Same instances, same
// slots, same results (integer sums in a different order); ~0.65x the instructions per wavefront.
extern "C" __global__ void __launch_bounds__(128) nbls_vm_kernel_split(KernelArgs ka) {
  extern __shared__ __attribute__((aligned(16))) u32 smem[];
  const u32 tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  u32 n_items = ka.n_items;
  if (ka.n_items_dev) { const u32 v = *ka.n_items_dev; n_items = v < n_items ? v : n_items; if (blockIdx.x * ka.G >= n_items) return; }
  const u32 shared_words = ka.nconst * SLOT_WORDS;
  for (u32 i = tid; i < shared_words; i += 128) smem[i] = ka.consts[i];
  const u32 W = ka.W;
  const u32 inst_id = lane / W;
  const u32 lane_in = (inst_id < ka.G) ? (lane - inst_id * W) : 0xffffu;
  LaneCtx cx;
  cx.inst = shared_words + inst_id * ka.slots * SLOT_WORDS;
  cx.item = blockIdx.x * ka.G + inst_id;
  cx.live = inst_id < ka.G && cx.item < n_items;
  if (ka.item_index && cx.live) cx.item = ka.item_index[cx.item];
  u64* xch = (u64*)(smem + (ka.nconst + ka.G * ka.slots) * SLOT_WORDS);   // exchange area: 28 columns x 64 lanes, column-major (conflict-free)
  __syncthreads();
  const uint4* descs4 = (const uint4*)ka.descs;
  Step st = ka.steps[0];
  Step nst = ka.steps[ka.nsteps > 1 ? 1 : 0];
  uint4 d0 = make_uint4(0, 0, 0, 0), d1 = make_uint4(0, 0, 0, 0);
  if (lane_in < st.nlanes) {
    const u32 o = (st.desc_off + lane_in * st.stride) >> 2;
    d0 = descs4[o];
    if (st.stride > 4) d1 = descs4[o + 1];
  }
  for (u32 s = 0; s < ka.nsteps; s++) {
    // header of step s+2 is requested now and first looked at one iteration later; the header of step s+1 arrived during
    // the previous step, so the descriptor prefetch below does not wait on global memory (a lone wavefront has nobody to
    // hide a ~2 us header round trip per step behind)
    const u32 sn = (s + 2 < ka.nsteps) ? s + 2 : ka.nsteps - 1;
    const Step nnst = ka.steps[sn];
    uint4 n0 = make_uint4(0, 0, 0, 0), n1 = make_uint4(0, 0, 0, 0);
    if (lane_in < nst.nlanes) {
      const u32 o = (nst.desc_off + lane_in * nst.stride) >> 2;
      n0 = descs4[o];
      if (nst.stride > 4) n1 = descs4[o + 1];
    }
    const bool active = lane_in < st.nlanes;
    u32 d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    const u32* gd = ka.descs + st.desc_off + lane_in * st.stride;
    if (st.kind == K_DOT && st.p0 >= 2) {        // uniform: split the product loop between the two waves
      const u32 h = (st.p0 + 1) / 2;
      u64 acc[2 * NL];
      if (active) {
        if (wave == 0) { acc_init(acc, d[0] >> 28); dot_products(acc, st, d, gd, smem, cx, 0, h); }
        else {
#pragma unroll
          for (int c = 0; c < 2 * NL; c++) acc[c] = 0;
          dot_products(acc, st, d, gd, smem, cx, h, st.p0);
#pragma unroll
          for (int c = 0; c < 2 * NL; c++) xch[c * 64 + lane] = acc[c];
        }
      }
      __syncthreads();
      if (active && wave == 0) {
#pragma unroll
        for (int c = 0; c < 2 * NL; c++) acc[c] += xch[c * 64 + lane];
        u32 res[NL];
        const u32 dst = dot_result(res, acc, true, st, d, smem, cx);
#pragma unroll
        for (int i = 0; i < NL; i++) smem[dst + i] = res[i];
      }
    } else if (active && wave == 0) {
      u32 res[NL];
      const u32 dst = exec_lane(st, d, gd, smem, cx, ka.bufs, res);
      if (dst != 0xffffffffu) {
#pragma unroll
        for (int i = 0; i < NL; i++) smem[dst + i] = res[i];
      }
    }
    __syncthreads();
    st = nst; nst = nnst; d0 = n0; d1 = n1;
  }
}

}  // namespace nbls

// host-side launcher (C linkage, used by nbls_api.cpp)
extern "C" int nbls_vm_launch(const nbls::KernelArgs* ka, unsigned lds_bytes, void* stream) {
  using namespace nbls;
  if (ka->n_items == 0) return 0;
  unsigned blocks = (ka->n_items + ka->G - 1) / ka->G;
  static bool attr_set = false;
  // NBLS_SPLIT: 0 = never, 1 = always, unset = for launches of at most 256 workgroups (one per CU): measured 18 % lower latency
  // there, break-even at 512 workgroups, a loss beyond (two co-resident wavefronts per CU contend for LDS)
  static const int split_mode = getenv("NBLS_SPLIT") ? atoi(getenv("NBLS_SPLIT")) : -1;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)nbls_vm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)nbls_vm_kernel_fair, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)nbls_vm_kernel_split, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  static const unsigned lds_floor = getenv("NBLS_LDS_FLOOR") ? (unsigned)atoi(getenv("NBLS_LDS_FLOOR")) : 0u;   // placement studies: caps workgroups per CU at 160 KB / floor
  if (lds_bytes < lds_floor) lds_bytes = lds_floor;
  const bool split = split_mode == 1 || (split_mode < 0 && blocks <= 256);
  if (split) hipLaunchKernelGGL(nbls_vm_kernel_split, dim3(blocks), dim3(128), lds_bytes + 2 * NLIMBS * 64 * 8, (hipStream_t)stream, *ka);
  else {
    static const int fair_mode = getenv("NBLS_FAIR") ? atoi(getenv("NBLS_FAIR")) : -1;   // 0 never, 1 always, unset: launches of 2..4 wavefronts per SIMD
    const bool fair = fair_mode >= 0 ? fair_mode != 0 : (blocks > 1024 && blocks <= 4096);
    if (fair) hipLaunchKernelGGL(nbls_vm_kernel_fair, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, *ka);
    else hipLaunchKernelGGL(nbls_vm_kernel, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, *ka);
  }
  return (int)hipGetLastError();
}
