// vm_kernel.hip -- the gfx950 wave-VM kernel: one wavefront (= one workgroup of 64 lanes) interprets a compiled
// step list for G work items at once; see vm.h / vm_exec.h.  Integer VALU work (v_mad_i64_i32 into lazy 64-bit columns):
// no MFMA by construction (independent 381-bit products are not a dense contraction).
//
// Control flow is scalar throughout: the step header (kind, product rounds, operand shapes, post-added terms) is uniform
// for the wavefront and is read with scalar loads, so the product loop contains no per-lane flag test and no exec-mask
// juggling -- a round is: four address additions, the LDS reads, the limb-wise combine the round's shape asks for, 196
// multiply-adds.  Per-lane data (LDS byte offsets) comes from the lane descriptors, fetched one round / one step ahead.
#include <hip/hip_runtime.h>
#include "config.h"
#include <cstdlib>
#include <atomic>
#include <mutex>
#include "vm_exec.h"

namespace nbls {

struct LaneSetup { u32 lane_in; LaneCtx cx; };

__device__ __forceinline__ u32 kernel_prologue(const KernelArgs& ka, char* lds, u32 tid, u32 nthreads, u32 lane, LaneSetup& ls, bool& exit_now) {
  u32 n_items = ka.n_items;
  exit_now = false;
  if (ka.n_items_dev) { const u32 v = *ka.n_items_dev; n_items = v < n_items ? v : n_items; if (blockIdx.x * ka.G >= n_items) { exit_now = true; return 0; } }
  // constants: replicated at the start of every instance region, or one shared copy at the start of the LDS image
  const u32 per_inst = ka.nconst * NL, copies = ka.shared_consts ? 1u : ka.G;
  for (u32 i = tid; i < copies * per_inst; i += nthreads) {
    const u32 g = i / per_inst, r = i - g * per_inst, c = r / NL, l = r - c * NL;
    *(u32*)(lds + g * ka.inst_bytes + c * ka.slot_bytes + 4 * l) = ka.consts[c * RAW_WORDS + l];
  }
  const u32 W = ka.W;
  const u32 inst_id = lane / W;
  ls.lane_in = (inst_id < ka.G) ? (lane - inst_id * W) : 0xffffu;
  ls.cx.shared = ka.shared_consts != 0;
  ls.cx.inst = (ka.shared_consts ? ka.nconst * ka.slot_bytes - 2u : 0u) + (inst_id < ka.G ? inst_id : 0) * ka.inst_bytes;   // shared constants: base - 2 (term_addr)
  ls.cx.item = blockIdx.x * ka.G + inst_id;
  ls.cx.live = inst_id < ka.G && ls.cx.item < n_items;
  if (ka.item_index && ls.cx.live) ls.cx.item = ka.item_index[ls.cx.item];
  return n_items;
}

// Lane split (LS = 4): sum of the 28 column accumulators over the four adjacent lanes of a lane-op, result in the first of them.  Two DPP stages
// (lane i += lane i + 1, then lane i += lane i + 2, inside rows of 16 lanes; groups of four never straddle a row).
__device__ __forceinline__ u32 dpp_shl1(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true); }   // row_shl:1
__device__ __forceinline__ u32 dpp_shl2(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x102, 0xf, 0xf, true); }   // row_shl:2
__device__ __forceinline__ void acc_sum4(u64* acc) {
#pragma unroll
  for (int c = 0; c < 2 * NL - 1; c++) {
    u64 v = acc[c];
    v += ((u64)dpp_shl1((u32)(v >> 32)) << 32) | dpp_shl1((u32)v);
    v += ((u64)dpp_shl2((u32)(v >> 32)) << 32) | dpp_shl2((u32)v);
    acc[c] = v;
  }
}

template <bool FAIR, bool SHARED, int LS = 1> __device__ __forceinline__ void vm_kernel_body(const KernelArgs& ka) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds = smem;
  const u32 lane = threadIdx.x;
  LaneSetup ls; bool exit_now;
  kernel_prologue(ka, lds, lane, 64, lane, ls, exit_now);
  if (exit_now) return;
  const u32 lane_in = ls.lane_in;
  LaneCtx cxv = ls.cx; cxv.shared = SHARED;   // compile-time: the address arithmetic of every operand depends on it
  const LaneCtx cx = cxv;
  if (ka.hwid_out && lane == 0) { ka.hwid_out[5 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); ka.hwid_out[5 * blockIdx.x + 1] = __builtin_readcyclecounter(); ka.hwid_out[5 * blockIdx.x + 3] = wall_clock64(); }   // HW_ID, XCC_ID, start tick (s_memtime), start time (s_memrealtime, 100 MHz): placement study
  __syncthreads();   // single wave: orders the constant fill before first use
  // Software-pipelined interpreter loop: the header of step s+2 (scalar), this lane's descriptor header and first product round of
  // step s+1 (vector) are requested before step s executes, so their latency overlaps the arithmetic.
  const uint4* descs4 = (const uint4*)ka.descs;
  Step st = ka.steps[0];
  Step nst = ka.steps[ka.nsteps > 1 ? 1 : 0];
  // Every lane reads 12 descriptor words whatever the step's stride (the descriptor stream is padded by 8 words) and idle lanes read lane 0's:
  // no conditional loads, no zero-filled registers (12 instructions per step).
  // lane split: a K_DOT lane-op has one descriptor per sub-lane (index = the physical lane), every other kind one per logical lane, executed by sub-lane 0
  const u32 lg = LS > 1 ? lane_in / (u32)LS : lane_in, sub = LS > 1 ? lane_in % (u32)LS : 0u;
  auto desc_index = [&](const Step& x) { const u32 idx = (LS > 1 && x.kind == K_DOT) ? lane_in : lg, cnt = (LS > 1 && x.kind == K_DOT) ? (u32)x.nlanes * LS : (u32)x.nlanes; return idx < cnt ? idx : 0u; };
  uint4 d0, d1, dr;
  {
    const u32 o = (st.desc_off + desc_index(st) * st.stride) >> 2;
    d0 = descs4[o]; d1 = descs4[o + 1]; dr = descs4[o + 2];
  }
  // Fairness between the wavefronts that share a SIMD: the issue arbiter prefers the oldest wavefront, which then runs at ~94 % of
  // its lone speed while a second one gets ~55 % and a third ~23 % (tools/placement.py), so the youngest finishes long after the
  // others and runs the tail alone.  Lowering the own priority with progress lets the wavefront that is behind catch up: +8..10 %
  // for launches that put 2-4 wavefronts on every SIMD in a single round (8192..16384 pairings).  The launcher picks the FAIR instantiation only
  // for those: with one wavefront per SIMD there is nothing to balance, and in steady state (many rounds, or several batches in
  // flight) the priorities cost ~2 %.
  const u32 quarter = (ka.nsteps >> 2) + 1;
  if (FAIR) __builtin_amdgcn_s_setprio(3);
  for (u32 s = 0; s < ka.nsteps; s++) {
    if (FAIR) { if (s == quarter) __builtin_amdgcn_s_setprio(2); else if (s == 3 * quarter) __builtin_amdgcn_s_setprio(1); }
    const u32 sn = (s + 2 < ka.nsteps) ? s + 2 : ka.nsteps - 1;
    const Step nnst = ka.steps[sn];
    uint4 n0, n1, nr;
    {
      const u32 o = (nst.desc_off + desc_index(nst) * nst.stride) >> 2;
      n0 = descs4[o]; n1 = descs4[o + 1]; nr = descs4[o + 2];
    }
    if (lg < st.nlanes && (LS == 1 || st.kind == K_DOT || sub == 0)) {
      const u32 d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
      u32 res[NL];
      u32 dst;
      if (st.kind == K_DOT) {   // uniform
        u64 acc[2 * NL];
        dot_init(acc, st, d[0]);
        const uint4* gr = descs4 + ((st.desc_off + lane_in * st.stride) >> 2) + 3;   // descriptor of round 1
        uint4 cur = dr;
        for (u32 r = 0; r < st.p0; r++) {   // uniform trip count; the next round's offsets are fetched ahead
          uint4 nx = cur;
          if (r + 1 < st.p0) nx = gr[r];
          dot_round(acc, round_shape(st, r), round_signs(d[1], r), cur.x, cur.y, cur.z, cur.w, lds, cx);
          cur = nx;
        }
        if (LS > 1) acc_sum4(acc);   // all sub-lanes of the lane-op are active here (lg < nlanes holds for the four of them alike)
        dst = (LS == 1 || sub == 0) ? dot_finish(res, acc, st, d, lds, cx, ka.qp_table) : 0xffffffffu;
      } else {
        dst = exec_lane(st, d, lds, cx, ka.bufs, res, ka.qp_table);
      }
      if (dst != 0xffffffffu) st14(lds, dst, res);
    }
    st = nst; nst = nnst; d0 = n0; d1 = n1; dr = nr;
  }
  if (ka.hwid_out && lane == 0) { ka.hwid_out[5 * blockIdx.x + 2] = __builtin_readcyclecounter(); ka.hwid_out[5 * blockIdx.x + 4] = wall_clock64(); }
}
// NBLS_WAVES_PER_EU (build-time experiment, tools/exp_variants.sh): cap the VGPR budget so that N wavefronts fit a SIMD (4 -> 128 registers, a few
// descriptor registers spill to scratch once per step)
#if !defined(NBLS_WAVES_PER_EU)
#define NBLS_WAVES_PER_EU 0
#endif
#if NBLS_WAVES_PER_EU > 0
#define NBLS_OCC __attribute__((amdgpu_waves_per_eu(NBLS_WAVES_PER_EU, NBLS_WAVES_PER_EU)))
#else
#define NBLS_OCC
#endif
// four instantiations: plain / fair (priority schedule), replicated / shared constants (_sc)
extern "C" __global__ void __launch_bounds__(64) NBLS_OCC nbls_vm_kernel(KernelArgs ka) { vm_kernel_body<false, false>(ka); }
extern "C" __global__ void __launch_bounds__(64) NBLS_OCC nbls_vm_kernel_fair(KernelArgs ka) { vm_kernel_body<true, false>(ka); }
extern "C" __global__ void __launch_bounds__(64) NBLS_OCC nbls_vm_kernel_sc(KernelArgs ka) { vm_kernel_body<false, true>(ka); }
extern "C" __global__ void __launch_bounds__(64) NBLS_OCC nbls_vm_kernel_fair_sc(KernelArgs ka) { vm_kernel_body<true, true>(ka); }
// latency variant: lane-split programs (Program::lsplit = 4), launches of at most one wavefront per SIMD
extern "C" __global__ void __launch_bounds__(64) nbls_vm_kernel_ls4(KernelArgs ka) { vm_kernel_body<false, false, 4>(ka); }

}  // namespace nbls

// host-side launcher (C linkage, used by runtime.cpp)
extern "C" int nbls_vm_launch(const nbls::KernelArgs* ka, unsigned lds_bytes, void* stream) {
  using namespace nbls;
  if (ka->n_items == 0) return 0;
  unsigned blocks = (ka->n_items + ka->G - 1) / ka->G;
  // the dynamic-LDS limit is a per-device function attribute: set it once on every device a launch is made on.  The common case (already set) takes no
  // lock: twelve host threads launch through here concurrently when twelve calls are kept in flight
  static std::atomic<bool> attr_set[64];
  static std::mutex attr_mu;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
      std::lock_guard<std::mutex> g(attr_mu);
      if (!attr_set[dev].load(std::memory_order_relaxed)) {
        hipFuncSetAttribute((const void*)nbls_vm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)nbls_vm_kernel_fair, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)nbls_vm_kernel_sc, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)nbls_vm_kernel_ls4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)nbls_vm_kernel_fair_sc, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[dev].store(true, std::memory_order_release);
      }
    }
  }
  static const unsigned lds_floor = (unsigned)env_long("NBLS_LDS_FLOOR", 0);   // placement studies: caps workgroups per CU at 160 KB / floor
  if (lds_bytes < lds_floor) lds_bytes = lds_floor;
  if (ka->lsplit == 4) {
    if (ka->shared_consts) return -1;   // lane-split programs are compiled with replicated constants only
    hipLaunchKernelGGL(nbls_vm_kernel_ls4, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, *ka);
  } else {
    static const int fair_mode = (int)env_long("NBLS_FAIR", -1);   // 0 never, 1 always, unset: launches of 2..4 wavefronts per SIMD
    const bool fair = fair_mode >= 0 ? fair_mode != 0 : (blocks > 1024 && blocks <= 4096);
    if (ka->shared_consts) {
      if (fair) hipLaunchKernelGGL(nbls_vm_kernel_fair_sc, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, *ka);
      else hipLaunchKernelGGL(nbls_vm_kernel_sc, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, *ka);
    } else {
      if (fair) hipLaunchKernelGGL(nbls_vm_kernel_fair, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, *ka);
      else hipLaunchKernelGGL(nbls_vm_kernel, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, *ka);
    }
  }
  return (int)hipGetLastError();
}
