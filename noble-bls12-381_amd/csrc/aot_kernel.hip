// aot_kernel.hip -- ahead-of-time specialised kernels of the hot step programs (aot.h): one kernel per program, its step loop a jump over the
// program's signature table (aot_sigs.inc, generated at build time); every signature is a straight-line body: product rounds unrolled, operand
// shapes / flags / post-added term counts compile-time constants, the arithmetic the interpreter's own (vm_exec.h: mac28, redc28, dot_finish).
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include "aot.h"
#include "vm_exec.h"
#include "aot_sigs.inc"

namespace nbls {

// One step with a compile-time header.  The descriptor words d0 / d1 / dr (header, post-added offsets, round 0) were fetched one step ahead.
template <u32 KIND, u32 P0, u32 P1, u32 LIN, u32 SH0, u32 SH1, u32 STRIDE>
__device__ __forceinline__ void aot_step(const u32 nlanes, const u32 desc_off, const u32 lane_in, const uint4 d0, const uint4 d1, const uint4 dr,
                                         const uint4* __restrict__ descs4, char* lds, const LaneCtx& cx, const KernelArgs& ka) {
  Step st;
  st.kind = (uint8_t)KIND; st.nlanes = (uint8_t)nlanes; st.p0 = (uint8_t)P0; st.p1 = (uint8_t)P1; st.desc_off = desc_off; st.stride = STRIDE; st.lin = LIN;
  st.shape[0] = SH0; st.shape[1] = SH1; st.rsv[0] = st.rsv[1] = 0;
  if (lane_in < nlanes) {
    const u32 d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    u32 res[NL];
    u32 dst;
    if constexpr (KIND == K_DOT) {
      u64 acc[2 * NL];
      dot_init(acc, st, d[0]);
      const uint4* gr = descs4 + ((desc_off + lane_in * STRIDE) >> 2) + 3;   // descriptor of round 1
      uint4 cur = dr;
#pragma unroll
      for (u32 r = 0; r < P0; r++) {
        uint4 nx = cur;
        if (r + 1 < P0) nx = gr[r];
        dot_round(acc, round_shape(st, r), round_signs(d[1], r), cur.x, cur.y, cur.z, cur.w, lds, cx);
        // the columns are made opaque between rounds: the optimiser would otherwise re-associate every column's sum over ALL rounds of the unrolled body
        // (every round's operands live at once: 330-480 registers); the scheduling barrier keeps the LDS reads of later rounds from being hoisted to the top
#pragma unroll
        for (int c = 0; c < 2 * NL - 1; c++) asm volatile("" : "+v"(acc[c]));
        __builtin_amdgcn_sched_barrier(0);
        cur = nx;
      }
      dst = dot_finish(res, acc, st, d, lds, cx, ka.qp_table);
    } else {
      dst = exec_lane(st, d, lds, cx, ka.bufs, res, ka.qp_table);
    }
    if (dst != 0xffffffffu) st14(lds, dst, res);
  }
}

template <class Dispatch>
__device__ __forceinline__ void aot_body(const KernelArgs& ka, Dispatch&& dispatch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds = smem;
  const u32 lane = threadIdx.x;
  u32 n_items = ka.n_items;
  if (ka.n_items_dev) { const u32 v = *ka.n_items_dev; n_items = v < n_items ? v : n_items; if (blockIdx.x * ka.G >= n_items) return; }
  // constants: replicated at the start of every instance region (the listed programs run five or six items per wavefront)
  const u32 per_inst = ka.nconst * NL;
  for (u32 i = lane; i < ka.G * per_inst; i += 64) {
    const u32 g = i / per_inst, r = i - g * per_inst, c = r / NL, l = r - c * NL;
    *(u32*)(lds + g * ka.inst_bytes + c * ka.slot_bytes + 4 * l) = ka.consts[c * RAW_WORDS + l];
  }
  const u32 W = ka.W, inst_id = lane / W;
  const u32 lane_in = (inst_id < ka.G) ? (lane - inst_id * W) : 0xffffu;
  LaneCtx cx;
  cx.shared = false;
  cx.inst = (inst_id < ka.G ? inst_id : 0) * ka.inst_bytes;
  cx.item = blockIdx.x * ka.G + inst_id;
  cx.live = inst_id < ka.G && cx.item < n_items;
  if (ka.item_index && cx.live) cx.item = ka.item_index[cx.item];
  __syncthreads();   // single wave: orders the constant fill before first use
  const uint4* descs4 = (const uint4*)ka.descs;
  const uint2* as = (const uint2*)ka.aot_steps;
  const u32 nsteps = ka.nsteps;
  uint2 h = as[0], nh = as[nsteps > 1 ? 1 : 0];
  auto desc_at = [&](const uint2& x) { const u32 nl = (x.x >> 8) & 0xffu, idx = lane_in < nl ? lane_in : 0u; return (x.y + idx * (x.x >> 16)) >> 2; };
  uint4 d0, d1, dr;
  { const u32 o = desc_at(h); d0 = descs4[o]; d1 = descs4[o + 1]; dr = descs4[o + 2]; }
  const u32 quarter = (nsteps >> 2) + 1;
  const bool fair = ka.fair != 0;
  if (fair) __builtin_amdgcn_s_setprio(3);
  for (u32 s = 0; s < nsteps; s++) {
    if (fair) { if (s == quarter) __builtin_amdgcn_s_setprio(2); else if (s == 3 * quarter) __builtin_amdgcn_s_setprio(1); }
    const uint2 nnh = as[(s + 2 < nsteps) ? s + 2 : nsteps - 1];
    uint4 n0, n1, nr;
    { const u32 o = desc_at(nh); n0 = descs4[o]; n1 = descs4[o + 1]; nr = descs4[o + 2]; }
    dispatch(h.x & 0xffu, (h.x >> 8) & 0xffu, h.y, lane_in, d0, d1, dr, descs4, lds, cx);
    h = nh; nh = nnh; d0 = n0; d1 = n1; dr = nr;
  }
}

// register budget: 168 VGPRs = three wavefronts per SIMD (the straight-line bodies would otherwise be scheduled with every LDS read hoisted: 220-300 registers)
#if !defined(NBLS_AOT_WAVES)
#define NBLS_AOT_WAVES 3
#endif
#define NBLS_AOT_OCC __attribute__((amdgpu_waves_per_eu(NBLS_AOT_WAVES, NBLS_AOT_WAVES)))
#define AOT_CASE(ID, KIND, P0, P1, LIN, SH0, SH1, STRIDE, CNT) \
  case ID: aot_step<KIND, P0, P1, LIN, SH0, SH1, STRIDE>(nlanes, desc_off, lane_in, d0, d1, dr, descs4, lds, cx, ka); break;
#define AOT_KERNEL(NAME, PID)                                                                                                                   \
  extern "C" __global__ void __launch_bounds__(64) NBLS_AOT_OCC nbls_aot_##NAME(KernelArgs ka) {                                                           \
    aot_body(ka, [&](u32 sig, u32 nlanes, u32 desc_off, u32 lane_in, const uint4& d0, const uint4& d1, const uint4& dr, const uint4* descs4, \
                     char* lds, const LaneCtx& cx) {                                                                                            \
      switch (sig) { AOT_SIGS_##NAME(AOT_CASE) default: break; }                                                                                \
    });                                                                                                                                         \
  }
NBLS_AOT_PROGRAMS(AOT_KERNEL)

// ---- host side
#define AOT_ROW(ID, KIND, P0, P1, LIN, SH0, SH1, STRIDE, CNT) {KIND, P0, P1, LIN, SH0, SH1, STRIDE},
#define AOT_TABLE(NAME, PID) static const AotSig sigs_##NAME[] = {AOT_SIGS_##NAME(AOT_ROW)};
NBLS_AOT_PROGRAMS(AOT_TABLE)
struct AotKernel { int prog_id; const void* fn; const AotSig* sigs; unsigned nsigs; };
#define AOT_ENTRY(NAME, PID) {(int)PID, (const void*)nbls_aot_##NAME, sigs_##NAME, (unsigned)(sizeof(sigs_##NAME) / sizeof(AotSig))},
static const AotKernel g_kernels[] = {NBLS_AOT_PROGRAMS(AOT_ENTRY)};
static const int g_nkernels = (int)(sizeof(g_kernels) / sizeof(g_kernels[0]));

}  // namespace nbls

extern "C" int nbls_aot_index(int prog_id) {
  for (int k = 0; k < nbls::g_nkernels; k++) if (nbls::g_kernels[k].prog_id == prog_id) return k;
  return -1;
}
extern "C" int nbls_aot_translate(int k, const nbls::Step* steps, unsigned nsteps, nbls::AotStep* out) {
  using namespace nbls;
  if (k < 0 || k >= g_nkernels) return -1;
  const AotKernel& K = g_kernels[k];
  for (unsigned s = 0; s < nsteps; s++) {
    const AotSig sg = aot_sig_of(steps[s]);
    unsigned id = 0;
    while (id < K.nsigs && !(K.sigs[id] == sg)) id++;
    if (id == K.nsigs || steps[s].stride >= 65536u) return -1;
    out[s].x = id | ((uint32_t)steps[s].nlanes << 8) | (steps[s].stride << 16);
    out[s].y = steps[s].desc_off;
  }
  return 0;
}
extern "C" int nbls_aot_launch(int k, const nbls::KernelArgs* ka, unsigned lds_bytes, void* stream) {
  using namespace nbls;
  if (k < 0 || k >= g_nkernels) return -1;
  if (ka->n_items == 0) return 0;
  const unsigned blocks = (ka->n_items + ka->G - 1) / ka->G;
  static std::atomic<bool> attr_set[64];
  static std::mutex attr_mu;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
      std::lock_guard<std::mutex> g(attr_mu);
      if (!attr_set[dev].load(std::memory_order_relaxed)) {
        for (int i = 0; i < g_nkernels; i++) (void)hipFuncSetAttribute(g_kernels[i].fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set[dev].store(true, std::memory_order_release);
      }
    }
  }
  KernelArgs a = *ka;
  a.fair = (blocks > 1024 && blocks <= 4096) ? 1u : 0u;   // launches of 2..4 wavefronts per SIMD (vm_kernel.hip)
  void* args[] = {&a};
  const hipError_t e = hipLaunchKernel(g_kernels[k].fn, dim3(blocks), dim3(64), args, lds_bytes, (hipStream_t)stream);
  return e == hipSuccess ? (int)hipGetLastError() : (int)e;
}
