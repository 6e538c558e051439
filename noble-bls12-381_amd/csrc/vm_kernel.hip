// vm_kernel.hip -- the gfx950 wave-VM kernel: one wavefront (= one workgroup of 64 lanes) interprets a compiled
// step list for G work items at once; see vm.h / vm_exec.h.  Integer VALU work (v_mad_i64_i32 into lazy 64-bit columns):
// no MFMA by construction (independent 381-bit products are not a dense contraction).
#include <hip/hip_runtime.h>
#include "vm_exec.h"

namespace nbls {

extern "C" __global__ void __launch_bounds__(64) nbls_vm_kernel(KernelArgs ka) {
  extern __shared__ __attribute__((aligned(16))) u32 smem[];
  const u32 lane = threadIdx.x;
  const u32 shared_words = ka.nconst * SLOT_WORDS;
  for (u32 i = lane; i < shared_words; i += 64) smem[i] = ka.consts[i];
  const u32 W = ka.W;
  const u32 inst_id = lane / W;
  const u32 lane_in = (inst_id < ka.G) ? (lane - inst_id * W) : 0xffffu;
  LaneCtx cx;
  cx.inst = shared_words + inst_id * ka.slots * SLOT_WORDS;
  cx.item = blockIdx.x * ka.G + inst_id;
  cx.live = inst_id < ka.G && cx.item < ka.n_items;
  __syncthreads();   // single wave: orders the constant fill before first use
  // Software-pipelined interpreter loop: the step header (scalar) and this lane's descriptor words for step s+1 are
  // requested before step s executes, so the L2 latency of the descriptor fetch overlaps the arithmetic.
  const uint4* descs4 = (const uint4*)ka.descs;
  Step st = ka.steps[0];
  uint4 d0 = make_uint4(0, 0, 0, 0), d1 = make_uint4(0, 0, 0, 0);
  if (lane_in < st.nlanes) {
    const u32 o = (st.desc_off + lane_in * st.stride) >> 2;
    d0 = descs4[o];
    if (st.stride > 4) d1 = descs4[o + 1];
  }
  for (u32 s = 0; s < ka.nsteps; s++) {
    const u32 sn = (s + 1 < ka.nsteps) ? s + 1 : s;
    const Step nst = ka.steps[sn];
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
    st = nst; d0 = n0; d1 = n1;
  }
}

}  // namespace nbls

// host-side launcher (C linkage, used by nbls_api.cpp)
extern "C" int nbls_vm_launch(const nbls::KernelArgs* ka, unsigned lds_bytes, void* stream) {
  using namespace nbls;
  if (ka->n_items == 0) return 0;
  unsigned blocks = (ka->n_items + ka->G - 1) / ka->G;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)nbls_vm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(nbls_vm_kernel, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, *ka);
  return (int)hipGetLastError();
}
