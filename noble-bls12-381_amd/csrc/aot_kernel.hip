// aot_kernel.hip -- ahead-of-time specialised kernels of the hot step programs (aot.h): one kernel per program, its step loop a jump over the
// program's signature table (aot_sigs.inc, generated at build time); every signature is a straight-line body (aot_exec.h): product rounds unrolled,
// operand shapes / flags / term counts compile-time constants, absolute LDS addresses, the K_DOT finish kept in the 64-bit columns.
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include "aot.h"
#include "aot_exec.h"
#include "aot_sigs.inc"

namespace nbls {

// descriptor access of one lane on the device: the first three 16-byte words were fetched one step ahead, the others are loaded where the body asks for them
struct DevDesc {
  V4 q0, q1, q2;
  const V4* blk;   // word w of this lane: blk[64 * w]
  __device__ __forceinline__ V4 quad(const u32 i) const { return i == 0 ? q0 : i == 1 ? q1 : i == 2 ? q2 : blk[64 * i]; }
};

template <class Dispatch>
__device__ __forceinline__ void aot_body(const AotArgs& ka, Dispatch&& dispatch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds = smem;
  const u32 lane = threadIdx.x;
  u32 n_items = ka.n_items;
  if (ka.n_items_dev) { const u32 v = *ka.n_items_dev; n_items = v < n_items ? v : n_items; if (blockIdx.x * ka.G >= n_items) return; }
  const u32 inst_id = lane / ka.W;
  u32 item = blockIdx.x * ka.G + inst_id;
  const bool live = inst_id < ka.G && item < n_items;
  if (ka.item_index && live) item = ka.item_index[item];
  const bool fair = ka.fair != 0;
  if (fair) __builtin_amdgcn_s_setprio(3);
  u32 total = 0, done = 0;
  for (u32 g = 0; g < ka.nseg; g++) total += ka.seg[g].nsteps;
  const u32 quarter = (total >> 2) + 1;
  for (u32 g = 0; g < ka.nseg; g++) {
    const AotSeg& sg = ka.seg[g];
    // a later segment of a chain reads, possibly on other lanes, scratch elements an earlier one stored: workgroup-scope release / acquire (one wavefront per
    // workgroup: the stores are complete and visible before the loads are issued); the LDS image is rebuilt for the segment's own layout
    if (g) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // constants: replicated at the start of every instance region, or one shared copy at the start of the LDS image
    const u32 per_inst = sg.nconst * NL, copies = sg.shared_consts ? 1u : ka.G;
    for (u32 i = lane; i < copies * per_inst; i += 64) {
      const u32 c0 = i / per_inst, r = i - c0 * per_inst, c = r / NL, l = r - c * NL;
      *(u32*)(lds + c0 * sg.inst_bytes + c * sg.slot_bytes + 4 * l) = sg.consts[c * RAW_WORDS + l];
    }
    __syncthreads();   // single wave: orders the constant fill before first use
    const V4* descs4 = (const V4*)sg.descs + lane;
    const uint2* as = (const uint2*)sg.steps;
    const u32 nsteps = sg.nsteps;
    uint2 h = as[0], nh = as[nsteps > 1 ? 1 : 0];
    DevDesc d;
    d.blk = descs4 + (size_t)h.y; d.q0 = d.blk[0]; d.q1 = d.blk[64]; d.q2 = d.blk[128];
    for (u32 s = 0; s < nsteps; s++, done++) {
      if (fair) { if (done == quarter) __builtin_amdgcn_s_setprio(2); else if (done == 3 * quarter) __builtin_amdgcn_s_setprio(1); }
      const uint2 nnh = as[(s + 2 < nsteps) ? s + 2 : nsteps - 1];
      DevDesc nd;
      nd.blk = descs4 + (size_t)nh.y; nd.q0 = nd.blk[0]; nd.q1 = nd.blk[64]; nd.q2 = nd.blk[128];   // the stream is padded: three words exist behind every block start
      dispatch(h.x & 0xffu, d, lds, item, live, sg.bufs);
      h = nh; nh = nnh; d = nd;
    }
  }
}

// register budget: 168 VGPRs = three wavefronts per SIMD
#if !defined(NBLS_AOT_WAVES)
#define NBLS_AOT_WAVES 3
#endif
#define NBLS_AOT_OCC __attribute__((amdgpu_waves_per_eu(NBLS_AOT_WAVES, NBLS_AOT_WAVES)))
#define AOT_CASE(ID, KIND, P0, FLAGS, T, SH0, SH1, CNT) \
  case ID: aot_step<KIND, P0, FLAGS, T, SH0, SH1>(d, lds, item, live, bufs, ka.qp_table, [&](u32 dst, const u32* res) __attribute__((always_inline)) { st14(lds, dst, res); }); break;
#define AOT_CASE_LS(ID, KIND, P0, FLAGS, T, SH0, SH1, CNT) \
  case ID: aot_step<KIND, P0, FLAGS, T, SH0, SH1, 4>(d, lds, item, live, bufs, ka.qp_table, [&](u32 dst, const u32* res) __attribute__((always_inline)) { st14(lds, dst, res); }); break;
#define AOT_CASE_LS2(ID, KIND, P0, FLAGS, T, SH0, SH1, CNT) \
  case ID: aot_step<KIND, P0, FLAGS, T, SH0, SH1, 2>(d, lds, item, live, bufs, ka.qp_table, [&](u32 dst, const u32* res) __attribute__((always_inline)) { st14(lds, dst, res); }); break;
// this translation unit is compiled NBLS_AOT_PARTS times (Makefile: -DNBLS_AOT_PART=i); every part declares all kernels and defines its own
#if !defined(NBLS_AOT_PART)
#define NBLS_AOT_PART 0
#endif
#define AOT_DECL(PART, NAME, P0, P1, P2, P3) extern "C" __global__ void nbls_aot_##NAME(AotArgs ka);
NBLS_AOT_KERNELS(AOT_DECL)
NBLS_AOT_LS_KERNELS(AOT_DECL)
NBLS_AOT_LS2_KERNELS(AOT_DECL)
#define AOT_KERNEL_BODY(NAME) AOT_KERNEL_BODY_(NAME, AOT_CASE, NBLS_AOT_OCC)
// (launched at one wavefront per SIMD at most, so no register budget to keep)
#define AOT_KERNEL_BODY_LS(NAME) AOT_KERNEL_BODY_(NAME, AOT_CASE_LS, __attribute__((amdgpu_waves_per_eu(1, 2))))   // lane-split programs: the columns of four adjacent lanes are summed before the reduction
#define AOT_KERNEL_BODY_LS2(NAME) AOT_KERNEL_BODY_(NAME, AOT_CASE_LS2, __attribute__((amdgpu_waves_per_eu(1, 2))))   // two-lane split (1025 .. 2048 items: one wavefront per SIMD at most)
#define AOT_KERNEL_BODY_(NAME, CASE, OCC)                                                                                                 \
  extern "C" __global__ void __launch_bounds__(64) OCC nbls_aot_##NAME(AotArgs ka) {                                           \
    aot_body(ka, [&](u32 sig, const DevDesc& d, char* lds, u32 item, bool live, const IOBuf* bufs) __attribute__((always_inline)) { \
      switch (sig) { AOT_SIGS_##NAME(CASE) default: break; }                                                                   \
    });                                                                                                                        \
  }
#include "aot_parts.inc"   // generated: AOT_KERNEL_BODY(name) for the kernels of this part

// ---- host side
#define AOT_ROW(ID, KIND, P0, FLAGS, T, SH0, SH1, CNT) {KIND, P0, FLAGS, T, SH0, SH1},
}  // namespace nbls

#if NBLS_AOT_PART == 0
namespace nbls {
#define AOT_TABLE(PART, NAME, P0, P1, P2, P3) static const AotSig sigs_##NAME[] = {AOT_SIGS_##NAME(AOT_ROW)};
NBLS_AOT_KERNELS(AOT_TABLE)
NBLS_AOT_LS_KERNELS(AOT_TABLE)
NBLS_AOT_LS2_KERNELS(AOT_TABLE)
struct AotKernel { int prog_id[4]; const void* fn; const AotSig* sigs; unsigned nsigs; const char* name; };
#define AOT_ENTRY(PART, NAME, P0, P1, P2, P3) {{(int)P0, (int)P1, (int)P2, (int)P3}, (const void*)nbls_aot_##NAME, sigs_##NAME, (unsigned)(sizeof(sigs_##NAME) / sizeof(AotSig)), "nbls_aot_" #NAME},
static const AotKernel g_kernels[] = {NBLS_AOT_KERNELS(AOT_ENTRY) NBLS_AOT_LS_KERNELS(AOT_ENTRY) NBLS_AOT_LS2_KERNELS(AOT_ENTRY)};
static const int g_nkernels = (int)(sizeof(g_kernels) / sizeof(g_kernels[0]));

}  // namespace nbls

extern "C" int nbls_aot_index(int prog_id) {
  if (prog_id < 0 || prog_id >= (int)nbls::P_COUNT) return -1;
  for (int k = 0; k < nbls::g_nkernels; k++) for (int j = 0; j < 4; j++) if (nbls::g_kernels[k].prog_id[j] == prog_id) return k;
  return -1;
}
extern "C" const char* nbls_aot_name(int k) { return k >= 0 && k < nbls::g_nkernels ? nbls::g_kernels[k].name : nullptr; }
extern "C" int nbls_aot_bind(int k, nbls::AotProgram* ap) {
  using namespace nbls;
  if (k < 0 || k >= g_nkernels || !ap) return -1;
  const AotKernel& K = g_kernels[k];
  std::vector<unsigned> map(ap->sigs.size());
  for (size_t i = 0; i < ap->sigs.size(); i++) {
    unsigned id = 0;
    while (id < K.nsigs && !(K.sigs[id] == ap->sigs[i])) id++;
    if (id == K.nsigs) return -1;
    map[i] = id;
  }
  for (auto& s : ap->steps) s.x = (s.x & ~0xffu) | map[s.x & 0xffu];
  return 0;
}
extern "C" int nbls_aot_launch(int k, const nbls::AotArgs* ka, unsigned lds_bytes, void* stream) {
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
  AotArgs a = *ka;
  a.fair = (blocks > 1024 && blocks <= 4096) ? 1u : 0u;   // launches of 2..4 wavefronts per SIMD (vm_kernel.hip)
  void* args[] = {&a};
  const hipError_t e = hipLaunchKernel(g_kernels[k].fn, dim3(blocks), dim3(64), args, lds_bytes, (hipStream_t)stream);
  return e == hipSuccess ? (int)hipGetLastError() : (int)e;
}
#endif   // NBLS_AOT_PART == 0
