// vm_sim.cpp -- host-side simulator of the wave VM.  TEST INFRASTRUCTURE: lets the CPU test-suite check the
// compiled step programs (scheduler, slot allocation, descriptors, limb arithmetic) against oracle/ without a
// GPU.  It is built into libnbls_sim.so, which only tests/ load; libnbls.so (the product) does not contain it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "programs.h"
#include "vm_exec.h"
#include "consts_gen.h"

using namespace nbls;

static void sim_run(const Program& p, unsigned n_items, const IOBuf* bufs) {
  const unsigned shared = p.nconst * 12 + 17 * 16;
  std::vector<u32> lds(lds_words(p.nconst, p.G, p.slots));
  unsigned blocks = (n_items + p.G - 1) / p.G;
  for (unsigned blk = 0; blk < blocks; blk++) {
    std::fill(lds.begin(), lds.end(), 0xdeadbeefu);
    memcpy(lds.data(), p.consts.data(), shared * 4);
    for (size_t s = 0; s < p.steps.size(); s++) {
      const Step& st = p.steps[s];
      struct Pending { u32 dst; u32 v[12]; };
      std::vector<Pending> pend;
      for (unsigned lane = 0; lane < 64; lane++) {
        unsigned inst = lane / p.W;
        if (inst >= p.G) continue;
        unsigned lane_in = lane - inst * p.W;
        if (lane_in >= st.nlanes) continue;
        LaneCtx cx; cx.pm2 = p.nconst * 12; cx.inst = shared + inst * p.slots * 12; cx.item = blk * p.G + inst; cx.live = cx.item < n_items;
        Pending pd;
        pd.dst = exec_lane(st, p.descs.data() + st.desc_off + lane_in * st.stride, lds.data(), cx, bufs, pd.v);
        if (pd.dst != 0xffffffffu) pend.push_back(pd);
      }
      for (auto& pd : pend) memcpy(&lds[pd.dst], pd.v, 48);
    }
  }
}

extern "C" {
// bufs: 8 pointers + 8 strides
__attribute__((visibility("default"))) int nbls_sim_run(int prog, unsigned n_items, uint8_t** ptrs, const uint64_t* strides) {
  if (prog < 0 || prog >= P_COUNT) return -1;
  IOBuf b[MAX_BUFS];
  for (int i = 0; i < MAX_BUFS; i++) { b[i].ptr = ptrs[i]; b[i].stride = strides[i]; }
  sim_run(get_program((ProgId)prog), n_items, b);
  return 0;
}
// out = in^(p-2) on raw Montgomery limbs (stands in for the fp inversion kernel)
__attribute__((visibility("default"))) void nbls_sim_fp_inv(unsigned n, const u32* in, u32* out) {
  const u32 P2[12] = NBLS_2P32;
  const u32 P1[12] = NBLS_P32;
  for (unsigned k = 0; k < n; k++) {
    u32 acc[12], x[12];
    memcpy(x, in + 12 * k, 48); memcpy(acc, NBLS_R1, 48);
    for (int i = NBLS_P_MINUS_2_BITS - 1; i >= 0; i--) {
      u32 t[12]; mont_mul12(t, acc, acc); csub<12>(t, P2); memcpy(acc, t, 48);
      if ((NBLS_EXP_P_MINUS_2[i >> 6] >> (i & 63)) & 1) { mont_mul12(t, acc, x); csub<12>(t, P2); memcpy(acc, t, 48); }
    }
    (void)P1;
    memcpy(out + 12 * k, acc, 48);
  }
}
__attribute__((visibility("default"))) void nbls_sim_stats() { for (int i = 0; i < P_COUNT; i++) print_stats(get_program((ProgId)i)); }
}
