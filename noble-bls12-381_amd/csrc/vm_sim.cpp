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
#include "fp_inv.h"

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
        u32 dw[8] = {0};
        const u32* gd = p.descs.data() + st.desc_off + lane_in * st.stride;
        memcpy(dw, gd, (st.stride < 8 ? st.stride : 8) * 4);
        pd.dst = exec_lane(st, dw, gd, lds.data(), cx, bufs, pd.v);
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
// out = in^-1 on raw Montgomery limbs: the same fp_mont_inverse routine the inversion kernel runs per lane
__attribute__((visibility("default"))) void nbls_sim_fp_inv(unsigned n, const u32* in, u32* out) {
  static std::vector<u32> table;
  if (table.empty()) { table.resize(382 * 12); make_inv_table(table.data()); }
  for (unsigned k = 0; k < n; k++) fp_mont_inverse(out + 12 * k, in + 12 * k, table.data());
}
// out = in^e on raw Montgomery limbs; which: 0 = (p+1)/4 on Fp, 1 = (p^2+7)/16 on Fp2, 2 = (p^2-9)/16 on Fp2 (stand-ins for the pow kernels)
static void mmh(u32* r, const u32* a, const u32* b) { const u32 P2[12] = NBLS_2P32; u32 t[12]; mont_mul12(t, a, b); csub<12>(t, P2); memcpy(r, t, 48); }
static void addh(u32* r, const u32* a, const u32* b) { const u32 P2[12] = NBLS_2P32; u32 t[12], c = 0; for (int i = 0; i < 12; i++) t[i] = addc(a[i], b[i], c, &c); csub<12>(t, P2); memcpy(r, t, 48); }
static void subh(u32* r, const u32* a, const u32* b) { const u32 P2[12] = NBLS_2P32; u32 t[12], br = 0, c = 0; for (int i = 0; i < 12; i++) t[i] = subb(a[i], b[i], br, &br); for (int i = 0; i < 12; i++) t[i] = addc(t[i], P2[i], c, &c); csub<12>(t, P2); memcpy(r, t, 48); }
static void fp2mulh(u32* r, const u32* a, const u32* b) {
  u32 t1[12], t2[12], s1[12], s2[12], m[12];
  mmh(t1, a, b); mmh(t2, a + 12, b + 12); addh(s1, a, a + 12); addh(s2, b, b + 12); mmh(m, s1, s2);
  u32 r0[12], r1[12]; subh(r0, t1, t2); subh(m, m, t1); subh(r1, m, t2); memcpy(r, r0, 48); memcpy(r + 12, r1, 48);
}
__attribute__((visibility("default"))) void nbls_sim_fp_pow(unsigned n, const u32* in, u32* out, int which) {
  const uint64_t* e = which == 0 ? NBLS_EXP_P_PLUS_1_DIV_4 : which == 1 ? NBLS_EXP_P2_PLUS_7_DIV_16 : NBLS_EXP_P2_MINUS_9_DIV_16;
  int bits = which == 0 ? NBLS_P_PLUS_1_DIV_4_BITS : which == 1 ? NBLS_P2_PLUS_7_DIV_16_BITS : NBLS_P2_MINUS_9_DIV_16_BITS;
  for (unsigned k = 0; k < n; k++) {
    if (which == 0) {
      u32 acc[12]; memcpy(acc, NBLS_R1, 48);
      for (int i = bits - 1; i >= 0; i--) { mmh(acc, acc, acc); if ((e[i >> 6] >> (i & 63)) & 1) mmh(acc, acc, in + 12 * k); }
      memcpy(out + 12 * k, acc, 48);
    } else {
      u32 acc[24] = {0}; memcpy(acc, NBLS_R1, 48);
      for (int i = bits - 1; i >= 0; i--) { fp2mulh(acc, acc, acc); if ((e[i >> 6] >> (i & 63)) & 1) fp2mulh(acc, acc, in + 24 * k); }
      memcpy(out + 24 * k, acc, 96);
    }
  }
}
__attribute__((visibility("default"))) void nbls_sim_stats() { for (int i = 0; i < P_COUNT; i++) print_stats(get_program((ProgId)i)); }
}
