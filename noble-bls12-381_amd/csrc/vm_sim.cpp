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
#include "pow_exec.h"
#include "pow_wide.h"
#include "wide_exec.h"
#include "g1_wide.h"
#include "fp_inv_wide.h"
#include "scalar_split.h"
#include "aot_exec.h"
#include "aot_layout.h"
#include "aot_sigs.inc"

using namespace nbls;

static void sim_run(const Program& p, unsigned n_items, const IOBuf* bufs) {
  const u32* qp_table = qp_table_words();
  const unsigned inst_bytes = p.inst_bytes();
  std::vector<V4> lds4((size_t)p.lds_bytes() / 16 + 1);
  char* lds = (char*)lds4.data();
  unsigned blocks = (n_items + p.G - 1) / p.G;
  for (unsigned blk = 0; blk < blocks; blk++) {
    memset(lds, 0xde, (size_t)p.lds_bytes());
    for (unsigned g = 0; g < (p.shared_consts ? 1u : p.G); g++) for (unsigned c = 0; c < p.nconst; c++) memcpy(lds + g * inst_bytes + c * p.slot_bytes, p.consts.data() + c * RAW_WORDS, NL * 4);
    for (size_t s = 0; s < p.steps.size(); s++) {
      const Step& st = p.steps[s];
      struct Pending { u32 dst; u32 v[NL]; };
      std::vector<Pending> pend;
      for (unsigned lane = 0; lane < 64; lane++) {
        unsigned inst = lane / p.W;
        if (inst >= p.G) continue;
        unsigned lane_in = lane - inst * p.W;
        // lane split: lane_in is the physical lane; a K_DOT lane-op is the sum of its sub-lanes' accumulators (computed when sub-lane 0 is visited),
        // every other kind runs on sub-lane 0 with the descriptor of its logical lane
        const unsigned S = p.lsplit, lg = lane_in / S, sub = lane_in % S;
        if (lg >= st.nlanes || sub != 0) continue;
        LaneCtx cx; cx.inst = p.inst_base(inst) - (p.shared_consts ? 2u : 0u); cx.shared = p.shared_consts; cx.item = blk * p.G + inst; cx.live = cx.item < n_items;
        Pending pd;
        u32 dw[8] = {0};
        const u32* gd = p.descs.data() + st.desc_off + (st.kind == K_DOT ? lane_in : lg) * st.stride;
        memcpy(dw, gd, (st.stride < 8 ? st.stride : 8) * 4);
        if (st.kind == K_DOT) {
          u64 acc[2 * NL];
          dot_init(acc, st, dw[0]);
          for (u32 r = 0; r < st.p0; r++) { const u32* rd = gd + DOT_HDR_WORDS + DOT_ROUND_WORDS * r; dot_round(acc, round_shape(st, r), round_signs(gd[1], r), rd[0], rd[1], rd[2], rd[3], lds, cx); }
          for (unsigned j = 1; j < S; j++) {   // the other sub-lanes: their own descriptors, their own (bias-free) accumulators
            const u32* gj = gd + j * st.stride;
            u64 aj[2 * NL];
            dot_init(aj, st, gj[0]);
            for (u32 r = 0; r < st.p0; r++) { const u32* rd = gj + DOT_HDR_WORDS + DOT_ROUND_WORDS * r; dot_round(aj, round_shape(st, r), round_signs(gj[1], r), rd[0], rd[1], rd[2], rd[3], lds, cx); }
            for (int c = 0; c < 2 * NL; c++) acc[c] += aj[c];
          }
          pd.dst = dot_finish(pd.v, acc, st, dw, lds, cx, qp_table);
        } else pd.dst = exec_lane(st, dw, lds, cx, bufs, pd.v, qp_table);
        if (pd.dst != 0xffffffffu) pend.push_back(pd);
      }
      for (auto& pd : pend) st14(lds, pd.dst, pd.v);
    }
  }
}

// ---- translated programs (aot.h) on the host: the step bodies of the ahead-of-time kernels (aot_exec.h), every lane's results committed after all lanes of the
// step have read their operands (on the device the LDS operations of a wavefront execute in order)
struct HostDesc {
  const u32* blk; u32 lane;   // blk: word 0 of lane 0
  V4 quad(u32 i) const { const u32* q = blk + ((size_t)i * 64 + lane) * 4; return V4{q[0], q[1], q[2], q[3]}; }
};
struct Pend { u32 dst; u32 v[NL]; };
#define SIM_CASE(ID, KIND, P0, FLAGS, T, SH0, SH1, CNT) \
  case ID: aot_step<KIND, P0, FLAGS, T, SH0, SH1>(d, lds, item, live, bufs, qp, [&](u32 dst, const u32* res) { Pend pd; pd.dst = dst; memcpy(pd.v, res, NL * 4); pend.push_back(pd); }); break;
#define SIM_TABLE(PART, NAME, Q0, Q1, Q2, Q3)                                                                                                        \
  static const AotSig sim_sigs_##NAME[] = {AOT_SIGS_##NAME(SIM_ROW)};                                                                      \
  static void sim_step_##NAME(u32 sig, const HostDesc& d, char* lds, u32 item, bool live, const IOBuf* bufs, const u32* qp, std::vector<Pend>& pend) { \
    switch (sig) { AOT_SIGS_##NAME(SIM_CASE) default: abort(); }                                                                          \
  }
#define SIM_ROW(ID, KIND, P0, FLAGS, T, SH0, SH1, CNT) {KIND, P0, FLAGS, T, SH0, SH1},
NBLS_AOT_KERNELS(SIM_TABLE)
// lane-split kernels (aot.h NBLS_AOT_LS_KERNELS): the same bodies with LS = 4.  The device sums the columns of the four sub-lanes of a lane-op with two DPP stages
// (lane i <- v[i] + v[i+1] + v[i+2] + v[i+3] inside rows of 16 lanes); here the lanes of a step are visited from 63 down to 0, every lane leaves its columns in
// g_ls_cols before it takes the sum, so the three partners of a sub-lane 0 are already there.
static u64 g_ls_cols[64][2 * NL];
static unsigned g_ls_lane = 0, g_ls_width = 4;
static void sim_ls_sum(u64* acc, int nc) {
  const unsigned i = g_ls_lane;
  memcpy(g_ls_cols[i], acc, (size_t)nc * sizeof(u64));
  for (unsigned k = 1; k < g_ls_width; k++) if (i + k < 64 && (i + k) / 16 == i / 16) for (int c = 0; c < nc; c++) acc[c] += g_ls_cols[i + k][c];
}
#undef SIM_CASE
#define SIM_CASE(ID, KIND, P0, FLAGS, T, SH0, SH1, CNT) \
  case ID: aot_step<KIND, P0, FLAGS, T, SH0, SH1, 4>(d, lds, item, live, bufs, qp, [&](u32 dst, const u32* res) { Pend pd; pd.dst = dst; memcpy(pd.v, res, NL * 4); pend.push_back(pd); }); break;
NBLS_AOT_LS_KERNELS(SIM_TABLE)
#undef SIM_CASE
#define SIM_CASE(ID, KIND, P0, FLAGS, T, SH0, SH1, CNT) \
  case ID: aot_step<KIND, P0, FLAGS, T, SH0, SH1, 2>(d, lds, item, live, bufs, qp, [&](u32 dst, const u32* res) { Pend pd; pd.dst = dst; memcpy(pd.v, res, NL * 4); pend.push_back(pd); }); break;
NBLS_AOT_LS2_KERNELS(SIM_TABLE)
typedef void (*SimStepFn)(u32, const HostDesc&, char*, u32, bool, const IOBuf*, const u32*, std::vector<Pend>&);
struct SimKernel { int prog_id[4]; SimStepFn fn; const AotSig* sigs; unsigned nsigs; unsigned ls; };   // ls: sub-lanes per lane-op (0: none)
#define SIM_ENTRY(PART, NAME, Q0, Q1, Q2, Q3) {{(int)Q0, (int)Q1, (int)Q2, (int)Q3}, sim_step_##NAME, sim_sigs_##NAME, (unsigned)(sizeof(sim_sigs_##NAME) / sizeof(AotSig)), 0},
#define SIM_ENTRY_LS(PART, NAME, Q0, Q1, Q2, Q3) {{(int)Q0, (int)Q1, (int)Q2, (int)Q3}, sim_step_##NAME, sim_sigs_##NAME, (unsigned)(sizeof(sim_sigs_##NAME) / sizeof(AotSig)), 4},
#define SIM_ENTRY_LS2(PART, NAME, Q0, Q1, Q2, Q3) {{(int)Q0, (int)Q1, (int)Q2, (int)Q3}, sim_step_##NAME, sim_sigs_##NAME, (unsigned)(sizeof(sim_sigs_##NAME) / sizeof(AotSig)), 2},
static const SimKernel g_sim_kernels[] = {NBLS_AOT_KERNELS(SIM_ENTRY) NBLS_AOT_LS_KERNELS(SIM_ENTRY_LS) NBLS_AOT_LS2_KERNELS(SIM_ENTRY_LS2)};
static int g_sim_aot = 0;
// 0: ran; -2: the program has no ahead-of-time kernel; -3: its signatures are not in the kernel's table
static int sim_run_aot(int prog, unsigned n_items, const IOBuf* bufs) {
  const SimKernel* K = nullptr;
  if (prog >= 0 && prog < (int)P_COUNT) for (auto& k : g_sim_kernels) for (int j = 0; j < 4; j++) if (k.prog_id[j] == prog) K = &k;
  if (!K) return -2;
  const Program& p = get_program((ProgId)prog);
  AotProgram ap;
  if (!aot_translate(p, ap).empty()) return -3;
  std::vector<unsigned> map(ap.sigs.size());
  for (size_t i = 0; i < ap.sigs.size(); i++) { unsigned id = 0; while (id < K->nsigs && !(K->sigs[id] == ap.sigs[i])) id++; if (id == K->nsigs) return -3; map[i] = id; }
  const u32* qp_table = qp_table_words();
  std::vector<V4> lds4((size_t)ap.lds_bytes / 16 + 1);
  char* lds = (char*)lds4.data();
  const unsigned blocks = (n_items + p.G - 1) / p.G;
  for (unsigned blk = 0; blk < blocks; blk++) {
    memset(lds, 0xde, (size_t)ap.lds_bytes);
    for (unsigned g = 0; g < (p.shared_consts ? 1u : p.G); g++) for (unsigned c = 0; c < p.nconst; c++) memcpy(lds + g * p.inst_bytes() + c * p.slot_bytes, p.consts.data() + c * RAW_WORDS, NL * 4);
    for (size_t s = 0; s < ap.steps.size(); s++) {
      std::vector<Pend> pend;
      aot_sim_ls_hook() = K->ls ? sim_ls_sum : nullptr; g_ls_width = K->ls ? K->ls : 4;
      for (unsigned v = 0; v < 64; v++) {
        const unsigned lane = K->ls ? 63 - v : v;      // lane-split kernels: from the top down (sim_ls_sum)
        const unsigned inst = lane / p.W, item = blk * p.G + inst;
        HostDesc d; d.blk = ap.descs.data() + (size_t)ap.steps[s].y * 4; d.lane = lane;
        g_ls_lane = lane;
        K->fn(map[ap.steps[s].x & 0xffu], d, lds, item, inst < p.G && item < n_items, bufs, qp_table, pend);
      }
      for (auto& pd : pend) st14(lds, pd.dst, pd.v);
    }
  }
  return 0;
}

// The one-limb-per-lane form (pow_wide.h) on the host: a value of type U is the array of the 32 lanes of a wavefront's first two rows, every cross-lane move a loop.  The
// arithmetic the device does in 32 / 64 bits is checked here for what the device silently assumes: no 64-bit column overflows, the low-word additions never carry.
static unsigned long g_wide_violations = 0;
struct WideHost {
  struct U { u32 v[32]; };
  struct W { u64 v[32]; };
  const u32* in; u32* out; bool fp2; U tab[POW_TAB];
  template <class F> static U map(F f) { U r; for (int k = 0; k < 32; k++) r.v[k] = f(k); return r; }
  U konst(const u32* t16) const { return map([&](int k) { return t16[k & 15]; }); }
  U sel(const U& a, const U& b) const { return map([&](int k) { return k >= 16 ? b.v[k] : a.v[k]; }); }
  U add(const U& a, const U& b) const { return map([&](int k) { if ((u64)a.v[k] + b.v[k] > 0xffffffffull) g_wide_violations++; return a.v[k] + b.v[k]; }); }
  U sub(const U& a, const U& b) const { return map([&](int k) { if (a.v[k] < b.v[k]) g_wide_violations++; return a.v[k] - b.v[k]; }); }
  U and_(const U& a, u32 m) const { return map([&](int k) { return a.v[k] & m; }); }
  U shr(const U& a, int s) const { return map([&](int k) { return a.v[k] >> s; }); }
  U mul_lo(const U& a, u32 c) const { return map([&](int k) { return a.v[k] * c; }); }
  U lo(const W& w) const { return map([&](int k) { return (u32)w.v[k]; }); }
  W zero() const { W w; for (int k = 0; k < 32; k++) w.v[k] = 0; return w; }
  W mad(const U& a, const U& b, const W& acc) const { W w; for (int k = 0; k < 32; k++) { const unsigned __int128 t = (unsigned __int128)acc.v[k] + (unsigned __int128)a.v[k] * b.v[k]; if (t >> 64) g_wide_violations++; w.v[k] = (u64)t; } return w; }
  W mad_s(const U& a, u32 s, const W& acc) const { W w; for (int k = 0; k < 32; k++) { const unsigned __int128 t = (unsigned __int128)acc.v[k] + (unsigned __int128)a.v[k] * s; if (t >> 64) g_wide_violations++; w.v[k] = (u64)t; } return w; }
  W shr28(const W& a) const { W w; for (int k = 0; k < 32; k++) w.v[k] = a.v[k] >> 28; return w; }
  W add_lo(const W& a, const U& x) const { W w; for (int k = 0; k < 32; k++) { if ((a.v[k] >> 32) || (a.v[k] + x.v[k]) >> 32) g_wide_violations++; w.v[k] = (a.v[k] & 0xffffffff00000000ull) | (u32)((u32)a.v[k] + x.v[k]); } return w; }
  U bcast(const U& a, int i) const { return map([&](int k) { return a.v[(k & 16) | i]; }); }
  U xchg(const U& a) const { return map([&](int k) { return a.v[k ^ 16]; }); }
  U shl1(const U& a) const { return map([&](int k) { return (k & 15) == 15 ? 0u : a.v[k + 1]; }); }
  U shr1(const U& a) const { return map([&](int k) { return (k & 15) == 0 ? 0u : a.v[k - 1]; }); }
  u32 lane_of(const U& a, int k) const { return a.v[k]; }
  U load() const { return map([&](int k) { return (k & 15) < NL && (fp2 || k < 16) ? in[k] : 0u; }); }      // in: the element's 16 (Fp) or 32 (Fp2) words
  void store(const U& a) { for (int k = 0; k < (fp2 ? 32 : 16); k++) out[k] = (k & 15) < NL ? a.v[k] : 0u; }
  void tab_put(int e, const U& a) { tab[e] = a; }
  U tab_get(int e) const { return tab[e]; }
};
// ---- the one-limb-per-lane interpreter (wide_exec.h, vm_wide_kernel.hip) on the host: a value is the array of a row's sixteen lanes; every lane-op of a step reads,
// then all results are committed (the device's two barriers).  The 32 / 64-bit assumptions of the device code are checked: g_wide_violations counts them.
struct WideRowHost {
  struct I { i32 v[16]; };
  struct W { i64 v[16]; };
  const char* lds; u32 base;
  u32 addr(u32 f) const { return (f & 2u) ? base + f - 2u : f; }
  template <class F> static I map(F f) { I r; for (int k = 0; k < 16; k++) r.v[k] = f(k); return r; }
  static i32 chk32(i64 x) { if (x > 0x7fffffffll || x < -0x80000000ll) { g_wide_violations++; if (getenv("NBLS_SIM_WIDE_DEBUG")) fprintf(stderr, "wide: 32-bit overflow %lld\n", (long long)x); } return (i32)x; }
  I zero() const { return map([](int) { return 0; }); }
  void fence() const {}
  bool skip_rows() const { return false; }
  I konst(u32 c) const { return map([&](int) { return (i32)c; }); }
  I add(const I& a, const I& b) const { return map([&](int k) { return chk32((i64)a.v[k] + b.v[k]); }); }
  I sub(const I& a, const I& b) const { return map([&](int k) { return chk32((i64)a.v[k] - b.v[k]); }); }
  I and_(const I& a, u32 m) const { return map([&](int k) { return (i32)((u32)a.v[k] & m); }); }
  I low13(const I& a) const { return map([&](int k) { return k < 13 ? (i32)((u32)a.v[k] & LMASK) : a.v[k]; }); }
  I carry13(const I& a) const { return map([&](int k) { return k < 13 ? a.v[k] : 0; }); }
  I sar(const I& a, int s) const { return map([&](int k) { return a.v[k] >> s; }); }
  I mul_lo(const I& a, u32 c) const { return map([&](int k) { return (i32)((u32)a.v[k] * c); }); }
  I mul_small(const I& a, u32 c) const { return map([&](int k) { return chk32((i64)a.v[k] * (i64)c); }); }
  I muls(const I& a, int c) const { return map([&](int k) { return chk32((i64)a.v[k] * (i64)c); }); }
  // fp_inv_wide.h
  I or_(const I& a, const I& b) const { return map([&](int k) { return a.v[k] | b.v[k]; }); }
  I spread(i32 s) const { return map([&](int) { return s; }); }
  i32 first(const I& a) const { for (int k = 1; k < 16; k++) if (a.v[k] != a.v[0]) g_wide_violations++; return a.v[0]; }      // (must be row-uniform)
  I plimbs() const { const u32 P[NL] = NBLS_P28; return map([&](int k) { return k < NL ? (i32)P[k] : 0; }); }
  u32 nonzero_mask(const I& a) const { u32 m = 0; for (int k = 0; k < 16; k++) if (a.v[k] != 0) m |= 1u << k; return m; }
  I gather(const I& a, u32 j) const { return map([&](int) { return a.v[j & 15u]; }); }
  I pick(u32 lanebit, const I& a, const I& b) const { return map([&](int k) { return ((lanebit >> k) & 1u) ? a.v[k] : b.v[k]; }); }
  I lane_eq(u32 j, const I& a, const I& b) const { return map([&](int k) { return (u32)k == j ? a.v[k] : b.v[k]; }); }
  I lo(const W& w) const { return map([&](int k) { return (i32)w.v[k]; }); }
  W wzero() const { W w; for (int k = 0; k < 16; k++) w.v[k] = 0; return w; }
  static i64 chk64(__int128 t) { if (t > (__int128)0x7fffffffffffffffll || t < -(__int128)0x7fffffffffffffffll - 1) { g_wide_violations++; if (getenv("NBLS_SIM_WIDE_DEBUG")) fprintf(stderr, "wide: 64-bit overflow\n"); } return (i64)t; }
  W mad(const I& a, const I& b, const W& acc) const { W w; for (int k = 0; k < 16; k++) w.v[k] = chk64((__int128)acc.v[k] + (__int128)a.v[k] * b.v[k]); return w; }
  W mad_p(const I& ml, const W& acc) const { const u32 P[NL] = NBLS_P28; W w; for (int k = 0; k < 16; k++) w.v[k] = chk64((__int128)acc.v[k] + (__int128)(k < NL ? P[k] : 0u) * ml.v[0]); return w; }
  W mad_pq(const I& q, const W& acc) const { const u32 P[NL] = NBLS_P28; W w; for (int k = 0; k < 16; k++) w.v[k] = chk64((__int128)acc.v[k] + (__int128)(k < NL ? P[k] : 0u) * q.v[k]); return w; }
  W sar28(const W& a) const { W w; for (int k = 0; k < 16; k++) w.v[k] = a.v[k] >> 28; return w; }
  W addw(const W& a, const I& x) const { W w; for (int k = 0; k < 16; k++) w.v[k] = chk64((__int128)a.v[k] + x.v[k]); return w; }
  W addww(const W& a, const W& b) const { W w; for (int k = 0; k < 16; k++) w.v[k] = chk64((__int128)a.v[k] + b.v[k]); return w; }
  I wred_q(const I& top) const { return map([&](int k) { i32 t = top.v[k] - 9; t = t < 0 ? 0 : t; const u32 q = (u32)(((u64)(u32)t * 2642610142u) >> 48); return (i32)(q > (u32)(QP_TABLE_ENTRIES - 1) ? (u32)(QP_TABLE_ENTRIES - 1) : q); }); }
  I bcast(const I& a, int i) const { return map([&](int) { return a.v[i]; }); }
  I shl1(const I& a) const { return map([&](int k) { return k == 15 ? 0 : a.v[k + 1]; }); }
  I shr1(const I& a) const { return map([&](int k) { return k == 0 ? 0 : a.v[k - 1]; }); }
  I ld(u32 f) const { const u32 off = addr(f); return map([&](int k) { i32 x; memcpy(&x, lds + off + 4 * k, 4); return x; }); }
  I gload(const u32* g, bool live) const { return map([&](int k) { return (live && k < NL) ? (i32)g[k] : 0; }); }
  void gstore(u32* g, const I& v, bool live) const { if (live) for (int k = 0; k < 16; k++) g[k] = (u32)v.v[k]; }
};
struct WideDescHost { const u32* w; u32 operator()(int k) const { return w[k]; } };
// 0: ran; -4: the program has a step the form does not implement (or shared constants / a lane split)
static int sim_run_wide(const Program& p, unsigned n_items, const IOBuf* bufs) {
  if (p.lsplit != 1 || p.W > 16) return -4;
  for (auto& st : p.steps) if (!wide_step_supported(st, p.descs.data())) return -4;
  const u32 base = p.inst_base(0);
  std::vector<u32> image((size_t)(base + p.inst_bytes()) / 4 + 8);
  char* lds = (char*)image.data();
  for (unsigned item = 0; item < n_items; item++) {
    std::fill(image.begin(), image.end(), 0u);
    for (unsigned c = 0; c < p.nconst; c++) memcpy(lds + c * p.slot_bytes, p.consts.data() + c * RAW_WORDS, NL * 4);
    for (auto& st : p.steps) {
      struct Pending { u32 dst; WideRowHost::I v; };
      std::vector<Pending> pend;
      for (unsigned row = 0; row < st.nlanes; row++) {
        WideRowHost l; l.lds = lds; l.base = base;
        WideOps<WideRowHost> o(l);
        WideDescHost d{p.descs.data() + st.desc_off + row * st.stride};
        Pending pd;
        if (wide_step(o, st, d, bufs, item, true, pd.dst, pd.v)) pend.push_back(pd);
      }
      for (auto& pd : pend) { const u32 a = (pd.dst & 2u) ? base + pd.dst - 2u : pd.dst; memcpy(lds + a, pd.v.v, 64); }
    }
  }
  return 0;
}
// g1_wide.h on the host: the rows of a round one after the other, their results written once all of them have read (the device: one wavefront, LDS in program order)
static const u32 GW_TABLE_HOST[GW_TABLE_WORDS] = GW_TABLE_INIT;
static void sim_g1_wide_combine(const u32* S, int nwin, int shift, u32* out) {
  std::vector<u32> image((size_t)GW_SLOTS * 16, 0u);
  char* lds = (char*)image.data();
  auto fetch = [&](const u32* pt, int s0, int s1, int s2) { const int sl[3] = {s0, s1, s2}; for (int r = 0; r < 3; r++) { memset(lds + sl[r] * 64, 0, 64); memcpy(lds + sl[r] * 64, pt + r * RAW_WORDS, NL * 4); } };
  auto run_round = [&](int q, int p0) {
    WideRowHost::I res[4];
    for (int row = 0; row < 4; row++) {
      WideRowHost l; l.lds = lds; l.base = 0;
      WideOps<WideRowHost> o(l); WideG1<WideRowHost> g(o);
      const u32* d = GW_TABLE_HOST + (4 * q + row) * GW_ROW_WORDS;
      res[row] = p0 == 1 ? g.round<1>(d) : g.round<2>(d);
    }
    for (int row = 0; row < 4; row++) memcpy(lds + GW_TABLE_HOST[(4 * q + row) * GW_ROW_WORDS + 4] * 64, res[row].v, 64);
  };
  fetch(S + (size_t)(nwin - 1) * 3 * RAW_WORDS, GW_X, GW_U, GW_Z);
  for (int w = nwin - 2; w >= 0; w--) {
    fetch(S + (size_t)w * 3 * RAW_WORDS, GW_X2, GW_Y2, GW_Z2);
    for (int i = 0; i < shift; i++) { run_round(0, 1); run_round(1, 1); }
    for (int q = 0; q < GW_ADD_ROUNDS; q++) run_round(GW_DBL_ROUNDS + q, 2);
  }
  for (int row = 0; row < 3; row++) {
    WideRowHost l; l.lds = lds; l.base = 0;
    WideOps<WideRowHost> o(l); WideG1<WideRowHost> g(o);
    const WideRowHost::I v = row == 0 ? l.ld(GW_X * 64u) : row == 1 ? l.sub(l.ld(GW_U * 64u), l.ld(GW_V * 64u)) : l.ld(GW_Z * 64u);
    const WideRowHost::I e = g.leave(v, row == 1 ? 2u : 1u);
    for (int k = 0; k < 16; k++) { if (e.v[k] < 0 || (k < NL - 1 && e.v[k] > (i32)LMASK)) g_wide_violations++; out[row * RAW_WORDS + k] = (u32)e.v[k]; }
  }
}
static int g_sim_wide = 0;
extern "C" {
// the one-limb-per-lane form for the programs it implements (others run as usual): 0 off, 1 on; nbls_sim_wide_violations: device assumptions violated since the last call (0 on a correct build)
__attribute__((visibility("default"))) void nbls_sim_set_wide(int on) { g_sim_wide = on; }
// the window combination of the G1 MSM with one limb per lane (g1_wide.h): S = nwin projective points of 3 raw elements, out = sum_w 2^(shift w) S_w (3 raw elements)
__attribute__((visibility("default"))) void nbls_sim_g1_wide_combine(const u32* S, int nwin, int shift, u32* out) { sim_g1_wide_combine(S, nwin, shift, out); }
__attribute__((visibility("default"))) unsigned long nbls_sim_wide_violations() { const unsigned long v = g_wide_violations; g_wide_violations = 0; return v; }
__attribute__((visibility("default"))) int nbls_sim_wide_supported(int prog) { if (prog < 0 || prog >= (int)P_COUNT) return 0; const Program& p = get_program((ProgId)prog); if (p.lsplit != 1 || p.W > 16) return 0; for (auto& st : p.steps) if (!wide_step_supported(st, p.descs.data())) return 0; return 1; }
// translated programs (aot.h) instead of the interpreter's semantics for the programs that have an ahead-of-time kernel: 0 off, 1 on
__attribute__((visibility("default"))) void nbls_sim_set_aot(int on) { g_sim_aot = on; }
__attribute__((visibility("default"))) int nbls_sim_has_aot(int prog) { if (prog < 0 || prog >= (int)P_COUNT) return 0; for (auto& k : g_sim_kernels) for (int j = 0; j < 4; j++) if (k.prog_id[j] == prog) return 1; return 0; }
// bufs: 8 pointers + 8 strides
__attribute__((visibility("default"))) int nbls_sim_run(int prog, unsigned n_items, uint8_t** ptrs, const uint64_t* strides) {
  if (prog < 0 || prog >= P_COUNT) return -1;
  IOBuf b[MAX_BUFS];
  for (int i = 0; i < MAX_BUFS; i++) { b[i].ptr = ptrs[i]; b[i].stride = strides[i]; }
  if (g_sim_wide) { const int r = sim_run_wide(get_program((ProgId)prog), n_items, b); if (r != -4) return r; }
  if (g_sim_aot) { const int r = sim_run_aot(prog, n_items, b); if (r != -2) return r; }
  sim_run(get_program((ProgId)prog), n_items, b);
  return 0;
}
// out = in^-1 on raw elements (16 words each): the same fp_mont_inverse routine the inversion kernel runs per lane
// the same inverse with one limb per lane (fp_inv_wide.h), one element after the other
__attribute__((visibility("default"))) void nbls_sim_fp_inv_wide(unsigned n, const u32* in, u32* out) {
  const u32 R3[NL] = NBLS_R3_INIT;
  for (unsigned e = 0; e < n; e++) {
    WideRowHost l; l.lds = nullptr; l.base = 0;
    WideOps<WideRowHost> o(l); WideInv<WideRowHost> w(o);
    WideRowHost::I y, r3;
    for (int k = 0; k < 16; k++) { y.v[k] = k < NL ? (i32)in[SLOT_WORDS * e + k] : 0; r3.v[k] = k < NL ? (i32)R3[k] : 0; }
    const WideRowHost::I r = w.invert(y, r3);
    for (int k = 0; k < 16; k++) { if (k < NL && (r.v[k] < 0 || (k < NL - 1 && r.v[k] > (i32)LMASK))) g_wide_violations++; out[SLOT_WORDS * e + k] = k < NL ? (u32)r.v[k] : 0u; }
  }
}
__attribute__((visibility("default"))) void nbls_sim_fp_inv(unsigned n, const u32* in, u32* out) {
  for (unsigned k = 0; k < n; k++) { u32 r[NL]; fp_mont_inverse(r, in + SLOT_WORDS * k); memcpy(out + SLOT_WORDS * k, r, NL * 4); out[SLOT_WORDS * k + 14] = out[SLOT_WORDS * k + 15] = 0; }
}
// out = in^e; which: 0 = (p+1)/4 on Fp, 1 = (p^2+7)/16 on Fp2, 2 = (p^2-9)/16 on Fp2, 3 = (p-3)/4 on Fp (stand-ins for the pow kernels; raw elements of 16 words)
static void mmh(u32* r, const u32* a, const u32* b) { u32 t[NL]; mont_mul28(t, a, b); memcpy(r, t, NL * 4); }
static void addh(u32* r, const u32* a, const u32* b) { u32 t[NL]; for (int i = 0; i < NL; i++) t[i] = a[i] + b[i]; carry_norm(t); memcpy(r, t, NL * 4); }
static void subh(u32* r, const u32* a, const u32* b) { const u32 BIAS[NL] = NBLS_BIAS16_28; u32 t[NL]; for (int i = 0; i < NL; i++) t[i] = a[i] + BIAS[i] - b[i]; carry_norm(t); memcpy(r, t, NL * 4); }
static void fp2mulh(u32* r, const u32* a, const u32* b) {   // elements: c0 at [0..13], c1 at [16..29]
  u32 t1[NL], t2[NL], s1[NL], s2[NL], m[NL];
  mmh(t1, a, b); mmh(t2, a + 16, b + 16); addh(s1, a, a + 16); addh(s2, b, b + 16); mmh(m, s1, s2);
  u32 r0[NL], r1[NL], u[NL]; subh(r0, t1, t2); addh(u, t1, t2); subh(r1, m, u);
  u32 one[NL]; memcpy(one, NBLS_R1, NL * 4); mmh(r0, r0, one); mmh(r1, r1, one);   // contract (keeps values far below the 16p subtraction bias)
  memset(r, 0, 32 * 4); memcpy(r, r0, NL * 4); memcpy(r + 16, r1, NL * 4);
}
// plain square-and-multiply (what nbls_sim_fp_pow was until round 5): kept as the independent check of the chains below
__attribute__((visibility("default"))) void nbls_sim_fp_pow_naive(unsigned n, const u32* in, u32* out, int which) {
  const uint64_t* e = which == 0 ? NBLS_EXP_P_PLUS_1_DIV_4 : which == 1 ? NBLS_EXP_P2_PLUS_7_DIV_16 : which == 2 ? NBLS_EXP_P2_MINUS_9_DIV_16 : NBLS_EXP_P_MINUS_3_DIV_4;
  int bits = which == 0 ? NBLS_P_PLUS_1_DIV_4_BITS : which == 1 ? NBLS_P2_PLUS_7_DIV_16_BITS : which == 2 ? NBLS_P2_MINUS_9_DIV_16_BITS : NBLS_P_MINUS_3_DIV_4_BITS;
  for (unsigned k = 0; k < n; k++) {
    if (which == 0 || which == 3) {
      u32 acc[16] = {0}; memcpy(acc, NBLS_R1, NL * 4);
      for (int i = bits - 1; i >= 0; i--) { mmh(acc, acc, acc); if ((e[i >> 6] >> (i & 63)) & 1) mmh(acc, acc, in + 16 * k); }
      memcpy(out + 16 * k, acc, 64);
    } else {
      u32 acc[32] = {0}; memcpy(acc, NBLS_R1, NL * 4);
      for (int i = bits - 1; i >= 0; i--) { fp2mulh(acc, acc, acc); if ((e[i >> 6] >> (i & 63)) & 1) fp2mulh(acc, acc, in + 32 * k); }
      memcpy(out + 32 * k, acc, 128);
    }
  }
}
// The chains of pow_exec.h on the host: the sequences (tables, sliding windows, the split of the Fp2 exponents) and the lane arithmetic the kernels of
// pow_kernels.hip run; a lane pair of the Fp2 kernel is one value with both components here.
struct FpHost {
  struct V { u32 v[NL]; };
  const u32* in; u32* out; u32 tab[POW_TAB][NL];
  void sqr(V& r, const V& a) { u32 t[NL]; mont_sqr28(t, a.v); memcpy(r.v, t, sizeof t); }
  void mul(V& r, const V& a, const V& b) { u32 t[NL]; mont_mul28(t, a.v, b.v); memcpy(r.v, t, sizeof t); }
  void copy(V& r, const V& a) { r = a; }
  void load(V& r) { memcpy(r.v, in, NL * 4); }
  void store(const V& a) { memset(out, 0, 64); memcpy(out, a.v, NL * 4); }
  void tab_put(int j, const V& a) { memcpy(tab[j], a.v, NL * 4); }
  void tab_get(V& r, unsigned j) { memcpy(r.v, tab[j], NL * 4); }
};
struct Fp2Host {
  struct V { u32 c[2][NL]; };
  const u32* in; u32* out; V tab[POW_TAB];
  void sqr(V& r, const V& a) { V t; fp2_sqr_c(t.c[0], a.c[0], a.c[1], false); fp2_sqr_c(t.c[1], a.c[1], a.c[0], true); r = t; }
  void mul(V& r, const V& a, const V& b) { V t; fp2_mul_c(t.c[0], a.c[0], a.c[1], b.c[0], b.c[1], false); fp2_mul_c(t.c[1], a.c[1], a.c[0], b.c[1], b.c[0], true); r = t; }
  void conj(V& r, const V& a) { const u32 BIAS[NL] = NBLS_BIAS16_28; V t = a; for (int k = 0; k < NL; k++) t.c[1][k] = BIAS[k] - a.c[1][k]; carry_norm(t.c[0]); carry_norm(t.c[1]); r = t; }
  void copy(V& r, const V& a) { r = a; }
  void load(V& r) { for (int h = 0; h < 2; h++) mont_mul28(r.c[h], in + 16 * h, NBLS_R1); }
  void store(const V& a) { memset(out, 0, 128); memcpy(out, a.c[0], NL * 4); memcpy(out + 16, a.c[1], NL * 4); }
  void tab_put(int j, const V& a) { tab[j] = a; }
  void tab_get(V& r, unsigned j) { r = tab[j]; }
};
// out = in^e through the kernels' chains; which: 0 = (p+1)/4 on Fp, 1 = (p^2+7)/16 on Fp2, 2 = (p^2-9)/16 on Fp2, 3 = (p-3)/4 on Fp (raw elements of 16 words)
__attribute__((visibility("default"))) void nbls_sim_fp_pow(unsigned n, const u32* in, u32* out, int which) {
  std::vector<unsigned char> ops;
  if (which == 1 || which == 2) {
    uint64_t K[6]; for (int j = 0; j < 6; j++) K[j] = NBLS_EXP_P_MINUS_3_DIV_4[j];
    K[0] -= 2;
    for (int j = 0; j < 6; j++) K[j] = (K[j] >> 2) | (j < 5 ? K[j + 1] << 62 : 0);
    ops = pow_make_ops(K, 377);
  } else ops = which == 0 ? pow_make_ops(NBLS_EXP_P_PLUS_1_DIV_4, NBLS_P_PLUS_1_DIV_4_BITS) : pow_make_ops(NBLS_EXP_P_MINUS_3_DIV_4, NBLS_P_MINUS_3_DIV_4_BITS);
  const int nops = (int)(ops.size() / 2);
  for (unsigned k = 0; k < n; k++) {
    if (which == 0 || which == 3) { FpHost o; o.in = in + 16 * k; o.out = out + 16 * k; fp_pow_seq(o, ops.data(), nops); }
    else { Fp2Host o; o.in = in + 32 * k; o.out = out + 32 * k; fp2_pow_seq(o, ops.data(), nops, which == 1 ? 8 : 7); }
  }
}
// out = in^e through the one-limb-per-lane chains (which as nbls_sim_fp_pow); returns the number of violated device assumptions (0 on a correct build)
__attribute__((visibility("default"))) unsigned long nbls_sim_fp_pow_wide(unsigned n, const u32* in, u32* out, int which) {
  std::vector<unsigned char> ops;
  if (which == 1 || which == 2) {
    uint64_t K[6]; for (int j = 0; j < 6; j++) K[j] = NBLS_EXP_P_MINUS_3_DIV_4[j];
    K[0] -= 2;
    for (int j = 0; j < 6; j++) K[j] = (K[j] >> 2) | (j < 5 ? K[j + 1] << 62 : 0);
    ops = pow_make_ops(K, 377);
  } else ops = which == 0 ? pow_make_ops(NBLS_EXP_P_PLUS_1_DIV_4, NBLS_P_PLUS_1_DIV_4_BITS) : pow_make_ops(NBLS_EXP_P_MINUS_3_DIV_4, NBLS_P_MINUS_3_DIV_4_BITS);
  const int nops = (int)(ops.size() / 2);
  const WideConsts consts = wide_consts();
  g_wide_violations = 0;
  for (unsigned k = 0; k < n; k++) {
    if (which == 0 || which == 3) { WideHost l; l.in = in + 16 * k; l.out = out + 16 * k; l.fp2 = false; WideField<WideHost, false> f(l, consts); fp_pow_seq(f, ops.data(), nops); }
    else { WideHost l; l.in = in + 32 * k; l.out = out + 32 * k; l.fp2 = true; WideField<WideHost, true> f(l, consts); fp2_pow_seq(f, ops.data(), nops, which == 1 ? 8 : 7); }
  }
  return g_wide_violations;
}
// op list statistics of an exponent (tests): squarings, multiplications
__attribute__((visibility("default"))) void nbls_sim_pow_ops(int which, unsigned* out2) {
  std::vector<unsigned char> ops;
  if (which == 1 || which == 2) {
    uint64_t K[6]; for (int j = 0; j < 6; j++) K[j] = NBLS_EXP_P_MINUS_3_DIV_4[j];
    K[0] -= 2;
    for (int j = 0; j < 6; j++) K[j] = (K[j] >> 2) | (j < 5 ? K[j + 1] << 62 : 0);
    ops = pow_make_ops(K, 377);
  } else ops = which == 0 ? pow_make_ops(NBLS_EXP_P_PLUS_1_DIV_4, NBLS_P_PLUS_1_DIV_4_BITS) : pow_make_ops(NBLS_EXP_P_MINUS_3_DIV_4, NBLS_P_MINUS_3_DIV_4_BITS);
  out2[0] = out2[1] = 0;
  for (size_t k = 0; k < ops.size() / 2; k++) { out2[0] += ops[2 * k]; if (k && ops[2 * k + 1] != 0xff) out2[1]++; }
}
// canonical representative of raw elements (tests compare results that may differ by multiples of p)
__attribute__((visibility("default"))) void nbls_sim_fp_canon(unsigned n, const u32* in, u32* out) {
  for (unsigned k = 0; k < n; k++) { u32 t[NL]; mont_mul28(t, in + 16 * k, NBLS_R1); csub_p(t); memset(out + 16 * k, 0, 64); memcpy(out + 16 * k, t, NL * 4); }
}
// static verification of a compiled program (trace.h verify_program): 0 = clean, else the first violation in msg
__attribute__((visibility("default"))) int nbls_sim_verify(int prog, char* msg, unsigned cap) {
  if (prog < 0 || prog >= P_COUNT) return -1;
  const std::string e = verify_program(get_program((ProgId)prog));
  if (msg && cap) { snprintf(msg, cap, "%s", e.c_str()); }
  return e.empty() ? 0 : 1;
}
// the verifier on a deliberately damaged copy of a program: descriptor word `word` (or, with step_field >= 0, a header field of step `word`) is
// XOR-ed with `flip`; returns 1 and the violation when the damage is caught
__attribute__((visibility("default"))) int nbls_sim_verify_damaged(int prog, unsigned word, unsigned flip, int step_field, char* msg, unsigned cap) {
  if (prog < 0 || prog >= P_COUNT) return -1;
  Program p = get_program((ProgId)prog);
  if (step_field < 0) { if (word >= p.descs.size()) return -1; p.descs[word] ^= flip; }
  else { if (word >= p.steps.size()) return -1; Step& st = p.steps[word]; if (step_field == 0) st.desc_off ^= flip; else if (step_field == 1) st.nlanes ^= (uint8_t)flip; else if (step_field == 2) st.p0 ^= (uint8_t)flip; else st.kind ^= (uint8_t)flip; }
  const std::string e = verify_program(p);
  if (msg && cap) snprintf(msg, cap, "%s", e.c_str());
  return e.empty() ? 0 : 1;
}
// slot placement of a program's translated form (aot_layout.h): out = {has a valid table row, LDS read cycles per wavefront as compiled, with the placement, without any conflict}
__attribute__((visibility("default"))) int nbls_sim_layout_info(int prog, unsigned long* out4) {
  if (prog < 0 || prog >= P_COUNT) return -1;
  const Program& p = get_program((ProgId)prog);
  const AotLayout* l = aot_layout_for(p);
  const AotLdsCost c0 = aot_layout_cost(p, AotLayout());
  const AotLdsCost c1 = l ? aot_layout_cost(p, *l) : c0;
  out4[0] = l ? 1 : 0; out4[1] = c0.cycles; out4[2] = c1.cycles; out4[3] = c0.floor;
  return 0;
}
// the scalar side of the endomorphism splits as the device runs it (scalar_split.h): dims = 2 / 4: base-|z| digits; dims = 0: the sign-aligned recoding (4 x 32 bytes per scalar)
__attribute__((visibility("default"))) void nbls_sim_scalar_split(unsigned n, unsigned dims, const uint8_t* scalars, uint8_t* out) {
  for (unsigned i = 0; i < n; i++) { if (dims) scalar_decompose(scalars + 32ull * i, dims, out + 32ull * dims * i); else scalar_sac_recode(scalars + 32ull * i, out + 128ull * i); }
}
__attribute__((visibility("default"))) int nbls_sim_program_count() { return (int)P_COUNT; }
__attribute__((visibility("default"))) void nbls_sim_stats() { for (int i = 0; i < P_COUNT; i++) print_stats(get_program((ProgId)i)); }
}
