// nbls_api.cpp -- C ABI (include/nbls.h) and device runtime of the pairing engine: program upload, scratch
// management and the launch pipelines.  No CPU arithmetic path exists here: if HIP or the GPU is unavailable every
// entry point fails with NBLS_ENOGPU / NBLS_EHIP.
#include <hip/hip_runtime.h>
#include "config.h"
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>
#include "nbls.h"
#include "programs.h"
#include "consts_gen.h"
#include "fp_inv.h"
#include "pow_exec.h"
#include "sha256.h"
#include "curve.h"   // G1_FIXED_WIN and the table geometry of pt_mul_fixed_g1
#include <thread>

extern "C" int nbls_vm_launch(const nbls::KernelArgs* ka, unsigned lds_bytes, void* stream);
extern "C" int nbls_vm_wide_launch(const nbls::KernelArgs* ka, unsigned lds_bytes, void* stream);
#include "wide_exec.h"   // wide_step_supported
#include "aot.h"
#include <map>
#include <tuple>
extern "C" int nbls_fp_inv_launch(unsigned n, const void* in, void* out, void* stream);
extern "C" int nbls_flag_compact_launch(unsigned n, const void* flags, void* list, void* count, void* stream);
extern "C" int nbls_xmd_launch(unsigned n, const void* msgs, const void* offsets, const void* dst, unsigned dst_len, void* out, unsigned len_in_bytes, void* bad_flag, void* stream);
extern "C" int nbls_msm_keys_launch(unsigned n, unsigned nwin, const void* scalars, void* keys, void* vals, void* stream);
extern "C" int nbls_msm_decompose_launch(unsigned n, unsigned dims, const void* scalars, void* out, void* stream);
extern "C" int nbls_msm_sac_launch(unsigned n, const void* scalars, void* out, void* stream);
extern "C" int nbls_msm_sort_launch(void* temp, size_t* temp_bytes, const void* keys_in, void* keys_out, const void* vals_in, void* vals_out, size_t m, int key_bits, void* stream);
extern "C" int nbls_msm_gather_launch(size_t m, unsigned elem_bytes, const void* idx, const void* src, void* dst, void* stream);
extern "C" int nbls_msm_rank_launch(void* temp, size_t* temp_bytes, size_t m, const void* keys, void* pos, void* maxrun_u32, void* stream);
extern "C" int nbls_msm_pairs_launch(size_t m, unsigned d, const void* keys, const void* pos, void* list, void* count_u32, void* stream);
extern "C" int nbls_msm_fill_launch(size_t count, unsigned elem_bytes, const void* ident, void* dst, void* stream);
extern "C" int nbls_msm_heads_launch(size_t m, unsigned elem_bytes, const void* keys, const void* P, void* buckets, void* stream);
extern "C" int nbls_msm_bitsel_launch(unsigned nwin, unsigned elem_bytes, const void* buckets, void* G, void* stream);
extern "C" int nbls_fp_pow_launch(unsigned n, const void* in, void* out, const void* ops, int nops, void* scratch, int is_fp2, void* stream);
extern "C" int nbls_pow_wide_launch(unsigned n, const void* in, void* out, const void* ops, int nops, int is_fp2, void* stream);

using namespace nbls;
static_assert(P_COUNT <= NBLS_N_PROGRAMS, "nbls_timing_read's arrays (NBLS_N_PROGRAMS + 1 entries) must cover every step program");
static const size_t RAW = RAW_FP_BYTES;     // one raw field element in HBM scratch (14 limbs + padding)
static const size_t F12 = 12 * RAW;        // raw Fp12
static const size_t LINE_BYTES = (size_t)LINE_ELEMS * RAW;   // one line table: 68 triples of Fp2 as raw elements (26,112 B)
static const size_t SPLIT_MILLER_MIN = 4097;   // pairs from which the Miller loop runs as LINES + ACC (see nbls_pairing_batch_dev).  Round 4 measured the two programs ahead from 4096 pairs on; on the round-5 / 6 build a 4096-pair call -- exactly one wavefront of the fused program (four items) on each of the 1024 SIMDs -- takes 2.19 ms fused against 2.32 ms split, 3072 pairs likewise, and from 4608 pairs on the split form wins (tools/ab_split_min.py, profiles/round6_ab_split_min.txt)
static const size_t LINES_CHUNK = 131072;   // pairs whose line tables are in HBM at a time (3.4 GB of the 288); larger batches run chunk by chunk on the same stream

#define EXPORT extern "C" __attribute__((visibility("default")))
static std::recursive_mutex g_null_mu;   // locked in place of a context's mutex when the caller passed no context (the call then fails with NBLS_EINVAL)

static const size_t EXPC_MIN_DEFAULT = (size_t)1 << 40;   // items from which the cyclotomic exponentiations run with compressed squarings (expx below): never, unless asked for
struct DevProgram {
  Step* steps = nullptr; u32* descs = nullptr; u32* consts = nullptr;
  const Program* p = nullptr;
  bool wide_ok = false;   // every step is one the one-limb-per-lane interpreter implements (wide_exec.h): launches of at most ctx->wide_max items run on nbls_vm_kernel_wide
  int aot = -1; AotStep* aot_steps = nullptr; u32* aot_descs = nullptr; u32 aot_lds = 0;   // ahead-of-time kernel of this program (aot.h) and the translated program, when every step's signature is in the kernel's table
};

struct nbls_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::recursive_mutex mu;   // held for the whole of every exported call (host-level calls re-enter it through the *_dev entry points)
  DevProgram prog[P_COUNT];
  std::vector<uint8_t> dst_host; uint8_t* dst_dev = nullptr;   // hash-to-curve domain-separation tag last used by nbls_verify_batch_msgs_dev, and its device copy
  std::map<std::tuple<int, int, int, int>, DevProgram> tower;   // single tower operations (nbls_tower_op_batch), uploaded on first use
  // scratch (device)
  uint8_t *F = nullptr, *N = nullptr, *NI = nullptr, *io_g1 = nullptr, *io_g2 = nullptr, *io_f12 = nullptr, *one12 = nullptr;
  uint8_t* T[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // t1..t7 of the final exponentiation, raw Fp12
  // general scratch pool for the codec / hash / sum pipelines (grown on demand)
  static const int NSB = 20;
  uint8_t* sb[NSB] = {nullptr}; size_t sb_cap[NSB] = {0};
  // staging buffers of the host-buffer entry points (HostIO): kept between calls -- a hipMalloc / hipFree pair per buffer and call cost more than the copies at small batches
  struct IoBlock { void* p; size_t cap; bool busy; }; std::vector<IoBlock> io_pool; size_t io_pool_bytes = 0;
  uint8_t* nib[4] = {nullptr, nullptr, nullptr, nullptr}; int nnib[4] = {0, 0, 0, 0};   // op lists (pow_exec.h) of the exponents (p+1)/4, (p^2+7)/16, (p^2-9)/16, (p-3)/4 and their lengths in ops
  uint8_t* neg_g1 = nullptr;    // -G1 generator, affine wire bytes (verify: e(-G, S))
  uint8_t* gen_g1 = nullptr;    // G1 generator, affine wire bytes (getPublicKey)
  uint8_t* g1_fixed = nullptr;  // fixed-base table of the generator (curve.h pt_mul_fixed_g1): raw projective multiples [d 2^(WIN w)]G, built on first use by the ladder
  // side stream for the one-element chains of verifyBatch (signature decompression: a 758-bit Fp2 exponentiation on a single
  // lane is ~4 ms of pure latency) so that they overlap the batch-wide kernels instead of serialising with them
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; uint8_t* side_scratch = nullptr;
  // large pairing batches run as two halves on two streams (nbls_pairing_batch_dev): item offset applied to every per-item buffer of a launch, second stream, events
  bool in_halves = false;   // the running pairing call is one of two halves on two streams: their launches fill each other's tails, so the final exponentiation's middle is NOT chained (run_chain)
  size_t ioff = 0; hipStream_t half_stream = nullptr; hipEvent_t ev_half_fork = nullptr, ev_half_join = nullptr;
  size_t halves_min = env_long("NBLS_HALVES_MIN", 8192) > 0 ? (size_t)env_long("NBLS_HALVES_MIN", 8192) : (size_t)-1;   // NBLS_HALVES_MIN=0: never (profiles of kernels running alone)
  // verifyBatch as a software pipeline (round 5, verify_pipeline): events of the chunks (two each), the "xmd met non-monotonic offsets" flag lives behind the statuses
  std::vector<hipEvent_t> pipe_ev; hipEvent_t ev_pipe_done = nullptr; std::vector<hipStream_t> pipe_streams;
  long verify_chunks = env_long("NBLS_VERIFY_CHUNKS", 2), verify_last_pct = env_long("NBLS_VERIFY_LAST_PCT", 12), verify_pipe_min = env_long("NBLS_VERIFY_PIPE_MIN", 32768);   // nbls_set_tuning(NBLS_TUNE_VERIFY_*)
  hipStream_t side2 = nullptr; hipEvent_t ev_join2 = nullptr;   // verifyBatch: key decoding runs beside message hashing (their exponentiation kernels are latency-bound and leave issue slots free)
  uint8_t* ident_g1 = nullptr; uint8_t* ident_g2 = nullptr;   // projective identity (0 : 1 : 0), raw
  size_t cap_F = 0, cap_io = 0;
  size_t split_min = SPLIT_MILLER_MIN;   // nbls_set_tuning(NBLS_TUNE_SPLIT_MILLER_MIN)
  size_t acc8_min = (size_t)env_long("NBLS_ACC8_MIN", 131072);   // pairs per product call from which eight line tables share an accumulator (below: four).  Measured (tools/ab_acc8.sh): at 65,537 pairs the halves have 4096 groups of eight = 820 wavefronts, less than one per SIMD, and the call is slower (24.5 against 23.4 ms); at 2^18 terms 33.2 against 33.8 ms
  // cyclotomic exponentiation with compressed squarings (expx): scratch per item -- compressed powers, decompression scratch, redo flags and list -- and two redo counters (one per half)
  uint8_t *KS = nullptr, *KD = nullptr, *Kflag = nullptr; uint32_t *Klist = nullptr, *Kcount = nullptr;
  size_t expc_min = (size_t)env_long("NBLS_EXPC_MIN", (long)EXPC_MIN_DEFAULT);   // nbls_set_tuning(NBLS_TUNE_EXPC_MIN)
  size_t pt_ls2_max = (size_t)env_long("NBLS_PT_LS2_MAX", 4096);                // nbls_set_tuning(NBLS_TUNE_PT_LS2_MAX): items up to which the G2 point chains run in their two-lane forms (pt_ls2_variant)
  size_t sac_max = (size_t)env_long("NBLS_G2_SAC_MAX", 6144);                   // nbls_set_tuning(NBLS_TUNE_SAC_MAX): keys up to which sign's ladder is the sign-aligned form (dev_point_mul)
  // round 6: launches of at most wide_max items run the programs that allow it on the one-limb-per-lane interpreter (vm_wide_kernel.hip: one item per workgroup of ceil(W / 4)
  // wavefronts, the Montgomery reduction spread over a row of lanes, two barriers per step) -- the multi-wavefront item form the round-5 review asked for.  Built, bit-exact
  // (tests/test_wide_sim.py, test_gpu_pairing.py::test_one_limb_per_lane_forms) and MEASURED SLOWER than the four-lane forms: the five exponentiations of one final exponentiation
  // take 0.93 ms against 0.66 ms (tools/wide_time.py, profiles/round6_wide_time.txt).  A step is 350 instructions in its rows + ~190 around them where the four-lane form has 660,
  // but every one of them waits for its predecessor (one column per lane: no independent work), and a lone wavefront then pays ~9-11 clocks per instruction instead of ~5.  Off by
  // default: NBLS_WIDE_MAX / NBLS_TUNE_WIDE_MAX (items; 0 = never), NBLS_WIDE_PROGS = 0: only the final exponentiation's programs.  The fixed-exponent powers, where the same
  // idea removes a 196-multiply-add reduction per squaring from ONE lane, are the case that pays (pow_wide.h: 0.7 -> 0.3 ms).
  size_t wide_max = (size_t)env_long("NBLS_WIDE_MAX", 0);
  size_t chain_max = (size_t)env_long("NBLS_CHAIN_MAX", 8192);                  // nbls_set_tuning(NBLS_TUNE_CHAIN_MAX); see run_chain
  u32* qp_table = nullptr;      // multiples of p for the weak reduction (vm_exec.h weak_reduce), device copy
  uint8_t* unit_lines = nullptr;   // a line table whose 68 lines are all 1 (c0 = 1, c1 = c2 = 0): the neutral partner of an odd last pair
  uint8_t* partial = nullptr;   // 576 bytes: the Fp12 partial of the *_partial entry points (multi-GPU reductions)
  uint8_t* L = nullptr; size_t cap_L = 0;   // line tables of the Miller loop (LINE_BYTES each), at most LINES_CHUNK of them
  // the scratch above is shared by every call on this context: a call that uses another stream than its predecessor waits for it (StreamOrder)
  hipStream_t last_stream = nullptr; hipEvent_t ev_last = nullptr; bool ev_last_set = false;
  int last_hip = 0;
  // optional per-kernel timing (HIP events on the launch stream); slot P_COUNT = inversion kernel
  bool timing = false;
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> tev;
  std::vector<hipEvent_t> ev_pool;   // timing events are recycled (nbls_timing_read returns them here) instead of created per launch
};

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->last_hip = (int)e_; return NBLS_EHIP; } } while (0)

// Checked mode (make debug builds it in with -DNBLS_CHECKED; NBLS_CHECKED=1 switches it on in any build): every program is verified statically
// before its first upload (verify_program: all LDS offsets, descriptor reads and buffer indices the kernel will ever use) and every launch
// checks its buffers against the extents the program touches.  A violation is reported on stderr and the call fails with NBLS_EINVAL.
static bool checked_mode() {
#if defined(NBLS_CHECKED)
  return true;
#else
  static const bool on = env_long("NBLS_CHECKED", 0) != 0;
  return on;
#endif
}
// NBLS_AOT=0 keeps every program on the interpreter (A/B runs, profiles of the interpreter)
static bool aot_enabled() { static const bool on = env_long("NBLS_AOT", 1) != 0; return on; }
static int upload_program(nbls_ctx* ctx, DevProgram& d, const Program& p, const int k);
static int upload(nbls_ctx* ctx, ProgId id) {
  DevProgram& d = ctx->prog[id];
  if (d.p) return NBLS_OK;
  return upload_program(ctx, d, get_program(id), aot_enabled() ? nbls_aot_index((int)id) : -1);
}
// k: index of the ahead-of-time kernel that serves the program, or -1
static void free_program(DevProgram& d) {
  for (void* p : {(void*)d.steps, (void*)d.descs, (void*)d.consts, (void*)d.aot_steps, (void*)d.aot_descs}) if (p) hipFree(p);
  d = DevProgram();
}
static int upload_program(nbls_ctx* ctx, DevProgram& d, const Program& p, const int k) {
  if (checked_mode()) { const std::string e = verify_program(p); if (!e.empty()) { fprintf(stderr, "nbls (checked): %s\n", e.c_str()); return NBLS_EINVAL; } }
  // a failure half way leaves nothing behind: d.p stays unset, so a retry uploads again, and would otherwise leak what the first attempt had allocated
  auto fail = [&]() { ctx->last_hip = (int)hipGetLastError(); free_program(d); return NBLS_EHIP; };
  if (hipMalloc(&d.steps, p.steps.size() * sizeof(Step)) != hipSuccess || hipMalloc(&d.descs, p.descs.size() * 4 + 64) != hipSuccess || hipMalloc(&d.consts, p.consts.size() * 4) != hipSuccess ||
      hipMemcpy(d.steps, p.steps.data(), p.steps.size() * sizeof(Step), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d.descs, p.descs.data(), p.descs.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d.consts, p.consts.data(), p.consts.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail();
  // ahead-of-time kernel (aot.h): translate the program; one whose signatures are not all in the kernel's table (build / environment mismatch) stays on the interpreter
  if (k >= 0) {
    AotProgram ap;
    const std::string why = aot_translate(p, ap);
    if (why.empty() && nbls_aot_bind(k, &ap) == 0) {
      if (hipMalloc(&d.aot_steps, ap.steps.size() * sizeof(AotStep)) != hipSuccess || hipMalloc(&d.aot_descs, ap.descs.size() * 4) != hipSuccess ||
          hipMemcpy(d.aot_steps, ap.steps.data(), ap.steps.size() * sizeof(AotStep), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d.aot_descs, ap.descs.data(), ap.descs.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail();
      d.aot = k; d.aot_lds = ap.lds_bytes;
    } else fprintf(stderr, "nbls: %s: %s; running on the interpreter\n", p.name.c_str(), why.empty() ? "step signatures differ from the ahead-of-time kernel's table" : why.c_str());
  }
  d.wide_ok = p.lsplit == 1 && p.W <= 16 && p.inst_base(0) + p.inst_bytes() <= 64 * 1024;
  for (const Step& st : p.steps) if (!wide_step_supported(st, p.descs.data())) { d.wide_ok = false; break; }
  d.p = &p;
  return NBLS_OK;
}
// does a launch of n items of this (uploaded) program take the one-limb-per-lane form?
static bool wide_applies(const nbls_ctx* ctx, const DevProgram& d, int id, size_t n) {
  if (!d.wide_ok || n > ctx->wide_max || id < 0) return false;
  static const long which = env_long("NBLS_WIDE_PROGS", 1);
  if (which) return true;
  return id == P_EXPX || id == P_FE_MID1 || id == P_FE_MID2 || id == P_FE_EASY || id == P_NORM_RAW || id == P_MUL2 || id == P_MUL2S;
}

static hipEvent_t timing_event(nbls_ctx* ctx) {
  if (!ctx->ev_pool.empty()) { hipEvent_t e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr; hipEventCreate(&e); return e;
}

static void aot_seg(AotSeg& g, const DevProgram& d, const IOBuf* bufs) {
  g.steps = d.aot_steps; g.descs = d.aot_descs; g.consts = d.consts;
  g.nsteps = (u32)d.p->steps.size(); g.nconst = d.p->nconst; g.inst_bytes = d.p->inst_bytes(); g.slot_bytes = d.p->slot_bytes; g.shared_consts = d.p->shared_consts ? 1u : 0u;
  for (int k = 0; k < MAX_BUFS; k++) g.bufs[k] = bufs[k];
}
static int run_dev(nbls_ctx* ctx, const DevProgram& d, int id, size_t n, std::initializer_list<std::pair<int, std::pair<const void*, size_t>>> bufs, hipStream_t s, const uint32_t* n_dev, const uint32_t* item_index);
static int run(nbls_ctx* ctx, ProgId id, size_t n, std::initializer_list<std::pair<int, std::pair<const void*, size_t>>> bufs, hipStream_t s, const uint32_t* n_dev = nullptr, const uint32_t* item_index = nullptr) {
  int r = upload(ctx, id); if (r) return r;
  return run_dev(ctx, ctx->prog[id], (int)id, n, bufs, s, n_dev, item_index);
}
// id: the timing slot of the launch (a ProgId), or -1 for programs outside the registry (single tower operations)
static int run_dev(nbls_ctx* ctx, const DevProgram& d, int id, size_t n, std::initializer_list<std::pair<int, std::pair<const void*, size_t>>> bufs, hipStream_t s, const uint32_t* n_dev, const uint32_t* item_index) {
  KernelArgs ka; memset(&ka, 0, sizeof ka);
  ka.steps = d.steps; ka.descs = d.descs; ka.consts = d.consts; ka.qp_table = ctx->qp_table;
  ka.nsteps = (u32)d.p->steps.size(); ka.nconst = d.p->nconst; ka.W = d.p->W; ka.G = d.p->G; ka.slot_bytes = d.p->slot_bytes; ka.inst_bytes = d.p->inst_bytes(); ka.shared_consts = d.p->shared_consts ? 1u : 0u; ka.lsplit = d.p->lsplit; ka.n_items = (u32)n; ka.n_items_dev = n_dev; ka.item_index = item_index;
  for (auto& b : bufs) { ka.bufs[b.first].ptr = (uint8_t*)b.second.first + ctx->ioff * b.second.second; ka.bufs[b.first].stride = b.second.second; }   // ioff: the second half of a split call
  if (checked_mode()) {
    for (int k = 0; k < MAX_BUFS; k++) {
      const u32 ext = d.p->buf_extent[k];
      if (!ext) continue;
      if (!ka.bufs[k].ptr || (ka.bufs[k].stride != 0 && ka.bufs[k].stride < ext)) {
        fprintf(stderr, "nbls (checked): %s: buffer %d: %s (stride %llu, the program touches %u bytes per item)\n", d.p->name.c_str(), k, ka.bufs[k].ptr ? "stride too small" : "not bound", (unsigned long long)ka.bufs[k].stride, ext);
        return NBLS_EINVAL;
      }
      // the last item's bytes must lie inside the allocation the pointer belongs to (items reached through an index list are not bounded by n)
      hipDeviceptr_t base = nullptr; size_t size = 0;
      if (n && !item_index && hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ka.bufs[k].ptr) == hipSuccess) {
        const size_t end = (size_t)((uint8_t*)ka.bufs[k].ptr - (uint8_t*)base) + (n - 1) * ka.bufs[k].stride + ext;
        if (end > size) {
          fprintf(stderr, "nbls (checked): %s: buffer %d: %zu items of stride %llu (+%u) end %zu bytes into an allocation of %zu\n", d.p->name.c_str(), k, n, (unsigned long long)ka.bufs[k].stride, ext, end, size);
          return NBLS_EINVAL;
        }
      } else (void)hipGetLastError();
    }
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->timing) { e0 = timing_event(ctx); e1 = timing_event(ctx); hipEventRecord(e0, s); }
  int e;
  if (wide_applies(ctx, d, id, n)) e = nbls_vm_wide_launch(&ka, d.p->inst_base(0) + d.p->inst_bytes(), s);
  else if (d.aot >= 0) {
    AotArgs a; memset(&a, 0, sizeof a);
    aot_seg(a.seg[0], d, ka.bufs);
    a.nseg = 1; a.W = ka.W; a.G = ka.G; a.n_items = ka.n_items; a.qp_table = ka.qp_table; a.item_index = ka.item_index; a.n_items_dev = ka.n_items_dev;
    e = nbls_aot_launch(d.aot, &a, d.aot_lds, s);
  } else e = nbls_vm_launch(&ka, d.p->lds_bytes(), s);
  if (ctx->timing) { hipEventRecord(e1, s); if (id >= 0) ctx->tev.push_back({id, {e0, e1}}); else { ctx->ev_pool.push_back(e0); ctx->ev_pool.push_back(e1); } }
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  return NBLS_OK;
}
static int run_inv(nbls_ctx* ctx, size_t n, hipStream_t s) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->timing) { e0 = timing_event(ctx); e1 = timing_event(ctx); hipEventRecord(e0, s); }
  int e = nbls_fp_inv_launch((unsigned)n, ctx->N + ctx->ioff * RAW, ctx->NI + ctx->ioff * RAW, s);
  if (ctx->timing) { hipEventRecord(e1, s); ctx->tev.push_back({(int)P_COUNT, {e0, e1}}); }
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  return NBLS_OK;
}

// A chain: several programs executed back to back by ONE launch (aot.h): every wavefront runs them in order for its own items, the values between them pass
// through the HBM scratch buffers the separate launches would use.  Falls back to one launch per program when some program is not on an ahead-of-time kernel,
// when the programs do not share a kernel / the lanes per item, or in checked mode (whose per-launch buffer checks live in run()).
typedef std::initializer_list<std::pair<int, std::pair<const void*, size_t>>> BufList;
struct ChainLink { ProgId id; BufList bufs; };
// ctx->chain_max: items up to which the middle of the final exponentiation runs as one chain (default 8192, NBLS_CHAIN_MAX / NBLS_TUNE_CHAIN_MAX): measured equal to seven
// launches up to 4096 pairings per call (2.371 against 2.374 ms), slower where a call runs as two halves on two streams (16,384: 6.53 against 6.28 ms), whose launches fill each
// other's tails -- and slower with calls in flight on other streams for the same reason: a chained wavefront is 427 k instructions long, so the rounds of wavefronts at the end of a
// burst are coarse (twenty 4096-pairing calls on twenty streams 2.82 against 2.85 M pairings/s, 512 calls twelve deep 3.05 against 3.08 M; tools/ab_chain20.sh).  The pool
// (nbls_pool_init, nbls_multi.cpp) therefore sets it to 0 for its contexts.
static bool chains_enabled() { static const bool on = env_long("NBLS_CHAIN", 1) != 0; return on; }
static int run_chain(nbls_ctx* ctx, size_t n, std::initializer_list<ChainLink> links, hipStream_t s) {
  int r;
  bool fuse = chains_enabled() && !checked_mode() && links.size() <= (size_t)AOT_MAX_SEGS;
  int k = -1; u32 W = 0, G = 0, lds = 0;
  for (auto& l : links) {
    if ((r = upload(ctx, l.id))) return r;
    const DevProgram& d = ctx->prog[l.id];
    if (d.aot < 0 || (k >= 0 && (d.aot != k || d.p->W != W || d.p->G != G))) fuse = false;
    k = d.aot; W = d.p->W; G = d.p->G; lds = std::max(lds, d.aot_lds);
  }
  if (!fuse) { for (auto& l : links) if ((r = run(ctx, l.id, n, l.bufs, s))) return r; return NBLS_OK; }
  AotArgs a; memset(&a, 0, sizeof a);
  for (auto& l : links) {
    IOBuf bufs[MAX_BUFS]; memset(bufs, 0, sizeof bufs);
    for (auto& b : l.bufs) { bufs[b.first].ptr = (uint8_t*)b.second.first + ctx->ioff * b.second.second; bufs[b.first].stride = b.second.second; }
    aot_seg(a.seg[a.nseg++], ctx->prog[l.id], bufs);
  }
  a.W = W; a.G = G; a.n_items = (u32)n; a.qp_table = ctx->qp_table;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->timing) { e0 = timing_event(ctx); e1 = timing_event(ctx); hipEventRecord(e0, s); }
  const int e = nbls_aot_launch(k, &a, lds, s);
  if (ctx->timing) { hipEventRecord(e1, s); ctx->tev.push_back({(int)links.begin()->id, {e0, e1}}); }   // the whole chain is booked on its first program
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  return NBLS_OK;
}

static int ensure_scratch(nbls_ctx* ctx, size_t n) {
  if (n <= ctx->cap_F) return NBLS_OK;
  size_t cap = n + n / 8 + 64;
  if (ctx->F) { hipFree(ctx->F); hipFree(ctx->N); hipFree(ctx->NI); for (auto& t : ctx->T) { hipFree(t); t = nullptr; } hipFree(ctx->KS); hipFree(ctx->KD); hipFree(ctx->Kflag); hipFree(ctx->Klist); hipFree(ctx->Kcount); }
  ctx->F = ctx->N = ctx->NI = ctx->KS = ctx->KD = ctx->Kflag = nullptr; ctx->Klist = ctx->Kcount = nullptr; ctx->cap_F = 0;
  HIPCHK(hipMalloc(&ctx->F, (cap + 2) * F12));
  HIPCHK(hipMalloc(&ctx->N, cap * RAW));
  HIPCHK(hipMalloc(&ctx->NI, cap * RAW));
  for (auto& t : ctx->T) HIPCHK(hipMalloc(&t, cap * F12));
  ctx->cap_F = cap;
  return NBLS_OK;
}
// scratch of the compressed-squaring exponentiation (2.8 KB per item): only a context that runs that path -- off by default, NBLS_TUNE_EXPC_MIN -- ever allocates it
static int ensure_expc_scratch(nbls_ctx* ctx) {
  if (ctx->Kcount) return NBLS_OK;     // the LAST allocation below: a set that failed half way is not taken for complete (sized with F / T; ensure_scratch frees it when they grow)
  const size_t cap = ctx->cap_F;
  auto fail = [&]() { for (void* p : {(void*)ctx->KS, (void*)ctx->KD, (void*)ctx->Kflag, (void*)ctx->Klist, (void*)ctx->Kcount}) if (p) hipFree(p); ctx->KS = ctx->KD = ctx->Kflag = nullptr; ctx->Klist = ctx->Kcount = nullptr; ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; };
  if (ctx->KS) { hipFree(ctx->KS); ctx->KS = nullptr; } if (ctx->KD) { hipFree(ctx->KD); ctx->KD = nullptr; } if (ctx->Kflag) { hipFree(ctx->Kflag); ctx->Kflag = nullptr; } if (ctx->Klist) { hipFree(ctx->Klist); ctx->Klist = nullptr; }
  if (hipMalloc(&ctx->KS, cap * EXPC_SQ_ELEMS * RAW) != hipSuccess || hipMalloc(&ctx->KD, cap * EXPC_DEC_ELEMS * RAW) != hipSuccess || hipMalloc(&ctx->Kflag, cap) != hipSuccess ||
      hipMalloc(&ctx->Klist, cap * 4) != hipSuccess || hipMalloc(&ctx->Kcount, 8) != hipSuccess) return fail();
  return NBLS_OK;
}
static int ensure_io(nbls_ctx* ctx, size_t n) {
  if (n <= ctx->cap_io) return NBLS_OK;
  size_t cap = n + 64;
  if (ctx->io_g1) { hipFree(ctx->io_g1); hipFree(ctx->io_g2); hipFree(ctx->io_f12); }
  ctx->io_g1 = ctx->io_g2 = ctx->io_f12 = nullptr; ctx->cap_io = 0;
  HIPCHK(hipMalloc(&ctx->io_g1, cap * 96));
  HIPCHK(hipMalloc(&ctx->io_g2, cap * 192));
  HIPCHK(hipMalloc(&ctx->io_f12, cap * 576));
  ctx->cap_io = cap;
  return NBLS_OK;
}

static int ensure_lines(nbls_ctx* ctx, size_t n) {
  if (n > LINES_CHUNK + 3) n = LINES_CHUNK + 3;
  if (n <= ctx->cap_L) return NBLS_OK;
  size_t cap = n + n / 8 + 8; if (cap > LINES_CHUNK + 3) cap = LINES_CHUNK + 3;
  if (ctx->L) hipFree(ctx->L);
  ctx->L = nullptr; ctx->cap_L = 0;
  HIPCHK(hipMalloc(&ctx->L, cap * LINE_BYTES));
  ctx->cap_L = cap;
  return NBLS_OK;
}
static int need(nbls_ctx* ctx, int i, size_t bytes, uint8_t** out) {
  if (bytes > ctx->sb_cap[i]) {
    if (ctx->sb[i]) hipFree(ctx->sb[i]);
    ctx->sb[i] = nullptr; ctx->sb_cap[i] = 0;
    size_t cap = bytes + bytes / 8 + 4096;
    HIPCHK(hipMalloc(&ctx->sb[i], cap));
    ctx->sb_cap[i] = cap;
  }
  *out = ctx->sb[i];
  return NBLS_OK;
}
// Launches of at most pow_wide_max elements -- a wavefront or two per SIMD -- run the one-limb-per-lane form (pow_wide.h, nbls_pow_wide_kernel: one wavefront per element, no scratch
// table): one verify / sign spends 0.3 instead of 0.7 ms in the Fp2 exponentiation of hash-to-G2.  NBLS_POW_WIDE_MAX (0 = never).
static size_t pow_wide_max() { static const size_t v = (size_t)env_long("NBLS_POW_WIDE_MAX", 1024); return v; }
static int run_pow(nbls_ctx* ctx, int which, size_t n, const void* in, void* out, hipStream_t s, uint8_t* scratch = nullptr) {
  int is_fp2 = which == 1 || which == 2;
  if (n <= pow_wide_max()) {
    const int e = nbls_pow_wide_launch((unsigned)n, in, out, ctx->nib[which], ctx->nnib[which], which == 1 ? 8 : which == 2 ? 7 : 0, s);
    if (e) { ctx->last_hip = e; return NBLS_EHIP; }
    return NBLS_OK;
  }
  if (!scratch) { int r = need(ctx, 11, n * POW_TAB * (is_fp2 ? 2 : 1) * RAW, &scratch); if (r) return r; }
  int e = nbls_fp_pow_launch((unsigned)n, in, out, ctx->nib[which], ctx->nnib[which], scratch, which == 1 ? 8 : which == 2 ? 7 : 0, s);   // Fp2: a^((p^2+7)/16) = b^K a^8, a^((p^2-9)/16) = b^K a^7
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  return NBLS_OK;
}
static int run_inv_buf(nbls_ctx* ctx, size_t n, const void* in, void* out, hipStream_t s) {
  int e = nbls_fp_inv_launch((unsigned)n, in, out, s);
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  return NBLS_OK;
}


// Calls on one context share its scratch buffers.  The mutex serialises the host side; on the device, work submitted to the
// SAME stream is ordered anyway, and a call that names a different stream than its predecessor is made to wait for it.
struct StreamOrder {
  nbls_ctx* ctx; hipStream_t s;
  StreamOrder(nbls_ctx* c, hipStream_t st) : ctx(c), s(st) {
    if (!ctx->ev_last) hipEventCreateWithFlags(&ctx->ev_last, hipEventDisableTiming);
    if (ctx->ev_last && ctx->ev_last_set && ctx->last_stream != s) hipStreamWaitEvent(s, ctx->ev_last, 0);
  }
  ~StreamOrder() { if (ctx->ev_last && hipEventRecord(ctx->ev_last, s) == hipSuccess) { ctx->ev_last_set = true; ctx->last_stream = s; } }
};

// A call that has forked work onto other streams of the context (the side streams of verifyBatch, the second half of a large pairing call) and then fails must not return while
// those streams still run: StreamOrder records the call's end on `s` only, and the next call would free, regrow or overwrite scratch the orphaned kernels use.
// Armed right after the fork; every error return in between synchronises the device, the success path disarms it.
struct ForkGuard {
  bool armed = true;
  ~ForkGuard() { if (armed) (void)hipDeviceSynchronize(); }
};
typedef std::pair<int, std::pair<const void*, size_t>> BufArg;
static inline BufArg B(int idx, const void* p, size_t stride) { return {idx, {p, stride}}; }

// F (n raw Fp12) -> one element in F[0]; returns pointer to the buffer holding the product
// Launches of at most one wavefront per SIMD take the time of one wavefront's instruction stream, so up to LS_MAX items (one item per wavefront
// on 1024 SIMDs) the lane-split variants run: the same formulas with every lane-op's products shared by four lanes, the columns summed across them before the one
// reduction (ahead-of-time kernels nbls_aot_miller_ls / nbls_aot_expx_ls, aot.h NBLS_AOT_LS_KERNELS; on the interpreter nbls_vm_kernel_ls4).  Measured
// (tools/ab_ls.sh): one pairing 1.72 against 2.09 ms, 1024 pairings 1.79 against 2.12 ms.  NBLS_LS_MAX overrides (0 = the throughput forms at every size).
static size_t ls_max() { static const size_t v = (size_t)env_long("NBLS_LS_MAX", 1024); return v; }
// round 5: from LS_MAX + 1 to LS2_MAX items (two items per wavefront on 1024 SIMDs) the TWO-lane forms (nbls_aot_miller_ls2 / nbls_aot_expx_ls2; no interpreter form exists, so they are
// used only where the program is bound to its ahead-of-time kernel).  Measured (tools/ab_ls2.sh): 2048 pairings 1.9 against 2.17 ms.  NBLS_LS2_MAX = 0 switches them off.
static size_t ls2_max() { static const size_t v = (size_t)env_long("NBLS_LS2_MAX", 2048); return v; }
static ProgId ls_variant(nbls_ctx* ctx, ProgId id, size_t n) {
  if (n <= ctx->wide_max && upload(ctx, id) == NBLS_OK && wide_applies(ctx, ctx->prog[id], (int)id, n)) return id;      // the one-limb-per-lane form runs the plain program
  if (n <= ls_max()) {
    switch (id) {
      case P_MILLER_BYTES: return P_MILLER_BYTES_LS;
      case P_MILLER_RAW: return P_MILLER_RAW_LS;
      case P_MILLER_FE: return P_MILLER_FE_LS;
      case P_EXPX: return P_EXPX_LS;
      default: return id;
    }
  }
  if (n <= ls2_max()) {
    ProgId v = id;
    switch (id) {
      case P_MILLER_BYTES: v = P_MILLER_BYTES_LS2; break;
      case P_MILLER_RAW: v = P_MILLER_RAW_LS2; break;
      case P_MILLER_FE: v = P_MILLER_FE_LS2; break;
      case P_EXPX: v = P_EXPX_LS2; break;
      default: return id;
    }
    if (upload(ctx, v) == NBLS_OK && ctx->prog[v].aot >= 0) return v;
  }
  return id;
}
// The G2 point chains of a single verify / sign (the two ladders of clearCofactor, sign's own ladder) in their two-lane forms (round 5; nbls_aot_g2pt_ls2): four items per wavefront,
// launches of at most 4096 items -- one wavefront per SIMD at most, where a shorter instruction stream is the whole gain.  NBLS_PT_LS2_MAX / NBLS_TUNE_PT_LS2_MAX (0 = never).
static ProgId pt_ls2_variant(nbls_ctx* ctx, ProgId id, size_t n) {
  if (n <= ctx->wide_max && upload(ctx, id) == NBLS_OK && wide_applies(ctx, ctx->prog[id], (int)id, n)) return id;
  if (n > ctx->pt_ls2_max) return id;
  const ProgId v = id == P_H2C_C1 ? P_H2C_C1_LS2 : id == P_H2C_C2 ? P_H2C_C2_LS2 : id == P_G2_MUL_SAC ? P_G2_MUL_SAC_LS2 : id;
  if (v != id && upload(ctx, v) == NBLS_OK && ctx->prog[v].aot >= 0) return v;   // (no interpreter form of the two-lane split exists)
  return id;
}
static int reduce_product(nbls_ctx* ctx, size_t n, uint8_t** result, hipStream_t s) {
  // round 5: IN PLACE.  With spacing d the live elements are F[0], F[d], F[2d], ... below n; one launch multiplies F[2 i d] by F[2 i d + d] into the former for every
  // complete pair, and an odd last element -- its index is a multiple of 2 d -- simply stays alive for the next level.  No level copies or pads anything (round 4: ping-pong
  // between F and F2 with a copy of ONE behind every odd level: 15 launches + 15 copies for 16,385 values, 0.3 ms at the end of a verifyBatch with the GPU otherwise idle).
  for (size_t d = 1; d < n; d *= 2) {
    const size_t pairs = ((n + d - 1) / d) / 2;
    if (!pairs) continue;
    const int r = run(ctx, P_MUL2S, pairs, {B(3, ctx->F, 2 * d * F12), B(4, ctx->F + d * F12, 2 * d * F12), B(5, ctx->F, 2 * d * F12)}, s);
    if (r) return r;
  }
  *result = ctx->F;
  return NBLS_OK;
}
// out = conj(in^|x|) for n unitary raw Fp12 elements (cyclotomicExp + conjugate, math.ts:845-852, 862).  Default: ONE program, 63 Granger-Scott squarings
// on the tripled state and five products on twelve lanes per item (P_EXPX; its lane-split variant up to LS_MAX items).
// Opt-in from expc_min items on (NBLS_TUNE_EXPC_MIN / NBLS_EXPC_MIN; default never): Karabina's compressed squarings -- 57 squarings on the four
// coordinates (g2, g3, g4, g5) at EIGHT lanes per item (P_EXPC_SQ), the powers 2^16, 2^48, 2^57 decompressed around one Fp inversion per item
// (P_EXPC_DEC_A -> inversion kernel -> P_EXPC_DEC_B, which also squares on to 2^60, 2^62, 2^63 and multiplies the six powers).  15 % fewer instructions
// per item, and measured no faster (profiles/round3_pmc_expc.csv, round3_expc_ab.txt; tools/pmc_expc.sh, tools/exp_expc.sh): alone at 65,536 items
// 1.18 + 0.17 + 1.00 ms + a 0.16 ms inversion launch against 2.45 ms for P_EXPX; twelve 4096-batches in flight 2.67 against 2.70 M pairings/s.  The squaring
// program issues at EXPX's rate; the decompression program keeps 50 slots live (two wavefronts per SIMD) and the exponent's set bits are too spread for a
// compressed form that cannot multiply.  Kept because it is correct on every input and answers the question whether it pays; the
// decompression divides by g2: an item with a vanishing g2 (the unit element, or a crafted
// input) is flagged by DEC_B and recomputed by the plain program over an index list kept on the device, so the result is the reference's for every input.
static int expx(nbls_ctx* ctx, size_t n, uint8_t* in, uint8_t* out, hipStream_t s) {
  int r;
  if (n < ctx->expc_min) return run(ctx, ls_variant(ctx, P_EXPX, n), n, {B(3, in, F12), B(5, out, F12)}, s);
  const size_t KSB = (size_t)EXPC_SQ_ELEMS * RAW, KDB = (size_t)EXPC_DEC_ELEMS * RAW;
  if ((r = ensure_expc_scratch(ctx))) return r;
  if ((r = run(ctx, P_EXPC_SQ, n, {B(3, in, F12), B(5, ctx->KS, KSB)}, s))) return r;
  if ((r = run(ctx, P_EXPC_DEC_A, n, {B(3, ctx->KS, KSB), B(4, ctx->N, RAW), B(5, ctx->KD, KDB)}, s))) return r;
  if ((r = run_inv(ctx, n, s))) return r;                 // N / NI are free once FE_EASY has run
  if ((r = run(ctx, P_EXPC_DEC_B, n, {B(3, ctx->KS, KSB), B(4, ctx->NI, RAW), B(6, ctx->KD, KDB), B(5, out, F12), B(7, ctx->Kflag, 1)}, s))) return r;
  uint32_t* count = ctx->Kcount + (ctx->ioff ? 1 : 0);    // the two halves of a split call run concurrently
  uint32_t* list = ctx->Klist + ctx->ioff;
  if (nbls_flag_compact_launch((unsigned)n, ctx->Kflag + ctx->ioff, list, count, s)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
  return run(ctx, P_EXPX, n, {B(3, in, F12), B(5, out, F12)}, s, count, list);   // workgroups beyond the listed items exit at once
}
// n raw Fp12 in `f_raw` (norms already in ctx->N) -> finalExponentiate -> wire bytes at d_out (math.ts:856-874)
static int final_exp_pipeline(nbls_ctx* ctx, size_t n, uint8_t* f_raw, void* d_out, hipStream_t s) {
  int r;
  uint8_t** T = ctx->T;
  if ((r = run_inv(ctx, n, s))) return r;
  if ((r = run(ctx, P_FE_EASY, n, {B(3, f_raw, F12), B(4, ctx->NI, RAW), B(5, T[0], F12)}, s))) return r;
  if (n < ctx->expc_min && n < ctx->chain_max && !ctx->in_halves && ls_variant(ctx, P_EXPX, n) == P_EXPX && !wide_applies(ctx, ctx->prog[P_EXPX], (int)P_EXPX, n)) {
    // the seven launches between the easy part and the final product as one chain (math.ts:862-867): t2 = t1^x, t3 = conj(t1^2) t2, t4 = t3^x, t5 = t4^x,
    // t6' = t5^x, t6 = t6' t2^2, t7 = t6^x
    if ((r = run_chain(ctx, n, {{P_EXPX, {B(3, T[0], F12), B(5, T[1], F12)}},
                                {P_FE_MID1, {B(3, T[0], F12), B(5, T[1], F12), B(6, T[2], F12)}},
                                {P_EXPX, {B(3, T[2], F12), B(5, T[3], F12)}},
                                {P_EXPX, {B(3, T[3], F12), B(5, T[4], F12)}},
                                {P_EXPX, {B(3, T[4], F12), B(5, T[6], F12)}},
                                {P_FE_MID2, {B(3, T[6], F12), B(5, T[1], F12), B(6, T[5], F12)}},
                                {P_EXPX, {B(3, T[5], F12), B(5, T[6], F12)}}}, s))) return r;
    return run(ctx, P_FE_FINAL, n, {B(0, T[0], F12), B(1, T[1], F12), B(2, T[2], F12), B(3, T[3], F12), B(4, T[4], F12), B(5, T[5], F12), B(6, T[6], F12), B(7, d_out, 576)}, s);
  }
  if ((r = expx(ctx, n, T[0], T[1], s))) return r;   // t2
  if ((r = run(ctx, P_FE_MID1, n, {B(3, T[0], F12), B(5, T[1], F12), B(6, T[2], F12)}, s))) return r;   // t3
  if ((r = expx(ctx, n, T[2], T[3], s))) return r;   // t4
  if ((r = expx(ctx, n, T[3], T[4], s))) return r;   // t5
  if ((r = expx(ctx, n, T[4], T[6], s))) return r;   // t6' (parked in T7's buffer)
  if ((r = run(ctx, P_FE_MID2, n, {B(3, T[6], F12), B(5, T[1], F12), B(6, T[5], F12)}, s))) return r;   // t6
  if ((r = expx(ctx, n, T[5], T[6], s))) return r;   // t7
  return run(ctx, P_FE_FINAL, n, {B(0, T[0], F12), B(1, T[1], F12), B(2, T[2], F12), B(3, T[3], F12), B(4, T[4], F12), B(5, T[5], F12), B(6, T[6], F12), B(7, d_out, 576)}, s);
}
// one raw Fp12 -> final exponentiation (or plain encoding) -> wire bytes on device
static int finish_single(nbls_ctx* ctx, uint8_t* f_raw, int final_exp, void* d_out, hipStream_t s) {
  int r;
  if (!final_exp) return run(ctx, P_RAW_TO_BYTES, 1, {B(3, f_raw, F12), B(2, d_out, 576)}, s);
  if ((r = run(ctx, P_NORM_RAW, 1, {B(3, f_raw, F12), B(4, ctx->N, RAW)}, s))) return r;
  return final_exp_pipeline(ctx, 1, f_raw, d_out, s);
}

EXPORT int nbls_init(int device_id, nbls_ctx** out) {
  if (!out) return NBLS_EINVAL;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return NBLS_ENOGPU;
  if (device_id < 0 || device_id >= count) return NBLS_EINVAL;
  nbls_ctx* ctx = new nbls_ctx();
  ctx->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess) { delete ctx; return NBLS_ENOGPU; }
  if (hipStreamCreate(&ctx->stream) != hipSuccess) { delete ctx; return NBLS_EHIP; }
  if (hipMalloc(&ctx->qp_table, (size_t)QP_TABLE_ENTRIES * RAW_WORDS * 4) != hipSuccess || hipMemcpy(ctx->qp_table, qp_table_words(), (size_t)QP_TABLE_ENTRIES * RAW_WORDS * 4, hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return NBLS_EHIP; }
  {
    std::vector<u32> ul((size_t)LINE_ELEMS * RAW_WORDS, 0);
    for (int j = 0; j < N_LINES; j++) memcpy(&ul[(size_t)6 * j * RAW_WORDS], NBLS_R1, NLIMBS * 4);   // c0.c0 = 1 in Montgomery form
    if (hipMalloc(&ctx->unit_lines, LINE_BYTES) != hipSuccess || hipMemcpy(ctx->unit_lines, ul.data(), LINE_BYTES, hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return NBLS_EHIP; }
  }
  // Montgomery one as a raw Fp12 (pads odd-sized product reductions)
  u32 one[12 * SLOT_WORDS]; memset(one, 0, sizeof one); memcpy(one, NBLS_R1, NLIMBS * 4);
  if (hipMalloc(&ctx->one12, F12) != hipSuccess || hipMemcpy(ctx->one12, one, F12, hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return NBLS_EHIP; }
  {
    const uint64_t* exps[4] = {NBLS_EXP_P_PLUS_1_DIV_4, NBLS_EXP_P2_PLUS_7_DIV_16, NBLS_EXP_P2_MINUS_9_DIV_16, NBLS_EXP_P_MINUS_3_DIV_4};
    const int bits[4] = {NBLS_P_PLUS_1_DIV_4_BITS, NBLS_P2_PLUS_7_DIV_16_BITS, NBLS_P2_MINUS_9_DIV_16_BITS, NBLS_P_MINUS_3_DIV_4_BITS};
    for (int k = 0; k < 4; k++) {
      // op lists of the exponents (pow_exec.h: sliding windows).  The two Fp2 exponents are (K p + 11 K + 8) and (K p + 11 K + 7) with K = (p - 11) / 16: the kernel
      // raises conj(a) a^11 to K (pow_kernels.hip), so both get the op list of K
      std::vector<unsigned char> ops;
      if (k == 1 || k == 2) {
        uint64_t K[6]; for (int j = 0; j < 6; j++) K[j] = NBLS_EXP_P_MINUS_3_DIV_4[j];
        K[0] -= 2;                                                        // (p - 3) / 4 - 2 = (p - 11) / 4 (no borrow: the low word ends in ...aaaa)
        for (int j = 0; j < 6; j++) K[j] = (K[j] >> 2) | (j < 5 ? K[j + 1] << 62 : 0);   // / 4
        ops = pow_make_ops(K, 377);
      } else ops = pow_make_ops(exps[k], bits[k]);
      ctx->nnib[k] = (int)(ops.size() / 2);
      if (hipMalloc(&ctx->nib[k], ops.size()) != hipSuccess || hipMemcpy(ctx->nib[k], ops.data(), ops.size(), hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return NBLS_EHIP; }
    }
    // -G1 in wire form: x || (p - y)   (standard integers, big-endian)
    uint8_t ng[96];
    auto be = [](uint8_t* o, const u32* limbs) { u32 w[12]; limbs_to_words(w, limbs); for (int i = 0; i < 12; i++) { u32 v = w[11 - i]; o[4 * i] = v >> 24; o[4 * i + 1] = v >> 16; o[4 * i + 2] = v >> 8; o[4 * i + 3] = v; } };
    be(ng, NBLS_G1X_RAW); be(ng + 48, NBLS_NEG_G1Y_RAW);
    uint8_t gg[96]; be(gg, NBLS_G1X_RAW); be(gg + 48, NBLS_G1Y_RAW);
    if (hipMalloc(&ctx->gen_g1, 96) != hipSuccess || hipMemcpy(ctx->gen_g1, gg, 96, hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return NBLS_EHIP; }
    u32 id1[3 * SLOT_WORDS] = {0}, id2[6 * SLOT_WORDS] = {0}; memcpy(id1 + SLOT_WORDS, NBLS_R1, NLIMBS * 4); memcpy(id2 + 2 * SLOT_WORDS, NBLS_R1, NLIMBS * 4);   // (0 : 1 : 0)
    if (hipMalloc(&ctx->neg_g1, 96) != hipSuccess || hipMemcpy(ctx->neg_g1, ng, 96, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&ctx->ident_g1, 3 * RAW) != hipSuccess || hipMemcpy(ctx->ident_g1, id1, 3 * RAW, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&ctx->ident_g2, 6 * RAW) != hipSuccess || hipMemcpy(ctx->ident_g2, id2, 6 * RAW, hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return NBLS_EHIP; }
  }
  for (int i = 0; i < P_COUNT; i++) { if (i == P_G1_MUL || i == P_G2_MUL || i == P_G1_MUL_W3 || i == P_G2_MUL_W3 || i == P_G1_MUL_FIXED || i == P_G2_MUL_GLS || i == P_G2_MUL_SAC || i == P_G2_MUL_SAC_LS2) continue;   // the scalar-multiplication ladders are uploaded on first use
    int r = upload(ctx, (ProgId)i); if (r) { int e = ctx->last_hip; (void)e; nbls_destroy(ctx); return r; } }
  *out = ctx;
  return NBLS_OK;
}

EXPORT void nbls_destroy(nbls_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  if (ctx->dst_dev) hipFree(ctx->dst_dev);
  for (auto& kv : ctx->tower) free_program(kv.second);
  for (auto& d : ctx->prog) free_program(d);
  for (uint8_t* p : {ctx->F, ctx->N, ctx->NI, ctx->io_g1, ctx->io_g2, ctx->io_f12, ctx->one12, ctx->gen_g1, ctx->g1_fixed, ctx->side_scratch, ctx->L, ctx->partial, ctx->unit_lines, ctx->KS, ctx->KD, ctx->Kflag, (uint8_t*)ctx->Klist, (uint8_t*)ctx->Kcount}) if (p) hipFree(p);
  for (uint8_t* p : ctx->T) if (p) hipFree(p);
  for (uint8_t* p : ctx->sb) if (p) hipFree(p);
  for (auto& b : ctx->io_pool) if (b.p) hipFree(b.p);
  for (uint8_t* p : ctx->nib) if (p) hipFree(p);
  if (ctx->qp_table) hipFree(ctx->qp_table);
  for (uint8_t* p : {ctx->neg_g1, ctx->ident_g1, ctx->ident_g2}) if (p) hipFree(p);
  if (ctx->side) hipStreamDestroy(ctx->side);
  if (ctx->half_stream) hipStreamDestroy(ctx->half_stream);
  for (hipEvent_t e : {ctx->ev_half_fork, ctx->ev_half_join}) if (e) hipEventDestroy(e);
  if (ctx->side2) hipStreamDestroy(ctx->side2);
  for (hipEvent_t e : {ctx->ev_fork, ctx->ev_join, ctx->ev_join2, ctx->ev_last, ctx->ev_pipe_done}) if (e) hipEventDestroy(e);
  for (hipEvent_t e : ctx->pipe_ev) hipEventDestroy(e);
  for (hipStream_t st : ctx->pipe_streams) hipStreamDestroy(st);
  for (auto& t : ctx->tev) { hipEventDestroy(t.second.first); hipEventDestroy(t.second.second); }
  for (hipEvent_t e : ctx->ev_pool) hipEventDestroy(e);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

EXPORT const char* nbls_strerror(int code) {
  switch (code) {
    case NBLS_OK: return "ok";
    case NBLS_EINVAL: return "invalid argument";
    case NBLS_EHIP: return "HIP runtime error";
    case NBLS_ENOSUP: return "not supported in this build";
    case NBLS_ENOGPU: return "no usable GPU";
    case NBLS_EDECODE: return "input point failed to decode";
    default: return "unknown error";
  }
}
EXPORT int nbls_last_hip_error(nbls_ctx* ctx) { return ctx ? ctx->last_hip : 0; }
EXPORT int nbls_device_synchronize(nbls_ctx* ctx) { if (!ctx) return NBLS_EINVAL; HIPCHK(hipSetDevice(ctx->device)); HIPCHK(hipDeviceSynchronize()); return NBLS_OK; }

static int pairing_core(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_g2, int with_final_exp, void* d_out, hipStream_t s, bool two_programs);
EXPORT int nbls_pairing_batch_dev(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_g2, int with_final_exp, void* d_out, void* stream) {
  if (!ctx || (n && (!d_g1 || !d_g2 || !d_out))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  StreamOrder order_(ctx, s);
  // A batch of 8192 pairs or more runs as two halves on two streams: every launch of a dependent chain ends in a partly filled round of wavefronts (EXPX at 65,536 pairs: 4.65
  // rounds of 2,816 resident wavefronts), and the tail of one half is filled by the other (65,536 pairs: 27.4 -> 25.9 ms).  Both halves use the caller's scratch
  // through an item offset (ctx->ioff, applied by run() to every per-item buffer) and the two-program Miller loop (what counts with work in flight is the instruction count).
  // (measured from 8192 pairs up: 8192 5.08 -> 4.69 ms, 16,384 8.53 -> 7.73, 24,576 11.96 -> 10.33, 32,768 14.96 -> 13.64, 65,536 27.5 -> 26.0; the exception is a batch that
  // fills the chip exactly three wavefronts deep in ONE round with the fused program, 12,288 pairs: 6.04 ms against 6.39)
  const bool one_full_round = n > 10752 && n <= 12288;
  if (n >= ctx->halves_min && !one_full_round && n <= LINES_CHUNK) {
    int r;
    if ((r = ensure_lines(ctx, n))) return r;
    if (with_final_exp && (r = ensure_scratch(ctx, n))) return r;
    if (!ctx->half_stream && (hipStreamCreateWithFlags(&ctx->half_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_half_fork, hipEventDisableTiming) != hipSuccess ||
                              hipEventCreateWithFlags(&ctx->ev_half_join, hipEventDisableTiming) != hipSuccess)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
    static const size_t split_pct = (size_t)std::min<long>(99, std::max<long>(1, env_long("NBLS_HALVES_SPLIT_PCT", 55)));      // size of the first half in per cent (clamped to 1 .. 99): slightly unequal halves do not run phase-locked (profiles/round5_ab_split.txt: 16,384 pairs 6.25 -> 6.14 ms, 65,536 within noise)
    const size_t h = ((n * split_pct / 100) + 63) & ~(size_t)63;
    if (h > 0 && h < n) {      // (a split that leaves one side empty -- rounding at a small n -- falls through to the single-stream path)
      HIPCHK(hipEventRecord(ctx->ev_half_fork, s)); HIPCHK(hipStreamWaitEvent(ctx->half_stream, ctx->ev_half_fork, 0));
      ForkGuard fork_guard;
      ctx->in_halves = true;     // (round 4 compared each HALF with chain_max: calls of 8192..16383 pairs ran their halves chained, the configuration measured as slower)
      r = pairing_core(ctx, h, d_g1, d_g2, with_final_exp, d_out, s, true);
      if (!r) { ctx->ioff = h; r = pairing_core(ctx, n - h, d_g1, d_g2, with_final_exp, d_out, ctx->half_stream, true); ctx->ioff = 0; }
      ctx->in_halves = false;
      HIPCHK(hipEventRecord(ctx->ev_half_join, ctx->half_stream)); HIPCHK(hipStreamWaitEvent(s, ctx->ev_half_join, 0));
      if (!r) fork_guard.armed = false;      // joined into s; a failed half leaves work in flight on both streams: the guard waits for it
      return r;
    }
  }
  return pairing_core(ctx, n, d_g1, d_g2, with_final_exp, d_out, s, false);
}
static int pairing_core(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_g2, int with_final_exp, void* d_out, hipStream_t s, bool two_programs) {
  int r;
  // One program or two?  LINES + ACC execute ~12 % fewer instructions per pairing (no idle lanes in the Fp12 steps, 20 instead of 37 lane-ops
  // per bit in the point chain) but are two dependent chains of 307 + 173 steps where the fused program has 349: a launch that is only a few
  // wavefronts per SIMD deep takes the time of its longest instruction stream: round 3 (the interpreter) kept the fused program below 49,152 pairs; with the
  // ahead-of-time kernels the two programs win from split_min = 4096 pairs on (SPLIT_MILLER_MIN above, tools/sweep_modes.sh), and with several calls in flight
  // the instruction count is what matters at every size (nbls_pool_init sets the threshold to 0).  NBLS_FUSED_MILLER = 1 / 0 forces one or the other.
  static const int fused_mode = (int)env_long("NBLS_FUSED_MILLER", -1);
  const bool fused = fused_mode >= 0 ? fused_mode != 0 : (!two_programs && n < ctx->split_min);
  if (fused) {
    if (!with_final_exp) return run(ctx, ls_variant(ctx, P_MILLER_BYTES, n), n, {B(0, d_g1, 96), B(1, d_g2, 192), B(2, d_out, 576)}, s);
    if ((r = ensure_scratch(ctx, n))) return r;
    if ((r = run(ctx, ls_variant(ctx, P_MILLER_FE, n), n, {B(0, d_g1, 96), B(1, d_g2, 192), B(3, ctx->F, F12), B(4, ctx->N, RAW)}, s))) return r;
    return final_exp_pipeline(ctx, n, ctx->F, d_out, s);
  }
  // calcPairingPrecomputes + millerLoop (math.ts:1331-1388) as two programs: line tables through HBM (LINE_BYTES per pair)
  if ((r = ensure_lines(ctx, n))) return r;
  if (with_final_exp && (r = ensure_scratch(ctx, n))) return r;
  for (size_t o = 0; o < n; o += LINES_CHUNK) {
    const size_t c = n - o < LINES_CHUNK ? n - o : LINES_CHUNK;
    const uint8_t *g1 = (const uint8_t*)d_g1 + o * 96, *g2 = (const uint8_t*)d_g2 + o * 192;
    if ((r = run(ctx, P_LINES_PQ, c, {B(0, g1, 96), B(1, g2, 192), B(3, ctx->L, LINE_BYTES)}, s))) return r;
    if (!with_final_exp) r = run(ctx, P_ACC_BYTES, c, {B(3, ctx->L, LINE_BYTES), B(2, (uint8_t*)d_out + o * 576, 576)}, s);
    else r = run(ctx, P_ACC_FE, c, {B(3, ctx->L, LINE_BYTES), B(5, ctx->F + o * F12, F12), B(4, ctx->N + o * RAW, RAW)}, s);
    if (r) return r;
  }
  if (!with_final_exp) return NBLS_OK;
  return final_exp_pipeline(ctx, n, ctx->F, d_out, s);
}

EXPORT int nbls_pairing_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, int with_final_exp, int validate, uint8_t* out, int8_t* status) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);   // scratch and I/O staging buffers belong to this call until it returns
  if (!ctx || (n && (!g1 || !g2 || !out))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  int r;
  std::vector<int8_t> st1, st2;
  if (validate) {   // P.assertValidity(); Q.assertValidity()  (index.ts:717-718)
    st1.resize(n); st2.resize(n);
    if ((r = nbls_g1_validate_batch(ctx, n, g1, st1.data())) || (r = nbls_g2_validate_batch(ctx, n, g2, st2.data()))) return r;
  }
  {
    std::lock_guard<std::recursive_mutex> g(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    if ((r = ensure_io(ctx, n))) return r;
    HIPCHK(hipMemcpyAsync(ctx->io_g1, g1, n * 96, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->io_g2, g2, n * 192, hipMemcpyHostToDevice, ctx->stream));
  }
  if ((r = nbls_pairing_batch_dev(ctx, n, ctx->io_g1, ctx->io_g2, with_final_exp, ctx->io_f12, ctx->stream))) return r;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipMemcpyAsync(out, ctx->io_f12, n * 576, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (status) memset(status, 0, n);
  if (validate) for (size_t i = 0; i < n; i++) {
    int8_t c = st1[i] ? st1[i] : (st2[i] ? (int8_t)(10 + st2[i]) : 0);
    if (c) { memset(out + 576 * i, 0, 576); if (status) status[i] = c; }
  }
  return NBLS_OK;
}

// n >= 1 pairs -> *m_out raw Miller values (products of up to eight Miller loops each) in ctx->F[0 .. *m_out); the caller multiplies them (reduce_product)
static int miller_values(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_g2, size_t* m_out, hipStream_t s) {
  int r;
  {
    // pairs are taken two at a time with a shared accumulator (one Fp12 squaring per bit for both); an odd last pair runs alone
    const size_t n2 = n / 2; size_t m = n2 + (n & 1);
    static const int fused_mode = (int)env_long("NBLS_FUSED_MILLER", -1);
    const bool fused = fused_mode >= 0 ? fused_mode != 0 : n < ctx->split_min;
    if (fused && n <= 4096) {
      // up to one wavefront per SIMD (4 pairs per wavefront): the call takes the time of ONE wavefront's instruction stream whatever it computes, so every
      // pair gets an item of its own (420 k instructions) rather than sharing an accumulator with a second one (630 k): a single verify 6.1 -> 5.5 ms
      m = n;
      if ((r = run(ctx, ls_variant(ctx, P_MILLER_RAW, n), n, {B(0, d_g1, 96), B(1, d_g2, 192), B(3, ctx->F, F12)}, s))) return r;
    } else if (fused) {
      if (n2 && (r = run(ctx, P_MILLER_RAW2, n2, {B(0, d_g1, 192), B(1, d_g2, 384), B(3, ctx->F, F12)}, s))) return r;
      if ((n & 1) && (r = run(ctx, P_MILLER_RAW, 1, {B(0, (const uint8_t*)d_g1 + (n - 1) * 96, 96), B(1, (const uint8_t*)d_g2 + (n - 1) * 192, 192), B(3, ctx->F + n2 * F12, F12)}, s))) return r;
    } else {
      // four pairs per item with ONE accumulator: f <- (f l1 l2 l3 l4)^2 per bit, a single Fp12 squaring for four Miller loops (16 % fewer
      // products per pair than two per item).  A last group of fewer than four pairs is filled up with the unit table (every line = 1:
      // multiplying by it changes nothing) instead of getting a launch -- and the latency of a whole Miller loop -- of its own.
      // round 4: EIGHT pairs per accumulator from acc8_min pairs per call (one squaring per eight line tables: 1,921 instead of 2,196 instructions per pair and bit)
      const size_t GR = n >= ctx->acc8_min ? 8 : 4;
      const ProgId acc = GR == 8 ? P_ACC8_RAW : P_ACC4_RAW;
      if ((r = ensure_lines(ctx, n + GR - 1))) return r;
      m = 0;
      for (size_t o = 0; o < n; o += LINES_CHUNK) {   // LINES_CHUNK is a multiple of eight: a chunk boundary never splits a group
        const size_t c = n - o < LINES_CHUNK ? n - o : LINES_CHUNK, cg = (c + GR - 1) / GR;
        // a large chunk runs as two halves (whole groups) on two streams, like nbls_pairing_batch_dev: the tail of LINES / ACC of one half under the other
        const size_t h = (c >= ctx->halves_min && ctx->ioff == 0) ? ((c / 2 + GR - 1) & ~(GR - 1)) : c;
        if (h < c && !ctx->half_stream && (hipStreamCreateWithFlags(&ctx->half_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_half_fork, hipEventDisableTiming) != hipSuccess ||
                                           hipEventCreateWithFlags(&ctx->ev_half_join, hipEventDisableTiming) != hipSuccess)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
        if (h < c) { HIPCHK(hipEventRecord(ctx->ev_half_fork, s)); HIPCHK(hipStreamWaitEvent(ctx->half_stream, ctx->ev_half_fork, 0)); }
        for (size_t lo = 0; lo < c; lo += h) {
          const size_t cc = lo ? c - lo : h, gg = (cc + GR - 1) / GR;      // two parts: [0, h) and everything behind it
          hipStream_t sh = lo ? ctx->half_stream : s;
          if ((r = run(ctx, P_LINES_PQ, cc, {B(0, (const uint8_t*)d_g1 + (o + lo) * 96, 96), B(1, (const uint8_t*)d_g2 + (o + lo) * 192, 192), B(3, ctx->L + lo * LINE_BYTES, LINE_BYTES)}, sh))) return r;
          for (size_t k = cc; k < GR * gg; k++) HIPCHK(hipMemcpyAsync(ctx->L + (lo + k) * LINE_BYTES, ctx->unit_lines, LINE_BYTES, hipMemcpyDeviceToDevice, sh));
          if ((r = run(ctx, acc, gg, {B(3, ctx->L + lo * LINE_BYTES, GR * LINE_BYTES), B(5, ctx->F + (m + lo / GR) * F12, F12)}, sh))) return r;
          if (lo) break;
        }
        if (h < c) { HIPCHK(hipEventRecord(ctx->ev_half_join, ctx->half_stream)); HIPCHK(hipStreamWaitEvent(s, ctx->ev_half_join, 0)); }
        m += cg;
      }
    }
    *m_out = m;
  }
  return NBLS_OK;
}
EXPORT int nbls_miller_product_dev(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_g2, int final_exp, void* d_out, void* stream) {
  if (!ctx || !d_out || (n && (!d_g1 || !d_g2))) return NBLS_EINVAL;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  StreamOrder order_(ctx, s);
  int r;
  if ((r = ensure_scratch(ctx, n ? n : 1))) return r;
  uint8_t* res = ctx->F;
  if (n == 0) { HIPCHK(hipMemcpyAsync(ctx->F, ctx->one12, F12, hipMemcpyDeviceToDevice, s)); }
  else {
    size_t m = 0;
    if ((r = miller_values(ctx, n, d_g1, d_g2, &m, s))) return r;
    if ((r = reduce_product(ctx, m, &res, s))) return r;
  }
  return finish_single(ctx, res, final_exp, d_out, s);
}

EXPORT int nbls_miller_product(nbls_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, int final_exp, int validate, uint8_t* out, int8_t* status) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);   // scratch and I/O staging buffers belong to this call until it returns
  if (!ctx || !out || (n && (!g1 || !g2))) return NBLS_EINVAL;
  int r;
  if (validate && n) {
    std::vector<int8_t> st1(n), st2(n); bool bad = false;
    if ((r = nbls_g1_validate_batch(ctx, n, g1, st1.data())) || (r = nbls_g2_validate_batch(ctx, n, g2, st2.data()))) return r;
    for (size_t i = 0; i < n; i++) { int8_t c = st1[i] ? st1[i] : (st2[i] ? (int8_t)(10 + st2[i]) : 0); if (status) status[i] = c; bad = bad || c; }
    if (bad) { memset(out, 0, 576); return NBLS_EDECODE; }
  }
  {
    std::lock_guard<std::recursive_mutex> g(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    if ((r = ensure_io(ctx, n ? n : 1))) return r;
    if (n) {
      HIPCHK(hipMemcpyAsync(ctx->io_g1, g1, n * 96, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(ctx->io_g2, g2, n * 192, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  if ((r = nbls_miller_product_dev(ctx, n, ctx->io_g1, ctx->io_g2, final_exp, ctx->io_f12, ctx->stream))) return r;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipMemcpyAsync(out, ctx->io_f12, 576, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (status) memset(status, 0, n);
  return NBLS_OK;
}

EXPORT int nbls_final_exp_batch_dev(nbls_ctx* ctx, size_t n, const void* d_in, void* d_out, void* stream) {
  if (!ctx || (n && (!d_in || !d_out))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  StreamOrder order_(ctx, s);
  int r;
  if ((r = ensure_scratch(ctx, n))) return r;
  if ((r = run(ctx, P_NORM_BYTES, n, {B(2, d_in, 576), B(3, ctx->F, F12), B(4, ctx->N, RAW)}, s))) return r;
  return final_exp_pipeline(ctx, n, ctx->F, d_out, s);
}

EXPORT int nbls_final_exp_batch(nbls_ctx* ctx, size_t n, const uint8_t* in, uint8_t* out) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);   // scratch and I/O staging buffers belong to this call until it returns
  if (!ctx || (n && (!in || !out))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  int r;
  {
    std::lock_guard<std::recursive_mutex> g(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    if ((r = ensure_io(ctx, 2 * n))) return r;
    HIPCHK(hipMemcpyAsync(ctx->io_f12, in, n * 576, hipMemcpyHostToDevice, ctx->stream));
  }
  uint8_t* d_out = ctx->io_f12 + n * 576;
  if ((r = nbls_final_exp_batch_dev(ctx, n, ctx->io_f12, d_out, ctx->stream))) return r;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipMemcpyAsync(out, d_out, n * 576, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return NBLS_OK;
}

// One tower operation on n elements (include/nbls.h): wire bytes in and out, everything on the device.  Inversions are two programs around the inversion kernel.
EXPORT int nbls_tower_op_batch(nbls_ctx* ctx, int field, int op, int param, size_t n, const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d, uint8_t* out) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);
  if (!ctx || (n && (!a || !out))) return NBLS_EINVAL;
  const Program* p0 = get_tower_program(field, op, param, 0);
  if (!p0) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  StreamOrder order_(ctx, s);
  const size_t esz = 48 * (size_t)field;
  // operand sizes: b is a full element for the binary operations, an Fp2 for the sparse products; c, d are Fp2
  const bool sparse = op == 10 || op == 11 || op == 12;
  const size_t bsz = b ? (sparse ? 96 : esz) : 0, csz = c ? 96 : 0, dsz = d ? 96 : 0;
  if ((p0->buf_extent[1] && !b) || (p0->buf_extent[2] && !c) || (p0->buf_extent[3] && !d)) return NBLS_EINVAL;
  int r;
  if ((r = ensure_scratch(ctx, n))) return r;
  const size_t need = n * (2 * esz + bsz + csz + dsz);
  uint8_t* io = nullptr;
  HIPCHK(hipMalloc(&io, need));
  uint8_t *da = io, *db = da + n * esz, *dc = db + n * bsz, *dd = dc + n * csz, *dout = dd + n * dsz;
  auto fail = [&](int code) { hipFree(io); return code; };
  if (hipMemcpyAsync(da, a, n * esz, hipMemcpyHostToDevice, s) != hipSuccess) return fail(NBLS_EHIP);
  if (b && hipMemcpyAsync(db, b, n * bsz, hipMemcpyHostToDevice, s) != hipSuccess) return fail(NBLS_EHIP);
  if (c && hipMemcpyAsync(dc, c, n * csz, hipMemcpyHostToDevice, s) != hipSuccess) return fail(NBLS_EHIP);
  if (d && hipMemcpyAsync(dd, d, n * dsz, hipMemcpyHostToDevice, s) != hipSuccess) return fail(NBLS_EHIP);
  auto launch = [&](int part) -> int {
    const Program* p = get_tower_program(field, op, param, part);
    if (!p) return NBLS_EINVAL;
    DevProgram& dp = ctx->tower[std::make_tuple(field, op, param, part)];
    if (!dp.p) { const int e = upload_program(ctx, dp, *p, -1); if (e) return e; }
    return run_dev(ctx, dp, -1, n, {B(0, da, esz), B(1, db, bsz), B(2, dc, csz), B(3, dd, dsz), B(4, ctx->N, RAW), B(5, ctx->NI, RAW), B(7, dout, esz)}, s, nullptr, nullptr);
  };
  if ((r = launch(0))) return fail(r);
  if (op == 5) {   // NBLS_TOP_INV
    if ((r = run_inv(ctx, n, s))) return fail(r);
    if ((r = launch(1))) return fail(r);
  }
  if (hipMemcpyAsync(out, dout, n * esz, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail(NBLS_EHIP);
  hipFree(io);
  return NBLS_OK;
}

// n Fp12 wire elements on the device -> their product, optionally final-exponentiated (multi-GPU: partials of all ranks)
EXPORT int nbls_fp12_product_final_dev(nbls_ctx* ctx, size_t n, const void* d_in, int final_exp, void* d_out, void* stream) {
  if (!ctx || !d_out || (n && !d_in)) return NBLS_EINVAL;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  StreamOrder order_(ctx, s);
  int r;
  if ((r = ensure_scratch(ctx, n ? n : 1))) return r;
  uint8_t* res = ctx->F;
  if (n == 0) { HIPCHK(hipMemcpyAsync(ctx->F, ctx->one12, F12, hipMemcpyDeviceToDevice, s)); }
  else {
    // wire bytes -> raw Montgomery (P_NORM_BYTES also writes N, which is ignored here)
    if ((r = run(ctx, P_NORM_BYTES, n, {B(2, d_in, 576), B(3, ctx->F, F12), B(4, ctx->N, RAW)}, s))) return r;
    if ((r = reduce_product(ctx, n, &res, s))) return r;
  }
  return finish_single(ctx, res, final_exp, d_out, s);
}

// Placement study: runs the EXPX program on n scratch items and returns, per workgroup, three words: HW_ID | XCC_ID << 32 of its wavefront, start and end tick (s_memtime).
EXPORT int nbls_placement_probe(nbls_ctx* ctx, size_t n, uint64_t* out_blocks) {
  if (!ctx || !n || !out_blocks) return NBLS_EINVAL;
  std::lock_guard<std::recursive_mutex> g(ctx->mu); HIPCHK(hipSetDevice(ctx->device));
  int r = ensure_scratch(ctx, n); if (r) return r;
  if ((r = ensure_io(ctx, n))) return r;
  const ProgId pid = env_set("NBLS_PROBE_MILLER") ? P_MILLER_FE : P_EXPX;   // NBLS_PROBE_MILLER: probe the (4x longer) Miller program instead
  r = upload(ctx, pid); if (r) return r;
  const DevProgram& d = ctx->prog[pid];
  const size_t blocks = (n + d.p->G - 1) / d.p->G;
  uint64_t* dbg = nullptr; HIPCHK(hipMalloc(&dbg, blocks * 40));
  KernelArgs ka; memset(&ka, 0, sizeof ka);
  ka.steps = d.steps; ka.descs = d.descs; ka.consts = d.consts; ka.qp_table = ctx->qp_table;
  ka.nsteps = (u32)d.p->steps.size(); ka.nconst = d.p->nconst; ka.W = d.p->W; ka.G = d.p->G; ka.slot_bytes = d.p->slot_bytes; ka.inst_bytes = d.p->inst_bytes(); ka.shared_consts = d.p->shared_consts ? 1u : 0u; ka.lsplit = d.p->lsplit; ka.n_items = (u32)n;
  ka.bufs[3].ptr = ctx->T[0]; ka.bufs[3].stride = F12; ka.bufs[5].ptr = ctx->T[1]; ka.bufs[5].stride = F12;
  if (pid == P_MILLER_FE) { ka.bufs[0].ptr = ctx->io_g1; ka.bufs[0].stride = 96; ka.bufs[1].ptr = ctx->io_g2; ka.bufs[1].stride = 192; ka.bufs[4].ptr = ctx->N; ka.bufs[4].stride = RAW; }
  ka.hwid_out = dbg;
  HIPCHK(hipMemsetAsync(ctx->T[0], 0, n * F12, ctx->stream));
  int e = nbls_vm_launch(&ka, d.p->lds_bytes(), ctx->stream);
  if (e) { hipFree(dbg); ctx->last_hip = e; return NBLS_EHIP; }
  HIPCHK(hipMemcpyAsync(out_blocks, dbg, blocks * 40, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream));
  hipFree(dbg);
  return NBLS_OK;
}
EXPORT int nbls_program_stats(nbls_ctx* ctx, int prog, uint32_t* o) {
  (void)ctx;
  if (prog < 0 || prog >= P_COUNT || !o) return NBLS_EINVAL;
  const Program& p = get_program((ProgId)prog);
  o[0] = (uint32_t)p.steps.size(); o[1] = p.n_dot_steps; o[2] = p.n_lin_steps; o[3] = p.n_dot_ops; o[4] = p.n_products; o[5] = p.n_lin_ops; o[6] = p.slots; o[7] = p.lds_bytes();
  return NBLS_OK;
}

// Per-kernel timing for the benchmark's roofline leg: enable, run, synchronise, then read accumulated milliseconds and
// launch counts per program (index P_COUNT = the inversion kernel).  ms/counts must hold P_COUNT+1 entries.
EXPORT int nbls_timing_enable(nbls_ctx* ctx, int on) {
  if (!ctx) return NBLS_EINVAL;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  for (auto& t : ctx->tev) { ctx->ev_pool.push_back(t.second.first); ctx->ev_pool.push_back(t.second.second); }
  ctx->tev.clear(); ctx->timing = on != 0; return NBLS_OK;
}
EXPORT int nbls_timing_read(nbls_ctx* ctx, float* ms, uint32_t* counts) {
  if (!ctx || !ms || !counts) return NBLS_EINVAL;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  for (int i = 0; i <= P_COUNT; i++) { ms[i] = 0; counts[i] = 0; }
  for (auto& t : ctx->tev) {
    HIPCHK(hipEventSynchronize(t.second.second));
    float m = 0; HIPCHK(hipEventElapsedTime(&m, t.second.first, t.second.second));
    ms[t.first] += m; counts[t.first]++;
    ctx->ev_pool.push_back(t.second.first); ctx->ev_pool.push_back(t.second.second);
  }
  ctx->tev.clear();
  return NBLS_OK;
}

// ================================================================================================================
// Validity, decoders, hash-to-G2, point sums, verifyBatch.  Device-side pipelines (dev_*) work on device pointers and
// enqueue on `s`; the exported wrappers stage host buffers.
// ================================================================================================================
static int dev_validate(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, void* d_status, hipStream_t s) {
  return g2 ? run(ctx, P_G2_VALIDATE, n, {B(1, d_pts, 192), B(7, d_status, 1)}, s) : run(ctx, P_G1_VALIDATE, n, {B(0, d_pts, 96), B(7, d_status, 1)}, s);
}
// PointG1.fromHex (48 B) / PointG2.fromSignature (96 B): compressed -> affine wire bytes + status
// slot0 / pow_slot: scratch-pool slots used (three from slot0, one for the exponentiation table), so that two chains can run on
// different streams at the same time
// mode (G2 only): 0 fromSignature 96 B, 1 fromSignature 192 B, 2 fromHex 96 B (no subgroup check, flag rules)
// io / ntot: the call works on items [io, io + n) of scratch arrays sized for ntot items (verify_pipeline: sub-batches of one call run side by side on slices of the same arrays)
static int dev_decompress(nbls_ctx* ctx, bool g2, size_t n, const void* d_in, void* d_out, void* d_status, hipStream_t s, int slot0 = 0, int pow_slot = 11, int mode = 0, size_t io = 0, size_t ntot = 0) {
  const size_t e = g2 ? (mode == 1 ? 192 : 96) : 48, q = g2 ? 2 * RAW : RAW, pq = POW_TAB * (g2 ? 2 : 1) * RAW;
  if (ntot < io + n) ntot = io + n;
  uint8_t *X, *R, *Cd, *pw; int r;
  if ((r = need(ctx, slot0, ntot * q, &X)) || (r = need(ctx, slot0 + 1, ntot * q, &R)) || (r = need(ctx, slot0 + 2, ntot * q, &Cd)) || (r = need(ctx, pow_slot, ntot * pq, &pw))) return r;
  X += io * q; R += io * q; Cd += io * q; pw += io * pq;
  const ProgId pa = !g2 ? P_G1_DEC_A : mode == 1 ? P_G2_DEC_A192 : P_G2_DEC_A, pb = !g2 ? P_G1_DEC_B : mode == 1 ? P_G2_DEC_B192 : mode == 2 ? P_G2_DEC_B_HEX : P_G2_DEC_B;
  if ((r = run(ctx, pa, n, {B(0, d_in, e), B(3, X, q), B(4, R, q)}, s))) return r;
  if ((r = run_pow(ctx, g2 ? 1 : 0, n, R, Cd, s, pw))) return r;
  return run(ctx, pb, n, {B(0, d_in, e), B(3, X, q), B(4, R, q), B(5, Cd, q), B(6, d_out, g2 ? 192 : 96), B(7, d_status, 1)}, s);
}
// 256 uniform bytes per message (expand_message_xmd output) -> hash point, affine wire bytes (PointG2.hashToCurve, index.ts:481-490)
// PointG2.clearCofactor (index.ts:659-672) on raw projective points: three programs: the t1-independent points, then one around each multiplication by x (programs.h P_H2C_C0 / C1 / C2).
// in -> out (may alias in or base), norm of Z -> N; base and S are scratch of n * 6 raw elements each, and `in` is scratch too from the second program on (t1 is stored over P)
static int dev_clear_g2(nbls_ctx* ctx, size_t n, void* in, uint8_t* base, uint8_t* S, void* out, void* N, hipStream_t s) {
  int r = run(ctx, P_H2C_C0, n, {B(3, in, 6 * RAW), B(6, base, 6 * RAW), B(5, S, 6 * RAW)}, s); if (r) return r;     // v = psi(P) -> base, u = psi^2(2P) - psi(P) - P -> S
  if ((r = run(ctx, pt_ls2_variant(ctx, P_H2C_C1, n), n, {B(3, in, 6 * RAW), B(6, base, 6 * RAW)}, s))) return r;                          // base = t1 + v over v, t1 = -[x]P over P
  return run(ctx, pt_ls2_variant(ctx, P_H2C_C2, n), n, {B(3, base, 6 * RAW), B(4, in, 6 * RAW), B(5, S, 6 * RAW), B(6, out, 6 * RAW), B(7, N, RAW)}, s);   // out may be in: every item reads its t1 before its result is stored
}
static int dev_hash_to_g2(nbls_ctx* ctx, size_t n, const void* d_uniform, void* d_out, hipStream_t s, size_t io = 0, size_t ntot = 0, uint8_t** proj = nullptr) {   // io / ntot: see dev_decompress; proj: stop at the raw projective points (scratch slot 1) -- the caller multiplies them (sign) and normalises once, at the end
  if (ntot < io + n) ntot = io + n;
  uint8_t *T, *E, *Pw, *Q, *N, *NI, *st, *St, *Pt2, *S, *tab; int r;
  if ((r = need(ctx, 0, ntot * 4 * RAW, &T)) || (r = need(ctx, 1, ntot * 6 * RAW, &E)) || (r = need(ctx, 2, ntot * 4 * RAW, &Pw)) || (r = need(ctx, 3, ntot * 6 * RAW, &Q)) ||
      (r = need(ctx, 4, ntot * RAW, &N)) || (r = need(ctx, 5, ntot * RAW, &NI)) || (r = need(ctx, 6, ntot, &st)) || (r = need(ctx, 18, ntot * 24 * RAW, &St)) || (r = need(ctx, 19, ntot * 12 * RAW, &Pt2)) ||
      (r = need(ctx, 13, ntot * 6 * RAW, &S)) || (r = need(ctx, 11, ntot * 4 * POW_TAB * RAW, &tab))) return r;
  T += io * 4 * RAW; E += io * 6 * RAW; Pw += io * 4 * RAW; Q += io * 6 * RAW; N += io * RAW; NI += io * RAW; st += io; St += io * 24 * RAW; Pt2 += io * 12 * RAW; S += io * 6 * RAW; tab += io * 4 * POW_TAB * RAW;
  // H2C_A: per message the two field elements t (T), the exponentiation inputs (E) and the rest of the SWU state (St: twelve raw elements per map)
  if ((r = run(ctx, P_H2C_A, n, {B(0, d_uniform, 256), B(3, T, 4 * RAW), B(4, E, 4 * RAW), B(5, St, 24 * RAW)}, s))) return r;
  if ((r = run_pow(ctx, 2, 2 * n, E, Pw, s, tab))) return r;
  // H2C_B1: one map per item (2 n items) -> its point on E2'; H2C_B2: the two points of a message -> their sum on E2 (round 3: one program, 62 slots, four workgroups per CU)
  if ((r = run(ctx, P_H2C_B1, 2 * n, {B(3, T, 2 * RAW), B(5, Pw, 2 * RAW), B(4, St, 12 * RAW), B(6, Pt2, 6 * RAW)}, s))) return r;
  if ((r = run(ctx, P_H2C_B2, n, {B(3, Pt2, 12 * RAW), B(6, E, 6 * RAW)}, s))) return r;        // E is free again: reuse it for the E2 point
  if ((r = dev_clear_g2(ctx, n, E, Q, S, E, N, s))) return r;
  if (proj) { *proj = E; return NBLS_OK; }
  if ((r = run_inv_buf(ctx, n, N, NI, s))) return r;
  return run(ctx, P_G2_TO_AFFINE, n, {B(3, E, 6 * RAW), B(4, NI, RAW), B(2, d_out, 192), B(7, st, 1)}, s);
}
// sum of n affine points (left fold of add == tree of complete additions): affine wire bytes + status (1 = sum is the zero point)
static int dev_point_sum(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, void* d_out, void* d_status, hipStream_t s) {
  const size_t a = g2 ? 192 : 96, p = g2 ? 6 * RAW : 3 * RAW;
  uint8_t *A, *Bf, *N, *NI; int r;
  if ((r = need(ctx, 0, (n + 2) * p, &A)) || (r = need(ctx, 1, (n / 2 + 2) * p, &Bf)) || (r = need(ctx, 4, RAW, &N)) || (r = need(ctx, 5, RAW, &NI))) return r;
  uint8_t* ident = g2 ? ctx->ident_g2 : ctx->ident_g1;
  if (n == 0) { HIPCHK(hipMemcpyAsync(A, ident, p, hipMemcpyDeviceToDevice, s)); }
  else if ((r = run(ctx, g2 ? P_G2_TO_PROJ : P_G1_TO_PROJ, n, {B(g2 ? 1 : 0, d_pts, a), B(3, A, p)}, s))) return r;
  uint8_t *src = A, *dst = Bf; size_t m = n ? n : 1;
  while (m > 1) {
    if (m & 1) { HIPCHK(hipMemcpyAsync(src + m * p, ident, p, hipMemcpyDeviceToDevice, s)); m++; }
    if ((r = run(ctx, g2 ? P_G2_ADD2 : P_G1_ADD2, m / 2, {B(3, src, 2 * p), B(5, dst, p)}, s))) return r;
    std::swap(src, dst); m /= 2;
  }
  if ((r = run(ctx, g2 ? P_G2_NORM : P_G1_NORM, 1, {B(3, src, p), B(4, N, RAW)}, s))) return r;
  if ((r = run_inv_buf(ctx, 1, N, NI, s))) return r;
  return run(ctx, g2 ? P_G2_TO_AFFINE : P_G1_TO_AFFINE, 1, {B(3, src, p), B(4, NI, RAW), B(2, d_out, a), B(7, d_status, 1)}, s);
}

// ---- host-buffer wrappers ------------------------------------------------------------------------------------
// Staging buffers on the device for one host-buffer call.  Round 5: taken from a pool the context keeps (best fit among the free blocks of at most four times the size; a miss
// allocates) and handed back when the call returns -- every such call ends with a stream synchronisation and holds the context's mutex throughout, so a block is never reused while
// the device still works on it.  The pool is capped (NBLS_IO_POOL_MB, default 1024): free blocks are released oldest first when it would grow past the cap.
// Round 6: a call that fails half way may leave copies or kernels in flight on its stream, so the destructor synchronises the stream before the blocks
// become free (on the success path the call has just synchronised: a query of an idle stream); buffers registered with secret() -- device copies of private keys -- are
// zeroed on the call's stream before its last synchronisation (wipe()), or here when the call did not get that far: the pool hands blocks to later, unrelated calls.
struct HostIO {
  nbls_ctx* ctx; std::vector<size_t> mine; hipStream_t s = nullptr; std::vector<std::pair<void*, size_t>> secrets; bool wiped = false;
  ~HostIO() {
    if (!mine.empty()) (void)hipStreamSynchronize(s ? s : ctx->stream);
    if (!wiped) for (auto& k : secrets) (void)hipMemset(k.first, 0, k.second);
    for (size_t i : mine) ctx->io_pool[i].busy = false;
  }
  void secret(void* p, size_t n) { if (p && n) secrets.push_back({p, n}); }
  // enqueue the zeroing of the key buffers behind the work that reads them (call it before the call's final synchronisation)
  int wipe(hipStream_t st) { for (auto& k : secrets) if (hipMemsetAsync(k.first, 0, k.second, st) != hipSuccess) return NBLS_EHIP; wiped = true; return NBLS_OK; }
  void* alloc(size_t n) {
    if (!n) n = 1;
    auto& pool = ctx->io_pool;
    size_t best = (size_t)-1;
    for (size_t i = 0; i < pool.size(); i++)
      if (!pool[i].busy && pool[i].p && pool[i].cap >= n && pool[i].cap / 4 <= n && (best == (size_t)-1 || pool[i].cap < pool[best].cap)) best = i;
    if (best != (size_t)-1) { pool[best].busy = true; mine.push_back(best); return pool[best].p; }
    static const size_t cap_bytes = (size_t)env_long("NBLS_IO_POOL_MB", 1024) << 20;
    const size_t cap = n + n / 8 + 256;
    for (size_t i = 0; i < pool.size() && ctx->io_pool_bytes + cap > cap_bytes; i++)
      if (!pool[i].busy && pool[i].p) { hipFree(pool[i].p); ctx->io_pool_bytes -= pool[i].cap; pool[i].p = nullptr; pool[i].cap = 0; }
    void* p = nullptr;
    if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    ctx->io_pool_bytes += cap;
    size_t slot = (size_t)-1;
    for (size_t i = 0; i < pool.size(); i++) if (!pool[i].p) { slot = i; break; }
    if (slot == (size_t)-1) { pool.push_back({nullptr, 0, false}); slot = pool.size() - 1; }
    pool[slot] = {p, cap, true};
    mine.push_back(slot);
    return p;
  }
};
#define LOCKED(ctx) std::lock_guard<std::recursive_mutex> g_((ctx)->mu); HIPCHK(hipSetDevice((ctx)->device)); hipStream_t s = (ctx)->stream

// ---- prepared G2 points: PointG2.pairingPrecomputes() (index.ts:703-711) and PointG1.millerLoop (index.ts:452-454) -----------------
// d_tables: n line tables of NBLS_LINE_TABLE_BYTES each, device-resident, in the engine's raw limb format
EXPORT int nbls_g2_prepare_dev(nbls_ctx* ctx, size_t n, const void* d_g2, void* d_tables, void* stream) {
  if (!ctx || (n && (!d_g2 || !d_tables))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  StreamOrder order_(ctx, s);
  return run(ctx, P_LINES_Q, n, {B(1, d_g2, 192), B(3, d_tables, LINE_BYTES)}, s);
}
// raw tables <-> the reference's value: 68 x [Fp2, Fp2, Fp2] as Fp2.toBytes (NBLS_LINE_WIRE_BYTES per point)
EXPORT int nbls_lines_to_wire_dev(nbls_ctx* ctx, size_t n, const void* d_tables, void* d_wire, void* stream) {
  if (!ctx || (n && (!d_tables || !d_wire))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  return run(ctx, P_LINES_BYTES, n * N_LINES, {B(3, d_tables, 6 * RAW), B(2, d_wire, 288)}, s);
}
EXPORT int nbls_lines_from_wire_dev(nbls_ctx* ctx, size_t n, const void* d_wire, void* d_tables, void* stream) {
  if (!ctx || (n && (!d_tables || !d_wire))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  return run(ctx, P_LINES_FROM_BYTES, n * N_LINES, {B(2, d_wire, 288), B(3, d_tables, 6 * RAW)}, s);
}
// millerLoop(table_i, P_i) for n items (table_stride = NBLS_LINE_TABLE_BYTES) or millerLoop(table, P_i) with ONE table for every item
// (table_stride = 0): raw Fp12 values in ctx->F
static int acc_prepared(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_tables, size_t table_stride, hipStream_t s) {
  if (table_stride != 0 && table_stride != LINE_BYTES) return NBLS_EINVAL;
  int r = ensure_scratch(ctx, n); if (r) return r;
  return run(ctx, P_ACC_Q, n, {B(0, d_g1, 96), B(3, d_tables, table_stride), B(5, ctx->F, F12)}, s);
}
EXPORT int nbls_pairing_prepared_dev(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_tables, size_t table_stride, int with_final_exp, void* d_out, void* stream) {
  if (!ctx || (n && (!d_g1 || !d_tables || !d_out))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  StreamOrder order_(ctx, s);
  int r = acc_prepared(ctx, n, d_g1, d_tables, table_stride, s); if (r) return r;
  if (!with_final_exp) return run(ctx, P_RAW_TO_BYTES, n, {B(3, ctx->F, F12), B(2, d_out, 576)}, s);
  if ((r = run(ctx, P_NORM_RAW, n, {B(3, ctx->F, F12), B(4, ctx->N, RAW)}, s))) return r;
  return final_exp_pipeline(ctx, n, ctx->F, d_out, s);
}
EXPORT int nbls_miller_product_prepared_dev(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_tables, size_t table_stride, int final_exp, void* d_out, void* stream) {
  if (!ctx || !d_out || (n && (!d_g1 || !d_tables))) return NBLS_EINVAL;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  StreamOrder order_(ctx, s);
  int r;
  uint8_t* res = ctx->F;
  if (n == 0) { if ((r = ensure_scratch(ctx, 1))) return r; HIPCHK(hipMemcpyAsync(ctx->F, ctx->one12, F12, hipMemcpyDeviceToDevice, s)); }
  else {
    if ((r = acc_prepared(ctx, n, d_g1, d_tables, table_stride, s))) return r;
    if ((r = reduce_product(ctx, n, &res, s))) return r;
  }
  return finish_single(ctx, res, final_exp, d_out, s);
}
// host buffers: affine G2 points -> tables in wire form (what PointG2.pairingPrecomputes() returns)
EXPORT int nbls_g2_prepare(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, uint8_t* out_wire) {
  if (!ctx || (n && (!g2_aff || !out_wire))) return NBLS_EINVAL; if (!n) return NBLS_OK;
  LOCKED(ctx); HostIO io{ctx}; void *d = io.alloc(n * 192), *t = io.alloc(n * LINE_BYTES), *w = io.alloc(n * (size_t)N_LINES * 288); if (!d || !t || !w) return NBLS_EHIP;
  HIPCHK(hipMemcpyAsync(d, g2_aff, n * 192, hipMemcpyHostToDevice, s));
  int r;
  if ((r = nbls_g2_prepare_dev(ctx, n, d, t, s)) || (r = nbls_lines_to_wire_dev(ctx, n, t, w, s))) return r;
  HIPCHK(hipMemcpyAsync(out_wire, w, n * (size_t)N_LINES * 288, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); return NBLS_OK;
}
// host buffers: n G1 points against n_tables (1 or n) tables in wire form; product != 0: one Fp12 (the product of the Miller values), else n
EXPORT int nbls_pairing_prepared(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, const uint8_t* tables_wire, size_t n_tables, int with_final_exp, int product, uint8_t* out_fp12) {
  if (!ctx || !out_fp12 || (n && (!g1_aff || !tables_wire)) || (n && n_tables != 1 && n_tables != n)) return NBLS_EINVAL;
  if (!n && !product) return NBLS_OK;
  LOCKED(ctx); HostIO io{ctx};
  const size_t wire = (size_t)N_LINES * 288, nout = product ? 1 : n;
  void *d = io.alloc(n * 96), *w = io.alloc(n_tables * wire), *t = io.alloc(n_tables * LINE_BYTES), *o = io.alloc(nout * 576); if (!d || !w || !t || !o) return NBLS_EHIP;
  int r;
  if (n) {
    HIPCHK(hipMemcpyAsync(d, g1_aff, n * 96, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(w, tables_wire, n_tables * wire, hipMemcpyHostToDevice, s));
    if ((r = nbls_lines_from_wire_dev(ctx, n_tables, w, t, s))) return r;
  }
  const size_t stride = n_tables == 1 && n > 1 ? 0 : LINE_BYTES;
  r = product ? nbls_miller_product_prepared_dev(ctx, n, d, t, stride, with_final_exp, o, s) : nbls_pairing_prepared_dev(ctx, n, d, t, stride, with_final_exp, o, s);
  if (r) return r;
  HIPCHK(hipMemcpyAsync(out_fp12, o, nout * 576, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); return NBLS_OK;
}

EXPORT int nbls_set_tuning(nbls_ctx* ctx, int key, long long value) {
  if (!ctx) return NBLS_EINVAL;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  switch (key) {
    case NBLS_TUNE_SPLIT_MILLER_MIN: if (value < 0) return NBLS_EINVAL; ctx->split_min = (size_t)value; return NBLS_OK;
    case NBLS_TUNE_HALVES_MIN: if (value < 0) return NBLS_EINVAL; ctx->halves_min = value == 0 ? (size_t)-1 : (size_t)value; return NBLS_OK;
    case NBLS_TUNE_EXPC_MIN: if (value < 0) return NBLS_EINVAL; ctx->expc_min = (size_t)value; return NBLS_OK;
    case NBLS_TUNE_CHAIN_MAX: if (value < 0) return NBLS_EINVAL; ctx->chain_max = (size_t)value; return NBLS_OK;
    case NBLS_TUNE_SAC_MAX: if (value < 0) return NBLS_EINVAL; ctx->sac_max = (size_t)value; return NBLS_OK;
    case NBLS_TUNE_PT_LS2_MAX: if (value < 0) return NBLS_EINVAL; ctx->pt_ls2_max = (size_t)value; return NBLS_OK;
    case NBLS_TUNE_WIDE_MAX: if (value < 0) return NBLS_EINVAL; ctx->wide_max = (size_t)value; return NBLS_OK;
    case NBLS_TUNE_VERIFY_CHUNKS: if (value < 0 || value > 16) return NBLS_EINVAL; ctx->verify_chunks = (long)value; return NBLS_OK;
    case NBLS_TUNE_VERIFY_LAST_PCT: if (value < 1 || value > 100) return NBLS_EINVAL; ctx->verify_last_pct = (long)value; return NBLS_OK;
    case NBLS_TUNE_VERIFY_PIPE_MIN: if (value < 0) return NBLS_EINVAL; ctx->verify_pipe_min = (long)value; return NBLS_OK;
    default: return NBLS_EINVAL;
  }
}
// Which kernel executes a program in THIS context: "nbls_aot_<name>" when the program was translated and bound to its ahead-of-time kernel at upload, else the
// interpreter ("nbls_vm_kernel": NBLS_AOT=0, a program without an ahead-of-time kernel, or step signatures that differ from the kernel's table -- a build mismatch).
// Programs uploaded on first use (the scalar-multiplication ladders) are uploaded by the query.  tests/test_gpu_binding.py and bench.py (`aot_programs`) read it.
EXPORT const char* nbls_program_kernel(nbls_ctx* ctx, int prog) {
  if (!ctx || prog < 0 || prog >= P_COUNT) return nullptr;
  std::lock_guard<std::recursive_mutex> g(ctx->mu);
  if (hipSetDevice(ctx->device) != hipSuccess || upload(ctx, (ProgId)prog)) return nullptr;
  const DevProgram& d = ctx->prog[prog];
  return d.aot >= 0 ? nbls_aot_name(d.aot) : d.p->lsplit == 4 ? "nbls_vm_kernel_ls4" : d.p->lsplit == 1 ? "nbls_vm_kernel" : "none (two-lane programs have no interpreter form)";
}
EXPORT int nbls_program_count(void) { return (int)P_COUNT; }
EXPORT const char* nbls_program_name(int prog) { return prog >= 0 && prog < P_COUNT ? get_program((ProgId)prog).name.c_str() : nullptr; }

EXPORT int nbls_g1_validate_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, int8_t* status) {
  if (!ctx || (n && (!g1_aff || !status))) return NBLS_EINVAL; if (!n) return NBLS_OK;
  LOCKED(ctx); HostIO io{ctx}; void *d = io.alloc(n * 96), *st = io.alloc(n); if (!d || !st) return NBLS_EHIP;
  HIPCHK(hipMemcpyAsync(d, g1_aff, n * 96, hipMemcpyHostToDevice, s));
  int r = dev_validate(ctx, false, n, d, st, s); if (r) return r;
  HIPCHK(hipMemcpyAsync(status, st, n, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); return NBLS_OK;
}
EXPORT int nbls_g2_validate_batch(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, int8_t* status) {
  if (!ctx || (n && (!g2_aff || !status))) return NBLS_EINVAL; if (!n) return NBLS_OK;
  LOCKED(ctx); HostIO io{ctx}; void *d = io.alloc(n * 192), *st = io.alloc(n); if (!d || !st) return NBLS_EHIP;
  HIPCHK(hipMemcpyAsync(d, g2_aff, n * 192, hipMemcpyHostToDevice, s));
  int r = dev_validate(ctx, true, n, d, st, s); if (r) return r;
  HIPCHK(hipMemcpyAsync(status, st, n, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); return NBLS_OK;
}
static int decompress_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* in, uint8_t* out, int8_t* status) {
  if (!ctx || (n && (!in || !out))) return NBLS_EINVAL; if (!n) return NBLS_OK;
  const size_t e = g2 ? 96 : 48;
  LOCKED(ctx); HostIO io{ctx}; void *d = io.alloc(n * e), *o = io.alloc(n * 2 * e), *st = io.alloc(n); if (!d || !o || !st) return NBLS_EHIP;
  HIPCHK(hipMemcpyAsync(d, in, n * e, hipMemcpyHostToDevice, s));
  int r = dev_decompress(ctx, g2, n, d, o, st, s); if (r) return r;
  HIPCHK(hipMemcpyAsync(out, o, n * 2 * e, hipMemcpyDeviceToHost, s));
  std::vector<int8_t> tmp(n); HIPCHK(hipMemcpyAsync(tmp.data(), st, n, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
  if (status) memcpy(status, tmp.data(), n);
  return NBLS_OK;
}
EXPORT int nbls_g1_decompress_batch(nbls_ctx* ctx, size_t n, const uint8_t* in48, uint8_t* out96, int8_t* status) { return decompress_host(ctx, false, n, in48, out96, status); }
EXPORT int nbls_g2_decompress_batch(nbls_ctx* ctx, size_t n, const uint8_t* in96, uint8_t* out192, int8_t* status) { return decompress_host(ctx, true, n, in96, out192, status); }

// expand_message_xmd for all messages on the device (xmd_kernel.hip): uploads the message blob, the n+1 offsets and the DST
// into the scratch pool and leaves len_in_bytes (64, 128 or 256) uniform bytes per message in *d_uniform.  Only a DST longer than 255 bytes is touched
// on the host (RFC 9380 5.3.3: replaced by its SHA-256 digest), which is per call, not per message.
static int dev_expand(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offs, const uint8_t* dst, size_t dst_len, uint8_t** d_uniform, hipStream_t s, unsigned len_in_bytes = 256) {
  for (size_t i = 0; i < n; i++) if (offs[i + 1] < offs[i]) return NBLS_EINVAL;
  const size_t total = offs[n] - offs[0];
  uint8_t dst_hash[32];
  if (dst_len > 255) { Sha256 c; c.update((const uint8_t*)"H2C-OVERSIZE-DST-", 17); c.update(dst, dst_len); c.final(dst_hash); dst = dst_hash; dst_len = 32; }
  uint8_t *dm, *dofs, *dd, *du; int r;
  if ((r = need(ctx, 7, total + 4, &dm)) || (r = need(ctx, 12, (n + 1) * 4 + 256, &dofs)) || (r = need(ctx, 8, n * (size_t)len_in_bytes, &du))) return r;
  dd = dofs + (n + 1) * 4;
  std::vector<uint32_t> rel(n + 1); for (size_t i = 0; i <= n; i++) rel[i] = offs[i] - offs[0];
  if (total) HIPCHK(hipMemcpyAsync(dm, msgs + offs[0], total, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(dofs, rel.data(), (n + 1) * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(dd, dst, dst_len, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));     // `rel` and a hashed DST live on this stack frame
  int e = nbls_xmd_launch((unsigned)n, dm, dofs, dd, (unsigned)dst_len, du, len_in_bytes, nullptr, s);
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  *d_uniform = du;
  return NBLS_OK;
}
EXPORT int nbls_hash_to_g2_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out192) {
  if (!ctx || (n && (!offsets || !out192 || !dst))) return NBLS_EINVAL; if (!n) return NBLS_OK;
  LOCKED(ctx); HostIO io{ctx}; void* o = io.alloc(n * 192); if (!o) return NBLS_EHIP;
  uint8_t* d; int r = dev_expand(ctx, n, msgs, offsets, dst, dst_len, &d, s); if (r) return r;
  if ((r = dev_hash_to_g2(ctx, n, d, o, s))) return r;
  HIPCHK(hipMemcpyAsync(out192, o, n * 192, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); return NBLS_OK;
}
static int sum_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* pts, uint8_t* out, int8_t* status) {
  if (!ctx || !out || (n && !pts)) return NBLS_EINVAL;
  const size_t a = g2 ? 192 : 96;
  LOCKED(ctx); HostIO io{ctx}; void *d = io.alloc(n * a), *o = io.alloc(a), *st = io.alloc(1); if (!d || !o || !st) return NBLS_EHIP;
  if (n) HIPCHK(hipMemcpyAsync(d, pts, n * a, hipMemcpyHostToDevice, s));
  int r = dev_point_sum(ctx, g2, n, d, o, st, s); if (r) return r;
  int8_t z = 0; HIPCHK(hipMemcpyAsync(out, o, a, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(&z, st, 1, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
  if (status) *status = z;
  return NBLS_OK;
}
EXPORT int nbls_g1_sum(nbls_ctx* ctx, size_t n, const uint8_t* pts96, uint8_t* out96, int8_t* status) { return sum_host(ctx, false, n, pts96, out96, status); }
EXPORT int nbls_g2_sum(nbls_ctx* ctx, size_t n, const uint8_t* pts192, uint8_t* out192, int8_t* status) { return sum_host(ctx, true, n, pts192, out192, status); }

// PointG1.hashToCurve (count = 2) / PointG1.encodeToCurve (count = 1) on 64 * count uniform bytes per message (index.ts:331-350)
static int dev_hash_to_g1(nbls_ctx* ctx, int count, size_t n, const void* d_uniform, void* d_out, hipStream_t s) {
  uint8_t *U, *E, *Pw, *Q, *Q2, *N, *NI, *st; int r;
  if ((r = need(ctx, 0, n * 2 * RAW, &U)) || (r = need(ctx, 1, n * 3 * RAW, &E)) || (r = need(ctx, 2, n * 2 * RAW, &Pw)) || (r = need(ctx, 3, n * 3 * RAW, &Q)) ||
      (r = need(ctx, 4, n * RAW, &N)) || (r = need(ctx, 5, n * RAW, &NI)) || (r = need(ctx, 6, n, &st))) return r;
  Q2 = E;   // E (the exponentiation inputs) is free again after the pow kernel
  const size_t us = (size_t)count * RAW;
  if ((r = run(ctx, count == 2 ? P_H2C1_A : P_ENC1_A, n, {B(0, d_uniform, 64 * (size_t)count), B(3, U, us), B(4, E, us)}, s))) return r;
  if ((r = run_pow(ctx, 3, (size_t)count * n, E, Pw, s))) return r;
  if ((r = run(ctx, count == 2 ? P_H2C1_B : P_ENC1_B, n, {B(3, U, us), B(5, Pw, us), B(6, Q, 3 * RAW)}, s))) return r;
  if ((r = run(ctx, P_G1_CLEAR, n, {B(3, Q, 3 * RAW), B(6, Q2, 3 * RAW), B(7, N, RAW)}, s))) return r;
  if ((r = run_inv_buf(ctx, n, N, NI, s))) return r;
  return run(ctx, P_G1_TO_AFFINE, n, {B(3, Q2, 3 * RAW), B(4, NI, RAW), B(2, d_out, 96), B(7, st, 1)}, s);
}
// PointG2.encodeToCurve on 128 uniform bytes per message (index.ts:491-497)
static int dev_encode_to_g2(nbls_ctx* ctx, size_t n, const void* d_uniform, void* d_out, hipStream_t s) {
  uint8_t *T, *E, *Pw, *Q, *N, *NI, *st; int r;
  if ((r = need(ctx, 0, n * 2 * RAW, &T)) || (r = need(ctx, 1, n * 6 * RAW, &E)) || (r = need(ctx, 2, n * 2 * RAW, &Pw)) || (r = need(ctx, 3, n * 6 * RAW, &Q)) ||
      (r = need(ctx, 4, n * RAW, &N)) || (r = need(ctx, 5, n * RAW, &NI)) || (r = need(ctx, 6, n, &st))) return r;
  if ((r = run(ctx, P_ENC2_A, n, {B(0, d_uniform, 128), B(3, T, 2 * RAW), B(4, E, 2 * RAW)}, s))) return r;
  if ((r = run_pow(ctx, 2, n, E, Pw, s))) return r;
  if ((r = run(ctx, P_ENC2_B, n, {B(3, T, 2 * RAW), B(5, Pw, 2 * RAW), B(6, E, 6 * RAW)}, s))) return r;
  uint8_t* S; if ((r = need(ctx, 13, n * 6 * RAW, &S))) return r;
  if ((r = dev_clear_g2(ctx, n, E, Q, S, E, N, s))) return r;
  if ((r = run_inv_buf(ctx, n, N, NI, s))) return r;
  return run(ctx, P_G2_TO_AFFINE, n, {B(3, E, 6 * RAW), B(4, NI, RAW), B(2, d_out, 192), B(7, st, 1)}, s);
}
// mode: 0 = PointG1.hashToCurve, 1 = PointG1.encodeToCurve, 2 = PointG2.encodeToCurve
static int hash_curve_host(nbls_ctx* ctx, int mode, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out) {
  if (!ctx || (n && (!offsets || !out || !dst))) return NBLS_EINVAL; if (!n) return NBLS_OK;
  const size_t a = mode == 2 ? 192 : 96;
  LOCKED(ctx); HostIO io{ctx}; void* o = io.alloc(n * a); if (!o) return NBLS_EHIP;
  uint8_t* d; int r = dev_expand(ctx, n, msgs, offsets, dst, dst_len, &d, s, mode == 1 ? 64 : 128); if (r) return r;
  r = mode == 2 ? dev_encode_to_g2(ctx, n, d, o, s) : dev_hash_to_g1(ctx, mode == 0 ? 2 : 1, n, d, o, s); if (r) return r;
  HIPCHK(hipMemcpyAsync(out, o, n * a, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); return NBLS_OK;
}
EXPORT int nbls_hash_to_g1_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out96) { return hash_curve_host(ctx, 0, n, msgs, offsets, dst, dst_len, out96); }
EXPORT int nbls_encode_to_g1_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out96) { return hash_curve_host(ctx, 1, n, msgs, offsets, dst, dst_len, out96); }
EXPORT int nbls_encode_to_g2_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out192) { return hash_curve_host(ctx, 2, n, msgs, offsets, dst, dst_len, out192); }

// PointG1.toHex(true) / PointG2.toSignature for non-zero affine points (index.ts:359-371, 586-602): bulk serialisation
static int compress_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* aff, uint8_t* out) {
  if (!ctx || (n && (!aff || !out))) return NBLS_EINVAL; if (!n) return NBLS_OK;
  const size_t a = g2 ? 192 : 96, c = a / 2;
  LOCKED(ctx); HostIO io{ctx}; void *d = io.alloc(n * a), *o = io.alloc(n * c); if (!d || !o) return NBLS_EHIP;
  HIPCHK(hipMemcpyAsync(d, aff, n * a, hipMemcpyHostToDevice, s));
  int r = run(ctx, g2 ? P_G2_COMPRESS : P_G1_COMPRESS, n, {B(0, d, a), B(2, o, c)}, s); if (r) return r;
  HIPCHK(hipMemcpyAsync(out, o, n * c, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
  return NBLS_OK;
}
EXPORT int nbls_g1_compress_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, uint8_t* out48) { return compress_host(ctx, false, n, g1_aff, out48); }
EXPORT int nbls_g2_compress_batch(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, uint8_t* out96) { return compress_host(ctx, true, n, g2_aff, out96); }

// ---- every wire form of the reference's point codecs, in bulk --------------------------------------------------------------
// PointG1.fromHex (index.ts:298-327): 48 compressed or 96 uncompressed bytes per point; PointG2.fromHex (index.ts:532-579): 96 compressed
// (flag rules, no subgroup check) or 192 uncompressed bytes; PointG2.fromSignature (index.ts:500-530): 96 or 192 bytes.
static int decode_host(nbls_ctx* ctx, int kind /* 0 g1.fromHex, 1 g2.fromHex, 2 g2.fromSignature */, size_t n, const uint8_t* in, size_t len, uint8_t* out, int8_t* status) {
  if (!ctx || (n && (!in || !out || !status))) return NBLS_EINVAL;
  const bool g2 = kind != 0;
  const size_t a = g2 ? 192 : 96;
  if (len != a && len != a / 2) return NBLS_EINVAL;
  if (!n) return NBLS_OK;
  LOCKED(ctx); HostIO io{ctx}; void *d = io.alloc(n * len), *o = io.alloc(n * a), *st = io.alloc(n); if (!d || !o || !st) return NBLS_EHIP;
  HIPCHK(hipMemcpyAsync(d, in, n * len, hipMemcpyHostToDevice, s));
  int r;
  if (kind == 0) r = len == 48 ? dev_decompress(ctx, false, n, d, o, st, s) : run(ctx, P_G1_FROM_RAW, n, {B(0, d, 96), B(6, o, 96), B(7, st, 1)}, s);
  else if (kind == 1) r = len == 96 ? dev_decompress(ctx, true, n, d, o, st, s, 0, 11, 2) : run(ctx, P_G2_FROM_RAW, n, {B(0, d, 192), B(6, o, 192), B(7, st, 1)}, s);
  else r = dev_decompress(ctx, true, n, d, o, st, s, 0, 11, len == 192 ? 1 : 0);
  if (r) return r;
  HIPCHK(hipMemcpyAsync(out, o, n * a, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(status, st, n, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
  return NBLS_OK;
}
EXPORT int nbls_g1_from_hex_batch(nbls_ctx* ctx, size_t n, const uint8_t* in, size_t len, uint8_t* out96, int8_t* status) { return decode_host(ctx, 0, n, in, len, out96, status); }
EXPORT int nbls_g2_from_hex_batch(nbls_ctx* ctx, size_t n, const uint8_t* in, size_t len, uint8_t* out192, int8_t* status) { return decode_host(ctx, 1, n, in, len, out192, status); }
EXPORT int nbls_g2_from_signature_batch(nbls_ctx* ctx, size_t n, const uint8_t* in, size_t len, uint8_t* out192, int8_t* status) { return decode_host(ctx, 2, n, in, len, out192, status); }
// PointG1.toHex / PointG2.toHex (index.ts:359-381, 603-631) for n valid affine points; zero[i] != 0 marks the zero point (its affine bytes are
// ignored): compressed 0xc0 00.., uncompressed 0x40 00..
static int encode_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* aff, const int8_t* zero, int compressed, uint8_t* out) {
  if (!ctx || (n && (!aff || !out))) return NBLS_EINVAL; if (!n) return NBLS_OK;
  const size_t a = g2 ? 192 : 96, c = compressed ? a / 2 : a;
  int r;
  if (compressed) { if ((r = compress_host(ctx, g2, n, aff, out))) return r; }
  else if (!g2) memcpy(out, aff, n * a);
  else {
    LOCKED(ctx); HostIO io{ctx}; void *d = io.alloc(n * a), *o = io.alloc(n * a); if (!d || !o) return NBLS_EHIP;
    HIPCHK(hipMemcpyAsync(d, aff, n * a, hipMemcpyHostToDevice, s));
    if ((r = run(ctx, P_G2_SWAP, n, {B(0, d, a), B(2, o, a)}, s))) return r;
    HIPCHK(hipMemcpyAsync(out, o, n * a, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
  }
  if (zero) for (size_t i = 0; i < n; i++) if (zero[i]) { memset(out + i * c, 0, c); out[i * c] = compressed ? 0xc0 : 0x40; }
  return NBLS_OK;
}
EXPORT int nbls_g1_to_hex_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, const int8_t* zero, int compressed, uint8_t* out) { return encode_host(ctx, false, n, g1_aff, zero, compressed, out); }
EXPORT int nbls_g2_to_hex_batch(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, const int8_t* zero, int compressed, uint8_t* out) { return encode_host(ctx, true, n, g2_aff, zero, compressed, out); }
// PointG1.clearCofactor (index.ts:401-405) / PointG2.clearCofactor (index.ts:659-672) for n affine points ON THE CURVE (any subgroup):
// status 1 = the result is the zero point
static int clear_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* aff, uint8_t* out, int8_t* status) {
  if (!ctx || (n && (!aff || !out || !status))) return NBLS_EINVAL; if (!n) return NBLS_OK;
  const size_t a = g2 ? 192 : 96, p = g2 ? 6 * RAW : 3 * RAW;
  LOCKED(ctx); HostIO io{ctx};
  void *d = io.alloc(n * a), *P = io.alloc(n * p), *Q = io.alloc(n * p), *N = io.alloc(n * RAW), *NI = io.alloc(n * RAW), *o = io.alloc(n * a), *st = io.alloc(n);
  if (!d || !P || !Q || !N || !NI || !o || !st) return NBLS_EHIP;
  HIPCHK(hipMemcpyAsync(d, aff, n * a, hipMemcpyHostToDevice, s));
  int r;
  if ((r = run(ctx, g2 ? P_G2_TO_PROJ : P_G1_TO_PROJ, n, {B(g2 ? 1 : 0, d, a), B(3, P, p)}, s))) return r;
  if (g2) { void* S2 = io.alloc(n * p); if (!S2) return NBLS_EHIP; if ((r = dev_clear_g2(ctx, n, P, (uint8_t*)Q, (uint8_t*)S2, Q, N, s))) return r; }
  else if ((r = run(ctx, P_G1_CLEAR, n, {B(3, P, p), B(6, Q, p), B(7, N, RAW)}, s))) return r;
  if ((r = run_inv_buf(ctx, n, N, NI, s))) return r;
  if ((r = run(ctx, g2 ? P_G2_TO_AFFINE : P_G1_TO_AFFINE, n, {B(3, Q, p), B(4, NI, RAW), B(2, o, a), B(7, st, 1)}, s))) return r;
  HIPCHK(hipMemcpyAsync(out, o, n * a, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(status, st, n, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
  return NBLS_OK;
}
EXPORT int nbls_g1_clear_cofactor_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, uint8_t* out96, int8_t* status) { return clear_host(ctx, false, n, g1_aff, out96, status); }
EXPORT int nbls_g2_clear_cofactor_batch(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, uint8_t* out192, int8_t* status) { return clear_host(ctx, true, n, g2_aff, out192, status); }

// [k_i]P_i for per-item 256-bit big-endian scalars (pt_stride 0 = one point for all items): ladder -> inversion -> affine
static int dev_point_mul(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, size_t pt_stride, const void* d_scalars, void* d_out, void* d_status, hipStream_t s, bool allow_fixed = true, bool in_subgroup = false);
// the fixed-base table of G1.BASE (curve.h pt_mul_fixed_g1): for every window w and digit d = 1 .. 2^WIN - 1 the point [d 2^(WIN w)]G as a raw projective point (x, y, 1), computed
// ONCE per context by the variable-base ladder itself (602 scalar multiplications with WIN = 3: a few hundred microseconds) -- no table of constants enters the source
static int ensure_g1_fixed(nbls_ctx* ctx, hipStream_t s) {
  if (ctx->g1_fixed) return NBLS_OK;
  const int WIN = G1_FIXED_WIN, NW = g1_fixed_windows(), NE = g1_fixed_entries();
  const size_t m = (size_t)NW * NE;
  std::vector<uint8_t> ks(m * 32, 0);
  for (int w = 0; w < NW; w++)
    for (int d = 1; d <= NE; d++) {
      uint8_t* k = &ks[((size_t)w * NE + d - 1) * 32];
      const int sh = WIN * w;                                         // d << sh as a 256-bit big-endian integer; digits that would pass bit 255 (the short top window) are never read: [1]G stands in
      if (sh + 32 - __builtin_clz((unsigned)d) > 256) { k[31] = 1; continue; }
      for (int bit = 0; bit < WIN; bit++) if ((d >> bit) & 1) { const int pos = sh + bit; k[31 - pos / 8] |= (uint8_t)(1u << (pos % 8)); }
    }
  uint8_t *dk = nullptr, *aff = nullptr, *st = nullptr, *tab = nullptr;
  auto fail = [&](int code) { for (uint8_t* p : {dk, aff, st, tab}) if (p) hipFree(p); return code; };
  if (hipMalloc(&dk, m * 32) != hipSuccess || hipMalloc(&aff, m * 96) != hipSuccess || hipMalloc(&st, m) != hipSuccess || hipMalloc(&tab, m * 3 * RAW) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); return fail(NBLS_EHIP); }
  if (hipMemcpyAsync(dk, ks.data(), m * 32, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); return fail(NBLS_EHIP); }
  int r = dev_point_mul(ctx, false, m, ctx->gen_g1, 0, dk, aff, st, s, false);
  if (!r) r = run(ctx, P_G1_TO_PROJ, m, {B(0, aff, 96), B(3, tab, 3 * RAW)}, s);
  if (!r && hipStreamSynchronize(s) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); r = NBLS_EHIP; }
  if (r) return fail(r);
  hipFree(dk); hipFree(aff); hipFree(st);
  ctx->g1_fixed = tab;
  return NBLS_OK;
}
static int dev_point_mul(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, size_t pt_stride, const void* d_scalars, void* d_out, void* d_status, hipStream_t s, bool allow_fixed, bool in_subgroup) {
  const size_t a = g2 ? 192 : 96, p = g2 ? 6 * RAW : 3 * RAW;
  uint8_t *Pj, *N, *NI; int r;
  // getPublicKey (the base point is G1.BASE for every item): no doublings, the multiples of the generator come from a table (round 5: 86 additions instead of 256 doublings + 128
  // additions; NBLS_G1_FIXED=0 keeps the ladder)
  static const bool fixed_on = env_long("NBLS_G1_FIXED", 1) != 0;
  const bool fixed = allow_fixed && fixed_on && !g2 && d_pts == ctx->gen_g1 && pt_stride == 0;
  if (fixed && (r = ensure_g1_fixed(ctx, s))) return r;
  if ((r = need(ctx, 0, n * p, &Pj)) || (r = need(ctx, 4, n * RAW, &N)) || (r = need(ctx, 5, n * RAW, &NI))) return r;
  // sign (the base points are hash outputs: in G2 by construction): the scalar split along psi, four 65-bit digits on one accumulator (codec.h pt_mul_gls_g2: 66 doublings + 132 additions
  // instead of 256 + 128; NBLS_G2_GLS=0 keeps the plain ladder).  The digits are made on the device by the MSM's decomposition kernel (branch-free long division by |z|).
  static const bool gls_on = env_long("NBLS_G2_GLS", 1) != 0;
  if (g2 && in_subgroup && gls_on) {
    // up to sac_max keys d_pts are RAW PROJECTIVE points (six raw elements each, pt_stride = 6 * RAW: sign_points() below) -- the hash points as cofactor clearing leaves them in
    // scratch slot 1, so the digits go to slot 2; above, affine wire points as everywhere else
    uint8_t* dig;
    if ((r = need(ctx, n <= ctx->sac_max ? 2 : 1, n * 128, &dig))) return r;
    // while every wavefront of the launch is resident at once the length of ONE wavefront's instruction stream is the time: the sign-aligned recoding with one addition per bit
    // (codec.h pt_mul_sac_g2: 65 doublings + 73 additions; its table of eight points takes 101 slots = three workgroups per CU = 768 wavefronts of 8 keys); above 6144 keys the
    // windowed form, whose table of four leaves room for six workgroups per CU (NBLS_G2_SAC_MAX / NBLS_TUNE_SAC_MAX; 0 = never)
    if (n <= ctx->sac_max) {
      if (nbls_msm_sac_launch((unsigned)n, d_scalars, dig, s)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
      if ((r = run(ctx, pt_ls2_variant(ctx, P_G2_MUL_SAC, n), n, {B(1, d_pts, pt_stride), B(2, dig, 128), B(3, Pj, p), B(4, N, RAW)}, s))) return r;
    } else {
    if (nbls_msm_decompose_launch((unsigned)n, 4, d_scalars, dig, s)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
    if ((r = run(ctx, P_G2_MUL_GLS, n, {B(1, d_pts, pt_stride), B(2, dig, 128), B(3, Pj, p), B(4, N, RAW)}, s))) return r;
    }
    HIPCHK(hipMemsetAsync(dig, 0, n * 128, s));      // the recoded digits ARE the private keys: not left in a scratch slot that later calls reuse
  } else
  if (fixed) {
    if ((r = run(ctx, P_G1_MUL_FIXED, n, {B(2, d_scalars, 32), B(5, ctx->g1_fixed, 0), B(3, Pj, p), B(4, N, RAW)}, s))) return r;
  } else {
  // up to one wavefront per SIMD (16 / 8 items per wavefront) the length of one wavefront's instruction stream counts: 3-bit windows (85 additions); above, wavefronts per CU
  // count: 2-bit windows, whose table of four leaves room for seven workgroups per CU instead of three / four (tools/mul_time.py)
  static const size_t w3_waves = (size_t)env_long("NBLS_MUL_W3_WAVES", 1024);
  const bool w3 = (n + (g2 ? 7 : 15)) / (g2 ? 8 : 16) <= w3_waves;
  if ((r = run(ctx, g2 ? (w3 ? P_G2_MUL_W3 : P_G2_MUL) : (w3 ? P_G1_MUL_W3 : P_G1_MUL), n, {B(g2 ? 1 : 0, d_pts, pt_stride), B(2, d_scalars, 32), B(3, Pj, p), B(4, N, RAW)}, s))) return r;
  }
  if ((r = run_inv_buf(ctx, n, N, NI, s))) return r;
  return run(ctx, g2 ? P_G2_TO_AFFINE : P_G1_TO_AFFINE, n, {B(3, Pj, p), B(4, NI, RAW), B(2, d_out, a), B(7, d_status, 1)}, s);
}
// sign's two halves: hash-to-G2, then the key ladder on the hash points.  Up to sac_max keys the points stay projective in between (no inversion, no affine program: the
// sign-aligned ladder reads raw projective points); above, they are normalised first (the windowed ladder's affine input keeps it at six workgroups per CU).  `h`: n * 192 bytes of
// device scratch for the affine points of the second form
static int sign_points(nbls_ctx* ctx, size_t n, const void* d_uniform, void* h, const void* d_keys32, void* d_out192, void* d_status, hipStream_t s) {
  int r;
  static const bool gls_on = env_long("NBLS_G2_GLS", 1) != 0;
  if (gls_on && n <= ctx->sac_max) {
    uint8_t* pj;
    if ((r = dev_hash_to_g2(ctx, n, d_uniform, nullptr, s, 0, 0, &pj))) return r;
    return dev_point_mul(ctx, true, n, pj, 6 * RAW, d_keys32, d_out192, d_status, s, true, true);
  }
  if ((r = dev_hash_to_g2(ctx, n, d_uniform, h, s))) return r;
  return dev_point_mul(ctx, true, n, h, 192, d_keys32, d_out192, d_status, s, true, true);      // H(m) is in G2: the ladder may split the key along psi
}
// scalar k is acceptable iff k mod r != 0 (normalizePrivKey, index.ts:269-279, reduces mod r and rejects zero); the ladder
// itself takes any 256-bit value since the points are in the order-r subgroup
static bool scalar_is_zero_mod_r(const uint8_t* k32) {
  static const uint8_t R_BE[32] = {0x73, 0xed, 0xa7, 0x53, 0x29, 0x9d, 0x7d, 0x48, 0x33, 0x39, 0xd8, 0x08, 0x09, 0xa1, 0xd8, 0x05,
                                   0x53, 0xbd, 0xa4, 0x02, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0xff, 0xff, 0xff, 0x00, 0x00, 0x00, 0x01};
  uint8_t m[32] = {0};   // m = 0, r, 2r, 3r  (4r > 2^256 - 1? 4r = 0x1cfb6..., 33 bytes: stop at 3r)
  for (int mult = 0; mult < 4; mult++) {
    if (memcmp(m, k32, 32) == 0) return true;
    unsigned c = 0; for (int i = 31; i >= 0; i--) { unsigned v = (unsigned)m[i] + R_BE[i] + c; m[i] = (uint8_t)v; c = v >> 8; }
    if (c) break;
  }
  return false;
}
static int mul_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* pts, const uint8_t* scalars32, uint8_t* out, int8_t* status) {
  if (!ctx || (n && (!scalars32 || !out)) || (g2 && n && !pts)) return NBLS_EINVAL; if (!n) return NBLS_OK;
  const size_t a = g2 ? 192 : 96;
  LOCKED(ctx); HostIO io{ctx}; void *dp = pts ? io.alloc(n * a) : nullptr, *dk = io.alloc(n * 32), *o = io.alloc(n * a), *st = io.alloc(n);
  if ((pts && !dp) || !dk || !o || !st) return NBLS_EHIP;
  io.secret(dk, n * 32);      // the scalars are private keys in getPublicKey / sign: zeroed before the staging block goes back to the pool
  if (pts) HIPCHK(hipMemcpyAsync(dp, pts, n * a, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(dk, scalars32, n * 32, hipMemcpyHostToDevice, s));
  int r = dev_point_mul(ctx, g2, n, pts ? dp : ctx->gen_g1, pts ? a : 0, dk, o, st, s); if (r) return r;
  std::vector<int8_t> tmp(n);
  if ((r = io.wipe(s))) return r;
  HIPCHK(hipMemcpyAsync(out, o, n * a, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(tmp.data(), st, n, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
  for (size_t i = 0; i < n; i++) if (scalar_is_zero_mod_r(scalars32 + 32 * i)) tmp[i] = 5;
  if (status) memcpy(status, tmp.data(), n);
  return NBLS_OK;
}
// PointG1.fromPrivateKey / getPublicKey core (index.ts:350-353, 738-740): [k_i]P_i, P = generator when g1_aff is NULL
EXPORT int nbls_g1_mul_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, const uint8_t* scalars32, uint8_t* out96, int8_t* status) { return mul_host(ctx, false, n, g1_aff, scalars32, out96, status); }
EXPORT int nbls_g2_mul_batch(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, const uint8_t* scalars32, uint8_t* out192, int8_t* status) { return mul_host(ctx, true, n, g2_aff, scalars32, out192, status); }
// ---- multi-scalar multiplication sum_i [k_i]P_i (the reference only has the unweighted sums aggregatePublicKeys /
// aggregateSignatures, index.ts:771-788).  Bucket method with 12-bit windows, every group operation a complete addition run as
// a step program over whole arrays:
//   1. keys (window, digit) for every (point, window); device radix sort of the n * nwin keys (hipCUB); the points follow.
//   2. segmented sum over the sorted list as a balanced tree inside every run of equal keys: in the round with stride d the
//      elements whose rank in their run is a multiple of 2d absorb the element d further on.  The pairs of a round are
//      listed by a compaction kernel (their number stays on the device: the step program reads it there) and added in place,
//      the step program addressing its operands through the list: n * nwin - (number of buckets hit) additions in total whatever the distribution of the digits;
//      ceil(log2(longest run)) rounds -- the longest run is the one value read back.  The head of every run ends up as its bucket sum.
//   3. sum_b b * B_b per window = sum_t 2^t * T_t with T_t = sum of the buckets whose index has bit t: 12 * 2^11 gathered
//      points per window, a balanced tree of 11 rounds of pairwise additions (data independent).
//   4. Horner over t inside every window (one item per window), then acc <- 2^12 * acc + S_w from the top window down.
// Result: affine wire bytes + status (1 = the sum is the zero point).  nbits bounds the scalars (< 2^nbits), 0 = 256.
#define MSMCHK(call) do { int e_ = (call); if (e_) { ctx->last_hip = e_; return NBLS_EHIP; } } while (0)
static int dev_msm(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, const void* d_scalars, unsigned nbits, void* d_out, void* d_status, hipStream_t s) {
  const size_t a = g2 ? 192 : 96, p = g2 ? 6 * RAW : 3 * RAW;
  const unsigned C = MSM_WINDOW_BITS;
  uint8_t* ident = g2 ? ctx->ident_g2 : ctx->ident_g1;
  if (nbits == 0 || nbits > 256) nbits = 256;
  if (n > ((size_t)1 << 22)) return NBLS_EINVAL;
  // Wide scalars are split with the curve endomorphisms (GLV / GLS): k = sum_i a_i |z|^i, [|z|^i]P is one cheap map of P.  G1:
  // 2 points with 129-bit scalars, G2: 4 points with 65-bit scalars -- the same number of bucket additions, but 10 / 5 instead
  // of 21 rounds of "12 doublings + 1 addition" on a single point at the end (latency-bound: 0.11 ms each).
  const bool split = nbits > 192;
  const unsigned dims = split ? (g2 ? 4 : 2) : 1;
  const size_t n_in = n;
  if (split) { n *= dims; nbits = g2 ? 65 : 129; }
  const unsigned nwin = (nbits + C - 1) / C;
  const size_t m = n * nwin, nb = (size_t)nwin << C, ng = ((size_t)nwin * C) << (C - 1);
  uint8_t *Pj, *P, *A, *K, *tmp, *Bk, *G, *Gh, *N, *NI, *acc, *cnt, *Ks = nullptr; int r;
  if (split && (r = need(ctx, 13, (n + 1) * 32, &Ks))) return r;
  size_t tmp_bytes = 0;
  size_t scan_bytes = 0;
  if (m) { MSMCHK(nbls_msm_sort_launch(nullptr, &tmp_bytes, nullptr, nullptr, nullptr, nullptr, m, 17, s)); MSMCHK(nbls_msm_rank_launch(nullptr, &scan_bytes, m, nullptr, nullptr, nullptr, s)); tmp_bytes = std::max(tmp_bytes, scan_bytes); }
  if ((r = need(ctx, 0, (n + 1) * p, &Pj)) || (r = need(ctx, 1, (m + 1) * p, &P)) || (r = need(ctx, 2, ((size_t)nwin + 1) * p, &A)) || (r = need(ctx, 3, (m + 1) * 24, &K)) ||
      (r = need(ctx, 4, RAW, &N)) || (r = need(ctx, 5, RAW, &NI)) || (r = need(ctx, 6, tmp_bytes + 16, &tmp)) || (r = need(ctx, 7, nb * p, &Bk)) || (r = need(ctx, 8, ng * p, &G)) ||
      (r = need(ctx, 9, (ng / 2 + 2) * p, &Gh)) || (r = need(ctx, 11, 64 * 4, &cnt))) return r;
  acc = Gh + ng / 2 * p;     // (slot 10 belongs to verifyBatch, which drops the context lock between its stages)
  uint32_t *kin = (uint32_t*)K, *vin = kin + m, *kout = vin + m, *vout = kout + m, *pos = vout + m, *list = pos + m;
  uint32_t* counters = (uint32_t*)cnt;    // [0] longest run, [1 + round] pairs of that round
  MSMCHK(nbls_msm_fill_launch(nb, (unsigned)p, ident, Bk, s));
  if (m) {
    if (split) {
      if ((r = run(ctx, g2 ? P_G2_MSM_PREP : P_G1_MSM_PREP, n_in, {B(g2 ? 1 : 0, d_pts, a), B(3, Pj, dims * p)}, s))) return r;
      MSMCHK(nbls_msm_decompose_launch((unsigned)n_in, dims, d_scalars, Ks, s));
    } else if ((r = run(ctx, g2 ? P_G2_TO_PROJ : P_G1_TO_PROJ, n, {B(g2 ? 1 : 0, d_pts, a), B(3, Pj, p)}, s))) return r;
    MSMCHK(nbls_msm_keys_launch((unsigned)n, nwin, split ? Ks : (const uint8_t*)d_scalars, kin, vin, s));
    MSMCHK(nbls_msm_sort_launch(tmp, &tmp_bytes, kin, kout, vin, vout, m, 17, s));
    MSMCHK(nbls_msm_gather_launch(m, (unsigned)p, vout, Pj, P, s));
    MSMCHK(nbls_msm_rank_launch(tmp, &scan_bytes, m, kout, pos, counters, s));
    uint32_t maxrun = 0;
    HIPCHK(hipMemcpyAsync(&maxrun, counters, 4, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
    int round = 0;
    for (size_t d = 1; d < maxrun; d *= 2, round++) {
      const size_t bound = m / (d + 1) + 1;      // every pair owns d + 1 list positions of its own
      uint32_t* c = counters + 1 + round;
      MSMCHK(nbls_msm_pairs_launch(m, (unsigned)d, kout, pos, list, c, s));
      // P[j] += P[j + d] for the listed j, in place: the step program addresses its buffers through the list (KernelArgs.item_index);
      // a listed element is never the partner of another one (ranks are multiples of 2d), so no pair touches another pair's points
      if ((r = run(ctx, g2 ? P_G2_ADD_AB : P_G1_ADD_AB, bound, {B(3, P, p), B(4, P + d * p, p), B(5, P, p)}, s, c, list))) return r;
    }
    MSMCHK(nbls_msm_heads_launch(m, (unsigned)p, kout, P, Bk, s));
  }
  MSMCHK(nbls_msm_bitsel_launch(nwin, (unsigned)p, Bk, G, s));
  uint8_t *src = G, *dst = Gh;
  for (size_t cnt = ng; cnt > (size_t)nwin * C; cnt /= 2) {
    if ((r = run(ctx, g2 ? P_G2_ADD2 : P_G1_ADD2, cnt / 2, {B(3, src, 2 * p), B(5, dst, p)}, s))) return r;
    std::swap(src, dst);
  }
  uint8_t* S = A;   // per-window sums
  if ((r = run(ctx, g2 ? P_G2_HORNER : P_G1_HORNER, nwin, {B(3, src, C * p), B(5, S, p)}, s))) return r;
  HIPCHK(hipMemcpyAsync(acc, S + (size_t)(nwin - 1) * p, p, hipMemcpyDeviceToDevice, s));
  for (int w = (int)nwin - 2; w >= 0; w--)
    if ((r = run(ctx, g2 ? P_G2_SHIFTADD : P_G1_SHIFTADD, 1, {B(3, acc, p), B(4, S + (size_t)w * p, p), B(5, acc, p)}, s))) return r;
  if ((r = run(ctx, g2 ? P_G2_NORM : P_G1_NORM, 1, {B(3, acc, p), B(4, N, RAW)}, s))) return r;
  if ((r = run_inv_buf(ctx, 1, N, NI, s))) return r;
  return run(ctx, g2 ? P_G2_TO_AFFINE : P_G1_TO_AFFINE, 1, {B(3, acc, p), B(4, NI, RAW), B(2, d_out, a), B(7, d_status, 1)}, s);
}
static unsigned scalars_bit_length(size_t n, const uint8_t* k32) {   // max over the batch
  int lead = 32;   // leading zero bytes common to all scalars
  for (size_t i = 0; i < n && lead; i++) { int z = 0; while (z < lead && k32[32 * i + z] == 0) z++; lead = z; }
  if (lead == 32) return 1;
  uint8_t top = 0; for (size_t i = 0; i < n; i++) top |= k32[32 * i + lead];
  unsigned bits = 8 * (31 - lead); while (top) { bits++; top >>= 1; }
  return bits;
}
static int msm_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* pts, const uint8_t* scalars32, uint8_t* out, int8_t* status) {
  if (!ctx || !out || (n && (!pts || !scalars32))) return NBLS_EINVAL;
  const size_t a = g2 ? 192 : 96;
  LOCKED(ctx); HostIO io{ctx}; void *dp = io.alloc(n * a), *dk = io.alloc(n * 32), *o = io.alloc(a), *st = io.alloc(1); if (!dp || !dk || !o || !st) return NBLS_EHIP;
  if (n) { HIPCHK(hipMemcpyAsync(dp, pts, n * a, hipMemcpyHostToDevice, s)); HIPCHK(hipMemcpyAsync(dk, scalars32, n * 32, hipMemcpyHostToDevice, s)); }
  int r = dev_msm(ctx, g2, n, dp, dk, n ? scalars_bit_length(n, scalars32) : 1, o, st, s); if (r) return r;
  int8_t z = 0; HIPCHK(hipMemcpyAsync(out, o, a, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(&z, st, 1, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
  if (status) *status = z;
  return NBLS_OK;
}
EXPORT int nbls_g1_msm(nbls_ctx* ctx, size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out96, int8_t* status) { return msm_host(ctx, false, n, pts96, scalars32, out96, status); }
EXPORT int nbls_g2_msm(nbls_ctx* ctx, size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out192, int8_t* status) { return msm_host(ctx, true, n, pts192, scalars32, out192, status); }
// device-resident variant: points (affine wire bytes), scalars (32 B big-endian, all < 2^nbits; nbits = 0 means 256), one affine result + int8 status
// in device memory; enqueued on `stream` (NULL = the context's stream) except for one 4-byte read-back in the middle
EXPORT int nbls_msm_dev(nbls_ctx* ctx, int g2, size_t n, const void* d_pts, const void* d_scalars32, unsigned nbits, void* d_out, void* d_status, void* stream) {
  if (!ctx || !d_out || !d_status || (n && (!d_pts || !d_scalars32))) return NBLS_EINVAL;
  std::lock_guard<std::recursive_mutex> g_(ctx->mu); HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  StreamOrder order_(ctx, s);
  return dev_msm(ctx, g2 != 0, n, d_pts, d_scalars32, nbits, d_out, d_status, s);
}

// sign(message_i, key_i) (index.ts:744-752): hashToCurve -> multiply by the key -> affine signature point (the caller
// compresses, PointG2.toSignature index.ts:586-602).  status: 0 ok, 5 key is 0 mod r.
EXPORT int nbls_sign_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, const uint8_t* keys32, uint8_t* out192, int8_t* status) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);   // scratch and I/O staging buffers belong to this call until it returns
  if (!ctx || (n && (!offsets || !out192 || !dst || !keys32))) return NBLS_EINVAL; if (!n) return NBLS_OK;
  LOCKED(ctx); HostIO io{ctx}; void *h = io.alloc(n * 192), *dk = io.alloc(n * 32), *o = io.alloc(n * 192), *st = io.alloc(n);
  if (!h || !dk || !o || !st) return NBLS_EHIP;
  io.secret(dk, n * 32);
  uint8_t* d; int r = dev_expand(ctx, n, msgs, offsets, dst, dst_len, &d, s); if (r) return r;
  HIPCHK(hipMemcpyAsync(dk, keys32, n * 32, hipMemcpyHostToDevice, s));
  if ((r = sign_points(ctx, n, d, h, dk, o, st, s))) return r;
  std::vector<int8_t> tmp(n);
  if ((r = io.wipe(s))) return r;
  HIPCHK(hipMemcpyAsync(out192, o, n * 192, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(tmp.data(), st, n, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
  for (size_t i = 0; i < n; i++) if (scalar_is_zero_mod_r(keys32 + 32 * i)) tmp[i] = 5;
  if (status) memcpy(status, tmp.data(), n);
  return NBLS_OK;
}

// The domain-separation tag of the device-resident hash entry points, kept on the device between calls (a service works under one tag): no copy, no synchronisation in the steady
// state.  A DST longer than 255 bytes is replaced by its SHA-256 digest (RFC 9380 5.3.3), as in dev_expand and the reference's expand_message_xmd (index.ts:207-231).
static int dst_on_device(nbls_ctx* ctx, const uint8_t* dst, size_t* dst_len, hipStream_t s, uint8_t** dd) {
  uint8_t dst_hash[32];
  if (*dst_len > 255) { Sha256 c; c.update((const uint8_t*)"H2C-OVERSIZE-DST-", 17); c.update(dst, *dst_len); c.final(dst_hash); dst = dst_hash; *dst_len = 32; }
  std::lock_guard<std::recursive_mutex> g_(ctx->mu); HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->dst_dev) HIPCHK(hipMalloc(&ctx->dst_dev, 256));
  if (ctx->dst_host.size() != *dst_len || memcmp(ctx->dst_host.data(), dst, *dst_len)) {
    HIPCHK(hipStreamSynchronize(s));   // an earlier call may still read the old tag
    HIPCHK(hipMemcpy(ctx->dst_dev, dst, *dst_len, hipMemcpyHostToDevice));
    ctx->dst_host.assign(dst, dst + *dst_len);
  }
  *dd = ctx->dst_dev;
  return NBLS_OK;
}
// sign with everything resident in HBM (round 5): message bytes + offsets, 32-byte keys -> affine signature points and status bytes; SHA-256 expand_message_xmd, hash-to-G2 and
// the constant-time ladder in one chain on `stream`.  Synchronises (the offsets are validated by the hashing kernel and the verdict is read back).  index.ts:744-752.
EXPORT int nbls_sign_batch_dev(nbls_ctx* ctx, size_t n, const void* d_msgs, const void* d_offsets, const uint8_t* dst, size_t dst_len, const void* d_keys32, void* d_out192, void* d_status, void* stream) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);
  if (!ctx || (n && (!d_offsets || !d_keys32 || !d_out192 || !d_status || !dst))) return NBLS_EINVAL;
  if (!n) return NBLS_OK;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  uint8_t* dd; int r;
  if ((r = dst_on_device(ctx, dst, &dst_len, s, &dd))) return r;
  StreamOrder order_(ctx, s);
  HostIO io{ctx}; io.s = s; void* h = io.alloc(n * 192); if (!h) return NBLS_EHIP;
  uint8_t* du;
  if ((r = need(ctx, 8, n * 256 + 16, &du))) return r;
  uint32_t* d_bad = (uint32_t*)(du + n * 256);
  HIPCHK(hipMemsetAsync(d_bad, 0, 4, s));
  const int e = nbls_xmd_launch((unsigned)n, (const uint8_t*)d_msgs, (const uint8_t*)d_offsets, dd, (unsigned)dst_len, du, 256, d_bad, s);
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  if ((r = sign_points(ctx, n, du, h, d_keys32, d_out192, d_status, s))) return r;
  uint32_t bad = 0; HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
  return bad ? NBLS_EINVAL : NBLS_OK;     // offsets[i + 1] < offsets[i] somewhere
}

// ---- verifyBatch as concurrent sub-batches (round 5) ------------------------------------------------------------------------------------------------------
// Round 4 ran the call in two phases -- decode the keys and hash every message (12.5 ms at 65,536 signatures, most of it ONE chain of dependent launches), read the
// statuses back, then the Miller product of all pairs (9.1 ms) -- and three such calls in flight took 18.4 ms each instead of 22.8: every launch of the chain leaves issue
// slots free (the exponentiation kernels fill the chip 1.33 rounds deep, ACC4 1.07 rounds, every launch ends in a partly filled round, the host reads the statuses in the
// middle).  First attempt of this round: chunks in SEQUENCE, the Miller loops of chunk c beside the hash chain of chunk c + 1 -- slower (24.5 ms with four chunks,
// profiles/round5_verify_sweep_sequential.txt): the hash chain has a latency floor of ~3.7 ms whatever its size (a 377-squaring exponentiation per lane), four chains one after
// the other are 15 ms of it, and sharing the SIMDs with Miller loops stretches them further.  What the in-flight figure really says is that INDEPENDENT chains fill each other's
// holes.  So the signatures are cut into K sub-batches of decreasing size that all start at once, each on a stream of its own: keys -> hash chain (odd sub-batches the other
// way round, so that equal kernels do not meet) -> LINES_PQ -> ACC over its own slice of the scratch arrays; every accumulator lands in ONE array that the in-place product tree
// reduces at the end, and the statuses are read back once, with the result: nothing is decided on the host before the end (an undecodable key only makes the product
// meaningless, and the statuses say so).  index.ts:792-821.
struct VerifyIn {
  const void* d_sig96;      // 96-byte signature, or NULL (a shard without the signature pair)
  const void* d_uniform;    // 256 B of expand_message_xmd output per message, or NULL when the messages themselves are given:
  const void* d_msgs; const void* d_offsets; const uint8_t* dst_dev; unsigned dst_len;
  const void* d_pk48;
};
static std::vector<size_t> verify_plan(nbls_ctx* ctx, size_t n) {
  const size_t K = (size_t)ctx->verify_chunks;
  // every size but the last is a multiple of g: whole groups of accumulators (4), whole wavefronts where the batch is large (64)
  const size_t g = n >= 4096 ? 64 : 4;
  if (K < 2 || K > 16 || n < (size_t)ctx->verify_pipe_min || n < 2 * g * K || n + 128 > LINES_CHUNK) return {n};   // (a call's line tables are one allocation of at most LINES_CHUNK)
  // sizes fall linearly from the first chunk to the last (verify_last_pct per cent of n): the last chunk's Miller loops run with nothing beside them, so it is the small one
  double last = (double)n * (double)ctx->verify_last_pct / 100.0, first = 2.0 * (double)n / (double)K - last;
  if (first < last) first = last = (double)n / (double)K;
  std::vector<size_t> v(K); size_t sum = 0;
  for (size_t c = 0; c + 1 < K; c++) { v[c] = (((size_t)(first + (last - first) * (double)c / (double)(K - 1)) + g - 1) / g) * g; sum += v[c]; if (sum >= n) return {n}; }
  v[K - 1] = n - sum;
  return v;
}
static int pipe_event(nbls_ctx* ctx, size_t i, hipEvent_t* e) {
  while (ctx->pipe_ev.size() <= i) { hipEvent_t ev = nullptr; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); ctx->pipe_ev.push_back(ev); }
  *e = ctx->pipe_ev[i];
  return NBLS_OK;
}
static int ensure_half_stream(nbls_ctx* ctx) {
  if (!ctx->half_stream && (hipStreamCreateWithFlags(&ctx->half_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_half_fork, hipEventDisableTiming) != hipSuccess ||
                            hipEventCreateWithFlags(&ctx->ev_half_join, hipEventDisableTiming) != hipSuccess)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
  return NBLS_OK;
}
// final_exp = 1: the product's final exponentiation as 576 wire bytes in `out` (host); 0: the product itself as wire bytes at d_out (device; a shard's partial).
// st: n statuses of the keys (+ 1 of the signature) as the decoders wrote them; *bad_offsets: the message offsets were not monotonic.
static int pipe_stream(nbls_ctx* ctx, size_t i, hipStream_t* st) {
  // NBLS_VERIFY_PRIO=1: the streams of the later sub-batches get the lowest priority the device offers (experiment: does the first sub-batch then finish its hash chain early?)
  static const long prio_mode = env_long("NBLS_VERIFY_PRIO", 0);
  while (ctx->pipe_streams.size() <= i) {
    hipStream_t x = nullptr;
    if (prio_mode) { int lo = 0, hi = 0; HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi)); HIPCHK(hipStreamCreateWithPriority(&x, hipStreamNonBlocking, lo)); }
    else HIPCHK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    ctx->pipe_streams.push_back(x);
  }
  *st = ctx->pipe_streams[i];
  return NBLS_OK;
}
static int verify_pipeline(nbls_ctx* ctx, size_t n, const VerifyIn& in, int final_exp, void* d_out, uint8_t* out, std::vector<int8_t>& st, int* bad_offsets, void* stream) {
  const size_t np = n + (in.d_sig96 ? 1 : 0);
  st.assign(np + 8, 0);
  std::lock_guard<std::recursive_mutex> g_(ctx->mu); HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  StreamOrder order_(ctx, s);
  int r;
  uint8_t *G1, *G2, *ST, *O, *du = nullptr;
  if ((r = need(ctx, 10, (n + 1) * (96 + 192) + (n + 1) + 576 + 64 + 16, &G1))) return r;
  G2 = G1 + (n + 1) * 96; O = G2 + (n + 1) * 192; ST = O + 576;
  uint32_t* d_bad = (uint32_t*)(ST + ((np + 3) & ~(size_t)3));   // one word behind the statuses
  if (!in.d_uniform && (r = need(ctx, 8, n * 256, &du))) return r;
  const std::vector<size_t> plan = verify_plan(ctx, n);
  const size_t K = plan.size();
  if (!ctx->ev_fork && hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
  HIPCHK(hipMemsetAsync(d_bad, 0, 4, s));
  HIPCHK(hipEventRecord(ctx->ev_fork, s));
  ForkGuard fork_guard;
  if (in.d_sig96) {
    // normP2: PointG2.fromSignature for the ONE signature, on the side stream with its own scratch (its Fp2 exponentiation on two lanes is pure latency)
    if (!ctx->side) {
      if (hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
          hipMalloc(&ctx->side_scratch, (6 + 2 * POW_TAB) * RAW) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
    }
    uint8_t *X = ctx->side_scratch, *Rr = X + 2 * RAW, *Cd = Rr + 2 * RAW, *pw = Cd + 2 * RAW;
    HIPCHK(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    if ((r = run(ctx, P_G2_DEC_A, 1, {B(0, in.d_sig96, 96), B(3, X, 2 * RAW), B(4, Rr, 2 * RAW)}, ctx->side))) return r;
    if ((r = run_pow(ctx, 1, 1, Rr, Cd, ctx->side, pw))) return r;
    if ((r = run(ctx, P_G2_DEC_B, 1, {B(0, in.d_sig96, 96), B(3, X, 2 * RAW), B(4, Rr, 2 * RAW), B(5, Cd, 2 * RAW), B(6, G2 + n * 192, 192), B(7, ST + n, 1)}, ctx->side))) return r;
    HIPCHK(hipEventRecord(ctx->ev_join, ctx->side));
  }
  if ((r = ensure_scratch(ctx, np))) return r;
  size_t m_off = 0;
  if (K == 1) {
    // one sub-batch: keys on a second stream beside the hash chain (both contain a per-lane exponentiation kernel that leaves issue slots free), then the Miller loops of all pairs
    if (!ctx->side2 && (hipStreamCreateWithFlags(&ctx->side2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming) != hipSuccess)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
    HIPCHK(hipStreamWaitEvent(ctx->side2, ctx->ev_fork, 0));
    if ((r = dev_decompress(ctx, false, n, in.d_pk48, G1, ST, ctx->side2, 14, 17))) return r;      // normP1: PointG1.fromHex; scratch slots 14..16 / 17
    HIPCHK(hipEventRecord(ctx->ev_join2, ctx->side2));
    const uint8_t* uni = (const uint8_t*)in.d_uniform;
    if (!uni) {
      const int e = nbls_xmd_launch((unsigned)n, in.d_msgs, in.d_offsets, in.dst_dev, in.dst_len, du, 256, d_bad, s);
      if (e) { ctx->last_hip = e; return NBLS_EHIP; }
      uni = du;
    }
    if ((r = dev_hash_to_g2(ctx, n, uni, G2, s))) return r;                                       // normP2Hash: PointG2.hashToCurve; slots 0..6 / 11 / 13 / 18 / 19
    HIPCHK(hipStreamWaitEvent(s, ctx->ev_join2, 0));
    if (in.d_sig96) {
      HIPCHK(hipMemcpyAsync(G1 + n * 96, ctx->neg_g1, 96, hipMemcpyDeviceToDevice, s));            // PointG1.BASE.negate()
      HIPCHK(hipStreamWaitEvent(s, ctx->ev_join, 0));
    }
    if ((r = miller_values(ctx, np, G1, G2, &m_off, s))) return r;
  } else {
    // scratch is sized once for the whole call (the sub-batches work on slices of it): grow it before anything is in flight
    if ((r = ensure_lines(ctx, np + 4 * K + 4))) return r;
    if (np + 4 * K + 4 > ctx->cap_L) return NBLS_EINVAL;      // (more pairs than one allocation of line tables holds: verify_plan does not cut such calls)
    size_t o = 0;
    for (size_t c = 0; c < K; c++) {
      const size_t nc = plan[c]; const bool last = c + 1 == K;
      hipStream_t sc = s; hipEvent_t evc;
      if (c && (r = pipe_stream(ctx, c - 1, &sc))) return r;
      if ((r = pipe_event(ctx, c, &evc))) return r;
      if (c) HIPCHK(hipStreamWaitEvent(sc, ctx->ev_fork, 0));
      auto keys = [&]() { return dev_decompress(ctx, false, nc, (const uint8_t*)in.d_pk48 + o * 48, G1 + o * 96, ST + o, sc, 14, 17, 0, o, n); };   // normP1: PointG1.fromHex
      auto hash = [&]() -> int {                                                                                                                  // normP2Hash: PointG2.hashToCurve
        const uint8_t* uni = (const uint8_t*)in.d_uniform + o * 256;
        if (!in.d_uniform) {
          const int e = nbls_xmd_launch((unsigned)nc, in.d_msgs, (const uint32_t*)in.d_offsets + o, in.dst_dev, in.dst_len, du + o * 256, 256, d_bad, sc);
          if (e) { ctx->last_hip = e; return NBLS_EHIP; }
          uni = du + o * 256;
        }
        return dev_hash_to_g2(ctx, nc, uni, G2 + o * 192, sc, o, n);
      };
      // experiment: the first (large) sub-batch decodes its keys on a stream of its own beside its hash chain, as round 4 did for the whole call: both contain an exponentiation kernel that leaves
      // issue slots free, and in sequence they put 3 ms in front of the longest chain of the call -- measured no better either (profiles/round5_ab_verify2.txt), off by default: NBLS_VERIFY_KEYS_SIDE=1
      static const bool keys_side = env_long("NBLS_VERIFY_KEYS_SIDE", 0) != 0;
      if (c == 0 && keys_side) {
        if (!ctx->side2 && (hipStreamCreateWithFlags(&ctx->side2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming) != hipSuccess)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
        HIPCHK(hipStreamWaitEvent(ctx->side2, ctx->ev_fork, 0));
        hipStream_t keep = sc; sc = ctx->side2;
        if ((r = keys())) return r;
        sc = keep;
        HIPCHK(hipEventRecord(ctx->ev_join2, ctx->side2));
        if ((r = hash())) return r;
        HIPCHK(hipStreamWaitEvent(sc, ctx->ev_join2, 0));
      } else if (c & 1) { if ((r = hash()) || (r = keys())) return r; }
      else { if ((r = keys()) || (r = hash())) return r; }
      size_t cc = nc;
      if (last && in.d_sig96) {
        HIPCHK(hipMemcpyAsync(G1 + n * 96, ctx->neg_g1, 96, hipMemcpyDeviceToDevice, sc));         // PointG1.BASE.negate()
        HIPCHK(hipStreamWaitEvent(sc, ctx->ev_join, 0));
        cc++;
      }
      // line tables per accumulator: four where a quarter of the sub-batch's pairs still are thousands of items, fewer where only the length of one wavefront's instruction stream counts
      const size_t GR = cc >= 8192 ? 4 : cc >= 2048 ? 2 : 1;
      const ProgId acc = GR == 4 ? P_ACC4_RAW : GR == 2 ? P_ACC2_RAW : P_ACC_RAW;
      const size_t gg = (cc + GR - 1) / GR;
      uint8_t* Lc = ctx->L + (o + 4 * c) * LINE_BYTES;      // its own line tables (+ up to three unit tables behind them)
      // experiment: the FIRST (large) sub-batch's Miller loops run alone once the small ones are done; as two halves on two streams, like nbls_pairing_batch_dev, so that the partly filled
      // last round of LINES / ACC of one half runs under the other -- measured NO better (profiles/round5_ab_verify2.txt: 12 % worse with the default split, even with a 75 / 25 split),
      // so the switch NBLS_VERIFY_HALVES=1 is off by default
      static const bool halves_on = env_long("NBLS_VERIFY_HALVES", 0) != 0;
      const size_t h = (c == 0 && halves_on && cc >= 2 * ctx->halves_min) ? (((cc / 2) + GR * 64 - 1) / (GR * 64)) * (GR * 64) : cc;   // whole groups, whole wavefronts
      if (h < cc) {
        if ((r = ensure_half_stream(ctx))) return r;
        HIPCHK(hipEventRecord(ctx->ev_half_fork, sc)); HIPCHK(hipStreamWaitEvent(ctx->half_stream, ctx->ev_half_fork, 0));
      }
      for (size_t lo = 0; lo < cc; lo += h) {
        const size_t part = lo ? cc - lo : h, pg = (part + GR - 1) / GR;
        hipStream_t sh = lo ? ctx->half_stream : sc;
        if ((r = run(ctx, P_LINES_PQ, part, {B(0, G1 + (o + lo) * 96, 96), B(1, G2 + (o + lo) * 192, 192), B(3, Lc + lo * LINE_BYTES, LINE_BYTES)}, sh))) return r;
        for (size_t k = part; k < GR * pg; k++) HIPCHK(hipMemcpyAsync(Lc + (lo + k) * LINE_BYTES, ctx->unit_lines, LINE_BYTES, hipMemcpyDeviceToDevice, sh));
        if ((r = run(ctx, acc, pg, {B(3, Lc + lo * LINE_BYTES, GR * LINE_BYTES), B(5, ctx->F + (m_off + lo / GR) * F12, F12)}, sh))) return r;
        if (lo) break;
      }
      if (h < cc) { HIPCHK(hipEventRecord(ctx->ev_half_join, ctx->half_stream)); HIPCHK(hipStreamWaitEvent(sc, ctx->ev_half_join, 0)); }
      m_off += gg;
      if (c) { HIPCHK(hipEventRecord(evc, sc)); HIPCHK(hipStreamWaitEvent(s, evc, 0)); }      // (enqueued on s behind sub-batch 0's own work)
      o += nc;
    }
  }
  uint8_t* res = ctx->F;
  if ((r = reduce_product(ctx, m_off, &res, s))) return r;
  if ((r = finish_single(ctx, res, final_exp, final_exp ? (void*)O : d_out, s))) return r;
  HIPCHK(hipMemcpyAsync(st.data(), ST, ((np + 3) & ~(size_t)3) + 4, hipMemcpyDeviceToHost, s));
  if (final_exp) HIPCHK(hipMemcpyAsync(out, O, 576, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  fork_guard.armed = false;      // synchronised: every forked stream was joined into s
  uint32_t bad = 0; memcpy(&bad, st.data() + ((np + 3) & ~(size_t)3), 4);
  if (bad_offsets) *bad_offsets = bad != 0;
  st.resize(np);
  return NBLS_OK;
}
static bool verify_pipe_enabled() { static const bool on = env_long("NBLS_VERIFY_PIPE", 1) != 0; return on; }
static bool fp12_wire_is_one(const uint8_t* out) { bool one = out[47] == 1; for (int i = 0; i < 576 && one; i++) if (i != 47 && out[i]) one = false; return one; }   // exp.equals(Fp12.ONE)
// the whole of verifyBatch behind the pipeline: decide from the statuses as the reference does (index.ts:792-821)
static int verify_decide(const std::vector<int8_t>& st, size_t n, const uint8_t* out, int* ok, int8_t* pk_status) {
  if (pk_status) memcpy(pk_status, st.data(), n);
  for (int8_t v : st) if (v > 1) return NBLS_EDECODE;                  // the reference throws before its try block
  for (int8_t v : st) if (v == 1) { *ok = 0; return NBLS_OK; }          // zero point -> pairing() throws -> false
  *ok = fp12_wire_is_one(out) ? 1 : 0;
  return NBLS_OK;
}
// verifyBatch(signature, messages, publicKeys) on wire inputs (index.ts:792-821): every message hashes to its own point
// (hex inputs are distinct objects in the reference), n pairings e(pk_i, H(m_i)) times e(-G, sig), one final exponentiation.
//   *ok = 1 / 0.  Return code: NBLS_OK, or NBLS_EDECODE when the reference would throw while decoding (before its try block):
//   invalid signature or public key encoding / subgroup.  A zero public key or zero signature gives *ok = 0 (pairing throws
//   inside the try block, index.ts:716, 818-820).
EXPORT int nbls_verify_batch_dev_inputs(nbls_ctx* ctx, size_t n, const void* d_sig96, const void* d_uniform, const void* d_pk48, int* ok, int8_t* pk_status, void* stream);
EXPORT int nbls_verify_batch(nbls_ctx* ctx, size_t n, const uint8_t* sig96, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* pk48, const uint8_t* dst, size_t dst_len, int* ok) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);   // scratch and I/O staging buffers belong to this call until it returns
  if (!ctx || !ok || !n || !sig96 || !offsets || !pk48 || !dst) return NBLS_EINVAL;
  void *d_sig, *d_uni, *d_pk;
  {
    LOCKED(ctx);
    uint8_t *b, *c; int r;
    if ((r = dev_expand(ctx, n, msgs, offsets, dst, dst_len, &b, s))) return r;
    if ((r = need(ctx, 9, n * 48 + 96, &c))) return r;
    HIPCHK(hipMemcpyAsync(c, pk48, n * 48, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c + n * 48, sig96, 96, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    d_sig = c + n * 48; d_uni = b; d_pk = c;
  }
  return nbls_verify_batch_dev_inputs(ctx, n, d_sig, d_uni, d_pk, ok, nullptr, nullptr);
}
// verifyBatch with EVERYTHING resident in HBM (bench.py's verifyBatch value): signature, the message bytes with their n + 1 offsets (uint32, relative to d_msgs),
// compressed keys.  SHA-256 expand_message_xmd (index.ts:207-231) runs first, on the same stream, then the call continues as nbls_verify_batch_dev_inputs.
EXPORT int nbls_verify_batch_msgs_dev(nbls_ctx* ctx, size_t n, const void* d_sig96, const void* d_msgs, const void* d_offsets, const void* d_pk48, const uint8_t* dst, size_t dst_len, int* ok, void* stream) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);
  if (!ctx || !ok || !n || !d_sig96 || !d_offsets || !d_pk48 || !dst) return NBLS_EINVAL;
  uint8_t* dd;
  { const int r = dst_on_device(ctx, dst, &dst_len, stream ? (hipStream_t)stream : ctx->stream, &dd); if (r) return r; }
  if (verify_pipe_enabled()) {
    VerifyIn in{d_sig96, nullptr, d_msgs, d_offsets, dd, (unsigned)dst_len, d_pk48};
    std::vector<int8_t> st; uint8_t out[576]; int bad = 0;
    int r = verify_pipeline(ctx, n, in, 1, nullptr, out, st, &bad, stream); if (r) return r;
    if (bad) return NBLS_EINVAL;   // offsets[i + 1] < offsets[i] somewhere (the kernel hashed an empty message there instead of reading 4 GB)
    return verify_decide(st, n, out, ok, nullptr);
  }
  uint8_t* du;
  {
    std::lock_guard<std::recursive_mutex> g_(ctx->mu);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    StreamOrder order_(ctx, s);
    int r;
    if ((r = need(ctx, 8, n * 256 + 16, &du))) return r;
    uint32_t* d_bad = (uint32_t*)(du + n * 256);
    HIPCHK(hipMemsetAsync(d_bad, 0, 4, s));
    const int e = nbls_xmd_launch((unsigned)n, (const uint8_t*)d_msgs, (const uint8_t*)d_offsets, dd, (unsigned)dst_len, du, 256, d_bad, s);
    if (e) { ctx->last_hip = e; return NBLS_EHIP; }
    uint32_t bad = 0; HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
    if (bad) return NBLS_EINVAL;
  }
  return nbls_verify_batch_dev_inputs(ctx, n, d_sig96, du, d_pk48, ok, nullptr, stream);
}
// Same with inputs resident in HBM: signature (96 B), expand_message_xmd outputs (256 B per message), public keys (48 B each).
// decode + hash stage shared by verifyBatch and its multi-GPU shard: keys -> G1 points, messages -> G2 hash points, and (when a
// signature is given) the pair (-G, S) appended; the pairs land in scratch slot 10 (g1 | g2), statuses in st (n or n + 1 entries)
static int verify_stage(nbls_ctx* ctx, size_t n, const void* d_sig96, const void* d_uniform, const void* d_pk48, std::vector<int8_t>& st, void* stream) {
  const size_t np = n + (d_sig96 ? 1 : 0);
  st.assign(np, 0);
  std::lock_guard<std::recursive_mutex> g_(ctx->mu); HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  StreamOrder order_(ctx, s);
  uint8_t *G1, *G2, *ST, *O; int r;
  if ((r = need(ctx, 10, (n + 1) * (96 + 192) + (n + 1) + 576 + 64, &G1))) return r;
  G2 = G1 + (n + 1) * 96; O = G2 + (n + 1) * 192; ST = O + 576;
  if (d_sig96) {
    // normP2: PointG2.fromSignature for the ONE signature, on the side stream with its own scratch (overlaps everything below).
    // The side stream is created on first use: HIP spreads streams over a few hardware queues in creation order, and contexts
    // that only run pairing batches (noble-bls12-381_amd/pipeline.py keeps several in flight) should each get a queue of their own.
    if (!ctx->side) {
      if (hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess || (!ctx->ev_fork && hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess) ||
          hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess || hipMalloc(&ctx->side_scratch, (6 + 2 * POW_TAB) * RAW) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
    }
    uint8_t *X = ctx->side_scratch, *Rr = X + 2 * RAW, *Cd = Rr + 2 * RAW, *pw = Cd + 2 * RAW;
    HIPCHK(hipEventRecord(ctx->ev_fork, s)); HIPCHK(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    if ((r = run(ctx, P_G2_DEC_A, 1, {B(0, d_sig96, 96), B(3, X, 2 * RAW), B(4, Rr, 2 * RAW)}, ctx->side))) return r;
    if ((r = run_pow(ctx, 1, 1, Rr, Cd, ctx->side, pw))) return r;
    if ((r = run(ctx, P_G2_DEC_B, 1, {B(0, d_sig96, 96), B(3, X, 2 * RAW), B(4, Rr, 2 * RAW), B(5, Cd, 2 * RAW), B(6, G2 + n * 192, 192), B(7, ST + n, 1)}, ctx->side))) return r;
    HIPCHK(hipEventRecord(ctx->ev_join, ctx->side));
  }
  // normP1 (PointG1.fromHex of the keys) on a second stream beside normP2Hash (PointG2.hashToCurve of the messages): both chains
  // contain a per-lane exponentiation kernel that fills the chip only two wavefronts deep and issues at half rate, so running
  // them side by side costs little more than the longer one.  Scratch slots 14..16 / 17 for the key chain (0..6 / 11 / 13 / 18 / 19 belong to the hash,
  // 7..9 / 12 hold the staged messages, keys and expand_message_xmd output of the host-buffer entry point).
  static const bool overlap = !env_set("NBLS_VERIFY_NO_OVERLAP");
  if (overlap) {
    if (!ctx->side2 && (hipStreamCreateWithFlags(&ctx->side2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming) != hipSuccess)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
    if (!ctx->ev_fork && hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
    HIPCHK(hipEventRecord(ctx->ev_fork, s)); HIPCHK(hipStreamWaitEvent(ctx->side2, ctx->ev_fork, 0));
    if ((r = dev_decompress(ctx, false, n, d_pk48, G1, ST, ctx->side2, 14, 17))) return r;
    HIPCHK(hipEventRecord(ctx->ev_join2, ctx->side2));
    if ((r = dev_hash_to_g2(ctx, n, d_uniform, G2, s))) return r;
    HIPCHK(hipStreamWaitEvent(s, ctx->ev_join2, 0));
  } else {
    if ((r = dev_decompress(ctx, false, n, d_pk48, G1, ST, s))) return r;                       // normP1: PointG1.fromHex
    if ((r = dev_hash_to_g2(ctx, n, d_uniform, G2, s))) return r;                               // normP2Hash: PointG2.hashToCurve
  }
  if (d_sig96) {
    HIPCHK(hipMemcpyAsync(G1 + n * 96, ctx->neg_g1, 96, hipMemcpyDeviceToDevice, s));         // PointG1.BASE.negate()
    HIPCHK(hipStreamWaitEvent(s, ctx->ev_join, 0));
  }
  HIPCHK(hipMemcpyAsync(st.data(), ST, np, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return NBLS_OK;
}
EXPORT int nbls_verify_batch_dev_inputs(nbls_ctx* ctx, size_t n, const void* d_sig96, const void* d_uniform, const void* d_pk48, int* ok, int8_t* pk_status, void* stream) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);   // scratch and I/O staging buffers belong to this call until it returns
  if (!ctx || !ok || !n || !d_sig96 || !d_uniform || !d_pk48) return NBLS_EINVAL;
  std::vector<int8_t> st;
  uint8_t out[576];
  if (verify_pipe_enabled()) {
    VerifyIn in{d_sig96, d_uniform, nullptr, nullptr, nullptr, 0, d_pk48};
    int r = verify_pipeline(ctx, n, in, 1, nullptr, out, st, nullptr, stream); if (r) return r;
    return verify_decide(st, n, out, ok, pk_status);
  }
  int r = verify_stage(ctx, n, d_sig96, d_uniform, d_pk48, st, stream); if (r) return r;
  if (pk_status) memcpy(pk_status, st.data(), n);
  for (size_t i = 0; i <= n; i++) if (st[i] > 1) return NBLS_EDECODE;       // the reference throws before its try block
  for (size_t i = 0; i <= n; i++) if (st[i] == 1) { *ok = 0; return NBLS_OK; }   // zero point -> pairing() throws -> false
  {
    uint8_t* base = ctx->sb[10];
    r = nbls_miller_product_dev(ctx, n + 1, base, base + (n + 1) * 96, 1, base + (n + 1) * 288, stream);
    if (r) return r;
    std::lock_guard<std::recursive_mutex> g_(ctx->mu);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIPCHK(hipMemcpyAsync(out, base + (n + 1) * 288, 576, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  *ok = fp12_wire_is_one(out) ? 1 : 0;
  return NBLS_OK;
}
// One rank's share of a verifyBatch that is spread over several GPUs: the Miller product of this rank's n
// (key, message) pairs -- times millerLoop(-G, S) on the one rank that passes the signature -- WITHOUT the final exponentiation,
// as 576 wire bytes in device memory.  The ranks exchange their partials (one all-gather) and finish with
// nbls_fp12_product_final_dev.  *zero_flag = 1 when a zero point was met (verifyBatch then answers false; d_out is not written).
EXPORT int nbls_verify_batch_partial_dev(nbls_ctx* ctx, size_t n, const void* d_sig96, const void* d_uniform, const void* d_pk48, void* d_out_fp12, int* zero_flag, int8_t* pk_status, void* stream) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);   // scratch and I/O staging buffers belong to this call until it returns
  if (!ctx || !zero_flag || !n || !d_uniform || !d_pk48 || !d_out_fp12) return NBLS_EINVAL;
  std::vector<int8_t> st;
  if (verify_pipe_enabled()) {
    // the pipeline decides nothing before its end: with an undecodable key or a zero point d_out_fp12 holds a meaningless product (round 4 left it unwritten); callers look at the return code and the flag first
    VerifyIn in{d_sig96, d_uniform, nullptr, nullptr, nullptr, 0, d_pk48};
    int r = verify_pipeline(ctx, n, in, 0, d_out_fp12, nullptr, st, nullptr, stream); if (r) return r;
    if (pk_status) memcpy(pk_status, st.data(), n);
    for (int8_t v : st) if (v > 1) return NBLS_EDECODE;
    *zero_flag = 0;
    for (int8_t v : st) if (v == 1) *zero_flag = 1;
    return NBLS_OK;
  }
  int r = verify_stage(ctx, n, d_sig96, d_uniform, d_pk48, st, stream); if (r) return r;
  if (pk_status) memcpy(pk_status, st.data(), n);
  for (int8_t v : st) if (v > 1) return NBLS_EDECODE;
  *zero_flag = 0;
  for (int8_t v : st) if (v == 1) { *zero_flag = 1; return NBLS_OK; }
  const size_t np = st.size();
  uint8_t* base = ctx->sb[10];
  // the pairs sit at stride n + 1 inside the scratch block whether or not the signature pair is present
  return nbls_miller_product_dev(ctx, np, base, base + (n + 1) * 96, 0, d_out_fp12, stream);
}

// ---- one device's share of a product that is spread over several GPUs, from HOST inputs: the partial stays on this context's device so that
// the caller can move it to the reducing device with hipMemcpyPeer (nbls_multi.cpp) or hand it to a collective.  `*_into`: the partial lands in a caller-owned
// 576-byte buffer on this context's device (what nbls_multi.cpp passes: one buffer per call, so that calls racing on one context cannot see each other's
// partials); the plain names return a buffer owned by the context, valid only until the context's next *_partial call.  The call returns when the partial is complete.
// destination of a partial: the caller's buffer (`*_into`: it must be device memory of at least 576 bytes on the context's device -- checked with
// hipPointerGetAttributes, a wild pointer is refused instead of written through), or -- the original entry points, whose *d_partial is a pure OUT
// parameter again (ABI 2; round 3 had silently made it IN/OUT) -- a buffer owned by the context
static int partial_buffer(nbls_ctx* ctx, void* d_dst, uint8_t** dst) {
  if (d_dst) {
    hipPointerAttribute_t at; memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, d_dst) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != ctx->device) { (void)hipGetLastError(); return NBLS_EINVAL; }
    hipDeviceptr_t base = nullptr; size_t size = 0;      // ... and 576 bytes must remain behind it inside its allocation
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)d_dst) != hipSuccess || (size_t)((uint8_t*)d_dst - (uint8_t*)base) + 576 > size) { (void)hipGetLastError(); return NBLS_EINVAL; }
    *dst = (uint8_t*)d_dst; return NBLS_OK;
  }
  if (!ctx->partial) HIPCHK(hipMalloc(&ctx->partial, 576));
  *dst = ctx->partial;
  return NBLS_OK;
}
static int miller_product_partial_core(nbls_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, int validate, void* d_dst, void** d_partial, int8_t* status) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);
  if (!ctx || (n && (!g1 || !g2))) return NBLS_EINVAL;
  int r;
  if (status) memset(status, 0, n);
  if (validate && n) {
    std::vector<int8_t> st1(n), st2(n);
    if ((r = nbls_g1_validate_batch(ctx, n, g1, st1.data())) || (r = nbls_g2_validate_batch(ctx, n, g2, st2.data()))) return r;
    bool bad = false;
    for (size_t i = 0; i < n; i++) { int8_t c = st1[i] ? st1[i] : (st2[i] ? (int8_t)(10 + st2[i]) : 0); if (status) status[i] = c; bad |= c != 0; }
    if (bad) return NBLS_EDECODE;
  }
  LOCKED(ctx);
  uint8_t* part;
  if ((r = partial_buffer(ctx, d_dst, &part))) return r;
  if (n) {
    if ((r = ensure_io(ctx, n))) return r;
    HIPCHK(hipMemcpyAsync(ctx->io_g1, g1, n * 96, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->io_g2, g2, n * 192, hipMemcpyHostToDevice, s));
  }
  if ((r = nbls_miller_product_dev(ctx, n, ctx->io_g1, ctx->io_g2, 0, part, s))) return r;
  HIPCHK(hipStreamSynchronize(s));
  if (d_partial) *d_partial = part;
  return NBLS_OK;
}
EXPORT int nbls_miller_product_partial(nbls_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, int validate, void** d_partial, int8_t* status) {
  if (!d_partial) return NBLS_EINVAL;
  return miller_product_partial_core(ctx, n, g1, g2, validate, nullptr, d_partial, status);
}
EXPORT int nbls_miller_product_partial_into(nbls_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, int validate, void* d_dst, int8_t* status) {
  if (!d_dst) return NBLS_EINVAL;
  return miller_product_partial_core(ctx, n, g1, g2, validate, d_dst, nullptr, status);
}
EXPORT int nbls_abi_version(void) { return NBLS_ABI_VERSION; }
// every environment switch the library has read so far, with the value in force (config.h); the string lives until the next call on this thread
EXPORT const char* nbls_config_describe(void) { static thread_local std::string s; s = env_describe(); return s.c_str(); }
static int verify_batch_partial_core(nbls_ctx* ctx, size_t n, const uint8_t* sig96 /* or NULL */, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* pk48,
                                     const uint8_t* dst, size_t dst_len, void* d_dst, void** d_partial, int* zero_flag, int8_t* pk_status) {
  std::lock_guard<std::recursive_mutex> whole_call_(ctx ? ctx->mu : g_null_mu);
  if (!ctx || !zero_flag || !n || !offsets || !pk48 || !dst) return NBLS_EINVAL;
  LOCKED(ctx);
  uint8_t *b, *c, *part; int r;
  if ((r = partial_buffer(ctx, d_dst, &part))) return r;
  if ((r = dev_expand(ctx, n, msgs, offsets, dst, dst_len, &b, s))) return r;
  if ((r = need(ctx, 9, n * 48 + 96, &c))) return r;
  HIPCHK(hipMemcpyAsync(c, pk48, n * 48, hipMemcpyHostToDevice, s));
  if (sig96) HIPCHK(hipMemcpyAsync(c + n * 48, sig96, 96, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  if ((r = nbls_verify_batch_partial_dev(ctx, n, sig96 ? c + n * 48 : nullptr, b, c, part, zero_flag, pk_status, nullptr))) return r;
  HIPCHK(hipStreamSynchronize(s));
  if (d_partial) *d_partial = part;
  return NBLS_OK;
}
EXPORT int nbls_verify_batch_partial(nbls_ctx* ctx, size_t n, const uint8_t* sig96, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* pk48,
                                     const uint8_t* dst, size_t dst_len, void** d_partial, int* zero_flag, int8_t* pk_status) {
  if (!d_partial) return NBLS_EINVAL;
  return verify_batch_partial_core(ctx, n, sig96, msgs, offsets, pk48, dst, dst_len, nullptr, d_partial, zero_flag, pk_status);
}
EXPORT int nbls_verify_batch_partial_into(nbls_ctx* ctx, size_t n, const uint8_t* sig96, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* pk48,
                                          const uint8_t* dst, size_t dst_len, void* d_dst, int* zero_flag, int8_t* pk_status) {
  if (!d_dst) return NBLS_EINVAL;
  return verify_batch_partial_core(ctx, n, sig96, msgs, offsets, pk48, dst, dst_len, d_dst, nullptr, zero_flag, pk_status);
}
EXPORT int nbls_context_device(nbls_ctx* ctx) { return ctx ? ctx->device : -1; }
