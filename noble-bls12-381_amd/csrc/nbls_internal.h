// nbls_internal.h -- what the translation units of the runtime share (round 6: csrc/nbls_api.cpp, 2,000 lines with every pipeline in one unit, split into
// runtime.cpp / tuning.cpp / pipelines_pairing.cpp / pipelines_codec.cpp / pipelines_verify.cpp; pool and multi-device handles: nbls_multi.cpp): the context, the
// launch helpers and the device-side pipelines the exported entry points are built from.  Internal functions have hidden visibility (csrc/Makefile: -fvisibility=hidden).
#pragma once
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
extern "C" int nbls_fp_inv_wide_launch(unsigned n, const void* in, void* out, void* stream);   // fp_inv_wide.h: four elements per wavefront, one limb per lane
// g1_wide.h: `sums` times sum_w 2^(shift w) S_w over nwin points, one wavefront each
extern "C" int nbls_g1_wide_combine_launch(const void* S, int nwin, int shift, void* out, unsigned sums, void* stream);
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
// pairs from which the Miller loop runs as LINES + ACC (see nbls_pairing_batch_dev).  Round 4 measured the two programs ahead from 4096 pairs on; on the round-5 / 6 build a 4096-pair call
// -- exactly one wavefront of the fused program (four items) on each of the 1024 SIMDs -- takes 2.19 ms fused against 2.32 ms split, 3072 pairs likewise, and from 4608 pairs on the split
// form wins (tools/ab_split_min.py, profiles/round6_ab_split_min.txt)
static const size_t SPLIT_MILLER_MIN = 4097;
static const size_t LINES_CHUNK = 131072;   // pairs whose line tables are in HBM at a time (3.4 GB of the 288); larger batches run chunk by chunk on the same stream

#define EXPORT extern "C" __attribute__((visibility("default")))
extern std::recursive_mutex g_null_mu;   // locked in place of a context's mutex when the caller passed no context (the call then fails with NBLS_EINVAL)

static const size_t EXPC_MIN_DEFAULT = (size_t)1 << 40;   // items from which the cyclotomic exponentiations run with compressed squarings (expx below): never, unless asked for
struct DevProgram {
  Step* steps = nullptr; u32* descs = nullptr; u32* consts = nullptr;
  const Program* p = nullptr;
  bool wide_ok = false;   // every step is one the one-limb-per-lane interpreter implements (wide_exec.h): launches of at most ctx->wide_max items run on nbls_vm_kernel_wide
  // ahead-of-time kernel of this program (aot.h) and the translated program, when every step's signature is in the kernel's table
  int aot = -1; AotStep* aot_steps = nullptr; u32* aot_descs = nullptr; u32 aot_lds = 0;
};

size_t ls_max();      // defaults of the lane-split thresholds (NBLS_LS_MAX / NBLS_LS2_MAX), tuning.cpp
size_t ls2_max();
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
  // pairs from which a call runs as two halves on two streams.  16,384 since the end of round 6 (8192 before): on today's kernels one stream is faster up to 15,360 pairs
  // (10,240: 3.91 against 4.73 ms; 12,800 - 15,360: 5.05 - 5.24 against 5.67 - 5.9), the halves from 16,384 (6.1 against 6.45; profiles/round6_ab_halves.txt)
  size_t halves_min = env_long("NBLS_HALVES_MIN", 16384) > 0 ? (size_t)env_long("NBLS_HALVES_MIN", 16384) : (size_t)-1;   // NBLS_HALVES_MIN=0: never (profiles of kernels running alone)
  // verifyBatch as a software pipeline (round 5, verify_pipeline): events of the chunks (two each), the "xmd met non-monotonic offsets" flag lives behind the statuses
  std::vector<hipEvent_t> pipe_ev; hipEvent_t ev_pipe_done = nullptr; std::vector<hipStream_t> pipe_streams;
  // nbls_set_tuning(NBLS_TUNE_VERIFY_*)
  long verify_chunks = env_long("NBLS_VERIFY_CHUNKS", 2), verify_last_pct = env_long("NBLS_VERIFY_LAST_PCT", 25), verify_pipe_min = env_long("NBLS_VERIFY_PIPE_MIN", 32768);
  hipStream_t side2 = nullptr; hipEvent_t ev_join2 = nullptr;   // verifyBatch: key decoding runs beside message hashing (their exponentiation kernels are latency-bound and leave issue slots free)
  uint8_t* ident_g1 = nullptr; uint8_t* ident_g2 = nullptr;   // projective identity (0 : 1 : 0), raw
  size_t cap_F = 0, cap_io = 0;
  size_t split_min = SPLIT_MILLER_MIN;   // nbls_set_tuning(NBLS_TUNE_SPLIT_MILLER_MIN)
  // pairs per product call from which eight line tables share an accumulator (below: four).  Measured (tools/ab_acc8.sh): at 65,537 pairs the halves have 4096 groups of eight = 820
  // wavefronts, less than one per SIMD, and the call is slower (24.5 against 23.4 ms); at 2^18 terms 33.2 against 33.8 ms
  // (round 6, on the kernels as they are now: four per accumulator is faster at 2^17 and 2^18 terms as well -- 16.35 against 16.89 ms, 30.9 against 32.2 ms: never by default)
  size_t acc8_min = (size_t)env_long("NBLS_ACC8_MIN", (long)1 << 40);
  // cyclotomic exponentiation with compressed squarings (expx): scratch per item -- compressed powers, decompression scratch, redo flags and list -- and two redo counters (one per half)
  uint8_t *KS = nullptr, *KD = nullptr, *Kflag = nullptr; uint32_t *Klist = nullptr, *Kcount = nullptr;
  size_t expc_min = (size_t)env_long("NBLS_EXPC_MIN", (long)EXPC_MIN_DEFAULT);   // nbls_set_tuning(NBLS_TUNE_EXPC_MIN)
  // nbls_set_tuning(NBLS_TUNE_PT_LS2_MAX): items up to which the G2 point chains run in their two-lane forms (pt_ls2_variant)
  size_t pt_ls2_max = (size_t)env_long("NBLS_PT_LS2_MAX", 4096);
  size_t sac_max = (size_t)env_long("NBLS_G2_SAC_MAX", 6144);                   // nbls_set_tuning(NBLS_TUNE_SAC_MAX): keys up to which sign's ladder is the sign-aligned form (dev_point_mul)
  // round 6: launches of at most wide_max items run the programs that allow it on the one-limb-per-lane interpreter (vm_wide_kernel.hip: one item per workgroup of ceil(W / 4)
  // wavefronts, the Montgomery reduction spread over a row of lanes, two barriers per step) -- the multi-wavefront item form the round-5 review asked for.  Built, bit-exact
  // (tests/test_wide_sim.py, test_gpu_pairing.py::test_one_limb_per_lane_forms) and MEASURED SLOWER than the four-lane forms: the five exponentiations of one final exponentiation
  // take 0.93 ms against 0.66 ms (tools/wide_time.py, profiles/round6_wide_time.txt).  A step is 350 instructions in its rows + ~190 around them where the four-lane form has 660,
  // but every one of them waits for its predecessor (one column per lane: no independent work), and a lone wavefront then pays ~9-11 clocks per instruction instead of ~5.  Off by
  // default: NBLS_WIDE_MAX / NBLS_TUNE_WIDE_MAX (items; 0 = never), NBLS_WIDE_PROGS = 0: only the final exponentiation's programs.  The fixed-exponent powers, where the same
  // idea removes a 196-multiply-add reduction per squaring from ONE lane, are the case that pays (pow_wide.h: 0.7 -> 0.3 ms).
  size_t wide_max = (size_t)env_long("NBLS_WIDE_MAX", 0);
  // round 6: messages from which hash-to-G2 takes its SWU square root by the norm method (two Fp exponentiations, programs P_H2C_NA / NM / NB) instead of one Fp2 exponentiation;
  // nbls_set_tuning(NBLS_TUNE_H2C_NORM_MIN); 0 = always.  Less work (1.96 against 2.53 ms of exponentiation kernels per 65,536 roots) but two dependent exponentiations where
  // there was one: a launch that does not fill the device pays their latency twice (sign of 8192 keys +0.09 ms), one that does, or that runs beside other work as the sub-batches
  // of verifyBatch do (the size counted there is the whole call's), gains (sign of 65,536 keys -0.7 ms, verifyBatch -0.3 ms alone and -0.5 ms with three calls in flight)
  size_t h2c_norm_min = (size_t)env_long("NBLS_H2C_NORM_MIN", 32768);
  // round 6: elements up to which an inversion launch runs with one limb per lane (fp_inv_wide.h; nbls_set_tuning(NBLS_TUNE_INV_WIDE_MAX)).  The form shortens ONE call (a 4096-pairing
  // call 2.19 -> 2.165 ms, one verify 2.76 -> 2.67 ms) at eight times the instructions per element: contexts that are kept busy side by side (nbls_pool_init) lower it to 256
  // (twelve 4096-pairing calls in flight: 3.03 M pairings/s against 2.97 M with 4096)
  size_t inv_wide_max = (size_t)env_long("NBLS_INV_WIDE_MAX", 4096);
  // items up to which the pairing programs run in their four-lane / two-lane forms (tuning.cpp ls_variant; nbls_set_tuning(NBLS_TUNE_LS_MAX / _LS2_MAX)).  Latency forms as well:
  // shorter for ONE call, up to four times the instructions per item -- nbls_pool_init sets both to 0 on its contexts (twelve 1024-pairing calls in flight: 1.52 -> 1.95 M pairings/s)
  size_t ls_max = ::ls_max(), ls2_max = ::ls2_max();
  size_t chain_max = (size_t)env_long("NBLS_CHAIN_MAX", 8192);                  // nbls_set_tuning(NBLS_TUNE_CHAIN_MAX); see run_chain
  u32* qp_table = nullptr;      // multiples of p for the weak reduction (vm_exec.h weak_reduce), device copy
  uint8_t* unit_lines = nullptr;   // a line table whose 68 lines are all 1 (c0 = 1, c1 = c2 = 0): the neutral partner of an odd last pair
  uint8_t* partial = nullptr;   // 576 bytes: the Fp12 partial of the *_partial entry points (multi-GPU reductions)
  uint8_t* L = nullptr; size_t cap_L = 0;   // line tables of the Miller loop (LINE_BYTES each), at most LINES_CHUNK of them
  // the scratch above is shared by every call on this context: a call that uses another stream than its predecessor waits for it (StreamOrder)
  hipStream_t last_stream = nullptr; hipEvent_t ev_last = nullptr; bool ev_last_set = false;
  uint8_t* pinned_out = nullptr; size_t pinned_out_cap = 0;   // the same for the way back (a call's inputs may still be in flight from `pinned` when its outputs are copied)
  uint8_t* pinned = nullptr; size_t pinned_cap = 0;   // page-locked host staging of the host-buffer entry points that pack their inputs (ensure_pinned; nbls_sign_batch)
  int last_hip = 0;
  // optional per-kernel timing (HIP events on the launch stream); slot P_COUNT = inversion kernel
  bool timing = false;
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> tev;
  std::vector<hipEvent_t> ev_pool;   // timing events are recycled (nbls_timing_read returns them here) instead of created per launch
};

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->last_hip = (int)e_; return NBLS_EHIP; } } while (0)

// A chain: several programs executed back to back by ONE launch (aot.h): every wavefront runs them in order for its own items, the values between them pass
// through the HBM scratch buffers the separate launches would use.  Falls back to one launch per program when some program is not on an ahead-of-time kernel,
// when the programs do not share a kernel / the lanes per item, or in checked mode (whose per-launch buffer checks live in run()).
typedef std::initializer_list<std::pair<int, std::pair<const void*, size_t>>> BufList;
struct ChainLink { ProgId id; BufList bufs; };
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

// ---- internal functions (hidden visibility; defined in runtime.cpp, tuning.cpp and the pipelines_*.cpp files)
struct VerifyIn;
bool checked_mode();
bool aot_enabled();
int upload_program(nbls_ctx* ctx, DevProgram& d, const Program& p, const int k);
int upload(nbls_ctx* ctx, ProgId id);
void free_program(DevProgram& d);
bool wide_applies(const nbls_ctx* ctx, const DevProgram& d, int id, size_t n);
hipEvent_t timing_event(nbls_ctx* ctx);
void aot_seg(AotSeg& g, const DevProgram& d, const IOBuf* bufs);
int run_dev(nbls_ctx* ctx, const DevProgram& d, int id, size_t n, std::initializer_list<std::pair<int, std::pair<const void*, size_t>>> bufs, hipStream_t s, const uint32_t* n_dev,
    const uint32_t* item_index);
int run(nbls_ctx* ctx, ProgId id, size_t n, std::initializer_list<std::pair<int, std::pair<const void*, size_t>>> bufs, hipStream_t s, const uint32_t* n_dev = nullptr,
    const uint32_t* item_index = nullptr);
int run_inv(nbls_ctx* ctx, size_t n, hipStream_t s);
bool chains_enabled();
int run_chain(nbls_ctx* ctx, size_t n, std::initializer_list<ChainLink> links, hipStream_t s);
int ensure_scratch(nbls_ctx* ctx, size_t n);
int ensure_expc_scratch(nbls_ctx* ctx);
int ensure_io(nbls_ctx* ctx, size_t n);
int ensure_lines(nbls_ctx* ctx, size_t n);
int need(nbls_ctx* ctx, int i, size_t bytes, uint8_t** out);
int ensure_side(nbls_ctx* ctx);
int ensure_side2(nbls_ctx* ctx);
int ensure_pinned(nbls_ctx* ctx, size_t bytes);
int ensure_pinned_out(nbls_ctx* ctx, size_t bytes);
size_t pow_wide_max();
int run_pow(nbls_ctx* ctx, int which, size_t n, const void* in, void* out, hipStream_t s, uint8_t* scratch = nullptr);
int run_inv_buf(nbls_ctx* ctx, size_t n, const void* in, void* out, hipStream_t s);
ProgId ls_variant(nbls_ctx* ctx, ProgId id, size_t n);
ProgId pt_ls2_variant(nbls_ctx* ctx, ProgId id, size_t n);
int reduce_product(nbls_ctx* ctx, size_t n, uint8_t** result, hipStream_t s);
int expx(nbls_ctx* ctx, size_t n, uint8_t* in, uint8_t* out, hipStream_t s);
int final_exp_pipeline(nbls_ctx* ctx, size_t n, uint8_t* f_raw, void* d_out, hipStream_t s);
int finish_single(nbls_ctx* ctx, uint8_t* f_raw, int final_exp, void* d_out, hipStream_t s);
int pairing_core(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_g2, int with_final_exp, void* d_out, hipStream_t s, bool two_programs);
int miller_values(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_g2, size_t* m_out, hipStream_t s);
int acc_prepared(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_tables, size_t table_stride, hipStream_t s);
int partial_buffer(nbls_ctx* ctx, void* d_dst, uint8_t** dst);
int miller_product_partial_core(nbls_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, int validate, void* d_dst, void** d_partial, int8_t* status);
int dev_validate(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, void* d_status, hipStream_t s);
int dev_decompress(nbls_ctx* ctx, bool g2, size_t n, const void* d_in, void* d_out, void* d_status, hipStream_t s, int slot0 = 0, int pow_slot = 11, int mode = 0, size_t io = 0, size_t ntot = 0);
int dev_clear_g2(nbls_ctx* ctx, size_t n, void* in, uint8_t* base, uint8_t* S, void* out, void* N, hipStream_t s);
int dev_hash_to_g2(nbls_ctx* ctx, size_t n, const void* d_uniform, void* d_out, hipStream_t s, size_t io = 0, size_t ntot = 0, uint8_t** proj = nullptr);
int dev_point_sum(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, void* d_out, void* d_status, hipStream_t s);
int decompress_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* in, uint8_t* out, int8_t* status);
int dev_expand(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offs, const uint8_t* dst, size_t dst_len, uint8_t** d_uniform, hipStream_t s, unsigned len_in_bytes = 256);
int sum_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* pts, uint8_t* out, int8_t* status);
int dev_hash_to_g1(nbls_ctx* ctx, int count, size_t n, const void* d_uniform, void* d_out, hipStream_t s);
int dev_encode_to_g2(nbls_ctx* ctx, size_t n, const void* d_uniform, void* d_out, hipStream_t s);
int hash_curve_host(nbls_ctx* ctx, int mode, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out);
int compress_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* aff, uint8_t* out);
int decode_host(nbls_ctx* ctx, int kind /* 0 g1.fromHex, 1 g2.fromHex, 2 g2.fromSignature */, size_t n, const uint8_t* in, size_t len, uint8_t* out, int8_t* status);
int encode_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* aff, const int8_t* zero, int compressed, uint8_t* out);
int clear_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* aff, uint8_t* out, int8_t* status);
int dev_point_mul(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, size_t pt_stride, const void* d_scalars, void* d_out, void* d_status, hipStream_t s, bool allow_fixed = true,
    bool in_subgroup = false, uint8_t* recoded = nullptr);
int ensure_g1_fixed(nbls_ctx* ctx, hipStream_t s);
int sign_points(nbls_ctx* ctx, size_t n, const void* d_uniform, void* h, const void* d_keys32, void* d_out192, void* d_status, hipStream_t s);
bool scalar_is_zero_mod_r(const uint8_t* k32);
int mul_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* pts, const uint8_t* scalars32, uint8_t* out, int8_t* status);
int dev_msm(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, const void* d_scalars, unsigned nbits, void* d_out, void* d_status, hipStream_t s);
unsigned scalars_bit_length(size_t n, const uint8_t* k32);
int msm_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* pts, const uint8_t* scalars32, uint8_t* out, int8_t* status);
int dst_on_device(nbls_ctx* ctx, const uint8_t* dst, size_t* dst_len, hipStream_t s, uint8_t** dd);
std::vector<size_t> verify_plan(nbls_ctx* ctx, size_t n);
int pipe_event(nbls_ctx* ctx, size_t i, hipEvent_t* e);
int ensure_half_stream(nbls_ctx* ctx);
int pipe_stream(nbls_ctx* ctx, size_t i, hipStream_t* st);
int verify_pipeline(nbls_ctx* ctx, size_t n, const VerifyIn& in, int final_exp, void* d_out, uint8_t* out, std::vector<int8_t>& st, int* bad_offsets, void* stream);
bool verify_pipe_enabled();
bool fp12_wire_is_one(const uint8_t* out);
int verify_decide(const std::vector<int8_t>& st, size_t n, const uint8_t* out, int* ok, int8_t* pk_status);
int verify_stage(nbls_ctx* ctx, size_t n, const void* d_sig96, const void* d_uniform, const void* d_pk48, std::vector<int8_t>& st, void* stream);
int verify_batch_partial_core(nbls_ctx* ctx, size_t n, const uint8_t* sig96 /* or NULL */, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* pk48,
                                     const uint8_t* dst, size_t dst_len, void* d_dst, void** d_partial, int* zero_flag, int8_t* pk_status);
