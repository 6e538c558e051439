// tuning.cpp -- which form of a program a launch takes (lane-split thresholds), the run-time tuning keys, per-kernel timing and the program statistics of the C ABI.
#include "nbls_internal.h"

// F (n raw Fp12) -> one element in F[0]; returns pointer to the buffer holding the product
// Launches of at most one wavefront per SIMD take the time of one wavefront's instruction stream, so up to LS_MAX items (one item per wavefront
// on 1024 SIMDs) the lane-split variants run: the same formulas with every lane-op's products shared by four lanes, the columns summed across them before the one
// reduction (ahead-of-time kernels nbls_aot_miller_ls / nbls_aot_expx_ls, aot.h NBLS_AOT_LS_KERNELS; on the interpreter nbls_vm_kernel_ls4).  Measured
// (tools/ab_ls.sh): one pairing 1.72 against 2.09 ms, 1024 pairings 1.79 against 2.12 ms.  NBLS_LS_MAX overrides (0 = the throughput forms at every size).
size_t ls_max() { static const size_t v = (size_t)env_long("NBLS_LS_MAX", 1024); return v; }
// round 5: from LS_MAX + 1 to LS2_MAX items (two items per wavefront on 1024 SIMDs) the TWO-lane forms (nbls_aot_miller_ls2 / nbls_aot_expx_ls2; no interpreter form exists, so they are
// used only where the program is bound to its ahead-of-time kernel).  Measured (tools/ab_ls2.sh): 2048 pairings 1.9 against 2.17 ms.  NBLS_LS2_MAX = 0 switches them off.
size_t ls2_max() { static const size_t v = (size_t)env_long("NBLS_LS2_MAX", 2048); return v; }
ProgId ls_variant(nbls_ctx* ctx, ProgId id, size_t n) {
  if (n <= ctx->wide_max && upload(ctx, id) == NBLS_OK && wide_applies(ctx, ctx->prog[id], (int)id, n)) return id;      // the one-limb-per-lane form runs the plain program
  if (n <= ctx->ls_max) {
    switch (id) {
      case P_MILLER_BYTES: return P_MILLER_BYTES_LS;
      case P_MILLER_RAW: return P_MILLER_RAW_LS;
      case P_MILLER_FE: return P_MILLER_FE_LS;
      case P_EXPX: return P_EXPX_LS;
      default: return id;
    }
  }
  if (n <= ctx->ls2_max) {
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
ProgId pt_ls2_variant(nbls_ctx* ctx, ProgId id, size_t n) {
  if (n <= ctx->wide_max && upload(ctx, id) == NBLS_OK && wide_applies(ctx, ctx->prog[id], (int)id, n)) return id;
  if (n > ctx->pt_ls2_max) return id;
  const ProgId v = id == P_H2C_C1 ? P_H2C_C1_LS2 : id == P_H2C_C2 ? P_H2C_C2_LS2 : id == P_G2_MUL_SAC ? P_G2_MUL_SAC_LS2 : id;
  if (v != id && upload(ctx, v) == NBLS_OK && ctx->prog[v].aot >= 0) return v;   // (no interpreter form of the two-lane split exists)
  return id;
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
  ka.nsteps = (u32)d.p->steps.size(); ka.nconst = d.p->nconst; ka.W = d.p->W; ka.G = d.p->G; ka.slot_bytes = d.p->slot_bytes; ka.inst_bytes = d.p->inst_bytes();
  ka.shared_consts = d.p->shared_consts ? 1u : 0u; ka.lsplit = d.p->lsplit; ka.n_items = (u32)n;
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
    case NBLS_TUNE_H2C_NORM_MIN: if (value < 0) return NBLS_EINVAL; ctx->h2c_norm_min = (size_t)value; return NBLS_OK;
    case NBLS_TUNE_INV_WIDE_MAX: if (value < 0) return NBLS_EINVAL; ctx->inv_wide_max = (size_t)value; return NBLS_OK;
    case NBLS_TUNE_LS_MAX: if (value < 0) return NBLS_EINVAL; ctx->ls_max = (size_t)value; return NBLS_OK;
    case NBLS_TUNE_LS2_MAX: if (value < 0) return NBLS_EINVAL; ctx->ls2_max = (size_t)value; return NBLS_OK;
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

