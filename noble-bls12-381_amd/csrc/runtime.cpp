// runtime.cpp -- device runtime of the pairing engine: program upload, launches (interpreter / ahead-of-time kernels / chains), scratch management, context life cycle.
// No CPU arithmetic path exists here: if HIP or the GPU is unavailable every entry point fails with NBLS_ENOGPU / NBLS_EHIP.
#include "nbls_internal.h"

std::recursive_mutex g_null_mu;   // locked in place of a context's mutex when the caller passed no context (the call then fails with NBLS_EINVAL)
// Checked mode (make debug builds it in with -DNBLS_CHECKED; NBLS_CHECKED=1 switches it on in any build): every program is verified statically
// before its first upload (verify_program: all LDS offsets, descriptor reads and buffer indices the kernel will ever use) and every launch
// checks its buffers against the extents the program touches.  A violation is reported on stderr and the call fails with NBLS_EINVAL.
bool checked_mode() {
#if defined(NBLS_CHECKED)
  return true;
#else
  static const bool on = env_long("NBLS_CHECKED", 0) != 0;
  return on;
#endif
}
// NBLS_AOT=0 keeps every program on the interpreter (A/B runs, profiles of the interpreter)
bool aot_enabled() { static const bool on = env_long("NBLS_AOT", 1) != 0; return on; }
int upload(nbls_ctx* ctx, ProgId id) {
  DevProgram& d = ctx->prog[id];
  if (d.p) return NBLS_OK;
  return upload_program(ctx, d, get_program(id), aot_enabled() ? nbls_aot_index((int)id) : -1);
}
// k: index of the ahead-of-time kernel that serves the program, or -1
void free_program(DevProgram& d) {
  for (void* p : {(void*)d.steps, (void*)d.descs, (void*)d.consts, (void*)d.aot_steps, (void*)d.aot_descs}) if (p) hipFree(p);
  d = DevProgram();
}
int upload_program(nbls_ctx* ctx, DevProgram& d, const Program& p, const int k) {
  if (checked_mode()) { const std::string e = verify_program(p); if (!e.empty()) { fprintf(stderr, "nbls (checked): %s\n", e.c_str()); return NBLS_EINVAL; } }
  // a failure half way leaves nothing behind: d.p stays unset, so a retry uploads again, and would otherwise leak what the first attempt had allocated
  auto fail = [&]() { ctx->last_hip = (int)hipGetLastError(); free_program(d); return NBLS_EHIP; };
  if (hipMalloc(&d.steps, p.steps.size() * sizeof(Step)) != hipSuccess || hipMalloc(&d.descs, p.descs.size() * 4 + 64) != hipSuccess || hipMalloc(&d.consts, p.consts.size() * 4) != hipSuccess ||
      hipMemcpy(d.steps, p.steps.data(), p.steps.size() * sizeof(Step), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d.descs, p.descs.data(), p.descs.size() * 4,
          hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d.consts, p.consts.data(), p.consts.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail();
  // ahead-of-time kernel (aot.h): translate the program; one whose signatures are not all in the kernel's table (build / environment mismatch) stays on the interpreter
  if (k >= 0) {
    AotProgram ap;
    const std::string why = aot_translate(p, ap);
    if (why.empty() && nbls_aot_bind(k, &ap) == 0) {
      if (hipMalloc(&d.aot_steps, ap.steps.size() * sizeof(AotStep)) != hipSuccess || hipMalloc(&d.aot_descs, ap.descs.size() * 4) != hipSuccess ||
          hipMemcpy(d.aot_steps, ap.steps.data(), ap.steps.size() * sizeof(AotStep), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d.aot_descs, ap.descs.data(), ap.descs.size() * 4,
              hipMemcpyHostToDevice) != hipSuccess) return fail();
      d.aot = k; d.aot_lds = ap.lds_bytes;
    } else fprintf(stderr, "nbls: %s: %s; running on the interpreter\n", p.name.c_str(), why.empty() ? "step signatures differ from the ahead-of-time kernel's table" : why.c_str());
  }
  d.wide_ok = p.lsplit == 1 && p.W <= 16 && p.inst_base(0) + p.inst_bytes() <= 64 * 1024;
  for (const Step& st : p.steps) if (!wide_step_supported(st, p.descs.data())) { d.wide_ok = false; break; }
  d.p = &p;
  return NBLS_OK;
}
// does a launch of n items of this (uploaded) program take the one-limb-per-lane form?
bool wide_applies(const nbls_ctx* ctx, const DevProgram& d, int id, size_t n) {
  if (!d.wide_ok || n > ctx->wide_max || id < 0) return false;
  static const long which = env_long("NBLS_WIDE_PROGS", 1);
  if (which) return true;
  return id == P_EXPX || id == P_FE_MID1 || id == P_FE_MID2 || id == P_FE_EASY || id == P_NORM_RAW || id == P_MUL2 || id == P_MUL2S;
}

hipEvent_t timing_event(nbls_ctx* ctx) {
  if (!ctx->ev_pool.empty()) { hipEvent_t e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr; hipEventCreate(&e); return e;
}

void aot_seg(AotSeg& g, const DevProgram& d, const IOBuf* bufs) {
  g.steps = d.aot_steps; g.descs = d.aot_descs; g.consts = d.consts;
  g.nsteps = (u32)d.p->steps.size(); g.nconst = d.p->nconst; g.inst_bytes = d.p->inst_bytes(); g.slot_bytes = d.p->slot_bytes; g.shared_consts = d.p->shared_consts ? 1u : 0u;
  for (int k = 0; k < MAX_BUFS; k++) g.bufs[k] = bufs[k];
}
int run(nbls_ctx* ctx, ProgId id, size_t n, std::initializer_list<std::pair<int, std::pair<const void*, size_t>>> bufs, hipStream_t s, const uint32_t* n_dev, const uint32_t* item_index) {
  int r = upload(ctx, id); if (r) return r;
  return run_dev(ctx, ctx->prog[id], (int)id, n, bufs, s, n_dev, item_index);
}
// id: the timing slot of the launch (a ProgId), or -1 for programs outside the registry (single tower operations)
int run_dev(nbls_ctx* ctx, const DevProgram& d, int id, size_t n, std::initializer_list<std::pair<int, std::pair<const void*, size_t>>> bufs, hipStream_t s, const uint32_t* n_dev,
    const uint32_t* item_index) {
  KernelArgs ka; memset(&ka, 0, sizeof ka);
  ka.steps = d.steps; ka.descs = d.descs; ka.consts = d.consts; ka.qp_table = ctx->qp_table;
  ka.nsteps = (u32)d.p->steps.size(); ka.nconst = d.p->nconst; ka.W = d.p->W; ka.G = d.p->G; ka.slot_bytes = d.p->slot_bytes; ka.inst_bytes = d.p->inst_bytes();
  ka.shared_consts = d.p->shared_consts ? 1u : 0u; ka.lsplit = d.p->lsplit; ka.n_items = (u32)n; ka.n_items_dev = n_dev; ka.item_index = item_index;
  for (auto& b : bufs) { ka.bufs[b.first].ptr = (uint8_t*)b.second.first + ctx->ioff * b.second.second; ka.bufs[b.first].stride = b.second.second; }   // ioff: the second half of a split call
  if (checked_mode()) {
    for (int k = 0; k < MAX_BUFS; k++) {
      const u32 ext = d.p->buf_extent[k];
      if (!ext) continue;
      if (!ka.bufs[k].ptr || (ka.bufs[k].stride != 0 && ka.bufs[k].stride < ext)) {
        fprintf(stderr, "nbls (checked): %s: buffer %d: %s (stride %llu, the program touches %u bytes per item)\n", d.p->name.c_str(), k, ka.bufs[k].ptr ? "stride too small" : "not bound",
            (unsigned long long)ka.bufs[k].stride, ext);
        return NBLS_EINVAL;
      }
      // the last item's bytes must lie inside the allocation the pointer belongs to (items reached through an index list are not bounded by n)
      hipDeviceptr_t base = nullptr; size_t size = 0;
      if (n && !item_index && hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ka.bufs[k].ptr) == hipSuccess) {
        const size_t end = (size_t)((uint8_t*)ka.bufs[k].ptr - (uint8_t*)base) + (n - 1) * ka.bufs[k].stride + ext;
        if (end > size) {
          fprintf(stderr, "nbls (checked): %s: buffer %d: %zu items of stride %llu (+%u) end %zu bytes into an allocation of %zu\n", d.p->name.c_str(), k, n,
              (unsigned long long)ka.bufs[k].stride, ext, end, size);
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
// Launches of at most ctx->inv_wide_max elements run the inversion with one limb per lane, four elements per wavefront (fp_inv_wide.h: the same binary GCD, ~25 k instead of ~48 k
// wave-instructions on the critical path of every single call); above, one element per lane.  NBLS_INV_WIDE_MAX / NBLS_TUNE_INV_WIDE_MAX (0 = never).
int run_inv(nbls_ctx* ctx, size_t n, hipStream_t s) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->timing) { e0 = timing_event(ctx); e1 = timing_event(ctx); hipEventRecord(e0, s); }
  const uint8_t* in = ctx->N + ctx->ioff * RAW; uint8_t* out = ctx->NI + ctx->ioff * RAW;
  int e = n <= ctx->inv_wide_max ? nbls_fp_inv_wide_launch((unsigned)n, in, out, s) : nbls_fp_inv_launch((unsigned)n, in, out, s);
  if (ctx->timing) { hipEventRecord(e1, s); ctx->tev.push_back({(int)P_COUNT, {e0, e1}}); }
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  return NBLS_OK;
}

// ctx->chain_max: items up to which the middle of the final exponentiation runs as one chain (default 8192, NBLS_CHAIN_MAX / NBLS_TUNE_CHAIN_MAX): measured equal to seven
// launches up to 4096 pairings per call (2.371 against 2.374 ms), slower where a call runs as two halves on two streams (16,384: 6.53 against 6.28 ms), whose launches fill each
// other's tails -- and slower with calls in flight on other streams for the same reason: a chained wavefront is 427 k instructions long, so the rounds of wavefronts at the end of a
// burst are coarse (twenty 4096-pairing calls on twenty streams 2.82 against 2.85 M pairings/s, 512 calls twelve deep 3.05 against 3.08 M; tools/ab_chain20.sh).  The pool
// (nbls_pool_init, nbls_multi.cpp) therefore sets it to 0 for its contexts.
bool chains_enabled() { static const bool on = env_long("NBLS_CHAIN", 1) != 0; return on; }
int run_chain(nbls_ctx* ctx, size_t n, std::initializer_list<ChainLink> links, hipStream_t s) {
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

int ensure_scratch(nbls_ctx* ctx, size_t n) {
  if (n <= ctx->cap_F) return NBLS_OK;
  size_t cap = n + n / 8 + 64;
  if (ctx->F) {
    hipFree(ctx->F); hipFree(ctx->N); hipFree(ctx->NI); for (auto& t : ctx->T) { hipFree(t); t = nullptr; } hipFree(ctx->KS); hipFree(ctx->KD); hipFree(ctx->Kflag);
    hipFree(ctx->Klist); hipFree(ctx->Kcount);
  }
  ctx->F = ctx->N = ctx->NI = ctx->KS = ctx->KD = ctx->Kflag = nullptr; ctx->Klist = ctx->Kcount = nullptr; ctx->cap_F = 0;
  HIPCHK(hipMalloc(&ctx->F, (cap + 2) * F12));
  HIPCHK(hipMalloc(&ctx->N, cap * RAW));
  HIPCHK(hipMalloc(&ctx->NI, cap * RAW));
  for (auto& t : ctx->T) HIPCHK(hipMalloc(&t, cap * F12));
  ctx->cap_F = cap;
  return NBLS_OK;
}
// scratch of the compressed-squaring exponentiation (2.8 KB per item): only a context that runs that path -- off by default, NBLS_TUNE_EXPC_MIN -- ever allocates it
int ensure_expc_scratch(nbls_ctx* ctx) {
  if (ctx->Kcount) return NBLS_OK;     // the LAST allocation below: a set that failed half way is not taken for complete (sized with F / T; ensure_scratch frees it when they grow)
  const size_t cap = ctx->cap_F;
  auto fail = [&]() { for (void* p : {(void*)ctx->KS, (void*)ctx->KD, (void*)ctx->Kflag, (void*)ctx->Klist,
      (void*)ctx->Kcount}) if (p) hipFree(p); ctx->KS = ctx->KD = ctx->Kflag = nullptr; ctx->Klist = ctx->Kcount = nullptr; ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; };
  if (ctx->KS) { hipFree(ctx->KS); ctx->KS = nullptr; } if (ctx->KD) { hipFree(ctx->KD); ctx->KD = nullptr; } if (ctx->Kflag) { hipFree(ctx->Kflag); ctx->Kflag = nullptr; }
  if (ctx->Klist) { hipFree(ctx->Klist); ctx->Klist = nullptr; }
  if (hipMalloc(&ctx->KS, cap * EXPC_SQ_ELEMS * RAW) != hipSuccess || hipMalloc(&ctx->KD, cap * EXPC_DEC_ELEMS * RAW) != hipSuccess || hipMalloc(&ctx->Kflag, cap) != hipSuccess ||
      hipMalloc(&ctx->Klist, cap * 4) != hipSuccess || hipMalloc(&ctx->Kcount, 8) != hipSuccess) return fail();
  return NBLS_OK;
}
int ensure_io(nbls_ctx* ctx, size_t n) {
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

int ensure_lines(nbls_ctx* ctx, size_t n) {
  if (n > LINES_CHUNK + 3) n = LINES_CHUNK + 3;
  if (n <= ctx->cap_L) return NBLS_OK;
  size_t cap = n + n / 8 + 8; if (cap > LINES_CHUNK + 3) cap = LINES_CHUNK + 3;
  if (ctx->L) hipFree(ctx->L);
  ctx->L = nullptr; ctx->cap_L = 0;
  HIPCHK(hipMalloc(&ctx->L, cap * LINE_BYTES));
  ctx->cap_L = cap;
  return NBLS_OK;
}
// The side streams (verifyBatch's one-element chains and key decoding; the validity programs of a small validated pairing call) are created on first use: HIP spreads
// streams over a few hardware queues in creation order, and contexts that only run pairing batches (pipeline.py keeps several in flight) should each get a queue of their own.
static int ensure_fork_event(nbls_ctx* ctx) {
  if (!ctx->ev_fork && hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
  return NBLS_OK;
}
int ensure_side(nbls_ctx* ctx) {     // ctx->side, ev_fork, ev_join and the side stream's own scratch (PointG2.fromSignature of ONE signature)
  int r = ensure_fork_event(ctx); if (r) return r;
  if (!ctx->side && hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess) { ctx->side = nullptr; ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
  if (!ctx->ev_join && hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
  if (!ctx->side_scratch && hipMalloc(&ctx->side_scratch, (6 + 2 * POW_TAB) * RAW) != hipSuccess) { ctx->side_scratch = nullptr; ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
  return NBLS_OK;
}
int ensure_side2(nbls_ctx* ctx) {    // ctx->side2, ev_fork, ev_join2
  int r = ensure_fork_event(ctx); if (r) return r;
  if (!ctx->side2 && hipStreamCreateWithFlags(&ctx->side2, hipStreamNonBlocking) != hipSuccess) { ctx->side2 = nullptr; ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
  if (!ctx->ev_join2 && hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
  return NBLS_OK;
}
int ensure_pinned(nbls_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pinned_cap) return NBLS_OK;
  if (ctx->pinned) { hipHostFree(ctx->pinned); ctx->pinned = nullptr; ctx->pinned_cap = 0; }
  const size_t cap = bytes + bytes / 4 + 4096;
  HIPCHK(hipHostMalloc((void**)&ctx->pinned, cap, hipHostMallocDefault));
  ctx->pinned_cap = cap;
  return NBLS_OK;
}
int ensure_pinned_out(nbls_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pinned_out_cap) return NBLS_OK;
  if (ctx->pinned_out) { hipHostFree(ctx->pinned_out); ctx->pinned_out = nullptr; ctx->pinned_out_cap = 0; }
  const size_t cap = bytes + bytes / 4 + 4096;
  HIPCHK(hipHostMalloc((void**)&ctx->pinned_out, cap, hipHostMallocDefault));
  ctx->pinned_out_cap = cap;
  return NBLS_OK;
}
int need(nbls_ctx* ctx, int i, size_t bytes, uint8_t** out) {
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
// table): one verify / sign spends 0.3 instead of 0.7 ms in the Fp2 exponentiation of hash-to-G2.  NBLS_POW_WIDE_MAX (0 = never); 3072 since the end of round 6 (1024 before:
// hash-to-G2 of 1024 messages 1.65 -> 1.40 ms, 2048 compressed signatures 1.10 -> 0.86 ms; from 4096 elements the one-lane kernels win, profiles/round6_ab_pow_wide_max.txt).
size_t pow_wide_max() { static const size_t v = (size_t)env_long("NBLS_POW_WIDE_MAX", 3072); return v; }
int run_pow(nbls_ctx* ctx, int which, size_t n, const void* in, void* out, hipStream_t s, uint8_t* scratch) {
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
int run_inv_buf(nbls_ctx* ctx, size_t n, const void* in, void* out, hipStream_t s) {
  int e = n <= ctx->inv_wide_max ? nbls_fp_inv_wide_launch((unsigned)n, in, out, s) : nbls_fp_inv_launch((unsigned)n, in, out, s);
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  return NBLS_OK;
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
  if (hipMalloc(&ctx->qp_table, (size_t)QP_TABLE_ENTRIES * RAW_WORDS * 4) != hipSuccess || hipMemcpy(ctx->qp_table, qp_table_words(), (size_t)QP_TABLE_ENTRIES * RAW_WORDS * 4,
      hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return NBLS_EHIP; }
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
    auto be = [](uint8_t* o, const u32* limbs) { u32 w[12]; limbs_to_words(w,
        limbs); for (int i = 0; i < 12; i++) { u32 v = w[11 - i]; o[4 * i] = v >> 24; o[4 * i + 1] = v >> 16; o[4 * i + 2] = v >> 8; o[4 * i + 3] = v; } };
    be(ng, NBLS_G1X_RAW); be(ng + 48, NBLS_NEG_G1Y_RAW);
    uint8_t gg[96]; be(gg, NBLS_G1X_RAW); be(gg + 48, NBLS_G1Y_RAW);
    if (hipMalloc(&ctx->gen_g1, 96) != hipSuccess || hipMemcpy(ctx->gen_g1, gg, 96, hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return NBLS_EHIP; }
    u32 id1[3 * SLOT_WORDS] = {0}, id2[6 * SLOT_WORDS] = {0}; memcpy(id1 + SLOT_WORDS, NBLS_R1, NLIMBS * 4); memcpy(id2 + 2 * SLOT_WORDS, NBLS_R1, NLIMBS * 4);   // (0 : 1 : 0)
    if (hipMalloc(&ctx->neg_g1, 96) != hipSuccess || hipMemcpy(ctx->neg_g1, ng, 96, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&ctx->ident_g1, 3 * RAW) != hipSuccess || hipMemcpy(ctx->ident_g1, id1, 3 * RAW, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&ctx->ident_g2, 6 * RAW) != hipSuccess || hipMemcpy(ctx->ident_g2, id2, 6 * RAW, hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return NBLS_EHIP; }
  }
  // the scalar-multiplication ladders are traced and uploaded on first use; so are (round 6) the programs of experiments and unit tests that no default path runs: the
  // compressed-squaring exponentiation (off unless NBLS_TUNE_EXPC_MIN asks for it), round 3's one-program forms of hash-to-G2 and the test-only pieces of it
  for (int i = 0; i < P_COUNT; i++) { if (i == P_G1_MUL || i == P_G2_MUL || i == P_G1_MUL_W3 || i == P_G2_MUL_W3 || i == P_G1_MUL_FIXED || i == P_G2_MUL_GLS || i == P_G2_MUL_SAC
      || i == P_G2_MUL_SAC_LS2 || i == P_EXPC_SQ || i == P_EXPC_DEC_A || i == P_EXPC_DEC_B || i == P_H2C_B || i == P_H2C_C || i == P_T_SWU || i == P_T_ISO || i == P_T_CLEAR) continue;
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
  for (uint8_t* p : {ctx->F, ctx->N, ctx->NI, ctx->io_g1, ctx->io_g2, ctx->io_f12, ctx->one12, ctx->gen_g1, ctx->g1_fixed, ctx->side_scratch, ctx->L, ctx->partial, ctx->unit_lines, ctx->KS,
      ctx->KD, ctx->Kflag, (uint8_t*)ctx->Klist, (uint8_t*)ctx->Kcount}) if (p) hipFree(p);
  for (uint8_t* p : ctx->T) if (p) hipFree(p);
  for (uint8_t* p : ctx->sb) if (p) hipFree(p);
  for (auto& b : ctx->io_pool) if (b.p) hipFree(b.p);
  for (uint8_t* p : ctx->nib) if (p) hipFree(p);
  if (ctx->pinned) hipHostFree(ctx->pinned);
  if (ctx->pinned_out) hipHostFree(ctx->pinned_out);
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

EXPORT int nbls_abi_version(void) { return NBLS_ABI_VERSION; }
// every environment switch the library has read so far, with the value in force (config.h); the string lives until the next call on this thread
EXPORT const char* nbls_config_describe(void) { static thread_local std::string s; s = env_describe(); return s.c_str(); }
EXPORT int nbls_context_device(nbls_ctx* ctx) { return ctx ? ctx->device : -1; }

