// pipelines_pairing.cpp -- pairing, Miller products, final exponentiation, prepared G2 points and the multi-GPU partial products (reference index.ts:703-722, math.ts:856-874, 1331-1388).
#include "nbls_internal.h"

int reduce_product(nbls_ctx* ctx, size_t n, uint8_t** result, hipStream_t s) {
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
int expx(nbls_ctx* ctx, size_t n, uint8_t* in, uint8_t* out, hipStream_t s) {
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
int final_exp_pipeline(nbls_ctx* ctx, size_t n, uint8_t* f_raw, void* d_out, hipStream_t s) {
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
int finish_single(nbls_ctx* ctx, uint8_t* f_raw, int final_exp, void* d_out, hipStream_t s) {
  int r;
  if (!final_exp) return run(ctx, P_RAW_TO_BYTES, 1, {B(3, f_raw, F12), B(2, d_out, 576)}, s);
  if ((r = run(ctx, P_NORM_RAW, 1, {B(3, f_raw, F12), B(4, ctx->N, RAW)}, s))) return r;
  return final_exp_pipeline(ctx, 1, f_raw, d_out, s);
}

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
    // size of the first half in per cent (clamped to 1 .. 99): slightly unequal halves do not run phase-locked (profiles/round5_ab_split.txt: 16,384 pairs 6.25 -> 6.14 ms, 65,536 within
    // noise)
    static const size_t split_pct = (size_t)std::min<long>(99, std::max<long>(1, env_long("NBLS_HALVES_SPLIT_PCT", 55)));
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
int pairing_core(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_g2, int with_final_exp, void* d_out, hipStream_t s, bool two_programs) {
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

// Host-buffer staging shared by nbls_pairing_batch and nbls_miller_product: the points go up ONCE (the validity programs read the same device copies the Miller
// loop reads), small calls through the context's page-locked block (one pageable copy costs a staging pass inside the runtime and a synchronisation of its own).
static const size_t PIN_STAGE_MAX = 1u << 20, VALIDATE_BESIDE_MAX = 1024;
static int stage_points(nbls_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, bool validate, void* d_st, hipStream_t s) {
  int r;
  if ((r = ensure_io(ctx, n ? n : 1)) || !n) return r;
  const uint8_t *h1 = g1, *h2 = g2;
  if (n * 288 <= PIN_STAGE_MAX) {
    if ((r = ensure_pinned(ctx, n * 288))) return r;
    memcpy(ctx->pinned, g1, n * 96); memcpy(ctx->pinned + n * 96, g2, n * 192);
    h1 = ctx->pinned; h2 = ctx->pinned + n * 96;
  }
  HIPCHK(hipMemcpyAsync(ctx->io_g1, h1, n * 96, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(ctx->io_g2, h2, n * 192, hipMemcpyHostToDevice, s));
  if (!validate) return NBLS_OK;
  if (n > VALIDATE_BESIDE_MAX) {
    if ((r = dev_validate(ctx, false, n, ctx->io_g1, d_st, s))) return r;
    return dev_validate(ctx, true, n, ctx->io_g2, (uint8_t*)d_st + n, s);
  }
  // A small call leaves most of the device idle, and the validity programs (two 64-bit scalar multiplications and a curve equation each; they touch only the points and
  // the status bytes) do not feed the Miller loop: they run BESIDE it on the context's two side streams; join_validation() makes `s` wait for them before the read-back.
  if ((r = ensure_side(ctx)) || (r = ensure_side2(ctx))) return r;
  HIPCHK(hipEventRecord(ctx->ev_fork, s));
  HIPCHK(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0)); HIPCHK(hipStreamWaitEvent(ctx->side2, ctx->ev_fork, 0));
  if ((r = dev_validate(ctx, true, n, ctx->io_g2, (uint8_t*)d_st + n, ctx->side))) return r;
  HIPCHK(hipEventRecord(ctx->ev_join, ctx->side));
  if ((r = dev_validate(ctx, false, n, ctx->io_g1, d_st, ctx->side2))) return r;
  HIPCHK(hipEventRecord(ctx->ev_join2, ctx->side2));
  return NBLS_OK;
}
static int join_validation(nbls_ctx* ctx, size_t n, bool validate, hipStream_t s) {
  if (!validate || n > VALIDATE_BESIDE_MAX) return NBLS_OK;
  HIPCHK(hipStreamWaitEvent(s, ctx->ev_join, 0)); HIPCHK(hipStreamWaitEvent(s, ctx->ev_join2, 0));
  return NBLS_OK;
}
// results (out_bytes from ctx->io_f12) and, when validating, the 2 n status bytes come back behind ONE synchronisation; codes as PointG1/G2.assertValidity order them
static int fetch_results(nbls_ctx* ctx, size_t n, size_t out_bytes, uint8_t* out, bool validate, const void* d_st, int8_t* st12, hipStream_t s) {
  int r;
  const size_t stb = validate ? 2 * n : 0;
  if (out_bytes + stb <= PIN_STAGE_MAX) {
    if ((r = ensure_pinned_out(ctx, out_bytes + stb))) return r;
    HIPCHK(hipMemcpyAsync(ctx->pinned_out, ctx->io_f12, out_bytes, hipMemcpyDeviceToHost, s));
    if (stb) HIPCHK(hipMemcpyAsync(ctx->pinned_out + out_bytes, d_st, stb, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    memcpy(out, ctx->pinned_out, out_bytes); if (stb) memcpy(st12, ctx->pinned_out + out_bytes, stb);
    return NBLS_OK;
  }
  HIPCHK(hipMemcpyAsync(out, ctx->io_f12, out_bytes, hipMemcpyDeviceToHost, s));
  if (stb) HIPCHK(hipMemcpyAsync(st12, d_st, stb, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return NBLS_OK;
}
static inline int8_t pair_code(const int8_t* st12, size_t n, size_t i) { return st12[i] ? st12[i] : (st12[n + i] ? (int8_t)(10 + st12[n + i]) : 0); }

EXPORT int nbls_pairing_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, int with_final_exp, int validate, uint8_t* out, int8_t* status) {
  if (!ctx || (n && (!g1 || !g2 || !out))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  LOCKED(ctx);   // scratch and I/O staging buffers belong to this call until it returns
  int r;
  HostIO io{ctx}; void* d_st = validate ? io.alloc(2 * n) : nullptr;   // P.assertValidity(); Q.assertValidity()  (index.ts:717-718)
  if (validate && !d_st) return NBLS_EHIP;
  std::vector<int8_t> st12(validate ? 2 * n : 0);
  ForkGuard fork;   // an error return below must not leave the side streams running over buffers the next call reuses
  if ((r = stage_points(ctx, n, g1, g2, validate, d_st, s))) return r;
  if ((r = nbls_pairing_batch_dev(ctx, n, ctx->io_g1, ctx->io_g2, with_final_exp, ctx->io_f12, s)) || (r = join_validation(ctx, n, validate, s))) return r;
  fork.armed = false;
  if ((r = fetch_results(ctx, n, n * 576, out, validate, d_st, st12.data(), s))) return r;
  if (status) memset(status, 0, n);
  if (validate) for (size_t i = 0; i < n; i++) {
    const int8_t c = pair_code(st12.data(), n, i);
    if (c) { memset(out + 576 * i, 0, 576); if (status) status[i] = c; }
  }
  return NBLS_OK;
}

// n >= 1 pairs -> *m_out raw Miller values (products of up to eight Miller loops each) in ctx->F[0 .. *m_out); the caller multiplies them (reduce_product)
int miller_values(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_g2, size_t* m_out, hipStream_t s) {
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
      // round 4: EIGHT pairs per accumulator from acc8_min pairs per call (one squaring per eight line tables: 1,921 instead of 2,196 instructions per pair and bit) -- off since round 6
      // round 6: FEWER pairs per accumulator where a launch is too small to fill the device with four: the accumulation program's instruction stream grows with the pairs per
      // item (four: 1.46 ms for one wavefront), and 4097 pairs in groups of four are 205 wavefronts on 1024 SIMDs.  By the pairs of ONE launch (a call of 8192 pairs and more
      // runs as two halves): one pair per item below acc2_min, two below acc4_min, four above; eight no longer pays at any size (profiles/round6_ab_acc_width.txt: 4098 pairs
      // 2.91 -> 2.13 ms, 8192 4.01 -> 2.77, 16,384 4.54 -> 3.79, 49,152 8.59 -> 7.85, 2^18 32.2 -> 30.9 ms)
      static const size_t acc2_min = (size_t)env_long("NBLS_ACC2_MIN", 6144), acc4_min = (size_t)env_long("NBLS_ACC4_MIN", 28672);
      const size_t chunk = n < LINES_CHUNK ? n : LINES_CHUNK, part = (chunk >= ctx->halves_min && ctx->ioff == 0) ? chunk / 2 : chunk;
      const size_t GR = n >= ctx->acc8_min ? 8 : part >= acc4_min ? 4 : part >= acc2_min ? 2 : 1;
      const ProgId acc = GR == 8 ? P_ACC8_RAW : GR == 4 ? P_ACC4_RAW : GR == 2 ? P_ACC2_RAW : P_ACC_RAW;
      if ((r = ensure_lines(ctx, n + GR - 1))) return r;
      m = 0;
      for (size_t o = 0; o < n; o += LINES_CHUNK) {   // LINES_CHUNK is a multiple of eight: a chunk boundary never splits a group
        const size_t c = n - o < LINES_CHUNK ? n - o : LINES_CHUNK, cg = (c + GR - 1) / GR;
        // a large chunk runs as two halves (whole groups) on two streams, like nbls_pairing_batch_dev: the tail of LINES / ACC of one half under the other
        const size_t h = (c >= ctx->halves_min && ctx->ioff == 0) ? ((c / 2 + GR - 1) & ~(GR - 1)) : c;
        if (h < c && !ctx->half_stream && (hipStreamCreateWithFlags(&ctx->half_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_half_fork,
            hipEventDisableTiming) != hipSuccess ||
                                           hipEventCreateWithFlags(&ctx->ev_half_join, hipEventDisableTiming) != hipSuccess)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
        if (h < c) { HIPCHK(hipEventRecord(ctx->ev_half_fork, s)); HIPCHK(hipStreamWaitEvent(ctx->half_stream, ctx->ev_half_fork, 0)); }
        for (size_t lo = 0; lo < c; lo += h) {
          const size_t cc = lo ? c - lo : h, gg = (cc + GR - 1) / GR;      // two parts: [0, h) and everything behind it
          hipStream_t sh = lo ? ctx->half_stream : s;
          if ((r = run(ctx, P_LINES_PQ, cc, {B(0, (const uint8_t*)d_g1 + (o + lo) * 96, 96), B(1, (const uint8_t*)d_g2 + (o + lo) * 192, 192), B(3, ctx->L + lo * LINE_BYTES, LINE_BYTES)},
              sh))) return r;
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
  if (!ctx || !out || (n && (!g1 || !g2))) return NBLS_EINVAL;
  LOCKED(ctx);   // scratch and I/O staging buffers belong to this call until it returns
  int r;
  const bool val = validate && n;
  HostIO io{ctx}; void* d_st = val ? io.alloc(2 * n) : nullptr;
  if (val && !d_st) return NBLS_EHIP;
  std::vector<int8_t> st12(val ? 2 * n : 0);
  ForkGuard fork;
  if ((r = stage_points(ctx, n, g1, g2, val, d_st, s))) return r;
  // an invalid point fails the whole product (the facade throws before any arithmetic); here the product of a small call is formed beside the checks and dropped if one fails
  if ((r = nbls_miller_product_dev(ctx, n, ctx->io_g1, ctx->io_g2, final_exp, ctx->io_f12, s)) || (r = join_validation(ctx, n, val, s))) return r;
  fork.armed = false;
  if ((r = fetch_results(ctx, n, 576, out, val, d_st, st12.data(), s))) return r;
  if (val) {
    bool bad = false;
    for (size_t i = 0; i < n; i++) { const int8_t c = pair_code(st12.data(), n, i); if (status) status[i] = c; bad = bad || c; }
    if (bad) { memset(out, 0, 576); return NBLS_EDECODE; }
  }
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
int acc_prepared(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_tables, size_t table_stride, hipStream_t s) {
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

// ---- one device's share of a product that is spread over several GPUs, from HOST inputs: the partial stays on this context's device so that
// the caller can move it to the reducing device with hipMemcpyPeer (nbls_multi.cpp) or hand it to a collective.  `*_into`: the partial lands in a caller-owned
// 576-byte buffer on this context's device (what nbls_multi.cpp passes: one buffer per call, so that calls racing on one context cannot see each other's
// partials); the plain names return a buffer owned by the context, valid only until the context's next *_partial call.  The call returns when the partial is complete.
// destination of a partial: the caller's buffer (`*_into`: it must be device memory of at least 576 bytes on the context's device -- checked with
// hipPointerGetAttributes, a wild pointer is refused instead of written through), or -- the original entry points, whose *d_partial is a pure OUT
// parameter again (ABI 2; round 3 had silently made it IN/OUT) -- a buffer owned by the context
int partial_buffer(nbls_ctx* ctx, void* d_dst, uint8_t** dst) {
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
int miller_product_partial_core(nbls_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, int validate, void* d_dst, void** d_partial, int8_t* status) {
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
