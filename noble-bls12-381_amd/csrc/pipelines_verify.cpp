// pipelines_verify.cpp -- verify / verifyBatch as concurrent sub-batches and their multi-GPU shares (reference index.ts:756-767, 792-821).
#include "nbls_internal.h"

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
std::vector<size_t> verify_plan(nbls_ctx* ctx, size_t n) {
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
int pipe_event(nbls_ctx* ctx, size_t i, hipEvent_t* e) {
  while (ctx->pipe_ev.size() <= i) { hipEvent_t ev = nullptr; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); ctx->pipe_ev.push_back(ev); }
  *e = ctx->pipe_ev[i];
  return NBLS_OK;
}
int ensure_half_stream(nbls_ctx* ctx) {
  if (!ctx->half_stream && (hipStreamCreateWithFlags(&ctx->half_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_half_fork, hipEventDisableTiming) != hipSuccess ||
                            hipEventCreateWithFlags(&ctx->ev_half_join, hipEventDisableTiming) != hipSuccess)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
  return NBLS_OK;
}
// final_exp = 1: the product's final exponentiation as 576 wire bytes in `out` (host); 0: the product itself as wire bytes at d_out (device; a shard's partial).
// st: n statuses of the keys (+ 1 of the signature) as the decoders wrote them; *bad_offsets: the message offsets were not monotonic.
int pipe_stream(nbls_ctx* ctx, size_t i, hipStream_t* st) {
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
int verify_pipeline(nbls_ctx* ctx, size_t n, const VerifyIn& in, int final_exp, void* d_out, uint8_t* out, std::vector<int8_t>& st, int* bad_offsets, void* stream) {
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
    if ((r = ensure_side(ctx))) return r;
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
    if ((r = ensure_side2(ctx))) return r;
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
      // issue slots free, and in sequence they put 3 ms in front of the longest chain of the call -- measured no better either (profiles/round5_ab_verify2.txt), off by default:
      // NBLS_VERIFY_KEYS_SIDE=1
      static const bool keys_side = env_long("NBLS_VERIFY_KEYS_SIDE", 0) != 0;
      if (c == 0 && keys_side) {
        if ((r = ensure_side2(ctx))) return r;
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
      // (round 6: four from 32,768 pairs instead of 8192 -- verifyBatch(32,768) 13.2 -> 12.0 ms, (65,536) unchanged; profiles/round6_ab_acc_width.txt)
      static const size_t v_acc4_min = (size_t)env_long("NBLS_VERIFY_ACC4_MIN", 32768), v_acc2_min = (size_t)env_long("NBLS_VERIFY_ACC2_MIN", 2048);
      const size_t GR = cc >= v_acc4_min ? 4 : cc >= v_acc2_min ? 2 : 1;
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
  // the result and the statuses lie behind one another (O | ST | bad-offsets word): ONE copy into page-locked memory (round 6: two copies into pageable memory before)
  const size_t st_bytes = ((np + 3) & ~(size_t)3) + 4, back = 576 + st_bytes;
  if ((r = ensure_pinned_out(ctx, back))) return r;
  HIPCHK(hipMemcpyAsync(ctx->pinned_out, O, back, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  fork_guard.armed = false;      // synchronised: every forked stream was joined into s
  if (final_exp) memcpy(out, ctx->pinned_out, 576);
  memcpy(st.data(), ctx->pinned_out + 576, st_bytes);
  uint32_t bad = 0; memcpy(&bad, st.data() + ((np + 3) & ~(size_t)3), 4);
  if (bad_offsets) *bad_offsets = bad != 0;
  st.resize(np);
  return NBLS_OK;
}
bool verify_pipe_enabled() { static const bool on = env_long("NBLS_VERIFY_PIPE", 1) != 0; return on; }
bool fp12_wire_is_one(const uint8_t* out) { bool one = out[47] == 1; for (int i = 0; i < 576 && one; i++) if (i != 47 && out[i]) one = false; return one; }   // exp.equals(Fp12.ONE)
// the whole of verifyBatch behind the pipeline: decide from the statuses as the reference does (index.ts:792-821)
int verify_decide(const std::vector<int8_t>& st, size_t n, const uint8_t* out, int* ok, int8_t* pk_status) {
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
    // Round 6: messages, offsets, tag, keys and the signature travel as ONE copy from a page-locked block (five copies from pageable memory and two synchronisations before
    // the chain even started: ~0.2 ms of a 2.8 ms verify); nothing waits on the host until the pipeline's single synchronisation at the end (the block belongs to the context,
    // whose mutex this call holds).
    for (size_t i = 0; i < n; i++) if (offsets[i + 1] < offsets[i]) return NBLS_EINVAL;
    const size_t total = offsets[n] - offsets[0];
    uint8_t dst_hash[32];
    if (dst_len > 255) { Sha256 c; c.update((const uint8_t*)"H2C-OVERSIZE-DST-", 17); c.update(dst, dst_len); c.final(dst_hash); dst = dst_hash; dst_len = 32; }
    const size_t o_off = (total + 15) & ~(size_t)15, o_dst = o_off + (((n + 1) * 4 + 15) & ~(size_t)15), o_pk = o_dst + 256, o_sig = o_pk + ((n * 48 + 15) & ~(size_t)15), in_bytes = o_sig + 96;
    LOCKED(ctx);
    uint8_t *c, *du; int r;
    if ((r = need(ctx, 9, in_bytes, &c)) || (r = need(ctx, 8, n * 256, &du)) || (r = ensure_pinned(ctx, in_bytes))) return r;
    uint8_t* pin = ctx->pinned;
    if (total) memcpy(pin, msgs + offsets[0], total);
    { uint32_t* rel = (uint32_t*)(pin + o_off); for (size_t i = 0; i <= n; i++) rel[i] = offsets[i] - offsets[0]; }
    memcpy(pin + o_dst, dst, dst_len); memcpy(pin + o_pk, pk48, n * 48); memcpy(pin + o_sig, sig96, 96);
    HIPCHK(hipMemcpyAsync(c, pin, in_bytes, hipMemcpyHostToDevice, s));
    const int e = nbls_xmd_launch((unsigned)n, c, c + o_off, c + o_dst, (unsigned)dst_len, du, 256, nullptr, s);
    if (e) { ctx->last_hip = e; return NBLS_EHIP; }
    d_sig = c + o_sig; d_uni = du; d_pk = c + o_pk;
  }
  return nbls_verify_batch_dev_inputs(ctx, n, d_sig, d_uni, d_pk, ok, nullptr, nullptr);
}
// verifyBatch with EVERYTHING resident in HBM (bench.py's verifyBatch value): signature, the message bytes with their n + 1 offsets (uint32, relative to d_msgs),
// compressed keys.  SHA-256 expand_message_xmd (index.ts:207-231) runs first, on the same stream, then the call continues as nbls_verify_batch_dev_inputs.
EXPORT int nbls_verify_batch_msgs_dev(nbls_ctx* ctx, size_t n, const void* d_sig96, const void* d_msgs, const void* d_offsets, const void* d_pk48, const uint8_t* dst, size_t dst_len,
    int* ok, void* stream) {
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
int verify_stage(nbls_ctx* ctx, size_t n, const void* d_sig96, const void* d_uniform, const void* d_pk48, std::vector<int8_t>& st, void* stream) {
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
    if ((r = ensure_side(ctx))) return r;
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
    if ((r = ensure_side2(ctx))) return r;
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
    // the pipeline decides nothing before its end: with an undecodable key or a zero point d_out_fp12 holds a meaningless product (round 4 left it unwritten); callers look at the return
    // code and the flag first
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

int verify_batch_partial_core(nbls_ctx* ctx, size_t n, const uint8_t* sig96 /* or NULL */, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* pk48,
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
