// pipelines_codec.cpp -- validity, decoders / encoders, hash-to-curve, point sums, scalar multiplication, MSM and sign (reference index.ts:207-327, 359-448, 481-690, 738-752, 771-788).
#include "nbls_internal.h"

// ================================================================================================================
// Validity, decoders, hash-to-G2, point sums, verifyBatch.  Device-side pipelines (dev_*) work on device pointers and
// enqueue on `s`; the exported wrappers stage host buffers.
// ================================================================================================================
int dev_validate(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, void* d_status, hipStream_t s) {
  return g2 ? run(ctx, P_G2_VALIDATE, n, {B(1, d_pts, 192), B(7, d_status, 1)}, s) : run(ctx, P_G1_VALIDATE, n, {B(0, d_pts, 96), B(7, d_status, 1)}, s);
}
// PointG1.fromHex (48 B) / PointG2.fromSignature (96 B): compressed -> affine wire bytes + status
// slot0 / pow_slot: scratch-pool slots used (three from slot0, one for the exponentiation table), so that two chains can run on
// different streams at the same time
// mode (G2 only): 0 fromSignature 96 B, 1 fromSignature 192 B, 2 fromHex 96 B (no subgroup check, flag rules)
// io / ntot: the call works on items [io, io + n) of scratch arrays sized for ntot items (verify_pipeline: sub-batches of one call run side by side on slices of the same arrays)
int dev_decompress(nbls_ctx* ctx, bool g2, size_t n, const void* d_in, void* d_out, void* d_status, hipStream_t s, int slot0, int pow_slot, int mode, size_t io, size_t ntot) {
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
int dev_clear_g2(nbls_ctx* ctx, size_t n, void* in, uint8_t* base, uint8_t* S, void* out, void* N, hipStream_t s) {
  int r = run(ctx, P_H2C_C0, n, {B(3, in, 6 * RAW), B(6, base, 6 * RAW), B(5, S, 6 * RAW)}, s); if (r) return r;     // v = psi(P) -> base, u = psi^2(2P) - psi(P) - P -> S
  if ((r = run(ctx, pt_ls2_variant(ctx, P_H2C_C1, n), n, {B(3, in, 6 * RAW), B(6, base, 6 * RAW)}, s))) return r;                          // base = t1 + v over v, t1 = -[x]P over P
  // out may be in: every item reads its t1 before its result is stored
  return run(ctx, pt_ls2_variant(ctx, P_H2C_C2, n), n, {B(3, base, 6 * RAW), B(4, in, 6 * RAW), B(5, S, 6 * RAW), B(6, out, 6 * RAW), B(7, N, RAW)}, s);
}
// io / ntot: see dev_decompress; proj: stop at the raw projective points (scratch slot 1) -- the caller multiplies them (sign) and normalises once, at the end
int dev_hash_to_g2(nbls_ctx* ctx, size_t n, const void* d_uniform, void* d_out, hipStream_t s, size_t io, size_t ntot, uint8_t** proj) {
  if (ntot < io + n) ntot = io + n;
  uint8_t *T, *E, *Pw, *Q, *N, *NI, *st, *St, *Pt2, *S, *tab; int r;
  const size_t ST = 32;   // raw elements of SWU state per message: 2 x 12 (P_H2C_A) or 2 x 16 (P_H2C_NA / NM)
  if ((r = need(ctx, 0, ntot * 4 * RAW, &T)) || (r = need(ctx, 1, ntot * 6 * RAW, &E)) || (r = need(ctx, 2, ntot * 4 * RAW, &Pw)) || (r = need(ctx, 3, ntot * 6 * RAW, &Q)) ||
      (r = need(ctx, 4, ntot * RAW, &N)) || (r = need(ctx, 5, ntot * RAW, &NI)) || (r = need(ctx, 6, ntot, &st)) || (r = need(ctx, 18, ntot * ST * RAW, &St)) || (r = need(ctx, 19,
          ntot * 12 * RAW, &Pt2)) ||
      (r = need(ctx, 13, ntot * 6 * RAW, &S)) || (r = need(ctx, 11, ntot * 4 * POW_TAB * RAW, &tab))) return r;
  T += io * 4 * RAW; E += io * 6 * RAW; Pw += io * 4 * RAW; Q += io * 6 * RAW; N += io * RAW; NI += io * RAW; st += io; St += io * ST * RAW; Pt2 += io * 12 * RAW; S += io * 6 * RAW;
  tab += io * 4 * POW_TAB * RAW;
  if (ntot >= ctx->h2c_norm_min) {   // ntot: the messages of the whole call (verifyBatch hashes its sub-batches side by side)
    // The SWU square root by the norm method (codec.h swu_norm_*): two Fp exponentiations with a short program between them where the Fp2 form spends one exponentiation of
    // twice the work.  Same points (the root's sign is fixed by sgn0 afterwards); two more launches, so single messages keep the Fp2 form.
    if ((r = run(ctx, P_H2C_NA, n, {B(0, d_uniform, 256), B(3, T, 4 * RAW), B(4, E, 2 * RAW), B(5, St, 32 * RAW)}, s))) return r;
    if ((r = run_pow(ctx, 0, 2 * n, E, Pw, s, tab))) return r;                                        // n = N(a)^((p+1)/4)
    if ((r = run(ctx, P_H2C_NM, 2 * n, {B(3, T, 2 * RAW), B(4, St, 16 * RAW), B(5, Pw, RAW), B(6, St + 9 * RAW, 16 * RAW), B(7, E, RAW)}, s))) return r;
    if ((r = run_pow(ctx, 3, 2 * n, E, Pw, s, tab))) return r;                                        // e = (delta d^3)^((p-3)/4)
    if ((r = run(ctx, P_H2C_NB, 2 * n, {B(3, T, 2 * RAW), B(4, St, 16 * RAW), B(5, Pw, RAW), B(6, Pt2, 6 * RAW)}, s))) return r;
  } else {
    // H2C_A: per message the two field elements t (T), the exponentiation inputs (E) and the rest of the SWU state (St: twelve raw elements per map)
    if ((r = run(ctx, P_H2C_A, n, {B(0, d_uniform, 256), B(3, T, 4 * RAW), B(4, E, 4 * RAW), B(5, St, 24 * RAW)}, s))) return r;
    if ((r = run_pow(ctx, 2, 2 * n, E, Pw, s, tab))) return r;
    // H2C_B1: one map per item (2 n items) -> its point on E2'
    if ((r = run(ctx, P_H2C_B1, 2 * n, {B(3, T, 2 * RAW), B(5, Pw, 2 * RAW), B(4, St, 12 * RAW), B(6, Pt2, 6 * RAW)}, s))) return r;
  }
  // H2C_B2: the two points of a message -> their sum on E2 (round 3: one program, 62 slots, four workgroups per CU)
  if ((r = run(ctx, P_H2C_B2, n, {B(3, Pt2, 12 * RAW), B(6, E, 6 * RAW)}, s))) return r;        // E is free again: reuse it for the E2 point
  if ((r = dev_clear_g2(ctx, n, E, Q, S, E, N, s))) return r;
  if (proj) { *proj = E; return NBLS_OK; }
  if ((r = run_inv_buf(ctx, n, N, NI, s))) return r;
  return run(ctx, P_G2_TO_AFFINE, n, {B(3, E, 6 * RAW), B(4, NI, RAW), B(2, d_out, 192), B(7, st, 1)}, s);
}
// sum of n affine points (left fold of add == tree of complete additions): affine wire bytes + status (1 = sum is the zero point)
int dev_point_sum(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, void* d_out, void* d_status, hipStream_t s) {
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
int decompress_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* in, uint8_t* out, int8_t* status) {
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
int dev_expand(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offs, const uint8_t* dst, size_t dst_len, uint8_t** d_uniform, hipStream_t s, unsigned len_in_bytes) {
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
int sum_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* pts, uint8_t* out, int8_t* status) {
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
int dev_hash_to_g1(nbls_ctx* ctx, int count, size_t n, const void* d_uniform, void* d_out, hipStream_t s) {
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
int dev_encode_to_g2(nbls_ctx* ctx, size_t n, const void* d_uniform, void* d_out, hipStream_t s) {
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
int hash_curve_host(nbls_ctx* ctx, int mode, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out) {
  if (!ctx || (n && (!offsets || !out || !dst))) return NBLS_EINVAL; if (!n) return NBLS_OK;
  const size_t a = mode == 2 ? 192 : 96;
  LOCKED(ctx); HostIO io{ctx}; void* o = io.alloc(n * a); if (!o) return NBLS_EHIP;
  uint8_t* d; int r = dev_expand(ctx, n, msgs, offsets, dst, dst_len, &d, s, mode == 1 ? 64 : 128); if (r) return r;
  r = mode == 2 ? dev_encode_to_g2(ctx, n, d, o, s) : dev_hash_to_g1(ctx, mode == 0 ? 2 : 1, n, d, o, s); if (r) return r;
  HIPCHK(hipMemcpyAsync(out, o, n * a, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); return NBLS_OK;
}
EXPORT int nbls_hash_to_g1_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out96) { return hash_curve_host(ctx, 0,
    n, msgs, offsets, dst, dst_len, out96); }
EXPORT int nbls_encode_to_g1_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out96) { return hash_curve_host(ctx, 1,
    n, msgs, offsets, dst, dst_len, out96); }
EXPORT int nbls_encode_to_g2_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out192) { return hash_curve_host(ctx,
    2, n, msgs, offsets, dst, dst_len, out192); }

// PointG1.toHex(true) / PointG2.toSignature for non-zero affine points (index.ts:359-371, 586-602): bulk serialisation
int compress_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* aff, uint8_t* out) {
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
int decode_host(nbls_ctx* ctx, int kind /* 0 g1.fromHex, 1 g2.fromHex, 2 g2.fromSignature */, size_t n, const uint8_t* in, size_t len, uint8_t* out, int8_t* status) {
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
int encode_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* aff, const int8_t* zero, int compressed, uint8_t* out) {
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
int clear_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* aff, uint8_t* out, int8_t* status) {
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
// the fixed-base table of G1.BASE (curve.h pt_mul_fixed_g1): for every window w and digit d = 1 .. 2^WIN - 1 the point [d 2^(WIN w)]G as a raw projective point (x, y, 1), computed
// ONCE per context by the variable-base ladder itself (602 scalar multiplications with WIN = 3: a few hundred microseconds) -- no table of constants enters the source
int ensure_g1_fixed(nbls_ctx* ctx, hipStream_t s) {
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
  if (hipMalloc(&dk, m * 32) != hipSuccess || hipMalloc(&aff, m * 96) != hipSuccess || hipMalloc(&st, m) != hipSuccess || hipMalloc(&tab,
      m * 3 * RAW) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); return fail(NBLS_EHIP); }
  if (hipMemcpyAsync(dk, ks.data(), m * 32, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); return fail(NBLS_EHIP); }
  int r = dev_point_mul(ctx, false, m, ctx->gen_g1, 0, dk, aff, st, s, false);
  if (!r) r = run(ctx, P_G1_TO_PROJ, m, {B(0, aff, 96), B(3, tab, 3 * RAW)}, s);
  if (!r && hipStreamSynchronize(s) != hipSuccess) { ctx->last_hip = (int)hipGetLastError(); r = NBLS_EHIP; }
  if (r) return fail(r);
  hipFree(dk); hipFree(aff); hipFree(st);
  ctx->g1_fixed = tab;
  return NBLS_OK;
}
int dev_point_mul(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, size_t pt_stride, const void* d_scalars, void* d_out, void* d_status, hipStream_t s, bool allow_fixed, bool in_subgroup,
                  uint8_t* recoded) {   // recoded: the sign-aligned digits of the scalars, made by the caller (sign_points: beside the hash chain), or NULL
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
    uint8_t* dig = recoded;
    if (!dig && (r = need(ctx, n <= ctx->sac_max ? 2 : 1, n * 128, &dig))) return r;
    // while every wavefront of the launch is resident at once the length of ONE wavefront's instruction stream is the time: the sign-aligned recoding with one addition per bit
    // (codec.h pt_mul_sac_g2: 65 doublings + 73 additions; its table of eight points takes 101 slots = three workgroups per CU = 768 wavefronts of 8 keys); above 6144 keys the
    // windowed form, whose table of four leaves room for six workgroups per CU (NBLS_G2_SAC_MAX / NBLS_TUNE_SAC_MAX; 0 = never)
    if (n <= ctx->sac_max) {
      if (!recoded && nbls_msm_sac_launch((unsigned)n, d_scalars, dig, s)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
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
int sign_points(nbls_ctx* ctx, size_t n, const void* d_uniform, void* h, const void* d_keys32, void* d_out192, void* d_status, hipStream_t s) {
  int r;
  static const bool gls_on = env_long("NBLS_G2_GLS", 1) != 0;
  if (gls_on && n <= ctx->sac_max) {
    // the recoding of the keys (0.045 ms for one key: a branch-free long division) does not need the hash points: it runs beside the hash chain on a side stream (round 6)
    uint8_t *pj, *dig;
    if ((r = need(ctx, 7, n * 128, &dig)) || (r = ensure_side2(ctx))) return r;      // slot 7: used by neither the hash chain nor the ladder
    HIPCHK(hipEventRecord(ctx->ev_fork, s)); HIPCHK(hipStreamWaitEvent(ctx->side2, ctx->ev_fork, 0));
    ForkGuard fork;
    if (nbls_msm_sac_launch((unsigned)n, d_keys32, dig, ctx->side2)) { ctx->last_hip = (int)hipGetLastError(); return NBLS_EHIP; }
    HIPCHK(hipEventRecord(ctx->ev_join2, ctx->side2));
    if ((r = dev_hash_to_g2(ctx, n, d_uniform, nullptr, s, 0, 0, &pj))) return r;
    HIPCHK(hipStreamWaitEvent(s, ctx->ev_join2, 0));
    fork.armed = false;
    return dev_point_mul(ctx, true, n, pj, 6 * RAW, d_keys32, d_out192, d_status, s, true, true, dig);
  }
  if ((r = dev_hash_to_g2(ctx, n, d_uniform, h, s))) return r;
  return dev_point_mul(ctx, true, n, h, 192, d_keys32, d_out192, d_status, s, true, true);      // H(m) is in G2: the ladder may split the key along psi
}
// scalar k is acceptable iff k mod r != 0 (normalizePrivKey, index.ts:269-279, reduces mod r and rejects zero); the ladder
// itself takes any 256-bit value since the points are in the order-r subgroup
bool scalar_is_zero_mod_r(const uint8_t* k32) {
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
int mul_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* pts, const uint8_t* scalars32, uint8_t* out, int8_t* status) {
  if (!ctx || (n && (!scalars32 || !out)) || (g2 && n && !pts)) return NBLS_EINVAL; if (!n) return NBLS_OK;
  const size_t a = g2 ? 192 : 96;
  LOCKED(ctx); HostIO io{ctx}; void *dp = pts ? io.alloc(n * a) : nullptr, *dk = io.alloc(n * 32), *o = io.alloc(n * a), *st = io.alloc(n);
  if ((pts && !dp) || !dk || !o || !st) return NBLS_EHIP;
  io.secret(dk, n * 32);      // the scalars are private keys in getPublicKey / sign: zeroed before the staging block goes back to the pool
  if (pts) HIPCHK(hipMemcpyAsync(dp, pts, n * a, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(dk, scalars32, n * 32, hipMemcpyHostToDevice, s));
  int r = dev_point_mul(ctx, g2, n, pts ? dp : ctx->gen_g1, pts ? a : 0, dk, o, st, s); if (r) return r;
  std::vector<int8_t> tmp(n), zero(n, 0);
  if ((r = io.wipe(s))) return r;
  HIPCHK(hipMemcpyAsync(out, o, n * a, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(tmp.data(), st, n, hipMemcpyDeviceToHost, s));
  for (size_t i = 0; i < n; i++) if (scalar_is_zero_mod_r(scalars32 + 32 * i)) zero[i] = 1;      // host work under the device's (round 6: 0.3 ms at 8192 keys when it followed the synchronisation)
  HIPCHK(hipStreamSynchronize(s));
  for (size_t i = 0; i < n; i++) if (zero[i]) tmp[i] = 5;
  if (status) memcpy(status, tmp.data(), n);
  return NBLS_OK;
}
// PointG1.fromPrivateKey / getPublicKey core (index.ts:350-353, 738-740): [k_i]P_i, P = generator when g1_aff is NULL
EXPORT int nbls_g1_mul_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, const uint8_t* scalars32, uint8_t* out96, int8_t* status) { return mul_host(ctx, false, n, g1_aff, scalars32,
    out96, status); }
EXPORT int nbls_g2_mul_batch(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, const uint8_t* scalars32, uint8_t* out192, int8_t* status) { return mul_host(ctx, true, n, g2_aff, scalars32,
    out192, status); }
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
int dev_msm(nbls_ctx* ctx, bool g2, size_t n, const void* d_pts, const void* d_scalars, unsigned nbits, void* d_out, void* d_status, hipStream_t s) {
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
  if (m) { MSMCHK(nbls_msm_sort_launch(nullptr, &tmp_bytes, nullptr, nullptr, nullptr, nullptr, m, 17, s)); MSMCHK(nbls_msm_rank_launch(nullptr, &scan_bytes, m, nullptr, nullptr, nullptr,
      s)); tmp_bytes = std::max(tmp_bytes, scan_bytes); }
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
  static const bool wide_combine = env_long("NBLS_MSM_WIDE", 1) != 0;
  // Horner over the bit-slices t of every window (sum_t 2^t T_t, one sum per window), then over the windows (below).  G1 (round 6): both with one limb per lane, a wavefront per
  // sum, the four products of a doubling level on its four rows (g1_wide.h: 0.74 -> 0.25 ms of a 65,536-point call); G2 and NBLS_MSM_WIDE=0: the step programs
  if (!g2 && wide_combine) MSMCHK(nbls_g1_wide_combine_launch(src, (int)C, 1, S, nwin, s));
  else if ((r = run(ctx, g2 ? P_G2_HORNER : P_G1_HORNER, nwin, {B(3, src, C * p), B(5, S, p)}, s))) return r;
  // acc <- 2^12 acc + S_w from the top window down: (nwin - 1) x (12 doublings + 1 addition) on ONE point
  if (!g2 && wide_combine) MSMCHK(nbls_g1_wide_combine_launch(S, (int)nwin, (int)C, acc, 1, s));
  else {
    HIPCHK(hipMemcpyAsync(acc, S + (size_t)(nwin - 1) * p, p, hipMemcpyDeviceToDevice, s));
    for (int w = (int)nwin - 2; w >= 0; w--)
      if ((r = run(ctx, g2 ? P_G2_SHIFTADD : P_G1_SHIFTADD, 1, {B(3, acc, p), B(4, S + (size_t)w * p, p), B(5, acc, p)}, s))) return r;
  }
  if ((r = run(ctx, g2 ? P_G2_NORM : P_G1_NORM, 1, {B(3, acc, p), B(4, N, RAW)}, s))) return r;
  if ((r = run_inv_buf(ctx, 1, N, NI, s))) return r;
  return run(ctx, g2 ? P_G2_TO_AFFINE : P_G1_TO_AFFINE, 1, {B(3, acc, p), B(4, NI, RAW), B(2, d_out, a), B(7, d_status, 1)}, s);
}
unsigned scalars_bit_length(size_t n, const uint8_t* k32) {   // max over the batch
  int lead = 32;   // leading zero bytes common to all scalars
  for (size_t i = 0; i < n && lead; i++) { int z = 0; while (z < lead && k32[32 * i + z] == 0) z++; lead = z; }
  if (lead == 32) return 1;
  uint8_t top = 0; for (size_t i = 0; i < n; i++) top |= k32[32 * i + lead];
  unsigned bits = 8 * (31 - lead); while (top) { bits++; top >>= 1; }
  return bits;
}
int msm_host(nbls_ctx* ctx, bool g2, size_t n, const uint8_t* pts, const uint8_t* scalars32, uint8_t* out, int8_t* status) {
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
  // Round 6: ONE pinned staging block in, one out.  Round 5 made six copies from / to pageable memory (messages, offsets, tag, keys; signatures, statuses) with a
  // synchronisation in the middle: 0.5 ms of a 4.4 ms call at 8192 keys (the same call on resident inputs: 3.85 ms).  Inputs are packed into the context's pinned buffer --
  // [message bytes | n + 1 relative offsets | tag (256 B) | keys] -- and travel as one asynchronous copy; signatures and statuses come back the same way.
  for (size_t i = 0; i < n; i++) if (offsets[i + 1] < offsets[i]) return NBLS_EINVAL;
  const size_t total = offsets[n] - offsets[0];
  uint8_t dst_hash[32];
  if (dst_len > 255) { Sha256 c; c.update((const uint8_t*)"H2C-OVERSIZE-DST-", 17); c.update(dst, dst_len); c.final(dst_hash); dst = dst_hash; dst_len = 32; }
  const size_t o_off = (total + 15) & ~(size_t)15, o_dst = o_off + (((n + 1) * 4 + 15) & ~(size_t)15), o_key = o_dst + 256, in_bytes = o_key + n * 32, out_bytes = n * 192 + n;
  LOCKED(ctx); HostIO io{ctx};
  uint8_t *h = (uint8_t*)io.alloc(n * 192), *din = (uint8_t*)io.alloc(in_bytes), *dout = (uint8_t*)io.alloc(out_bytes), *du;
  if (!h || !din || !dout) return NBLS_EHIP;
  io.secret(din + o_key, n * 32);
  int r;
  const size_t pin_out = (in_bytes + 63) & ~(size_t)63;      // the way back has its own part of the pinned block: nothing waits between the ladder and the copy out
  if ((r = need(ctx, 8, n * 256, &du)) || (r = ensure_pinned(ctx, pin_out + out_bytes))) return r;
  uint8_t* pin = ctx->pinned;
  if (total) memcpy(pin, msgs + offsets[0], total);
  { uint32_t* rel = (uint32_t*)(pin + o_off); for (size_t i = 0; i <= n; i++) rel[i] = offsets[i] - offsets[0]; }
  memcpy(pin + o_dst, dst, dst_len); memcpy(pin + o_key, keys32, n * 32);
  HIPCHK(hipMemcpyAsync(din, pin, in_bytes, hipMemcpyHostToDevice, s));
  const int e = nbls_xmd_launch((unsigned)n, din, din + o_off, din + o_dst, (unsigned)dst_len, du, 256, nullptr, s);
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  if ((r = sign_points(ctx, n, du, h, din + o_key, dout, dout + n * 192, s))) return r;
  if ((r = io.wipe(s))) return r;
  HIPCHK(hipMemcpyAsync(pin + pin_out, dout, out_bytes, hipMemcpyDeviceToHost, s));
  std::vector<int8_t> tmp(n, 0);
  for (size_t i = 0; i < n; i++) if (scalar_is_zero_mod_r(keys32 + 32 * i)) tmp[i] = 5;      // (host work under the device's: normalizePrivKey's rejection, index.ts:269-279)
  HIPCHK(hipStreamSynchronize(s));
  memset(pin + o_key, 0, n * 32);        // the host copy of the keys does not outlive the call either
  memcpy(out192, pin + pin_out, n * 192);
  for (size_t i = 0; i < n; i++) if (tmp[i] != 5) tmp[i] = (int8_t)pin[pin_out + n * 192 + i];
  if (status) memcpy(status, tmp.data(), n);
  return NBLS_OK;
}

// The domain-separation tag of the device-resident hash entry points, kept on the device between calls (a service works under one tag): no copy, no synchronisation in the steady
// state.  A DST longer than 255 bytes is replaced by its SHA-256 digest (RFC 9380 5.3.3), as in dev_expand and the reference's expand_message_xmd (index.ts:207-231).
int dst_on_device(nbls_ctx* ctx, const uint8_t* dst, size_t* dst_len, hipStream_t s, uint8_t** dd) {
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
EXPORT int nbls_sign_batch_dev(nbls_ctx* ctx, size_t n, const void* d_msgs, const void* d_offsets, const uint8_t* dst, size_t dst_len, const void* d_keys32, void* d_out192, void* d_status,
    void* stream) {
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

