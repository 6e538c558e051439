/*
 * nbls_napi.c -- thin N-API addon: exposes the C ABI of libnbls.so (include/nbls.h) to Node.  No arithmetic here.
 * libnbls.so is loaded with dlopen at module init so the addon builds with plain gcc (no HIP needed):
 *     gcc -shared -fPIC -I/usr/include/node -I../../include nbls_napi.c -o nbls_napi.node -ldl
 * Calls are synchronous (they block for the duration of the GPU work) except verifyBatchAsync, which runs on a libuv worker thread
 * (napi_create_async_work) and resolves a Promise: the facade's verifyBatch uses it for wire-format inputs (the calls that can take tens of milliseconds).  Typed arrays are passed by reference (napi_get_typedarray_info), no copies.
 */
#include <node_api.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nbls.h"

static void* lib;
#define SYM(name) static __typeof__(&name) p_##name;
SYM(nbls_init) SYM(nbls_destroy) SYM(nbls_strerror) SYM(nbls_pairing_batch) SYM(nbls_miller_product) SYM(nbls_final_exp_batch)
SYM(nbls_g1_validate_batch) SYM(nbls_g2_validate_batch) SYM(nbls_g1_decompress_batch) SYM(nbls_g2_decompress_batch)
SYM(nbls_hash_to_g2_batch) SYM(nbls_g1_sum) SYM(nbls_g2_sum) SYM(nbls_verify_batch) SYM(nbls_g1_mul_batch) SYM(nbls_g2_mul_batch) SYM(nbls_sign_batch) SYM(nbls_hash_to_g1_batch) SYM(nbls_encode_to_g1_batch) SYM(nbls_encode_to_g2_batch) SYM(nbls_g1_msm) SYM(nbls_g2_msm)
SYM(nbls_init_multi) SYM(nbls_destroy_multi) SYM(nbls_multi_device_count) SYM(nbls_multi_context) SYM(nbls_multi_pairing_batch) SYM(nbls_multi_miller_product) SYM(nbls_multi_verify_batch)
SYM(nbls_g2_prepare) SYM(nbls_pairing_prepared)
SYM(nbls_g1_from_hex_batch) SYM(nbls_g2_from_hex_batch) SYM(nbls_g2_from_signature_batch) SYM(nbls_g1_clear_cofactor_batch) SYM(nbls_g2_clear_cofactor_batch)
static nbls_ctx* ctx;
static nbls_multi* multi;   /* several GPUs behind one handle (initMulti): ctx is then its first context; the batch calls shard over all of them */
#define MULTI() (multi && p_nbls_multi_device_count(multi) > 1)

#define CHECK(env, call) do { if ((call) != napi_ok) { napi_throw_error(env, NULL, "N-API call failed: " #call); return NULL; } } while (0)
static napi_value throw_code(napi_env env, int code) { char m[128]; snprintf(m, sizeof m, "nbls: %s (code %d)", p_nbls_strerror ? p_nbls_strerror(code) : "error", code); napi_throw_error(env, NULL, m); return NULL; }

static int get_bytes(napi_env env, napi_value v, uint8_t** data, size_t* len) {
  bool is_ta = false; napi_is_typedarray(env, v, &is_ta);
  if (is_ta) { napi_typedarray_type t; napi_value ab; size_t off; void* d; if (napi_get_typedarray_info(env, v, &t, len, &d, &ab, &off) != napi_ok) return 0; if (t == napi_uint32_array || t == napi_int32_array) *len *= 4;   /* length in bytes */ *data = (uint8_t*)d; return 1; }
  bool is_buf = false; napi_is_buffer(env, v, &is_buf);
  if (is_buf) { void* d; if (napi_get_buffer_info(env, v, &d, len) != napi_ok) return 0; *data = (uint8_t*)d; return 1; }
  return 0;
}
static napi_value new_u8(napi_env env, size_t n, uint8_t** data) {
  napi_value ab, ta; void* d; if (napi_create_arraybuffer(env, n, &d, &ab) != napi_ok) return NULL; *data = (uint8_t*)d;
  if (napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &ta) != napi_ok) return NULL; return ta;
}
static napi_value result2(napi_env env, napi_value out, napi_value status) {
  napi_value o; napi_create_object(env, &o); napi_set_named_property(env, o, "out", out); napi_set_named_property(env, o, "status", status); return o;
}
#define ARGS(n) size_t argc = n; napi_value argv[n]; CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL)); if (argc < n) { napi_throw_type_error(env, NULL, "missing arguments"); return NULL; }
#define NEED_CTX() if (!ctx) { napi_throw_error(env, NULL, "nbls: call init(deviceId) first"); return NULL; }
#define COUNT_FROM_OFFSETS(n, lo) if ((lo) < 4 || (lo) % 4) { napi_throw_range_error(env, NULL, "offsets must hold n + 1 entries"); return NULL; } size_t n = (lo) / 4 - 1;
#define ALLOCATED(v) if (!(v)) { napi_throw_error(env, NULL, "nbls: out of memory"); return NULL; }
#define BYTES(i, d, l) uint8_t* d; size_t l; if (!get_bytes(env, argv[i], &d, &l)) { napi_throw_type_error(env, NULL, "expected Uint8Array"); return NULL; }

/* Contexts for asynchronous calls in flight: verifyBatchAsync takes them round-robin, so that `await Promise.all([verifyBatch(..), verifyBatch(..), ..])`
 * overlaps the calls on the GPU (three 65,536-signature calls in flight: 21 ms per call amortised against 26 ms one at a time, bench.py).  pool[0] = ctx;
 * NBLS_CONTEXTS (default 1) or init(device, contexts) sets the size; the extra contexts are created on first use. */
#define MAX_POOL 8
static nbls_ctx* pool[MAX_POOL]; static int pool_size = 1, pool_dev = 0; static unsigned pool_next = 0;
static void pool_clear(void) { for (int i = 1; i < MAX_POOL; i++) if (pool[i]) { p_nbls_destroy(pool[i]); pool[i] = NULL; } pool[0] = NULL; pool_next = 0; }
static nbls_ctx* pool_take(void) {
  if (pool_size <= 1 || !ctx) return ctx;
  const int k = (int)(pool_next++ % (unsigned)pool_size);
  if (k == 0) return ctx;
  if (!pool[k] && p_nbls_init(pool_dev, &pool[k]) != 0) { pool[k] = NULL; return ctx; }
  return pool[k];
}
static napi_value Init(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2]; CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1) { napi_throw_type_error(env, NULL, "missing arguments"); return NULL; }
  int32_t dev = 0, nctx = 0; napi_get_value_int32(env, argv[0], &dev);
  napi_valuetype t1 = napi_undefined; if (argc > 1) napi_typeof(env, argv[1], &t1);
  if (t1 == napi_number) napi_get_value_int32(env, argv[1], &nctx);
  else { const char* e = getenv("NBLS_CONTEXTS"); nctx = e ? atoi(e) : 1; }
  if (multi) { p_nbls_destroy_multi(multi); multi = NULL; ctx = NULL; }
  pool_clear();
  if (ctx) { p_nbls_destroy(ctx); ctx = NULL; }
  int r = p_nbls_init(dev, &ctx); if (r) return throw_code(env, r);
  pool[0] = ctx; pool_dev = dev; pool_size = nctx < 1 ? 1 : (nctx > MAX_POOL ? MAX_POOL : nctx);
  napi_value t; napi_get_boolean(env, true, &t); return t;
}
/* initMulti(Int32Array of device ids | null for every visible device) -> number of devices.  pairingBatch, millerProduct and verifyBatch(Async)
 * then shard over all of them (nbls_multi_*, include/nbls.h); every other call runs on the first device. */
static napi_value InitMulti(napi_env env, napi_callback_info info) {
  ARGS(1); napi_valuetype t; napi_typeof(env, argv[0], &t);
  int ids[64]; int n = 0;
  if (t != napi_null && t != napi_undefined) {
    uint8_t* d; size_t l; if (!get_bytes(env, argv[0], &d, &l) || l % 4 || l / 4 > 64) { napi_throw_type_error(env, NULL, "expected Int32Array of at most 64 device ids or null"); return NULL; }
    n = (int)(l / 4); memcpy(ids, d, l);
  }
  if (multi) { p_nbls_destroy_multi(multi); multi = NULL; ctx = NULL; }
  pool_clear(); pool_size = 1;   /* asynchronous calls run on the multi-device handle */
  if (ctx) { p_nbls_destroy(ctx); ctx = NULL; }
  int r = p_nbls_init_multi(n, n ? ids : NULL, &multi); if (r) return throw_code(env, r);
  ctx = p_nbls_multi_context(multi, 0);
  napi_value v; napi_create_int32(env, p_nbls_multi_device_count(multi), &v); return v;
}
/* decodePoints(kind, bytes, len) -> {out, status}: kind 0 PointG1.fromHex, 1 PointG2.fromHex, 2 PointG2.fromSignature; len = bytes per encoded point */
static napi_value DecodePoints(napi_env env, napi_callback_info info) {
  ARGS(3); NEED_CTX(); int32_t kind = 0, len = 0; napi_get_value_int32(env, argv[0], &kind); BYTES(1, in, l); napi_get_value_int32(env, argv[2], &len);
  if (len <= 0 || l % (size_t)len) { napi_throw_range_error(env, NULL, "bad encoded point length"); return NULL; }
  size_t n = l / (size_t)len, a = kind == 0 ? 96 : 192;
  uint8_t *out, *st; napi_value vo = new_u8(env, n * a, &out), vs = new_u8(env, n, &st); ALLOCATED(vo); ALLOCATED(vs);
  int r = kind == 0 ? p_nbls_g1_from_hex_batch(ctx, n, in, (size_t)len, out, (int8_t*)st) : kind == 1 ? p_nbls_g2_from_hex_batch(ctx, n, in, (size_t)len, out, (int8_t*)st)
                    : p_nbls_g2_from_signature_batch(ctx, n, in, (size_t)len, out, (int8_t*)st);
  if (r) return throw_code(env, r); return result2(env, vo, vs);
}
/* clearCofactor(g2, affine points) -> {out, status} */
static napi_value ClearCofactor(napi_env env, napi_callback_info info) {
  ARGS(2); NEED_CTX(); bool g2; napi_get_value_bool(env, argv[0], &g2); BYTES(1, in, l); size_t a = g2 ? 192 : 96, n = l / a;
  if (l != n * a) { napi_throw_range_error(env, NULL, "bad point array length"); return NULL; }
  uint8_t *out, *st; napi_value vo = new_u8(env, n * a, &out), vs = new_u8(env, n, &st); ALLOCATED(vo); ALLOCATED(vs);
  int r = g2 ? p_nbls_g2_clear_cofactor_batch(ctx, n, in, out, (int8_t*)st) : p_nbls_g1_clear_cofactor_batch(ctx, n, in, out, (int8_t*)st);
  if (r) return throw_code(env, r); return result2(env, vo, vs);
}
/* g2Prepare(g2 affine n*192) -> Uint8Array n*19584: PointG2.pairingPrecomputes() as 68 x [Fp2, Fp2, Fp2] in Fp2.toBytes order */
static napi_value G2Prepare(napi_env env, napi_callback_info info) {
  ARGS(1); NEED_CTX(); BYTES(0, in, l); size_t n = l / 192; if (l != n * 192) { napi_throw_range_error(env, NULL, "bad point array length"); return NULL; }
  uint8_t* out; napi_value vo = new_u8(env, n * NBLS_LINE_WIRE_BYTES, &out); ALLOCATED(vo);
  int r = p_nbls_g2_prepare(ctx, n, in, out); if (r) return throw_code(env, r); return vo;
}
/* pairingPrepared(g1 affine n*96, tables (n or 1)*19584, withFinalExp, product) -> Uint8Array n*576 (or 576 for product) */
static napi_value PairingPrepared(napi_env env, napi_callback_info info) {
  ARGS(4); NEED_CTX(); BYTES(0, g1, l1); BYTES(1, tab, lt); bool fe, prod; napi_get_value_bool(env, argv[2], &fe); napi_get_value_bool(env, argv[3], &prod);
  size_t n = l1 / 96, nt = lt / NBLS_LINE_WIRE_BYTES; if (l1 != n * 96 || lt != nt * NBLS_LINE_WIRE_BYTES || (n && nt != 1 && nt != n)) { napi_throw_range_error(env, NULL, "bad array lengths"); return NULL; }
  uint8_t* out; napi_value vo = new_u8(env, prod ? 576 : n * 576, &out); ALLOCATED(vo);
  int r = p_nbls_pairing_prepared(ctx, n, g1, tab, nt, fe, prod, out); if (r) return throw_code(env, r); return vo;
}
/* pairingBatch(g1, g2, withFinalExp, validate) -> {out, status} */
static napi_value PairingBatch(napi_env env, napi_callback_info info) {
  ARGS(4); NEED_CTX(); BYTES(0, g1, l1); BYTES(1, g2, l2); bool fe, val; napi_get_value_bool(env, argv[2], &fe); napi_get_value_bool(env, argv[3], &val);
  size_t n = l1 / 96; if (l1 != n * 96 || l2 != n * 192) { napi_throw_range_error(env, NULL, "bad point array lengths"); return NULL; }
  uint8_t *out, *st; napi_value vo = new_u8(env, n * 576, &out), vs = new_u8(env, n, &st);
  ALLOCATED(vo); ALLOCATED(vs);
  int r = MULTI() ? p_nbls_multi_pairing_batch(multi, n, g1, g2, fe, val, out, (int8_t*)st) : p_nbls_pairing_batch(ctx, n, g1, g2, fe, val, out, (int8_t*)st); if (r) return throw_code(env, r);
  return result2(env, vo, vs);
}
/* millerProduct(g1, g2, finalExp, validate) -> {out, status, code} */
static napi_value MillerProduct(napi_env env, napi_callback_info info) {
  ARGS(4); NEED_CTX(); BYTES(0, g1, l1); BYTES(1, g2, l2); bool fe, val; napi_get_value_bool(env, argv[2], &fe); napi_get_value_bool(env, argv[3], &val);
  size_t n = l1 / 96; if (l1 != n * 96 || l2 != n * 192) { napi_throw_range_error(env, NULL, "bad point array lengths"); return NULL; }
  uint8_t *out, *st; napi_value vo = new_u8(env, 576, &out), vs = new_u8(env, n, &st);
  ALLOCATED(vo); ALLOCATED(vs);
  int r = MULTI() ? p_nbls_multi_miller_product(multi, n, g1, g2, fe, val, out, (int8_t*)st) : p_nbls_miller_product(ctx, n, g1, g2, fe, val, out, (int8_t*)st); if (r && r != NBLS_EDECODE) return throw_code(env, r);
  napi_value o = result2(env, vo, vs), c; napi_create_int32(env, r, &c); napi_set_named_property(env, o, "code", c); return o;
}
static napi_value FinalExpBatch(napi_env env, napi_callback_info info) {
  ARGS(1); NEED_CTX(); BYTES(0, in, l); size_t n = l / 576; uint8_t* out; napi_value vo = new_u8(env, n * 576, &out);
  int r = p_nbls_final_exp_batch(ctx, n, in, out); if (r) return throw_code(env, r); return vo;
}
#define POINT_FN(NAME, CALL, IN_SZ, OUT_SZ) \
  static napi_value NAME(napi_env env, napi_callback_info info) { ARGS(1); NEED_CTX(); BYTES(0, in, l); size_t n = l / (IN_SZ); \
    uint8_t *out, *st; napi_value vo = new_u8(env, n * (OUT_SZ), &out), vs = new_u8(env, n, &st); \
    int r = CALL; if (r) return throw_code(env, r); return result2(env, vo, vs); }
POINT_FN(G1Decompress, p_nbls_g1_decompress_batch(ctx, n, in, out, (int8_t*)st), 48, 96)
POINT_FN(G2Decompress, p_nbls_g2_decompress_batch(ctx, n, in, out, (int8_t*)st), 96, 192)
POINT_FN(G1Validate, p_nbls_g1_validate_batch(ctx, n, in, (int8_t*)st), 96, 0)
POINT_FN(G2Validate, p_nbls_g2_validate_batch(ctx, n, in, (int8_t*)st), 192, 0)
static napi_value G1Sum(napi_env env, napi_callback_info info) { ARGS(1); NEED_CTX(); BYTES(0, in, l); uint8_t *out, *st; napi_value vo = new_u8(env, 96, &out), vs = new_u8(env, 1, &st);
  int r = p_nbls_g1_sum(ctx, l / 96, in, out, (int8_t*)st); if (r) return throw_code(env, r); return result2(env, vo, vs); }
static napi_value G2Sum(napi_env env, napi_callback_info info) { ARGS(1); NEED_CTX(); BYTES(0, in, l); uint8_t *out, *st; napi_value vo = new_u8(env, 192, &out), vs = new_u8(env, 1, &st);
  int r = p_nbls_g2_sum(ctx, l / 192, in, out, (int8_t*)st); if (r) return throw_code(env, r); return result2(env, vo, vs); }
/* hashToG2(msgs, offsets(Uint32Array n+1), dst) -> Uint8Array n*192 */
static napi_value HashToG2(napi_env env, napi_callback_info info) {
  ARGS(3); NEED_CTX(); BYTES(0, msgs, lm); BYTES(1, offs, lo); BYTES(2, dst, ld); COUNT_FROM_OFFSETS(n, lo); (void)lm;
  uint8_t* out; napi_value vo = new_u8(env, n * 192, &out);
  int r = p_nbls_hash_to_g2_batch(ctx, n, msgs, (const uint32_t*)offs, dst, ld, out); if (r) return throw_code(env, r); return vo;
}
/* g1Mul(points96 | null, scalars32) / g2Mul(points192, scalars32) -> {out, status}: [k_i]P_i, G1 generator when points is null */
static napi_value G1Mul(napi_env env, napi_callback_info info) {
  ARGS(2); NEED_CTX(); napi_valuetype t; napi_typeof(env, argv[0], &t); uint8_t* pts = NULL; size_t lp = 0;
  if (t != napi_null && t != napi_undefined && !get_bytes(env, argv[0], &pts, &lp)) { napi_throw_type_error(env, NULL, "expected Uint8Array or null"); return NULL; }
  BYTES(1, sc, ls); size_t n = ls / 32; if (pts && lp != n * 96) { napi_throw_range_error(env, NULL, "bad point array length"); return NULL; }
  uint8_t *out, *st; napi_value vo = new_u8(env, n * 96, &out), vs = new_u8(env, n, &st);
  int r = p_nbls_g1_mul_batch(ctx, n, pts, sc, out, (int8_t*)st); if (r) return throw_code(env, r); return result2(env, vo, vs);
}
static napi_value G2Mul(napi_env env, napi_callback_info info) {
  ARGS(2); NEED_CTX(); BYTES(0, pts, lp); BYTES(1, sc, ls); size_t n = ls / 32; if (lp != n * 192) { napi_throw_range_error(env, NULL, "bad point array length"); return NULL; }
  uint8_t *out, *st; napi_value vo = new_u8(env, n * 192, &out), vs = new_u8(env, n, &st);
  int r = p_nbls_g2_mul_batch(ctx, n, pts, sc, out, (int8_t*)st); if (r) return throw_code(env, r); return result2(env, vo, vs);
}
/* g1Msm(points96, scalars32) / g2Msm(points192, scalars32) -> {out: one affine point, status}: sum_i [k_i]P_i (status 1 = zero point) */
static napi_value G1Msm(napi_env env, napi_callback_info info) {
  ARGS(2); NEED_CTX(); BYTES(0, pts, lp); BYTES(1, sc, ls); size_t n = ls / 32; if (lp != n * 96) { napi_throw_range_error(env, NULL, "bad point array length"); return NULL; }
  uint8_t *out, *st; napi_value vo = new_u8(env, 96, &out), vs = new_u8(env, 1, &st);
  int r = p_nbls_g1_msm(ctx, n, pts, sc, out, (int8_t*)st); if (r) return throw_code(env, r); return result2(env, vo, vs);
}
static napi_value G2Msm(napi_env env, napi_callback_info info) {
  ARGS(2); NEED_CTX(); BYTES(0, pts, lp); BYTES(1, sc, ls); size_t n = ls / 32; if (lp != n * 192) { napi_throw_range_error(env, NULL, "bad point array length"); return NULL; }
  uint8_t *out, *st; napi_value vo = new_u8(env, 192, &out), vs = new_u8(env, 1, &st);
  int r = p_nbls_g2_msm(ctx, n, pts, sc, out, (int8_t*)st); if (r) return throw_code(env, r); return result2(env, vo, vs);
}
/* signBatch(msgs, offsets(Uint32Array n+1), dst, keys32) -> {out: n*192 affine signature points, status} */
static napi_value SignBatch(napi_env env, napi_callback_info info) {
  ARGS(4); NEED_CTX(); BYTES(0, msgs, lm); BYTES(1, offs, lo); BYTES(2, dst, ld); BYTES(3, keys, lk); (void)lm;
  COUNT_FROM_OFFSETS(n, lo); if (lk != n * 32) { napi_throw_range_error(env, NULL, "bad key array length"); return NULL; }
  uint8_t *out, *st; napi_value vo = new_u8(env, n * 192, &out), vs = new_u8(env, n, &st);
  int r = p_nbls_sign_batch(ctx, n, msgs, (const uint32_t*)offs, dst, ld, keys, out, (int8_t*)st); if (r) return throw_code(env, r); return result2(env, vo, vs);
}
/* hashToCurve(mode, msgs, offsets, dst): mode 0 = G1 hashToCurve, 1 = G1 encodeToCurve, 2 = G2 encodeToCurve -> Uint8Array n*96 / n*192 */
static napi_value HashToCurve(napi_env env, napi_callback_info info) {
  ARGS(4); NEED_CTX(); int32_t mode = 0; napi_get_value_int32(env, argv[0], &mode); BYTES(1, msgs, lm); BYTES(2, offs, lo); BYTES(3, dst, ld); (void)lm;
  COUNT_FROM_OFFSETS(n, lo); size_t a = mode == 2 ? 192 : 96; uint8_t* out; napi_value vo = new_u8(env, n * a, &out);
  int r = mode == 0 ? p_nbls_hash_to_g1_batch(ctx, n, msgs, (const uint32_t*)offs, dst, ld, out) : mode == 1 ? p_nbls_encode_to_g1_batch(ctx, n, msgs, (const uint32_t*)offs, dst, ld, out)
                    : p_nbls_encode_to_g2_batch(ctx, n, msgs, (const uint32_t*)offs, dst, ld, out);
  if (r) return throw_code(env, r); return vo;
}
/* verifyBatch(sig96, msgs, offsets, pks48, dst) -> {code, ok} */
static napi_value VerifyBatch(napi_env env, napi_callback_info info) {
  ARGS(5); NEED_CTX(); BYTES(0, sig, ls); BYTES(1, msgs, lm); BYTES(2, offs, lo); BYTES(3, pks, lp); BYTES(4, dst, ld); (void)lm; (void)ls;
  COUNT_FROM_OFFSETS(n, lo); if (lp != n * 48) { napi_throw_range_error(env, NULL, "bad public key array length"); return NULL; }
  int ok = 0; int r = MULTI() ? p_nbls_multi_verify_batch(multi, n, sig, msgs, (const uint32_t*)offs, pks, dst, ld, &ok) : p_nbls_verify_batch(ctx, n, sig, msgs, (const uint32_t*)offs, pks, dst, ld, &ok);
  if (r && r != NBLS_EDECODE) return throw_code(env, r);
  napi_value o, c, k; napi_create_object(env, &o); napi_create_int32(env, r, &c); napi_get_boolean(env, ok != 0, &k);
  napi_set_named_property(env, o, "code", c); napi_set_named_property(env, o, "ok", k); return o;
}

/* verifyBatchAsync(sig96, msgs, offsets, pks48, dst) -> Promise<{code, ok}>: the same call on a libuv worker thread (napi_create_async_work),
 * so that the event loop keeps running during the ~1-100 ms a batch takes.  The typed arrays are kept alive by references until
 * the work completes; the engine context serialises concurrent calls itself. */
typedef struct {
  napi_async_work work; napi_deferred deferred; napi_ref refs[5];
  const uint8_t *sig, *msgs, *pks, *dst; const uint32_t* offs; size_t n, dst_len;
  nbls_ctx* c;      /* the pool context this call runs on */
  int rc, ok;
} verify_job;
static void verify_execute(napi_env env, void* data) { verify_job* j = (verify_job*)data; (void)env;
  j->rc = MULTI() ? p_nbls_multi_verify_batch(multi, j->n, j->sig, j->msgs, j->offs, j->pks, j->dst, j->dst_len, &j->ok) : p_nbls_verify_batch(j->c, j->n, j->sig, j->msgs, j->offs, j->pks, j->dst, j->dst_len, &j->ok); }
static void verify_complete(napi_env env, napi_status status, void* data) {
  verify_job* j = (verify_job*)data;
  for (int i = 0; i < 5; i++) napi_delete_reference(env, j->refs[i]);
  if (status != napi_ok || (j->rc && j->rc != NBLS_EDECODE)) {
    char m[128]; snprintf(m, sizeof m, "nbls: %s (code %d)", p_nbls_strerror ? p_nbls_strerror(j->rc) : "error", j->rc);
    napi_value msg, err; napi_create_string_utf8(env, m, NAPI_AUTO_LENGTH, &msg); napi_create_error(env, NULL, msg, &err); napi_reject_deferred(env, j->deferred, err);
  } else {
    napi_value o, c, k; napi_create_object(env, &o); napi_create_int32(env, j->rc, &c); napi_get_boolean(env, j->ok != 0, &k);
    napi_set_named_property(env, o, "code", c); napi_set_named_property(env, o, "ok", k); napi_resolve_deferred(env, j->deferred, o);
  }
  napi_delete_async_work(env, j->work); free(j);
}
static napi_value VerifyBatchAsync(napi_env env, napi_callback_info info) {
  ARGS(5); NEED_CTX(); BYTES(0, sig, ls); BYTES(1, msgs, lm); BYTES(2, offs, lo); BYTES(3, pks, lp); BYTES(4, dst, ld); (void)lm; (void)ls;
  COUNT_FROM_OFFSETS(n, lo); if (lp != n * 48) { napi_throw_range_error(env, NULL, "bad public key array length"); return NULL; }
  verify_job* j = (verify_job*)calloc(1, sizeof *j); if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
  j->sig = sig; j->msgs = msgs; j->offs = (const uint32_t*)offs; j->pks = pks; j->dst = dst; j->dst_len = ld; j->n = n; j->c = pool_take();
  for (int i = 0; i < 5; i++) napi_create_reference(env, argv[i], 1, &j->refs[i]);
  napi_value promise, name; napi_create_promise(env, &j->deferred, &promise); napi_create_string_utf8(env, "nbls_verify_batch", NAPI_AUTO_LENGTH, &name);
  if (napi_create_async_work(env, NULL, name, verify_execute, verify_complete, j, &j->work) != napi_ok || napi_queue_async_work(env, j->work) != napi_ok) {
    for (int i = 0; i < 5; i++) napi_delete_reference(env, j->refs[i]); free(j); napi_throw_error(env, NULL, "napi_create_async_work failed"); return NULL; }
  return promise;
}


/* signBatchAsync(msgs, offsets, dst, keys32) -> Promise<{out: n*192 affine signature points, status}>: nbls_sign_batch on a libuv worker thread (the reference's
 * sign is async, index.ts:744-752).  The output arrays are created here, on the main thread, and kept alive by references like the inputs. */
typedef struct {
  napi_async_work work; napi_deferred deferred; napi_ref refs[6];
  const uint8_t *msgs, *dst, *keys; const uint32_t* offs; size_t n, dst_len; uint8_t *out; int8_t* st;
  nbls_ctx* c; int rc;
} sign_job;
static void sign_execute(napi_env env, void* data) { sign_job* j = (sign_job*)data; (void)env;
  j->rc = p_nbls_sign_batch(j->c, j->n, j->msgs, j->offs, j->dst, j->dst_len, j->keys, j->out, j->st); }
static void sign_complete(napi_env env, napi_status status, void* data) {
  sign_job* j = (sign_job*)data;
  if (status != napi_ok || j->rc) {
    char m[128]; snprintf(m, sizeof m, "nbls: %s (code %d)", p_nbls_strerror ? p_nbls_strerror(j->rc) : "error", j->rc);
    napi_value msg, err; napi_create_string_utf8(env, m, NAPI_AUTO_LENGTH, &msg); napi_create_error(env, NULL, msg, &err); napi_reject_deferred(env, j->deferred, err);
  } else {
    napi_value vo, vs; napi_get_reference_value(env, j->refs[4], &vo); napi_get_reference_value(env, j->refs[5], &vs);
    napi_resolve_deferred(env, j->deferred, result2(env, vo, vs));
  }
  for (int i = 0; i < 6; i++) napi_delete_reference(env, j->refs[i]);
  napi_delete_async_work(env, j->work); free(j);
}
static napi_value SignBatchAsync(napi_env env, napi_callback_info info) {
  ARGS(4); NEED_CTX(); BYTES(0, msgs, lm); BYTES(1, offs, lo); BYTES(2, dst, ld); BYTES(3, keys, lk); (void)lm;
  COUNT_FROM_OFFSETS(n, lo); if (lk != n * 32) { napi_throw_range_error(env, NULL, "bad key array length"); return NULL; }
  sign_job* j = (sign_job*)calloc(1, sizeof *j); if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
  uint8_t *out, *st; napi_value vo = new_u8(env, n * 192, &out), vs = new_u8(env, n, &st);
  j->msgs = msgs; j->offs = (const uint32_t*)offs; j->dst = dst; j->dst_len = ld; j->keys = keys; j->n = n; j->out = out; j->st = (int8_t*)st; j->c = pool_take();
  for (int i = 0; i < 4; i++) napi_create_reference(env, argv[i], 1, &j->refs[i]);
  napi_create_reference(env, vo, 1, &j->refs[4]); napi_create_reference(env, vs, 1, &j->refs[5]);
  napi_value promise, name; napi_create_promise(env, &j->deferred, &promise); napi_create_string_utf8(env, "nbls_sign_batch", NAPI_AUTO_LENGTH, &name);
  if (napi_create_async_work(env, NULL, name, sign_execute, sign_complete, j, &j->work) != napi_ok || napi_queue_async_work(env, j->work) != napi_ok) {
    for (int i = 0; i < 6; i++) napi_delete_reference(env, j->refs[i]); free(j); napi_throw_error(env, NULL, "napi_create_async_work failed"); return NULL; }
  return promise;
}

static napi_value ModuleInit(napi_env env, napi_value exports) {
  const char* path = getenv("NBLS_LIB");
  char buf[4096];
  if (!path) { Dl_info di; if (dladdr((void*)ModuleInit, &di) && di.dli_fname) { snprintf(buf, sizeof buf, "%s", di.dli_fname); char* s = strrchr(buf, '/'); if (s) { *s = 0; s = strrchr(buf, '/'); if (s) { snprintf(s, sizeof buf - (s - buf), "/libnbls.so"); path = buf; } } } }
  lib = dlopen(path ? path : "libnbls.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) { napi_throw_error(env, NULL, dlerror()); return exports; }
#define LOAD(name) p_##name = (__typeof__(p_##name))dlsym(lib, #name); if (!p_##name) { napi_throw_error(env, NULL, "libnbls.so lacks " #name); return exports; }
  LOAD(nbls_init) LOAD(nbls_destroy) LOAD(nbls_strerror) LOAD(nbls_pairing_batch) LOAD(nbls_miller_product) LOAD(nbls_final_exp_batch)
  LOAD(nbls_g1_validate_batch) LOAD(nbls_g2_validate_batch) LOAD(nbls_g1_decompress_batch) LOAD(nbls_g2_decompress_batch)
  LOAD(nbls_hash_to_g2_batch) LOAD(nbls_g1_sum) LOAD(nbls_g2_sum) LOAD(nbls_verify_batch) LOAD(nbls_g1_mul_batch) LOAD(nbls_g2_mul_batch) LOAD(nbls_sign_batch) LOAD(nbls_hash_to_g1_batch) LOAD(nbls_encode_to_g1_batch) LOAD(nbls_encode_to_g2_batch) LOAD(nbls_g1_msm) LOAD(nbls_g2_msm)
  LOAD(nbls_init_multi) LOAD(nbls_destroy_multi) LOAD(nbls_multi_device_count) LOAD(nbls_multi_context) LOAD(nbls_multi_pairing_batch) LOAD(nbls_multi_miller_product) LOAD(nbls_multi_verify_batch) LOAD(nbls_g2_prepare) LOAD(nbls_pairing_prepared)
  LOAD(nbls_g1_from_hex_batch) LOAD(nbls_g2_from_hex_batch) LOAD(nbls_g2_from_signature_batch) LOAD(nbls_g1_clear_cofactor_batch) LOAD(nbls_g2_clear_cofactor_batch)
  {   /* the ABI the addon was written against (include/nbls.h NBLS_ABI_VERSION): an older or newer library is refused at load instead of misread at run time */
    int (*abi)(void) = (int (*)(void))dlsym(lib, "nbls_abi_version");
    if (!abi || abi() != NBLS_ABI_VERSION) { napi_throw_error(env, NULL, "libnbls.so: ABI version differs from the one this addon was built for (include/nbls.h NBLS_ABI_VERSION)"); return exports; }
  }
  napi_property_descriptor d[] = {
    {"init", 0, Init, 0, 0, 0, napi_enumerable, 0}, {"pairingBatch", 0, PairingBatch, 0, 0, 0, napi_enumerable, 0}, {"millerProduct", 0, MillerProduct, 0, 0, 0, napi_enumerable, 0},
    {"finalExpBatch", 0, FinalExpBatch, 0, 0, 0, napi_enumerable, 0}, {"g1Decompress", 0, G1Decompress, 0, 0, 0, napi_enumerable, 0}, {"g2Decompress", 0, G2Decompress, 0, 0, 0, napi_enumerable, 0},
    {"g1Validate", 0, G1Validate, 0, 0, 0, napi_enumerable, 0}, {"g2Validate", 0, G2Validate, 0, 0, 0, napi_enumerable, 0}, {"g1Sum", 0, G1Sum, 0, 0, 0, napi_enumerable, 0},
    {"g2Sum", 0, G2Sum, 0, 0, 0, napi_enumerable, 0}, {"hashToG2", 0, HashToG2, 0, 0, 0, napi_enumerable, 0}, {"verifyBatch", 0, VerifyBatch, 0, 0, 0, napi_enumerable, 0},
    {"g1Mul", 0, G1Mul, 0, 0, 0, napi_enumerable, 0}, {"g2Mul", 0, G2Mul, 0, 0, 0, napi_enumerable, 0}, {"signBatch", 0, SignBatch, 0, 0, 0, napi_enumerable, 0},
    {"hashToCurve", 0, HashToCurve, 0, 0, 0, napi_enumerable, 0}, {"g1Msm", 0, G1Msm, 0, 0, 0, napi_enumerable, 0}, {"g2Msm", 0, G2Msm, 0, 0, 0, napi_enumerable, 0}, {"verifyBatchAsync", 0, VerifyBatchAsync, 0, 0, 0, napi_enumerable, 0}, {"signBatchAsync", 0, SignBatchAsync, 0, 0, 0, napi_enumerable, 0},
    {"initMulti", 0, InitMulti, 0, 0, 0, napi_enumerable, 0}, {"g2Prepare", 0, G2Prepare, 0, 0, 0, napi_enumerable, 0}, {"pairingPrepared", 0, PairingPrepared, 0, 0, 0, napi_enumerable, 0},
    {"decodePoints", 0, DecodePoints, 0, 0, 0, napi_enumerable, 0}, {"clearCofactor", 0, ClearCofactor, 0, 0, 0, napi_enumerable, 0}};
  napi_define_properties(env, exports, sizeof d / sizeof d[0], d);
  return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, ModuleInit)
