/*
 * nbls.h -- C ABI of the MI355X batched BLS12-381 pairing engine (libnbls.so).
 *
 * This is the drop-in boundary beneath the reference's exported TypeScript API (paulmillr/noble-bls12-381 v1.4.0).
 * The reference has no native/FFI layer; every entry point below replaces the part of a reference function that
 * runs between decoding its arguments and encoding its result, and names that function (file:line in the reference).
 * INTEGRATION.md shows the N-API / ctypes stubs that bind these symbols behind noble's names.
 *
 * Conventions
 *  - Wire format = the reference's own byte encodings: field elements are 48-byte big-endian (Fp.toBytes,
 *    math.ts:284-290); Fp2 = c0||c1 (math.ts:540-549); Fp12 = 576 bytes in Fp12.toBytes order (math.ts:875-884).
 *    G1 affine = x||y (96 B); G2 affine = x.c0||x.c1||y.c0||y.c1 (192 B).
 *  - All buffers are caller-owned and contiguous; the library keeps no pointer after a call returns.
 *  - Functions return 0 or a negative NBLS_E* code and never throw or abort.  Per-item `status` (may be NULL):
 *    0 ok, 1 point at infinity, 2 not on curve, 3 not in the prime-order subgroup, 4 bad encoding;
 *    +10 when the offending point is the G2 argument of a pairing.
 *  - `*_dev` variants take DEVICE pointers (HIP) and a hipStream_t (as void*); they enqueue work and do not synchronise.
 *  - A context is bound to one GPU; calls on one context are serialised internally (thread-safe per context).
 */
#ifndef NBLS_H
#define NBLS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbls_ctx nbls_ctx;

#define NBLS_OK 0
#define NBLS_EINVAL (-1)      /* bad argument */
#define NBLS_EHIP (-2)        /* HIP runtime error (nbls_last_hip_error) */
#define NBLS_ENOSUP (-3)      /* not implemented in this build */
#define NBLS_ENOGPU (-4)      /* no usable gfx950 device */
#define NBLS_EDECODE (-5)     /* an input point failed to decode where the reference throws (see per-item status) */

/* Create / destroy an engine context on HIP device `device_id` (compiles the step programs, uploads them). */
int nbls_init(int device_id, nbls_ctx** out);
void nbls_destroy(nbls_ctx* ctx);
const char* nbls_strerror(int code);
int nbls_last_hip_error(nbls_ctx* ctx);

/* pairing(P, Q, withFinalExponent) for n independent pairs -- reference index.ts:715-722.
 * validate != 0 reproduces P.assertValidity()/Q.assertValidity() (index.ts:383-388, 633-638) and the infinity
 * check (index.ts:716) as per-item status; items with a non-zero status get an all-zero output. */
int nbls_pairing_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, const uint8_t* g2_aff, int with_final_exp, int validate,
                       uint8_t* out_fp12, int8_t* status);
int nbls_pairing_batch_dev(nbls_ctx* ctx, size_t n, const void* d_g1_aff, const void* d_g2_aff, int with_final_exp,
                           void* d_out_fp12, void* stream);

/* prod_i millerLoop(P_i, Q_i), optionally followed by ONE shared finalExponentiate -- the core of verify
 * (index.ts:763-766) and verifyBatch (index.ts:811-816). */
int nbls_miller_product(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, const uint8_t* g2_aff, int final_exp, int validate,
                        uint8_t* out_fp12, int8_t* status);
int nbls_miller_product_dev(nbls_ctx* ctx, size_t n, const void* d_g1_aff, const void* d_g2_aff, int final_exp, void* d_out_fp12, void* stream);

/* Prepared G2 points.  PointG2.pairingPrecomputes() (index.ts:703-711) memoises calcPairingPrecomputes (math.ts:1331-1371): the 68 line
 * triples [Fp2, Fp2, Fp2] of the Miller loop, which depend on Q alone; PointG1.millerLoop (index.ts:452-454) then runs millerLoop
 * (math.ts:1373-1388) over them.  A table lives on the device in the engine's raw limb format (NBLS_LINE_TABLE_BYTES per point) or
 * crosses the ABI as the reference's value, 68 x (c0 || c1 || c2) in Fp2.toBytes order (NBLS_LINE_WIRE_BYTES per point).  Many P
 * against one Q (one message signed by many keys, index.ts:804-812) pass table_stride = 0 / n_tables = 1. */
#define NBLS_LINE_TABLE_BYTES 26112
#define NBLS_LINE_WIRE_BYTES 19584
int nbls_g2_prepare(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, uint8_t* out_tables_wire);
int nbls_g2_prepare_dev(nbls_ctx* ctx, size_t n, const void* d_g2_aff, void* d_tables, void* stream);
int nbls_lines_to_wire_dev(nbls_ctx* ctx, size_t n, const void* d_tables, void* d_tables_wire, void* stream);
int nbls_lines_from_wire_dev(nbls_ctx* ctx, size_t n, const void* d_tables_wire, void* d_tables, void* stream);
/* pairing(P_i, Q_i) / prod_i millerLoop(P_i, Q_i) with prepared Q: table_stride = NBLS_LINE_TABLE_BYTES (a table per item) or 0 (one table) */
int nbls_pairing_prepared_dev(nbls_ctx* ctx, size_t n, const void* d_g1_aff, const void* d_tables, size_t table_stride, int with_final_exp,
                              void* d_out_fp12, void* stream);
int nbls_miller_product_prepared_dev(nbls_ctx* ctx, size_t n, const void* d_g1_aff, const void* d_tables, size_t table_stride, int final_exp,
                                     void* d_out_fp12, void* stream);
/* host buffers: n_tables = n or 1 tables in wire form; product != 0 returns ONE Fp12 (the product, optionally final-exponentiated) */
int nbls_pairing_prepared(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, const uint8_t* tables_wire, size_t n_tables, int with_final_exp,
                          int product, uint8_t* out_fp12);

/* Fp12.finalExponentiate for n elements -- reference math.ts:856-874. */
int nbls_final_exp_batch(nbls_ctx* ctx, size_t n, const uint8_t* in_fp12, uint8_t* out_fp12);
int nbls_final_exp_batch_dev(nbls_ctx* ctx, size_t n, const void* d_in_fp12, void* d_out_fp12, void* stream);

/* Single tower operations on the device (round 4): the reference's own unit-test surface for Fp / Fp2 / Fp6 / Fp12 (test/fp.test.ts, fp2.test.ts, fp12.test.ts) as
 * one batched entry point, so that the tower rows of the scope table have known-answer tests of their own on the GPU and not only through pairings.
 * field = 1, 2, 6, 12 (elements of 48 * field wire bytes, Fp.toBytes / Fp2.toBytes / Fp12.toBytes order); op = NBLS_TOP_*; param = the power of a Frobenius map.
 * a: n elements; b: n elements (binary operations) or n Fp2 elements (sparse products); c, d: n Fp2 elements (sparse products); unused operands NULL.
 * Replaces: Fp add / subtract / negate / multiply / square / invert (math.ts:223-273, 134-156), Fp2 ... multiplyByB / mulByNonresidue / frobeniusMap / invert
 * (math.ts:451-539), Fp6 ... multiplyBy1 / multiplyBy01 / frobeniusMap / invert (math.ts:601-688), Fp12 ... multiplyBy014 / conjugate / frobeniusMap / invert /
 * cyclotomicSquare / cyclotomicExp(x) (math.ts:732-852).  Inverting zero is undefined (the reference throws): the output element is then unspecified.
 * Returns NBLS_EINVAL for a combination the reference does not have (e.g. conjugate on Fp6). */
#define NBLS_TOP_ADD 0
#define NBLS_TOP_SUB 1
#define NBLS_TOP_NEG 2
#define NBLS_TOP_MUL 3
#define NBLS_TOP_SQR 4
#define NBLS_TOP_INV 5
#define NBLS_TOP_FROBENIUS 6        /* Fp2, Fp6, Fp12: frobeniusMap(param), 0 <= param <= 11 */
#define NBLS_TOP_CONJUGATE 7        /* Fp2 (= frobeniusMap(1)), Fp12 */
#define NBLS_TOP_MUL_BY_NONRESIDUE 8 /* Fp2: * (1 + u); Fp6: * v */
#define NBLS_TOP_MUL_BY_B 9         /* Fp2: * 4 (1 + u) */
#define NBLS_TOP_MUL_BY_1 10        /* Fp6: multiplyBy1(b) */
#define NBLS_TOP_MUL_BY_01 11       /* Fp6: multiplyBy01(b, c) */
#define NBLS_TOP_MUL_BY_014 12      /* Fp12: multiplyBy014(b, c, d) */
#define NBLS_TOP_CYCLOTOMIC_SQUARE 13 /* Fp12, unitary input */
#define NBLS_TOP_CYCLOTOMIC_EXP 14  /* Fp12, unitary input: cyclotomicExp(CURVE.x) */
int nbls_tower_op_batch(nbls_ctx* ctx, int field, int op, int param, size_t n, const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d, uint8_t* out);


/* Raw device partial product for multi-GPU reductions: prod_i millerLoop(P_i,Q_i) WITHOUT final exponentiation as
 * 576 wire bytes on the device (one per rank; ranks exchange them and finish with nbls_fp12_product_final_dev). */
int nbls_fp12_product_final_dev(nbls_ctx* ctx, size_t n, const void* d_in_fp12, int final_exp, void* d_out_fp12, void* stream);

/* P.assertValidity() for affine points -- reference index.ts:383-388 (G1), 633-638 (G2): status 0 / 2 / 3. */
int nbls_g1_validate_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, int8_t* status);
int nbls_g2_validate_batch(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, int8_t* status);

/* PointG1.fromHex for 48-byte compressed keys (index.ts:298-327) and PointG2.fromSignature for 96-byte compressed
 * signatures (index.ts:500-530): affine output + status (0 ok, 1 zero point, 3 not in subgroup, 4 no square root). */
int nbls_g1_decompress_batch(nbls_ctx* ctx, size_t n, const uint8_t* in48, uint8_t* out96, int8_t* status);
int nbls_g2_decompress_batch(nbls_ctx* ctx, size_t n, const uint8_t* in96, uint8_t* out192, int8_t* status);

/* Every wire form of the reference's point codecs, in bulk.  `len` = bytes per encoded point.
 *   nbls_g1_from_hex_batch        PointG1.fromHex (index.ts:298-327): len 48 compressed | 96 uncompressed (x || y, infinity flag 0x40)
 *   nbls_g2_from_hex_batch        PointG2.fromHex (index.ts:532-579): len 96 compressed -- flag rules, root chosen by the S bit, NO subgroup
 *                                 check, exactly as the reference -- | 192 uncompressed (x.c1 || x.c0 || y.c1 || y.c0, infinity flag 0x40)
 *   nbls_g2_from_signature_batch  PointG2.fromSignature (index.ts:500-530): len 96, or 192 (z1 and z2 read as 96-byte integers)
 * Output: canonical affine wire bytes.  status: 0 ok, 1 zero point, 2 not on curve, 3 not in the prime-order subgroup, 4 no square root,
 * 6 invalid encoding flag (first byte & 0xe0 in {0x20, 0x60, 0xe0}), 7 infinity flag with other bits set, 8 compression bit clear on 96 bytes.
 *   nbls_g*_to_hex_batch          PointG1.toHex / PointG2.toHex (index.ts:359-381, 603-631) of valid points: compressed != 0 -> 48 / 96 bytes,
 *                                 else 96 / 192 bytes (G2 in c1 || c0 order); zero (may be NULL) marks zero points.
 *   nbls_g*_clear_cofactor_batch  PointG1.clearCofactor (index.ts:401-405), PointG2.clearCofactor (index.ts:659-672) for points on the curve. */
int nbls_g1_from_hex_batch(nbls_ctx* ctx, size_t n, const uint8_t* in, size_t len, uint8_t* out96, int8_t* status);
int nbls_g2_from_hex_batch(nbls_ctx* ctx, size_t n, const uint8_t* in, size_t len, uint8_t* out192, int8_t* status);
int nbls_g2_from_signature_batch(nbls_ctx* ctx, size_t n, const uint8_t* in, size_t len, uint8_t* out192, int8_t* status);
int nbls_g1_to_hex_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, const int8_t* zero, int compressed, uint8_t* out);
int nbls_g2_to_hex_batch(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, const int8_t* zero, int compressed, uint8_t* out);
int nbls_g1_clear_cofactor_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, uint8_t* out96, int8_t* status);
int nbls_g2_clear_cofactor_batch(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, uint8_t* out192, int8_t* status);

/* PointG2.hashToCurve(msg, {DST}) for n messages (index.ts:481-490): msgs are concatenated, message i = msgs[offsets[i] ..
 * offsets[i+1]); SHA-256 expand_message_xmd (index.ts:207-231) and everything after it run on the GPU. */
int nbls_hash_to_g2_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out192);

/* PointG1.hashToCurve / PointG1.encodeToCurve (index.ts:331-350) and PointG2.encodeToCurve (index.ts:491-497): hash_to_field with
 * (count, m) = (2, 1) / (1, 1) / (1, 2), simplified SWU on the isogenous curve, isogeny, cofactor clearing.  Messages as above. */
int nbls_hash_to_g1_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out96);
int nbls_encode_to_g1_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out96);
int nbls_encode_to_g2_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len, uint8_t* out192);

/* Sum of n affine points: the reduce step of aggregatePublicKeys / aggregateSignatures (index.ts:771-788). *status = 1 when
 * the sum is the zero point (output then all-zero). */
int nbls_g1_sum(nbls_ctx* ctx, size_t n, const uint8_t* pts96, uint8_t* out96, int8_t* status);
int nbls_g2_sum(nbls_ctx* ctx, size_t n, const uint8_t* pts192, uint8_t* out192, int8_t* status);

/* PointG1.toHex(true) / toRawBytes(true) (index.ts:355-371) and PointG2.toSignature (index.ts:586-602) for n NON-ZERO affine points:
 * x with the compression flag (bit 383) and the sign flag (bit 381) = floor(2y / p).  The zero point (not representable as
 * affine wire bytes; the sum / multiplication calls report it as status 1) encodes as 0xc0 00...00 on the caller's side. */
int nbls_g1_compress_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, uint8_t* out48);
int nbls_g2_compress_batch(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff, uint8_t* out96);

/* [k_i]P_i for per-item scalars (32 bytes big-endian each; any value, the reference reduces mod r first: normalizePrivKey
 * index.ts:269-279).  g1_aff == NULL multiplies the G1 generator: the core of getPublicKey / PointG1.fromPrivateKey
 * (index.ts:350-353, 738-740).  Fixed-window ladder with masked table selection: instruction stream and memory access pattern do
 * not depend on the scalar.  status: 0 ok, 1 result is the zero point, 5 scalar is 0 mod r (the reference throws). */
int nbls_g1_mul_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff /* n*96 or NULL */, const uint8_t* scalars32, uint8_t* out96, int8_t* status);
int nbls_g2_mul_batch(nbls_ctx* ctx, size_t n, const uint8_t* g2_aff /* n*192 */, const uint8_t* scalars32, uint8_t* out192, int8_t* status);

/* Multi-scalar multiplication sum_i [k_i]P_i (weighted form of the reference's aggregatePublicKeys / aggregateSignatures,
 * index.ts:771-788, whose unweighted sums are nbls_g1_sum / nbls_g2_sum; the building block of random-linear-combination batch
 * verification of distinct signatures).  Points: affine wire bytes in the prime-order subgroup (nbls_g*_validate_batch); scalars:
 * 32 bytes big-endian each, any value.  Bucket method (12-bit windows) on the device; NOT constant time in the scalars (public
 * coefficients), use nbls_g*_mul_batch for secret ones.  out: one affine point; status: 0 ok, 1 the sum is the zero point. */
int nbls_g1_msm(nbls_ctx* ctx, size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out96, int8_t* status);
int nbls_g2_msm(nbls_ctx* ctx, size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out192, int8_t* status);
/* the same with points, scalars, result and status resident in device memory; all scalars < 2^nbits (0 = 256) */
int nbls_msm_dev(nbls_ctx* ctx, int g2, size_t n, const void* d_pts, const void* d_scalars32, unsigned nbits, void* d_out, void* d_status, void* stream);

/* sign(message_i, privateKey_i) -- reference index.ts:744-752: PointG2.hashToCurve(message) multiplied by the key; output is
 * the affine signature point (192 B), which PointG2.toSignature (index.ts:586-602) compresses on the caller's side.
 * Messages as in nbls_hash_to_g2_batch. */
int nbls_sign_batch(nbls_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* dst, size_t dst_len,
                    const uint8_t* keys32, uint8_t* out192, int8_t* status);
/* Same with everything resident in device memory (round 5): message bytes, n + 1 uint32 offsets into them, 32-byte big-endian keys -> n affine points (192 B) and n status
 * bytes in device memory; SHA-256 expand_message_xmd, hash-to-G2 and the ladder run as one chain on `stream` (NULL = the context's).  status: 0 ok, 1 the result is the zero
 * point, i.e. the key is 0 mod r (the host-buffer call reports 5 there, as normalizePrivKey's message demands; keys >= r are reduced, index.ts:269-279).  Synchronises;
 * NBLS_EINVAL if the offsets decrease somewhere. */
int nbls_sign_batch_dev(nbls_ctx* ctx, size_t n, const void* d_msgs, const void* d_offsets, const uint8_t* dst, size_t dst_len,
                        const void* d_keys32, void* d_out192, void* d_status, void* stream);

/* verifyBatch(signature, messages, publicKeys) on wire inputs -- reference index.ts:792-821 with every message distinct.
 * *ok = 1/0; returns NBLS_EDECODE where the reference throws while decoding its arguments. */
int nbls_verify_batch(nbls_ctx* ctx, size_t n, const uint8_t* sig96, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* pk48,
                      const uint8_t* dst, size_t dst_len, int* ok);
/* Same, inputs already in HBM: signature, 256-byte expand_message_xmd outputs, compressed keys.  Synchronises (returns *ok). */
/* Same with the MESSAGES resident in HBM (bytes + n + 1 uint32 offsets relative to d_msgs): expand_message_xmd runs on the device first, on the same stream
 * (the whole of verifyBatch index.ts:792-821 for wire-format inputs with nothing done on the host).  dst_len <= 255. */
int nbls_verify_batch_msgs_dev(nbls_ctx* ctx, size_t n, const void* d_sig96, const void* d_msgs, const void* d_offsets, const void* d_pk48, const uint8_t* dst, size_t dst_len,
                               int* ok, void* stream);
int nbls_verify_batch_dev_inputs(nbls_ctx* ctx, size_t n, const void* d_sig96, const void* d_uniform256, const void* d_pk48, int* ok,
                                 int8_t* pk_status /* n, may be NULL */, void* stream);

/* One rank's share of a verifyBatch spread over several GPUs (one process per GPU): the Miller product of this rank's n
 * (key, message) pairs, times millerLoop(-G, S) on the ONE rank that passes the signature (d_sig96 = NULL elsewhere), WITHOUT the
 * final exponentiation, as 576 wire bytes in device memory.  Ranks all-gather their partials and finish with
 * nbls_fp12_product_final_dev (product + shared finalExponentiate, index.ts:811-817), then compare with Fp12.ONE.
 * *zero_flag = 1: a zero point was met (verifyBatch answers false).  NBLS_EDECODE as nbls_verify_batch.  ABI 3: the call decides nothing on the host before its end, so d_out_fp12 IS
 * written in both cases -- with a meaningless product; look at the return code and the flag first. */
int nbls_verify_batch_partial_dev(nbls_ctx* ctx, size_t n, const void* d_sig96 /* or NULL */, const void* d_uniform256, const void* d_pk48,
                                  void* d_out_fp12, int* zero_flag, int8_t* pk_status /* n, may be NULL */, void* stream);

/* The same as one device's share of a product that is spread over several GPUs, from HOST inputs: on return *d_partial points at 576 wire
 * bytes on the context's device, ready for hipMemcpyPeer / a collective; the call returns when the partial is complete.  *d_partial is a pure
 * OUT parameter (ABI 2; ABI 1 of round 3 read it as well): it receives a buffer owned by the context, valid only until the context's next
 * *_partial call.  The `_into` forms write the partial to a caller-owned 576-byte device buffer on the context's device instead -- required when
 * several threads may use the context, because each call then owns its partial; a pointer that is not device memory of that device is refused
 * (NBLS_EINVAL).  An empty shard (n = 0, product only) yields the unit element. */
int nbls_miller_product_partial(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, const uint8_t* g2_aff, int validate, void** d_partial, int8_t* status);
int nbls_verify_batch_partial(nbls_ctx* ctx, size_t n, const uint8_t* sig96 /* or NULL */, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* pk48,
                              const uint8_t* dst, size_t dst_len, void** d_partial, int* zero_flag, int8_t* pk_status /* n, may be NULL */);
int nbls_miller_product_partial_into(nbls_ctx* ctx, size_t n, const uint8_t* g1_aff, const uint8_t* g2_aff, int validate, void* d_dst576, int8_t* status);
int nbls_verify_batch_partial_into(nbls_ctx* ctx, size_t n, const uint8_t* sig96 /* or NULL */, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* pk48,
                                   const uint8_t* dst, size_t dst_len, void* d_dst576, int* zero_flag, int8_t* pk_status /* n, may be NULL */);
const char* nbls_config_describe(void);   /* "NBLS_X=value(env|default) ...": every environment switch the library has read so far and the value in force -- print it next to an A/B result */
/* 4 (round 6): nbls_hw_queues, NBLS_TUNE_WIDE_MAX, NBLS_TUNE_H2C_NORM_MIN, NBLS_TUNE_INV_WIDE_MAX, NBLS_TUNE_LS_MAX / _LS2_MAX (additions only); the library sets GPU_MAX_HW_QUEUES = 22 at load when the variable is unset (see nbls_pool_init below).
   3 (round 5): nbls_program_kernel, nbls_pool_*, nbls_sign_batch_dev, NBLS_TUNE_VERIFY_* / _SAC_MAX / _PT_LS2_MAX (additions only); nbls_verify_batch_partial_dev writes d_out_fp12 even when it reports a zero point or a decode error
   (contents then meaningless); 2: *_partial take *d_partial as OUT only, *_partial_into added, nbls_tower_op_batch, nbls_verify_batch_msgs_dev.  The bindings check it at load. */
#define NBLS_ABI_VERSION 4
int nbls_abi_version(void);
int nbls_context_device(nbls_ctx* ctx);

/* ---- several calls in flight on ONE device (round 5): `depth` contexts, each with its own stream and scratch, fed round-robin (what noble-bls12-381_amd/pipeline.py does, for C
 * callers).  A 4096-pairing call alone fills the chip one wavefront deep and runs at ~0.27 of the multiply-add roofline; twelve overlapping calls (nbls_pool_*) reach ~0.45.
 * nbls_pool_pairing_batch_dev enqueues pairing(P_i, Q_i) (reference index.ts:715-722) for n device-resident pairs on the next context's stream and returns at once;
 * *slot (may be NULL) = the context used -- keep one output buffer per slot; nbls_pool_next_slot tells it in advance.
 * Hardware queues: the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read once when the runtime initialises; streams that share a queue
 * serialise (a pool of twelve on four queues: 2.50 M instead of 2.98 M pairings/s, tests/c/pool_rate.c).  Round 6: libnbls.so sets GPU_MAX_HW_QUEUES = 22 when it is loaded and the variable is unset
 * (NBLS_KEEP_HW_QUEUES=1: never), which is in time for every process that loads the library before its first HIP call -- a C program linked against it, the N-API addon, a Python
 * process that imports the binding first.  A process that initialised HIP earlier sets the variable itself; nbls_pool_init warns on stderr when `depth` exceeds the value in force.
 * nbls_hw_queues: the value in the environment (0 = unset: the runtime's default of 4); *set_by_library (may be NULL) = 1 when the library put it there. */
int nbls_hw_queues(int* set_by_library);
typedef struct nbls_pool nbls_pool;
int nbls_pool_init(int device_id, int depth /* 1 .. 64 */, nbls_pool** out);
void nbls_pool_destroy(nbls_pool* p);
int nbls_pool_depth(const nbls_pool* p);
nbls_ctx* nbls_pool_context(nbls_pool* p, int i);
int nbls_pool_next_slot(nbls_pool* p);
int nbls_pool_pairing_batch_dev(nbls_pool* p, size_t n, const void* d_g1, const void* d_g2, int with_final_exp, void* d_out, int* slot);
int nbls_pool_synchronize(nbls_pool* p);   /* everything submitted so far is complete (synchronises the device) */

/* Several GPUs of one node behind one handle (one context, host thread and stream per device; contiguous shards).  n_devices = 0 takes every
 * visible device; device_ids = NULL means 0 .. n_devices-1.  Independent pairings need no exchange; the product paths reduce every shard
 * to a 576-byte Fp12 partial, gather the partials on the first device with hipMemcpyPeer (xGMI) and run ONE shared final exponentiation
 * there.  Same arguments, results and error behaviour as the single-device calls of the same name. */
typedef struct nbls_multi nbls_multi;
int nbls_init_multi(int n_devices, const int* device_ids, nbls_multi** out);
void nbls_destroy_multi(nbls_multi* m);
int nbls_multi_device_count(const nbls_multi* m);
nbls_ctx* nbls_multi_context(nbls_multi* m, int i);
int nbls_multi_peer_access(const nbls_multi* m, int i);   /* 1: copies between device i and the reducing (first listed) device go peer to peer over xGMI; 0: staged by the runtime; -1: bad index */
int nbls_multi_pairing_batch(nbls_multi* m, size_t n, const uint8_t* g1_aff, const uint8_t* g2_aff, int with_final_exp, int validate,
                             uint8_t* out_fp12, int8_t* status);
int nbls_multi_miller_product(nbls_multi* m, size_t n, const uint8_t* g1_aff, const uint8_t* g2_aff, int final_exp, int validate,
                              uint8_t* out_fp12, int8_t* status);
int nbls_multi_verify_batch(nbls_multi* m, size_t n, const uint8_t* sig96, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* pk48,
                            const uint8_t* dst, size_t dst_len, int* ok);

/* Introspection for the benchmark / tests. */
int nbls_program_stats(nbls_ctx* ctx, int prog, uint32_t* out8);   /* steps, mul_steps, lin_steps, mul_ops, lin_ops, lin_terms, slots, lds_bytes */
int nbls_device_synchronize(nbls_ctx* ctx);
/* Placement study (tools/placement.py): runs one step program on n scratch items; out_blocks[5b..5b+4] = HW_ID | XCC_ID << 32, start tick, end tick (s_memtime), start, end time (s_memrealtime, 100 MHz) of workgroup b. */
int nbls_placement_probe(nbls_ctx* ctx, size_t n, uint64_t* out_blocks);
/* Per-kernel HIP-event timing (benchmark roofline leg): ms[i]/counts[i] for program i, last entry = inversion kernel. */
#define NBLS_N_PROGRAMS 128
/* Tuning knobs of a context (defaults are the measured optimum; tests use them to force a code path).
 * NBLS_TUNE_SPLIT_MILLER_MIN: number of pairs from which the Miller loop runs as two programs (line tables through HBM) instead of one. */
#define NBLS_TUNE_SPLIT_MILLER_MIN 1
#define NBLS_TUNE_HALVES_MIN 2         /* pairs from which nbls_pairing_batch_dev runs a batch as two halves on two streams (default 16384 since the end of round 6, 8192 before; 0 = never) */
#define NBLS_TUNE_EXPC_MIN 3           /* items from which the cyclotomic exponentiations of the final exponentiation use Karabina's compressed squarings
                                          (default: never -- 15 % fewer instructions but no faster as measured, see csrc/pipelines_pairing.cpp expx; 0 = always) */
#define NBLS_TUNE_CHAIN_MAX 4          /* items below which EXPX, FE_MID1, EXPX x 3, FE_MID2, EXPX of a final exponentiation are ONE launch (default 8192; 0 = seven launches:
                                          what a caller that keeps several calls in flight on other contexts wants, csrc/runtime.cpp run_chain) */
#define NBLS_TUNE_VERIFY_CHUNKS 5      /* verifyBatch as concurrent sub-batches (csrc/pipelines_verify.cpp verify_pipeline): number of sub-batches the signatures are cut into (default 2; 0 or 1 = one) */
#define NBLS_TUNE_VERIFY_LAST_PCT 6    /* ... size of the last sub-batch in per cent of the batch (default 25; the sizes fall linearly from the first to the last) */
#define NBLS_TUNE_VERIFY_PIPE_MIN 7    /* ... signatures from which a call is cut at all (default 32768) */
#define NBLS_TUNE_SAC_MAX 8            /* keys up to which sign's ladder (points known to lie in G2) uses the sign-aligned recoding with one addition per bit (default 6144: every
                                        * wavefront of the launch resident at once; 0 = never: the windowed psi-split ladder at every size) */
#define NBLS_TUNE_PT_LS2_MAX 9         /* items up to which the G2 point chains of verify / sign (clearCofactor's two ladders, sign's ladder) run in their two-lane forms (default 4096; 0 = never) */
#define NBLS_TUNE_WIDE_MAX 10          /* round 6, experiment: items up to which the programs that allow it run on the one-limb-per-lane interpreter (one item per workgroup of three wavefronts, two barriers per step); bit-exact, measured slower than the four-lane forms (0.93 against 0.66 ms for a final exponentiation's five exponentiations), so the default is 0 = never */
#define NBLS_TUNE_H2C_NORM_MIN 11     /* round 6: messages from which hash-to-G2 (also inside sign / verify / verifyBatch) takes the square root of its SWU map by the norm method -- two Fp exponentiations and a short program between them instead of one Fp2 exponentiation of twice the work; the same points; less work but two dependent exponentiations, so it pays where the device is full or the chain runs beside other work: the size compared is the whole call's (default 32768; 0 = always) */
#define NBLS_TUNE_INV_WIDE_MAX 12     /* round 6: elements up to which an Fp inversion launch runs with one limb per lane, four elements per wavefront (shorter for ONE call, eight times the instructions per element: default 4096; nbls_pool_init sets 256 on its contexts; 0 = never) */
#define NBLS_TUNE_LS_MAX 13           /* items up to which the pairing programs run in their four-lane forms (default 1024; nbls_pool_init sets 0 on its contexts: the forms shorten ONE call at up to four times the instructions per item) */
#define NBLS_TUNE_LS2_MAX 14          /* ... in their two-lane forms, above LS_MAX (default 2048; pool contexts 0) */
int nbls_set_tuning(nbls_ctx* ctx, int key, long long value);
int nbls_program_count(void);                 /* number of step programs; timing slot nbls_program_count() = the inversion kernel */
const char* nbls_program_name(int prog);
/* the kernel that executes program `prog` in this context: "nbls_aot_<name>" (ahead-of-time specialised, the product path) or "nbls_vm_kernel[_ls4]" (the interpreter:
   NBLS_AOT=0, or the build-time and run-time compilations of the program disagree); NULL on a bad index.  The string is static. */
const char* nbls_program_kernel(nbls_ctx* ctx, int prog);
int nbls_timing_enable(nbls_ctx* ctx, int on);
int nbls_timing_read(nbls_ctx* ctx, float* ms /*[NBLS_N_PROGRAMS+1]*/, uint32_t* counts /*[NBLS_N_PROGRAMS+1]*/);

#ifdef __cplusplus
}
#endif
#endif
