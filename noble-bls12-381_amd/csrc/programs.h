// programs.h -- registry of compiled step programs (see programs.cpp for buffer conventions).
#pragma once
#include "trace.h"
namespace nbls {
enum ProgId {
  P_MILLER_BYTES = 0,  // (G1, G2) -> conj-Miller value as wire bytes          [pairing(P, Q, false)]
  P_MILLER_RAW,        // (G1, G2) -> F
  P_MILLER_FE,         // (G1, G2) -> F, N = norm to invert
  P_NORM_RAW,          // F -> N
  P_NORM_BYTES,        // Fp12 wire bytes -> F, N
  P_FE_EASY,           // F, N^-1 -> t1 = f^((p^6-1)(p^2+1))                      (math.ts:859-861)
  P_EXPX,              // A -> conj(A^|x|) for unitary A (cyclotomicExp + conjugate) (math.ts:845-852, 862)
  P_FE_MID1,           // A, B -> conj(cyclotomicSquare(A)) * B                    (math.ts:863)
  P_FE_MID2,           // A, B -> A * cyclotomicSquare(B)                          (math.ts:866)
  P_FE_FINAL,          // t1..t7 -> final product as wire bytes                    (math.ts:868-873)
  P_MUL2,              // F[2i] * F[2i+1] -> F'[i]
  P_RAW_TO_BYTES,      // F -> wire bytes
  P_G1_VALIDATE,       // G1 affine (buf 0) -> int8 status (buf 7): 0 ok, 2 not on curve, 3 not in subgroup   (index.ts:383-388)
  P_G2_VALIDATE,       // G2 affine (buf 1) -> int8 status (buf 7)                                           (index.ts:633-638)
  P_G1_DEC_A,          // 48 B compressed G1 (buf 0) -> x (3), x^3+4 (4)                                       (index.ts:301-310)
  P_G1_DEC_B,          // compressed (0), x (3), rhs (4), rhs^((p+1)/4) (5) -> affine bytes (6), status (7)     (index.ts:311-326)
  P_G2_DEC_A,          // 96 B compressed G2 signature (buf 0) -> x (3), x^3+b (4)                             (index.ts:500-515)
  P_G2_DEC_B,          // compressed (0), x (3), rhs (4), rhs^((p^2+7)/16) (5) -> affine bytes (6), status (7)  (index.ts:516-529)
  P_H2C_A,             // 256 uniform bytes (buf 0) -> u0,u1 (3), SWU exponentiation inputs (4)                 (index.ts:256-263, math.ts:1220-1241)
  P_H2C_B,             // u0,u1 (3), powers (5) -> SWU maps, sum on E', 3-isogeny: projective point on E2 (6)    (math.ts:1243-1266, index.ts:487-488)
  P_G1_TO_PROJ, P_G1_ADD2, P_G1_NORM, P_G1_TO_AFFINE,     // point sums: aggregatePublicKeys (index.ts:771-778)
  P_G2_TO_PROJ, P_G2_ADD2, P_G2_NORM, P_G2_TO_AFFINE,     // aggregateSignatures (index.ts:781-788), hash-to-G2 output
  P_T_SWU, P_T_ISO, P_T_CLEAR,   // test-only pieces of hash-to-G2 (unit parity against golden vectors): SWU map, 3-isogeny, cofactor clearing
  P_H2C_C,             // projective point (3) -> clearCofactor -> projective hash point (6), norm of Z (7)      (index.ts:489, 659-672)
  P_MILLER_RAW2,       // two (G1, G2) pairs per item -> raw Fp12 of millerLoop x millerLoop with a shared accumulator (one Fp12 squaring per bit)
  P_G1_COMPRESS, P_G2_COMPRESS,   // affine wire point (buf 0) -> 48 / 96 compressed bytes (buf 2)   (PointG1.toHex(true) index.ts:359-371, PointG2.toSignature 586-602)
  // G1 hash-to-curve / encode-to-curve and G2 encode-to-curve (index.ts:331-350, 491-497); "count" field elements per message
  P_H2C1_A, P_ENC1_A,   // 64 * count uniform bytes (buf 0) -> u (3), tv4 = gx1 gxd^3 (4)                              (math.ts:1272-1299)
  P_H2C1_B, P_ENC1_B,   // u (3), tv4^((p-3)/4) (5) -> SWU point(s), sum on E1', 11-isogeny: projective point on E1 (6)  (math.ts:1300-1313, 1327)
  P_G1_CLEAR,           // projective point (3) -> clearCofactor -> projective point (6), Z (7)                            (index.ts:401-405)
  P_ENC2_A, P_ENC2_B,   // G2 encodeToCurve: 128 uniform bytes (0) -> u (3), SWU exponentiation input (4) ; u (3), power (5) -> projective point on E2 (6)
  P_G1_MUL, P_G2_MUL,            // [k]P for per-item 256-bit scalars: point (buf 0 / 1), scalar 32 B (buf 2) -> projective (3), norm of Z (4)   (getPublicKey / sign, index.ts:738-752)
  // multi-scalar multiplication (bucket method, 12-bit windows; pipelines_codec.cpp dev_msm): points are raw projective (3 / 6 field elements)
  P_G1_ADD_AB, P_G2_ADD_AB,       // A[i] (buf 3) + B[i] (buf 4) -> buf 5 (may alias buf 3)
  P_G1_HORNER, P_G2_HORNER,       // T[0..11] of one window (buf 3) -> sum_t 2^t T[t] (buf 5)
  P_G1_SHIFTADD, P_G2_SHIFTADD,   // 2^12 * acc (buf 3) + S (buf 4) -> buf 5
  P_G1_MSM_PREP,                  // affine P (buf 0) -> P, [z^2]P = -phi(P) = (beta x, -y) as projective points (buf 3): scalars split in base z^2
  P_G2_MSM_PREP,                  // affine Q (buf 1) -> Q, [|z|]Q = -psi(Q), [z^2]Q = psi^2(Q), [|z|^3]Q = -psi^3(Q) (buf 3): scalars split in base |z|
  // Miller loop in two programs (round 2), the reference's own decomposition (calcPairingPrecomputes math.ts:1331-1371 + millerLoop 1373-1388):
  // the point chain of Q (little parallelism: many items per wavefront) writes its 68 line triples to HBM, the Fp12 accumulation (12 lanes
  // per item, no idle lane) reads them back.  A line table is NBLS_LINE_BYTES = 68 * 6 raw elements per point.
  P_LINES_PQ,          // G1 (buf 0), G2 (buf 1) -> lines (buf 3) with the G1 coordinates folded in: (c0, c1 * Px, c2 * Py)
  P_LINES_Q,           // G2 (buf 1) -> lines (buf 3) as calcPairingPrecomputes returns them (prepared Q: PointG2.pairingPrecomputes, index.ts:703-711)
  P_LINES_BYTES,       // one line triple per item: 6 raw elements (buf 3) -> c0 || c1 || c2 as 3 x 96 wire bytes (buf 2); a table is 68 items
  P_LINES_FROM_BYTES,  // the inverse (a caller hands a table back in wire form)
  P_ACC_BYTES,         // folded lines (buf 3) -> conj-Miller value as wire bytes (buf 2)             [pairing(P, Q, false)]
  P_ACC_RAW,           // folded lines (buf 3) -> F (buf 5)
  P_ACC_FE,            // folded lines (buf 3) -> F (buf 5), N = norm to invert (buf 4)
  P_ACC2_RAW,          // two folded line tables per item (buf 3) -> F (buf 5): one shared accumulator, one Fp12 squaring per bit for both
  P_ACC_Q,             // prepared lines (buf 3) + G1 (buf 0) -> F (buf 5)                           [PointG1.millerLoop, index.ts:452-454]
  // wire-format completeness (round 2): the remaining decoders of the reference and the uncompressed G2 byte order
  P_G2_DEC_A192, P_G2_DEC_B192,   // PointG2.fromSignature on 192 bytes (index.ts:500-530 with half = 96)
  P_G2_DEC_B_HEX,                 // PointG2.fromHex on 96 compressed bytes (index.ts:532-562): flag rules, root by the S bit, no subgroup check
  P_G1_FROM_RAW, P_G2_FROM_RAW,   // uncompressed 96 / 192 bytes (buf 0) -> canonical affine wire bytes (buf 6), status (buf 7)   (index.ts:317-321, 563-575)
  P_G2_SWAP,                      // x.c0 x.c1 y.c0 y.c1 <-> x.c1 x.c0 y.c1 y.c0 (buf 0 -> buf 2)   (PointG2.toHex(false), index.ts:622-629)
  P_H2C_C1, P_H2C_C2,             // PointG2.clearCofactor (index.ts:659-672) one program around each multiplication by x: projective P (3), v (6) -> base = t1 + v (6), t1 = -[x]P (3, over P) ;
                                  // base (3), t1 (4), u (5) -> projective result u - [x]base - t1 (6), norm of Z (7)
  P_ACC4_RAW,                     // four folded line tables per item (buf 3) -> F (buf 5): one Fp12 squaring per bit for four Miller loops
  // lane-split variants (Program::lsplit = 4: every K_DOT lane-op on four adjacent lanes, one item per wavefront) for launches of at most one wavefront
  // per SIMD: the same formulas, a third fewer instructions per wavefront
  P_MILLER_BYTES_LS, P_MILLER_RAW_LS, P_MILLER_FE_LS, P_EXPX_LS,
  // cyclotomic exponentiation with Karabina's compressed squarings (round 3): the same map as P_EXPX in three programs around one Fp inversion
  P_EXPC_SQ,           // A (buf 3) -> the compressed coordinates (g2, g3, g4, g5) of (3 A)^(2^k) for k = 16, 48, 57 (buf 5: 3 x 8 raw elements): 57 compressed squarings, 8 lanes per item
  P_EXPC_DEC_A,        // compressed powers (buf 3) -> product of the three |2 g2|^2 (buf 4: the element to invert), numerators times conj(g2), all-but-one products (and a third of them), the g1-free part of g0, zero flag (buf 5: 19 raw elements)
  P_EXPC_DEC_B,        // compressed powers (3), inverse (4), DEC_A scratch (6) -> conj(A^|x|) (buf 5), int8 status (buf 7): 1 = some g2 was zero, the item must be recomputed by P_EXPX
  P_ACC8_RAW,          // eight folded line tables per item (buf 3) -> F (buf 5): one Fp12 squaring per bit for eight Miller loops (round 4: Miller products of 131,072 pairs and more, acc8_min in nbls_internal.h)
  P_H2C_C0,            // clearCofactor, the part that does not depend on [x]P: projective P (3) -> v = psi(P) (6), u = psi^2(2P) - psi(P) - P (5), read back by P_H2C_C1 / C2 after their ladders
  P_H2C_B1,            // one SWU map per item (2 n items): t (buf 3: Fp2), its exponentiation (buf 5: Fp2) -> projective point on E2' (buf 6: 6 raw elements)
  P_H2C_B2,            // the two points of a message (buf 3: 12 raw elements) -> their sum mapped to E2 by the 3-isogeny (buf 6), index.ts:487-488
  P_G1_MUL_W3, P_G2_MUL_W3,     // the ladders with 3-bit windows (85 instead of 128 additions: a shorter instruction stream, but a table that limits the wavefronts per CU): launches of at most one wavefront per SIMD
  P_MUL2S,             // A (buf 3) * B (buf 4) -> buf 5 (may alias buf 3): one level of the IN-PLACE product tree (round 5, reduce_product in pipelines_pairing.cpp):
                       // level k multiplies F[i 2^(k+1)] by F[i 2^(k+1) + 2^k] into the former, so no level copies or pads anything
  P_G1_MUL_FIXED,      // [k]G1.BASE by a fixed-base table (buf 5, shared by every item: 86 windows x 7 raw projective points with the default G1_FIXED_WIN = 3), scalar 32 B (buf 2) -> projective (3), Z (4): getPublicKey without doublings (round 5, curve.h pt_mul_fixed_g1)
  P_G2_MUL_GLS,        // [k]Q for Q in G2: affine Q (buf 1), the four base-|z| digits of k as 4 x 32 B big-endian (buf 2) -> projective (3), norm of Z (4): sign's ladder with the scalar split along psi (round 5, codec.h pt_mul_gls_g2)
  // two-lane split (round 5): launches of 1025 .. 2048 items, TWO items per wavefront, every K_DOT lane-op on two adjacent lanes (one DPP stage sums the columns)
  P_MILLER_BYTES_LS2, P_MILLER_RAW_LS2, P_MILLER_FE_LS2, P_EXPX_LS2,
  P_G2_MUL_SAC,        // [k]Q for Q in G2 with the four digits recoded sign-aligned (scalar_split.h): raw projective Q (buf 1), 4 x 32 B big-endian (signs + correction flag, index bits 0..2)
                       // (buf 2) -> projective (3), norm of Z (4): ONE addition per bit from a table of eight (round 5, codec.h pt_mul_sac_g2): launches of at most 6144 keys (three workgroups per CU)
  // two-lane split of the G2 point chains (round 5): launches of at most 4096 items (four items per wavefront): the two ladders of clearCofactor and sign's ladder -- a single verify / sign is
  // a chain of one-item launches whose time is the length of their instruction streams
  P_H2C_C1_LS2, P_H2C_C2_LS2, P_G2_MUL_SAC_LS2,
  // hash-to-G2 with the SWU square root by the norm method (round 6, codec.h swu_norm_*): two Fp exponentiations instead of one in Fp2, for calls of NBLS_H2C_NORM_MIN messages and more
  P_H2C_NA,            // 256 uniform bytes (buf 0) -> t0, t1 (3), N(u conj(v)) of both maps (4: the first exponentiation's input), state (5: per map zt2, num, den, a = u conj(v), d = N(v): 9 raw elements of 16)
  P_H2C_NM,            // one map per item: t (3), state (4), n = N(a)^((p+1)/4) (5) -> num of the chosen x, a1 / 2, delta (6: 4 raw elements), delta d^3 (7: the second exponentiation's input)
  P_H2C_NB,            // one map per item: t (3), state (4), e = (delta d^3)^((p-3)/4) (5) -> projective point on E2' (6), what P_H2C_B1 produces
  P_COUNT
};
// |x| = 2^63 + 2^62 + 2^60 + 2^57 + 2^48 + 2^16: the compressed chain runs to 2^57 and its values at the set bits 16, 48, 57 are decompressed; the powers
// 2^60, 2^62, 2^63 follow from the decompressed 2^57 power by 3 + 2 + 1 plain squarings (cheaper than three more decompressions)
static const int EXPC_POWERS = 3, EXPC_TOP = 57;
static const int EXPC_SQ_ELEMS = 8 * EXPC_POWERS, EXPC_DEC_ELEMS = 6 * EXPC_POWERS + 1;   // raw elements per item of the two scratch areas
static const int N_LINES = 68;                 // 63 doubling steps + 5 addition steps (bits of |x|)
static const int LINE_ELEMS = 6 * N_LINES;     // raw field elements per line table
static const int MSM_WINDOW_BITS = 12;
const Program& get_program(ProgId id);
// Single tower operations as step programs (nbls_tower_op_batch, include/nbls.h: KATs of math.ts:223-273, 451-539, 601-688, 732-852 on the device), built on first use.
// field 1 / 2 / 6 / 12; op = NBLS_TOP_* ; part 0 = the operation (or the first half of an inversion: x -> element to invert), 1 = the second half of an inversion.
// Buffers: 0 a, 1 b, 2 c, 3 d (wire bytes, 48 * field each; b / c / d of a sparse product are Fp2), 7 out; inversions: 4 the Fp element to invert (raw), 5 its inverse (raw).
// Returns nullptr for a combination the reference does not have.
const Program* get_tower_program(int field, int op, int param, int part);
void print_stats(const Program& p);
}  // namespace nbls
