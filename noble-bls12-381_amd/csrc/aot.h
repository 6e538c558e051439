// aot.h -- ahead-of-time specialised kernels for the hot step programs (round 4).
//
// The generic kernel (vm_kernel.hip) interprets a step list: per step it decodes a 32-byte header on the scalar unit and branches on the
// kind, on the number of product rounds, on the operand shape of every operand of every round and on the number of post-added terms.
// The hot programs -- the cyclotomic exponentiation (math.ts:845-852), the Miller accumulation (math.ts:1376-1386) and the line computation
// (math.ts:1337-1368) -- use only a handful of distinct step SIGNATURES (10 / 12 / 18 of them; 62 of EXPX's 97 steps are ONE signature, the
// cyclotomic squaring).  At build time `aot_gen` (aot_gen.cpp: the host compiler run over the programs listed below) writes the signature table
// of every listed program to aot_sigs.inc, and aot_kernel.hip instantiates, per program, ONE kernel whose step loop is a jump over that table:
// every signature is a straight-line body with the rounds unrolled and the shapes, flags and term counts resolved at compile time.
//
// First measurement (round 4): specialising the interpreter's own step code this way changes nothing for EXPX and ACC_FE (1132 against ~1170 VALU
// instructions per squaring step, same time): the interpreter's dispatch was never the cost.  What a specialised body CAN do, because it is not shared with
// other shapes, is a cheaper formulation of the work around the multiply-adds (aot_exec.h):
//   * descriptors hold ABSOLUTE LDS addresses per physical lane (no address arithmetic), idle lanes get a descriptor that reads the zero constant and writes
//     a junk slot (no lane predicate);
//   * the finish of a K_DOT lane-op stays in the 64-bit columns: multiplier as a shift, post-added terms as one multiply-add per limb with a per-lane signed
//     coefficient (t -+ 2 g is ONE term), the bias offs * p and the weak reduction's - q p as ONE multiply-add pass with the coefficient offs - q, q estimated
//     from the top two columns (no table, no global load) -- and ONE carry pass at the end instead of two.
// A translated program is a list of (signature id, descriptor block); a program whose steps are not all in its kernel's table (a build / environment mismatch)
// stays on the interpreter.
#pragma once
#include <string>
#include <vector>
#include "vm.h"
#include "programs.h"

// Ahead-of-time kernels: X(kernel name, up to four ProgIds it serves; P_COUNT = none).  A kernel's signature table is the union over its programs; programs that
// share a kernel (and the lanes per item) can run as a CHAIN -- one launch that executes them back to back for the same items, values passing through the same
// HBM scratch as between separate launches but without a grid-wide boundary: the seven launches EXPX, FE_MID1, EXPX x 3, FE_MID2, EXPX in the middle of a final
// exponentiation (math.ts:862-867) are one.
#define NBLS_AOT_PARTS 8   /* aot_kernel.hip is compiled once per part (Makefile): X's first argument is the part that holds the kernel's device code */
#define NBLS_AOT_KERNELS(X)                                                  \
  X(0, expx, P_EXPX, P_FE_MID1, P_FE_MID2, P_COUNT)                          \
  X(0, acc_fe, P_ACC_FE, P_ACC_RAW, P_ACC_BYTES, P_COUNT)                    \
  X(0, lines_pq, P_LINES_PQ, P_COUNT, P_COUNT, P_COUNT)                      \
  X(0, acc4_raw, P_ACC4_RAW, P_ACC8_RAW, P_ACC2_RAW, P_COUNT)                \
  X(1, fe_easy, P_FE_EASY, P_COUNT, P_COUNT, P_COUNT)                        \
  X(1, fe_final, P_FE_FINAL, P_COUNT, P_COUNT, P_COUNT)                      \
  X(1, miller_fe, P_MILLER_FE, P_MILLER_RAW, P_MILLER_BYTES, P_COUNT)        \
  X(1, mul2, P_MUL2, P_MUL2S, P_COUNT, P_COUNT)                             \
  X(1, norm, P_NORM_RAW, P_NORM_BYTES, P_RAW_TO_BYTES, P_COUNT)              \
  X(2, h2c_a, P_H2C_A, P_COUNT, P_COUNT, P_COUNT)                            \
  X(2, h2c_b, P_H2C_B1, P_H2C_B2, P_COUNT, P_COUNT)                          \
  X(2, h2c_c1, P_H2C_C1, P_H2C_C0, P_COUNT, P_COUNT)                         \
  X(2, h2c_c2, P_H2C_C2, P_COUNT, P_COUNT, P_COUNT)                          \
  X(2, g1_dec, P_G1_DEC_A, P_G1_DEC_B, P_COUNT, P_COUNT)                     \
  X(2, g2_dec, P_G2_DEC_A, P_G2_DEC_B, P_COUNT, P_COUNT)                     \
  X(2, g2_to_affine, P_G2_TO_AFFINE, P_G2_NORM, P_G2_TO_PROJ, P_G2_ADD2)     \
  X(3, g2_mul, P_G2_MUL, P_G2_MUL_W3, P_G2_MUL_GLS, P_G2_MUL_SAC)         \
  X(3, g1_mul, P_G1_MUL, P_G1_MUL_W3, P_G1_MUL_FIXED, P_COUNT)               \
  X(4, g1_validate, P_G1_VALIDATE, P_COUNT, P_COUNT, P_COUNT)                \
  X(4, g2_validate, P_G2_VALIDATE, P_COUNT, P_COUNT, P_COUNT)                \
  X(4, g1_sum, P_G1_TO_PROJ, P_G1_ADD2, P_G1_NORM, P_G1_TO_AFFINE)           \
  X(5, g1_msm, P_G1_ADD_AB, P_G1_HORNER, P_G1_SHIFTADD, P_G1_MSM_PREP)       \
  X(5, g2_msm, P_G2_ADD_AB, P_G2_HORNER, P_G2_SHIFTADD, P_G2_MSM_PREP)       \
  X(6, lines_q, P_LINES_Q, P_ACC_Q, P_LINES_BYTES, P_LINES_FROM_BYTES)       \
  X(6, miller_raw2, P_MILLER_RAW2, P_COUNT, P_COUNT, P_COUNT)                \
  X(6, g2_dec192, P_G2_DEC_A192, P_G2_DEC_B192, P_G2_DEC_B_HEX, P_G2_SWAP)   \
  X(6, from_raw, P_G1_FROM_RAW, P_G2_FROM_RAW, P_G1_COMPRESS, P_G2_COMPRESS) \
  X(7, h2c1, P_H2C1_A, P_H2C1_B, P_ENC1_A, P_ENC1_B)                         \
  X(7, g1_clear, P_G1_CLEAR, P_COUNT, P_COUNT, P_COUNT)                      \
  X(7, enc2, P_ENC2_A, P_ENC2_B, P_COUNT, P_COUNT)                            \
  X(4, h2c_n, P_H2C_NA, P_H2C_NM, P_H2C_NB, P_COUNT)

// Lane-split kernels (latency form for launches of at most one wavefront per SIMD): ONE item per wavefront, every K_DOT lane-op spread over four adjacent
// lanes that each accumulate a share of its products; the 28 columns are summed across the four lanes (two DPP stages) before the one reduction on the first of them.
#define NBLS_AOT_LS_KERNELS(X)                                                              \
  X(7, expx_ls, P_EXPX_LS, P_COUNT, P_COUNT, P_COUNT)                                      \
  X(3, miller_ls, P_MILLER_FE_LS, P_MILLER_RAW_LS, P_MILLER_BYTES_LS, P_COUNT)

// Two-lane split (round 5): launches of 1025 .. 2048 items -- at most one wavefront per SIMD with TWO items per wavefront --, every K_DOT lane-op on two adjacent lanes, ONE DPP
// stage.  Round 3 costed this form from the interpreter's numbers and left it; on the specialised kernels 2048 pairings take 1.9 instead of 2.17 ms (profiles/round5_ab_ls2.txt).
#define NBLS_AOT_LS2_KERNELS(X)                                                             \
  X(4, expx_ls2, P_EXPX_LS2, P_COUNT, P_COUNT, P_COUNT)                                    \
  X(5, miller_ls2, P_MILLER_FE_LS2, P_MILLER_RAW_LS2, P_MILLER_BYTES_LS2, P_COUNT)                    \
  X(6, g2pt_ls2, P_H2C_C1_LS2, P_H2C_C2_LS2, P_G2_MUL_SAC_LS2, P_COUNT)

namespace nbls {

// K_DOT flags of a signature
static const uint32_t AF_MULTSH = 1;   // some lane multiplies the reduced sum by 2 or 4 (a shift of the columns)
static const uint32_t AF_MULT3 = 2;    // some lane multiplies by 3
static const uint32_t AF_OFFS = 4;     // some lane adds a multiple of p (keeps a sum with negative products non-negative)
static const uint32_t AF_WRED = 8;     // weak reduction after the post-added terms
static const uint32_t AF_HALVE = 16;   // some lane halves its result
// Signature of a step.  K_DOT: p0 = product rounds, flags = AF_*, t = post-added terms after merging equal slots (max over the lanes), sh0 / sh1 = round shapes
// (vm.h SH_*).  K_LIN: p0 = added terms, t = subtracted terms, flags = AF_WRED | AF_HALVE.  K_LOAD: p0 = bytes.  K_STORE: p0 = 1 raw.
struct AotSig { uint32_t kind, p0, flags, t, sh0, sh1; };
static inline bool operator==(const AotSig& a, const AotSig& b) { return a.kind == b.kind && a.p0 == b.p0 && a.flags == b.flags && a.t == b.t && a.sh0 == b.sh0 && a.sh1 == b.sh1; }
// one step of a translated program (scalar loads): x = signature id | 16-byte words per lane << 8 ; y = index (in 16-byte words) of the step's descriptor block.
// A block is laid out [word][64 lanes]: a wavefront's load of word w is one contiguous kilobyte.
struct AotStep { uint32_t x, y; };
// Lane descriptor words (absolute LDS byte addresses):
//   K_DOT : w0 = dst | (m >> 1) << 16 (2 bits) | (m == 3) << 18 | halve << 19 | offs << 20 (4 bits) ; w1 = per-lane term signs (4 bits per round, shape mode 3) ;
//           then t post-added terms: address | coefficient << 16 (signed 16 bits), the header padded to whole 16-byte words ; then one 16-byte word per
//           product round: a0, a1, b0, b1
//   K_LIN : w0 = dst | halve << 16 ; then p0 + t term addresses, two 16-bit fields per word (added ones first)
//   K_LOAD / K_LOADW / K_STORE / K_STOREW : w0 = slot | buffer << 16 | active << 31 ; w1 = byte offset inside the item
//   every other kind : the interpreter's descriptor (vm.h) with its 16-bit fields made absolute; K_STATUS: bit 31 of w0 = active
static const int AOT_DOT_HDR = 2;
static inline uint32_t aot_dot_hdr_quads(uint32_t t) { return (AOT_DOT_HDR + t + 3) / 4; }   // 16-byte words of a K_DOT descriptor before its rounds
struct AotProgram {
  std::vector<AotSig> sigs;          // distinct signatures in order of first use
  std::vector<uint32_t> sig_count;   // steps per signature
  std::vector<AotStep> steps;        // x holds an index into `sigs` until remapped to a kernel's table
  std::vector<uint32_t> descs;       // 4 words per 16-byte word
  uint32_t lds_bytes = 0;            // the program's LDS image + the junk slot idle lanes write to
};
// Launch arguments of an ahead-of-time kernel: up to AOT_MAX_SEGS programs executed back to back by every wavefront for its items (a chain; one segment = one
// program with its own constants, LDS layout and buffer bindings; all segments share the lanes per item).
static const int AOT_MAX_SEGS = 8;
struct AotSeg {
  const AotStep* steps; const uint32_t* descs; const uint32_t* consts;
  uint32_t nsteps, nconst, inst_bytes, slot_bytes, shared_consts, pad;
  IOBuf bufs[MAX_BUFS];
};
struct AotArgs {
  AotSeg seg[AOT_MAX_SEGS];
  uint32_t nseg, W, G, n_items, fair, pad;
  const uint32_t* qp_table; const uint32_t* item_index; const uint32_t* n_items_dev;
};
// Translate a compiled program.  Returns an empty string, or why the program cannot run on an ahead-of-time kernel (lane split, a step kind the kernels do not implement).
// aot_translate places the LDS slots as the generated table says (aot_layout.h; NBLS_LDS_LAYOUT=0: as compiled); aot_translate_with takes the placement (nullptr: as compiled).
struct AotLayout;
std::string aot_translate(const Program& p, AotProgram& out);
std::string aot_translate_with(const Program& p, AotProgram& out, const AotLayout* layout);

}  // namespace nbls

// kernel side (aot_kernel.hip)
extern "C" const char* nbls_aot_name(int k);   // "nbls_aot_<name>" of kernel k
extern "C" int nbls_aot_index(int prog_id);   // index of the ahead-of-time kernel that serves a ProgId, or -1
// remap the signature indices of a translated program to kernel k's table: 0, or -1 when a signature is not in the table
extern "C" int nbls_aot_bind(int k, nbls::AotProgram* ap);
extern "C" int nbls_aot_launch(int k, const nbls::AotArgs* a, unsigned lds_bytes, void* stream);
