// vm.h -- program format of the wavefront Fp engine ("wave VM").
//
// Design: the pairing path is straight-line arithmetic over Fp with a fixed control flow
// (the bits of the BLS parameter x).  The host traces the tower formulas symbolically (trace.h), schedules the
// resulting DAG of Fp operations into wave-wide STEPS (<= W lane-operations each, one kind per step) and
// allocates LDS slots.  One wavefront executes the step list for G = 64/W work items ("instances") at once:
// every lane performs one lane-operation per step -- typically a K_DOT: a sum of up to 8 products of 381-bit values with
// one Montgomery reduction -- operands and results live in LDS slots (14 limbs of 28 bits), Montgomery radix R = 2^392;
// stored values are non-negative, normalised (limbs < 2^28) and small multiples of p -- the 11 bits of headroom make
// conditional subtractions unnecessary except when a canonical representative is required.
// Within one wavefront LDS operations execute in order, so no barriers are needed anywhere.
//
// Format, second generation (round 2).  Everything the kernel would otherwise decide per lane is decided by the host
// compiler and handed over in one of two forms:
//   * UNIFORM data in the step header (read with scalar loads, branched on with scalar branches): the number of product
//     rounds of a K_DOT step, the operand SHAPE of every round (single slot / sum / difference, normalise first), how
//     many post-added terms follow the reduction.  Lanes that need less are padded with the zero constant, so no lane
//     ever tests a flag of its own inside the product loop.
//   * PER-LANE data in the descriptors: LDS byte offsets (relative to the lane's instance region) of the slots to read
//     and write.  The program's constants are replicated at the start of every instance region, so an operand address
//     is one addition (programs with eight or more instances per wavefront keep a single shared copy instead, at three more instructions
//     per operand, because replication would cost them occupancy); the slot stride is a property of the program (80 bytes = conflict-free 16-byte reads when the
//     LDS budget allows it, 64 otherwise), not of the kernel.
#pragma once
#include <stdint.h>

namespace nbls {

enum StepKind : uint8_t {
  K_LOAD = 0,    // slot <- p0 (0 = 48) big-endian bytes of an input buffer (raw integer, < 2^384)
  K_LIN = 2,     // slot <- sum of up to 7 +-slots (a k*p constant term keeps it non-negative), optionally halved
  K_STORE = 3,   // 48 big-endian bytes of an output buffer <- canonical(slot)   (slot must hold value/R already); p0 = 1: the raw integer as is
  K_LOADW = 4,   // slot <- one raw element (14 limbs) of a scratch buffer
  K_STOREW = 5,  // scratch buffer <- the 14 limbs of slot
  K_ISZ = 6,     // slot <- (value == 0 mod p) ? 1 : 0   (raw integer flag)
  K_SEL = 7,     // slot <- flag ? a : b
  K_STATUS = 8,  // int8 status[item] <- first code whose flag is 0, else 0
  K_CANON = 9,   // slot <- canonical representative in [0,p)
  K_CMP = 10,    // flag slot <- predicate on raw integers: p0 = 0: a > b ; 1: a is odd
  K_FLAG = 11,   // flag slot <- boolean op of two flags: p0 = 0 and, 1 or, 2 xor, 3 and-not (a & !b)
  K_BIT = 13,    // flag slot <- bit (w1) of a raw integer slot            (wire-format flag bits, index.ts:305-314)
  K_BITAND = 14, // slot <- a & b (raw 384-bit integers)                   (value mod 2^381, index.ts:309)
  K_DOT = 12,    // slot <- m * (mont(sum_i A_i * B_i) + offs * p) +- up to 4 slots, ONE Montgomery reduction for the whole sum;
                 //         every A_i, B_i is x, x + y or x - y as signed limbs (-x is 0 - x).  Subsumes Fp2/Fp6/Fp12 recombination
};

// Step header: 32 bytes, uniform for the wavefront (scalar loads).
struct Step {
  uint8_t kind;
  uint8_t nlanes;     // active lanes per instance (<= W)
  uint8_t p0, p1;     // DOT: p0 = product rounds (max over the lanes), p1 = flags (DOTF_*); LIN: p0 = added terms, p1 = subtracted terms (max over the
                      //      lanes; a lane with fewer terms is padded with the zero constant); LOAD: p0 = bytes (0 = 48); STORE: p0 = 1 raw; CMP / FLAG: p0 = predicate
  uint32_t desc_off;  // word offset of this step's lane descriptors
  uint32_t stride;    // words per lane descriptor (multiple of 4)
  uint32_t lin;       // DOT: post-added terms after the reduction: bits 0..2 added, bits 4..6 subtracted (max over the lanes, zero-padded)
  uint32_t shape[2];  // DOT: 8 bits per product round (SH_*), rounds 0..3 in shape[0], 4..7 in shape[1]
  uint32_t rsv[2];
};
static_assert(sizeof(Step) == 32, "Step header is read as eight dwords");

// per-round operand shape (uniform for the wavefront): a lane whose operand has no second term adds / subtracts the zero constant
static const uint32_t SH_A_MODE = 0x03;     // 0 single slot, 1 x + y, 2 x - y, 3 per-lane signs +-x +- y (word 1 of the lane descriptor: 4 sign bits per round)
static const uint32_t SH_A_NORM = 0x04;     // normalise the (sum) operand's limbs before multiplying
static const uint32_t SH_B_SHIFT = 3;       // the same three bits for operand B
static const uint32_t DOTF_MULT = 1, DOTF_HALVE = 2, DOTF_OFFS = 4;   // some lane of the step has m > 1 / halves its result / has offs > 0
static const uint32_t DOTF_WRED = 8;     // weak reduction after the post-added terms: subtract q p with q estimated from the top limb (table lookup), result below 3.02 p
static const int QP_TABLE_ENTRIES = 128; // q p for q = 0 .. 127, 16 words each (14 normalised limbs + padding)

static const int MAX_LIN_TERMS = 7;    // signed limb-wise sums must stay inside (-2^31, 2^31): 7 x 2^28
static const int NLIMBS = 14;
static const int RAW_WORDS = 16;       // one raw field element in HBM scratch: 14 limbs + 2 padding words
static const int SLOT_WORDS = RAW_WORDS;   // (host-side arrays of raw elements)
static const int RAW_FP_BYTES = 64;
static const int MAX_BUFS = 8;
static const int MAX_DOT_PRODUCTS = 8;
static const int MAX_DOT_LINEAR = 4;
// Operand fields are LDS byte offsets (multiples of 16, below 64 KB) relative to the instance region; with shared constants bit 1 marks a slot, an offset without it is an absolute constant offset.
// K_DOT lane descriptor (words):  w0 = dst | m << 16 | halve << 19 | offs << 20 ;  w1 = per-lane term signs, 4 bits per round (a0, a1, b0, b1; mode 3) ;  w2, w3 reserved ;
//                                 w4..w7 = eight 16-bit offsets of post-added terms (added ones first, then subtracted ones, each group zero-padded
//                                          to the step's count) -- only read when the step has any ;
//                                 then per product round 4 words: a0, a1, b0, b1 (32-bit offsets; a1 / b1 = 0 when the round's shape has no second term)
// K_LIN lane descriptor:          w0 = dst | halve << 16 ; w1..w7 = 16-bit offsets, added terms first
static const int DOT_HDR_WORDS = 8;
static const int DOT_ROUND_WORDS = 4;

struct IOBuf { uint8_t* ptr; uint64_t stride; };

struct KernelArgs {
  const Step* steps;
  const uint32_t* descs;
  const uint32_t* consts;   // nconst * RAW_WORDS words
  const uint32_t* qp_table; // QP_TABLE_ENTRIES * RAW_WORDS words: multiples of p for the weak reduction (DOTF_WRED)
  uint32_t nsteps, nconst;
  uint32_t W, G;            // lanes per instance, instances per wave (G * W <= 64)
  uint32_t slot_bytes;      // LDS slot stride of this program (64 or 80)
  uint32_t inst_bytes;      // LDS bytes per instance region: (nconst + slots) * slot_bytes, or slots * slot_bytes with shared constants
  uint32_t n_items;
  uint32_t lsplit;          // 1, or 4: lane-split program (every K_DOT lane-op on four adjacent lanes; W counts physical lanes)
  uint32_t shared_consts;   // 0: constants replicated at the start of every instance region; 1: one copy at the start of the LDS image, the instance
                            //    regions behind it, constant operands marked by bit 1 of their offset (programs with 8 or more instances per wavefront)
  IOBuf bufs[MAX_BUFS];
  const uint32_t* item_index;    // optional (NULL): buffers are addressed with item_index[item] instead of item (gathered operands, results scattered back in place)
  const uint32_t* n_items_dev;   // optional (NULL): the item count lives in device memory (min with n_items, which then only sizes the launch)
  uint64_t* hwid_out;       // optional (NULL): per workgroup, HW_ID | XCC_ID << 32 of its wavefront -- placement studies (tools/placement.py)
  const void* aot_steps;    // ahead-of-time kernels (aot.h): the translated step list (AotStep per step); unused by the interpreter
  uint32_t fair;            // ahead-of-time kernels: 1 = lower the own priority with progress (the interpreter has separate _fair instantiations)
};

}  // namespace nbls
