// vm.h -- program format of the wavefront Fp engine ("wave VM").
//
// Design: the pairing path is straight-line arithmetic over Fp with a fixed control flow
// (the bits of the BLS parameter x).  The host traces the tower formulas symbolically (trace.h), schedules the
// resulting DAG of Fp operations into wave-wide STEPS (<= W lane-operations each, one kind per step) and
// allocates LDS slots.  One wavefront executes the step list for G = 64/W work items ("instances") at once:
// every lane performs one lane-operation per step -- typically a K_DOT: a sum of up to 8 products of 381-bit values with
// one Montgomery reduction -- operands and results live in LDS slots (64 B each: 14 limbs of 28 bits + padding),
// Montgomery radix R = 2^392; stored values are non-negative, normalised (limbs < 2^28) and small multiples of p -- the
// 11 bits of headroom make conditional subtractions unnecessary except when a canonical representative is required.
// Within one wavefront LDS operations execute in order, so no barriers are needed anywhere.
#pragma once
#include <stdint.h>

namespace nbls {

enum StepKind : uint8_t {
  K_LOAD = 0,    // slot <- p0 (0 = 48) big-endian bytes of an input buffer (raw integer, < 2^384)
  K_MUL = 1,     // (retired: the first engine of this round; K_DOT subsumes it)
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
                 //         every A_i, B_i is x, -x, x + y or x - y as signed limbs.  Subsumes Fp2/Fp6/Fp12 recombination
};

struct Step {
  uint8_t kind;
  uint8_t nlanes;     // active lanes per instance (<= W)
  uint8_t p0, p1;     // kind-specific (LIN: p0 = max terms over the step's lanes; DOT: p0 = max products, pad = max linear terms;
                      //                LOAD: p0 = bytes (0 = 48); CMP / FLAG: p0 = predicate)
  uint32_t desc_off;  // word offset of this step's descriptors
  uint32_t stride;    // words per lane descriptor
  uint32_t pad;
};

// operand encoding (16 bit): bits 0..12 slot index, bit 13 = constant region, bits 14..15 = mode
static const uint32_t OP_SLOT_MASK = 0x1fff;
static const uint32_t OP_CONST = 0x2000;
static const uint32_t OP_MODE_SHIFT = 14;   // LIN / DOT linear term: bit 14 = negative
static const int MAX_LIN_TERMS = 7;    // signed limb-wise sums must stay inside (-2^31, 2^31): 7 x 2^28
static const int SLOT_WORDS = 16;      // 14 limbs + 2 padding words (16-byte aligned LDS / HBM scratch elements)
static const int NLIMBS = 14;
static const int RAW_FP_BYTES = 64;    // one raw field element in HBM scratch
static const int MAX_BUFS = 8;
// K_DOT lane descriptor: w0 = dst | k<<16 | L<<20 | m<<24 | halve<<27 | offs<<28 ; w1 reserved ; w2,w3 = 4 linear terms (u16: slot|const|neg<<14)
//   then per product 2 words: (a0 | a1<<16), (b0 | b1<<16); first term u16 = slot | const<<13 | negate-operand<<14 | normalise<<15, second term u16 = slot | const<<13 | subtract<<14 | present<<15
static const int MAX_DOT_PRODUCTS = 8;
static const int MAX_DOT_LINEAR = 4;
static const uint32_t OP_NEG = 0x4000;
static const uint32_t OP_PRESENT = 0x8000;   // on the second term of a product operand: term present
static const uint32_t OP_NORM = 0x8000;      // on the first term: normalise the (sum) operand's limbs before multiplying

struct IOBuf { uint8_t* ptr; uint64_t stride; };

struct KernelArgs {
  const Step* steps;
  const uint32_t* descs;
  const uint32_t* consts;   // nconst * SLOT_WORDS words
  uint32_t nsteps, nconst;
  uint32_t W, G;            // lanes per instance, instances per wave (G * W <= 64)
  uint32_t slots;           // LDS slots per instance
  uint32_t n_items;
  IOBuf bufs[MAX_BUFS];
  const uint32_t* item_index;    // optional (NULL): buffers are addressed with item_index[item] instead of item (gathered operands, results scattered back in place)
  const uint32_t* n_items_dev;   // optional (NULL): the item count lives in device memory (min with n_items, which then only sizes the launch)
  uint64_t* hwid_out;       // optional (NULL): per workgroup, HW_ID | XCC_ID << 32 of its wavefront -- placement studies (tools/placement.py)
};

static inline uint32_t lds_words(uint32_t nconst, uint32_t G, uint32_t slots) { return nconst * SLOT_WORDS + G * slots * SLOT_WORDS; }

}  // namespace nbls
