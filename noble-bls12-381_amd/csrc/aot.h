// aot.h -- ahead-of-time specialised kernels for the hot step programs (round 4).
//
// The generic kernel (vm_kernel.hip) interprets a step list: per step it decodes a 32-byte header on the scalar unit and branches on the
// kind, on the number of product rounds, on the operand shape of every operand of every round and on the number of post-added terms
// (~59 taken branches per step).  The hot programs -- the cyclotomic exponentiation (math.ts:845-852), the Miller accumulation
// (math.ts:1376-1386) and the line computation (math.ts:1337-1368) -- use only a handful of distinct step SIGNATURES (kind, rounds,
// flags, shapes, post-added terms: 10 / 12 / 18 of them; 62 of EXPX's 97 steps are ONE signature, the cyclotomic squaring).  At build time
// `aot_gen` (aot_gen.cpp: the host compiler run over the programs listed below) writes the signature table of every listed program to
// aot_sigs.inc, and aot_kernel.hip instantiates, per program, ONE kernel whose step loop is a jump over that table: every signature is a
// straight-line body with the rounds unrolled, the shapes, flags and term counts resolved at compile time (no header decode, no shape
// dispatch, no round loop) and the same mac28 / redc28 arithmetic as the interpreter (vm_exec.h), so results are bit-identical.
// At run time a program's step list is translated into (signature id, active lanes, descriptor offset) words; a program whose steps are
// not all in its kernel's table (a build / environment mismatch) falls back to the interpreter.
#pragma once
#include "vm.h"
#include "programs.h"

// programs with an ahead-of-time kernel: X(name, ProgId)
#define NBLS_AOT_PROGRAMS(X) \
  X(expx, P_EXPX)            \
  X(acc_fe, P_ACC_FE)        \
  X(lines_pq, P_LINES_PQ)    \
  X(acc4_raw, P_ACC4_RAW)

namespace nbls {

struct AotSig { uint32_t kind, p0, p1, lin, sh0, sh1, stride; };
static inline bool operator==(const AotSig& a, const AotSig& b) { return a.kind == b.kind && a.p0 == b.p0 && a.p1 == b.p1 && a.lin == b.lin && a.sh0 == b.sh0 && a.sh1 == b.sh1 && a.stride == b.stride; }
static inline AotSig aot_sig_of(const Step& st) { return AotSig{st.kind, st.p0, st.p1, st.lin, st.shape[0], st.shape[1], st.stride}; }
// one step of the translated list (scalar loads): x = signature id | active lanes << 8 | descriptor stride (words) << 16 ; y = word offset of the step's descriptors
struct AotStep { uint32_t x, y; };

}  // namespace nbls

// host side (aot_kernel.hip)
extern "C" int nbls_aot_index(int prog_id);   // index of the ahead-of-time kernel for a ProgId, or -1
// translate a compiled program for kernel `k`: fills `out` (one AotStep per step) and returns 0, or -1 when a step's signature is not in the kernel's table
extern "C" int nbls_aot_translate(int k, const nbls::Step* steps, unsigned nsteps, nbls::AotStep* out);
extern "C" int nbls_aot_launch(int k, const nbls::KernelArgs* ka, unsigned lds_bytes, void* stream);
