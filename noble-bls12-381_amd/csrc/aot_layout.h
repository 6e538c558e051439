// aot_layout.h -- placement of a translated program's LDS slots (round 5).
//
// A translated program (aot.h) addresses LDS with absolute per-lane addresses, so WHERE a slot of an instance lives is free: any injective map of (instance, slot) to
// slot-sized places gives the same results.  The compiled programs number their slots by a linear scan over value lifetimes, and with that numbering half of the LDS cycles
// of every hot kernel are bank conflicts (profiles/round4_pmc_b65536.json: ACC_FE 0.52, EXPX 0.50, LINES_PQ 0.44): a wavefront's ds_read_b128 is served in four groups
// of sixteen lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- and two lanes of a group that read different addresses in the same 16-byte bank
// group ((address / 16) mod 16) cost an extra cycle.  The model below reproduces the measured fractions (EXPX 0.49, ACC_FE 0.52, LINES_PQ 0.38) and a search over
// per-instance slot permutations (simulated annealing, incremental cost) brings them down; the result is generated at build time (aot_gen -> aot_layout.inc) and applied by
// aot_translate.  Programs with replicated constants and programs without a table entry keep the compiled placement.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "aot.h"

namespace nbls {

struct AotLayout {
  std::vector<std::vector<uint16_t>> pos;   // pos[g][s]: place of slot s inside instance g's region (a permutation of 0 .. slots-1 per instance)
  bool empty() const { return pos.empty(); }
};
// LDS read cycles of one wavefront of the program under a placement (identity when empty) and the cycles the same reads take without any conflict
struct AotLdsCost { unsigned long cycles, floor; };
AotLdsCost aot_layout_cost(const Program& p, const AotLayout& l);
// search; deterministic for a given (program, iterations, seed)
AotLayout aot_layout_search(const Program& p, long iterations, unsigned seed);
// the generated table: a placement for the program, or nullptr.  Defined by aot_layout_table.cpp (libnbls.so, the simulator) or by a stub (aot_gen, which produces the table)
struct AotLayoutEntry { const char* name; uint32_t G, slots, nsteps, hash; const uint16_t* pos; };
uint32_t aot_program_hash(const Program& p);   // FNV-1a over the compiled steps and descriptors: a table row is valid for exactly the program it was searched on
const AotLayoutEntry* aot_layout_table(size_t* n);
const AotLayout* aot_layout_for(const Program& p);

}  // namespace nbls
