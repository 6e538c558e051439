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
  P_FE_HARD,           // F, N^-1 -> finalExponentiate(F) as wire bytes
  P_MUL2,              // F[2i] * F[2i+1] -> F'[i]
  P_RAW_TO_BYTES,      // F -> wire bytes
  P_COUNT
};
const Program& get_program(ProgId id);
void print_stats(const Program& p);
}  // namespace nbls
