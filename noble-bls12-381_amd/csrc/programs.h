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
  P_COUNT
};
const Program& get_program(ProgId id);
void print_stats(const Program& p);
}  // namespace nbls
