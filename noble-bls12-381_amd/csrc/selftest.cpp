// selftest.cpp -- host-side self test of the wave VM's compiler and simulator for the sanitizer build (make debug -> nbls_selftest, built with
// -fsanitize=address,undefined).  TEST INFRASTRUCTURE (not part of libnbls.so): compiles every step program, verifies each one statically
// (verify_program), and runs the pairing of the two generators through the simulator -- Miller loop as one program and as LINES + ACC -- printing
// the first coefficient of the Miller value, which tests/test_debug_build.py compares with the reference's (SURVEY 8(c) anchor).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "programs.h"
#include "consts_gen.h"
#include "vm_exec.h"

extern "C" int nbls_sim_run(int prog, unsigned n_items, uint8_t** ptrs, const uint64_t* strides);
using namespace nbls;

static void be48(uint8_t* o, const u32* limbs) { u32 w[12]; limbs_to_words(w, limbs); for (int i = 0; i < 12; i++) { u32 v = w[11 - i]; o[4 * i] = v >> 24; o[4 * i + 1] = v >> 16; o[4 * i + 2] = v >> 8; o[4 * i + 3] = v; } }

int main() {
  int bad = 0;
  for (int i = 0; i < P_COUNT; i++) {
    const Program& p = get_program((ProgId)i);
    const std::string e = verify_program(p);
    if (!e.empty()) { printf("VERIFY FAILED %s\n", e.c_str()); bad++; }
  }
  printf("programs %d verified, failures %d\n", (int)P_COUNT, bad);
  // pairing(G1, G2, false): generators in wire form
  uint8_t g1[96], g2[192], out1[576], out2[576];
  be48(g1, NBLS_G1X_RAW); be48(g1 + 48, NBLS_G1Y_RAW);
  static const char* G2HEX = "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
                             "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
                             "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"
                             "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be";   // the G2 generator (CURVE.G2x, CURVE.G2y)
  for (int i = 0; i < 192; i++) { unsigned v; sscanf(G2HEX + 2 * i, "%2x", &v); g2[i] = (uint8_t)v; }
  {
    uint8_t* ptrs[8] = {g1, g2, out1, nullptr, nullptr, nullptr, nullptr, nullptr}; uint64_t strides[8] = {96, 192, 576, 0, 0, 0, 0, 0};
    if (nbls_sim_run(P_MILLER_BYTES, 1, ptrs, strides)) return 2;
  }
  {
    std::vector<uint8_t> lines((size_t)LINE_ELEMS * RAW_FP_BYTES);
    uint8_t* ptrs[8] = {g1, g2, out2, lines.data(), nullptr, nullptr, nullptr, nullptr}; uint64_t strides[8] = {96, 192, 576, lines.size(), 0, 0, 0, 0};
    if (nbls_sim_run(P_LINES_PQ, 1, ptrs, strides) || nbls_sim_run(P_ACC_BYTES, 1, ptrs, strides)) return 2;
  }
  if (memcmp(out1, out2, 576)) { printf("MISMATCH between the fused and the split Miller loop\n"); bad++; }
  printf("miller c0.c0.c0 ");
  for (int i = 0; i < 48; i++) printf("%02x", out1[i]);
  printf("\n");
  return bad ? 1 : 0;
}
