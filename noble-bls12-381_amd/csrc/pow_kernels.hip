// pow_kernels.hip -- serial exponentiations with one work item per LANE (the wave VM would leave 63 lanes idle on a
// single dependent chain).  Used for the one Fp inversion of the final exponentiation (Fp.invert, math.ts:134-156:
// the reference uses extended Euclid; a^(p-2) is the same canonical field element).
#include <hip/hip_runtime.h>
#include "vm_exec.h"
#include "consts_gen.h"

namespace nbls {

__constant__ uint64_t c_exp_pm2[6] = {NBLS_EXP_P_MINUS_2[0], NBLS_EXP_P_MINUS_2[1], NBLS_EXP_P_MINUS_2[2], NBLS_EXP_P_MINUS_2[3], NBLS_EXP_P_MINUS_2[4], NBLS_EXP_P_MINUS_2[5]};

// out[i] = in[i]^(p-2); raw Montgomery limbs (12 words per element), values in [0,2p) in and out.
extern "C" __global__ void __launch_bounds__(64) nbls_fp_inv_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out) {
  const u32 P2[12] = NBLS_2P32;
  const u32 R1[12] = {NBLS_R1[0], NBLS_R1[1], NBLS_R1[2], NBLS_R1[3], NBLS_R1[4], NBLS_R1[5], NBLS_R1[6], NBLS_R1[7], NBLS_R1[8], NBLS_R1[9], NBLS_R1[10], NBLS_R1[11]};
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 x[12], acc[12], t[12];
#pragma unroll
  for (int k = 0; k < 12; k++) { x[k] = in[12 * i + k]; acc[k] = R1[k]; }
  for (int b = NBLS_P_MINUS_2_BITS - 1; b >= 0; b--) {
    mont_mul12(t, acc, acc); csub<12>(t, P2);
    unsigned bit = (unsigned)((c_exp_pm2[b >> 6] >> (b & 63)) & 1);   // uniform
    if (bit) { mont_mul12(acc, t, x); csub<12>(acc, P2); }
    else {
#pragma unroll
      for (int k = 0; k < 12; k++) acc[k] = t[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 12; k++) out[12 * i + k] = acc[k];
}

}  // namespace nbls

extern "C" int nbls_fp_inv_launch(unsigned n, const void* in, void* out, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(nbls::nbls_fp_inv_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out);
  return (int)hipGetLastError();
}
