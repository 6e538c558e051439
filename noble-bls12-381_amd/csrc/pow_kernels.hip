// pow_kernels.hip -- serial per-element routines with one work item per LANE (the wave VM would leave 63 lanes idle on
// a single dependent chain): the Fp inversion of the final exponentiation (see fp_inv.h).
#include <hip/hip_runtime.h>
#include "vm_exec.h"
#include "consts_gen.h"
#include "fp_inv.h"

namespace nbls {

// out[i] = in[i]^-1 on raw Montgomery limbs (12 words per element), values in [0,2p) in and out.
extern "C" __global__ void __launch_bounds__(64) nbls_fp_inv_kernel(unsigned n, const u32* __restrict__ in, u32* __restrict__ out, const u32* __restrict__ table) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 x[12], y[12];
#pragma unroll
  for (int k = 0; k < 12; k++) x[k] = in[12 * i + k];
  fp_mont_inverse(y, x, table);
#pragma unroll
  for (int k = 0; k < 12; k++) out[12 * i + k] = y[k];
}

}  // namespace nbls

extern "C" int nbls_fp_inv_launch(unsigned n, const void* in, void* out, const void* table, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(nbls::nbls_fp_inv_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const nbls::u32*)in, (nbls::u32*)out, (const nbls::u32*)table);
  return (int)hipGetLastError();
}
