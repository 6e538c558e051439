#include <hip/hip_runtime.h>
typedef unsigned long long u64; typedef unsigned int u32;
__device__ __constant__ u32 P32[12] = {0xffffaaabu,0xb9feffffu,0xb153ffffu,0x1eabfffeu,0xf6b0f624u,0x6730d2a0u,
                                       0xf38512bfu,0x64774b84u,0x434bacd7u,0x4b1ba7b6u,0x397fe69au,0x1a0111eau};
#define N0INV 0xfffcfffdu
// Variant C: row products via mad(a_j,b_i,t_j) (32-bit addend zero-extended), then one 32-bit carry chain per row.
__device__ __forceinline__ void montmul_C(u32* __restrict__ r, const u32* a, const u32* b) {
  u32 t[13];
#pragma unroll
  for (int i = 0; i < 13; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    u32 lo[12], hi[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { u64 x = (u64)a[j] * b[i] + t[j]; lo[j] = (u32)x; hi[j] = (u32)(x >> 32); }
    // t = lo + (hi << 32) + t[12]<<(12*32)
    u32 c = 0, s[13];
    s[0] = lo[0];
#pragma unroll
    for (int j = 1; j < 12; j++) { u32 co; s[j] = __builtin_addc(lo[j], hi[j - 1], c, &co); c = co; }
    { u32 co; s[12] = __builtin_addc(t[12], hi[11], c, &co); c = co; }
    u32 top = c;
    u32 m = s[0] * N0INV;
#pragma unroll
    for (int j = 0; j < 12; j++) { u64 x = (u64)m * P32[j] + s[j]; lo[j] = (u32)x; hi[j] = (u32)(x >> 32); }
    c = 0;
#pragma unroll
    for (int j = 1; j < 12; j++) { u32 co; t[j - 1] = __builtin_addc(lo[j], hi[j - 1], c, &co); c = co; }
    { u32 co; t[11] = __builtin_addc(s[12], hi[11], c, &co); c = co; }
    t[12] = top + c;
  }
#pragma unroll
  for (int i = 0; i < 12; i++) r[i] = t[i];
}
__global__ void __launch_bounds__(256) k_mont(u32* out, const u32* in, int iters) {
  u32 a[12], b[12];
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 12; i++) { a[i] = in[i] ^ (tid * 0x01000193u); b[i] = in[12 + i] + tid; }
  for (int it = 0; it < iters; it++) {
    u32 r[12];
    montmul_C(r, a, b);
#pragma unroll
    for (int i = 0; i < 12; i++) { b[i] = a[i]; a[i] = r[i]; }
  }
  u32 s = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) s ^= a[i];
  out[tid] = s;
}
