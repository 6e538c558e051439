// Prototype: throughput of a 12x32-bit CIOS Montgomery multiplication mod the BLS12-381 prime,
// one independent product chain per lane (the shape of a VM MUL step). Compares formulations.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__device__ __constant__ u32 P32[12] = {0xffffaaabu,0xb9feffffu,0xb153ffffu,0x1eabfffeu,0xf6b0f624u,0x6730d2a0u,
                                       0xf38512bfu,0x64774b84u,0x434bacd7u,0x4b1ba7b6u,0x397fe69au,0x1a0111eau};
#define N0INV 0xfffcfffdu

// Variant A: textbook CIOS, u64 temporaries, compiler picks instructions.
__device__ __forceinline__ void montmul_A(u32* __restrict__ r, const u32* a, const u32* b) {
  u32 t[14];
#pragma unroll
  for (int i = 0; i < 14; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) { u64 x = (u64)a[j] * b[i] + t[j] + c; t[j] = (u32)x; c = x >> 32; }
    u64 x = (u64)t[12] + c; t[12] = (u32)x; t[13] = (u32)(x >> 32);
    u32 m = t[0] * N0INV;
    c = ((u64)m * P32[0] + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 12; j++) { u64 y = (u64)m * P32[j] + t[j] + c; t[j - 1] = (u32)y; c = y >> 32; }
    x = (u64)t[12] + c; t[11] = (u32)x; t[12] = t[13] + (u32)(x >> 32);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) r[i] = t[i];
}

// Variant B: product-scanning with 64-bit column accumulators split lo/hi (no per-MAD carry adds):
// each column k keeps two u64 sums: Slo_k = sum of low halves... implemented as: acc (u64) += lo32(prod) ; acc_next += hi32(prod)
// -> uses mul_lo + mul_hi (2 mults per product) : likely worse; kept for measurement.
__device__ __forceinline__ void montmul_B(u32* __restrict__ r, const u32* a, const u32* b) {
  // operand scanning, row i: (t) += a*b[i] using mad_u64_u32 on even/odd interleave:
  // t kept as 13 words; we compute row products into two carry-free vectors E (even j) and O (odd j) then add.
  u32 t[14];
#pragma unroll
  for (int i = 0; i < 14; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    u64 e[6], o[6];
#pragma unroll
    for (int j = 0; j < 6; j++) { e[j] = (u64)a[2*j] * b[i]; o[j] = (u64)a[2*j+1] * b[i]; }
    // t += E  (words 0..11), t += O<<32 (words 1..12)
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      u64 x = (u64)t[2*j] + (u32)e[j] + c; t[2*j] = (u32)x; c = x >> 32;
      x = (u64)t[2*j+1] + (u32)(e[j] >> 32) + c; t[2*j+1] = (u32)x; c = x >> 32;
    }
    u64 x = (u64)t[12] + c; t[12] = (u32)x; t[13] = (u32)(x >> 32);
    c = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      u64 y = (u64)t[2*j+1] + (u32)o[j] + c; t[2*j+1] = (u32)y; c = y >> 32;
      y = (u64)t[2*j+2] + (u32)(o[j] >> 32) + c; t[2*j+2] = (u32)y; c = y >> 32;
    }
    t[13] += (u32)c;
    u32 m = t[0] * N0INV;
#pragma unroll
    for (int j = 0; j < 6; j++) { e[j] = (u64)P32[2*j] * m; o[j] = (u64)P32[2*j+1] * m; }
    c = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      u64 x2 = (u64)t[2*j] + (u32)e[j] + c; t[2*j] = (u32)x2; c = x2 >> 32;
      x2 = (u64)t[2*j+1] + (u32)(e[j] >> 32) + c; t[2*j+1] = (u32)x2; c = x2 >> 32;
    }
    x = (u64)t[12] + c; t[12] = (u32)x; t[13] += (u32)(x >> 32);
    c = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      u64 y = (u64)t[2*j+1] + (u32)o[j] + c; t[2*j+1] = (u32)y; c = y >> 32;
      y = (u64)t[2*j+2] + (u32)(o[j] >> 32) + c; t[2*j+2] = (u32)y; c = y >> 32;
    }
    t[13] += (u32)c;
#pragma unroll
    for (int j = 0; j < 13; j++) t[j] = t[j + 1];
    t[13] = 0;
  }
#pragma unroll
  for (int i = 0; i < 12; i++) r[i] = t[i];
}

template <int V>
__global__ void __launch_bounds__(256) k_mont(u32* out, const u32* in, int iters) {
  u32 a[12], b[12];
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 12; i++) { a[i] = in[i] ^ (tid * 0x01000193u); b[i] = in[12 + i] + tid; }
  a[11] &= 0x0fffffff; b[11] &= 0x0fffffff;
  for (int it = 0; it < iters; it++) {
    u32 r[12];
    if (V == 0) montmul_A(r, a, b); else montmul_B(r, a, b);
#pragma unroll
    for (int i = 0; i < 12; i++) { b[i] = a[i]; a[i] = r[i]; }
  }
  u32 s = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) s ^= a[i];
  out[tid] = s;
}

template <int V>
int run(const char* name, int wavesPerSimd, u32* d_out, u32* d_in) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = 256 * wavesPerSimd, iters = 2000;
  hipLaunchKernelGGL(k_mont<V>, dim3(blocks), dim3(256), 0, 0, d_out, d_in, 10);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mont<V>, dim3(blocks), dim3(256), 0, 0, d_out, d_in, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double total = (double)iters * blocks * 256.0;
  printf("%-10s waves/SIMD=%d %8.3f ms  %8.2f G montmul/s  (%.1f clk per wave-montmul per SIMD @2.4GHz)\n", name, wavesPerSimd, best,
         total / (best * 1e-3) / 1e9, (best * 1e-3) * 2.4e9 / ((double)iters * wavesPerSimd));
  return 0;
}
int main() {
  u32 *d_out, *d_in; CK(hipMalloc(&d_out, 256 * 8 * 256 * 4)); CK(hipMalloc(&d_in, 24 * 4));
  u32 h[24]; for (int i = 0; i < 24; i++) h[i] = 0x9e3779b9u * (i + 1);
  CK(hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice));
  for (int w : {1, 2, 4}) { run<0>("cios_A", w, d_out, d_in); run<1>("evenodd_B", w, d_out, d_in); }
  return 0;
}
