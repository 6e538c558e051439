// Microbenchmark: issue rates of the integer/fp64 multiply instructions that a 381-bit
// Montgomery multiplier can be built from, on gfx950. Output feeds DESIGN.md's roofline peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
__device__ inline u32 __umulhi24x(u32 a, u32 b){ u32 r; asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
#define ITERS 4096
#define UNR 16

__global__ void k_mad64(u32* out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  u64 acc[UNR];
  for (int j = 0; j < UNR; j++) acc[j] = a + j;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int j = 0; j < UNR; j++) acc[j] = (u64)(u32)acc[j] * b + acc[j];  // v_mad_u64_u32
  }
  u64 s = 0; for (int j = 0; j < UNR; j++) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}
__global__ void k_mullo(u32* out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = a | 1;
  u32 acc[UNR];
  for (int j = 0; j < UNR; j++) acc[j] = a + j;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int j = 0; j < UNR; j++) acc[j] = acc[j] * b;  // v_mul_lo_u32
  }
  u32 s = 0; for (int j = 0; j < UNR; j++) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi(u32* out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = a | 0x80000001u;
  u32 acc[UNR];
  for (int j = 0; j < UNR; j++) acc[j] = a + j * 77777u;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int j = 0; j < UNR; j++) acc[j] = __umulhi(acc[j], b) + 0x10000001u;  // v_mul_hi_u32 + add
  }
  u32 s = 0; for (int j = 0; j < UNR; j++) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad24(u32* out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = (a | 1) & 0xffffff;
  u32 acc[UNR];
  for (int j = 0; j < UNR; j++) acc[j] = a + j;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int j = 0; j < UNR; j++) acc[j] = __umul24(acc[j], b) + acc[j];  // v_mad_u32_u24
  }
  u32 s = 0; for (int j = 0; j < UNR; j++) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi24(u32* out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = (a | 1) & 0xffffff;
  u32 acc[UNR];
  for (int j = 0; j < UNR; j++) acc[j] = a + j;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int j = 0; j < UNR; j++) acc[j] = __umulhi24x(acc[j], b) ^ a;  // v_mul_hi_u32_u24
  }
  u32 s = 0; for (int j = 0; j < UNR; j++) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(u32* out, u32 seed) {
  double a = 1.0 + (threadIdx.x + seed) * 1e-9, b = 0.999999;
  double acc[UNR];
  for (int j = 0; j < UNR; j++) acc[j] = a + j;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int j = 0; j < UNR; j++) acc[j] = __builtin_fma(acc[j], b, a);  // v_fma_f64
  }
  double s = 0; for (int j = 0; j < UNR; j++) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s;
}
__global__ void k_fma32(u32* out, u32 seed) {
  float a = 1.0f + (threadIdx.x + seed) * 1e-6f, b = 0.99999f;
  float acc[UNR];
  for (int j = 0; j < UNR; j++) acc[j] = a + j;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int j = 0; j < UNR; j++) acc[j] = __builtin_fmaf(acc[j], b, a);  // v_fma_f32
  }
  float s = 0; for (int j = 0; j < UNR; j++) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s;
}
__global__ void k_addc(u32* out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed;
  u64 acc[UNR];
  for (int j = 0; j < UNR; j++) acc[j] = ((u64)a << 32) + j;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int j = 0; j < UNR; j++) acc[j] = acc[j] + (acc[j] >> 7) ;  // 64-bit add = add_co + addc (+ shifts)
  }
  u64 s = 0; for (int j = 0; j < UNR; j++) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}

template <typename K>
int run(const char* name, K kern, double ops_per_thread, int wavesPerSimd, u32* d_out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = 256 * wavesPerSimd;  // 256-thread blocks: 4 waves each, one per SIMD
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, (u32)r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double total = ops_per_thread * blocks * 256.0;
  double rate = total / (best * 1e-3);
  // lanes/clk/CU at 2.4 GHz
  printf("%-12s waves/SIMD=%d  %8.3f ms  %8.3f Tops/s  = %6.2f lane-ops/clk/CU @2.4GHz\n", name, wavesPerSimd, best, rate / 1e12,
         rate / (256.0 * 2.4e9));
  return 0;
}
int main() {
  u32* d_out; CK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(u32) * 4));
  double ops = (double)ITERS * UNR;
  for (int w : {1, 2, 4, 8}) {
    run("fma_f32", k_fma32, ops, w, d_out);
    run("mad_u64_u32", k_mad64, ops, w, d_out);
    run("mul_lo_u32", k_mullo, ops, w, d_out);
    run("mul_hi_u32+add", k_mulhi, ops, w, d_out);
    run("mad_u32_u24", k_mad24, ops, w, d_out);
    run("mul_hi_u24", k_mulhi24, ops, w, d_out);
    run("fma_f64", k_fma64, ops, w, d_out);
    run("add64(shift)", k_addc, ops, w, d_out);
    printf("\n");
  }
  return 0;
}
