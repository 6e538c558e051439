// Microbenchmark (round 2): does it matter how far apart DEPENDENT v_mad_i64_i32 are?  The compiler turns the unrolled 14 x 14 limb product of the wave VM
// into per-column chains (each multiply-add consumes the result of the one just before it: dependency distance 1..2 instead of 14), so this measures
// D interleaved accumulator chains (every multiply-add depends on the one D instructions earlier) at 1 / 2 / 3 / 4 / 8 wavefronts per SIMD.
// Output: clocks per wave-instruction per SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
#define TOTAL 28672   // multiply-adds per lane (a multiple of every D below)

template <int D> __global__ void k_chain(u32* out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = a | 1;
  u64 acc[D];
  for (int j = 0; j < D; j++) acc[j] = ((u64)a << 20) + j;
  for (int i = 0; i < TOTAL / D / 4; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int j = 0; j < D; j++) asm volatile("v_mad_i64_i32 %0, s[10:11], %1, %2, %0" : "+v"(acc[j]) : "v"(b), "v"(a) : "s10", "s11");
    }
  }
  u64 s = 0; for (int j = 0; j < D; j++) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}
template <int D> int run(int wavesPerSimd, u32* d_out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = 256 * wavesPerSimd;   // 256-thread blocks: four wavefronts, one per SIMD
  hipLaunchKernelGGL(k_chain<D>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_chain<D>, dim3(blocks), dim3(256), 0, 0, d_out, (u32)r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("distance %2d  waves/SIMD=%d  %7.3f ms  %6.2f clk per multiply-add per SIMD\n", D, wavesPerSimd, best, best * 1e-3 * 2.4e9 / ((double)TOTAL * wavesPerSimd));
  return 0;
}
int main() {
  u32* d_out; CK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(u32)));
  for (int w : {1, 2, 3, 4, 8}) {
    run<1>(w, d_out); run<2>(w, d_out); run<4>(w, d_out); run<7>(w, d_out); run<14>(w, d_out);
    printf("\n");
  }
  return 0;
}
