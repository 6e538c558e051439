// Microbenchmark (round 3): v_mad_i64_i32 is a VOP3B instruction -- it writes a carry-out SGPR pair (SDST) besides its 64-bit result.  The compiler gives
// every multiply-add of the wave VM the same dead pair (s[6:7]).  Does that write-after-write chain on ONE scalar pair limit a lone wavefront (7.3 clocks
// per multiply-add against 5.5 for simple VALU work)?  Variants: the same pair every time, 2 / 4 pairs in rotation, and VCC.
// Output: clocks per wave-instruction per SIMD at 2.4 GHz for 1 / 2 / 3 / 4 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
#define TOTAL 28672
#define D 14

#define MAD(S, CL0, CL1, k) asm volatile("v_mad_i64_i32 %0, " S ", %1, %2, %0" : "+v"(acc[k]) : "v"(b), "v"(a) : CL0, CL1)
template <int V> __global__ void k_sdst(u32* out, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = a | 1;
  u64 acc[D];
  for (int j = 0; j < D; j++) acc[j] = ((u64)a << 20) + j;
  for (int i = 0; i < TOTAL / D; i++) {
#pragma unroll
    for (int j = 0; j < D; j++) {
      if (V == 0) MAD("s[10:11]", "s10", "s11", j);
      else if (V == 1) { if (j & 1) MAD("s[12:13]", "s12", "s13", j); else MAD("s[10:11]", "s10", "s11", j); }
      else if (V == 2) { switch (j & 3) { case 0: MAD("s[10:11]", "s10", "s11", j); break; case 1: MAD("s[12:13]", "s12", "s13", j); break; case 2: MAD("s[14:15]", "s14", "s15", j); break; default: MAD("s[16:17]", "s16", "s17", j); } }
      else MAD("vcc", "vcc", "vcc", j);
    }
  }
  u64 s = 0; for (int j = 0; j < D; j++) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}
// the same stream without any SDST: 32-bit low and high halves of the product with v_mul_lo_u32 / v_mul_hi_u32 + add (what a carry-free formulation would issue)
__global__ void k_mul24(u32* out, u32 seed) {
  u32 a = (threadIdx.x * 2654435761u + seed) & 0xffffff, b = (a | 1) & 0xffffff;
  u32 acc[D];
  for (int j = 0; j < D; j++) acc[j] = a + j;
  for (int i = 0; i < TOTAL / D; i++) {
#pragma unroll
    for (int j = 0; j < D; j++) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(b), "v"(a));
  }
  u32 s = 0; for (int j = 0; j < D; j++) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K> int run(const char* name, K kern, int wavesPerSimd, u32* d_out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = 256 * wavesPerSimd;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, (u32)r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-22s waves/SIMD=%d  %7.3f ms  %6.2f clk per instruction per SIMD\n", name, wavesPerSimd, best, best * 1e-3 * 2.4e9 / ((double)TOTAL * wavesPerSimd));
  return 0;
}
int main() {
  u32* d_out; CK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(u32)));
  for (int w : {1, 2, 3, 4}) {
    run("same SDST pair", k_sdst<0>, w, d_out); run("2 SDST pairs", k_sdst<1>, w, d_out); run("4 SDST pairs", k_sdst<2>, w, d_out); run("SDST = vcc", k_sdst<3>, w, d_out);
    run("v_mad_u32_u24 (no SDST)", k_mul24, w, d_out);
    printf("\n");
  }
  return 0;
}
