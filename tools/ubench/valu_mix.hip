// Microbenchmark (round 2): what a VALU instruction of the wave VM's mix costs on gfx950 when the SIMD is saturated (8 wavefronts
// per SIMD) -- the 64-bit multiply-add next to the "simple" integer instructions of operand formation and reduction -- and
// what the shader clock is while doing so.  Each kernel runs 16 independent dependency chains of ONE instruction (inline
// asm, so the compiler can neither fuse nor drop them); `mix` alternates one v_mad_i64_i32 with one v_add_u32.
// Output: cycles per wave-instruction per SIMD (from s_memtime of a resident wave and from wall-clock at the measured clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
#define ITERS 2048
#define UNR 16

#define KERNEL32(name, ASM)                                                                 \
__global__ void name(u32* out, u64* clk, u32 seed) {                                        \
  u32 a = threadIdx.x * 2654435761u + seed, b = a | 1;                                      \
  u32 acc[UNR];                                                                             \
  for (int j = 0; j < UNR; j++) acc[j] = a + j;                                             \
  u64 t0 = __builtin_readcyclecounter();                                                    \
  for (int i = 0; i < ITERS; i++) {                                                         \
    _Pragma("unroll") for (int j = 0; j < UNR; j++) asm volatile(ASM : "+v"(acc[j]) : "v"(b), "v"(a)); \
  }                                                                                         \
  u64 t1 = __builtin_readcyclecounter();                                                    \
  u32 s = 0; for (int j = 0; j < UNR; j++) s += acc[j];                                     \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                           \
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                          \
}
#define KERNEL64(name, ASM)                                                                 \
__global__ void name(u32* out, u64* clk, u32 seed) {                                        \
  u32 a = threadIdx.x * 2654435761u + seed, b = a | 1;                                      \
  u64 acc[UNR];                                                                             \
  for (int j = 0; j < UNR; j++) acc[j] = ((u64)a << 20) + j;                                \
  u64 t0 = __builtin_readcyclecounter();                                                    \
  for (int i = 0; i < ITERS; i++) {                                                         \
    _Pragma("unroll") for (int j = 0; j < UNR; j++) asm volatile(ASM : "+v"(acc[j]) : "v"(b), "v"(a)); \
  }                                                                                         \
  u64 t1 = __builtin_readcyclecounter();                                                    \
  u64 s = 0; for (int j = 0; j < UNR; j++) s += acc[j];                                     \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);                     \
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                          \
}
KERNEL32(k_add, "v_add_u32 %0, %0, %1")
KERNEL32(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 28")
KERNEL32(k_ashr32, "v_ashrrev_i32 %0, 3, %0")
KERNEL32(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 3, 20")
KERNEL64(k_mad_i64, "v_mad_i64_i32 %0, s[10:11], %1, %2, %0")
KERNEL64(k_mad_u64, "v_mad_u64_u32 %0, s[10:11], %1, %2, %0")
KERNEL64(k_ashr64, "v_ashrrev_i64 %0, 3, %0")
KERNEL64(k_lshladd64, "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL64(k_mov64, "v_mov_b64 %0, %0")
// one multiply-add followed by one simple instruction on another chain
__global__ void k_mix(u32* out, u64* clk, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = a | 1;
  u64 acc[UNR]; u32 x[UNR];
  for (int j = 0; j < UNR; j++) { acc[j] = ((u64)a << 20) + j; x[j] = a + j; }
  u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int j = 0; j < UNR; j++) { asm volatile("v_mad_i64_i32 %0, s[10:11], %1, %2, %0" : "+v"(acc[j]) : "v"(b), "v"(a)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[j]) : "v"(b)); }
  }
  u64 t1 = __builtin_readcyclecounter();
  u64 s = 0; for (int j = 0; j < UNR; j++) s += acc[j] + x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
// two simple instructions per multiply-add
__global__ void k_mix2(u32* out, u64* clk, u32 seed) {
  u32 a = threadIdx.x * 2654435761u + seed, b = a | 1;
  u64 acc[UNR]; u32 x[UNR];
  for (int j = 0; j < UNR; j++) { acc[j] = ((u64)a << 20) + j; x[j] = a + j; }
  u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int j = 0; j < UNR; j++) { asm volatile("v_mad_i64_i32 %0, s[10:11], %1, %2, %0" : "+v"(acc[j]) : "v"(b), "v"(a)); asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %2" : "+v"(x[j]) : "v"(b), "v"(a)); }
  }
  u64 t1 = __builtin_readcyclecounter();
  u64 s = 0; for (int j = 0; j < UNR; j++) s += acc[j] + x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <typename K>
int run(const char* name, K kern, double insts_per_wave, int wavesPerSimd, u32* d_out, u64* d_clk) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = 256 * wavesPerSimd;  // 256-thread blocks: 4 waves each, one per SIMD
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, d_clk, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, d_clk, (u32)r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  u64 clk[64]; CK(hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost));
  double c = 0; for (int i = 0; i < 64; i++) c += (double)clk[i]; c /= 64;   // s_memtime ticks (100 MHz constant clock on gfx9: convert through wall time below)
  // cycles per wave-instruction per SIMD from wall-clock at 2.4 GHz: every SIMD runs wavesPerSimd waves of insts_per_wave instructions
  double cyc24 = best * 1e-3 * 2.4e9 / (insts_per_wave * wavesPerSimd);
  printf("%-14s waves/SIMD=%d  %8.3f ms  %6.2f clk/inst/SIMD @2.4GHz   (memtime ticks per wave %.0f = %.3f ms @100MHz)\n", name, wavesPerSimd, best, cyc24, c, c / 1e5);
  return 0;
}
int main() {
  u32* d_out; CK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(u32) * 4));
  u64* d_clk; CK(hipMalloc(&d_clk, 256 * 8 * 4 * sizeof(u64)));
  double n = (double)ITERS * UNR;
  for (int w : {1, 2, 8}) {
    run("v_mad_i64_i32", k_mad_i64, n, w, d_out, d_clk);
    run("v_mad_u64_u32", k_mad_u64, n, w, d_out, d_clk);
    run("v_add_u32", k_add, n, w, d_out, d_clk);
    run("v_sub_u32", k_sub, n, w, d_out, d_clk);
    run("v_and_b32", k_and, n, w, d_out, d_clk);
    run("v_xor_b32", k_xor, n, w, d_out, d_clk);
    run("v_mov_b32", k_mov, n, w, d_out, d_clk);
    run("v_lshl_add_u32", k_lshladd, n, w, d_out, d_clk);
    run("v_add3_u32", k_add3, n, w, d_out, d_clk);
    run("v_alignbit_b32", k_alignbit, n, w, d_out, d_clk);
    run("v_ashrrev_i32", k_ashr32, n, w, d_out, d_clk);
    run("v_bfe_u32", k_bfe, n, w, d_out, d_clk);
    run("v_mul_lo_u32", k_mullo, n, w, d_out, d_clk);
    run("v_ashrrev_i64", k_ashr64, n, w, d_out, d_clk);
    run("v_lshl_add_u64", k_lshladd64, n, w, d_out, d_clk);
    run("v_mov_b64", k_mov64, n, w, d_out, d_clk);
    run("mad+add", k_mix, 2 * n, w, d_out, d_clk);
    run("mad+add+xor", k_mix2, 3 * n, w, d_out, d_clk);
    printf("\n");
  }
  return 0;
}
