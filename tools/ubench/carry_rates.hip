// Issue rates of the carry-chain instructions around v_mad_u64_u32 (inline asm, dependent/independent mixes).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32; typedef unsigned long long u64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
#define REP8(x) x x x x x x x x
template <int V> __global__ void __launch_bounds__(64) k(u32* out, int iters) {
  u32 a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = a0 | 1;
  u64 q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;
  for (int it = 0; it < iters; it++) {
    if (V == 0) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %1, %0\n v_mov_b32 %3, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }   // 32 v_mov
    if (V == 1) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }  // 32 adds in vcc chains
    if (V == 2) { u64 c; REP8(asm volatile("v_addc_co_u32_e64 %0, %4, 0, %0, %4\n v_addc_co_u32_e64 %1, %4, 0, %1, %4\n v_addc_co_u32_e64 %2, %4, 0, %2, %4\n v_addc_co_u32_e64 %3, %4, 0, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&s"(c));) (void)c; }  // 32 addc e64 sgpr
    if (V == 3) { u64 c; REP8(asm volatile("v_mad_u64_u32 %0, %4, %5, %6, %0\n v_mad_u64_u32 %1, %4, %5, %6, %1\n v_mad_u64_u32 %2, %4, %5, %6, %2\n v_mad_u64_u32 %3, %4, %5, %6, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "=&s"(c) : "v"(a0), "v"(b));) (void)c; }  // 32 mad (4 independent chains)
    if (V == 4) { u64 c; REP8(asm volatile("v_mad_u64_u32 %0, %8, %9, %10, %0\n v_addc_co_u32_e64 %4, %8, 0, %4, %8\n v_mad_u64_u32 %1, %8, %9, %10, %1\n v_addc_co_u32_e64 %5, %8, 0, %5, %8\n v_mad_u64_u32 %2, %8, %9, %10, %2\n v_addc_co_u32_e64 %6, %8, 0, %6, %8\n v_mad_u64_u32 %3, %8, %9, %10, %3\n v_addc_co_u32_e64 %7, %8, 0, %7, %8"
                      : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(c) : "v"(a0), "v"(b));) (void)c; }   // 32 mad + 32 addc, adjacent pairs
    if (V == 5) { u64 c0, c1, c2, c3; REP8(asm volatile("v_mad_u64_u32 %0, %8, %12, %13, %0\n v_mad_u64_u32 %1, %9, %12, %13, %1\n v_mad_u64_u32 %2, %10, %12, %13, %2\n v_mad_u64_u32 %3, %11, %12, %13, %3\n v_addc_co_u32_e64 %4, %8, 0, %4, %8\n v_addc_co_u32_e64 %5, %9, 0, %5, %9\n v_addc_co_u32_e64 %6, %10, 0, %6, %10\n v_addc_co_u32_e64 %7, %11, 0, %7, %11"
                      : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3) : "v"(a0), "v"(b));) (void)c0; }   // grouped: 4 mads then 4 addcs
    if (V == 6) { REP8(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");) }
    if (V == 7) { u64 c; REP8(asm volatile("v_mad_u64_u32 %0, %4, %5, %6, %0\n v_mov_b32 %7, %8\n v_mad_u64_u32 %1, %4, %5, %6, %1\n v_mov_b32 %8, %7\n v_mad_u64_u32 %2, %4, %5, %6, %2\n v_mov_b32 %7, %8\n v_mad_u64_u32 %3, %4, %5, %6, %3\n v_mov_b32 %8, %7" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "=&s"(c) : "v"(a0), "v"(b), "v"(a4), "v"(a5));) (void)c; }  // mad + mov interleaved
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (u32)(q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7);
}
template <int V> int run(const char* name, int w, u32* d, double ninstr) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int iters = 2000; hipLaunchKernelGGL(k<V>, dim3(1024 * w), dim3(64), 0, 0, d, 10); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<V>, dim3(1024 * w), dim3(64), 0, 0, d, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  printf("%-34s waves/SIMD=%d  %.2f clk per instruction per SIMD (2.4 GHz)\n", name, w, best * 1e-3 * 2.4e9 / (iters * ninstr * w));
  return 0;
}
int main() {
  u32* d; CK(hipMalloc(&d, 1024 * 8 * 64 * 4));
  for (int w : {1, 4}) {
    run<0>("v_mov_b32", w, d, 32); run<1>("v_add_co/v_addc_co (vcc chain)", w, d, 32); run<2>("v_addc_co_u32_e64 (sgpr carry)", w, d, 32); run<3>("v_mad_u64_u32 (4 chains)", w, d, 32);
    run<4>("mad+addc adjacent pairs", w, d, 64); run<5>("4 mads then 4 addcs", w, d, 64); run<6>("s_nop 0", w, d, 32); run<7>("mad + v_mov interleaved", w, d, 64);
  }
  return 0;
}
