// lone_wave.hip -- issue rate of ONE wavefront per SIMD for the instruction mixes of the VM kernel:
//   (a) 196 independent v_mad_i64_i32 (row-wise order: consecutive MADs hit different accumulators)
//   (b) the same products ordered column-wise (each MAD consumes the previous MAD's result)
//   (c) independent 32-bit adds
// Prints clocks per instruction for 1, 2, 3 and 4 waves per SIMD (grid = waves x 1024 SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef long long i64; typedef int i32; typedef unsigned long long u64; typedef unsigned u32;

template <int MODE>
__global__ void __launch_bounds__(64) k(u32* out, u64* clk, int iters, u32 seed) {
  i32 a[14], b[14]; i64 acc[28];
  for (int i = 0; i < 14; i++) { a[i] = (seed * (i + 1) + threadIdx.x) & 0xfffffff; b[i] = (seed * (i + 3) ^ threadIdx.x) & 0xfffffff; }
  for (int i = 0; i < 28; i++) acc[i] = 0;
  u64 t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 14; i++)
#pragma unroll
        for (int j = 0; j < 14; j++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i + j]) : "v"(a[j]), "v"(b[i]) : "vcc");
    } else if (MODE == 1) {
#pragma unroll
      for (int c = 0; c < 27; c++)
#pragma unroll
        for (int i = 0; i < 14; i++) { int j = c - i; if (j >= 0 && j < 14) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a[j]), "v"(b[i]) : "vcc"); }
    } else {
#pragma unroll
      for (int r = 0; r < 7; r++)
#pragma unroll
        for (int i = 0; i < 28; i++) { u32 lo = (u32)acc[i]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a[i % 14])); acc[i] = lo; }
    }
    a[0] ^= (u32)acc[27] & 0xff;            // loop-carried dependence: keeps the body from being hoisted
    asm volatile("" ::: "memory");
  }
  u64 t1 = __builtin_readcyclecounter();
  u32 s = 0; for (int i = 0; i < 28; i++) s ^= (u32)acc[i] ^ (u32)(acc[i] >> 32);
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
  const int iters = 2000;
  for (int mode = 0; mode < 3; mode++) for (int wps = 1; wps <= 4; wps++) {
    int blocks = 1024 * wps; u32* out; u64* clk; hipMalloc(&out, blocks * 64 * 4); hipMalloc(&clk, blocks * 8);
    for (int rep = 0; rep < 2; rep++) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 12345u);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 12345u);
      else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 12345u);
      hipDeviceSynchronize();
    }
    std::vector<u64> h(blocks); hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    printf("mode %d (%s) waves/SIMD %d: %.2f wave-clocks per instruction (196 instr per iteration), %.2f SIMD clocks per instruction\n", mode,
           mode == 0 ? "mad row-wise" : mode == 1 ? "mad column-wise" : "add32", wps, avg / iters / 196.0, avg / iters / 196.0 / wps);
    hipFree(out); hipFree(clk);
  }
  return 0;
}
