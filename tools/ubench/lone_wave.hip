// lone_wave.hip -- issue rate of ONE wavefront per SIMD for the instruction mixes of the VM kernel:
//   (a) 196 independent v_mad_i64_i32 (row-wise order: consecutive MADs hit different accumulators)
//   (b) the same products ordered column-wise (each MAD consumes the previous MAD's result)
//   (c) independent 32-bit adds
//   (d) (a) + per block 16 ds_read_b128 from per-lane slots and 28 adds on the loaded words (operand formation)
//   (e) (d) + lane-divergent branches around half of the adds (exec-mask juggling as in dot_operand)
//   (f) (e) + one dependent global load per block (descriptor word)
//   (g) (a) with the block replicated 24 times in straight-line code (~38 KB of instructions: the VM kernel's code footprint)
// Prints clocks per instruction for 1, 2, 3 and 4 waves per SIMD (grid = waves x 1024 SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef long long i64; typedef int i32; typedef unsigned long long u64; typedef unsigned u32;

template <int MODE>
__global__ void __launch_bounds__(64) k(u32* out, u64* clk, int iters, u32 seed, const u32* gmem) {
  constexpr int LW = (MODE >= 3 && MODE <= 5) ? 64 * 16 : 16;      // 4 KB of slots only where they are used, so LDS never limits occupancy
  __shared__ __attribute__((aligned(16))) u32 lds[LW];
  for (int i = threadIdx.x; i < LW; i += 64) lds[i] = i * seed;
  __syncthreads();
  u32 gi = threadIdx.x;
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
    } else if (MODE == 6) {
#pragma unroll
      for (int rep = 0; rep < 24; rep++) {
#pragma unroll
        for (int i = 0; i < 14; i++)
#pragma unroll
          for (int j = 0; j < 14; j++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i + j]) : "v"(a[(j + rep) % 14]), "v"(b[i]) : "vcc");
      }
    } else if (MODE >= 3) {
      // operand formation: four 14-limb loads (slot = f(lane, it)), combine, then the 196 MADs
      u32 e = (MODE >= 5) ? gmem[gi & 1023] : (u32)it * 2654435761u;
      if (MODE >= 5) gi = gi * 5 + e;
      const uint4* L = (const uint4*)lds;
      u32 x[16], y[16];
      { uint4 q0 = L[((threadIdx.x + e) & 15) * 4 + 0], q1 = L[((threadIdx.x + e) & 15) * 4 + 1], q2 = L[((threadIdx.x + e) & 15) * 4 + 2], q3 = L[((threadIdx.x + e) & 15) * 4 + 3];
        x[0]=q0.x;x[1]=q0.y;x[2]=q0.z;x[3]=q0.w;x[4]=q1.x;x[5]=q1.y;x[6]=q1.z;x[7]=q1.w;x[8]=q2.x;x[9]=q2.y;x[10]=q2.z;x[11]=q2.w;x[12]=q3.x;x[13]=q3.y;x[14]=q3.z;x[15]=q3.w; }
      { uint4 q0 = L[((threadIdx.x * 3 + e) & 15) * 4 + 0], q1 = L[((threadIdx.x * 3 + e) & 15) * 4 + 1], q2 = L[((threadIdx.x * 3 + e) & 15) * 4 + 2], q3 = L[((threadIdx.x * 3 + e) & 15) * 4 + 3];
        y[0]=q0.x;y[1]=q0.y;y[2]=q0.z;y[3]=q0.w;y[4]=q1.x;y[5]=q1.y;y[6]=q1.z;y[7]=q1.w;y[8]=q2.x;y[9]=q2.y;y[10]=q2.z;y[11]=q2.w;y[12]=q3.x;y[13]=q3.y;y[14]=q3.z;y[15]=q3.w; }
      if (MODE >= 4) {
        if ((threadIdx.x ^ e) & 1) {
#pragma unroll
          for (int i = 0; i < 14; i++) a[i] = (i32)(x[i] + y[i]);
        } else {
#pragma unroll
          for (int i = 0; i < 14; i++) a[i] = (i32)(x[i] - y[i]);
        }
        if ((threadIdx.x ^ e) & 2) {
#pragma unroll
          for (int i = 0; i < 14; i++) b[i] = (i32)(0u - y[i]);
        } else {
#pragma unroll
          for (int i = 0; i < 14; i++) b[i] = (i32)y[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 14; i++) { a[i] = (i32)(x[i] + y[i]); b[i] = (i32)(x[i] - y[i]); }
      }
#pragma unroll
      for (int i = 0; i < 14; i++)
#pragma unroll
        for (int j = 0; j < 14; j++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i + j]) : "v"(a[j]), "v"(b[i]) : "vcc");
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
  u32* gmem; hipMalloc(&gmem, 4096); hipMemset(gmem, 1, 4096);
  for (int mode = 0; mode < 7; mode++) for (int wps = 1; wps <= (mode == 0 ? 8 : 4); wps++) {
    int blocks = 1024 * wps; u32* out; u64* clk; hipMalloc(&out, blocks * 64 * 4); hipMalloc(&clk, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0, 0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 12345u, gmem);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 12345u, gmem);
      else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 12345u, gmem);
      else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 12345u, gmem);
      else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 12345u, gmem);
      else if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 12345u, gmem);
      else hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, out, clk, iters / 24, 12345u, gmem);
      hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<u64> h(blocks); hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    printf("mode %d (%s) waves/SIMD %d: %.0f ticks per block per wave, %.0f per SIMD (blocks of 196 MADs); kernel %.3f ms -> %.1f T lane-MAD/s\n", mode,
           mode == 0 ? "mad row-wise" : mode == 1 ? "mad column-wise" : mode == 2 ? "add32" : mode == 3 ? "lds+adds+mad" : mode == 4 ? "lds+divergent adds+mad" : mode == 5 ? "global+lds+divergent+mad" : "mad, 38 KB straight-line (per 24 blocks)", wps, avg / iters, avg / iters / wps, ms, (mode == 2 ? 0.0 : (double)blocks * 64 * 196 * (mode == 6 ? (iters / 24) * 24 : iters) / (ms * 1e-3) / 1e12));
    hipFree(out); hipFree(clk);
  }
  return 0;
}
