// Microbenchmark: what clock and package power does a SATURATED stream of v_mad_u64_u32 sustain on gfx950 -- is the 2.4 GHz of the roofline peak reachable by the instruction the
// engine is made of?  Runs one of three kernels back to back for a few seconds and prints the achieved multiply-add rate; tools/mad_power.sh samples rocm-smi beside it.
//   mode 0: multiply-adds only, 32-bit random operands     mode 1: the same with 28-bit operands (the engine's limbs)     mode 2: 196 multiply-adds per 28 LDS dword reads + 14 writes
//           (mode 2 is LDS-bound as written -- dword reads, 14.8 T multiply-adds/s at 880 W -- and says nothing about the engine, whose operand reads are 16-byte reads)
//   mode 3: v_mad_i64_i32 on SIGNED 28-bit operands, half of them negative (what the engine's operands look like: sign-extended limbs)
// Usage: mad_power <mode> <seconds> [waves per SIMD = 8]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define ITERS 2048
template <int MODE> __global__ void __launch_bounds__(64) k(u32* out, u32 seed) {
  __shared__ u32 lds[64 * 40];      // two 80-byte slots per lane (the engine's stride where it fits: 16-byte reads without the 4-way conflicts of 64-byte slots)
  const u32 mask = MODE == 0 ? 0xffffffffu : 0x0fffffffu;      // (mode 3 sign-extends instead)
  u32 a[14], b[14];
  for (int j = 0; j < 14; j++) { a[j] = (threadIdx.x * 2654435761u + seed * (j + 1)) & mask; b[j] = (a[j] * 40503u + j) & mask; lds[threadIdx.x * 40 + j] = a[j]; lds[threadIdx.x * 40 + 20 + j] = b[j]; }
  u64 acc[28];
  for (int j = 0; j < 28; j++) acc[j] = j;
  for (int i = 0; i < ITERS; i++) {
    if (MODE == 2) {      // operands come from LDS every iteration (a neighbour lane's slot, as in the engine: lane-ops read what other lanes wrote); the offset is opaque to the compiler
      u32 off = ((threadIdx.x + 1 + (i & 3)) & 63) * 40;
      asm volatile("" : "+v"(off));
#pragma unroll
      for (int j = 0; j < 14; j++) { a[j] = lds[off + j]; b[j] = lds[off + 20 + j]; }
    }
    if (MODE == 3) {
#pragma unroll
      for (int x = 0; x < 14; x++)
#pragma unroll
        for (int y = 0; y < 14; y++) acc[x + y] = (u64)((long long)(int)a[x] * (long long)(int)b[y] + (long long)acc[x + y]);      // 196 v_mad_i64_i32
    } else {
#pragma unroll
    for (int x = 0; x < 14; x++)
#pragma unroll
      for (int y = 0; y < 14; y++) acc[x + y] = (u64)a[x] * b[y] + acc[x + y];      // 196 v_mad_u64_u32, the engine's product block
    }
    if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 14; j++) lds[threadIdx.x * 40 + j] = ((u32)acc[j] + i) & mask;
    } else if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 14; j++) a[j] = (u32)((int)((a[j] + (u32)acc[j + 7]) << 4) >> 4);      // sign-extended 28-bit value: about half of them negative
    } else {
#pragma unroll
      for (int j = 0; j < 14; j++) a[j] = (a[j] + (u32)acc[j + 7]) & mask;
    }
#pragma unroll
    for (int j = 0; j < 28; j++) acc[j] &= 0xffffffffffffull;
  }
  u64 s = lds[(threadIdx.x * 40 + 5) % 2560]; for (int j = 0; j < 28; j++) s += acc[j];
  out[blockIdx.x * 64 + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0; const double secs = argc > 2 ? atof(argv[2]) : 3.0; const int wps = argc > 3 ? atoi(argv[3]) : 8;
  const int blocks = 1024 * wps;
  u32* out; CK(hipMalloc(&out, (size_t)blocks * 64 * 4));
  auto launch = [&](u32 seed) { if (mode == 0) k<0><<<blocks, 64>>>(out, seed); else if (mode == 1) k<1><<<blocks, 64>>>(out, seed); else if (mode == 3) k<3><<<blocks, 64>>>(out, seed); else k<2><<<blocks, 64>>>(out, seed); };
  launch(1); CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now(); long n = 0; double dt = 0;
  do { for (int r = 0; r < 8; r++) launch((u32)(n + r + 2)); CK(hipDeviceSynchronize()); n += 8; dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } while (dt < secs);
  const double mads = (double)n * blocks * 64 * ITERS * 196;
  printf("mode %d, %d waves per SIMD: %.2f T lane-MAD/s over %.1f s (nominal peak at 2.4 GHz: 39.32; %.1f %% of it)\n", mode, wps, mads / dt / 1e12, dt, mads / dt / 39.32e10);
  return 0;
}
