// Microbenchmark (round 2, exploratory -- VERDICT item 9): could the limb products of the Fp12 tower run on the matrix cores?
//
// A 381-bit product is a convolution, not a contraction, but a schoolbook ROW of an Fp12 product multiplies ONE coefficient a against 12
// coefficients b_j, and with 8-bit limbs (48 per element) that row is the matrix product  T(a) [96 x 48, Toeplitz]  x  B [48 x 12]:
// six v_mfma_i32_16x16x64_i8 (M = 96 = 6 x 16 output columns, N = 16 >= 12 products, K = 64 >= 48 limbs) per 12 products, with exact
// lazy accumulation in the i32 result (a column holds up to 48 x 2^14 per product).  Three things are measured on a saturated chip:
//   mfma_row    the six MFMAs per row, accumulators kept live across `terms` rows (a lazy dot product), operands assumed free
//   toeplitz    the same plus forming the A operand of every MFMA from a byte string at an arbitrary byte offset (5 dword reads + 4
//               v_alignbyte per lane and tile: what a Toeplitz matrix costs when it is never materialised)
//   normalise   one byte-normalisation of the 96 result columns of 16 products (3 carry-save passes over the 24 accumulator registers
//               of a lane with row-shifted neighbours), needed before a result can be an operand again or enter a Montgomery reduction
//   mac28       the engine's limb product: 196 v_mad_i64_i32 per product and lane, 64 products per wave
// Output: wave-clocks per Fp product for each (2.4 GHz nominal), so that DESIGN.md can state the verdict with numbers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
#define ROWS 2048      // schoolbook rows per wave in a launch

// six accumulators (96 output columns x 16 products), `terms` rows accumulated lazily, then one XOR-fold so that nothing is dead
__global__ void k_mfma_row(u32* out, u32 seed, int terms) {
  v4i c[6]; for (int t = 0; t < 6; t++) c[t] = v4i{0, 0, 0, 0};
  v4i a = {(int)(threadIdx.x * 2654435761u + seed), (int)seed, 3, 4}, b = {(int)threadIdx.x, 7, (int)seed, 9};
  for (int r = 0; r < ROWS; r++) {
#pragma unroll
    for (int t = 0; t < 6; t++) c[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[t], 0, 0, 0);
    if (terms && (r % terms) == terms - 1) { a[0] ^= c[0][0]; }   // the accumulators are consumed every `terms` rows
  }
  u32 s = 0; for (int t = 0; t < 6; t++) s ^= c[t][0] ^ c[t][1] ^ c[t][2] ^ c[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the A operand of every MFMA is formed from LDS bytes at a per-lane, per-tile byte offset (Toeplitz rows are shifted copies of a)
__global__ void k_toeplitz(u32* out, u32 seed) {
  __shared__ u32 abytes[64 * 4];       // 48 limb bytes of `a` (reversed, zero padded) for four items
  abytes[threadIdx.x & 255] = threadIdx.x * 2654435761u + seed;
  __syncthreads();
  v4i c[6]; for (int t = 0; t < 6; t++) c[t] = v4i{0, 0, 0, 0};
  v4i b = {(int)threadIdx.x, 7, (int)seed, 9};
  const u32 lane = threadIdx.x & 63;
  for (int r = 0; r < ROWS; r++) {
#pragma unroll
    for (int t = 0; t < 6; t++) {
      const u32 off = (16 * t + (lane & 15) + 16 * (lane >> 4) + r) & 63;     // byte offset of this lane's 16-byte window
      const u32* p = abytes + (off >> 2);
      const u32 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4], sh = (off & 3) * 8;
      v4i a = {(int)__builtin_amdgcn_alignbyte(w1, w0, off & 3), (int)__builtin_amdgcn_alignbyte(w2, w1, off & 3), (int)__builtin_amdgcn_alignbyte(w3, w2, off & 3), (int)__builtin_amdgcn_alignbyte(w4, w3, off & 3)};
      (void)sh;
      c[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[t], 0, 0, 0);
    }
  }
  u32 s = 0; for (int t = 0; t < 6; t++) s ^= c[t][0] ^ c[t][1] ^ c[t][2] ^ c[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// byte-normalisation of the 96 columns of 16 products held as 24 i32 per lane: three carry-save passes; a column's lower neighbours are
// the previous registers of the lane or, across a 4-row boundary, the registers of the lane 16 below (ds_bpermute)
__global__ void k_normalise(u32* out, u32 seed) {
  int c[24];
  for (int i = 0; i < 24; i++) c[i] = (int)((threadIdx.x * 2654435761u + seed * (i + 1)) & 0xfffff);
  const int src = (int)(((threadIdx.x & 63) - 16) & 63) * 4;
  for (int r = 0; r < ROWS / 8; r++) {
#pragma unroll
    for (int pass = 0; pass < 3; pass++) {
      int hi_prev1 = __builtin_amdgcn_ds_bpermute(src, c[23] >> 8), hi_prev2 = __builtin_amdgcn_ds_bpermute(src, c[22] >> 8);
#pragma unroll
      for (int i = 0; i < 24; i++) {
        const int hi = c[i] >> 8, lo = c[i] & 0xff;
        c[i] = lo + (hi_prev1 & 0xff) + (hi_prev2 >> 8);
        hi_prev2 = hi_prev1; hi_prev1 = hi;
      }
    }
    c[0] += r;
  }
  u32 s = 0; for (int i = 0; i < 24; i++) s ^= (u32)c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the engine's product: 14 x 14 limbs of 28 bits into 28 signed 64-bit columns, one product per lane
__global__ void k_mac28(u32* out, u32 seed) {
  u64 acc[28]; for (int i = 0; i < 28; i++) acc[i] = 0;
  u32 a[14], b[14];
  for (int i = 0; i < 14; i++) { a[i] = (threadIdx.x * 2654435761u + seed + i) & 0xfffffff; b[i] = (a[i] * 40503u + i) & 0xfffffff; }
  for (int r = 0; r < ROWS / 8; r++) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
#pragma unroll
      for (int j = 0; j < 14; j++) acc[i + j] = (u64)((long long)acc[i + j] + (long long)(int)a[j] * (long long)(int)b[i]);
    }
    // both operands change every iteration (28 cheap instructions beside the 196 multiply-adds), so that no product is loop-invariant
#pragma unroll
    for (int i = 0; i < 14; i++) { a[i] = (a[i] + (u32)acc[i] + r) & 0xfffffff; b[i] = (b[i] ^ (u32)acc[i + 14]) & 0xfffffff; }
  }
  u64 s = 0; for (int i = 0; i < 28; i++) s ^= acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}

template <typename F> float time_it(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; r++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}
int main() {
  u32* d_out; CK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(u32)));
  const int waves_per_simd = 4, blocks = 256 * waves_per_simd;     // 256-thread blocks: one wave per SIMD each
  const double simd_clk = 2.4e9;
  auto per_product = [&](float ms, double products_per_wave) { return ms * 1e-3 * simd_clk / (products_per_wave * waves_per_simd); };
  for (int terms : {1, 4}) {
    float ms = time_it([&] { hipLaunchKernelGGL(k_mfma_row, dim3(blocks), dim3(256), 0, 0, d_out, 1u, terms); });
    printf("mfma_row   terms=%d  %8.3f ms  %6.2f clk per MFMA per SIMD  -> %6.2f clk per Fp product (12 useful products per 6 MFMA)\n", terms, ms, ms * 1e-3 * simd_clk / (ROWS * 6.0 * waves_per_simd), per_product(ms, ROWS * 12.0));
  }
  {
    float ms = time_it([&] { hipLaunchKernelGGL(k_toeplitz, dim3(blocks), dim3(256), 0, 0, d_out, 1u); });
    printf("toeplitz            %8.3f ms  -> %6.2f clk per Fp product (A operand formed from bytes at arbitrary offsets)\n", ms, per_product(ms, ROWS * 12.0));
  }
  {
    float ms = time_it([&] { hipLaunchKernelGGL(k_normalise, dim3(blocks), dim3(256), 0, 0, d_out, 1u); });
    printf("normalise           %8.3f ms  -> %6.2f clk per normalisation of one product's 96 columns (16 products per wave pass)\n", ms, ms * 1e-3 * simd_clk / ((ROWS / 8) * 16.0 * waves_per_simd));
  }
  {
    float ms = time_it([&] { hipLaunchKernelGGL(k_mac28, dim3(blocks), dim3(256), 0, 0, d_out, 1u); });
    printf("mac28               %8.3f ms  -> %6.2f clk per Fp product (196 v_mad_i64_i32, 64 products per wave)\n", ms, per_product(ms, (ROWS / 8) * 64.0));
  }
  return 0;
}
