// Microbenchmark of the engine's own limb routines (vm_exec.h) as dependent chains per lane, at 1/2/4/8 waves per SIMD:
// mont_mul12 (fused CIOS), wide_mac (24-word product + accumulate) and wide_redc.  Tells latency-bound (1 wave/SIMD)
// from issue-bound cost per routine.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../noble-bls12-381_amd/csrc/vm_exec.h"
using namespace nbls;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int V>
__global__ void __launch_bounds__(64) k(u32* out, const u32* in, int iters) {
  u32 a[12], b[12];
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 12; i++) { a[i] = in[i] ^ (tid * 0x01000193u); b[i] = in[12 + i] + tid; }
  a[11] &= 0x1fffffff; b[11] &= 0x1fffffff;
  const u32 P2[12] = NBLS_2P32;
  for (int it = 0; it < iters; it++) {
    u32 r[13];
    if (V == 0) { mont_mul12(r, a, b); csub<12>(r, P2); }
    if (V == 1) { u32 acc[25]; for (int i = 0; i < 25; i++) acc[i] = 0; wide_mac(acc, a, b); wide_redc(r, acc); csub<12>(r, P2); }
    if (V == 2) { u32 acc[25]; for (int i = 0; i < 25; i++) acc[i] = 0; wide_mac(acc, a, b); wide_mac(acc, b, a); wide_mac(acc, a, a); wide_mac(acc, b, b); wide_redc(r, acc); csub<12>(r, P2); r[11] &= 0x1fffffff; }
    if (V == 3) { LazyAcc L; lazy_zero(L); lazy_mac(L, a, b); u32 acc[25]; lazy_normalize(acc, L); wide_redc(r, acc); csub<12>(r, P2); }
    if (V == 4) { LazyAcc L; lazy_zero(L); lazy_mac(L, a, b); lazy_mac(L, b, a); lazy_mac(L, a, a); lazy_mac(L, b, b); u32 acc[25]; lazy_normalize(acc, L); wide_redc(r, acc); csub<12>(r, P2); r[11] &= 0x1fffffff; }
#pragma unroll
    for (int i = 0; i < 12; i++) { b[i] = a[i]; a[i] = r[i]; }
  }
  u32 s = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) s = s * 31 + a[i];
  out[tid] = s;
}
template <int V>
int run(const char* name, int wavesPerSimd, u32* d_out, u32* d_in, double units) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = 1024 * wavesPerSimd, iters = 500;
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, 10);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-22s waves/SIMD=%d %8.3f ms  wave-latency per iteration %8.0f clk ; per 144-MAD unit: %6.0f clk latency, %6.0f clk SIMD-issue\n", name, wavesPerSimd, best,
         best * 1e-3 * 2.4e9 / iters, best * 1e-3 * 2.4e9 / iters / units, best * 1e-3 * 2.4e9 / iters / units / wavesPerSimd);
  return 0;
}
int main() {
  u32 *d_out, *d_in; CK(hipMalloc(&d_out, 1024 * 8 * 64 * 4)); CK(hipMalloc(&d_in, 24 * 4));
  u32 h[24]; for (int i = 0; i < 24; i++) h[i] = 0x9e3779b9u * (i + 1);
  CK(hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice));
  // correctness: lazy variants must reproduce the carry-chain variants bit for bit on every lane
  {
    u32 *o1, *o2; CK(hipMalloc(&o1, 4096 * 64 * 4)); CK(hipMalloc(&o2, 4096 * 64 * 4));
    std::vector<u32> h1(4096 * 64), h2(4096 * 64);
    hipLaunchKernelGGL(k<1>, dim3(4096), dim3(64), 0, 0, o1, d_in, 200); hipLaunchKernelGGL(k<3>, dim3(4096), dim3(64), 0, 0, o2, d_in, 200);
    CK(hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < h1.size(); i++) bad += h1[i] != h2[i];
    printf("lazy_mac vs wide_mac (1 product, 200 chained iterations, %zu lanes): %zu mismatches\n", h1.size(), bad);
    hipLaunchKernelGGL(k<2>, dim3(4096), dim3(64), 0, 0, o1, d_in, 200); hipLaunchKernelGGL(k<4>, dim3(4096), dim3(64), 0, 0, o2, d_in, 200);
    CK(hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost));
    bad = 0; for (size_t i = 0; i < h1.size(); i++) bad += h1[i] != h2[i];
    printf("lazy_mac vs wide_mac (4 products): %zu mismatches\n", bad);
    // single wave per SIMD, back-to-back issue (worst case for hazards)
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(64), 0, 0, o1, d_in, 2000); hipLaunchKernelGGL(k<3>, dim3(256), dim3(64), 0, 0, o2, d_in, 2000);
    CK(hipMemcpy(h1.data(), o1, 256 * 64 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, 256 * 64 * 4, hipMemcpyDeviceToHost));
    bad = 0; for (size_t i = 0; i < 256 * 64; i++) bad += h1[i] != h2[i];
    printf("sparse launch (1 wave per CU), 2000 iterations: %zu mismatches\n", bad);
  }
  for (int w : {1, 2, 4}) { run<1>("wide_mac+redc", w, d_out, d_in, 2); run<3>("lazy_mac+redc", w, d_out, d_in, 2); run<2>("4x wide_mac + redc", w, d_out, d_in, 5); run<4>("4x lazy_mac + redc", w, d_out, d_in, 5); }
  return 0;
}
