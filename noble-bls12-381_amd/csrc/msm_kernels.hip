// msm_kernels.hip -- data movement of the multi-scalar multiplication sum_i [k_i]P_i (bucket method).
// The group arithmetic itself runs as wave-VM step programs (P_G*_ADD_AB, P_G*_ADD2, P_G*_HORNER, P_G*_SHIFTADD); the
// kernels here only decide WHICH points meet: window digits of the scalars, a device radix sort of (window, digit) keys
// (hipCUB), and gathers by index.  Points are raw projective elements of E = 192 (G1) or 384 (G2) bytes, moved as 16-byte
// vectors, one vector per thread, consecutive threads on consecutive vectors of the same point (coalesced).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include "scalar_split.h"

namespace {
typedef uint32_t u32;
typedef uint64_t u64;
const int C = 12;   // window bits == MSM_WINDOW_BITS (programs.h)

// keys[w * n + i] = w << C | digit_w(k_i), vals[w * n + i] = i; scalars are 32-byte big-endian integers
__global__ void msm_keys_kernel(u32 n, u32 nwin, const uint8_t* __restrict__ scalars, u32* __restrict__ keys, u32* __restrict__ vals) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32* k = (const u32*)(scalars + 32ull * i);
  u32 w[9];
#pragma unroll
  for (int j = 0; j < 8; j++) w[j] = __builtin_bswap32(k[7 - j]);
  w[8] = 0;
  for (u32 win = 0; win < nwin; win++) {
    const u32 bit = C * win, word = bit >> 5, off = bit & 31;
    const u64 v = (u64)w[word] | ((u64)w[word + 1] << 32);
    const u32 digit = (u32)(v >> off) & ((1u << C) - 1);
    keys[(u64)win * n + i] = (win << C) | digit;
    vals[(u64)win * n + i] = i;
  }
}

// Scalar decomposition for the endomorphism split and its sign-aligned recoding: scalar_split.h (written once, compiled here and into the test-only simulator).
__global__ void msm_decompose_kernel(u32 n, u32 dims, const uint8_t* __restrict__ scalars, uint8_t* __restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  nbls::scalar_decompose(scalars + 32ull * i, dims, out + 32ull * dims * i);
}
__global__ void msm_sac_kernel(u32 n, const uint8_t* __restrict__ scalars, uint8_t* __restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  nbls::scalar_sac_recode(scalars + 32ull * i, out + 128ull * i);
}

// dst[j] = src[idx[j]]   (q = 16-byte vectors per element)
__global__ void msm_gather_kernel(u64 m, u32 q, const u32* __restrict__ idx, const uint4* __restrict__ src, uint4* __restrict__ dst) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * q) return;
  const u64 e = t / q; const u32 part = (u32)(t - e * q);
  dst[t] = src[(u64)idx[e] * q + part];
}

// rank of every element inside its run of equal keys: starts[j] = j at the head of a run, else 0; an inclusive max-scan (hipCUB)
// turns that into the index of the run's head, rank = j - head.  The last element of a run reports the run length; the longest
// run bounds the number of rounds of the segmented sum.
__global__ void msm_heads_flag_kernel(u64 m, const u32* __restrict__ keys, u32* __restrict__ starts) {
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  starts[j] = (j > 0 && keys[j - 1] != keys[j]) ? (u32)j : 0u;
}
__global__ void msm_rank_kernel(u64 m, const u32* __restrict__ keys, u32* __restrict__ pos /* in: head index, out: rank */, u32* __restrict__ maxrun) {
  __shared__ u32 blockmax;                       // one global atomic per workgroup (same-address atomics serialise: one per run cost 0.24 ms at 90 k runs)
  if (threadIdx.x == 0) blockmax = 0;
  __syncthreads();
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) {
    const u32 r = (u32)j - pos[j];
    pos[j] = r;
    if (j + 1 == m || keys[j + 1] != keys[j]) atomicMax(&blockmax, r + 1);
  }
  __syncthreads();
  if (threadIdx.x == 0 && blockmax) atomicMax(maxrun, blockmax);
}

// Segmented sum over the sorted list as a balanced tree inside every run: in the round with stride d the elements whose rank
// is a multiple of 2d absorb the element d places further on (when it is in the same run).  This kernel lists those
// elements (ballot + prefix counts); the addition program then addresses its operands through the list (KernelArgs.item_index).
// The order of the list is irrelevant: every pair is independent.
__global__ void __launch_bounds__(1024) msm_pairs_kernel(u64 m, u32 d, const u32* __restrict__ keys, const u32* __restrict__ pos, u32* __restrict__ list, u32* __restrict__ count) {
  // 4096 elements per workgroup, ONE global atomic per workgroup (same-address atomics serialise: one per wavefront cost 1.5 ms at 2^24 elements)
  __shared__ u32 wbase[4][16];
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u64 j0 = (u64)blockIdx.x * 4096 + tid;
  bool act[4]; u64 mask[4];
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const u64 j = j0 + it * 1024;
    act[it] = false;
    if (j < m && (pos[j] & (2 * d - 1)) == 0 && j + d < m) act[it] = keys[j + d] == keys[j];
    mask[it] = __ballot(act[it]);
    if (lane == 0) wbase[it][wave] = (u32)__popcll(mask[it]);
  }
  __syncthreads();
  if (tid == 0) {
    u32 tot = 0;
    for (int it = 0; it < 4; it++) for (int w = 0; w < 16; w++) { const u32 c = wbase[it][w]; wbase[it][w] = tot; tot += c; }
    const u32 g = tot ? atomicAdd(count, tot) : 0;
    for (int it = 0; it < 4; it++) for (int w = 0; w < 16; w++) wbase[it][w] += g;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; it++)
    if (act[it]) list[wbase[it][wave] + (u32)__popcll(mask[it] & ((1ull << lane) - 1))] = (u32)(j0 + it * 1024);
}

__global__ void msm_fill_kernel(u64 count, u32 q, const uint4* __restrict__ ident, uint4* __restrict__ dst) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * q) return;
  dst[t] = ident[t % q];
}

// buckets[key] = P[j] for the head j of every run
__global__ void msm_heads_kernel(u64 m, u32 q, const u32* __restrict__ keys, const uint4* __restrict__ P, uint4* __restrict__ buckets) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * q) return;
  const u64 e = t / q; const u32 part = (u32)(t - e * q);
  const u32 key = keys[e];
  if (e > 0 && keys[e - 1] == key) return;
  buckets[(u64)key * q + part] = P[t];
}

// sum_b b * B_b = sum_t 2^t * (sum of the buckets whose index has bit t set): G[(w * C + t) * 2^(C-1) + j] = the j-th such bucket
__global__ void msm_bitsel_kernel(u64 count, u32 q, const uint4* __restrict__ buckets, uint4* __restrict__ G) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * q) return;
  const u64 e = t / q; const u32 part = (u32)(t - e * q);
  const u32 j = (u32)(e & ((1u << (C - 1)) - 1)); const u32 grp = (u32)(e >> (C - 1));
  const u32 w = grp / C, bit = grp % C;
  const u32 b = ((j >> bit) << (bit + 1)) | (1u << bit) | (j & ((1u << bit) - 1));
  G[t] = buckets[((u64)w * (1u << C) + b) * q + part];
}

inline unsigned blocks_for(u64 threads) { return (unsigned)((threads + 255) / 256); }
}  // namespace

extern "C" {
int nbls_msm_keys_launch(unsigned n, unsigned nwin, const void* scalars, void* keys, void* vals, void* stream) {
  hipLaunchKernelGGL(msm_keys_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, n, nwin, (const uint8_t*)scalars, (u32*)keys, (u32*)vals);
  return (int)hipGetLastError();
}
int nbls_msm_decompose_launch(unsigned n, unsigned dims, const void* scalars, void* out, void* stream) {
  hipLaunchKernelGGL(msm_decompose_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, n, dims, (const uint8_t*)scalars, (uint8_t*)out);
  return (int)hipGetLastError();
}
int nbls_msm_sac_launch(unsigned n, const void* scalars, void* out, void* stream) {
  hipLaunchKernelGGL(msm_sac_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, n, (const uint8_t*)scalars, (uint8_t*)out);
  return (int)hipGetLastError();
}
// temp == NULL: returns the scratch size in *temp_bytes
int nbls_msm_sort_launch(void* temp, size_t* temp_bytes, const void* keys_in, void* keys_out, const void* vals_in, void* vals_out, size_t m, int key_bits, void* stream) {
  return (int)hipcub::DeviceRadixSort::SortPairs(temp, *temp_bytes, (const u32*)keys_in, (u32*)keys_out, (const u32*)vals_in, (u32*)vals_out, (int)m, 0, key_bits, (hipStream_t)stream);
}
int nbls_msm_gather_launch(size_t m, unsigned elem_bytes, const void* idx, const void* src, void* dst, void* stream) {
  const u32 q = elem_bytes / 16;
  hipLaunchKernelGGL(msm_gather_kernel, dim3(blocks_for((u64)m * q)), dim3(256), 0, (hipStream_t)stream, (u64)m, q, (const u32*)idx, (const uint4*)src, (uint4*)dst);
  return (int)hipGetLastError();
}
// temp == NULL: returns the scan's scratch size in *temp_bytes
int nbls_msm_rank_launch(void* temp, size_t* temp_bytes, size_t m, const void* keys, void* pos, void* maxrun_u32, void* stream) {
  if (!temp) return (int)hipcub::DeviceScan::InclusiveScan(nullptr, *temp_bytes, (const u32*)pos, (u32*)pos, hipcub::Max(), (int)m, (hipStream_t)stream);
  hipMemsetAsync(maxrun_u32, 0, 4, (hipStream_t)stream);
  hipLaunchKernelGGL(msm_heads_flag_kernel, dim3(blocks_for(m)), dim3(256), 0, (hipStream_t)stream, (u64)m, (const u32*)keys, (u32*)pos);
  int e = (int)hipcub::DeviceScan::InclusiveScan(temp, *temp_bytes, (const u32*)pos, (u32*)pos, hipcub::Max(), (int)m, (hipStream_t)stream);
  if (e) return e;
  hipLaunchKernelGGL(msm_rank_kernel, dim3(blocks_for(m)), dim3(256), 0, (hipStream_t)stream, (u64)m, (const u32*)keys, (u32*)pos, (u32*)maxrun_u32);
  return (int)hipGetLastError();
}
int nbls_msm_pairs_launch(size_t m, unsigned d, const void* keys, const void* pos, void* list, void* count_u32, void* stream) {
  hipMemsetAsync(count_u32, 0, 4, (hipStream_t)stream);
  hipLaunchKernelGGL(msm_pairs_kernel, dim3((unsigned)((m + 4095) / 4096)), dim3(1024), 0, (hipStream_t)stream, (u64)m, d, (const u32*)keys, (const u32*)pos, (u32*)list, (u32*)count_u32);
  return (int)hipGetLastError();
}
int nbls_msm_fill_launch(size_t count, unsigned elem_bytes, const void* ident, void* dst, void* stream) {
  const u32 q = elem_bytes / 16;
  hipLaunchKernelGGL(msm_fill_kernel, dim3(blocks_for((u64)count * q)), dim3(256), 0, (hipStream_t)stream, (u64)count, q, (const uint4*)ident, (uint4*)dst);
  return (int)hipGetLastError();
}
int nbls_msm_heads_launch(size_t m, unsigned elem_bytes, const void* keys, const void* P, void* buckets, void* stream) {
  const u32 q = elem_bytes / 16;
  hipLaunchKernelGGL(msm_heads_kernel, dim3(blocks_for((u64)m * q)), dim3(256), 0, (hipStream_t)stream, (u64)m, q, (const u32*)keys, (const uint4*)P, (uint4*)buckets);
  return (int)hipGetLastError();
}
int nbls_msm_bitsel_launch(unsigned nwin, unsigned elem_bytes, const void* buckets, void* G, void* stream) {
  const u32 q = elem_bytes / 16;
  const u64 count = (u64)nwin * C << (C - 1);
  hipLaunchKernelGGL(msm_bitsel_kernel, dim3(blocks_for(count * q)), dim3(256), 0, (hipStream_t)stream, count, q, (const uint4*)buckets, (uint4*)G);
  return (int)hipGetLastError();
}
}
