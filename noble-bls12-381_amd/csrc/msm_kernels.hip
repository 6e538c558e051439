// msm_kernels.hip -- data movement of the multi-scalar multiplication sum_i [k_i]P_i (bucket method).
// The group arithmetic itself runs as wave-VM step programs (P_G*_ADD_AB, P_G*_ADD2, P_G*_HORNER, P_G*_SHIFTADD); the
// kernels here only decide WHICH points meet: window digits of the scalars, a device radix sort of (window, digit) keys
// (hipCUB), and gathers by index.  Points are raw projective elements of E = 192 (G1) or 384 (G2) bytes, moved as 16-byte
// vectors, one vector per thread, consecutive threads on consecutive vectors of the same point (coalesced).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

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

// Scalar decomposition for the endomorphism split: k = a0 + a1 |z| + a2 |z|^2 + a3 |z|^3 in base |z| = 0xd201000000010000 (the
// BLS parameter; a0..a2 < 2^64, a3 < 2^65 for k < 2^256).  dims = 4 (G2): the four digits; dims = 2 (G1): k mod z^2 = a0 + a1 |z|
// and k div z^2 = a2 + a3 |z|.  Output: dims scalars per input scalar, 32 bytes big-endian each (the format msm_keys_kernel reads).
__device__ inline u64 div_step(u64* limbs, int nl) {   // limbs (little-endian u64) /= |z|, returns the remainder; bitwise, |z| has its top bit set
  const u64 Z = 0xd201000000010000ull;
  u64 rem = 0;
  for (int i = nl - 1; i >= 0; i--) {
    u64 q = 0; const u64 v = limbs[i];
    for (int b = 63; b >= 0; b--) {
      const u64 carry = rem >> 63;
      rem = (rem << 1) | ((v >> b) & 1);
      const u64 ge = (u64)(carry | (u64)(rem >= Z));      // branch-free: sign's ladder feeds SECRET scalars through this division (round 5)
      rem -= Z & (0 - ge);
      q = (q << 1) | ge;
    }
    limbs[i] = q;
  }
  return rem;
}
__device__ inline void store_be(uint8_t* out, u64 l0, u64 l1, u64 l2) {   // 32-byte big-endian of l0 + l1 2^64 + l2 2^128
  u32* o = (u32*)out;
  o[0] = 0; o[1] = 0; o[2] = __builtin_bswap32((u32)(l2 >> 32)); o[3] = __builtin_bswap32((u32)l2);
  o[4] = __builtin_bswap32((u32)(l1 >> 32)); o[5] = __builtin_bswap32((u32)l1); o[6] = __builtin_bswap32((u32)(l0 >> 32)); o[7] = __builtin_bswap32((u32)l0);
}
__global__ void msm_decompose_kernel(u32 n, u32 dims, const uint8_t* __restrict__ scalars, uint8_t* __restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32* k = (const u32*)(scalars + 32ull * i);
  u64 l[4];
#pragma unroll
  for (int j = 0; j < 4; j++) l[j] = ((u64)__builtin_bswap32(k[6 - 2 * j]) << 32) | __builtin_bswap32(k[7 - 2 * j]);
  const u64 a0 = div_step(l, 4);
  const u64 a1 = div_step(l, 4);          // l = k div z^2 (< 2^129)
  uint8_t* o = out + 32ull * dims * i;
  if (dims == 2) {
    const u64 Z = 0xd201000000010000ull;
    const u64 lo = a1 * Z, hi = __umul64hi(a1, Z);
    const u64 s0 = lo + a0, s1 = hi + (s0 < lo);
    store_be(o, s0, s1, 0);
    store_be(o + 32, l[0], l[1], l[2]);
  } else {
    const u64 a2 = div_step(l, 3);          // l = a3 (< 2^65)
    store_be(o, a0, 0, 0); store_be(o + 32, a1, 0, 0); store_be(o + 64, a2, 0, 0); store_be(o + 96, l[0], l[1], 0);
  }
}

// The four digits recoded SIGN-ALIGNED for the one-addition-per-bit ladder of sign (codec.h pt_mul_sac_g2; Faz-Hernandez, Longa, Sanchez 2013): with a0 made odd (a0 + 1 when even:
// the ladder subtracts Q again), a0 = sum_i s_i 2^i over 66 digits s_i = +-1 with s_i = 2 bit_(i+1)(a0) - 1 and s_65 = +1; every other digit is rewritten over the same signs,
// a_j = sum_i s_i e_ji 2^i with e_ji = a_j mod 2 and a_j <- (a_j >> 1) + (e_ji and s_i = -1).  Output per scalar, 4 x 32 bytes big-endian: [bits 0..65: s_i = +1, bit 66: a0 was even],
// then the bits e_1i, e_2i, e_3i.  Branch-free: the scalars are secret keys.
__global__ void msm_sac_kernel(u32 n, const uint8_t* __restrict__ scalars, uint8_t* __restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32* k = (const u32*)(scalars + 32ull * i);
  u64 l[4];
#pragma unroll
  for (int j = 0; j < 4; j++) l[j] = ((u64)__builtin_bswap32(k[6 - 2 * j]) << 32) | __builtin_bswap32(k[7 - 2 * j]);
  u64 alo[4], ahi[4];
  alo[0] = div_step(l, 4); alo[1] = div_step(l, 4); alo[2] = div_step(l, 3); alo[3] = l[0];
  ahi[0] = ahi[1] = ahi[2] = 0; ahi[3] = l[1];
  const u64 even = (alo[0] & 1) ^ 1;
  alo[0] |= 1;
  // signs: bit i of (slo, shi) set <=> s_i = +1
  const u64 slo = alo[0] >> 1, shi = 2;          // bits 63, 64 clear (a0 < 2^64), bit 65 set
  uint8_t* o = out + 128ull * i;
  store_be(o, slo, shi | (even << 2), 0);
  for (int j = 1; j < 4; j++) {
    u64 lo = alo[j], hi = ahi[j], elo = 0, ehi = 0;
    for (int b = 0; b < 66; b++) {
      const u64 e = lo & 1;
      const u64 sp = b < 64 ? (slo >> b) & 1 : (shi >> (b - 64)) & 1;
      if (b < 64) elo |= e << b; else ehi |= e << (b - 64);
      lo = (lo >> 1) | (hi << 63); hi >>= 1;
      const u64 inc = e & (sp ^ 1);
      lo += inc; hi += (u64)(lo < inc);
    }
    store_be(o + 32 * j, elo, ehi, 0);
  }
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
