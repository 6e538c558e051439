// xmd_kernel.hip -- expand_message_xmd (RFC 9380 section 5.3.1) with SHA-256 on the GPU, one message per lane: the
// hash_to_field front end of PointG2.hashToCurve (reference index.ts:207-231 expand_message_xmd, 39-48 sha256).
// Produces the 256 uniform bytes per message that H2C_A consumes (count = 2, m = 2, L = 64; index.ts:239-267), so a
// verifyBatch / sign call needs no host pre-pass over the messages.  Bytes are fed one at a time into a block buffer in LDS (DevSha below).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbls {
typedef uint32_t u32;

__constant__ u32 SHA_K[64] = {
  0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
  0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
  0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
  0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};

// The 64-byte block under construction lives in LDS, word-interleaved over the lanes (word w of lane l at wbuf[w][l]: every lane on its own bank): a byte is ONE ds_write_b8 at
// a computed address.  Round 5: the register-resident block of round 2 needed a 16-way predicated OR per byte to avoid dynamic register indexing -- 48 k of the 76 k instructions
// of one message's expansion, 0.18 ms in front of every single verify / sign.
struct DevSha {
  u32 h[8], n;
  u32 (*wb)[64];     // the workgroup's block buffer
  u32 lane;
  __device__ static u32 rotr(u32 x, int k) { return (x >> k) | (x << (32 - k)); }
  __device__ void clear() {
#pragma unroll
    for (int i = 0; i < 16; i++) wb[i][lane] = 0;
  }
  __device__ void init() {
    h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a; h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
    n = 0;
    clear();
  }
  __device__ void compress() {
    u32 s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = wb[i][lane];
    rounds(s);
    clear();
  }
  // one block given as sixteen words in registers (the fixed-layout blocks of b_1 .. b_ell: no trip through the byte buffer)
  __device__ void rounds(u32* s) {
    u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
      if (i >= 16) {
        const u32 w15 = s[(i + 1) & 15], w2 = s[(i + 14) & 15];
        s[i & 15] += (rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3)) + s[(i + 9) & 15] + (rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10));
      }
      const u32 t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + s[i & 15];
      const u32 t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  __device__ void byte(u32 v) {
    const u32 pos = n & 63;
    ((uint8_t*)&wb[pos >> 2][lane])[3 - (pos & 3)] = (uint8_t)v;     // big-endian byte (pos & 3) of a word read back as a little-endian dword
    n++;
    if ((n & 63) == 0) compress();
  }
  __device__ void word(u32 v) {
    if ((n & 3) == 0) { wb[(n & 63) >> 2][lane] = v; n += 4; if ((n & 63) == 0) compress(); }
    else { byte(v >> 24); byte((v >> 16) & 0xff); byte((v >> 8) & 0xff); byte(v & 0xff); }
  }
  __device__ void finish(u32* out8) {
    const u32 bits = n * 8;     // messages are far below 512 MB
    byte(0x80);
    // the padding zeros are in the buffer already (init / compress clear it): only the position moves (round 6: up to 55 byte operations before)
    if ((n & 63) > 56) { n = (n | 63u) + 1u; compress(); }
    n = (n & ~63u) + 56u;
    word(0); word(bits);
#pragma unroll
    for (int i = 0; i < 8; i++) out8[i] = h[i];
  }
};

// out[len * i ..] = expand_message_xmd(msg_i, DST, len), len = 64, 128 or 256 (hash_to_field for G1 encode / G1 hash, G2 encode / G2 hash);
// dst_len <= 255 (longer DSTs are pre-hashed by the caller, RFC 9380 5.3.3)
extern "C" __global__ void __launch_bounds__(64) nbls_xmd_kernel(unsigned n, const uint8_t* __restrict__ msgs, const u32* __restrict__ offsets,
                                                                 const uint8_t* __restrict__ dst, unsigned dst_len, uint8_t* __restrict__ out, unsigned len, u32* __restrict__ bad) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* m = msgs + offsets[i];
  u32 mlen = offsets[i + 1] - offsets[i];
  // offsets that are not monotonic would make this a ~4 GB read: hash an empty message instead and tell the caller (device-resident offsets cannot be checked on the host)
  if (offsets[i + 1] < offsets[i]) { mlen = 0; if (bad) atomicOr(bad, 1u); }
  __shared__ u32 wbuf[16][64];
  DevSha c;
  c.wb = wbuf; c.lane = threadIdx.x;
  u32 b0[8], bi[8];
  // b_0 = H(Z_pad || msg || l_i_b_str || 0 || DST_prime)
  c.init();
  // Z_pad is one whole block of zeros: the state behind it is a constant (round 6: one compression less per message)
  c.h[0] = 0xda5698be; c.h[1] = 0x17b9b469; c.h[2] = 0x62335799; c.h[3] = 0x779fbeca; c.h[4] = 0x8ce5d491; c.h[5] = 0xc0d26243; c.h[6] = 0xbafef9ea; c.h[7] = 0x1837a9d8;
  c.n = 64;
  for (u32 k = 0; k < mlen; k++) c.byte(m[k]);
  c.byte(len >> 8); c.byte(len & 0xff); c.byte(0x00);           // I2OSP(len_in_bytes, 2), then I2OSP(0, 1)
  for (u32 k = 0; k < dst_len; k++) c.byte(dst[k]);
  c.byte(dst_len);
  c.finish(b0);
  // b_1 = H(b_0 || 1 || DST_prime) ; b_j = H((b_0 xor b_(j-1)) || j || DST_prime)
  // Round 6: for DSTs of at most 85 bytes (the ciphersuites' are 43) these inputs are one or two blocks whose layout does not depend on the message: eight state words, then
  // j || DST_prime || 0x80 || zeros || bit length -- the same bytes for every j but the first.  The tail is laid out ONCE (24 words in the lane's LDS column), and every b_j is
  // two compressions from registers: the ~100 byte operations per j (0.06 of the 0.13 ms in front of every single verify / sign) are gone.
  if (dst_len <= 85u) {
    __shared__ u32 tail[24][64];
    const u32 lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 24; k++) tail[k][lane] = 0;
    auto put = [&](u32 pos, u32 v) { ((uint8_t*)&tail[pos >> 2][lane])[3 - (pos & 3)] = (uint8_t)v; };      // pos: byte offset behind the eight state words
    for (u32 k = 0; k < dst_len; k++) put(1 + k, dst[k]);
    put(1 + dst_len, dst_len);
    put(2 + dst_len, 0x80);
    const bool two = dst_len > 21u;                                                                        // 34 + dst_len bytes + 0x80 + the 8-byte length: one block up to 64
    tail[two ? 23 : 7][lane] = (32u + 2u + dst_len) * 8u;                                                  // the bit length, big-endian in the last word of the last block
    for (u32 j = 1; j <= len / 32; j++) {
      u32 s1[16], s2[16];
#pragma unroll
      for (int k = 0; k < 8; k++) { s1[k] = j == 1 ? b0[k] : (b0[k] ^ bi[k]); s1[8 + k] = tail[k][lane]; }
      s1[8] |= j << 24;
#pragma unroll
      for (int k = 0; k < 16; k++) s2[k] = tail[8 + k][lane];
      c.h[0] = 0x6a09e667; c.h[1] = 0xbb67ae85; c.h[2] = 0x3c6ef372; c.h[3] = 0xa54ff53a; c.h[4] = 0x510e527f; c.h[5] = 0x9b05688c; c.h[6] = 0x1f83d9ab; c.h[7] = 0x5be0cd19;
      c.rounds(s1); if (two) c.rounds(s2);
#pragma unroll
      for (int k = 0; k < 8; k++) bi[k] = c.h[k];
      u32* o = (u32*)(out + (size_t)len * i + 32 * (j - 1));
#pragma unroll
      for (int k = 0; k < 8; k++) o[k] = __builtin_bswap32(bi[k]);
    }
    return;
  }
  for (u32 j = 1; j <= len / 32; j++) {
    c.init();
    for (int k = 0; k < 8; k++) c.word(j == 1 ? b0[k] : (b0[k] ^ bi[k]));
    c.byte(j);
    for (u32 k = 0; k < dst_len; k++) c.byte(dst[k]);
    c.byte(dst_len);
    c.finish(bi);
    u32* o = (u32*)(out + (size_t)len * i + 32 * (j - 1));
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = __builtin_bswap32(bi[k]);
  }
}
}  // namespace nbls

extern "C" int nbls_xmd_launch(unsigned n, const void* msgs, const void* offsets, const void* dst, unsigned dst_len, void* out, unsigned len_in_bytes, void* bad_flag, void* stream) {   // bad_flag: NULL, or a device word that is set when an offset pair is not monotonic
  if (n == 0) return 0;
  hipLaunchKernelGGL(nbls::nbls_xmd_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, (const uint8_t*)msgs, (const nbls::u32*)offsets, (const uint8_t*)dst, dst_len, (uint8_t*)out, len_in_bytes, (nbls::u32*)bad_flag);
  return (int)hipGetLastError();
}
