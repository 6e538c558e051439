// sha256.h -- host-side SHA-256 (used per CALL for an oversize DST only, RFC 9380 5.3.3; the per-message hashing is
// xmd_kernel.hip) and a host expand_message_xmd kept as the readable statement of what the kernel computes.
// Stands for the reference's expand_message_xmd over node's crypto SHA-256 (index.ts:39-48, 207-231).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
namespace nbls {
struct Sha256 {
  uint32_t h[8]; uint8_t buf[64]; uint64_t len;
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  Sha256() { static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19}; memcpy(h, iv, 32); len = 0; }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
      0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
      0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
      0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
      0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) w[i] = w[i - 16] + (rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10));
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
      uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const uint8_t* p, size_t n) {
    while (n) { size_t off = len & 63, take = 64 - off; if (take > n) take = n; memcpy(buf + off, p, take); len += take; p += take; n -= take; if ((len & 63) == 0) block(buf); }
  }
  void final(uint8_t* out) {
    uint64_t bits = len * 8; uint8_t pad = 0x80; update(&pad, 1);
    uint8_t z = 0; while ((len & 63) != 56) update(&z, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i)); update(lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
  }
};
// out[len_in_bytes]; returns false when ell > 255 (the reference throws 'Invalid xmd length')
static inline bool expand_message_xmd(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, uint8_t* out, size_t len_in_bytes) {
  uint8_t dst_hash[32];
  if (dst_len > 255) { Sha256 c; c.update((const uint8_t*)"H2C-OVERSIZE-DST-", 17); c.update(dst, dst_len); c.final(dst_hash); dst = dst_hash; dst_len = 32; }
  size_t ell = (len_in_bytes + 31) / 32; if (ell > 255) return false;
  uint8_t dlen = (uint8_t)dst_len, zpad[64] = {0}, lib[2] = {(uint8_t)(len_in_bytes >> 8), (uint8_t)len_in_bytes}, zero = 0, b0[32], bi[32];
  { Sha256 c; c.update(zpad, 64); c.update(msg, msg_len); c.update(lib, 2); c.update(&zero, 1); c.update(dst, dst_len); c.update(&dlen, 1); c.final(b0); }
  uint8_t idx = 1;
  { Sha256 c; c.update(b0, 32); c.update(&idx, 1); c.update(dst, dst_len); c.update(&dlen, 1); c.final(bi); }
  size_t done = 0;
  for (size_t i = 1;; i++) {
    size_t take = len_in_bytes - done < 32 ? len_in_bytes - done : 32; memcpy(out + done, bi, take); done += take;
    if (done >= len_in_bytes) break;
    uint8_t x[32]; for (int k = 0; k < 32; k++) x[k] = b0[k] ^ bi[k];
    idx = (uint8_t)(i + 1);
    Sha256 c; c.update(x, 32); c.update(&idx, 1); c.update(dst, dst_len); c.update(&dlen, 1); c.final(bi);
  }
  return true;
}
}  // namespace nbls
