// aot_layout.cpp -- see aot_layout.h
#include "aot_layout.h"
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <random>

namespace nbls {
namespace {

struct Read { uint16_t a[64]; };   // one 14-limb read of a step: LDS byte address per physical lane (identity placement)
const int G128[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                         {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
// LDS cycles of one slot read (ld14 = ds_read_b128 at +0, +16, +32 and ds_read_b64 at +48) given the per-lane addresses; 3 * 4 + 2 = 14 without conflicts
unsigned cost_read(const uint32_t* a) {
  unsigned cyc = 0;
  for (int off = 0; off < 48; off += 16)
    for (int g = 0; g < 4; g++) {
      uint32_t seen[16][16]; int cnt[16] = {0}; int mx = 1;
      for (int i = 0; i < 16; i++) {
        const uint32_t ad = a[G128[g][i]] + off; const int bg = (ad >> 4) & 15;
        bool dup = false; for (int k = 0; k < cnt[bg]; k++) if (seen[bg][k] == ad) { dup = true; break; }
        if (!dup) { seen[bg][cnt[bg]++] = ad; if (cnt[bg] > mx) mx = cnt[bg]; }
      }
      cyc += mx;
    }
  for (int g = 0; g < 2; g++) {
    uint32_t seen[32][32]; int cnt[32] = {0}; int mx = 1;
    for (int i = 0; i < 32; i++) {
      const uint32_t ad = a[32 * g + i] + 48; const int bp = (ad >> 3) & 31;
      bool dup = false; for (int k = 0; k < cnt[bp]; k++) if (seen[bp][k] == ad) { dup = true; break; }
      if (!dup) { seen[bp][cnt[bp]++] = ad; if (cnt[bp] > mx) mx = cnt[bp]; }
    }
    cyc += mx;
  }
  return cyc;
}
// every slot read of one wavefront, from the identity translation
std::vector<Read> collect_reads(const Program& p) {
  std::vector<Read> reads;
  AotProgram ap;
  if (!aot_translate_with(p, ap, nullptr).empty()) return reads;
  auto word = [&](const AotStep& st, uint32_t lane, uint32_t i) { return ap.descs[((size_t)st.y + (i / 4) * 64 + lane) * 4 + (i & 3)]; };
  for (const AotStep& st : ap.steps) {
    const AotSig& sg = ap.sigs[st.x & 0xff];
    auto add = [&](auto f) { Read r; for (int l = 0; l < 64; l++) r.a[l] = (uint16_t)f(l); reads.push_back(r); };
    if (sg.kind == K_DOT) {
      const uint32_t HW = 4 * aot_dot_hdr_quads(sg.t);
      for (uint32_t t = 0; t < sg.t; t++) add([&](int l) { return word(st, l, AOT_DOT_HDR + t) & 0xffffu; });
      for (uint32_t r = 0; r < sg.p0; r++) {
        const uint32_t shape = ((r < 4 ? sg.sh0 : sg.sh1) >> (8 * (r & 3))) & 0xffu, sa = shape & 3u, sb = (shape >> SH_B_SHIFT) & 3u;
        add([&](int l) { return word(st, l, HW + 4 * r + 0); });
        if (sa) add([&](int l) { return word(st, l, HW + 4 * r + 1); });
        add([&](int l) { return word(st, l, HW + 4 * r + 2); });
        if (sb) add([&](int l) { return word(st, l, HW + 4 * r + 3); });
      }
    } else if (sg.kind == K_LIN) {
      for (uint32_t t = 0; t < sg.p0 + sg.t; t++) add([&](int l) { return (word(st, l, 1 + t / 2) >> (16 * (t & 1))) & 0xffffu; });
    } else if (sg.kind == K_STORE || sg.kind == K_STOREW) {
      add([&](int l) { return word(st, l, 0) & 0xffffu; });
    }
  }
  return reads;
}
struct Geo { uint32_t C, I, stride, G, slots, end; };
Geo geo_of(const Program& p) { return {p.shared_consts ? p.nconst * p.slot_bytes : 0u, p.inst_bytes(), p.slot_bytes, p.G, p.slots, p.lds_bytes()}; }
inline uint32_t place(const Geo& q, const AotLayout& l, uint32_t a) {
  if (a < q.C || a >= q.end) return a;
  const uint32_t g = (a - q.C) / q.I, s = ((a - q.C) % q.I) / q.stride;
  return q.C + g * q.I + (uint32_t)l.pos[g][s] * q.stride;
}
AotLayout identity(const Geo& q) { AotLayout l; l.pos.assign(q.G, std::vector<uint16_t>(q.slots)); for (uint32_t g = 0; g < q.G; g++) for (uint32_t s = 0; s < q.slots; s++) l.pos[g][s] = (uint16_t)s; return l; }

}  // namespace

AotLdsCost aot_layout_cost(const Program& p, const AotLayout& l) {
  const std::vector<Read> reads = collect_reads(p);
  const Geo q = geo_of(p);
  const AotLayout id = identity(q);
  const AotLayout& use = l.empty() ? id : l;
  unsigned long c = 0; uint32_t a[64];
  for (const Read& r : reads) { for (int i = 0; i < 64; i++) a[i] = place(q, use, r.a[i]); c += cost_read(a); }
  return {c, 14ul * reads.size()};
}

AotLayout aot_layout_search(const Program& p, long iterations, unsigned seed) {
  const Geo q = geo_of(p);
  if (!p.shared_consts || q.slots < 2 || q.G == 0) return AotLayout();
  const std::vector<Read> reads = collect_reads(p);
  if (reads.empty()) return AotLayout();
  // reads that touch (g, s)
  std::vector<std::vector<uint32_t>> touch((size_t)q.G * q.slots);
  for (size_t k = 0; k < reads.size(); k++) {
    std::vector<uint32_t> seen;
    for (int i = 0; i < 64; i++) {
      const uint32_t a = reads[k].a[i];
      if (a < q.C || a >= q.end) continue;
      const uint32_t key = ((a - q.C) / q.I) * q.slots + ((a - q.C) % q.I) / q.stride;
      if (std::find(seen.begin(), seen.end(), key) == seen.end()) { seen.push_back(key); touch[key].push_back((uint32_t)k); }
    }
  }
  AotLayout cur = identity(q);
  std::vector<unsigned> rc(reads.size());
  uint32_t a[64];
  auto eval = [&](size_t k) { for (int i = 0; i < 64; i++) a[i] = place(q, cur, reads[k].a[i]); return cost_read(a); };
  unsigned long total = 0;
  for (size_t k = 0; k < reads.size(); k++) { rc[k] = eval(k); total += rc[k]; }
  AotLayout best = cur; unsigned long best_total = total;
  std::mt19937 rng(seed);
  std::vector<uint32_t> aff; std::vector<unsigned> saved;
  for (long it = 0; it < iterations; it++) {
    const double T = 3.0 * (1.0 - (double)it / (double)iterations) + 0.02;
    const uint32_t g = rng() % q.G, i = rng() % q.slots, j = rng() % q.slots;
    if (i == j) continue;
    aff = touch[(size_t)g * q.slots + i];
    for (uint32_t k : touch[(size_t)g * q.slots + j]) if (std::find(aff.begin(), aff.end(), k) == aff.end()) aff.push_back(k);
    if (aff.empty()) continue;
    std::swap(cur.pos[g][i], cur.pos[g][j]);
    saved.resize(aff.size());
    long delta = 0;
    for (size_t x = 0; x < aff.size(); x++) { saved[x] = rc[aff[x]]; const unsigned c = eval(aff[x]); delta += (long)c - (long)saved[x]; rc[aff[x]] = c; }
    const bool accept = delta <= 0 || (double)(rng() >> 8) / 16777216.0 < std::exp(-(double)delta / T);
    if (accept) { total = (unsigned long)((long)total + delta); if (total < best_total) { best_total = total; best = cur; } }
    else { std::swap(cur.pos[g][i], cur.pos[g][j]); for (size_t x = 0; x < aff.size(); x++) rc[aff[x]] = saved[x]; }
  }
  return best;
}

uint32_t aot_program_hash(const Program& p) {
  uint32_t h = 2166136261u;
  auto mix = [&](const void* d, size_t n) { const unsigned char* b = (const unsigned char*)d; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 16777619u; } };
  const uint32_t hdr[6] = {p.W, p.G, p.slots, p.nconst, p.slot_bytes, (uint32_t)p.lsplit};
  mix(hdr, sizeof hdr);
  if (!p.steps.empty()) mix(p.steps.data(), p.steps.size() * sizeof(Step));
  if (!p.descs.empty()) mix(p.descs.data(), p.descs.size() * 4);
  return h;
}
const AotLayout* aot_layout_for(const Program& p) {
  static std::mutex mu;
  static std::map<std::string, AotLayout> cache;
  std::lock_guard<std::mutex> g(mu);
  auto it = cache.find(p.name);
  if (it != cache.end()) return it->second.empty() ? nullptr : &it->second;
  AotLayout l;
  size_t n = 0; const AotLayoutEntry* t = aot_layout_table(&n);
  for (size_t k = 0; k < n; k++) {
    if (p.name != t[k].name) continue;
    // a table made for another compilation of the program (the host compiler changed and the table was not regenerated) is ignored: the compiled placement is always valid
    if (t[k].G != p.G || t[k].slots != p.slots || t[k].nsteps != p.steps.size() || !p.shared_consts || t[k].hash != aot_program_hash(p)) break;
    l.pos.assign(p.G, std::vector<uint16_t>(p.slots));
    bool ok = true;
    for (uint32_t gi = 0; gi < p.G && ok; gi++) {
      std::vector<bool> used(p.slots, false);
      for (uint32_t s = 0; s < p.slots; s++) { const uint16_t v = t[k].pos[(size_t)gi * p.slots + s]; if (v >= p.slots || used[v]) { ok = false; break; } used[v] = true; l.pos[gi][s] = v; }
    }
    if (!ok) l = AotLayout();
    break;
  }
  auto& slot = cache[p.name]; slot = l;
  return slot.empty() ? nullptr : &slot;
}

}  // namespace nbls
