// lds_model.cpp -- host tool (round 5): LDS bank-conflict model of a translated step program (aot_host.cpp) on gfx950 and a search for a slot placement with
// fewer conflicts.  Model (MI355X guide, LDS section): ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, the same + 32; within a
// group every 16-byte access occupies one of 16 bank groups ((a / 16) mod 16); equal addresses broadcast; the group costs max over bank groups of the number of distinct
// addresses.  ds_read_b64 (the last 8 bytes of a slot): two 32-lane groups, 32 bank pairs ((a / 8) mod 32).
// Build: g++ -O2 -std=c++17 -I noble-bls12-381_amd/csrc -I include tools/ldsmodel/lds_model.cpp noble-bls12-381_amd/csrc/build/{trace,programs,aot_host,config}.o -o /tmp/lds_model
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <set>
#include <string>
#include <vector>
#include "aot.h"
using namespace nbls;

struct Read { uint16_t a[64]; };   // one ld14 of a step: address per physical lane
static const int G128[4][16] = {{0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27}, {4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31},
                                {32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59}, {36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63}};
static unsigned cost_read(const uint32_t* a) {   // LDS cycles of one ld14 (3 x b128 + b64) given per-lane addresses; minimum 3 * 4 + 2 = 14
  unsigned cyc = 0;
  for (int off = 0; off < 48; off += 16)
    for (int g = 0; g < 4; g++) {
      uint32_t seen[16][16]; int cnt[16] = {0};
      int mx = 1;
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

struct Layout { uint32_t C, I, stride, G, slots, junk; std::vector<std::vector<uint32_t>> pos; std::vector<uint32_t> base; uint32_t nstride; };   // pos[g][s]: position of slot s inside instance g's region
static uint32_t remap(const Layout& L, uint32_t a) {
  if (a < L.C || a >= L.junk) return a >= L.junk ? L.base[L.G] : a;   // constants stay; the junk slot moves behind the new image
  const uint32_t g = (a - L.C) / L.I, s = ((a - L.C) % L.I) / L.stride;
  return L.base[g] + L.pos[g][s] * L.nstride;
}
int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: lds_model <program name> [iterations] [new stride]\n"); return 1; }
  int id = -1;
  for (int i = 0; i < P_COUNT; i++) if (get_program((ProgId)i).name == argv[1]) id = i;
  if (id < 0) { fprintf(stderr, "no such program\n"); return 1; }
  const long iters = argc > 2 ? atol(argv[2]) : 200000;
  const Program& p = get_program((ProgId)id);
  AotProgram ap; const std::string e = aot_translate(p, ap);
  if (!e.empty()) { fprintf(stderr, "%s\n", e.c_str()); return 1; }
  std::vector<Read> reads; std::vector<unsigned> weight;
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
    }
  }
  Layout L; L.C = p.shared_consts ? p.nconst * p.slot_bytes : 0; L.I = p.inst_bytes(); L.stride = p.slot_bytes; L.G = p.G; L.slots = p.slots; L.junk = p.lds_bytes();
  if (!p.shared_consts) { fprintf(stderr, "replicated constants: not modelled\n"); return 1; }
  L.nstride = argc > 3 ? atoi(argv[3]) : p.slot_bytes;
  L.pos.assign(L.G, std::vector<uint32_t>(L.slots)); for (uint32_t g = 0; g < L.G; g++) for (uint32_t s = 0; s < L.slots; s++) L.pos[g][s] = s;
  const bool per_inst = argc > 5 && atoi(argv[5]);
  L.base.resize(L.G + 1); for (uint32_t g = 0; g <= L.G; g++) L.base[g] = L.C + g * L.slots * L.nstride;
  auto total = [&](const Layout& l) { unsigned long c = 0; uint32_t a[64]; for (const Read& r : reads) { for (int i = 0; i < 64; i++) a[i] = remap(l, r.a[i]); c += cost_read(a); } return c; };
  const unsigned long base_cycles = 14ul * reads.size();
  unsigned long cur = total(L);
  printf("%s: W %u G %u slots %u stride %u -> %u, image %u B, %zu slot reads per wavefront; identity placement: %lu LDS read cycles, %lu of them conflicts (%.3f)\n", p.name.c_str(), p.W, p.G, p.slots, p.slot_bytes, L.nstride,
         p.lds_bytes(), reads.size(), cur, cur - base_cycles, (double)(cur - base_cycles) / cur);
  // search: swap two slot positions, or shift an instance base by 16 bytes within the slack of `budget` bytes
  const uint32_t budget = argc > 4 ? atoi(argv[4]) : L.C + L.G * L.slots * L.nstride;   // bytes the image may take (without the junk slot)
  std::mt19937 rng(12345);
  Layout best = L; unsigned long bestc = cur;
  double T = 40.0;
  for (long it = 0; it < iters; it++) {
    Layout n = L;
    if (rng() % 4 || budget <= L.C + L.G * L.slots * L.nstride) { uint32_t i = rng() % L.slots, j = rng() % L.slots; if (per_inst) { uint32_t g = rng() % L.G; std::swap(n.pos[g][i], n.pos[g][j]); } else for (uint32_t g = 0; g < L.G; g++) std::swap(n.pos[g][i], n.pos[g][j]); }
    else {   // move the bases of instances g.. by +-16
      uint32_t g = 1 + rng() % L.G; int d = (rng() & 1) ? 16 : -16;
      bool ok = true;
      for (uint32_t k = g; k <= L.G; k++) n.base[k] += d;
      if (n.base[g] < n.base[g - 1] + L.slots * L.nstride || n.base[L.G] > budget) ok = false;
      if (!ok) continue;
    }
    const unsigned long c = total(n);
    if (c <= cur || std::exp((double)((long)cur - (long)c) / T) * 4294967296.0 > (double)rng()) { L = n; cur = c; if (c < bestc) { bestc = c; best = n; } }
    T = 40.0 * (1.0 - (double)it / iters) + 0.05;
  }
  printf("  best placement found: %lu cycles, %lu conflicts (%.3f); image %u B\n  pos:", bestc, bestc - base_cycles, (double)(bestc - base_cycles) / bestc, best.base[L.G]);
  for (uint32_t g = 0; g < (per_inst ? L.G : 1); g++) { for (uint32_t s = 0; s < L.slots; s++) printf(" %u", best.pos[g][s]); printf(" |"); }
  printf("\n  base:"); for (uint32_t g = 0; g <= L.G; g++) printf(" %u", best.base[g]); printf("\n");
  return 0;
}
