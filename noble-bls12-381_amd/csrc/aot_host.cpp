// aot_host.cpp -- translation of a compiled step program (trace.h Program: step headers + per-lane descriptors with offsets relative to the instance
// region) into the form the ahead-of-time kernels execute (aot.h): step signatures and per-PHYSICAL-lane descriptors with absolute LDS addresses.
// Host code shared by aot_gen (build time: which signatures exist), libnbls.so (upload) and the test-only simulator.
#include "aot.h"
#include "aot_layout.h"
#include "config.h"
#include <map>

namespace nbls {

std::string aot_translate(const Program& p, AotProgram& out) {
  static const bool placed = env_long("NBLS_LDS_LAYOUT", 1) != 0;
  return aot_translate_with(p, out, placed ? aot_layout_for(p) : nullptr);
}
std::string aot_translate_with(const Program& p, AotProgram& out, const AotLayout* layout) {
  out = AotProgram();
  const u32 S = p.lsplit;   // lane split: a K_DOT lane-op has one descriptor per sub-lane (index = the physical lane), every other kind one per logical lane, executed by sub-lane 0
  if (S != 1 && S != 2 && S != 4) return p.name + ": lane split " + std::to_string(S) + " has no ahead-of-time body";
  if (p.lds_bytes() + 64 > 65536) return p.name + ": LDS image above 64 KB (16-bit address fields)";
  const u32 junk = p.lds_bytes();             // one slot behind the image: destination of idle lanes
  out.lds_bytes = p.lds_bytes() + 64;
  const u32 zero = 0;                         // constant 0 of instance 0 (replicated constants) or of the shared copy: the zero element (Builder::Builder)
  auto abs_addr = [&](u32 f, u32 g) -> u32 {  // vm_exec.h term_addr, resolved for instance g
    f &= 0xffffu;
    if (p.shared_consts) {
      if (!(f & 2u)) return f;                       // a constant of the shared copy
      f -= 2u;
      if (layout) { const u32 sl = f / p.slot_bytes; f = (u32)layout->pos[g][sl] * p.slot_bytes + (f - sl * p.slot_bytes); }   // the slot's place inside its instance region (aot_layout.h)
      return p.inst_base(g) + f;
    }
    return p.inst_base(g) + f;
  };
  for (size_t s = 0; s < p.steps.size(); s++) {
    const Step& st = p.steps[s];
    AotSig sg{st.kind, 0, 0, 0, 0, 0};
    std::vector<std::vector<u32>> lane_words(64);
    auto old_desc = [&](u32 lane_in) { return p.descs.data() + st.desc_off + (size_t)lane_in * st.stride; };
    // li: index of the lane's descriptor in the compiled program; sub: its sub-lane (0 when the program is not split)
    auto active = [&](u32 lane, u32& g, u32& li) {
      g = lane / p.W; const u32 phys = lane - g * p.W;
      if (g >= p.G) return false;
      if (st.kind == K_DOT) { li = phys; return phys < (u32)st.nlanes * S; }
      li = phys / S; return phys % S == 0 && li < st.nlanes;
    };
    switch (st.kind) {
      case K_DOT: {
        const u32 nadd = st.lin & 7, nsub = (st.lin >> 4) & 7;
        // merged post-added terms of every lane-op: slot -> coefficient
        const u32 ndesc = (u32)st.nlanes * S;
        std::vector<std::vector<std::pair<u32, int>>> post(ndesc);
        u32 T = 0, flags = 0;
        for (u32 li = 0; li < ndesc; li += S) {   // sub-lane 0 of every lane-op carries its destination, multiplier, bias and post-added terms
          const u32* d = old_desc(li);
          std::map<u32, int> m;
          for (u32 t = 0; t < nadd + nsub; t++) { const u32 f = (d[4 + t / 2] >> (16 * (t & 1))) & 0xffffu; if (f != 0) m[f] += t < nadd ? 1 : -1; }
          for (auto& kv : m) if (kv.second) post[li].push_back({kv.first, kv.second});
          T = std::max<u32>(T, (u32)post[li].size());
          const u32 mult = (d[0] >> 16) & 7;
          if (mult == 2 || mult == 4) flags |= AF_MULTSH;
          if (mult == 3) flags |= AF_MULT3;
          if ((d[0] >> 20) & 0xf) flags |= AF_OFFS;
          if (d[0] & (1u << 19)) flags |= AF_HALVE;
        }
        if (st.p1 & DOTF_WRED) flags |= AF_WRED;
        sg.p0 = st.p0; sg.flags = flags; sg.t = T; sg.sh0 = st.shape[0]; sg.sh1 = st.shape[1];
        const u32 HW = 4 * aot_dot_hdr_quads(T), nw = HW + 4 * st.p0;   // header padded to whole 16-byte words: round r is word HW / 4 + r
        for (u32 lane = 0; lane < 64; lane++) {
          std::vector<u32>& w = lane_words[lane]; w.assign(nw, zero);
          u32 g, li;
          if (!active(lane, g, li)) { w[0] = junk; w[1] = 0; for (u32 t = 0; t < T; t++) w[AOT_DOT_HDR + t] = zero; continue; }
          const u32* d = old_desc(li);
          const u32 mult = (d[0] >> 16) & 7;
          if (li % S) {   // sub-lanes 1 .. S-1: products only; the columns are summed into sub-lane 0, which finishes the lane-op
            w[0] = junk; w[1] = d[1];
            for (u32 r = 0; r < st.p0; r++) for (u32 q = 0; q < 4; q++) w[HW + 4 * r + q] = abs_addr(d[DOT_HDR_WORDS + DOT_ROUND_WORDS * r + q], g);
            continue;
          }
          w[0] = abs_addr(d[0], g) | ((mult >> 1) << 16) | ((mult == 3 ? 1u : 0u) << 18) | (d[0] & (1u << 19)) | (d[0] & (0xfu << 20));
          w[1] = d[1];
          for (u32 t = 0; t < T; t++) w[AOT_DOT_HDR + t] = t < post[li].size() ? (abs_addr(post[li][t].first, g) | ((u32)(post[li][t].second & 0xffff) << 16)) : zero;
          for (u32 r = 0; r < st.p0; r++) for (u32 q = 0; q < 4; q++) w[HW + 4 * r + q] = abs_addr(d[DOT_HDR_WORDS + DOT_ROUND_WORDS * r + q], g);
        }
        break;
      }
      case K_LIN: {
        const u32 nadd = st.p0, nsub = st.p1, nt = nadd + nsub;
        bool halve = false;
        for (u32 li = 0; li < st.nlanes; li++) if (old_desc(li)[0] & (1u << 16)) halve = true;
        sg.p0 = nadd; sg.t = nsub; sg.flags = ((st.lin & 1) ? AF_WRED : 0u) | (halve ? AF_HALVE : 0u);
        const u32 nw = 1 + (nt + 1) / 2;
        for (u32 lane = 0; lane < 64; lane++) {
          std::vector<u32>& w = lane_words[lane]; w.assign(nw, 0);
          u32 g, li;
          const bool act = active(lane, g, li);
          const u32* d = act ? old_desc(li) : nullptr;
          w[0] = act ? (abs_addr(d[0], g) | (d[0] & (1u << 16))) : junk;
          for (u32 t = 0; t < nt; t++) {
            const u32 a = act ? abs_addr((d[1 + t / 2] >> (16 * (t & 1))) & 0xffffu, g) : zero;
            w[1 + t / 2] |= a << (16 * (t & 1));
          }
        }
        break;
      }
      case K_LOAD: case K_LOADW: case K_STORE: case K_STOREW: {
        sg.p0 = st.p0;
        for (u32 lane = 0; lane < 64; lane++) {
          std::vector<u32>& w = lane_words[lane]; w.assign(2, 0);
          u32 g, li;
          if (!active(lane, g, li)) { w[0] = junk; continue; }   // inactive: bit 31 clear
          const u32* d = old_desc(li);
          w[0] = abs_addr(d[0], g) | (((d[0] >> 16) & 7u) << 16) | (1u << 31);
          w[1] = d[1];
        }
        break;
      }
      case K_ISZ: case K_CANON: case K_SEL: case K_CMP: case K_FLAG: case K_BITAND: case K_BIT: case K_STATUS: {
        // the remaining kinds run the interpreter's own lane code (vm_exec.h exec_lane) on descriptors whose 16-bit fields are absolute addresses
        sg.p0 = st.p0;
        const u32 nw = st.kind == K_STATUS ? 8u : 2u;
        for (u32 lane = 0; lane < 64; lane++) {
          std::vector<u32>& w = lane_words[lane]; w.assign(nw, 0);
          u32 g, li;
          if (!active(lane, g, li)) { w[0] = st.kind == K_STATUS ? 0u : junk; continue; }   // sources: the zero constant; a status step with no flags and the active bit clear
          const u32* d = old_desc(li);
          auto lo = [&](u32 x) { return abs_addr(x & 0xffffu, g); };
          auto hi = [&](u32 x) { return abs_addr(x >> 16, g) << 16; };
          switch (st.kind) {
            case K_ISZ: case K_CANON: w[0] = lo(d[0]) | hi(d[0]); break;
            case K_SEL: w[0] = lo(d[0]) | hi(d[0]); w[1] = lo(d[1]) | hi(d[1]); break;
            case K_CMP: case K_FLAG: case K_BITAND: w[0] = lo(d[0]); w[1] = lo(d[1]) | hi(d[1]); break;
            case K_BIT: w[0] = lo(d[0]) | hi(d[0]); w[1] = d[1]; break;
            default: {   // K_STATUS: w0 = flag count | buffer << 16 ; then flag address | code << 16
              const u32 n = d[0] & 0xffu;
              w[0] = n | (d[0] & (7u << 16)) | (1u << 31);
              for (u32 k = 0; k < n && k < 7; k++) w[1 + k] = lo(d[1 + k]) | (d[1 + k] & 0xffff0000u);
            }
          }
        }
        break;
      }
      default: return p.name + ": step kind " + std::to_string((int)st.kind) + " has no ahead-of-time body";
    }
    // signature index
    size_t k = 0;
    while (k < out.sigs.size() && !(out.sigs[k] == sg)) k++;
    if (k == out.sigs.size()) { out.sigs.push_back(sg); out.sig_count.push_back(0); }
    out.sig_count[k]++;
    // descriptor block [word4][lane]
    const u32 nw4 = (u32)(lane_words[0].size() + 3) / 4;
    if (k > 255 || nw4 > 255) return p.name + ": signature / descriptor size out of range";
    AotStep as; as.x = (u32)k | (nw4 << 8); as.y = (u32)(out.descs.size() / 4);
    out.descs.resize(out.descs.size() + (size_t)nw4 * 64 * 4, 0);
    for (u32 lane = 0; lane < 64; lane++) for (size_t i = 0; i < lane_words[lane].size(); i++) out.descs[((size_t)as.y + (i / 4) * 64 + lane) * 4 + (i & 3)] = lane_words[lane][i];
    out.steps.push_back(as);
  }
  // the kernels fetch three 16-byte words of the next step unconditionally: pad the stream
  out.descs.resize(out.descs.size() + 3 * 64 * 4, 0);
  return std::string();
}

}  // namespace nbls
