// trace.cpp -- scheduler, slot allocator and descriptor emitter of the wave VM (see trace.h).
#include "trace.h"
#include <cassert>
#include <queue>
#include "consts_gen.h"

namespace nbls {

Builder::Builder() {
  cur() = this;
  static const u32 zero[12] = {0};
  zero_atom = const_atom(zero);
  one_atom = const_atom(NBLS_R1);
  r2_atom = const_atom(NBLS_R2);
  rawone_atom = const_atom(NBLS_RAW_ONE);
}

SFp input(int buf, int off) {
  Builder* B = Builder::cur();
  Node n; n.kind = K_LOAD; n.buf = buf; n.off = off;
  int raw = B->add_node(n);
  Node m; m.kind = K_MUL; m.a0 = raw; m.b0 = B->r2_atom;       // x * R^2 / R = x R : to Montgomery form (any x < 2^384)
  return SFp(B->add_node(m));
}
void output(const SFp& x, int buf, int off) {
  Builder* B = Builder::cur();
  Node m; m.kind = K_MUL; m.a0 = materialize(x); m.b0 = B->rawone_atom;   // x / R : out of Montgomery form
  int v = B->add_node(m);
  Node n; n.kind = K_STORE; n.a0 = v; n.buf = buf; n.off = off; n.live = true;
  B->add_node(n);
}

static void node_deps(const Node& n, std::vector<int>& d) {
  d.clear();
  switch (n.kind) {
    case K_MUL: d = {n.a0, n.a1, n.b0, n.b1}; break;
    case K_LIN: for (auto& t : n.lin) d.push_back(t.first); break;
    case K_STORE: case K_STOREW: case K_ISZ: case K_CANON: d = {n.a0}; break;
    case K_SEL: d = {n.b0, n.a0, n.a1}; break;
    case K_CMP: case K_FLAG: d = {n.a0, n.a1}; break;
    case K_STATUS: for (auto& t : n.stat) d.push_back(t.first); break;
    default: break;
  }
  d.erase(std::remove(d.begin(), d.end(), -1), d.end());
}

static void make_pm2(std::vector<u32>& out) {
  // k * 2p for k = 0..16, 13 significant words, padded to 16
  u32 acc[16] = {0};
  for (int k = 0; k <= 16; k++) {
    for (int i = 0; i < 16; i++) out.push_back(acc[i]);
    uint64_t c = 0;
    for (int i = 0; i < 16; i++) { uint64_t s = (uint64_t)acc[i] + (i < 12 ? NBLS_2P[i] : 0) + c; acc[i] = (u32)s; c = s >> 32; }
  }
}

Program Builder::compile(const std::string& name, int W) {
  Program P; P.name = name; P.W = W; P.G = 64 / W;
  const int N = (int)nodes.size();
  std::vector<int> d;
  // 1. liveness from sinks
  for (int i = N - 1; i >= 0; i--) {
    if (!nodes[i].live) continue;
    node_deps(nodes[i], d);
    for (int x : d) nodes[x].live = true;
  }
  // 2. users / dependency counts (constants are always available)
  for (int i = 0; i < N; i++) {
    Node& n = nodes[i];
    if (!n.live || n.kind == 0xff) continue;
    node_deps(n, d);
    std::sort(d.begin(), d.end()); d.erase(std::unique(d.begin(), d.end()), d.end());
    for (int x : d) if (nodes[x].kind != 0xff) { nodes[x].users.push_back(i); n.ndeps++; }
  }
  // 3. heights (critical path to a sink)
  auto cost = [&](const Node& n) { return n.kind == K_MUL ? 20 : n.kind == K_LIN ? 3 + (int)n.lin.size() / 3 : 2; };
  for (int i = N - 1; i >= 0; i--) {
    Node& n = nodes[i];
    if (!n.live || n.kind == 0xff) continue;
    int h = 0; for (int u : n.users) h = std::max(h, nodes[u].height);
    n.height = h + cost(n);
  }
  // 4. list scheduling.  Ready queues per (kind, p0).
  auto qkey = [&](const Node& n) { return (int)n.kind * 256 + ((n.kind == K_CMP || n.kind == K_FLAG) ? n.p0 : 0); };
  auto cmp = [&](int a, int b) { return nodes[a].height < nodes[b].height || (nodes[a].height == nodes[b].height && a > b); };
  typedef std::priority_queue<int, std::vector<int>, decltype(cmp)> PQ;
  std::map<int, PQ> ready;
  int remaining = 0;
  for (int i = 0; i < N; i++) {
    Node& n = nodes[i];
    if (!n.live || n.kind == 0xff) continue;
    remaining++;
    if (n.ndeps == 0) ready.emplace(qkey(n), PQ(cmp)).first->second.push(i);
  }
  std::vector<std::vector<int>> step_nodes;
  const int MULKEY = K_MUL * 256;
  while (remaining > 0) {
    int key = -1;
    auto itm = ready.find(MULKEY);
    size_t nmul = itm == ready.end() ? 0 : itm->second.size();
    if (nmul >= (size_t)W) key = MULKEY;
    else {
      static const int order[] = {K_LOAD, K_LOADW, K_LIN, K_ISZ, K_FLAG, K_CMP, K_CANON, K_SEL, K_STOREW, K_STORE, K_STATUS};
      for (int k : order) {
        for (auto& kv : ready) if (kv.first / 256 == k && !kv.second.empty()) { key = kv.first; break; }
        if (key >= 0) break;
      }
      if (key < 0) key = MULKEY;
    }
    PQ& q = ready.find(key)->second;
    assert(!q.empty());
    std::vector<int> chosen;
    while (!q.empty() && (int)chosen.size() < W) { chosen.push_back(q.top()); q.pop(); }
    int sidx = (int)step_nodes.size();
    for (size_t l = 0; l < chosen.size(); l++) { nodes[chosen[l]].step = sidx; nodes[chosen[l]].lane = (int)l; }
    step_nodes.push_back(chosen);
    remaining -= (int)chosen.size();
    for (int c : chosen) for (int u : nodes[c].users) if (--nodes[u].ndeps == 0) ready.emplace(qkey(nodes[u]), PQ(cmp)).first->second.push(u);
  }
  // 5. slot allocation (linear scan; a destination may reuse a slot whose last read is in the same step)
  for (int i = 0; i < N; i++) { Node& n = nodes[i]; if (!n.live || n.kind == 0xff) continue; for (int u : n.users) n.last_use = std::max(n.last_use, nodes[u].step); }
  std::vector<int> free_slots; int nslots = 0;
  std::vector<std::vector<int>> dying(step_nodes.size());
  for (int i = 0; i < N; i++) { Node& n = nodes[i]; if (n.live && n.kind != 0xff && n.last_use >= 0) dying[n.last_use].push_back(i); }
  auto has_dst = [](uint8_t k) { return k != K_STORE && k != K_STOREW && k != K_STATUS; };
  for (size_t s = 0; s < step_nodes.size(); s++) {
    for (int x : dying[s]) free_slots.push_back(nodes[x].slot);
    for (int c : step_nodes[s]) {
      if (!has_dst(nodes[c].kind)) continue;
      if (free_slots.empty()) nodes[c].slot = nslots++; else { nodes[c].slot = free_slots.back(); free_slots.pop_back(); }
      if (nodes[c].last_use < 0) free_slots.push_back(nodes[c].slot);   // defensive: result never read
    }
  }
  assert(nslots <= (int)OP_SLOT_MASK);
  P.slots = nslots;
  // 6. emit
  auto op = [&](int atom) -> u32 {
    if (atom < 0) return OP_CONST | 0;   // const slot 0 is zero
    const Node& n = nodes[atom];
    if (n.kind == 0xff) return OP_CONST | (u32)n.const_idx;
    assert(n.slot >= 0);
    return (u32)n.slot;
  };
  for (size_t s = 0; s < step_nodes.size(); s++) {
    const std::vector<int>& L = step_nodes[s];
    const Node& n0 = nodes[L[0]];
    Step st; memset(&st, 0, sizeof st);
    st.kind = n0.kind; st.nlanes = (uint8_t)L.size(); st.desc_off = (u32)P.descs.size();
    st.stride = 4;
    if (n0.kind == K_LIN) {
      size_t mx = 0; for (int c : L) mx = std::max(mx, nodes[c].lin.size());
      st.p0 = (uint8_t)mx; st.stride = mx > 6 ? 8 : 4;
      int stages = 0; while ((1u << stages) < mx) stages++;
      st.p1 = (uint8_t)stages;
      P.n_lin_steps++; P.n_lin_ops += (u32)L.size();
    } else if (n0.kind == K_MUL) {
      for (int c : L) { if (nodes[c].a1 >= 0) st.p0 |= 1; if (nodes[c].b1 >= 0) st.p0 |= 2; }
      P.n_mul_steps++; P.n_mul_ops += (u32)L.size();
    } else {
      st.p0 = n0.p0; P.n_other_steps++;
      if (n0.kind == K_STATUS) st.stride = 8;
    }
    for (int c : L) {
      const Node& n = nodes[c];
      std::vector<u32> w(st.stride, 0);
      switch (n.kind) {
        case K_MUL:
          w[0] = op(n.a0) | ((n.a1 >= 0 ? (op(n.a1) | ((u32)n.am << OP_MODE_SHIFT)) : 0u) << 16);
          w[1] = op(n.b0) | ((n.b1 >= 0 ? (op(n.b1) | ((u32)n.bm << OP_MODE_SHIFT)) : 0u) << 16);
          w[2] = (u32)n.slot;
          break;
        case K_LIN:
          w[0] = (u32)n.slot | ((u32)n.lin.size() << 16) | (n.halve ? (1u << 24) : 0u);
          for (size_t t = 0; t < n.lin.size(); t++) {
            u32 term = op(n.lin[t].first) | (n.lin[t].second < 0 ? (1u << OP_MODE_SHIFT) : 0u);
            w[1 + t / 2] |= term << (16 * (t & 1));
          }
          P.n_lin_terms += (u32)n.lin.size();
          break;
        case K_LOAD: case K_LOADW: w[0] = (u32)n.slot | ((u32)n.buf << 16); w[1] = (u32)n.off; break;
        case K_STORE: case K_STOREW: w[0] = op(n.a0) | ((u32)n.buf << 16); w[1] = (u32)n.off; break;
        case K_ISZ: case K_CANON: w[0] = (u32)n.slot | (op(n.a0) << 16); break;
        case K_SEL: w[0] = (u32)n.slot | (op(n.b0) << 16); w[1] = op(n.a0) | (op(n.a1) << 16); break;
        case K_CMP: case K_FLAG: w[0] = (u32)n.slot; w[1] = op(n.a0) | (op(n.a1) << 16); break;
        case K_STATUS:
          assert(n.stat.size() <= 7);
          w[0] = (u32)n.stat.size() | ((u32)n.buf << 16);
          for (size_t k = 0; k < n.stat.size(); k++) w[1 + k] = op(n.stat[k].first) | ((u32)n.stat[k].second << 16);
          break;
        default: assert(0);
      }
      P.descs.insert(P.descs.end(), w.begin(), w.end());
    }
    P.steps.push_back(st);
  }
  P.nconst = (u32)const_words.size() / 12;
  P.consts = const_words;
  make_pm2(P.consts);
  return P;
}

}  // namespace nbls
