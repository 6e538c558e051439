// trace.cpp -- form materialisation, scheduler, slot allocator and descriptor emitter of the wave VM (see trace.h).
#include "trace.h"
#include "config.h"
#include <cassert>
#include <cmath>
#include <queue>
#include "consts_gen.h"
#include "vm_exec.h"

namespace nbls {

static const double P_OVER_R = 1.0 / 1234.0;   // p / 2^392 = 0.00079..., rounded up
static const int MAX_OFFS = 15;                 // DOT: 4-bit multiple of p baked into the accumulator
static const double OUT_CAP = 10.0;             // results above this bound get their post-added terms folded into the dot product
static const double OP_CAP = 24.0;              // product operands above this bound are contracted first (keeps bounds from compounding)

static void add_mod_p(u32* x, const u32* y) {   // canonical 14-limb values
  for (int i = 0; i < NLIMBS; i++) x[i] += y[i];
  carry_norm(x);
  csub_p(x);
}

Builder::Builder() {
  cur() = this;
  static const u32 zero[NLIMBS] = {0};
  zero_atom = const_atom(zero);
  one_atom = const_atom(NBLS_R1);
  r2_atom = const_atom(NBLS_R2);
  rawone_atom = const_atom(NBLS_RAW_ONE);
}

int Builder::small_const(int c) {
  assert(c > 0 && c < 4096);
  auto it = small_consts.find(c);
  if (it != small_consts.end()) return it->second;
  u32 acc[NLIMBS] = {0}, dbl[NLIMBS];
  memcpy(dbl, NBLS_R1, NLIMBS * 4);
  for (int k = c; k; k >>= 1) { if (k & 1) add_mod_p(acc, dbl); u32 t[NLIMBS]; memcpy(t, dbl, NLIMBS * 4); add_mod_p(dbl, t); }
  int id = const_atom(acc);
  small_consts[c] = id;
  return id;
}

int Builder::frac_const(int c, int m) {
  auto it = frac_consts.find({c, m}); if (it != frac_consts.end()) return it->second;
  u32 unit[NLIMBS];   // 1/m in Montgomery form
  if (m == 1) memcpy(unit, NBLS_R1, NLIMBS * 4);
  else if (m == 2) memcpy(unit, NBLS_HALF, NLIMBS * 4);
  else if (m == 3) memcpy(unit, NBLS_INV3, NLIMBS * 4);
  else { assert(m == 4); memcpy(unit, NBLS_HALF, NLIMBS * 4); u32 acc[NLIMBS] = {0};   // 1/4 = (1/2) * (1/2): halve 1/2 = (x + (x odd ? p : 0)) / 2
         memcpy(acc, NBLS_HALF, NLIMBS * 4); halve28(acc); csub_p(acc); memcpy(unit, acc, NLIMBS * 4); }
  u32 acc[NLIMBS] = {0};
  for (int k = 0; k < std::abs(c); k++) add_mod_p(acc, unit);
  if (c < 0) { u32 P[NLIMBS] = NBLS_P_INIT, z = 0; for (int i = 0; i < NLIMBS; i++) z |= acc[i]; if (z) { for (int i = 0; i < NLIMBS; i++) acc[i] = P[i] + (i < NLIMBS - 1 ? (1u << 28) : 0) - acc[i] - (i > 0 ? 1 : 0); carry_norm(acc); csub_p(acc); } }
  int id = const_atom(acc); frac_consts[{c, m}] = id; return id;
}

SFp input(int buf, int off) {
  Builder* B = Builder::cur();
  Node n; n.kind = K_LOAD; n.buf = buf; n.off = off; n.raw = true; n.bound = 9.85;
  int raw = B->add_node(n);
  // x * R^2 / R = x R : to Montgomery form (valid for any x < 2^384)
  Operand a; a.s0 = raw; Operand b; b.s0 = B->r2_atom;
  SFp r; r.f.push_back({PROD_BASE + B->product(a, b), 1});
  return SFp(materialize(r));
}
SFp input_raw(int buf, int off, int nbytes) {
  Node n; n.kind = K_LOAD; n.buf = buf; n.off = off; n.raw = true; n.bound = 9.85; n.p0 = (uint8_t)(nbytes == 48 ? 0 : nbytes);
  return SFp(Builder::cur()->add_node(n));
}
SFp to_mont(const SFp& raw) {
  Builder* B = Builder::cur();
  Operand a; a.s0 = materialize(raw); Operand b; b.s0 = B->r2_atom;
  SFp r; r.f.push_back({PROD_BASE + B->product(a, b), 1});
  return SFp(materialize(r));
}
SFp raw_const(const u32* limbs) { return SFp(Builder::cur()->const_atom(limbs)); }
void output(const SFp& x, int buf, int off) {
  Builder* B = Builder::cur();
  Operand a; a.s0 = materialize(x); Operand b; b.s0 = B->rawone_atom;      // x / R : out of Montgomery form
  SFp r; r.f.push_back({PROD_BASE + B->product(a, b), 1});
  Node n; n.kind = K_STORE; n.a0 = materialize(r); n.buf = buf; n.off = off; n.live = true;
  B->add_node(n);
}

// 48 big-endian bytes <- a raw integer below 2^384 held in a slot (no Montgomery conversion, no reduction mod p)
void output_raw(const SFp& x, int buf, int off) {
  Node n; n.kind = K_STORE; n.p0 = 1; n.a0 = materialize(x); n.buf = buf; n.off = off; n.live = true;
  Builder::cur()->add_node(n);
}

int Builder::kp_atom(int k) {
  assert(k >= 1 && k <= 64);
  u32 P[NLIMBS] = NBLS_P_INIT, acc[NLIMBS] = {0};
  for (int i = 0; i < k; i++) { for (int j = 0; j < NLIMBS; j++) acc[j] += P[j]; carry_norm(acc); }
  int id = const_atom(acc); nodes[id].bound = k; if (env_set("NBLS_DUMP_KP")) fprintf(stderr, "kp %d\n", k); return id;
}

int Builder::contract(int atom) {
  if (atom_bound(atom) <= 1.5 || nodes[atom].kind == 0xff) return atom;
  auto it = contract_cache.find(atom); if (it != contract_cache.end()) return it->second;
  Node n; n.kind = K_DOT; Operand a; a.s0 = atom; Operand b; b.s0 = one_atom;
  n.prods.push_back({a, b, false}); n.mult = 1;
  n.bound = atom_bound(atom) * P_OVER_R + 1.0;
  int id = add_node(n); contract_cache[atom] = id; return id;
}

// Bring an atom's bound under `cap`: if the lane-op that produces it can still take a weak reduction (a table lookup and 14 subtractions at its
// end) that is used -- retroactively, which only makes the bounds its earlier consumers assumed conservative -- else a contraction lane-op.
int Builder::lower_bound(int atom, double cap) {
  if (atom_bound(atom) <= cap) return atom;
  Node& n = nodes[atom];
  const double after = n.halve ? 3.02 / 2 + 0.5 : 3.02;
  if (use_wred && !n.raw && !n.wred && after <= cap && n.bound < 100.0 &&
      ((n.kind == K_DOT && n.mult + (int)n.lin.size() <= 7 && !n.prods.empty()) || n.kind == K_LIN)) { n.wred = true; n.bound = after; return atom; }
  return contract(atom);
}

// Emit one DOT node for (prods, lin) -- caller guarantees the limits.
static const double WRED_BOUND = 3.02, WRED_MAX_IN = 100.0;   // weak_reduce (vm_exec.h): output bound; largest input bound the table covers with margin
static int emit_dot(Builder* B, std::vector<DotProduct> prods, int mult, std::vector<std::pair<int, int>> lin, bool halve_it, bool wred = false) {
  for (auto& p : prods) {
    auto cap = [&](int& a) { if (a >= 0 && B->atom_bound(a) > OP_CAP && !B->nodes[a].raw) a = B->lower_bound(a, OP_CAP); };
    cap(p.a.s0); cap(p.a.s1); cap(p.b.s0); cap(p.b.s1);
  }
  for (auto& t : lin) t.first = B->lower_bound(t.first, t.second < 0 ? B->neg_cap : OP_CAP);
  Node n; n.kind = K_DOT; n.mult = mult; n.halve = halve_it;
  // value bounds: REDC(T) lies in (T/R, T/R + p); products that may be negative are compensated by offs * p
  double Vpos = 0, Vneg = 0, Lpos = 0, Lneg = 0;
  for (auto& p : prods) {
    double v = B->operand_bound(p.a) * B->operand_bound(p.b);
    bool maybe_neg = p.neg || (p.a.s1 >= 0 && p.a.n1) || (p.b.s1 >= 0 && p.b.n1);
    bool surely_neg = p.neg && !(p.a.s1 >= 0 && p.a.n1) && !(p.b.s1 >= 0 && p.b.n1);
    if (maybe_neg) Vneg += v;
    if (!surely_neg) Vpos += v;
  }
  for (auto& t : lin) (t.second < 0 ? Lneg : Lpos) += B->atom_bound(t.first);
  int offs = (int)std::ceil(Vneg * P_OVER_R + Lneg / mult - 1e-9);
  if (offs > MAX_OFFS) { fprintf(stderr, "DOT offset %d: Vneg=%.1f Lneg=%.1f mult=%d\n", offs, Vneg, Lneg, mult); }
  assert(offs <= MAX_OFFS);
  n.offs = offs;
  // lane-split programs (round 6, aot_exec.h aot_ls_presum_mask): every sub-lane that holds a product runs its own reduction, each adding less than p
  const double redc_p = B->lane_split > 1 ? (double)std::min<size_t>((size_t)B->lane_split, std::max<size_t>(prods.size(), 1)) : 1.0;
  double T = mult * (Vpos * P_OVER_R + redc_p + offs) + Lpos;
  if (wred) { assert(T < WRED_MAX_IN && mult + (int)lin.size() <= 7); n.wred = true; T = WRED_BOUND; }
  if (halve_it) T = T / 2 + 0.5;
  assert(T < 1000.0);
  // limb budget of the signed column accumulators: sum of c_a * c_b <= 8, c = 2 for an un-normalised sum operand
  auto is_sum = [](const Operand& o) { return o.s1 >= 0 && !o.n1; };
  for (;;) {
    int total = 0, worst = -1, worst_cc = 0;
    for (size_t i = 0; i < prods.size(); i++) {
      auto& p = prods[i];
      int ca = is_sum(p.a) && !p.norm_a ? 2 : 1, cb = is_sum(p.b) && !p.norm_b ? 2 : 1;
      total += ca * cb;
      if (ca * cb > worst_cc) { worst_cc = ca * cb; worst = (int)i; }
    }
    if (total <= 8) break;
    assert(worst >= 0 && worst_cc > 1);
    auto& p = prods[worst];
    if (is_sum(p.a) && !p.norm_a) p.norm_a = true; else p.norm_b = true;
  }
  n.prods = prods; n.lin = lin;
  n.bound = T;
  return B->add_node(n);
}
static int emit_lin(Builder* B, std::vector<std::pair<int, int>> terms, bool halve_it) {
  for (auto& t : terms) t.first = B->lower_bound(t.first, t.second < 0 ? B->neg_cap : OP_CAP);
  // negative terms are limb-wise subtractions; a constant k p keeps the value non-negative
  auto finish = [&](std::vector<std::pair<int, int>>& ts) {
    double neg = 0; for (auto& t : ts) if (t.second < 0) neg += B->atom_bound(t.first);
    if (neg > 0) ts.push_back({B->kp_atom((int)std::ceil(neg - 1e-9)), 1});
  };
  auto bound_of = [&](const std::vector<std::pair<int, int>>& ts) { double T = 0; for (auto& t : ts) if (t.second > 0) T += B->atom_bound(t.first); return T; };
  while ((int)terms.size() > MAX_LIN_TERMS - 1) {
    Node n; n.kind = K_LIN; n.lin.assign(terms.begin(), terms.begin() + MAX_LIN_TERMS - 1); finish(n.lin); n.bound = bound_of(n.lin);
    int id = B->add_node(n);
    terms.erase(terms.begin(), terms.begin() + MAX_LIN_TERMS - 1);
    terms.insert(terms.begin(), {id, 1});
  }
  Node n; n.kind = K_LIN; n.lin = terms; finish(n.lin); n.halve = halve_it; n.bound = bound_of(n.lin);
  if (halve_it) n.bound = n.bound / 2 + 0.5;
  assert(n.bound < 1000.0 && (int)n.lin.size() <= MAX_LIN_TERMS);
  return B->add_node(n);
}

int materialize(const SFp& x, bool halve_it) {
  Builder* B = Builder::cur();
  if (x.f.empty()) return B->zero_atom;
  if (!halve_it && x.f.size() == 1 && x.f[0].second == 1 && x.f[0].first < PROD_BASE) return (int)x.f[0].first;
  auto& cse = halve_it ? B->halve_cse : B->mat_cse;
  auto it = cse.find(x.f);
  if (it != cse.end()) return it->second;

  std::vector<std::pair<int, int>> prods;   // (product id, coef)
  std::vector<std::pair<int, int>> atoms;   // (atom, coef)
  for (auto& t : x.f) {
    if (t.first >= PROD_BASE) prods.push_back({(int)(t.first - PROD_BASE), t.second});
    else if (std::abs(t.second) > 4 || (std::abs(t.second) == 4 && !x.pure_atoms())) {   // 4 x as a pure sum stays a LIN lane-op (x + x + x + x)
      // large integer multiple of an atom: multiply by the constant inside the dot product instead of repeated addition
      Operand a; a.s0 = (int)t.first; Operand c; c.s0 = B->small_const(std::abs(t.second));
      prods.push_back({B->product(a, c), t.second > 0 ? 1 : -1});
    } else atoms.push_back({(int)t.first, t.second});
  }
  // common multiplier m of the products: every |coef| / m must be 1, or 2 with one single-atom operand (2x = x + x)
  int g = 0; for (auto& p : prods) { int a = std::abs(p.second), b = g; while (b) { int t = a % b; a = b; b = t; } g = a; }
  int mult = 1;
  for (int i = 0; i < 4; i++) {
    // the largest multiplier first (a factor costs the lane-op a shift of its result), or -- prefer_doubling -- the smallest: the factor then rides in doubled
    // single-slot operands (x + x), which is free in a step whose rounds form two-term operands for other lanes anyway, and a lane-op without a multiplier
    // finishes in one carry pass
    const int m = B->prefer_doubling ? 1 + i : 4 - i;
    if (g % m) continue;
    bool ok = true;
    for (auto& p : prods) {
      int q = std::abs(p.second) / m; const ProdKey& k = B->prods[p.first];
      if (!(q == 1 || (q == 2 && (k.a.s1 < 0 || k.b.s1 < 0)) || (q == 4 && k.a.s1 < 0 && k.b.s1 < 0))) { ok = false; break; }
    }
    if (ok) { mult = m; break; }
  }
  std::vector<DotProduct> dps;
  for (auto& p : prods) {
    ProdKey k = B->prods[p.first];
    int q = std::abs(p.second) / mult; bool neg = p.second < 0;
    if (q == 4 && k.a.s1 < 0 && k.b.s1 < 0) { k.a.s1 = k.a.s0; k.a.n1 = false; k.b.s1 = k.b.s0; k.b.n1 = false; q = 1; }   // 4 x y = (x + x)(y + y)
    else if (q == 2 && k.a.s1 < 0) { k.a.s1 = k.a.s0; k.a.n1 = false; q = 1; }
    else if (q == 2 && k.b.s1 < 0) { k.b.s1 = k.b.s0; k.b.n1 = false; q = 1; }
    if (q == 1) dps.push_back({k.a, k.b, neg});
    else {
      // awkward coefficient: evaluate the product on its own, then use it as a linear term
      SFp one; one.f.push_back({PROD_BASE + p.first, 1});
      int a = materialize(one);
      int c = p.second;
      if (std::abs(c) > 3) { Operand oa; oa.s0 = a; Operand oc; oc.s0 = B->small_const(std::abs(c)); dps.push_back({oa, oc, c < 0}); assert(mult == 1); }
      else atoms.push_back({a, c});
    }
  }
  std::vector<std::pair<int, int>> lin;
  for (auto& a : atoms) for (int k = 0; k < std::abs(a.second); k++) lin.push_back({a.first, a.second > 0 ? 1 : -1});

  int id;
  bool want_wred = false;
  if (!dps.empty() && !lin.empty()) {
    // If the post-added terms would make the result large, fold them into the accumulator as products with small
    // constants (x * (c/m)): the Montgomery reduction then contracts everything to about m p.
    double T = mult * 3.0; for (auto& t : lin) T += std::min(B->atom_bound(t.first), t.second < 0 ? B->neg_cap : OP_CAP);
    // ... or, cheaper, keep them as post-added terms and reduce weakly afterwards (a table lookup and 14 subtractions instead of a round of 196
    // multiply-adds per term): possible while the un-reduced value stays inside the table and the limb sums inside 32 bits
    double Tfull = mult * 3.0; for (auto& t : lin) Tfull += std::min(B->atom_bound(t.first), t.second < 0 ? B->neg_cap : OP_CAP) * (t.second < 0 ? 2 : 1);
    if (T > OUT_CAP && B->use_wred && (int)lin.size() <= MAX_DOT_LINEAR && mult + (int)lin.size() <= 7 && Tfull < WRED_MAX_IN / 2 && (int)dps.size() <= B->max_dot) want_wred = true;
    else if (T > OUT_CAP) {
      std::map<int, int> net; for (auto& t : lin) net[t.first] += t.second;
      for (auto& kv : net) if (kv.second) { Operand a; a.s0 = kv.first; Operand c; c.s0 = B->frac_const(kv.second, mult); dps.push_back({a, c, false}); }
      lin.clear();
    }
  }
  if (dps.empty()) {
    id = emit_lin(B, lin, halve_it);
  } else {
    if ((int)lin.size() > MAX_DOT_LINEAR) { int a = emit_lin(B, lin, false); lin.clear(); lin.push_back({a, 1}); }
    // chunk the products: at most 8 per lane-op (column accumulators hold 112 limb products)
    while (dps.size() > (size_t)B->max_dot) {
      std::vector<DotProduct> chunk(dps.begin(), dps.begin() + B->max_dot);
      int a = emit_dot(B, chunk, mult, {}, false);
      dps.erase(dps.begin(), dps.begin() + B->max_dot);
      if ((int)lin.size() >= MAX_DOT_LINEAR) { int l = emit_lin(B, lin, false); lin.clear(); lin.push_back({l, 1}); }
      lin.push_back({a, 1});
    }
    id = emit_dot(B, dps, mult, lin, halve_it, want_wred);
  }
  cse[x.f] = id;
  return id;
}

static void node_deps(const Node& n, std::vector<int>& d) {
  d.clear();
  switch (n.kind) {
    case K_DOT:
      for (auto& p : n.prods) { d.push_back(p.a.s0); d.push_back(p.a.s1); d.push_back(p.b.s0); d.push_back(p.b.s1); }
      for (auto& t : n.lin) d.push_back(t.first);
      break;
    case K_LIN: for (auto& t : n.lin) d.push_back(t.first); break;
    case K_STORE: case K_STOREW: case K_ISZ: case K_CANON: d = {n.a0}; break;
    case K_SEL: d = {n.b0, n.a0, n.a1}; break;
    case K_CMP: case K_FLAG: case K_BITAND: d = {n.a0, n.a1}; break;
    case K_BIT: d = {n.a0}; break;
    case K_STATUS: for (auto& t : n.stat) d.push_back(t.first); break;
    default: break;
  }
  d.erase(std::remove(d.begin(), d.end(), -1), d.end());
}

Program Builder::compile(const std::string& name, int W) {
  const int S = lane_split;
  assert(S >= 1 && W * S <= 64 && (S & (S - 1)) == 0);
  Program P; P.name = name; P.lsplit = (u32)S; P.W = (u32)(W * S); P.G = 64 / (W * S);   // W stays the logical lane count below
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
  auto cost = [&](const Node& n) { return n.kind == K_DOT ? 8 + 14 * (int)n.prods.size() : n.kind == K_LIN ? 3 + (int)n.lin.size() / 3 : 2; };
  for (int i = N - 1; i >= 0; i--) {
    Node& n = nodes[i];
    if (!n.live || n.kind == 0xff) continue;
    int h = 0; for (int u : n.users) h = std::max(h, nodes[u].height);
    n.height = h + cost(n);
  }
  // 4. list scheduling.  Ready queues per (kind, p0).
  // DOT lane-ops are bucketed by weight class so that a step's lanes do similar amounts of work (step time = max k)
  auto dot_class = [&](const Node& n) { size_t k = n.prods.size(); return (int)k <= light_max ? 0 : 1; };
  auto qkey = [&](const Node& n) { return (int)n.kind * 256 + ((n.kind == K_CMP || n.kind == K_FLAG || n.kind == K_LOAD || n.kind == K_STORE) ? n.p0 : n.kind == K_DOT ? dot_class(n) : 0); };
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
  const int DOTKEY = K_DOT * 256;
  // sched_window (0 = unlimited): a ready lane-op may issue only if its critical-path height is within the window of the
  // most critical ready lane-op.  Without it a short side chain (the point updates of a Miller loop with a shared
  // accumulator) runs many iterations ahead of the main chain and its results pile up in LDS.
  const int window = sched_window > 0 ? sched_window : (1 << 30);
  while (remaining > 0) {
    int front = -1;
    for (auto& kv : ready) if (!kv.second.empty()) front = std::max(front, nodes[kv.second.top()].height);
    const int minh = front - window;
    auto eligible = [&](PQ& q) { return !q.empty() && nodes[q.top()].height >= minh; };
    auto take = [&](PQ& q, std::vector<int>& out) { while (eligible(q) && (int)out.size() < W) { out.push_back(q.top()); q.pop(); } };
    int key = -1;
    // a full DOT step of one class is always worth issuing; otherwise drain the cheap kinds first, then issue the DOT
    // class that holds the most urgent (highest critical path) ready lane-op
    std::vector<int> cand[2];
    for (int c = 1; c >= 0; c--) { auto itq = ready.find(DOTKEY + c); if (itq != ready.end()) take(itq->second, cand[c]); }
    // results that only wait to be written out keep their slots alive: a batch of ready stores goes first
    if (store_batch > 0) {
      int nst = 0, kst = -1;
      for (auto& kv : ready) if (kv.first / 256 == K_STOREW || kv.first / 256 == K_STORE) { int c = 0; PQ q = kv.second; while (!q.empty() && c < W) { q.pop(); c++; } if (c > nst) { nst = c; kst = kv.first; } }
      if (nst >= std::min(store_batch, W)) key = kst;
    }
    if (key >= 0) {}
    else if ((int)cand[1].size() >= W) key = DOTKEY + 1;
    else if ((int)cand[0].size() >= W) key = DOTKEY;
    if (key < 0) {
      static const int order[] = {K_LOAD, K_LOADW, K_BIT, K_BITAND, K_LIN, K_ISZ, K_FLAG, K_CMP, K_CANON, K_SEL, K_STOREW, K_STORE, K_STATUS};
      for (int k : order) {
        for (auto& kv : ready) if (kv.first / 256 == k && eligible(kv.second)) { key = kv.first; break; }
        if (key >= 0) break;
      }
      if (key < 0) {
        int h1 = cand[1].empty() ? -1 : nodes[cand[1][0]].height, h0 = cand[0].empty() ? -1 : nodes[cand[0][0]].height;
        key = h1 >= h0 ? DOTKEY + 1 : DOTKEY;
        assert(h1 >= 0 || h0 >= 0);
      }
    }
    std::vector<int> chosen;
    if (key / 256 == K_DOT) {
      const int c = key - DOTKEY;
      chosen = cand[c];
      if (c == 1) { for (int x : cand[0]) { if ((int)chosen.size() < W) chosen.push_back(x); else ready.find(DOTKEY)->second.push(x); } }   // heavy DOT step: light lane-ops ride along in the free lanes at no cost
      else for (int x : cand[1]) ready.find(DOTKEY + 1)->second.push(x);
    } else {
      for (int c = 0; c < 2; c++) for (int x : cand[c]) ready.find(DOTKEY + c)->second.push(x);
      take(ready.find(key)->second, chosen);
    }
    assert(!chosen.empty());
    int sidx = (int)step_nodes.size();
    for (size_t l = 0; l < chosen.size(); l++) { nodes[chosen[l]].step = sidx; nodes[chosen[l]].lane = (int)l; }
    step_nodes.push_back(chosen);
    remaining -= (int)chosen.size();
    for (int c : chosen) for (int u : nodes[c].users) if (--nodes[u].ndeps == 0) ready.emplace(qkey(nodes[u]), PQ(cmp)).first->second.push(u);
  }
  // 4b. operand shapes.  A product's minus sign goes where it is free: reverse a difference, else turn a single slot x into 0 - x
  // (a difference with the zero constant).  Then, because every lane of a wavefront walks the same product rounds and the kernel
  // branches on the shape of a ROUND (uniform, in the step header), the products of each lane-op are permuted (and their operands
  // swapped) so that round i has the same operand shape in as many lanes of the step as possible: a second term costs 14 limb
  // additions for the whole wavefront as soon as ANY lane has one (the others add the zero constant), mixed signs cost three times
  // that, a normalisation 41 instructions.
  auto form_mask = [](const Operand& o, bool norm, bool neg0) { return (o.s1 >= 0 ? (o.n1 ? 2 : 1) : 0) | (norm ? 4 : 0) | (neg0 ? 8 : 0); };
  auto form_cost = [](int m) { return (((m & 3) == 3 || (m & 8)) ? 46 : (m & 3) ? 18 : 0) + ((m & 4) ? 41 : 0); };
  std::vector<std::vector<std::pair<int, int>>> step_shapes(step_nodes.size());   // per DOT step: (shape of A, shape of B) per round
  for (size_t si = 0; si < step_nodes.size(); si++) {
    auto& L = step_nodes[si];
    if (nodes[L[0]].kind != K_DOT) continue;
    for (int c : L) for (auto& p : nodes[c].prods) {
      if (!p.neg) continue;
      if (p.a.s1 >= 0 && p.a.n1) std::swap(p.a.s0, p.a.s1);
      else if (p.b.s1 >= 0 && p.b.n1) std::swap(p.b.s0, p.b.s1);
      else if (p.a.s1 < 0) { p.a.s1 = p.a.s0; p.a.s0 = zero_atom; p.a.n1 = true; }
      else if (p.b.s1 < 0) { p.b.s1 = p.b.s0; p.b.s0 = zero_atom; p.b.n1 = true; }
      else { p.a.n1 = true; p.neg0_a = true; }   // -(x + y) = (-x) - y: both signs per lane (shape mode 3)
      p.neg = false;
    }
    size_t mk = 0; for (int c : L) mk = std::max(mk, nodes[c].prods.size());
    std::vector<int> ua(mk, 0), ub(mk, 0);
    std::vector<int> order(L.begin(), L.end());
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return nodes[x].prods.size() > nodes[y].prods.size(); });
    for (int c : order) {
      auto& pr = nodes[c].prods;
      std::vector<DotProduct> placed(mk); std::vector<char> used(mk, 0);
      std::vector<size_t> idx(pr.size()); for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
      auto weight = [&](const DotProduct& p) { return form_cost(form_mask(p.a, p.norm_a, p.neg0_a)) + form_cost(form_mask(p.b, p.norm_b, p.neg0_b)); };
      std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return weight(pr[x]) > weight(pr[y]); });
      size_t hi = 0;
      for (size_t i : idx) {
        const DotProduct& p = pr[i];
        int ma = form_mask(p.a, p.norm_a, p.neg0_a), mb = form_mask(p.b, p.norm_b, p.neg0_b);
        int best = -1, best_cost = 1 << 30; bool best_swap = false;
        for (size_t j = 0; j < mk; j++) {
          if (used[j]) continue;
          if (j >= pr.size() && j > hi) break;   // keep the lane's products packed at the low rounds
          for (int sw = 0; sw < 2; sw++) {
            int xa = sw ? mb : ma, xb = sw ? ma : mb;
            int inc = form_cost(ua[j] | xa) - form_cost(ua[j]) + form_cost(ub[j] | xb) - form_cost(ub[j]);
            if (inc < best_cost) { best_cost = inc; best = (int)j; best_swap = sw; }
          }
        }
        if (best >= (int)pr.size()) best = -1;
        if (env_set("NBLS_NO_ALIGN")) { best = (int)i; best_swap = false; }
        if (best < 0) { for (size_t j = 0; j < pr.size(); j++) if (!used[j]) { best = (int)j; break; } best_swap = false; }
        DotProduct q = p;
        if (best_swap) { std::swap(q.a, q.b); std::swap(q.norm_a, q.norm_b); std::swap(q.neg0_a, q.neg0_b); }
        placed[best] = q; used[best] = 1; hi = std::max(hi, (size_t)best + 1);
        ua[best] |= form_mask(q.a, q.norm_a, q.neg0_a); ub[best] |= form_mask(q.b, q.norm_b, q.neg0_b);
      }
      placed.resize(pr.size());
      pr = placed;
    }
    // canonical operand order of a round: A carries the larger shape (mode, then normalise), so that the kernel's specialised round bodies
    // (vm_exec.h dot_round) cover the shape pairs that occur
    {
      auto rank = [](int u) { int mode = ((u & 8) || (u & 3) == 3) ? 3 : (u & 3); return mode * 2 + ((u >> 2) & 1); };
      for (size_t j = 0; j < mk; j++) {
        if (rank(ua[j]) >= rank(ub[j])) continue;
        std::swap(ua[j], ub[j]);
        for (int c : L) { auto& pr = nodes[c].prods; if (j < pr.size()) { DotProduct& q = pr[j]; std::swap(q.a, q.b); std::swap(q.norm_a, q.norm_b); std::swap(q.neg0_a, q.neg0_b); } }
      }
    }
    for (size_t j = 0; j < mk; j++) { step_shapes[si].push_back({ua[j], ub[j]}); P.n_round_ops += 2; P.n_op_mode[(ua[j] & 8) ? 3 : (ua[j] & 3)]++; P.n_op_mode[(ub[j] & 8) ? 3 : (ub[j] & 3)]++; P.n_op_norm += ((ua[j] >> 2) & 1) + ((ub[j] >> 2) & 1); }
    if (env_set("NBLS_DUMP_STEPS")) { fprintf(stderr, "%s dot step lanes=%zu k:", name.c_str(), L.size()); for (int c : L) fprintf(stderr, " %zu", nodes[c].prods.size()); fprintf(stderr, "\n"); }
    if (env_set("NBLS_DUMP_NODES")) for (int c : L) { fprintf(stderr, "  node %d m=%d lin=%zu:", c, nodes[c].mult, nodes[c].lin.size()); for (auto& q : nodes[c].prods) fprintf(stderr, " (%d%s%d)x(%d%s%d)", q.a.s0, q.a.s1 < 0 ? "" : (q.a.n1 ? "-" : "+"), q.a.s1 < 0 ? 0 : q.a.s1, q.b.s0, q.b.s1 < 0 ? "" : (q.b.n1 ? "-" : "+"), q.b.s1 < 0 ? 0 : q.b.s1); fprintf(stderr, "\n"); }
    if (env_set("NBLS_DUMP_FORMS")) { fprintf(stderr, "%s forms:", name.c_str()); for (size_t j = 0; j < mk; j++) fprintf(stderr, " %x/%x", ua[j], ub[j]); fprintf(stderr, "\n"); }
    // cost model of the step (VALU instructions per wavefront): rounds of 196 multiply-adds + 4 address additions + the shape work, one
    // reduction (196 + ~100), post-processing
    double est = 60 + 196 + 100;
    for (size_t j = 0; j < mk; j++) est += 196 + 4 + form_cost(ua[j]) + form_cost(ub[j]);
    size_t mp = 0, mn = 0; bool any_mult = false, any_halve = false;
    for (int c : L) { size_t np = 0, nn = 0; for (auto& t : nodes[c].lin) (t.second < 0 ? nn : np)++; mp = std::max(mp, np); mn = std::max(mn, nn); any_mult |= nodes[c].mult > 1; any_halve |= nodes[c].halve; }
    est += (mp + mn) * 16 + (any_mult ? 28 : 0) + ((any_mult || mp + mn) ? 41 : 0) + (any_halve ? 60 : 0);
    P.est_valu += est;
  }
  for (auto& L : step_nodes) {
    const Node& n0 = nodes[L[0]];
    if (n0.kind == K_DOT) continue;
    if (n0.kind == K_LIN) { size_t mp = 0, mn = 0; bool h = false; for (int c : L) { size_t np = 0, nn = 0; for (auto& t : nodes[c].lin) (t.second < 0 ? nn : np)++; mp = std::max(mp, np); mn = std::max(mn, nn); h |= nodes[c].halve; } P.est_valu += 50 + (mp + mn) * 16 + 41 + (h ? 60 : 0); }
    else P.est_valu += 120;
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
    if (env_set("NBLS_DUMP_LIVE")) fprintf(stderr, "%s live step %zu kind=%d slots_in_use=%d\n", name.c_str(), s, step_nodes[s].empty() ? -1 : (int)nodes[step_nodes[s][0]].kind, nslots - (int)free_slots.size());
  }
  P.slots = nslots;
  P.nconst = (u32)const_words.size() / SLOT_WORDS;
  // LDS layout: every instance region holds the constants (replicated, so that an operand address is base + offset whatever it
  // names) followed by the slots.  Slot stride: 64 bytes.  An 80-byte stride spreads the 16-byte reads of lanes that name different
  // slots over all 16 bank groups (64 bytes: 4 of 16) and removes most bank-conflict cycles, but measured slower on the Miller
  // programs (1.98 vs 1.43 ms at 4096 pairings, 14.4 vs 13.0 ms at 65,536) and equal on EXPX: the kernel is bound by VALU issue, not by
  // LDS cycles.  NBLS_SLOT_BYTES=80 selects it for experiments.
  P.slot_bytes = 64;
  // constants: a copy per instance (operand address = base + offset) unless that costs real LDS: with 8 or 16 instances per wavefront the copies
  // of a dozen constants are 5-10 KB and push the point programs from 8 to 6 wavefronts per CU; those keep one shared copy, marked by bit 1
  P.shared_consts = env_long("NBLS_SHARED_CONSTS", -1) >= 0 ? env_long("NBLS_SHARED_CONSTS", -1) != 0 : shared_consts >= 0 ? shared_consts != 0 : P.G >= 8;
  {   // slot stride: 80 bytes where the larger image still leaves room for twelve workgroups per CU (three wavefronts per SIMD is what the register budget
      // allows anyway); NBLS_SLOT_BYTES = 64 / 80 forces one or the other
    const long e = env_long("NBLS_SLOT_BYTES", 0);
    P.slot_bytes = 80;
    const bool fits = P.lds_bytes() <= (160u * 1024u) / 12u;
    P.slot_bytes = e ? (e == 80 ? 80 : 64) : (fits ? 80 : 64);
  }
  assert(P.lds_bytes() < 65536 * 2 && P.inst_bytes() < 32768);
  // 6. emit
  const u32 SLOT_FLAG = P.shared_consts ? 2u : 0u;
  auto op = [&](int atom) -> u32 {   // LDS byte offset: inside the instance region (with SLOT_FLAG when the constants are shared), or an absolute constant offset
    if (atom < 0) return 0u;   // const slot 0 is zero
    const Node& n = nodes[atom];
    if (n.kind == 0xff) return (u32)n.const_idx * P.slot_bytes;
    assert(n.slot >= 0);
    return ((P.shared_consts ? 0u : P.nconst) + (u32)n.slot) * P.slot_bytes | SLOT_FLAG;
  };
  const u32 ZERO_FIELD = op(-1);
  for (size_t s = 0; s < step_nodes.size(); s++) {
    const std::vector<int>& L = step_nodes[s];
    const Node& n0 = nodes[L[0]];
    Step st; memset(&st, 0, sizeof st);
    st.kind = n0.kind; st.nlanes = (uint8_t)L.size(); st.desc_off = (u32)P.descs.size();
    st.stride = 4;
    size_t mp = 0, mn = 0;   // DOT / LIN: added and subtracted terms, max over the lanes
    if (n0.kind == K_LIN || n0.kind == K_DOT) for (int c : L) { size_t np = 0, nn = 0; for (auto& t : nodes[c].lin) (t.second < 0 ? nn : np)++; mp = std::max(mp, np); mn = std::max(mn, nn); }
    if (n0.kind == K_LIN) {
      if (mp == 0) mp = 1;   // the kernel starts from the first added field (the zero constant when no lane adds anything)
      assert(mp <= (size_t)MAX_LIN_TERMS && mn <= (size_t)MAX_LIN_TERMS);
      st.p0 = (uint8_t)mp; st.p1 = (uint8_t)mn; st.stride = mp + mn > 6 ? 8 : 4;
      for (int c : L) if (nodes[c].wred) st.lin |= 1;
      P.n_lin_steps++; P.n_lin_ops += (u32)L.size();
    } else if (n0.kind == K_DOT) {
      size_t mk = step_shapes[s].size();
      for (int c : L) { P.n_products += (u32)nodes[c].prods.size(); if (nodes[c].mult > 1) st.p1 |= DOTF_MULT; if (nodes[c].halve) st.p1 |= DOTF_HALVE; if (nodes[c].offs > 0) st.p1 |= DOTF_OFFS; if (nodes[c].wred) st.p1 |= DOTF_WRED; }
      assert(mk <= (size_t)MAX_DOT_PRODUCTS && mp <= (size_t)MAX_DOT_LINEAR && mn <= (size_t)MAX_DOT_LINEAR);
      st.lin = (u32)mp | ((u32)mn << 4);
      if (S > 1) {   // lane split: physical round r holds the products r * S .. r * S + S - 1 (one per sub-lane); its shape is the union of theirs
        std::vector<std::pair<int, int>> u((mk + S - 1) / S, {0, 0});
        for (size_t j = 0; j < mk; j++) { u[j / S].first |= step_shapes[s][j].first; u[j / S].second |= step_shapes[s][j].second; }
        step_shapes[s] = u; mk = u.size();
      }
      st.p0 = (uint8_t)mk;
      for (size_t j = 0; j < mk; j++) {
        auto sh3 = [](int m) { return (u32)(((m & 8) ? 3 : (m & 3)) | (m & 4)); };
        const u32 sh = sh3(step_shapes[s][j].first) | (sh3(step_shapes[s][j].second) << SH_B_SHIFT);
        st.shape[j / 4] |= sh << (8 * (j & 3));
      }
      st.stride = (u32)(DOT_HDR_WORDS + DOT_ROUND_WORDS * mk);
      P.n_dot_steps++; P.n_dot_ops += (u32)L.size(); P.n_prod_slots += (u32)(mk * W);
    } else {
      st.p0 = n0.p0; P.n_other_steps++;
      if (n0.kind == K_STATUS) st.stride = 8;
    }
    for (int c : L) for (int sub = 0; sub < (n0.kind == K_DOT ? S : 1); sub++) {
      const Node& n = nodes[c];
      std::vector<u32> w(st.stride, 0);
      if (ZERO_FIELD && (n.kind == K_DOT || n.kind == K_LIN)) {   // padding fields name the zero constant
        const int first = n.kind == K_DOT ? 4 : 1, cnt = n.kind == K_DOT ? 8 : std::min<int>(14, 2 * ((int)st.stride - 1));
        for (int t = 0; t < cnt; t++) w[first + t / 2] |= ZERO_FIELD << (16 * (t & 1));
        if (n.kind == K_DOT) for (u32 r = 0; r < st.p0; r++) for (int q = 0; q < 4; q++) w[DOT_HDR_WORDS + DOT_ROUND_WORDS * r + q] = ZERO_FIELD;
      }
      auto put16 = [&](std::vector<u32>& ww, int first, int t, u32 v) { u32& x = ww[first + t / 2]; x = (x & ~(0xffffu << (16 * (t & 1)))) | ((v & 0xffffu) << (16 * (t & 1))); };
      auto put_lin = [&](int first) {   // added terms first, then subtracted ones, each group padded with the zero constant (offset 0)
        int ia = 0, is = 0;
        for (auto& t : n.lin) { if (t.second > 0) put16(w, first, ia++, op(t.first)); else put16(w, first, (int)mp + is++, op(t.first)); }
      };
      switch (n.kind) {
        case K_DOT:
          assert(n.prods.size() <= (size_t)MAX_DOT_PRODUCTS && n.lin.size() <= (size_t)MAX_DOT_LINEAR && n.mult >= 1 && n.mult <= 4);
          w[0] = op(c) | ((u32)n.mult << 16) | (n.halve ? (1u << 19) : 0u) | ((u32)(sub == 0 ? n.offs : 0) << 20);   // the bias enters once: sub-lane 0
          put_lin(4);
          for (size_t pi = (size_t)sub; pi < n.prods.size(); pi += (size_t)S) {
            const DotProduct& p = n.prods[pi];
            const size_t i = pi / (size_t)S;    // physical round of this sub-lane
            u32* r = &w[DOT_HDR_WORDS + DOT_ROUND_WORDS * i];
            const int sa = step_shapes[s][i].first, sb = step_shapes[s][i].second;
            const bool ma = (sa & 3) == 3 || (sa & 8), mb = (sb & 3) == 3 || (sb & 8);   // per-lane signs in bit 0 of the offsets
            r[0] = op(p.a.s0); r[1] = p.a.s1 >= 0 ? op(p.a.s1) : ZERO_FIELD;
            r[2] = op(p.b.s0); r[3] = p.b.s1 >= 0 ? op(p.b.s1) : ZERO_FIELD;
            const u32 neg = ((ma && p.neg0_a) ? 1u : 0u) | ((ma && p.a.s1 >= 0 && p.a.n1) ? 2u : 0u) | ((mb && p.neg0_b) ? 4u : 0u) | ((mb && p.b.s1 >= 0 && p.b.n1) ? 8u : 0u);
            w[1] |= neg << (4 * i);   // per-lane signs of the round's four terms (rounds of shape mode 3)
            P.n_norm_operands += p.norm_a + p.norm_b; P.n_comb_operands += (p.a.s1 >= 0) + (p.b.s1 >= 0);
          }
          break;
        case K_LIN:
          w[0] = op(c) | (n.halve ? (1u << 16) : 0u);
          put_lin(1);
          P.n_lin_terms += (u32)n.lin.size();
          break;
        case K_LOAD: case K_LOADW: w[0] = op(c) | ((u32)n.buf << 16); w[1] = (u32)n.off;
          P.buf_extent[n.buf] = std::max(P.buf_extent[n.buf], (u32)n.off + (n.kind == K_LOADW ? (u32)RAW_FP_BYTES : (n.p0 ? (u32)n.p0 : 48u))); break;
        case K_STORE: case K_STOREW: w[0] = op(n.a0) | ((u32)n.buf << 16); w[1] = (u32)n.off;
          P.buf_extent[n.buf] = std::max(P.buf_extent[n.buf], (u32)n.off + (n.kind == K_STOREW ? (u32)RAW_FP_BYTES : 48u)); break;
        case K_ISZ: case K_CANON: w[0] = op(c) | (op(n.a0) << 16); break;
        case K_SEL: w[0] = op(c) | (op(n.b0) << 16); w[1] = op(n.a0) | (op(n.a1) << 16); break;
        case K_CMP: case K_FLAG: case K_BITAND: w[0] = op(c); w[1] = op(n.a0) | (op(n.a1) << 16); break;
        case K_BIT: w[0] = op(c) | (op(n.a0) << 16); w[1] = (u32)n.off; break;
        case K_STATUS:
          assert(n.stat.size() <= 7);
          w[0] = (u32)n.stat.size() | ((u32)n.buf << 16); P.buf_extent[n.buf] = std::max(P.buf_extent[n.buf], 1u);
          for (size_t k = 0; k < n.stat.size(); k++) w[1 + k] = op(n.stat[k].first) | ((u32)n.stat[k].second << 16);
          break;
        default: assert(0);
      }
      P.descs.insert(P.descs.end(), w.begin(), w.end());
    }
    if (env_set("NBLS_DUMP_SEQ")) fprintf(stderr, "%s step %zu kind=%d lanes=%d p0=%d p1=%d lin=0x%x\n", name.c_str(), s, st.kind, st.nlanes, st.p0, st.p1, st.lin);
    P.steps.push_back(st);
  }
  P.consts = const_words;
  if (env_set("NBLS_DUMP_CONTRACT")) { int nc = 0, nw = 0; for (auto& kv : contract_cache) if (nodes[kv.second].live) nc++; for (auto& n : nodes) if (n.live && n.wred) nw++; fprintf(stderr, "%s: %d live contraction lane-ops, %d weak reductions\n", name.c_str(), nc, nw); }
  return P;
}

const u32* qp_table_words() {
  static std::vector<u32> tab;
  static bool built = false;
  if (!built) {
    const u32 P[NLIMBS] = NBLS_P_INIT;
    tab.assign((size_t)QP_TABLE_ENTRIES * RAW_WORDS, 0);
    u32 acc[NLIMBS] = {0};
    for (int q = 1; q < QP_TABLE_ENTRIES; q++) {
      for (int j = 0; j < NLIMBS; j++) acc[j] += P[j];
      carry_norm(acc);
      memcpy(&tab[(size_t)q * RAW_WORDS], acc, NLIMBS * 4);
    }
    built = true;
  }
  return tab.data();
}

std::string verify_program(const Program& p) {
  char msg[256];
  const u32 ib = p.inst_bytes(), cbytes = p.nconst * p.slot_bytes;
  const bool sh = p.shared_consts;
  auto bad = [&](size_t s, unsigned lane, const char* what, u32 v) { snprintf(msg, sizeof msg, "%s: step %zu lane %u: %s (0x%x)", p.name.c_str(), s, lane, what, v); return std::string(msg); };
  if (p.slot_bytes != 64 && p.slot_bytes != 80) return p.name + ": slot stride";
  if ((u64)p.lds_bytes() > 160 * 1024) return p.name + ": LDS image exceeds 160 KB";
  if (p.W * p.G > 64 || p.W == 0) return p.name + ": lanes";
  if (p.consts.size() != (size_t)p.nconst * RAW_WORDS) return p.name + ": constant table size";
  // an operand offset: replicated constants -> anything inside the instance region; shared -> bit 1 marks a slot of the instance region, without it
  // the offset is that of a constant (absolute)
  auto inside = [&](u32 f, u32 flagmask) {
    if (f & flagmask) return false;
    const u32 o = f & ~15u;
    if (sh && !(f & 2u)) return o + 56 <= cbytes;
    return o + 56 <= ib;
  };
  // declared aliases: no load of the input buffer after the first store to the output buffer (whatever the lanes: the items of a wavefront run in lock step, but a store of one
  // lane-op and a later load of another would see the new bytes)
  for (auto& al : p.aliases) {
    bool stored = false;
    for (size_t s = 0; s < p.steps.size(); s++) {
      const Step& st = p.steps[s];
      if (st.kind != K_LOAD && st.kind != K_LOADW && st.kind != K_STORE && st.kind != K_STOREW) continue;
      for (unsigned l = 0; l < st.nlanes; l++) {
        const u32 buf = (p.descs[st.desc_off + l * st.stride] >> 16) & 7u;
        if ((st.kind == K_STORE || st.kind == K_STOREW) && (int)buf == al.first) stored = true;
        if ((st.kind == K_LOAD || st.kind == K_LOADW) && (int)buf == al.second && stored) return bad(s, l, "load of an aliased input after the first store to its output", buf);
      }
    }
  }
  for (size_t s = 0; s < p.steps.size(); s++) {
    const Step& st = p.steps[s];
    if (st.nlanes == 0 || st.nlanes > p.W) return bad(s, 0, "active lanes", st.nlanes);
    if (st.stride % 4 || st.stride < 4) return bad(s, 0, "descriptor stride", st.stride);
    const unsigned ndesc = st.kind == K_DOT ? st.nlanes * p.lsplit : st.nlanes;   // lane split: a K_DOT lane-op has one descriptor per sub-lane
    if (st.nlanes * (st.kind == K_DOT ? p.lsplit : 1u) > p.W) return bad(s, 0, "active lanes", st.nlanes);
    if ((u64)st.desc_off + (u64)ndesc * st.stride > p.descs.size() || st.desc_off % 4) return bad(s, 0, "descriptors outside the program", st.desc_off);
    for (unsigned l = 0; l < ndesc; l++) {
      const u32* d = p.descs.data() + st.desc_off + l * st.stride;
      auto src = [&](u32 f) { return inside(f & 0xffffu, sh ? 13u : 15u); };                                  // a readable slot (constants included)
      auto dst = [&](u32 f) { f &= 0xffffu; return (f & 15u) == (sh ? 2u : 0u) && (sh || f >= cbytes) && (f & ~15u) + 56 <= ib; };  // a writable slot (never a constant)
      auto term = [&](u32 f) { return inside(f, sh ? 13u : 15u); };       // product term: 32-bit offset
      switch (st.kind) {
        case K_DOT: {
          if (st.p0 > MAX_DOT_PRODUCTS || st.stride != (u32)(DOT_HDR_WORDS + DOT_ROUND_WORDS * st.p0)) return bad(s, l, "product rounds / stride", st.p0);
          if (!dst(d[0])) return bad(s, l, "destination", d[0]);
          const u32 nadd = st.lin & 7, nsub = (st.lin >> 4) & 7;
          if (nadd > MAX_DOT_LINEAR || nsub > MAX_DOT_LINEAR || (st.lin & ~0x77u)) return bad(s, l, "post-added term counts", st.lin);
          for (u32 t = 0; t < nadd + nsub; t++) { const u32 f = (d[4 + t / 2] >> (16 * (t & 1))) & 0xffffu; if (!src(f)) return bad(s, l, "post-added term", f); }
          const u32 mult = (d[0] >> 16) & 7; if (mult < 1 || mult > 4) return bad(s, l, "multiplier", mult);
          if (mult > 1 && !(st.p1 & DOTF_MULT)) return bad(s, l, "multiplier without the step flag", mult);
          if ((d[0] & (1u << 19)) && !(st.p1 & DOTF_HALVE)) return bad(s, l, "halving without the step flag", d[0]);
          if (((d[0] >> 20) & 0xf) && !(st.p1 & DOTF_OFFS)) return bad(s, l, "offset without the step flag", d[0]);
          if (st.p0 < 8 && (d[1] >> (4 * st.p0))) return bad(s, l, "sign bits beyond the last round", d[1]);
          for (u32 r = 0; r < st.p0; r++) {
            const u32* rd = d + DOT_HDR_WORDS + DOT_ROUND_WORDS * r;
            const u32 sh = ((r < 4 ? st.shape[0] : st.shape[1]) >> (8 * (r & 3))) & 0xff;
            for (int o = 0; o < 2; o++) {
              const u32 mode = (sh >> (3 * o)) & 3, f0 = rd[2 * o], f1 = rd[2 * o + 1];
              if (!term(f0)) return bad(s, l, "first term of an operand", f0);
              if (mode == 0 ? f1 != 0 : !term(f1)) return bad(s, l, "second term of an operand", f1);
              if (mode != 3 && ((d[1] >> (4 * r + 2 * o)) & 3u)) return bad(s, l, "sign bits on an operand without per-lane signs", d[1]);
            }
            if (sh & 0xc0) return bad(s, l, "round shape", sh);
          }
          break;
        }
        case K_LIN: {
          if (!dst(d[0])) return bad(s, l, "destination", d[0]);
          const u32 nt = (u32)st.p0 + st.p1;
          if (st.p0 < 1 || st.p0 > MAX_LIN_TERMS || st.p1 > MAX_LIN_TERMS || 1 + (nt + 1) / 2 > st.stride) return bad(s, l, "term counts", nt);
          for (u32 t = 0; t < nt; t++) { const u32 f = (d[1 + t / 2] >> (16 * (t & 1))) & 0xffffu; if (!src(f)) return bad(s, l, "term", f); }
          break;
        }
        case K_LOAD: case K_LOADW: if (!dst(d[0]) || ((d[0] >> 16) & 0xffff) >= (u32)MAX_BUFS) return bad(s, l, "load", d[0]); if (st.kind == K_LOAD && (st.p0 % 4 || st.p0 > 48)) return bad(s, l, "load width", st.p0); break;
        case K_STORE: case K_STOREW: if (!src(d[0]) || ((d[0] >> 16) & 0xffff) >= (u32)MAX_BUFS) return bad(s, l, "store", d[0]); break;
        case K_ISZ: case K_CANON: if (!dst(d[0]) || !src(d[0] >> 16)) return bad(s, l, "unary", d[0]); break;
        case K_BIT: if (!dst(d[0]) || !src(d[0] >> 16) || d[1] >= 392) return bad(s, l, "bit", d[1]); break;
        case K_SEL: if (!dst(d[0]) || !src(d[0] >> 16) || !src(d[1]) || !src(d[1] >> 16)) return bad(s, l, "select", d[1]); break;
        case K_CMP: case K_FLAG: case K_BITAND: if (!dst(d[0]) || !src(d[1]) || !src(d[1] >> 16)) return bad(s, l, "binary", d[1]); if (st.kind != K_BITAND && st.p0 > 3) return bad(s, l, "predicate", st.p0); break;
        case K_STATUS: {
          const u32 n = d[0] & 0xff; if (n > 7 || st.stride < 8 || ((d[0] >> 16) & 0xffff) >= (u32)MAX_BUFS) return bad(s, l, "status", d[0]);
          for (u32 k = 0; k < n; k++) if (!src(d[1 + k])) return bad(s, l, "status flag", d[1 + k]);
          break;
        }
        default: return bad(s, l, "step kind", st.kind);
      }
    }
  }
  return std::string();
}

}  // namespace nbls
