// trace.h -- host-side "compiler" of the wave VM: symbolic Fp values, lazy linear forms, DAG construction,
// list scheduling into wave-wide steps and LDS slot allocation.  Runs once per program at nbls_init().
//
// A symbolic Fp value (SFp) is a linear form  sum coef_i * atom_i  over ATOMS = values that live in an LDS slot
// (inputs, constants, products, materialised sums, select results).  Additions, subtractions, negations and
// multiplications by small integers only edit the form; a form is materialised by a K_LIN lane-op when it is
// needed as a multiplication operand with more than two terms, as an output, or when it grows past TMAX terms.
// Every atom is kept in [0,2p); a MUL operand may be (x) or (x +- y) -- the pre-addition is fused into the
// multiplication lane-op.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "vm.h"

namespace nbls {

typedef uint32_t u32;

struct Program {
  std::string name;
  std::vector<Step> steps;
  std::vector<u32> descs;
  std::vector<u32> consts;   // nconst*12 words + PM2 table (17*16 words)
  u32 nconst = 0, W = 64, G = 1, slots = 0;
  // statistics
  u32 n_mul_steps = 0, n_lin_steps = 0, n_other_steps = 0, n_mul_ops = 0, n_lin_ops = 0, n_lin_terms = 0;
  u32 lds_bytes() const { return lds_words(nconst, G, slots) * 4; }
};

typedef std::vector<std::pair<int, int>> Form;   // (atom, coef), sorted by atom, no zero coefs

struct Node {
  uint8_t kind = 0;        // StepKind, or 0xff for constants
  uint8_t p0 = 0;
  bool halve = false;       // LIN: divide the reduced sum by two (mod p)
  int a0 = -1, a1 = -1, am = 0, b0 = -1, b1 = -1, bm = 0;   // MUL operands / generic sources (a0 = src, a1 = second src, b0 = flag)
  std::vector<std::pair<int, int>> lin;                    // LIN terms (atom, sign)
  std::vector<std::pair<int, int>> stat;                   // STATUS (flag atom, code)
  int buf = 0, off = 0;
  int const_idx = -1;
  // scheduling state
  bool live = false;
  int step = -1, lane = 0, slot = -1, last_use = -1, height = 0, ndeps = 0;
  std::vector<int> users;
};

struct SFp;
struct Builder {
  std::vector<Node> nodes;
  std::vector<u32> const_words;                      // 12 words per constant
  std::map<std::vector<u32>, int> const_map;         // limbs -> node id
  std::map<std::vector<int>, int> mul_cse;
  std::map<Form, int> lin_cse, halve_cse;
  int zero_atom = -1, one_atom = -1, r2_atom = -1, rawone_atom = -1;
  int TMAX = 10;
  static Builder*& cur() { static thread_local Builder* b = nullptr; return b; }
  Builder();
  ~Builder() { if (cur() == this) cur() = nullptr; }

  int add_node(const Node& n) { nodes.push_back(n); return (int)nodes.size() - 1; }
  int const_atom(const u32* limbs) {
    std::vector<u32> key(limbs, limbs + 12);
    auto it = const_map.find(key);
    if (it != const_map.end()) return it->second;
    Node n; n.kind = 0xff; n.const_idx = (int)const_words.size() / 12;
    const_words.insert(const_words.end(), key.begin(), key.end());
    int id = add_node(n); const_map[key] = id; return id;
  }
  Program compile(const std::string& name, int W);
};

// ---------------------------------------------------------------------------------------------- SFp
struct SFp {
  Form f;
  SFp() {}
  explicit SFp(int atom) { f.push_back({atom, 1}); }
  bool is_zero() const { return f.empty(); }
  int weight() const { int w = 0; for (auto& t : f) w += std::abs(t.second); return w; }
};

static inline Form form_add(const Form& a, const Form& b, int sb) {
  Form r; r.reserve(a.size() + b.size());
  size_t i = 0, j = 0;
  while (i < a.size() || j < b.size()) {
    if (j >= b.size() || (i < a.size() && a[i].first < b[j].first)) r.push_back(a[i++]);
    else if (i >= a.size() || b[j].first < a[i].first) { r.push_back({b[j].first, sb * b[j].second}); j++; }
    else { int c = a[i].second + sb * b[j].second; if (c) r.push_back({a[i].first, c}); i++; j++; }
  }
  return r;
}

int materialize(const SFp& x);

static inline SFp lin_combine(SFp a, SFp b, int sb) {
  Builder* B = Builder::cur();
  SFp r; r.f = form_add(a.f, b.f, sb);
  if (r.weight() > B->TMAX) {
    // keep forms small: put the heavier operand into a slot first
    if (a.weight() >= b.weight() && a.weight() > 1) a = SFp(materialize(a)); else if (b.weight() > 1) b = SFp(materialize(b));
    r.f = form_add(a.f, b.f, sb);
    if (r.weight() > B->TMAX) {
      if (a.weight() > 1) a = SFp(materialize(a));
      if (b.weight() > 1) b = SFp(materialize(b));
      r.f = form_add(a.f, b.f, sb);
    }
  }
  return r;
}
static inline SFp operator+(const SFp& a, const SFp& b) { return lin_combine(a, b, 1); }
static inline SFp operator-(const SFp& a, const SFp& b) { return lin_combine(a, b, -1); }
static inline SFp operator-(const SFp& a) { SFp r = a; for (auto& t : r.f) t.second = -t.second; return r; }
static inline SFp scale(const SFp& a, int k) {
  Builder* B = Builder::cur();
  SFp x = a;
  if (std::abs(k) * x.weight() > B->TMAX && x.weight() > 1) x = SFp(materialize(x));
  SFp r; if (k == 0) return r;
  r = x; for (auto& t : r.f) t.second *= k; return r;
}

inline int materialize(const SFp& x) {
  Builder* B = Builder::cur();
  if (x.f.empty()) return B->zero_atom;
  if (x.f.size() == 1 && x.f[0].second == 1) return x.f[0].first;
  auto it = B->lin_cse.find(x.f);
  if (it != B->lin_cse.end()) return it->second;
  // expand coefficients into repeated terms; split when too many
  std::vector<std::pair<int, int>> terms;
  for (auto& t : x.f) for (int k = 0; k < std::abs(t.second); k++) terms.push_back({t.first, t.second > 0 ? 1 : -1});
  while ((int)terms.size() > MAX_LIN_TERMS) {
    // fold the first chunk into its own LIN atom
    Node n; n.kind = K_LIN; n.lin.assign(terms.begin(), terms.begin() + (MAX_LIN_TERMS));
    int id = B->add_node(n);
    terms.erase(terms.begin(), terms.begin() + (MAX_LIN_TERMS));
    terms.insert(terms.begin(), {id, 1});
  }
  Node n; n.kind = K_LIN; n.lin = terms;
  int id = B->add_node(n);
  B->lin_cse[x.f] = id;
  return id;
}

// x/2 mod p of a linear form, as one LIN lane-op with the halve flag
static inline SFp halve(const SFp& x) {
  Builder* B = Builder::cur();
  if (x.f.empty()) return SFp();
  auto it = B->halve_cse.find(x.f);
  if (it != B->halve_cse.end()) return SFp(it->second);
  SFp y = x;
  if (y.weight() > MAX_LIN_TERMS) y = SFp(materialize(y));
  Node n; n.kind = K_LIN; n.halve = true;
  for (auto& t : y.f) for (int k = 0; k < std::abs(t.second); k++) n.lin.push_back({t.first, t.second > 0 ? 1 : -1});
  int id = B->add_node(n);
  B->halve_cse[x.f] = id;
  return SFp(id);
}

struct MulOperand { int sign, a0, a1, mode; };
static inline int igcd(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }
// `sign` carries an integer factor pulled out of the operand: (g * form) * y = g * (form * y)
static inline MulOperand mul_operand(const SFp& x) {
  Form f = x.f;
  int g = 0; for (auto& t : f) g = igcd(g, std::abs(t.second));
  if (f.size() == 1 && g == 2) g = 1;            // 2x is served by the fused pre-addition x + x
  if (g > 1) for (auto& t : f) t.second /= g;
  if (f.size() == 1 && std::abs(f[0].second) == 1) return {g * f[0].second, f[0].first, -1, 0};
  if (f.size() == 1 && std::abs(f[0].second) == 2) return {f[0].second / 2, f[0].first, f[0].first, 1};
  if (f.size() == 2 && std::abs(f[0].second) == 1 && std::abs(f[1].second) == 1) {
    if (f[0].second == 1) return {g, f[0].first, f[1].first, f[1].second == 1 ? 1 : 2};
    if (f[1].second == 1) return {g, f[1].first, f[0].first, 2};
    return {-g, f[0].first, f[1].first, 1};
  }
  SFp y; y.f = f;
  return {g, materialize(y), -1, 0};
}

static inline SFp mul(const SFp& a, const SFp& b) {
  Builder* B = Builder::cur();
  if (a.is_zero() || b.is_zero()) return SFp();
  MulOperand A = mul_operand(a), Bo = mul_operand(b);
  // multiplication by the constant one
  if (A.a1 < 0 && A.a0 == B->one_atom) { SFp r = b; if (A.sign < 0) r = -r; return r; }
  if (Bo.a1 < 0 && Bo.a0 == B->one_atom) { SFp r = a; if (Bo.sign < 0) r = -r; return r; }
  std::vector<int> ka = {A.a0, A.a1, A.mode}, kb = {Bo.a0, Bo.a1, Bo.mode};
  if (kb < ka) { std::swap(ka, kb); std::swap(A, Bo); }
  std::vector<int> key = ka; key.insert(key.end(), kb.begin(), kb.end());
  int id;
  auto it = B->mul_cse.find(key);
  if (it != B->mul_cse.end()) id = it->second;
  else {
    Node n; n.kind = K_MUL; n.a0 = A.a0; n.a1 = A.a1; n.am = A.mode; n.b0 = Bo.a0; n.b1 = Bo.a1; n.bm = Bo.mode;
    id = B->add_node(n); B->mul_cse[key] = id;
  }
  SFp r(id); r.f[0].second = A.sign * Bo.sign; return r;
}
static inline SFp sqr(const SFp& a) { return mul(a, a); }

// ---- constants, inputs, outputs
static inline SFp constant(const u32* mont_limbs) {
  Builder* B = Builder::cur();
  bool z = true; for (int i = 0; i < 12; i++) z = z && mont_limbs[i] == 0;
  if (z) return SFp();
  return SFp(B->const_atom(mont_limbs));
}
SFp input(int buf, int off);          // big-endian wire bytes -> Montgomery value
void output(const SFp& x, int buf, int off);
static inline SFp inputw(int buf, int off) { Node n; n.kind = K_LOADW; n.buf = buf; n.off = off; return SFp(Builder::cur()->add_node(n)); }
static inline void outputw(const SFp& x, int buf, int off) { Node n; n.kind = K_STOREW; n.a0 = materialize(x); n.buf = buf; n.off = off; n.live = true; Builder::cur()->add_node(n); }
// flags (raw 0/1 integers in a slot)
static inline SFp is_zero(const SFp& x) { Node n; n.kind = K_ISZ; n.a0 = materialize(x); return SFp(Builder::cur()->add_node(n)); }
static inline SFp select(const SFp& flag, const SFp& a, const SFp& b) {
  Node n; n.kind = K_SEL; n.b0 = materialize(flag); n.a0 = materialize(a); n.a1 = materialize(b); return SFp(Builder::cur()->add_node(n));
}
static inline SFp canon(const SFp& x) { Node n; n.kind = K_CANON; n.a0 = materialize(x); return SFp(Builder::cur()->add_node(n)); }
static inline SFp cmp_gt(const SFp& a, const SFp& b) { Node n; n.kind = K_CMP; n.p0 = 0; n.a0 = materialize(a); n.a1 = materialize(b); return SFp(Builder::cur()->add_node(n)); }
static inline SFp is_odd(const SFp& a) { Node n; n.kind = K_CMP; n.p0 = 1; n.a0 = materialize(a); n.a1 = n.a0; return SFp(Builder::cur()->add_node(n)); }
static inline SFp flag_op(int op, const SFp& a, const SFp& b) { Node n; n.kind = K_FLAG; n.p0 = (uint8_t)op; n.a0 = materialize(a); n.a1 = materialize(b); return SFp(Builder::cur()->add_node(n)); }
static inline SFp f_and(const SFp& a, const SFp& b) { return flag_op(0, a, b); }
static inline SFp f_or(const SFp& a, const SFp& b) { return flag_op(1, a, b); }
static inline SFp f_xor(const SFp& a, const SFp& b) { return flag_op(2, a, b); }
static inline SFp f_andnot(const SFp& a, const SFp& b) { return flag_op(3, a, b); }
// int8 status[item] = code of the first flag that is 0 (flags listed most significant first), else 0
static inline void status_out(const std::vector<std::pair<SFp, int>>& checks, int buf) {
  Node n; n.kind = K_STATUS; n.buf = buf; n.live = true;
  for (auto& c : checks) n.stat.push_back({materialize(c.first), c.second});
  Builder::cur()->add_node(n);
}

}  // namespace nbls
