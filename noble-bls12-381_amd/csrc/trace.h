// trace.h -- host-side "compiler" of the wave VM: symbolic Fp values, lazy bilinear forms, DAG construction,
// list scheduling into wave-wide steps and LDS slot allocation.  Runs once per program at nbls_init().
//
// A symbolic Fp value (SFp) is a lazy form
//        sum_i c_i * atom_i  +  sum_j d_j * (A_j x B_j)
// over ATOMS (values that live in an LDS slot: inputs, constants, results of earlier lane-ops) and PENDING PRODUCTS
// whose operands A_j, B_j are (+-atom) or (+-atom +- atom).  Additions, subtractions, negations and multiplications by
// small integers only edit the form; multiplying two forms yields a one-product form.  A form is materialised -- turned
// into ONE lane-op -- only when it is needed as a multiplication operand that is not of the (x) / (x +- y) shape, as an
// output, or when it outgrows a lane-op.  The lane-op (K_DOT) evaluates  m * REDC(sum_j A_j * B_j) +- atoms  with a
// single Montgomery reduction, so the Karatsuba / tower recombination of Fp2, Fp6 and Fp12 arithmetic costs no extra
// steps: an Fp12 coefficient is produced directly from the operand slots.  Every atom is kept in [0, 2p).
#pragma once
#include <algorithm>
#include "config.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>
#include "vm.h"

namespace nbls {

typedef uint32_t u32;

struct Program {
  std::string name;
  std::vector<Step> steps;
  std::vector<u32> descs;
  std::vector<u32> consts;   // nconst * RAW_WORDS words
  u32 nconst = 0, W = 64, G = 1, slots = 0;
  u32 slot_bytes = 64;       // LDS slot stride chosen for this program (Builder::compile)
  // statistics
  u32 n_dot_steps = 0, n_lin_steps = 0, n_other_steps = 0, n_dot_ops = 0, n_products = 0, n_prod_slots = 0, n_lin_ops = 0, n_lin_terms = 0, n_norm_operands = 0, n_neg_operands = 0, n_comb_operands = 0;
  u32 n_round_ops = 0, n_op_mode[4] = {0, 0, 0, 0}, n_op_norm = 0;   // product-round operands (two per round) by shape
  double est_valu = 0;       // cost-model estimate of VALU instructions per wave (see Builder::compile)
  // buffers a caller may bind to the SAME memory (out, in): verify_program checks that every load of `in` precedes the first store to `out` (round 6; the in-place product tree
  // binds buffer 5 of fp12_mul2s onto buffer 3 -- safe because every output coefficient depends on all inputs, and now checked instead of assumed)
  std::vector<std::pair<int, int>> aliases;
  std::vector<u32> buf_extent = std::vector<u32>(MAX_BUFS, 0);   // per buffer index: bytes of one item the program touches (max offset + size); launch check in checked builds
  u32 lsplit = 1;            // lane split: every K_DOT lane-op is spread over `lsplit` adjacent lanes, each accumulating a share of the products; the columns are summed
                             // across them before the one reduction (latency variant for launches of at most one wavefront per SIMD).  W is the PHYSICAL lane count.
  bool shared_consts = false;   // one shared copy of the constants instead of one per instance (chosen by compile() for G >= 8)
  u32 inst_bytes() const { return (shared_consts ? slots : nconst + slots) * slot_bytes; }   // one instance region: [constants +] slots
  u32 inst_base(u32 g) const { return (shared_consts ? nconst * slot_bytes : 0) + g * inst_bytes(); }
  u32 lds_bytes() const { return inst_base(G); }
};

// operand of a pending product: s0 (never negated after canonicalisation) and optional s1 with sign
struct Operand {
  int s0 = -1, s1 = -1; bool n1 = false;
  bool operator<(const Operand& o) const { return std::tie(s0, s1, n1) < std::tie(o.s0, o.s1, o.n1); }
  bool operator==(const Operand& o) const { return s0 == o.s0 && s1 == o.s1 && n1 == o.n1; }
};
struct ProdKey { Operand a, b; bool operator<(const ProdKey& o) const { return std::tie(a, b) < std::tie(o.a, o.b); } };

typedef long long TermKey;                         // atom id, or PROD_BASE + product id
static const TermKey PROD_BASE = 1LL << 40;
typedef std::vector<std::pair<TermKey, int>> Form;   // sorted by key, no zero coefficients

struct DotProduct { Operand a, b; bool neg; bool norm_a = false, norm_b = false; bool neg0_a = false, neg0_b = false; };   // neg: the product enters with a minus sign (resolved into an operand shape by compile()); norm_*: normalise that (sum) operand first; neg0_*: the operand's first term is negated too (-(x + y), per-lane signs)

struct Node {
  uint8_t kind = 0;        // StepKind, or 0xff for constants
  uint8_t p0 = 0;
  bool halve = false;      // DOT/LIN: divide the reduced result by two (mod p)
  bool wred = false;       // DOT/LIN: weak reduction after the post-added terms (result below 3.02 p whatever the terms' bounds)
  bool raw = false;        // K_LOAD result: any 384-bit integer (not < 2p)
  int a0 = -1, a1 = -1, b0 = -1;                           // generic sources (a0 = src, a1 = second src, b0 = flag)
  std::vector<DotProduct> prods; int mult = 1; int offs = 0;   // DOT: m * (REDC(sum of products) + offs * p) +- lin
  std::vector<std::pair<int, int>> lin;                    // DOT / LIN linear terms (atom, sign)
  std::vector<std::pair<int, int>> stat;                   // STATUS (flag atom, code)
  int buf = 0, off = 0;
  int const_idx = -1;
  double bound = 2.0;      // magnitude bound of the stored value in units of p (values are normalised 28-bit limbs)
  // scheduling state
  bool live = false;
  int step = -1, lane = 0, slot = -1, last_use = -1, height = 0, ndeps = 0;
  std::vector<int> users;
};

struct SFp;
struct Builder {
  std::vector<Node> nodes;
  std::vector<u32> const_words;                      // SLOT_WORDS words per constant (14 limbs + padding)
  std::map<std::vector<u32>, int> const_map;         // limbs -> node id
  std::vector<ProdKey> prods;
  std::map<ProdKey, int> prod_map;
  std::map<Form, int> mat_cse, halve_cse;
  std::map<int, int> small_consts;                   // integer -> constant atom (Montgomery form)
  std::map<int, int> contract_cache;                 // atom -> contracted atom
  int zero_atom = -1, one_atom = -1, r2_atom = -1, rawone_atom = -1;
  int TMAX = 12;         // soft cap on the size of a lazy form
  int max_dot = MAX_DOT_PRODUCTS;   // products per lane-op: a lower cap splits heavy lane-ops (first chunk, then the rest + the first as a post-added term) so that the few heaviest lanes do not set the length of a step
  int light_max = 2;     // lane-ops with at most this many products form the "light" class of the scheduler (they ride in the free lanes of heavy steps; MAX_DOT_PRODUCTS = one class)
  double neg_cap = 6.0;  // negated linear terms above this bound are contracted first (their bound is paid as a +k p offset)
  bool use_wred = !env_set("NBLS_NO_WRED");   // large post-added terms: weak reduction (table of multiples of p) instead of folding them into the dot product
  int store_batch = 0;   // > 0: a store step is issued as soon as this many stores are ready (programs that stream results out: the values do not linger in LDS)
  int lane_split = 1;    // see Program::lsplit (compile(name, W) takes the LOGICAL lanes per item; the program runs on W * lane_split)
  int shared_consts = -1;   // -1: one shared copy of the constants for programs with eight or more items per wavefront (compile()); 0 / 1: replicated / shared
  int sched_window = 0;  // scheduler look-ahead limit in critical-path units (0 = unlimited), see compile()
  bool prefer_doubling = env_long("NBLS_PREFER_DBL", 1) != 0;   // materialize(): integer factors of a lane-op's products as doubled operands rather than a multiplier on the reduced sum
  static Builder*& cur() { static thread_local Builder* b = nullptr; return b; }
  Builder();
  ~Builder() { if (cur() == this) cur() = nullptr; }

  int add_node(const Node& n) { nodes.push_back(n); return (int)nodes.size() - 1; }
  int const_atom(const u32* limbs) {
    std::vector<u32> key(limbs, limbs + NLIMBS);
    auto it = const_map.find(key);
    if (it != const_map.end()) return it->second;
    Node n; n.kind = 0xff; n.const_idx = (int)const_words.size() / SLOT_WORDS; n.bound = 1.0;
    const_words.insert(const_words.end(), key.begin(), key.end());
    const_words.push_back(0); const_words.push_back(0);
    int id = add_node(n); const_map[key] = id; return id;
  }
  int small_const(int c);                            // Montgomery form of a small positive integer
  int frac_const(int c, int m);                      // Montgomery form of c / m mod p  (c small, possibly negative; m in 1..4)
  std::map<std::pair<int, int>, int> frac_consts;
  int product(const Operand& a, const Operand& b) {
    ProdKey k{a, b}; if (k.b < k.a) std::swap(k.a, k.b);
    auto it = prod_map.find(k); if (it != prod_map.end()) return it->second;
    prods.push_back(k); prod_map[k] = (int)prods.size() - 1; return (int)prods.size() - 1;
  }
  bool is_const(int atom) const { return nodes[atom].kind == 0xff; }
  // magnitude bound of an atom in units of p
  double atom_bound(int atom) const { return nodes[atom].bound; }
  // magnitude bound of a (signed) product operand
  double operand_bound(const Operand& o) const { return atom_bound(o.s0) + (o.s1 >= 0 ? atom_bound(o.s1) : 0.0); }
  int kp_atom(int k);                                // the constant k * p (normalised limbs), used to keep sums with negative terms non-negative
  int contract(int atom);                            // x -> x * R / R : same value mod p, bound ~1
  int lower_bound(int atom, double cap);             // bound <= cap by a weak reduction at the producer where possible, else by contraction
  Program compile(const std::string& name, int W);
};

// ---------------------------------------------------------------------------------------------- SFp
struct SFp {
  Form f;
  SFp() {}
  explicit SFp(int atom) { f.push_back({(TermKey)atom, 1}); }
  bool is_zero() const { return f.empty(); }
  int weight() const { return (int)f.size(); }
  bool pure_atoms() const { for (auto& t : f) if (t.first >= PROD_BASE) return false; return true; }
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

int materialize(const SFp& x, bool halve_it = false);

static inline SFp lin_combine(SFp a, SFp b, int sb) {
  Builder* B = Builder::cur();
  SFp r; r.f = form_add(a.f, b.f, sb);
  if (r.weight() > B->TMAX) {
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
static inline SFp scale(const SFp& a, int k) { SFp r; if (k == 0) return r; r = a; for (auto& t : r.f) t.second *= k; return r; }
static inline SFp halve(const SFp& x) { if (x.f.empty()) return SFp(); return SFp(materialize(x, true)); }

// Turn a form into a product operand: returns the integer factor pulled out (may be negative).  The form must
// consist of one or two atoms with equal |coefficient|; anything else is materialised first.
static inline int as_operand(const SFp& x, Operand& o) {
  const Form& f = x.f;
  bool ok = x.pure_atoms() && (f.size() == 1 || (f.size() == 2 && std::abs(f[0].second) == std::abs(f[1].second)));
  if (!ok) { o.s0 = materialize(x); o.s1 = -1; o.n1 = false; return 1; }
  int g = std::abs(f[0].second);
  if (f.size() == 1) {
    if (g % 2 == 0) { o.s0 = o.s1 = (int)f[0].first; o.n1 = false; return f[0].second / 2; }   // 2x = x + x in the pre-addition
    o.s0 = (int)f[0].first; o.s1 = -1; o.n1 = false; return f[0].second;
  }
  int c0 = f[0].second / g, c1 = f[1].second / g;
  if (c0 == 1) { o.s0 = (int)f[0].first; o.s1 = (int)f[1].first; o.n1 = c1 < 0; return g; }
  if (c1 == 1) { o.s0 = (int)f[1].first; o.s1 = (int)f[0].first; o.n1 = true; return g; }
  o.s0 = (int)f[0].first; o.s1 = (int)f[1].first; o.n1 = false; return -g;
}

static inline SFp mul(const SFp& a, const SFp& b) {
  Builder* B = Builder::cur();
  if (a.is_zero() || b.is_zero()) return SFp();
  Operand A, Bo;
  int ga = as_operand(a, A), gb = as_operand(b, Bo);
  if (A.s1 < 0 && A.s0 == B->one_atom) return scale(b, ga);
  if (Bo.s1 < 0 && Bo.s0 == B->one_atom) return scale(a, gb);
  if (A.s1 < 0 && A.s0 == B->zero_atom) return SFp();
  if (Bo.s1 < 0 && Bo.s0 == B->zero_atom) return SFp();
  SFp r; r.f.push_back({PROD_BASE + B->product(A, Bo), ga * gb});
  return r;
}
static inline SFp sqr(const SFp& a) { return mul(a, a); }

// ---- constants, inputs, outputs
static inline SFp constant(const u32* mont_limbs) {
  Builder* B = Builder::cur();
  bool z = true; for (int i = 0; i < NLIMBS; i++) z = z && mont_limbs[i] == 0;
  if (z) return SFp();
  return SFp(B->const_atom(mont_limbs));
}
SFp input(int buf, int off);          // big-endian wire bytes -> Montgomery value
void output_raw(const SFp& x, int buf, int off);       // raw integer (< 2^384) -> 48 big-endian bytes, no reduction
SFp input_raw(int buf, int off, int nbytes = 48);   // big-endian integer of nbytes (multiple of 4, <= 48), NOT in Montgomery form
SFp to_mont(const SFp& raw);          // raw integer (< 2^384) -> Montgomery value
SFp raw_const(const u32* limbs);      // constant raw integer
static inline SFp bit_flag(const SFp& raw, int bit) { Node n; n.kind = K_BIT; n.bound = 0.001; n.a0 = materialize(raw); n.off = bit; return SFp(Builder::cur()->add_node(n)); }
static inline SFp bit_and(const SFp& a, const SFp& b) { Node n; n.kind = K_BITAND; n.bound = 1.3; n.a0 = materialize(a); n.a1 = materialize(b); return SFp(Builder::cur()->add_node(n)); }
static inline SFp f_not(const SFp& a);
void output(const SFp& x, int buf, int off);
// raw scratch element k of an item sits at byte k * RAW_FP_BYTES; `off` is given as k * 48 (one wire-sized element per raw element)
static inline int raw_off(int off) { return off / 48 * RAW_FP_BYTES; }
static inline SFp inputw(int buf, int off) { Node n; n.kind = K_LOADW; n.buf = buf; n.off = raw_off(off); n.bound = 8.0; return SFp(Builder::cur()->add_node(n)); }
static inline void outputw(const SFp& x, int buf, int off) {
  Builder* B = Builder::cur(); Node n; n.kind = K_STOREW; n.a0 = materialize(x);
  n.a0 = B->lower_bound(n.a0, 8.0);     // scratch elements are reloaded with bound 8
  n.buf = buf; n.off = raw_off(off); n.live = true; B->add_node(n);
}
// flags (raw 0/1 integers in a slot)
static inline SFp is_zero(const SFp& x) { Builder* B = Builder::cur(); Node n; n.kind = K_ISZ; n.a0 = B->contract(materialize(x)); n.bound = 0.001; return SFp(B->add_node(n)); }   // zero test needs a value below 2p
static inline SFp select(const SFp& flag, const SFp& a, const SFp& b) {
  Builder* B = Builder::cur(); Node n; n.kind = K_SEL; n.b0 = materialize(flag); n.a0 = materialize(a); n.a1 = materialize(b);
  n.bound = std::max(B->atom_bound(n.a0), B->atom_bound(n.a1)); return SFp(B->add_node(n));
}
static inline SFp canon(const SFp& x) { Node n; n.kind = K_CANON; n.a0 = materialize(x); n.bound = 1.0; return SFp(Builder::cur()->add_node(n)); }   // input must be below 2p (callers contract first)
static inline SFp cmp_gt(const SFp& a, const SFp& b) { Node n; n.kind = K_CMP; n.bound = 0.001; n.p0 = 0; n.a0 = materialize(a); n.a1 = materialize(b); return SFp(Builder::cur()->add_node(n)); }
static inline SFp is_odd(const SFp& a) { Node n; n.kind = K_CMP; n.bound = 0.001; n.p0 = 1; n.a0 = materialize(a); n.a1 = n.a0; return SFp(Builder::cur()->add_node(n)); }
static inline SFp flag_op(int op, const SFp& a, const SFp& b) { Node n; n.kind = K_FLAG; n.bound = 0.001; n.p0 = (uint8_t)op; n.a0 = materialize(a); n.a1 = materialize(b); return SFp(Builder::cur()->add_node(n)); }
static inline SFp f_and(const SFp& a, const SFp& b) { return flag_op(0, a, b); }
static inline SFp f_or(const SFp& a, const SFp& b) { return flag_op(1, a, b); }
static inline SFp f_xor(const SFp& a, const SFp& b) { return flag_op(2, a, b); }
static inline SFp f_andnot(const SFp& a, const SFp& b) { return flag_op(3, a, b); }
static inline SFp f_not(const SFp& a) { return f_andnot(SFp(Builder::cur()->rawone_atom), a); }
// int8 status[item] = code of the first flag that is 0 (flags listed most significant first), else 0
static inline void status_out(const std::vector<std::pair<SFp, int>>& checks, int buf) {
  Node n; n.kind = K_STATUS; n.buf = buf; n.live = true;
  for (auto& c : checks) n.stat.push_back({materialize(c.first), c.second});
  Builder::cur()->add_node(n);
}

// Static verifier of a compiled program (checked builds, NBLS_CHECKED=1, and the CPU test-suite): programs are straight-line with static
// addresses, so EVERY LDS access, descriptor read and buffer offset the kernel will ever make can be checked on the host before the first launch:
// offsets inside the instance region and 16-byte aligned, destinations outside the constant region, descriptors inside the program, round
// shapes consistent with the lane descriptors, buffer indices valid.  Returns an empty string or the first violation.
std::string verify_program(const Program& p);
// q p for q = 0 .. QP_TABLE_ENTRIES-1 as normalised limbs, RAW_WORDS words per entry (weak_reduce, vm_exec.h); built once on the host
const u32* qp_table_words();

}  // namespace nbls
