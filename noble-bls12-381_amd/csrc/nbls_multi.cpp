// nbls_multi.cpp -- several GPUs of one node behind ONE handle: a context per device, one host thread and one
// stream per device for the duration of a call, contiguous shards.
//   * independent pairings: no exchange at all, results land in disjoint slices of the caller's buffer;
//   * Miller product / verifyBatch: every device reduces its shard to ONE Fp12 partial (576 bytes, no final exponentiation), the partials
//     are copied device-to-device (hipMemcpyPeer over xGMI) into the first device's gather buffer, which multiplies them and runs the one
//     shared final exponentiation.  The exchange is 576 bytes per device -- pure latency -- so no collective library is involved.
// nbls_pool_* (the end of this file): several contexts of ONE device fed round-robin -- the in-flight form of bench.py's headline, for C callers.
// This layer uses only the public single-device ABI (include/nbls.h) plus HIP for the peer copies.
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "nbls.h"
#include "config.h"

#define EXPORT extern "C" __attribute__((visibility("default")))

// Buffers of ONE product call in flight: a 576-byte partial on every device and, on dev[0], the gather area (one partial per device, then 576
// bytes of result).  A call takes a free set (or allocates one) and returns it when it is done, so that calls racing on one handle -- the
// N-API addon runs nbls_multi_verify_batch on libuv worker threads -- never see each other's partials: the contexts serialise the per-device
// work themselves, and everything between a context's *_partial call and the end of finish() touches only the call's own set.
struct GatherLanes;
struct CallBuffers {
  std::vector<uint8_t*> part;   // part[g] on dev[g]
  uint8_t* gather = nullptr;    // on dev[0]
  GatherLanes* lanes = nullptr; // copy streams / events of this set (created with it, on dev[0])
};
// One host thread per device for the LIFETIME of the handle (round 5; round 4 created G - 1 std::threads per call, which shows in the latency of a single verify on
// eight GPUs): a call posts its per-device work to the workers of devices 1 .. G-1, runs device 0's share itself and waits on a latch.  Calls racing on one handle
// queue up per device in arrival order -- the contexts serialise their work anyway.
struct Worker {
  std::thread th; std::mutex mu; std::condition_variable cv; std::deque<std::function<void()>> q; bool stop = false;
  void run() {
    for (;;) {
      std::function<void()> f;
      { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return stop || !q.empty(); }); if (q.empty()) return; f = std::move(q.front()); q.pop_front(); }
      f();
    }
  }
  void post(std::function<void()> f) { { std::lock_guard<std::mutex> l(mu); q.push_back(std::move(f)); } cv.notify_one(); }
  void start() { th = std::thread([this] { run(); }); }
  void finish() { { std::lock_guard<std::mutex> l(mu); stop = true; } cv.notify_one(); if (th.joinable()) th.join(); }
};
struct nbls_multi {
  std::vector<Worker*> workers;        // workers[g] serves device g of the handle (none for g = 0: the caller's thread)
  std::vector<nbls_ctx*> ctx;
  std::vector<int> dev;
  std::vector<int> peer;               // per device: 1 = peer access to / from the reducing device enabled (nbls_multi_peer_access)
  std::mutex mu;                       // guards `free_sets` only
  std::vector<CallBuffers*> free_sets;
  std::vector<CallBuffers*> all_sets;
};
// the gather of one product call (round 6): every device's partial travels on a stream of its own and the reduction waits for all of them through events -- the copies of an
// eight-device node overlap instead of running one after the other (round 5: a loop of synchronous hipMemcpyPeer).  The streams live on the reducing device and belong to the
// call's buffer set, so racing calls do not share them.
struct GatherLanes { std::vector<hipStream_t> s; std::vector<hipEvent_t> e; hipStream_t fin = nullptr; };

static void free_set(nbls_multi* m, CallBuffers* b) {
  for (size_t g = 0; g < b->part.size(); g++) if (b->part[g]) { hipSetDevice(m->dev[g]); hipFree(b->part[g]); }
  hipSetDevice(m->dev[0]);
  if (b->gather) hipFree(b->gather);
  if (b->lanes) {
    for (hipStream_t st : b->lanes->s) if (st) hipStreamDestroy(st);
    for (hipEvent_t e : b->lanes->e) if (e) hipEventDestroy(e);
    if (b->lanes->fin) hipStreamDestroy(b->lanes->fin);
    delete b->lanes;
  }
  delete b;
}
static CallBuffers* take_set(nbls_multi* m) {
  {
    std::lock_guard<std::mutex> g(m->mu);
    if (!m->free_sets.empty()) { CallBuffers* b = m->free_sets.back(); m->free_sets.pop_back(); return b; }
  }
  int prev = 0; hipGetDevice(&prev);
  CallBuffers* b = new CallBuffers();
  b->part.assign(m->dev.size(), nullptr);
  bool ok = true;
  for (size_t g = 0; g < m->dev.size() && ok; g++) ok = hipSetDevice(m->dev[g]) == hipSuccess && hipMalloc(&b->part[g], 576) == hipSuccess;
  ok = ok && hipSetDevice(m->dev[0]) == hipSuccess && hipMalloc(&b->gather, 576 * (m->dev.size() + 1)) == hipSuccess;
  if (ok) {
    b->lanes = new GatherLanes();
    b->lanes->s.assign(m->dev.size(), nullptr); b->lanes->e.assign(m->dev.size(), nullptr);
    for (size_t g = 0; g < m->dev.size() && ok; g++) ok = hipStreamCreateWithFlags(&b->lanes->s[g], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&b->lanes->e[g],
        hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&b->lanes->fin, hipStreamNonBlocking) == hipSuccess;
  }
  hipSetDevice(prev);
  if (!ok) { free_set(m, b); (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> g(m->mu);
  m->all_sets.push_back(b);
  return b;
}
static void give_set(nbls_multi* m, CallBuffers* b) { std::lock_guard<std::mutex> g(m->mu); m->free_sets.push_back(b); }
struct SetLease {   // returns the set on every exit path
  nbls_multi* m; CallBuffers* b;
  explicit SetLease(nbls_multi* m_) : m(m_), b(take_set(m_)) {}
  ~SetLease() { if (b) give_set(m, b); }
};

EXPORT void nbls_destroy_multi(nbls_multi* m) {
  if (!m) return;
  for (Worker* w : m->workers) if (w) { w->finish(); delete w; }
  for (CallBuffers* b : m->all_sets) free_set(m, b);
  for (nbls_ctx* c : m->ctx) nbls_destroy(c);
  delete m;
}

EXPORT int nbls_init_multi(int n_devices, const int* device_ids, nbls_multi** out) {
  if (!out || n_devices < 0) return NBLS_EINVAL;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return NBLS_ENOGPU;
  if (n_devices == 0) n_devices = count;   // 0 = every visible device
  if (n_devices > count && !device_ids) return NBLS_EINVAL;
  nbls_multi* m = new nbls_multi();
  for (int i = 0; i < n_devices; i++) {
    const int d = device_ids ? device_ids[i] : i;
    nbls_ctx* c = nullptr;
    int r = nbls_init(d, &c);
    if (r) { nbls_destroy_multi(m); return r; }
    m->ctx.push_back(c); m->dev.push_back(d);
  }
  m->workers.assign(m->ctx.size(), nullptr);
  for (size_t g = 1; g < m->ctx.size(); g++) { m->workers[g] = new Worker(); m->workers[g]->start(); }
  { CallBuffers* b = take_set(m); if (!b) { nbls_destroy_multi(m); return NBLS_EHIP; } give_set(m, b); }   // the first call's buffers
  // peer access lets hipMemcpyPeer go straight over xGMI; without it the copy is staged, which is still correct
  // (round 4: both directions between the reducing device -- the FIRST listed one, whatever its id -- and every other device, and the outcome is kept for
  // nbls_multi_peer_access so that a test on a multi-GPU box can say whether the gather really went peer to peer)
  m->peer.assign(m->dev.size(), 1);
  for (size_t i = 1; i < m->dev.size(); i++) {
    if (m->dev[i] == m->dev[0]) continue;   // the same device listed twice: a plain device-to-device copy
    int can01 = 0, can10 = 0;
    const bool ok = hipDeviceCanAccessPeer(&can01, m->dev[0], m->dev[i]) == hipSuccess && can01 && hipDeviceCanAccessPeer(&can10, m->dev[i], m->dev[0]) == hipSuccess && can10;
    if (ok) {
      hipError_t e1 = hipSuccess, e2 = hipSuccess;
      if (hipSetDevice(m->dev[0]) == hipSuccess) e1 = hipDeviceEnablePeerAccess(m->dev[i], 0);
      if (hipSetDevice(m->dev[i]) == hipSuccess) e2 = hipDeviceEnablePeerAccess(m->dev[0], 0);
      m->peer[i] = (e1 == hipSuccess || e1 == hipErrorPeerAccessAlreadyEnabled) && (e2 == hipSuccess || e2 == hipErrorPeerAccessAlreadyEnabled) ? 1 : 0;
    } else m->peer[i] = 0;
  }
  (void)hipGetLastError();
  hipSetDevice(m->dev[0]);
  *out = m;
  return NBLS_OK;
}
// 1: copies between device i of the handle and the reducing (first) device go peer to peer (or i is that device); 0: staged through the host by the runtime; -1: bad index
EXPORT int nbls_multi_peer_access(const nbls_multi* m, int i) { return m && i >= 0 && i < (int)m->peer.size() ? m->peer[i] : -1; }
EXPORT int nbls_multi_device_count(const nbls_multi* m) { return m ? (int)m->ctx.size() : 0; }
EXPORT nbls_ctx* nbls_multi_context(nbls_multi* m, int i) { return m && i >= 0 && i < (int)m->ctx.size() ? m->ctx[i] : nullptr; }

// contiguous shards [lo, hi) of n items over the devices (the first n % G devices take one item more)
static void shard(size_t n, size_t G, size_t g, size_t* lo, size_t* hi) { const size_t q = n / G, r = n % G; *lo = g * q + (g < r ? g : r); *hi = *lo + q + (g < r ? 1 : 0); }

template <class F> static int on_every_device(nbls_multi* m, size_t G, F f) {
  std::vector<int> rc(G, 0);
  std::mutex mu; std::condition_variable cv; size_t pending = G > 1 ? G - 1 : 0;
  for (size_t g = 1; g < G; g++) m->workers[g]->post([&, g] { rc[g] = f(g); std::lock_guard<std::mutex> l(mu); if (--pending == 0) cv.notify_one(); });
  rc[0] = f(0);
  { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return pending == 0; }); }
  for (size_t g = 0; g < G; g++) if (rc[g] && rc[g] != NBLS_EDECODE) return rc[g];
  for (size_t g = 0; g < G; g++) if (rc[g]) return rc[g];
  return NBLS_OK;
}

// pairing(P_i, Q_i) for n independent pairs, sharded over the devices (BASELINE configs[3]) -- reference index.ts:715-722
EXPORT int nbls_multi_pairing_batch(nbls_multi* m, size_t n, const uint8_t* g1, const uint8_t* g2, int with_final_exp, int validate, uint8_t* out, int8_t* status) {
  if (!m || m->ctx.empty() || (n && (!g1 || !g2 || !out))) return NBLS_EINVAL;
  const size_t G = m->ctx.size();
  return on_every_device(m, G, [&](size_t g) {
    size_t lo, hi; shard(n, G, g, &lo, &hi);
    if (hi == lo) return (int)NBLS_OK;
    return nbls_pairing_batch(m->ctx[g], hi - lo, g1 + lo * 96, g2 + lo * 192, with_final_exp, validate, out + lo * 576, status ? status + lo : nullptr);
  });
}

// gather the partials of the first G devices on the first device (the call's own gather area), multiply, one shared final exponentiation, result to the host
static int finish(nbls_multi* m, CallBuffers* b, size_t G, int final_exp, uint8_t* out) {
  if (hipSetDevice(m->dev[0]) != hipSuccess) return NBLS_EHIP;
  GatherLanes* L = b->lanes;
  // every partial on its own stream (the producers have synchronised: on_every_device returned), one event each, the reduction's stream waits for all of them
  for (size_t g = 0; g < G; g++) {
    const hipError_t e = g == 0 || m->dev[g] == m->dev[0] ? hipMemcpyAsync(b->gather + 576 * g, b->part[g], 576, hipMemcpyDeviceToDevice, L->s[g])
                                                           : hipMemcpyPeerAsync(b->gather + 576 * g, m->dev[0], b->part[g], m->dev[g], 576, L->s[g]);
    if (e != hipSuccess || hipEventRecord(L->e[g], L->s[g]) != hipSuccess || hipStreamWaitEvent(L->fin, L->e[g], 0) != hipSuccess) { (void)hipDeviceSynchronize(); return NBLS_EHIP; }
  }
  uint8_t* res = b->gather + 576 * m->dev.size();
  int r = nbls_fp12_product_final_dev(m->ctx[0], G, b->gather, final_exp, res, L->fin);
  if (r) { (void)hipDeviceSynchronize(); return r; }
  if (hipMemcpyAsync(out, res, 576, hipMemcpyDeviceToHost, L->fin) != hipSuccess || hipStreamSynchronize(L->fin) != hipSuccess) return NBLS_EHIP;
  return NBLS_OK;
}

// prod_i millerLoop(P_i, Q_i) with ONE shared final exponentiation (BASELINE configs[4]) -- the core of verify / verifyBatch, index.ts:763-766, 811-816
EXPORT int nbls_multi_miller_product(nbls_multi* m, size_t n, const uint8_t* g1, const uint8_t* g2, int final_exp, int validate, uint8_t* out, int8_t* status) {
  if (!m || m->ctx.empty() || !out || (n && (!g1 || !g2))) return NBLS_EINVAL;
  const size_t G = m->ctx.size();
  SetLease lease(m);
  if (!lease.b) return NBLS_EHIP;
  CallBuffers* b = lease.b;
  int r = on_every_device(m, G, [&](size_t g) {
    size_t lo, hi; shard(n, G, g, &lo, &hi);   // an empty shard contributes the unit element
    void* p = b->part[g];
    return nbls_miller_product_partial_into(m->ctx[g], hi - lo, g1 + lo * 96, g2 + lo * 192, validate, p, status ? status + lo : nullptr);
  });
  if (r) { if (r == NBLS_EDECODE) memset(out, 0, 576); return r; }   // as nbls_miller_product: a rejected input zeroes the output
  return finish(m, b, G, final_exp, out);
}

// verifyBatch(signature, messages, publicKeys) with every message distinct (BASELINE configs[2] over several GPUs) -- reference index.ts:792-821.
// Every device decodes the keys and hashes the messages of its shard; the first one also decodes the signature and adds millerLoop(-G, S).
EXPORT int nbls_multi_verify_batch(nbls_multi* m, size_t n, const uint8_t* sig96, const uint8_t* msgs, const uint32_t* offsets, const uint8_t* pk48,
                                   const uint8_t* dst, size_t dst_len, int* ok) {
  if (!m || m->ctx.empty() || !ok || !n || !sig96 || !offsets || !pk48 || !dst) return NBLS_EINVAL;
  size_t G = m->ctx.size();
  if (G > n) G = n;   // fewer signatures than devices: the surplus devices stay idle
  SetLease lease(m);
  if (!lease.b) return NBLS_EHIP;
  CallBuffers* b = lease.b;
  std::vector<int> zero(G, 0);
  int r = on_every_device(m, G, [&](size_t g) {
    size_t lo, hi; shard(n, G, g, &lo, &hi);
    void* p = b->part[g];
    return nbls_verify_batch_partial_into(m->ctx[g], hi - lo, g == 0 ? sig96 : nullptr, msgs, offsets + lo, pk48 + lo * 48, dst, dst_len, p, &zero[g], nullptr);
  });
  if (r) return r;
  for (int z : zero) if (z) { *ok = 0; return NBLS_OK; }   // a zero point: pairing() throws, verifyBatch answers false
  uint8_t e[576];
  r = finish(m, b, G, 1, e);
  if (r) return r;
  bool one = e[47] == 1; for (int i = 0; i < 576 && one; i++) if (i != 47 && e[i]) one = false;   // exp.equals(Fp12.ONE)
  *ok = one ? 1 : 0;
  return NBLS_OK;
}

// ---- a pool of contexts on ONE device (round 5) --------------------------------------------------------------------------------------------------------
// A 4096-pairing call fills the chip one wavefront deep, and a lone wavefront issues at little more than half rate: a service that has independent
// batches keeps several calls in flight on contexts of their own.  Round 4 had this only as a Python helper (noble-bls12-381_amd/pipeline.py) and in the JS facade; the pool is
// the same thing behind the C ABI: `depth` contexts, each with its own stream and scratch, tuned for overlapping calls (the two-program Miller loop at every size, the
// final exponentiation's middle as seven launches: what counts with other calls' wavefronts on the SIMDs is the instruction count and fine launches), fed round-robin.
// nbls_pool_pairing_batch_dev enqueues and returns; results are complete after nbls_pool_synchronize (or a synchronisation of the device by the caller).
//
// Hardware queues (round 6).  The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read ONCE when the runtime initialises; with fewer
// queues than contexts in flight, streams share a queue and their kernels serialise: a pool of twelve on four queues runs at 2.50 M instead of 2.98 M pairings/s (tests/c/pool_rate.c).  Round 5 left
// this to the caller (bench.py set the variable, the JS facade and a plain C caller did not).  Now the library sets GPU_MAX_HW_QUEUES = 22 when it is LOADED and the variable is
// unset (a constructor: a program linked against libnbls.so, the N-API addon and a Python process that imports the binding before its first HIP call all get it without doing
// anything; 22: above ~24 user queues the process oversubscribes the device's queue slots and every later stream pays a queue switch per launch, profiles/round4_ab_queues20.txt).
// NBLS_KEEP_HW_QUEUES=1 leaves the environment alone.  A process that initialised HIP before loading the library keeps what it had: nbls_hw_queues() reports the value in the
// environment (0 = unset, i.e. the runtime's 4) and whether the library put it there, and nbls_pool_init warns once on stderr when the depth asked for exceeds it.
static int g_queues_set_by_library = 0;
__attribute__((constructor)) static void nbls_on_load() {
  if (getenv("NBLS_KEEP_HW_QUEUES")) return;
  if (!getenv("GPU_MAX_HW_QUEUES")) { setenv("GPU_MAX_HW_QUEUES", "22", 0); g_queues_set_by_library = 1; }
}
EXPORT int nbls_hw_queues(int* set_by_library) {
  if (set_by_library) *set_by_library = g_queues_set_by_library;
  const char* v = getenv("GPU_MAX_HW_QUEUES");
  return v && *v ? atoi(v) : 0;
}
struct nbls_pool { std::vector<nbls_ctx*> ctx; std::mutex mu; size_t next = 0; int device = 0; };
EXPORT void nbls_pool_destroy(nbls_pool* p) { if (!p) return; for (nbls_ctx* c : p->ctx) nbls_destroy(c); delete p; }
EXPORT int nbls_pool_init(int device_id, int depth, nbls_pool** out) {
  if (!out || depth < 1 || depth > 64) return NBLS_EINVAL;
  {
    const int q = nbls_hw_queues(nullptr), eff = q > 0 ? q : 4;
    static bool warned = false;
    if (depth > eff && !warned) {
      warned = true;
      fprintf(stderr, "nbls: pool of %d contexts on %d hardware queues (GPU_MAX_HW_QUEUES %s): streams will share queues and their kernels serialise; "
                      "set GPU_MAX_HW_QUEUES >= depth (<= 22) before the first HIP call of the process\n",
              depth, eff, q > 0 ? "as set" : "unset");
    }
  }
  nbls_pool* p = new nbls_pool(); p->device = device_id;
  for (int i = 0; i < depth; i++) {
    nbls_ctx* c = nullptr;
    const int r = nbls_init(device_id, &c);
    if (r) { nbls_pool_destroy(p); return r; }
    p->ctx.push_back(c);
    // NBLS_PIPELINE_CHAIN=1: A/B switch (tools/ab_pipeline.sh)
    if (depth > 1) {
      nbls_set_tuning(c, NBLS_TUNE_SPLIT_MILLER_MIN, 0);
      if (nbls::env_long("NBLS_PIPELINE_CHAIN", 0) != 1) nbls_set_tuning(c, NBLS_TUNE_CHAIN_MAX, 0);
      nbls_set_tuning(c, NBLS_TUNE_LS_MAX, 0); nbls_set_tuning(c, NBLS_TUNE_LS2_MAX, 0);      // the lane-split forms too: with calls side by side the plain programs carry more (1024-pair calls: 1.52 -> 1.95 M pairings/s)
      nbls_set_tuning(c, NBLS_TUNE_INV_WIDE_MAX, 256);      // contexts kept busy side by side: instructions count, not one call's latency (nbls_internal.h inv_wide_max)
    }
  }
  *out = p;
  return NBLS_OK;
}
EXPORT int nbls_pool_depth(const nbls_pool* p) { return p ? (int)p->ctx.size() : 0; }
EXPORT nbls_ctx* nbls_pool_context(nbls_pool* p, int i) { return p && i >= 0 && i < (int)p->ctx.size() ? p->ctx[i] : nullptr; }
// pairing(P_i, Q_i) for n device-resident pairs (index.ts:715-722) on the next context's own stream; returns at once.  *slot (optional) = index of the context used: a caller that
// keeps calls in flight gives every slot its own output buffer (the pool does not order two calls that write the same memory).
EXPORT int nbls_pool_pairing_batch_dev(nbls_pool* p, size_t n, const void* d_g1, const void* d_g2, int with_final_exp, void* d_out, int* slot) {
  if (!p || p->ctx.empty()) return NBLS_EINVAL;
  size_t k;
  { std::lock_guard<std::mutex> l(p->mu); k = p->next++ % p->ctx.size(); }
  if (slot) *slot = (int)k;
  return nbls_pairing_batch_dev(p->ctx[k], n, d_g1, d_g2, with_final_exp, d_out, nullptr);
}
EXPORT int nbls_pool_next_slot(nbls_pool* p) { if (!p || p->ctx.empty()) return -1; std::lock_guard<std::mutex> l(p->mu); return (int)(p->next % p->ctx.size()); }
EXPORT int nbls_pool_synchronize(nbls_pool* p) { return p && !p->ctx.empty() ? nbls_device_synchronize(p->ctx[0]) : NBLS_EINVAL; }
