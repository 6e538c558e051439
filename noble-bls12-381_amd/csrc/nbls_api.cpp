// nbls_api.cpp -- C ABI (include/nbls.h) and device runtime of the pairing engine: program upload, scratch
// management and the launch pipelines.  No CPU arithmetic path exists here: if HIP or the GPU is unavailable every
// entry point fails with NBLS_ENOGPU / NBLS_EHIP.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>
#include "nbls.h"
#include "programs.h"
#include "consts_gen.h"
#include "fp_inv.h"

extern "C" int nbls_vm_launch(const nbls::KernelArgs* ka, unsigned lds_bytes, void* stream);
extern "C" int nbls_fp_inv_launch(unsigned n, const void* in, void* out, const void* table, void* stream);

using namespace nbls;

#define EXPORT extern "C" __attribute__((visibility("default")))

struct DevProgram {
  Step* steps = nullptr; u32* descs = nullptr; u32* consts = nullptr;
  const Program* p = nullptr;
};

struct nbls_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  DevProgram prog[P_COUNT];
  // scratch (device)
  uint8_t *F = nullptr, *F2 = nullptr, *N = nullptr, *NI = nullptr, *io_g1 = nullptr, *io_g2 = nullptr, *io_f12 = nullptr, *one12 = nullptr, *inv_table = nullptr;
  uint8_t* T[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // t1..t7 of the final exponentiation, raw Fp12
  size_t cap_F = 0, cap_io = 0;
  int last_hip = 0;
  // optional per-kernel timing (HIP events on the launch stream); slot P_COUNT = inversion kernel
  bool timing = false;
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> tev;
};

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->last_hip = (int)e_; return NBLS_EHIP; } } while (0)

static int upload(nbls_ctx* ctx, ProgId id) {
  DevProgram& d = ctx->prog[id];
  if (d.p) return NBLS_OK;
  const Program& p = get_program(id);
  HIPCHK(hipMalloc(&d.steps, p.steps.size() * sizeof(Step)));
  HIPCHK(hipMalloc(&d.descs, p.descs.size() * 4 + 64));
  HIPCHK(hipMalloc(&d.consts, p.consts.size() * 4));
  HIPCHK(hipMemcpy(d.steps, p.steps.data(), p.steps.size() * sizeof(Step), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.descs, p.descs.data(), p.descs.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.consts, p.consts.data(), p.consts.size() * 4, hipMemcpyHostToDevice));
  d.p = &p;
  return NBLS_OK;
}

static int run(nbls_ctx* ctx, ProgId id, size_t n, std::initializer_list<std::pair<int, std::pair<const void*, size_t>>> bufs, hipStream_t s) {
  int r = upload(ctx, id); if (r) return r;
  const DevProgram& d = ctx->prog[id];
  KernelArgs ka; memset(&ka, 0, sizeof ka);
  ka.steps = d.steps; ka.descs = d.descs; ka.consts = d.consts;
  ka.nsteps = (u32)d.p->steps.size(); ka.nconst = d.p->nconst; ka.W = d.p->W; ka.G = d.p->G; ka.slots = d.p->slots; ka.n_items = (u32)n;
  for (auto& b : bufs) { ka.bufs[b.first].ptr = (uint8_t*)b.second.first; ka.bufs[b.first].stride = b.second.second; }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, s); }
  int e = nbls_vm_launch(&ka, d.p->lds_bytes(), s);
  if (ctx->timing) { hipEventRecord(e1, s); ctx->tev.push_back({(int)id, {e0, e1}}); }
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  return NBLS_OK;
}
static int run_inv(nbls_ctx* ctx, size_t n, hipStream_t s) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, s); }
  int e = nbls_fp_inv_launch((unsigned)n, ctx->N, ctx->NI, ctx->inv_table, s);
  if (ctx->timing) { hipEventRecord(e1, s); ctx->tev.push_back({(int)P_COUNT, {e0, e1}}); }
  if (e) { ctx->last_hip = e; return NBLS_EHIP; }
  return NBLS_OK;
}

static int ensure_scratch(nbls_ctx* ctx, size_t n) {
  if (n <= ctx->cap_F) return NBLS_OK;
  size_t cap = n + n / 8 + 64;
  if (ctx->F) { hipFree(ctx->F); hipFree(ctx->F2); hipFree(ctx->N); hipFree(ctx->NI); for (auto& t : ctx->T) hipFree(t); }
  ctx->cap_F = 0;
  HIPCHK(hipMalloc(&ctx->F, (cap + 2) * 576));
  HIPCHK(hipMalloc(&ctx->F2, (cap / 2 + 2) * 576));
  HIPCHK(hipMalloc(&ctx->N, cap * 48));
  HIPCHK(hipMalloc(&ctx->NI, cap * 48));
  for (auto& t : ctx->T) HIPCHK(hipMalloc(&t, cap * 576));
  ctx->cap_F = cap;
  return NBLS_OK;
}
static int ensure_io(nbls_ctx* ctx, size_t n) {
  if (n <= ctx->cap_io) return NBLS_OK;
  size_t cap = n + 64;
  if (ctx->io_g1) { hipFree(ctx->io_g1); hipFree(ctx->io_g2); hipFree(ctx->io_f12); }
  ctx->cap_io = 0;
  HIPCHK(hipMalloc(&ctx->io_g1, cap * 96));
  HIPCHK(hipMalloc(&ctx->io_g2, cap * 192));
  HIPCHK(hipMalloc(&ctx->io_f12, cap * 576));
  ctx->cap_io = cap;
  return NBLS_OK;
}

typedef std::pair<int, std::pair<const void*, size_t>> BufArg;
static inline BufArg B(int idx, const void* p, size_t stride) { return {idx, {p, stride}}; }

// F (n raw Fp12) -> one element in F[0] (or F2[0]); returns pointer to the buffer holding the product
static int reduce_product(nbls_ctx* ctx, size_t n, uint8_t** result, hipStream_t s) {
  uint8_t *src = ctx->F, *dst = ctx->F2;
  size_t m = n;
  while (m > 1) {
    if (m & 1) { HIPCHK(hipMemcpyAsync(src + m * 576, ctx->one12, 576, hipMemcpyDeviceToDevice, s)); m++; }
    int r = run(ctx, P_MUL2, m / 2, {B(3, src, 1152), B(5, dst, 576)}, s); if (r) return r;
    std::swap(src, dst); m /= 2;
  }
  *result = src;
  return NBLS_OK;
}
// n raw Fp12 in `f_raw` (norms already in ctx->N) -> finalExponentiate -> wire bytes at d_out (math.ts:856-874)
static int final_exp_pipeline(nbls_ctx* ctx, size_t n, uint8_t* f_raw, void* d_out, hipStream_t s) {
  int r;
  uint8_t** T = ctx->T;
  if ((r = run_inv(ctx, n, s))) return r;
  if ((r = run(ctx, P_FE_EASY, n, {B(3, f_raw, 576), B(4, ctx->NI, 48), B(5, T[0], 576)}, s))) return r;
  if ((r = run(ctx, P_EXPX, n, {B(3, T[0], 576), B(5, T[1], 576)}, s))) return r;                       // t2
  if ((r = run(ctx, P_FE_MID1, n, {B(3, T[0], 576), B(5, T[1], 576), B(6, T[2], 576)}, s))) return r;   // t3
  if ((r = run(ctx, P_EXPX, n, {B(3, T[2], 576), B(5, T[3], 576)}, s))) return r;                       // t4
  if ((r = run(ctx, P_EXPX, n, {B(3, T[3], 576), B(5, T[4], 576)}, s))) return r;                       // t5
  if ((r = run(ctx, P_EXPX, n, {B(3, T[4], 576), B(5, T[6], 576)}, s))) return r;                       // t6' (parked in T7's buffer)
  if ((r = run(ctx, P_FE_MID2, n, {B(3, T[6], 576), B(5, T[1], 576), B(6, T[5], 576)}, s))) return r;   // t6
  if ((r = run(ctx, P_EXPX, n, {B(3, T[5], 576), B(5, T[6], 576)}, s))) return r;                       // t7
  return run(ctx, P_FE_FINAL, n, {B(0, T[0], 576), B(1, T[1], 576), B(2, T[2], 576), B(3, T[3], 576), B(4, T[4], 576), B(5, T[5], 576), B(6, T[6], 576), B(7, d_out, 576)}, s);
}
// one raw Fp12 -> final exponentiation (or plain encoding) -> wire bytes on device
static int finish_single(nbls_ctx* ctx, uint8_t* f_raw, int final_exp, void* d_out, hipStream_t s) {
  int r;
  if (!final_exp) return run(ctx, P_RAW_TO_BYTES, 1, {B(3, f_raw, 576), B(2, d_out, 576)}, s);
  if ((r = run(ctx, P_NORM_RAW, 1, {B(3, f_raw, 576), B(4, ctx->N, 48)}, s))) return r;
  return final_exp_pipeline(ctx, 1, f_raw, d_out, s);
}

EXPORT int nbls_init(int device_id, nbls_ctx** out) {
  if (!out) return NBLS_EINVAL;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return NBLS_ENOGPU;
  if (device_id < 0 || device_id >= count) return NBLS_EINVAL;
  nbls_ctx* ctx = new nbls_ctx();
  ctx->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess) { delete ctx; return NBLS_ENOGPU; }
  if (hipStreamCreate(&ctx->stream) != hipSuccess) { delete ctx; return NBLS_EHIP; }
  // Montgomery one as a raw Fp12 (pads odd-sized product reductions)
  u32 one[144]; memset(one, 0, sizeof one); memcpy(one, NBLS_R1, 48);
  if (hipMalloc(&ctx->one12, 576) != hipSuccess || hipMemcpy(ctx->one12, one, 576, hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return NBLS_EHIP; }
  {
    std::vector<u32> tab(382 * 12); make_inv_table(tab.data());
    if (hipMalloc(&ctx->inv_table, tab.size() * 4) != hipSuccess || hipMemcpy(ctx->inv_table, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return NBLS_EHIP; }
  }
  for (int i = 0; i < P_COUNT; i++) { int r = upload(ctx, (ProgId)i); if (r) { int e = ctx->last_hip; (void)e; nbls_destroy(ctx); return r; } }
  *out = ctx;
  return NBLS_OK;
}

EXPORT void nbls_destroy(nbls_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  for (auto& d : ctx->prog) { if (d.steps) hipFree(d.steps); if (d.descs) hipFree(d.descs); if (d.consts) hipFree(d.consts); }
  for (uint8_t* p : {ctx->F, ctx->F2, ctx->N, ctx->NI, ctx->io_g1, ctx->io_g2, ctx->io_f12, ctx->one12, ctx->inv_table}) if (p) hipFree(p);
  for (uint8_t* p : ctx->T) if (p) hipFree(p);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

EXPORT const char* nbls_strerror(int code) {
  switch (code) {
    case NBLS_OK: return "ok";
    case NBLS_EINVAL: return "invalid argument";
    case NBLS_EHIP: return "HIP runtime error";
    case NBLS_ENOSUP: return "not supported in this build";
    case NBLS_ENOGPU: return "no usable GPU";
    default: return "unknown error";
  }
}
EXPORT int nbls_last_hip_error(nbls_ctx* ctx) { return ctx ? ctx->last_hip : 0; }
EXPORT int nbls_device_synchronize(nbls_ctx* ctx) { if (!ctx) return NBLS_EINVAL; HIPCHK(hipSetDevice(ctx->device)); HIPCHK(hipDeviceSynchronize()); return NBLS_OK; }

EXPORT int nbls_pairing_batch_dev(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_g2, int with_final_exp, void* d_out, void* stream) {
  if (!ctx || (n && (!d_g1 || !d_g2 || !d_out))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  int r;
  if (!with_final_exp) return run(ctx, P_MILLER_BYTES, n, {B(0, d_g1, 96), B(1, d_g2, 192), B(2, d_out, 576)}, s);
  if ((r = ensure_scratch(ctx, n))) return r;
  if ((r = run(ctx, P_MILLER_FE, n, {B(0, d_g1, 96), B(1, d_g2, 192), B(3, ctx->F, 576), B(4, ctx->N, 48)}, s))) return r;
  return final_exp_pipeline(ctx, n, ctx->F, d_out, s);
}

EXPORT int nbls_pairing_batch(nbls_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, int with_final_exp, int validate, uint8_t* out, int8_t* status) {
  if (!ctx || (n && (!g1 || !g2 || !out))) return NBLS_EINVAL;
  if (validate) return NBLS_ENOSUP;
  if (n == 0) return NBLS_OK;
  int r;
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    if ((r = ensure_io(ctx, n))) return r;
    HIPCHK(hipMemcpyAsync(ctx->io_g1, g1, n * 96, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->io_g2, g2, n * 192, hipMemcpyHostToDevice, ctx->stream));
  }
  if ((r = nbls_pairing_batch_dev(ctx, n, ctx->io_g1, ctx->io_g2, with_final_exp, ctx->io_f12, ctx->stream))) return r;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIPCHK(hipMemcpyAsync(out, ctx->io_f12, n * 576, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (status) memset(status, 0, n);
  return NBLS_OK;
}

EXPORT int nbls_miller_product_dev(nbls_ctx* ctx, size_t n, const void* d_g1, const void* d_g2, int final_exp, void* d_out, void* stream) {
  if (!ctx || !d_out || (n && (!d_g1 || !d_g2))) return NBLS_EINVAL;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  int r;
  if ((r = ensure_scratch(ctx, n ? n : 1))) return r;
  uint8_t* res = ctx->F;
  if (n == 0) { HIPCHK(hipMemcpyAsync(ctx->F, ctx->one12, 576, hipMemcpyDeviceToDevice, s)); }
  else {
    if ((r = run(ctx, P_MILLER_RAW, n, {B(0, d_g1, 96), B(1, d_g2, 192), B(3, ctx->F, 576)}, s))) return r;
    if ((r = reduce_product(ctx, n, &res, s))) return r;
  }
  return finish_single(ctx, res, final_exp, d_out, s);
}

EXPORT int nbls_miller_product(nbls_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, int final_exp, int validate, uint8_t* out, int8_t* status) {
  if (!ctx || !out || (n && (!g1 || !g2))) return NBLS_EINVAL;
  if (validate) return NBLS_ENOSUP;
  int r;
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    if ((r = ensure_io(ctx, n ? n : 1))) return r;
    if (n) {
      HIPCHK(hipMemcpyAsync(ctx->io_g1, g1, n * 96, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(ctx->io_g2, g2, n * 192, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  if ((r = nbls_miller_product_dev(ctx, n, ctx->io_g1, ctx->io_g2, final_exp, ctx->io_f12, ctx->stream))) return r;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIPCHK(hipMemcpyAsync(out, ctx->io_f12, 576, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (status) memset(status, 0, n);
  return NBLS_OK;
}

EXPORT int nbls_final_exp_batch_dev(nbls_ctx* ctx, size_t n, const void* d_in, void* d_out, void* stream) {
  if (!ctx || (n && (!d_in || !d_out))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  int r;
  if ((r = ensure_scratch(ctx, n))) return r;
  if ((r = run(ctx, P_NORM_BYTES, n, {B(2, d_in, 576), B(3, ctx->F, 576), B(4, ctx->N, 48)}, s))) return r;
  return final_exp_pipeline(ctx, n, ctx->F, d_out, s);
}

EXPORT int nbls_final_exp_batch(nbls_ctx* ctx, size_t n, const uint8_t* in, uint8_t* out) {
  if (!ctx || (n && (!in || !out))) return NBLS_EINVAL;
  if (n == 0) return NBLS_OK;
  int r;
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    if ((r = ensure_io(ctx, 2 * n))) return r;
    HIPCHK(hipMemcpyAsync(ctx->io_f12, in, n * 576, hipMemcpyHostToDevice, ctx->stream));
  }
  uint8_t* d_out = ctx->io_f12 + n * 576;
  if ((r = nbls_final_exp_batch_dev(ctx, n, ctx->io_f12, d_out, ctx->stream))) return r;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIPCHK(hipMemcpyAsync(out, d_out, n * 576, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return NBLS_OK;
}

// n Fp12 wire elements on the device -> their product, optionally final-exponentiated (multi-GPU: partials of all ranks)
EXPORT int nbls_fp12_product_final_dev(nbls_ctx* ctx, size_t n, const void* d_in, int final_exp, void* d_out, void* stream) {
  if (!ctx || !d_out || (n && !d_in)) return NBLS_EINVAL;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  int r;
  if ((r = ensure_scratch(ctx, n ? n : 1))) return r;
  uint8_t* res = ctx->F;
  if (n == 0) { HIPCHK(hipMemcpyAsync(ctx->F, ctx->one12, 576, hipMemcpyDeviceToDevice, s)); }
  else {
    // wire bytes -> raw Montgomery (P_NORM_BYTES also writes N, which is ignored here)
    if ((r = run(ctx, P_NORM_BYTES, n, {B(2, d_in, 576), B(3, ctx->F, 576), B(4, ctx->N, 48)}, s))) return r;
    if ((r = reduce_product(ctx, n, &res, s))) return r;
  }
  return finish_single(ctx, res, final_exp, d_out, s);
}

EXPORT int nbls_program_stats(nbls_ctx* ctx, int prog, uint32_t* o) {
  (void)ctx;
  if (prog < 0 || prog >= P_COUNT || !o) return NBLS_EINVAL;
  const Program& p = get_program((ProgId)prog);
  o[0] = (uint32_t)p.steps.size(); o[1] = p.n_dot_steps; o[2] = p.n_lin_steps; o[3] = p.n_dot_ops; o[4] = p.n_products; o[5] = p.n_lin_ops; o[6] = p.slots; o[7] = p.lds_bytes();
  return NBLS_OK;
}

// Per-kernel timing for the benchmark's roofline leg: enable, run, synchronise, then read accumulated milliseconds and
// launch counts per program (index P_COUNT = the inversion kernel).  ms/counts must hold P_COUNT+1 entries.
EXPORT int nbls_timing_enable(nbls_ctx* ctx, int on) {
  if (!ctx) return NBLS_EINVAL;
  std::lock_guard<std::mutex> g(ctx->mu);
  for (auto& t : ctx->tev) { hipEventDestroy(t.second.first); hipEventDestroy(t.second.second); }
  ctx->tev.clear(); ctx->timing = on != 0; return NBLS_OK;
}
EXPORT int nbls_timing_read(nbls_ctx* ctx, float* ms, uint32_t* counts) {
  if (!ctx || !ms || !counts) return NBLS_EINVAL;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  for (int i = 0; i <= P_COUNT; i++) { ms[i] = 0; counts[i] = 0; }
  for (auto& t : ctx->tev) {
    HIPCHK(hipEventSynchronize(t.second.second));
    float m = 0; HIPCHK(hipEventElapsedTime(&m, t.second.first, t.second.second));
    ms[t.first] += m; counts[t.first]++;
    hipEventDestroy(t.second.first); hipEventDestroy(t.second.second);
  }
  ctx->tev.clear();
  return NBLS_OK;
}
