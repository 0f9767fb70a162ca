// Error reporting, version, and the optional HIP-event profiler of libhgn_mp.so.
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <vector>
#include "hgn_host.h"

namespace hgn {

static thread_local char g_err[512] = "";
thread_local int g_prof_tag = 0;
// the DEFAULT for calls that leave `products` at 0: one word, written by hgn_set_matmul_products only (relaxed atomic)
static std::atomic<int> g_products{3};
int matmul_products(int per_call) { return per_call ? per_call : g_products.load(std::memory_order_relaxed); }
// products of the backward / weight-gradient kernels: the forward's for 6, 3 and 1; mode 2 (ONE fp16 product in the forward) differentiates
// with the two-term fp16 products of mode 3 -- fp16 operands need per-row / per-block scales in the backward anyway (gradients leave
// fp16's range), and with the scales in place the second term costs little: the gradients are then exact derivatives of the
// reduced-precision forward up to fp32 rounding, which is what gives the mode a gradient tolerance at all
int bwd_products(int per_call) { const int p = matmul_products(per_call); return p == 2 ? 3 : p; }
bool valid_products(int p) { return p == 0 || p == 1 || p == 2 || p == 3 || p == 6; }

int hgn_fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int hgn_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return HGN_E_LAUNCH;
  }
  return HGN_OK;
}

struct ProfRec { int kid; double units; hipEvent_t a, b; };
static std::mutex g_mu;
static bool g_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;

ProfScope::ProfScope(int kernel_id, double units, hipStream_t s) : kid(kernel_id), stream(s), on(false), slot(-1) {
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on) return;
  ProfRec r;
  r.kid = kernel_id; r.units = units;
  if (!g_pool.empty()) { r.a = g_pool.back().first; r.b = g_pool.back().second; g_pool.pop_back(); }
  else { if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return; }
  hipEventRecord(r.a, stream);
  g_recs.push_back(r);
  slot = (int)g_recs.size() - 1;
  on = true;
}
ProfScope::~ProfScope() {
  if (!on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot >= 0 && slot < (int)g_recs.size()) hipEventRecord(g_recs[slot].b, stream);
}

}  // namespace hgn

using namespace hgn;

extern "C" const char* hgn_last_error(void) { return g_err; }
extern "C" int hgn_version(void) { return 100; }

extern "C" int hgn_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  return HGN_OK;
}
extern "C" int hgn_set_matmul_products(int n) {
  if (n != 1 && n != 2 && n != 3 && n != 6)
    return hgn_fail(HGN_E_INVALID, "hgn_set_matmul_products: 6 / 3 (fp32-accurate: three bf16 / two scaled fp16 terms), 1 (single bf16 product) or 2 (single fp16 product in the forward)");
  g_products.store(n, std::memory_order_relaxed);
  return HGN_OK;
}
extern "C" int hgn_get_matmul_products(void) { return g_products.load(std::memory_order_relaxed); }
extern "C" int hgn_prof_tag(int tag) { g_prof_tag = tag; return HGN_OK; }
extern "C" int hgn_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) g_pool.push_back({r.a, r.b});
  g_recs.clear();
  return HGN_OK;
}
extern "C" int hgn_prof_collect(double* total_ms, int64_t* count, double* units) {
  if (!total_ms || !count || !units) return hgn_fail(HGN_E_INVALID, "hgn_prof_collect: null output");
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < HGN_NUM_KERNEL_IDS; ++i) { total_ms[i] = 0; count[i] = 0; units[i] = 0; }
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) return hgn_fail(HGN_E_LAUNCH, "hgn_prof_collect: event sync failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return hgn_fail(HGN_E_LAUNCH, "hgn_prof_collect: elapsed failed");
    if (r.kid >= 0 && r.kid < HGN_NUM_KERNEL_IDS) { total_ms[r.kid] += ms; count[r.kid] += 1; units[r.kid] += r.units; }
  }
  return HGN_OK;
}
