// Launch tracing for benchmarks: while a trace is open, every kernel launch of this library is bracketed
// by two events on the stream it is launched on, and recorded with the kernel's symbol (as a profiler
// prints it) and the algorithmic work the launcher declared for it.  Off by default: one global flag is
// read per launch.  Not for use under graph capture.
#include "common.h"
#include "../../include/dvae_hip.h"
#include <cxxabi.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

int g_dvae_trace_on = 0;

namespace {
struct Rec {
  const void* fn;
  hipEvent_t a, b;
  double work;
  int unit;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;          // events created so far, reused by later traces
size_t g_pool_next = 0;
int64_t g_cap = 0;
thread_local double t_work = 0.0;
thread_local int t_unit = DVAE_UNIT_NONE;
thread_local hipEvent_t t_start = nullptr;
std::unordered_map<const void*, std::string> g_names;

hipEvent_t take_event() {
  if (g_pool_next < g_pool.size()) return g_pool[g_pool_next++];
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  g_pool.push_back(e);
  ++g_pool_next;
  return e;
}
}  // namespace

void dvae_trace_work(double work, int unit) {
  t_work = work;
  t_unit = unit;
}

void dvae_trace_pre(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_mu);
  t_start = nullptr;
  if ((int64_t)g_recs.size() >= g_cap) return;
  t_start = take_event();
  if (t_start) (void)hipEventRecord(t_start, s);
}

void dvae_trace_post(const void* fn, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_mu);
  const double w = t_work;
  const int u = t_unit;
  t_work = 0.0;
  t_unit = DVAE_UNIT_NONE;
  if (!t_start) return;
  hipEvent_t e = take_event();
  if (!e) return;
  (void)hipEventRecord(e, s);
  g_recs.push_back(Rec{fn, t_start, e, w, u});
  t_start = nullptr;
}

extern "C" int dvae_trace_begin(int64_t max_launches) {
  DVAE_REQUIRE(max_launches > 0 && max_launches <= (1 << 22), "trace_begin: max_launches must be in 1..4194304");
  std::lock_guard<std::mutex> lock(g_mu);
  g_recs.clear();
  g_recs.reserve((size_t)max_launches);
  g_pool_next = 0;
  g_cap = max_launches;
  g_dvae_trace_on = 1;
  return 0;
}

extern "C" int64_t dvae_trace_end(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_dvae_trace_on = 0;
  return (int64_t)g_recs.size();
}

extern "C" int dvae_trace_read(int64_t i, char* name, int64_t name_cap, float* ms, double* work, int* unit) {
  std::lock_guard<std::mutex> lock(g_mu);
  DVAE_REQUIRE(!g_dvae_trace_on, "trace_read: close the trace with dvae_trace_end first");
  DVAE_REQUIRE(i >= 0 && i < (int64_t)g_recs.size(), "trace_read: record %ld out of range", (long)i);
  DVAE_REQUIRE(name && name_cap > 1 && ms && work && unit, "trace_read: null argument");
  const Rec& r = g_recs[(size_t)i];
  if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(ms, r.a, r.b) != hipSuccess) {
    dvae_set_error("trace_read: the events of record %ld cannot be read: %s", (long)i,
                   hipGetErrorString(hipGetLastError()));
    return -EIO;
  }
  auto it = g_names.find(r.fn);
  if (it == g_names.end()) {
    const char* raw = hipKernelNameRefByPtr(r.fn, nullptr);
    std::string s = raw ? raw : "?";
    int st = 0;
    char* dem = raw ? abi::__cxa_demangle(raw, nullptr, nullptr, &st) : nullptr;
    if (dem && st == 0) s = dem;
    free(dem);
    it = g_names.emplace(r.fn, s).first;
  }
  snprintf(name, (size_t)name_cap, "%s", it->second.c_str());
  *work = r.work;
  *unit = r.unit;
  return 0;
}
