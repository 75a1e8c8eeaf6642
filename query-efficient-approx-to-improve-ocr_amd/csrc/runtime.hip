// Library-wide state: last-error string, ABI version, per-kernel-class event timing.
#include "common.h"
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>

namespace {
thread_local char g_err[512] = "";

struct ProfClass {
  bool on = false;
  std::vector<hipEvent_t> start, stop;  // event pool, reused across resets
  size_t used = 0;
  double flops = 0.0, bytes = 0.0, flops_split = 0.0, flops_f16 = 0.0;  // flops_split / flops_f16: the parts that ran as 3-way bf16 / 2-way fp16 split MFMAs
  std::vector<double> launch_flops, launch_bytes;
  std::vector<int> launch_tag;              // which kernel took the launch (class-specific code, 0 = unspecified)
  bool open = false;
};
ProfClass g_prof[QEA_PROF_NCLASS];
std::mutex g_prof_mu;
}  // namespace

void qea_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
std::atomic<int>& mfma_mode() {
  static std::atomic<int> mode([] {
    const char* e = getenv("QEA_MFMA");
    return (e && strcmp(e, "f32") == 0) ? QEA_MFMA_F32 : QEA_MFMA_SPLIT_BF16;
  }());
  return mode;
}
}  // namespace

bool qea_split_bf16_enabled() { return mfma_mode().load(std::memory_order_relaxed) == QEA_MFMA_SPLIT_BF16; }

extern "C" int qea_set_mfma_mode(int mode) {
  if (mode == -1) return mfma_mode().load();
  QEA_REQUIRE(mode == QEA_MFMA_SPLIT_BF16 || mode == QEA_MFMA_F32, "qea_set_mfma_mode: bad mode %d", mode);
  return mfma_mode().exchange(mode);
}

extern "C" const char* qea_last_error(void) { return g_err; }
extern "C" int qea_version(void) { return 9; }

void qea_prof_begin(int klass, hipStream_t s) {
  ProfClass& pc = g_prof[klass];
  if (!pc.on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (pc.used == pc.start.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    pc.start.push_back(a);
    pc.stop.push_back(b);
  }
  hipEventRecord(pc.start[pc.used], s);
  pc.open = true;
}

void qea_prof_end(int klass, hipStream_t s, double flops, double bytes, int split, int tag) {
  ProfClass& pc = g_prof[klass];
  if (!pc.on || !pc.open) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEventRecord(pc.stop[pc.used], s);
  pc.used++;
  pc.launch_flops.push_back(flops);
  pc.launch_bytes.push_back(bytes);
  pc.launch_tag.push_back(tag);
  pc.flops += flops;
  pc.bytes += bytes;
  if (split == 1) pc.flops_split += flops;
  if (split == 2) pc.flops_f16 += flops;
  pc.open = false;
}

// an error return between begin and end: the class is closed again, the (re-recordable) start event slot stays unused
void qea_prof_abort(int klass) {
  ProfClass& pc = g_prof[klass];
  if (!pc.on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  pc.open = false;
}

extern "C" int qea_prof_enable(int klass, int on) {
  QEA_REQUIRE(klass >= 0 && klass < QEA_PROF_NCLASS, "qea_prof_enable: bad class %d", klass);
  g_prof[klass].on = on != 0;
  return QEA_OK;
}

extern "C" int qea_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& pc : g_prof) {
    pc.used = 0;
    pc.launch_flops.clear();
    pc.launch_bytes.clear();
    pc.launch_tag.clear();
    pc.flops = pc.bytes = pc.flops_split = pc.flops_f16 = 0.0;
    pc.open = false;
  }
  return QEA_OK;
}

extern "C" int qea_prof_read(int klass, double* ms, double* flops, double* bytes, int64_t* launches) {
  QEA_REQUIRE(klass >= 0 && klass < QEA_PROF_NCLASS, "qea_prof_read: bad class %d", klass);
  ProfClass& pc = g_prof[klass];
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double total = 0.0;
  for (size_t i = 0; i < pc.used; ++i) {
    if (hipEventSynchronize(pc.stop[i]) != hipSuccess) {
      qea_set_error("qea_prof_read: event sync failed");
      return QEA_ERR_LAUNCH;
    }
    float t = 0.f;
    hipEventElapsedTime(&t, pc.start[i], pc.stop[i]);
    total += t;
  }
  if (ms) *ms = total;
  if (flops) *flops = pc.flops;
  if (bytes) *bytes = pc.bytes;
  if (launches) *launches = (int64_t)pc.used;
  return QEA_OK;
}

extern "C" int qea_prof_read_tagged(int klass, int32_t tag, double* ms, double* flops, double* bytes, int64_t* launches) {
  QEA_REQUIRE(klass >= 0 && klass < QEA_PROF_NCLASS, "qea_prof_read_tagged: bad class %d", klass);
  ProfClass& pc = g_prof[klass];
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double total = 0.0, fl = 0.0, by = 0.0;
  int64_t n = 0;
  for (size_t i = 0; i < pc.used; ++i) {
    if (pc.launch_tag[i] != tag) continue;
    if (hipEventSynchronize(pc.stop[i]) != hipSuccess) {
      qea_set_error("qea_prof_read_tagged: event sync failed");
      return QEA_ERR_LAUNCH;
    }
    float t = 0.f;
    hipEventElapsedTime(&t, pc.start[i], pc.stop[i]);
    total += t;
    fl += pc.launch_flops[i];
    by += pc.launch_bytes[i];
    ++n;
  }
  if (ms) *ms = total;
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  if (launches) *launches = n;
  return QEA_OK;
}

extern "C" int qea_prof_read_split_f16(int klass, double* flops) {
  QEA_REQUIRE(klass >= 0 && klass < QEA_PROF_NCLASS && flops, "qea_prof_read_split_f16: bad arguments");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  *flops = g_prof[klass].flops_f16;
  return QEA_OK;
}

extern "C" int qea_prof_read_split_bf16(int klass, double* flops) {
  QEA_REQUIRE(klass >= 0 && klass < QEA_PROF_NCLASS && flops, "qea_prof_read_split_bf16: bad arguments");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  *flops = g_prof[klass].flops_split;
  return QEA_OK;
}

extern "C" int qea_prof_read_launches(int klass, double* ms, double* flops, int64_t capacity, int64_t* count) {
  QEA_REQUIRE(klass >= 0 && klass < QEA_PROF_NCLASS && count, "qea_prof_read_launches: bad arguments");
  ProfClass& pc = g_prof[klass];
  std::lock_guard<std::mutex> lk(g_prof_mu);
  *count = (int64_t)pc.used;
  for (size_t i = 0; i < pc.used && (int64_t)i < capacity; ++i) {
    if (hipEventSynchronize(pc.stop[i]) != hipSuccess) {
      qea_set_error("qea_prof_read_launches: event sync failed");
      return QEA_ERR_LAUNCH;
    }
    float t = 0.f;
    hipEventElapsedTime(&t, pc.start[i], pc.stop[i]);
    if (ms) ms[i] = t;
    if (flops) flops[i] = pc.launch_flops[i];
  }
  return QEA_OK;
}
