// cst_core.hip — error reporting, version/arch query, per-kernel-class profiling table.
#include "cst_common.h"
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void cst_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int cst_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    cst_set_error("%s: %s", what, hipGetErrorString(e));
    return CST_ERR_LAUNCH;
  }
  return CST_OK;
}

// ---- profiling table ---------------------------------------------------------------------
struct ProfRec {
  hipEvent_t start, stop;
  double flops, bytes;
  char tag[96];
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof[CST_K_NUM];

CstProfScope::CstProfScope(int cls_, hipStream_t s_, double flops, double bytes) : cls(cls_), s(s_), slot(-1) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  r.flops = flops;
  r.bytes = bytes;
  r.tag[0] = 0;
  if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  (void)hipEventRecord(r.start, s);
  g_prof[cls].push_back(r);
  slot = (int)g_prof[cls].size() - 1;
}
bool cst_prof_is_on() { return g_prof_on; }

void CstProfScope::tag(const char* fmt, ...) {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_prof[cls][slot].tag, sizeof(g_prof[cls][slot].tag), fmt, ap);
  va_end(ap);
}

CstProfScope::~CstProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof[cls][slot].stop, s);
}

extern "C" {

const char* cst_last_error(void) { return g_err; }
int cst_version(void) { return CST_ABI_VERSION; }

int cst_device_arch_ok(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

void cst_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (on) {
    for (int c = 0; c < CST_K_NUM; ++c) {
      for (auto& r : g_prof[c]) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
      }
      g_prof[c].clear();
    }
  }
  g_prof_on = on != 0;
}

int64_t cst_prof_query(int cls, double* total_ms, double* flops, double* bytes) {
  if (cls < 0 || cls >= CST_K_NUM) return -1;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double ms = 0, fl = 0, by = 0;
  for (auto& r : g_prof[cls]) {
    (void)hipEventSynchronize(r.stop);
    float t = 0;
    if (hipEventElapsedTime(&t, r.start, r.stop) == hipSuccess) ms += t;
    fl += r.flops;
    by += r.bytes;
  }
  if (total_ms) *total_ms = ms;
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  return (int64_t)g_prof[cls].size();
}

/* one text line per recorded launch of the class, in launch order: "<ms> <flops> <bytes> <tag>\n"; returns the bytes needed
 * (including the terminating 0) — call with cap = 0 to size the buffer */
int64_t cst_prof_dump(int cls, char* buf, int64_t cap) {
  if (cls < 0 || cls >= CST_K_NUM) return -1;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int64_t need = 1, at = 0;
  for (auto& r : g_prof[cls]) {
    (void)hipEventSynchronize(r.stop);
    float t = 0;
    (void)hipEventElapsedTime(&t, r.start, r.stop);
    char line[192];
    const int n = snprintf(line, sizeof(line), "%.6f %.6g %.6g %s\n", t, r.flops, r.bytes, r.tag);
    need += n;
    if (buf && at + n < cap) { memcpy(buf + at, line, (size_t)n); at += n; }
  }
  if (buf && cap > 0) buf[at < cap ? at : cap - 1] = 0;
  return need;
}

}  // extern "C"
