// ROCTX ranges around the phases of a call (digit + sort passes, accumulation, tail, host fold) so that a
// `rocprofv3 --marker-trace` timeline shows them by name.  Bound at run time like RCCL: librocprofiler-sdk-roctx (what
// rocprofv3 reads) or the older libroctx64; nothing is loaded, and a range costs one branch, unless LEMSM_ROCTX=1.
#pragma once
#include <dlfcn.h>
#include <stdlib.h>

namespace lemsm {

struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  static Roctx& get() {
    static Roctx r = load();
    return r;
  }
  static Roctx load() {
    Roctx r;
    const char* on = getenv("LEMSM_ROCTX");
    if (!on || on[0] != '1') return r;
    for (const char* nm : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void* h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (!h) continue;
      r.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      r.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (r.push && r.pop) break;
      r.push = nullptr; r.pop = nullptr;
    }
    return r;
  }
};

// scoped range: RoctxRange r("lemsm: accumulate");
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char* name) : on(Roctx::get().push != nullptr) { if (on) Roctx::get().push(name); }
  ~RoctxRange() { if (on) Roctx::get().pop(); }
  RoctxRange(const RoctxRange&) = delete;
  RoctxRange& operator=(const RoctxRange&) = delete;
};

}  // namespace lemsm
