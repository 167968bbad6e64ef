// trace.h — roctx ranges around the C ABI's entry points and launches (the reference has
// no tracing; rocprofv3 --marker-trace shows these next to the kernel timeline).  The marker library is
// looked up at run time — librocprofiler-sdk-roctx.so (what rocprofv3 preloads), then libroctx64.so —
// and only when it is already in the process or FPHIP_ROCTX=1 asks for it: no link-time dependency,
// one pointer test per range otherwise.
#ifndef FPHIP_TRACE_H
#define FPHIP_TRACE_H

#include <dlfcn.h>
#include <stdlib.h>

namespace fphip_trace
{
typedef int (*push_fn)(const char *);
typedef int (*pop_fn)(void);
struct Api
{
  push_fn push = nullptr;
  pop_fn pop   = nullptr;
  Api()
  {
    const char *libs[2] = {"librocprofiler-sdk-roctx.so", "libroctx64.so"};
    const char *want    = getenv("FPHIP_ROCTX");
    for (int i = 0; i < 2 && !push; ++i)
    {
      void *h = dlopen(libs[i], RTLD_NOW | RTLD_NOLOAD);
      if (!h && want && atoi(want) != 0)
        h = dlopen(libs[i], RTLD_NOW | RTLD_GLOBAL);
      if (!h)
        continue;
      push = (push_fn)dlsym(h, "roctxRangePushA");
      pop  = (pop_fn)dlsym(h, "roctxRangePop");
      if (!push || !pop)
        push = nullptr;
    }
  }
};
inline Api &api()
{
  static Api a;
  return a;
}
struct Range
{
  bool on;
  explicit Range(const char *name) : on(api().push != nullptr)
  {
    if (on)
      api().push(name);
  }
  ~Range()
  {
    if (on)
      api().pop();
  }
  Range(const Range &)            = delete;
  Range &operator=(const Range &) = delete;
};
}  // namespace fphip_trace
#define FPHIP_TRACE_CAT2(a, b) a##b
#define FPHIP_TRACE_CAT(a, b) FPHIP_TRACE_CAT2(a, b)
#define FPHIP_RANGE(name) fphip_trace::Range FPHIP_TRACE_CAT(fphip_range_, __LINE__)(name)
#endif
