// dev_cache.h — the implementation of dev_mem.h's fphip_dev_alloc / fphip_dev_free, included by exactly ONE translation
// unit of the library (enum_host.hip) — and by tests/native/dev_cache_host.cpp, which puts counting stand-ins in front
// of hipMalloc / hipFree / hipGetDevice / hipStreamSynchronize (FPHIP_DEV_CACHE_TEST) and drives the policy on the CPU.
#ifndef FPHIP_DEV_CACHE_H
#define FPHIP_DEV_CACHE_H

#include <cstdlib>
#include <mutex>
#include <vector>

// ---- process-wide cache of device blocks (dev_mem.h) ----------------------------------------------------------------
// Freed blocks are kept per device and handed out again to requests they fit (at most a quarter plus 1 MB larger than
// asked for); hipFree — a device-wide synchronisation — only happens when the cache exceeds FPHIP_DEV_CACHE_GB
// (default 96) or when hipMalloc itself runs out of memory.
namespace
{
struct DevBlock
{
  void *p;
  size_t bytes;
  int device;
};
std::mutex g_dev_mutex;
std::vector<DevBlock> g_dev_free;                 // idle blocks
std::vector<DevBlock> g_dev_live;                 // blocks handed out (size and device of a pointer)
size_t g_dev_cached = 0;

// takes blocks of `device` out of the cache until it holds at most keep_bytes; the caller hands them to hipFree
// AFTER it has let go of the mutex (hipFree synchronises the device — with a persistent kernel in flight that is
// served by threads which allocate, freeing under the lock would never return)
std::vector<void *> dev_cache_take_locked(int device, size_t keep_bytes)
{
  std::vector<void *> out;
  for (size_t i = 0; i < g_dev_free.size() && g_dev_cached > keep_bytes;)
  {
    if (g_dev_free[i].device != device)
    {
      ++i;
      continue;
    }
    out.push_back(g_dev_free[i].p);
    g_dev_cached -= g_dev_free[i].bytes;
    g_dev_free[i] = g_dev_free.back();
    g_dev_free.pop_back();
  }
  return out;
}
}  // namespace

// (the block belongs to the device of the stream it is asked for — what the stream-ordered allocator did by
//  construction; the calling thread's current device is put back before the function returns)
namespace
{
struct DevGuard
{
  int prev = -1;
  bool moved = false;
  ~DevGuard()
  {
    if (moved)
      (void)hipSetDevice(prev);
  }
};
}  // namespace

hipError_t fphip_dev_alloc(void **p, size_t bytes, hipStream_t s)
{
  *p = nullptr;
  if (bytes == 0)
    bytes = 256;
  bytes = (bytes + 255) & ~(size_t)255;
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess)
    return e;
  DevGuard guard;
  if (s)
  {
    hipDevice_t sdev = device;
    if (hipStreamGetDevice(s, &sdev) == hipSuccess && (int)sdev != device)
    {
      guard.prev = device;
      if ((e = hipSetDevice((int)sdev)) != hipSuccess)
        return e;
      guard.moved = true;
      device      = (int)sdev;
    }
    else
      (void)hipGetLastError();
  }
  {
    std::lock_guard<std::mutex> lk(g_dev_mutex);
    size_t best = g_dev_free.size();
    for (size_t i = 0; i < g_dev_free.size(); ++i)
    {
      const DevBlock &b = g_dev_free[i];
      if (b.device == device && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + ((size_t)1 << 20) &&
          (best == g_dev_free.size() || b.bytes < g_dev_free[best].bytes))
        best = i;
    }
    if (best != g_dev_free.size())
    {
      const DevBlock b = g_dev_free[best];
      g_dev_free[best] = g_dev_free.back();
      g_dev_free.pop_back();
      g_dev_cached -= b.bytes;
      g_dev_live.push_back(b);
      *p = b.p;
      return hipSuccess;
    }
  }
  void *q = nullptr;
  e       = hipMalloc(&q, bytes);
  if (e != hipSuccess)
  {  // out of memory with blocks in the cache: give them back and try once more
    (void)hipGetLastError();
    std::vector<void *> give;
    {
      std::lock_guard<std::mutex> lk(g_dev_mutex);
      give = dev_cache_take_locked(device, 0);
    }
    for (void *x : give)
      (void)hipFree(x);
    e = hipMalloc(&q, bytes);
    if (e != hipSuccess)
      return e;
  }
  std::lock_guard<std::mutex> lk(g_dev_mutex);
  g_dev_live.push_back(DevBlock{q, bytes, device});
  *p = q;
  return hipSuccess;
}

void fphip_dev_free(void *p, hipStream_t s)
{
  if (!p)
    return;
  // The block may be handed to another owner the moment it is in the cache: whatever its owner's stream still has
  // queued on it (a memset in front of a re-allocation, say) must be over.  The stream is idle at almost every call
  // site — this is a no-op there — and it is the OWNER's stream, never a stranger's.
  if (s)
    (void)hipStreamSynchronize(s);
  static const size_t cap = []
  {
    const char *v = getenv("FPHIP_DEV_CACHE_GB");
    const long gb = v ? atol(v) : 96;
    return (size_t)(gb > 0 ? gb : 1) << 30;
  }();
  std::vector<void *> give;
  bool ours = false;
  {
    std::lock_guard<std::mutex> lk(g_dev_mutex);
    for (size_t i = 0; i < g_dev_live.size(); ++i)
      if (g_dev_live[i].p == p)
      {
        const DevBlock b = g_dev_live[i];
        g_dev_live[i]    = g_dev_live.back();
        g_dev_live.pop_back();
        g_dev_free.push_back(b);
        g_dev_cached += b.bytes;
        if (g_dev_cached > cap)
          give = dev_cache_take_locked(b.device, cap / 2);
        ours = true;
        break;
      }
  }
  if (!ours)
    give.push_back(p);  // (not one of ours: cannot happen; stay correct anyway)
  for (void *x : give)
    (void)hipFree(x);
}

#endif
