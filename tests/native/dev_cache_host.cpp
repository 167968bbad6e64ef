// dev_cache_host.cpp — fplll_amd/csrc/dev_cache.h (the library's cache of freed device blocks) driven on the CPU:
// hipMalloc / hipFree / hipGetDevice / hipStreamSynchronize are counting stand-ins here, the policy is the library's
// own code.  Prints one line per check ("ok ..." / "FAIL ..."); exit code 1 on any failure.
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <map>

typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };

static std::map<void *, size_t> g_driver;  // what the "driver" holds
static size_t g_driver_bytes = 0, g_driver_limit = (size_t)1 << 40;
static int g_mallocs = 0, g_frees = 0, g_syncs = 0, g_device = 0;

static hipError_t hipMalloc(void **p, size_t bytes)
{
  if (g_driver_bytes + bytes > g_driver_limit)
    return hipErrorOutOfMemory;
  *p = malloc(16);  // (a unique address; never dereferenced)
  g_driver[*p] = bytes;
  g_driver_bytes += bytes;
  ++g_mallocs;
  return hipSuccess;
}
static hipError_t hipFree(void *p)
{
  auto it = g_driver.find(p);
  if (it == g_driver.end())
  {
    printf("FAIL hipFree of a pointer the driver does not hold\n");
    exit(1);
  }
  g_driver_bytes -= it->second;
  g_driver.erase(it);
  free(p);
  ++g_frees;
  return hipSuccess;
}
static hipError_t hipGetDevice(int *d)
{
  *d = g_device;
  return hipSuccess;
}
static hipError_t hipGetLastError() { return hipSuccess; }
typedef int hipDevice_t;
static int g_setdev = 0;
static hipError_t hipSetDevice(int d)
{
  g_device = d;
  ++g_setdev;
  return hipSuccess;
}
// (a "stream" of this harness is its device number + 1, as a pointer; nullptr: no stream)
static hipError_t hipStreamGetDevice(hipStream_t s, hipDevice_t *d)
{
  *d = (int)(size_t)s - 1;
  return hipSuccess;
}
static hipError_t hipStreamSynchronize(hipStream_t)
{
  ++g_syncs;
  return hipSuccess;
}

#define FPHIP_DEV_CACHE_TEST 1
hipError_t fphip_dev_alloc(void **p, size_t bytes, hipStream_t s);
void fphip_dev_free(void *p, hipStream_t s);
#include "../../fplll_amd/csrc/dev_cache.h"

static int g_bad = 0;
#define CHECK(cond, what)                     \
  do                                          \
  {                                           \
    if (cond)                                 \
      printf("ok %s\n", what);                \
    else                                      \
    {                                         \
      printf("FAIL %s\n", what);              \
      g_bad = 1;                              \
    }                                         \
  } while (0)

int main()
{
  const size_t MB = (size_t)1 << 20, GB = (size_t)1 << 30;
  void *a = nullptr, *b = nullptr, *c = nullptr;
  // a freed block comes back for a request it fits, without the driver
  CHECK(fphip_dev_alloc(&a, 100 * MB, nullptr) == hipSuccess && g_mallocs == 1, "first allocation goes to the driver");
  fphip_dev_free(a, (hipStream_t)0x1);
  CHECK(g_frees == 0 && g_syncs == 1, "a free waits for the owner's stream and keeps the block");
  CHECK(fphip_dev_alloc(&b, 90 * MB, nullptr) == hipSuccess && b == a && g_mallocs == 1, "re-used for a request it fits");
  fphip_dev_free(b, nullptr);
  CHECK(g_syncs == 1, "no stream, no wait");
  // ... but not for one it would waste (more than a quarter + 1 MB larger) or that does not fit
  CHECK(fphip_dev_alloc(&b, 10 * MB, nullptr) == hipSuccess && b != a && g_mallocs == 2, "not wasted on a small request");
  CHECK(fphip_dev_alloc(&c, 101 * MB, nullptr) == hipSuccess && c != a && g_mallocs == 3, "not handed to a larger request");
  // best fit among several
  fphip_dev_free(b, nullptr);  // 10 MB
  fphip_dev_free(c, nullptr);  // 101 MB; cache: 100, 10, 101
  void *d = nullptr;
  CHECK(fphip_dev_alloc(&d, 99 * MB, nullptr) == hipSuccess && d == a, "the smallest block that fits");
  // sizes are rounded up to 256 bytes; zero-byte requests get a block
  void *z = nullptr, *z2 = nullptr;
  CHECK(fphip_dev_alloc(&z, 0, nullptr) == hipSuccess && z && g_driver[z] == 256, "zero bytes: a 256-byte block");
  CHECK(fphip_dev_alloc(&z2, 257, nullptr) == hipSuccess && g_driver[z2] == 512, "rounded up to 256");
  // blocks of another device are not handed out
  fphip_dev_free(d, nullptr);  // the 100 MB block, device 0
  g_device = 1;
  void *e = nullptr;
  const int m0 = g_mallocs;
  CHECK(fphip_dev_alloc(&e, 100 * MB, nullptr) == hipSuccess && e != a && g_mallocs == m0 + 1, "per device");
  fphip_dev_free(e, nullptr);
  g_device = 0;
  // the cache is trimmed when it exceeds its cap (FPHIP_DEV_CACHE_GB=1 in the environment of this test), by half,
  // on the device of the block that tipped it
  const int f0 = g_frees;
  void *big[6];
  for (int i = 0; i < 6; ++i)
    CHECK(fphip_dev_alloc(&big[i], 300 * MB + i * 400 * MB / 6, nullptr) == hipSuccess, "a large block");
  for (int i = 0; i < 6; ++i)
    fphip_dev_free(big[i], nullptr);
  size_t cached0 = 0;
  for (auto &kv : g_driver)
    cached0 += kv.second;
  CHECK(g_frees > f0, "over the cap: blocks go back to the driver");
  CHECK(cached0 <= 1 * GB + 700 * MB, "... until about half the cap is left on that device");
  // out of memory: the cache is emptied for the caller's device and the allocation tried again
  g_driver_limit = g_driver_bytes + 50 * MB;
  void *o = nullptr;
  const int f1 = g_frees;
  CHECK(fphip_dev_alloc(&o, 200 * MB, nullptr) == hipSuccess && g_frees > f1, "out of memory: cache emptied, then it fits");
  g_driver_limit = g_driver_bytes;  // nothing left at all
  void *o2 = nullptr;
  CHECK(fphip_dev_alloc(&o2, 200 * MB, nullptr) != hipSuccess && o2 == nullptr, "... and an error when it still does not");
  // a pointer that never came from fphip_dev_alloc goes straight to hipFree
  g_driver_limit = (size_t)1 << 40;
  void *foreign = nullptr;
  hipMalloc(&foreign, 4096);
  const int f2 = g_frees;
  fphip_dev_free(foreign, nullptr);
  CHECK(g_frees == f2 + 1, "a foreign pointer is freed, not cached");
  fphip_dev_free(nullptr, nullptr);
  CHECK(true, "null is ignored");
  // a block is allocated on the device of the stream it is asked for, and the thread's device is put back
  g_device = 0;
  void *on1 = nullptr;
  const int m1 = g_mallocs, sd0 = g_setdev;
  CHECK(fphip_dev_alloc(&on1, 7 * MB, (hipStream_t)(size_t)2) == hipSuccess && g_mallocs == m1 + 1 && g_device == 0 &&
            g_setdev == sd0 + 2,
        "allocated on the stream's device, current device restored");
  fphip_dev_free(on1, (hipStream_t)(size_t)2);
  void *again0 = nullptr, *again1 = nullptr;
  const int m2 = g_mallocs;
  CHECK(fphip_dev_alloc(&again0, 7 * MB, (hipStream_t)(size_t)1) == hipSuccess && again0 != on1 && g_mallocs == m2 + 1,
        "... and not handed to a stream of another device");
  CHECK(fphip_dev_alloc(&again1, 7 * MB, (hipStream_t)(size_t)2) == hipSuccess && again1 == on1, "... but to one of its own");
  return g_bad;
}
