// dev_mem.h — device allocations of the host layer are STREAM-ORDERED (hipMallocAsync / hipFreeAsync
// on the context's stream).  hipFree synchronises the whole device: it waits for every stream, also
// those of other contexts of the process — measured here: destroying a small batch object took
// 28.6 s because another context had a 30 s reduction in flight (tests/perf/free_sync_probe.py).
// One process may drive several contexts at once (in-process multi-GPU plugin, the long config-size
// runs of the test-suite beside the rest of it), so nothing in this library may stall on a stranger's
// kernel.  Every API call waits for its own stream before it returns, so a buffer is idle when it is
// freed.  An allocation is COMPLETE when fphip_dev_alloc returns (it waits for the stream — which is
// idle at every call site: allocations happen at object creation and between launches, never on a
// hot path), so the blocking null-stream hipMemcpy calls of the host layer may touch it at once;
// without the wait the API leaves a use from another stream undefined (the pool could hand out a
// block whose hipFreeAsync is still pending on the stream).
#ifndef FPHIP_DEV_MEM_H
#define FPHIP_DEV_MEM_H

#include <hip/hip_runtime.h>

static inline hipError_t fphip_dev_alloc(void **p, size_t bytes, hipStream_t s)
{
  hipError_t e = hipMallocAsync(p, bytes, s);
  if (e != hipSuccess)
    return e;
  return hipStreamSynchronize(s);
}
static inline void fphip_dev_free(void *p, hipStream_t s)
{
  if (!p)
    return;
  if (hipFreeAsync(p, s) != hipSuccess)
  {
    (void)hipGetLastError();
    (void)hipFree(p);
  }
}

// Pinned, host-coherent buffers (mailboxes, the enumeration context's solution ring and staging
// block) are CACHED for the life of the process: hipHostMalloc / hipHostFree synchronise the whole
// device like hipFree does — measured: closing a small enumeration context took 349 s because the
// config-3 tour of another context was in flight.  (Defined in enum_host.hip.)
__attribute__((visibility("hidden"))) void *fphip_pinned_get(size_t bytes);
__attribute__((visibility("hidden"))) void fphip_pinned_put(void *p);
#endif
