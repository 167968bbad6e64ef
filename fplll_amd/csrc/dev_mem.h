// dev_mem.h — device allocations of the host layer: plain hipMalloc behind a process-wide CACHE of freed blocks.
//
// Two things shaped this.  (1) hipFree synchronises the whole device: it waits for every stream, also those of other
// contexts of the process — measured: destroying a small batch object took 28.6 s because another context had a 30 s
// reduction in flight (tests/perf/free_sync_probe.py).  One process may drive several contexts at once (in-process
// multi-GPU plugin, minutes-long runs beside short ones), so nothing in this library may stall on a stranger's
// kernel: a freed block goes to the cache, not to the driver.  (2) Rounds 3-6 used the runtime's stream-ordered
// allocator for that (hipMallocAsync / hipFreeAsync).  On ROCm 7.2 / gfx950 memory it has JUST handed out is not
// stable under its first kernel: tests/perf/micro/fresh_alloc_probe.hip — 8192 waves store a pattern into a private
// region each and read it back — counts 2.8·10^6 and 1.4·10^7 wrong words in 200 free / allocate / launch rounds, 2.3·10^6 with
// the pool's release threshold raised, and 0 with hipMalloc / hipFree.  That was the defect behind the run-to-run
// differences of the sub-solution walk (its call re-allocated the global-stack scratch microseconds in front of the
// kernel) and, in all likelihood, behind a batch object's upload that vanished once.
//
// Contract: a buffer is idle when it is freed (every API call waits for its own stream before it returns); a block
// from the cache holds whatever its last owner left in it (callers memset what they need zeroed — as they had to
// under the pool as well).  fphip_dev_free waits for the stream it is given — the owner's — before the block enters the
// cache; fphip_dev_alloc ignores its stream argument.
#ifndef FPHIP_DEV_MEM_H
#define FPHIP_DEV_MEM_H

#include <hip/hip_runtime.h>

// (defined in enum_host.hip: one cache for the library)
__attribute__((visibility("hidden"))) hipError_t fphip_dev_alloc(void **p, size_t bytes, hipStream_t s);
__attribute__((visibility("hidden"))) void fphip_dev_free(void *p, hipStream_t s);

// Pinned, host-coherent buffers (mailboxes, the enumeration context's solution ring and staging
// block) are CACHED for the life of the process: hipHostMalloc / hipHostFree synchronise the whole
// device like hipFree does — measured: closing a small enumeration context took 349 s because the
// config-3 tour of another context was in flight.  (Defined in enum_host.hip.)
__attribute__((visibility("hidden"))) void *fphip_pinned_get(size_t bytes);
__attribute__((visibility("hidden"))) void fphip_pinned_put(void *p);
#endif
