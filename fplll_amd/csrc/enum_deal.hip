// enum_deal.hip — the content-sorted snake deal of the first walk round of a multi-rank enumeration, on the device.
//
// Every rank holds the same SET of final tasks in an order of its own (enum_host.hip, "Multi-GPU").  The deal
// sorts the tasks by content — (partial distance of the root ascending: the heaviest subtree first, then a 64-bit
// key of the coefficient prefix) — and hands position p of the sorted list to rank snake(p) = 0..W-1, W-1..0, ...
// Until round 5 the keys went to the HOST, which sorted 65 536 indices with an indirect comparison: 6-9 ms per
// call, on every rank — a tenth of an eight-GPU walk of the benchmark blocks (41 ms).  Here: two stable radix sorts
// (by key, then by the distance's bit pattern — non-negative doubles order like their bit patterns) and one
// scatter, all on the call's stream; the host only computes the length of its share.  Structural precedent:
// enumlib's sorted work list (enum-parallel/enumeration.h:417-422).  FPHIP_DEAL_HOST=1 keeps the host path (A/B).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace fphip
{

__global__ void deal_iota_kernel(unsigned *idx, unsigned n)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    idx[i] = i;
}
__global__ void deal_gather_pd_kernel(const double *pd, const unsigned *idx, unsigned long long *out, unsigned n)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    out[i] = (unsigned long long)__double_as_longlong(pd[idx[i]]);
}
// position p of the sorted list belongs to rank snake(p); this rank's share keeps the sorted order
__global__ void deal_scatter_kernel(const unsigned *order, const unsigned *slot_of, unsigned *mine, unsigned n, unsigned W,
                                    unsigned rank)
{
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n)
    return;
  const unsigned r     = p % (2 * W);
  const unsigned owner = r < W ? r : 2 * W - 1 - r;
  if (owner != rank)
    return;
  const unsigned t = order[p];
  mine[2 * (p / (2 * W)) + (r >= W ? 1u : 0u)] = slot_of ? slot_of[t] : t;
}

// number of positions p < n with snake(p) == rank
static unsigned share_len(unsigned n, unsigned W, unsigned rank)
{
  const unsigned full = n / (2 * W), rem = n % (2 * W);
  unsigned c = 2 * full;
  if (rem > rank)
    ++c;
  if (rem > 2 * W - 1 - rank)
    ++c;
  return c;
}

// keys[n], pd[n] (device, the task list's order), slot_of (device or null): -> mine[share] (device).  work: device
// scratch of at least deal_work_bytes(n) bytes.  Returns the length of the share, or ~0u on a HIP error.
size_t deal_work_bytes(unsigned n)
{
  size_t tmp = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                           (const unsigned *)nullptr, (unsigned *)nullptr, (int)n);
  return ((tmp + 255) & ~(size_t)255) + (size_t)n * (8 + 8 + 4 + 4) + 1024;
}

unsigned deal_tasks_device(hipStream_t s, const unsigned long long *keys, const double *pd, const unsigned *slot_of,
                           unsigned n, unsigned W, unsigned rank, unsigned *mine, void *work, size_t work_bytes)
{
  if (n == 0)
    return 0;
  size_t tmp = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                           (const unsigned *)nullptr, (unsigned *)nullptr, (int)n);
  tmp = (tmp + 255) & ~(size_t)255;
  if (work_bytes < tmp + (size_t)n * 24)
    return ~0u;
  char *w                = (char *)work;
  void *d_tmp            = w;
  unsigned long long *ka = (unsigned long long *)(w + tmp);
  unsigned long long *kb = ka + n;
  unsigned *ia           = (unsigned *)(kb + n);
  unsigned *ib           = ia + n;
  const unsigned g       = (n + 255) / 256;
  hipLaunchKernelGGL(deal_iota_kernel, dim3(g), dim3(256), 0, s, ia, n);
  // stable by key ...
  if (hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp, keys, ka, ia, ib, (int)n, 0, 64, s) != hipSuccess)
    return ~0u;
  // ... then stable by the distance: (distance, key) ascending
  hipLaunchKernelGGL(deal_gather_pd_kernel, dim3(g), dim3(256), 0, s, pd, ib, kb, n);
  if (hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp, kb, ka, ib, ia, (int)n, 0, 64, s) != hipSuccess)
    return ~0u;
  hipLaunchKernelGGL(deal_scatter_kernel, dim3(g), dim3(256), 0, s, ia, slot_of, mine, n, W, rank);
  if (hipGetLastError() != hipSuccess)
    return ~0u;
  return share_len(n, W, rank);
}

}  // namespace fphip
