// fresh_alloc_probe.hip — a stand-alone probe for DESIGN.md section 6's hypothesis: device memory that
// hipMallocAsync has JUST handed out (after a hipFreeAsync of a smaller block and a stream synchronise: the default
// pool releases unused blocks at every synchronisation, so the new block is obtained from the driver anew) is
// cleared or remapped underneath its first user.  Every wavefront of a launch shaped like the enumeration walk
// (8192 waves, a private 12 KB region each) stores a pattern into its region and reads it back over and over for a
// few hundred microseconds; any word that does not read back as stored is counted.  Expected on a sound stack: 0.
//
//   hipcc --offload-arch=gfx950 -O2 -o fresh_alloc_probe tests/perf/micro/fresh_alloc_probe.hip && ./fresh_alloc_probe [rounds] [keep]
//   keep = 1: raise the pool's release threshold first (what a fphip context does since round 6: FPHIP_POOL_KEEP)
//   keep = 2: plain hipMalloc / hipFree instead of the stream-ordered calls
//
// Not run in round 6 (the GPU budget ended with the experiment that suggested it).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                      \
  do                                                                                                  \
  {                                                                                                   \
    hipError_t e_ = (x);                                                                              \
    if (e_ != hipSuccess)                                                                             \
    {                                                                                                 \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                         \
      return 2;                                                                                       \
    }                                                                                                 \
  } while (0)

__global__ void __launch_bounds__(128) probe(double *base, size_t per_wave, int passes, unsigned long long *bad,
                                             unsigned long long *first_bad_wave)
{
  const int lane    = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  double *mine      = base + wave * per_wave;
  unsigned long long wrong = 0;
  for (int p = 0; p < passes; ++p)
  {
    // "push": rows of 64 doubles, like the column stack's global slots
    for (size_t i = lane; i < per_wave; i += 64)
      mine[i] = (double)(wave * 4096 + i) + 0.25 * (p & 3);
    // "step": read them back a little later
    for (size_t i = lane; i < per_wave; i += 64)
      wrong += (mine[i] != (double)(wave * 4096 + i) + 0.25 * (p & 3));
  }
  if (wrong)
  {
    atomicAdd(bad, wrong);
    atomicMin(first_bad_wave, (unsigned long long)wave);
  }
}

int main(int argc, char **argv)
{
  const int rounds = argc > 1 ? atoi(argv[1]) : 50;
  const int keep   = argc > 2 ? atoi(argv[2]) : 0;
  CHECK(hipSetDevice(0));
  if (keep == 1)
  {
    hipMemPool_t pool;
    CHECK(hipDeviceGetDefaultMemPool(&pool, 0));
    uint64_t thr = ~(uint64_t)0;
    CHECK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
  }
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long *bad;
  CHECK(hipMalloc(&bad, 16));
  const size_t waves = 8192, small = 1456, large = 1520;  // doubles per wave: the 63- and the 64-level launches
  unsigned long long total = 0;
  for (int r = 0; r < rounds; ++r)
  {
    double *a = nullptr, *b = nullptr;
    // a context that has walked 63-level launches ...
    if (keep == 2)
      CHECK(hipMalloc((void **)&a, (waves * small + 256) * sizeof(double)));
    else
      CHECK(hipMallocAsync((void **)&a, (waves * small + 256) * sizeof(double), s));
    CHECK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(probe, dim3(waves / 2), dim3(128), 0, s, a, small, 2, bad, bad + 1);
    CHECK(hipStreamSynchronize(s));
    // ... meets its first 64-level one: free, allocate, synchronise, launch
    const unsigned long long init[2] = {0ull, ~0ull};
    CHECK(hipMemcpy(bad, init, 16, hipMemcpyHostToDevice));
    if (keep == 2)
    {
      CHECK(hipFree(a));
      CHECK(hipMalloc((void **)&b, (waves * large + 256) * sizeof(double)));
    }
    else
    {
      CHECK(hipFreeAsync(a, s));
      CHECK(hipMallocAsync((void **)&b, (waves * large + 256) * sizeof(double), s));
    }
    CHECK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(probe, dim3(waves / 2), dim3(128), 0, s, b, large, 64, bad, bad + 1);
    CHECK(hipStreamSynchronize(s));
    unsigned long long out[2];
    CHECK(hipMemcpy(out, bad, 16, hipMemcpyDeviceToHost));
    if (out[0])
      printf("round %d: %llu words read back wrong, first wave %llu\n", r, out[0], out[1]);
    total += out[0];
    if (keep == 2)
      CHECK(hipFree(b));
    else
      CHECK(hipFreeAsync(b, s));
    CHECK(hipStreamSynchronize(s));
  }
  printf("%d rounds, release threshold %s: %llu words read back wrong\n", rounds, keep == 2 ? "n/a (hipMalloc / hipFree)" : (keep ? "raised" : "default"), total);
  return total ? 1 : 0;
}
