// pruner_engine.h — the one expensive primitive of the pruner and its two back ends.
//
// Everything the pruner's cost model needs beyond O(n) host arithmetic is the volume of an "even
// simplex": V_k(y) = k! * vol{ 0 <= t_1 <= ... <= t_k , t_j <= y_j / y_k } for a bound vector y and
// k = 1 .. m (fplll: Pruner::relative_volume, pruner/pruner_simplex.h:34-46) — O(k^2) dependent
// operations each, O(m^3) for one cost value, and a search asks for thousands of them.  The model
// (pruner_search.hip) therefore never evaluates ONE candidate: it hands a whole batch of bound
// vectors plus a job list (vector, k) to a VolumeEngine.  The device engine runs one lane per job
// with the polynomial of the lane as a column of LDS (pruner_volume.hip); the host engine is the same
// recurrence as a loop, for callers without a context and for batches too small to pay for a launch.
// Both produce the SAME doubles (one fixed operation sequence of +, *, / per value, no contraction).
#ifndef FPHIP_PRUNER_ENGINE_H
#define FPHIP_PRUNER_ENGINE_H

#include <atomic>
#include <cstddef>

namespace fphip_pruner
{
struct VolumeJob
{
  int vec;  // row of the bounds matrix
  int k;    // 1 .. m: which volume of that row
};

class VolumeEngine
{
public:
  virtual ~VolumeEngine() {}
  // bounds: nvec rows of m doubles; out[j] = V_{jobs[j].k}(row jobs[j].vec).  false = device failure
  // (message through error()).
  virtual bool run(const double *bounds, int nvec, int m, const VolumeJob *jobs, int njobs, double *out) = 0;
  // how many candidates of a sequential search are worth evaluating ahead of the decision that
  // consumes them (1 = none: the host pays for every wasted candidate, the device does not)
  virtual int lookahead() const = 0;
  virtual const char *error() const { return ""; }
  // accounting (tests, bench): jobs evaluated by a kernel / inline on the host, kernel launches
  // (atomic: the host engine is one object shared by the service workers of an in-loop BKZ)
  std::atomic<unsigned long long> device_jobs{0}, host_jobs{0}, launches{0};
};

// V_k(y) on the host; poly = k + 1 doubles of scratch
double simplex_volume(const double *y, int k, double *poly);
VolumeEngine *host_volume_engine();  // stateless singleton
// A device engine owns a stream of its own (it must make progress while a persistent reduction kernel
// occupies the context's stream), pinned staging and device buffers; not thread-safe: one per thread.
VolumeEngine *create_device_volume_engine(int device, char *err, size_t errlen);
void destroy_volume_engine(VolumeEngine *e);
}  // namespace fphip_pruner
#endif
