// Replays a serial chain of per-time-step kernel launches from cached hipGraphs.
//
// The recurrent loops of this library (LSTM steps, denominator frames) are T dependent launches of
// the same kernel with a different step index.  Launched eagerly they are host-bound (~5 us per
// launch vs 4-6 us kernels); captured once into a hipGraph the launches are queued on the GPU
// back to back (MI355X_MICROARCH: dependent kernel boundary ~1.5 us, same as eager, no host cost).
//
// To make one captured graph serve every call (any T, any buffers) the kernels take a pointer to
// a parameter block in device memory plus a baked-in local index; the block holds the call's
// pointers/sizes and a running `base` step:  step = base + local, skipped when step >= T.
// A call writes the block with one tiny kernel (arguments travel by value, so no host buffer
// has to outlive the call), then replays graphs of 64 (and 8) steps until T is covered; the last
// node of each graph advances `base`.
#pragma once
#include <cstdlib>
#include <map>
#include <string>
#include <tuple>

#include "common.h"

namespace pk2 {

struct StepCounter { int base; int T; };

static __global__ void step_counter_bump(StepCounter* c, int n) { c->base += n; }
static __global__ void step_counter_set(StepCounter* c, int T) { c->base = 0; c->T = T; }

template <typename P>
__global__ void param_block_store(P value, P* dst) { *dst = value; }

class StepGraphs {
 public:
  // launch(stream, local_index) must enqueue exactly one kernel launch for local step `local`.
  template <typename LaunchFn>
  int run(const std::string& key, int T, StepCounter* counter, hipStream_t stream, LaunchFn launch) {
    hipLaunchKernelGGL(step_counter_set, dim3(1), dim3(1), 0, stream, counter, T);
    int done = 0;
    while (done < T) {
      const int len = pick_len(T - done);
      hipGraphExec_t exec;
      int rc = get(key, len, counter, launch, &exec);
      if (rc) return rc;
      PK2_HIP(hipGraphLaunch(exec, stream));
      done += len;
    }
    return PK2_OK;
  }

  // Two independent chains (each its own counter) replayed as the two parallel branches of one graph.
  template <typename LaunchA, typename LaunchB>
  int run2(const std::string& key, int T, StepCounter* ca, StepCounter* cb, hipStream_t stream, LaunchA la,
           LaunchB lb) {
    hipLaunchKernelGGL(step_counter_set, dim3(1), dim3(1), 0, stream, ca, T);
    hipLaunchKernelGGL(step_counter_set, dim3(1), dim3(1), 0, stream, cb, T);
    int done = 0;
    while (done < T) {
      const int len = (T - done >= kBig) ? kBig : kSmall;
      hipGraphExec_t exec;
      auto it = cache_.find({key, len});
      if (it != cache_.end()) {
        exec = it->second;
      } else {
        if (!cap_) PK2_HIP(hipStreamCreateWithFlags(&cap_, hipStreamNonBlocking));
        if (!cap2_) {
          PK2_HIP(hipStreamCreateWithFlags(&cap2_, hipStreamNonBlocking));
          PK2_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
          PK2_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
        }
        hipGraph_t graph;
        PK2_HIP(hipStreamBeginCapture(cap_, hipStreamCaptureModeThreadLocal));
        PK2_HIP(hipEventRecord(ev_fork_, cap_));
        PK2_HIP(hipStreamWaitEvent(cap2_, ev_fork_, 0));
        for (int j = 0; j < len; ++j) la(cap_, j);
        hipLaunchKernelGGL(step_counter_bump, dim3(1), dim3(1), 0, cap_, ca, len);
        for (int j = 0; j < len; ++j) lb(cap2_, j);
        hipLaunchKernelGGL(step_counter_bump, dim3(1), dim3(1), 0, cap2_, cb, len);
        PK2_HIP(hipEventRecord(ev_join_, cap2_));
        PK2_HIP(hipStreamWaitEvent(cap_, ev_join_, 0));
        PK2_HIP(hipStreamEndCapture(cap_, &graph));
        PK2_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        PK2_HIP(hipGraphDestroy(graph));
        cache_[{key, len}] = exec;
      }
      PK2_HIP(hipGraphLaunch(exec, stream));
      done += len;
    }
    return PK2_OK;
  }

  // Destroys the cached graphs of `key` (all lengths).  The caller makes sure none of them is still executing.
  void erase(const std::string& key) {
    for (auto it = cache_.begin(); it != cache_.end();) {
      if (it->first.first == key) { (void)hipGraphExecDestroy(it->second); it = cache_.erase(it); } else { ++it; }
    }
  }

 private:
  static constexpr int kBig = 64, kSmall = 8;
  // Graph lengths: the longest tier that fits the remaining steps (a last partial graph of kSmall early-exits).
  // A longer first tier (PK2_GRAPH_STEPS=256|512) was measured: no change in the step time (46.7 / 46.6 / 46.6 ms),
  // the graph boundaries are not where the time goes; the default stays at 64-step graphs.
  static int pick_len(int remaining) {
    static const int huge = [] { const char* e = getenv("PK2_GRAPH_STEPS"); const int v = e ? atoi(e) : kBig; return v < kBig ? kBig : v; }();
    if (remaining >= huge) return huge;
    if (remaining >= kBig) return kBig;
    return kSmall;
  }
  hipStream_t cap2_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  std::map<std::pair<std::string, int>, hipGraphExec_t> cache_;
  hipStream_t cap_ = nullptr;

  template <typename LaunchFn>
  int get(const std::string& key, int len, StepCounter* counter, LaunchFn launch, hipGraphExec_t* out) {
    auto it = cache_.find({key, len});
    if (it != cache_.end()) { *out = it->second; return PK2_OK; }
    if (!cap_) PK2_HIP(hipStreamCreateWithFlags(&cap_, hipStreamNonBlocking));
    hipGraph_t graph;
    PK2_HIP(hipStreamBeginCapture(cap_, hipStreamCaptureModeThreadLocal));
    for (int j = 0; j < len; ++j) launch(cap_, j);
    hipLaunchKernelGGL(step_counter_bump, dim3(1), dim3(1), 0, cap_, counter, len);
    PK2_HIP(hipStreamEndCapture(cap_, &graph));
    hipGraphExec_t exec;
    PK2_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    PK2_HIP(hipGraphDestroy(graph));
    cache_[{key, len}] = exec;
    *out = exec;
    return PK2_OK;
  }
};

// Device-resident parameter block + step counter, one per (kernel family, stream).
template <typename P>
struct ParamSlot {
  P* params = nullptr;
  StepCounter* counter = nullptr;
};

template <typename P>
int get_param_slot(std::map<std::pair<int, hipStream_t>, ParamSlot<P>>& slots, int family, hipStream_t stream,
                   ParamSlot<P>** out) {
  auto key = std::make_pair(family, stream);
  auto it = slots.find(key);
  if (it == slots.end()) {
    ParamSlot<P> s;
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&s.params), sizeof(P)));
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&s.counter), sizeof(StepCounter)));
    it = slots.emplace(key, s).first;
  }
  *out = &it->second;
  return PK2_OK;
}

}  // namespace pk2
