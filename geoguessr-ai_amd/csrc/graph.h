// Captured HIP graphs for launch-bound calls (one serving panorama, small training batches): a whole-encoder call is a fixed sequence of a few hundred
// kernel launches whose arguments depend only on the call's own arguments, so the second call with the same arguments is captured on a private
// stream and every later one replays the instantiated graph on the caller's stream.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <string>

// Key: every argument the launch sequence depends on, as bytes (pointers included: a graph node holds the addresses it was captured with).
struct GgGraphKey {
    std::string bytes;
    template <class T> GgGraphKey& add(const T& v) { bytes.append(reinterpret_cast<const char*>(&v), sizeof(T)); return *this; }
    GgGraphKey& add_bytes(const void* p, size_t n) { const uint64_t len = n; add(len); if (p && n) bytes.append(reinterpret_cast<const char*>(p), n); return *this; }
};

// mode: GG_GRAPH=0 never, GG_GRAPH=1 always, unset: when the caller says the call is launch-bound.  Never while the per-launch event timing is on.
bool gg_graph_wanted(bool launch_bound);
// Runs `body(stream)` eagerly the first time a key is seen, captures it the second time and replays it afterwards.  `body` must only enqueue work on the
// stream it is given (no synchronisation, no allocation, no host-side dependence on device results).
int gg_graph_run(const GgGraphKey& key, hipStream_t stream, const std::function<int(hipStream_t)>& body);
