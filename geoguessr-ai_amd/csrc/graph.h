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

// gemm_f32.hip: the split-K form of gg_gemm_nt_f32 reduces through library-owned slabs.  A captured graph gets slabs of its own (so that a replay computes what
// the eager call computed, and two graphs replayed on two streams share nothing): the cache watches whether a key's eager run used the form, allocates before
// the capture, registers the buffer for the capture stream and frees it with the graph.
long gg_gemm_f32_splitk_uses();
size_t gg_gemm_f32_splitk_bytes();
void gg_gemm_f32_capture_scratch(hipStream_t capture_stream, float* slabs);
void gg_gemm_f32_release_scratch();     // frees the per-(device, stream) slabs of eager launches (gg_graph_clear)
