// HIP-graph cache behind gg_tinyvit_forward / gg_tinyvit_backward (graph.h).
#include "graph.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/gg.h"
#include "prof.h"

bool gg_prof_is_on();

namespace {
struct Slot {
    std::string key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool refused = false;          // capture / instantiation / a launch failed once: this key stays eager
    bool capturing = false;        // a thread is capturing this key right now (outside the lock): everybody else runs it eagerly meanwhile
    bool first_run = false;        // the key's first (eager) run is still in flight on some thread: wants_slabs is not known yet, nobody captures it meanwhile
    bool wants_slabs = false;      // the key's eager run issued split-K GEMMs: its graph gets split-K slabs of its own
    float* slabs = nullptr;
    uint64_t tick = 0;
};
std::mutex g_mu;
std::vector<Slot> g_slots;
uint64_t g_tick = 0;
std::map<int, hipStream_t> g_capture;      // per device
int g_mode = -2;                   // -2: read GG_GRAPH on first use; -1 auto; 0 off; 1 on
long g_captures = 0, g_replays = 0, g_eager = 0;
constexpr size_t kMaxSlots = 16;
constexpr long kMaxWasted = 8;       // evicted captured graphs after which no new graph is captured (a caller whose buffers change address every call)
long g_wasted = 0;

void drop(Slot& s) {
    if (s.exec) (void)hipGraphExecDestroy(s.exec);
    if (s.graph) (void)hipGraphDestroy(s.graph);
    if (s.slabs) (void)hipFree(s.slabs);
    s.exec = nullptr; s.graph = nullptr; s.slabs = nullptr;
}
int mode() {
    if (g_mode == -2) {
        const char* e = getenv("GG_GRAPH");
        g_mode = !e || !*e ? -1 : (atoi(e) != 0 ? 1 : 0);
    }
    return g_mode;
}
}  // namespace

bool gg_graph_wanted(bool launch_bound) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (gg_prof_is_on()) return false;
    const int m = mode();
    return m == 1 || (m == -1 && launch_bound);
}

int gg_graph_run(const GgGraphKey& key_in, hipStream_t stream, const std::function<int(hipStream_t)>& body) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return body(stream); }
    GgGraphKey key = key_in;
    key.add(dev);                                          // a graph belongs to the device it was captured on
    std::unique_lock<std::mutex> lk(g_mu);
    Slot* s = nullptr;
    for (auto& c : g_slots) if (c.key == key.bytes) { s = &c; break; }
    if (!s) {
        // first sighting: run eagerly (one-time function attributes, lazily allocated scratch and static tables are set on this path) and remember the key
        if (g_slots.size() >= kMaxSlots) {
            size_t lru = 0;
            for (size_t i = 1; i < g_slots.size(); ++i) if (g_slots[i].tick < g_slots[lru].tick) lru = i;
            if (g_slots[lru].exec) ++g_wasted;            // a captured graph leaves the cache: the caller's addresses churn
            drop(g_slots[lru]);
            g_slots.erase(g_slots.begin() + lru);
        }
        g_slots.emplace_back();
        g_slots.back().key = key.bytes;
        g_slots.back().tick = ++g_tick;
        g_slots.back().first_run = true;               // (another thread that meets the key before this run has finished must not capture: wants_slabs is decided below)
        ++g_eager;
        lk.unlock();
        const long uses0 = gg_gemm_f32_splitk_uses();
        const int rc = body(stream);
        const bool used = gg_gemm_f32_splitk_uses() != uses0;
        lk.lock();
        for (auto& c : g_slots) if (c.key == key.bytes) { c.wants_slabs = used; c.first_run = false; break; }
        return rc;
    }
    s->tick = ++g_tick;
    if (s->refused || s->first_run) { ++g_eager; lk.unlock(); return body(stream); }
    if (!s->exec && g_wasted >= kMaxWasted) { ++g_eager; lk.unlock(); return body(stream); }      // captures keep being evicted unused-again: stop paying for new ones
    if (s->capturing) { ++g_eager; lk.unlock(); return body(stream); }
    if (!s->exec) {
        // second sighting: capture on this device's private stream (the caller's may be the legacy default stream, which cannot be captured).  The capture runs
        // WITHOUT the cache lock -- it walks the whole network's launch code, and another thread's call on another device must not wait for it -- under a
        // per-device capture lock (one private stream per device); the slot is marked so that nobody replays or re-captures it half-built, and is looked up
        // again by key afterwards (the slot vector may have moved)
        static std::mutex cap_mu[16];
        const std::string skey = s->key;
        const bool wants_slabs = s->wants_slabs;
        s->capturing = true;
        lk.unlock();
        hipGraph_t g = nullptr;
        hipGraphExec_t x = nullptr;
        float* slabs = nullptr;
        bool ok = false;
        // no slabs for a key whose eager run used the split-K form: a graph captured without them would run those GEMMs unsplit and its replays would differ
        // from the eager call in rounding -- the key stays eager instead (replays return the eager call's bits, or there are no replays)
        const bool no_slabs = wants_slabs && hipMalloc((void**)&slabs, gg_gemm_f32_splitk_bytes()) != hipSuccess;
        if (no_slabs) { (void)hipGetLastError(); slabs = nullptr; }
        if (!no_slabs) {
            std::lock_guard<std::mutex> cl(cap_mu[dev & 15]);
            hipStream_t cap = nullptr;
            { std::lock_guard<std::mutex> l2(g_mu); cap = g_capture[dev]; }
            if (!cap && hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) == hipSuccess) { std::lock_guard<std::mutex> l2(g_mu); g_capture[dev] = cap; }
            if (cap && slabs) gg_gemm_f32_capture_scratch(cap, slabs);
            if (cap && hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = body(cap);
                const hipError_t ec = hipStreamEndCapture(cap, &g);
                // rc != 0: nothing was enqueued.  The failure may belong to the capture (an allocation, a query): this key stays eager -- a genuine argument
                // error fails again below with its own message
                if (rc == 0 && ec == hipSuccess && g && hipGraphInstantiate(&x, g, nullptr, nullptr, 0) == hipSuccess && x) ok = true;
            }
            if (cap) gg_gemm_f32_capture_scratch(cap, nullptr);
            if (!ok) { (void)hipGetLastError(); if (x) (void)hipGraphExecDestroy(x); if (g) (void)hipGraphDestroy(g); x = nullptr; g = nullptr; }
        }
        if (!ok && slabs) { (void)hipFree(slabs); slabs = nullptr; }
        lk.lock();
        s = nullptr;
        for (auto& c : g_slots) if (c.key == skey) { s = &c; break; }
        if (!s) {                                          // evicted meanwhile
            if (x) (void)hipGraphExecDestroy(x);
            if (g) (void)hipGraphDestroy(g);
            if (slabs) (void)hipFree(slabs);
            ++g_eager; lk.unlock();
            return body(stream);
        }
        s->capturing = false;
        if (!ok) { s->refused = true; ++g_eager; lk.unlock(); return body(stream); }
        s->graph = g; s->exec = x; s->slabs = slabs;
        ++g_captures;
    }
    if (hipGraphLaunch(s->exec, stream) != hipSuccess) {
        // nothing was enqueued: this key stays eager from now on (a later call must not fail the same way), and this call runs eagerly
        (void)hipGetLastError();
        s->refused = true;
        drop(*s);
        ++g_eager;
        lk.unlock();
        return body(stream);
    }
    ++g_replays;
    if (g_wasted > 0 && (g_replays & 15) == 0) --g_wasted;      // the waste counter decays while graphs are being reused: a caller whose addresses churned for a while is not locked out for good
    return 0;
}

extern "C" int gg_graph_set_mode(int m) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (m < -1 || m > 1) { gg_set_error("gg_graph_set_mode: mode must be -1 (auto), 0 (off) or 1 (on)"); return -1; }
    g_mode = m;
    return 0;
}
extern "C" int gg_graph_stats(int64_t* captures, int64_t* replays, int64_t* eager) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (captures) *captures = g_captures;
    if (replays) *replays = g_replays;
    if (eager) *eager = g_eager;
    return (int)g_slots.size();
}
extern "C" int gg_graph_clear(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& s : g_slots) drop(s);
    g_slots.clear();
    g_wasted = 0;
    gg_gemm_f32_release_scratch();
    return 0;
}
