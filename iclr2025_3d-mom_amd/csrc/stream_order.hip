// Stream ordering for the host mirror: "stream A waits for what stream B holds now" and a small ring of reusable marks, as plain C
// calls.  The Python side orders a step's second stream with a dozen of these per iteration; through torch's Stream / Event objects
// each costs 8-10 us of host time (device-index resolution, object construction), which is a tenth of the render() + backward()
// path's host budget (tools/host_profile.py); a ctypes call into hipEventRecord + hipStreamWaitEvent costs about 1.5 us.
#include "mom_common.h"
#include <mutex>
#include <stdlib.h>

namespace {
constexpr int kDevices = 64;
struct PerDevice {
    hipEvent_t marks[MOM_STREAM_MARKS] = {};
    hipEvent_t pair = nullptr;                  // scratch event of mom_stream_wait_stream (record + wait capture the state at once)
};
PerDevice g_dev[kDevices];
std::mutex g_mu;

PerDevice* current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kDevices) return nullptr;
    return &g_dev[d];
}
}  // namespace

// Flags of every event this library orders streams with.  These events are only ever waited for by another stream of the SAME device
// (hipStreamWaitEvent), never by the host, so the marker needs no system-scope release of its own: the kernels on either side carry
// their own agent-scope fences, which is all two streams of one device need (and all two consecutive kernels of one stream get).
// tools/probe/marker_cost.hip: an event record between two kernels costs the recording stream 6.2 us with the default fence and 3.4 us
// without -- the fused step's main stream carries three per iteration.  MOM_EVENT_SYSTEM_FENCE=1 restores the default.
unsigned mom_order_event_flags()
{
    static const unsigned flags = [] {
        const char* e = getenv("MOM_EVENT_SYSTEM_FENCE");
        return (unsigned)hipEventDisableTiming | ((e && e[0] == '1') ? 0u : (unsigned)hipEventDisableSystemFence);
    }();
    return flags;
}

namespace {
bool ensure(hipEvent_t* e) { return *e || hipEventCreateWithFlags(e, mom_order_event_flags()) == hipSuccess; }
}  // namespace

int mom_stream_wait_stream(mom_stream_t waiter, mom_stream_t signaler)
{
    if (waiter == signaler) return MOM_OK;
    std::lock_guard<std::mutex> lock(g_mu);
    PerDevice* pd = current_device();
    if (!pd || !ensure(&pd->pair)) return MOM_ELAUNCH;
    // re-recording an event a stream still waits for is allowed: a wait refers to the record that preceded it
    if (hipEventRecord(pd->pair, (hipStream_t)signaler) != hipSuccess) return MOM_ELAUNCH;
    return hipStreamWaitEvent((hipStream_t)waiter, pd->pair, 0) == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

int mom_stream_mark(int slot, mom_stream_t stream)
{
    if (slot < 0 || slot >= MOM_STREAM_MARKS) return MOM_EINVAL;
    std::lock_guard<std::mutex> lock(g_mu);
    PerDevice* pd = current_device();
    if (!pd || !ensure(&pd->marks[slot])) return MOM_ELAUNCH;
    return hipEventRecord(pd->marks[slot], (hipStream_t)stream) == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

int mom_stream_wait_mark(mom_stream_t stream, int slot)
{
    if (slot < 0 || slot >= MOM_STREAM_MARKS) return MOM_EINVAL;
    std::lock_guard<std::mutex> lock(g_mu);
    PerDevice* pd = current_device();
    if (!pd || !pd->marks[slot]) return MOM_EINVAL;          // never recorded
    return hipStreamWaitEvent((hipStream_t)stream, pd->marks[slot], 0) == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

int mom_zero_async(void* ptr, size_t bytes, mom_stream_t stream)
{
    if (!bytes) return MOM_OK;
    if (!ptr) return MOM_EINVAL;
    return hipMemsetAsync(ptr, 0, bytes, (hipStream_t)stream) == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
