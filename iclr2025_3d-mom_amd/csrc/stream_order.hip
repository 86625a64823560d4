// Stream ordering for the host mirror: "stream A waits for what stream B holds now" and a small ring of reusable marks, as plain C
// calls.  The Python side orders a step's second stream with a dozen of these per iteration; through torch's Stream / Event objects
// each costs 8-10 us of host time (device-index resolution, object construction), which is a tenth of the render() + backward()
// path's host budget (tools/host_profile.py); a ctypes call into hipEventRecord + hipStreamWaitEvent costs about 1.5 us.
#include "mom_common.h"
#include <mutex>

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
bool ensure(hipEvent_t* e) { return *e || hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess; }
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
