// Per-kernel HIP-event timing slots (bench.py's live roofline figure).  Disabled by default: the
// launchers call mom_prof_begin/end unconditionally and these return immediately unless a slot is enabled.
#include "mom_common.h"
#include <mutex>
#include <vector>

namespace {
struct Slot {
    bool on = false;
    int period = 1;              // time every period-th launch of the slot
    long long seen = 0;
    bool armed = false;          // this launch carries events
    std::vector<hipEvent_t> ev;  // begin/end pairs
    size_t used = 0;
    double total_ms = 0.0;
    long long count = 0;
};
Slot g_slots[MOM_PROF_SLOTS];
std::mutex g_mu;
const char* kNames[MOM_PROF_SLOTS] = {"preprocess_fwd", "tile_hist", "tile_scan", "tile_scatter", "tile_sort", "render_fwd",
                                      "render_bwd", "preprocess_bwd", "hexplane_fwd", "hexplane_bwd", "adam", "l1_loss",
                                      "plane_reg", "mlp_fwd", "mlp_bwd", "reserved"};

void drain(Slot& s)
{
    for (size_t i = 0; i + 1 < s.used; i += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(s.ev[i + 1]) == hipSuccess && hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]) == hipSuccess) {
            s.total_ms += ms;
            s.count++;
        }
    }
    s.used = 0;
}
}  // namespace

void mom_prof_begin(int slot, hipStream_t s)
{
    Slot& S = g_slots[slot];
    if (!S.on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    // An event pair around a kernel costs the stream two bubbles of ~6.5 us (the command processor cannot run a marker packet
    // underneath its neighbours the way it overlaps consecutive dispatches; tools/gap_stats.py): 13 us on a 960 us step.  A
    // sampled average costs a fraction of that and estimates the same mean.
    S.armed = (S.seen++ % S.period) == 0;
    if (!S.armed) return;
    if (S.used + 2 > S.ev.size()) {
        if (S.ev.size() >= 8192) drain(S);
        else
            for (int k = 0; k < 2; k++) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return;
                S.ev.push_back(e);
            }
    }
    (void)hipEventRecord(S.ev[S.used], s);
}
void mom_prof_end(int slot, hipStream_t s)
{
    Slot& S = g_slots[slot];
    if (!S.on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!S.armed || S.used + 2 > S.ev.size()) return;
    S.armed = false;
    (void)hipEventRecord(S.ev[S.used + 1], s);
    S.used += 2;
}

extern "C" int mom_profile_enable(int slot, int on)
{
    if (slot < 0 || slot >= MOM_PROF_SLOTS) return MOM_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    g_slots[slot].on = on != 0;
    g_slots[slot].period = on > 1 ? on : 1;
    g_slots[slot].seen = 0;
    g_slots[slot].armed = false;
    return MOM_OK;
}
extern "C" int mom_profile_read(int slot, double* total_ms, long long* count, int reset)
{
    if (slot < 0 || slot >= MOM_PROF_SLOTS || !total_ms || !count) return MOM_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    Slot& S = g_slots[slot];
    drain(S);
    *total_ms = S.total_ms;
    *count = S.count;
    if (reset) { S.total_ms = 0.0; S.count = 0; }
    return MOM_OK;
}
extern "C" const char* mom_profile_name(int slot) { return (slot < 0 || slot >= MOM_PROF_SLOTS) ? "" : kNames[slot]; }
