// capi.hip -- library-level entry points of libpcl_hip.so (version, error string, launch helper).
#include "common.h"
#include <math.h>
#include <string.h>

namespace pcl {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
TimeHook& time_hook() {
    static thread_local TimeHook h = {nullptr, nullptr, "", "", ""};
    return h;
}
void set_launch_tag(const char* tag) {
    TimeHook& h = time_hook();
    strncpy(h.cur, tag ? tag : "", sizeof h.cur - 1);
    h.cur[sizeof h.cur - 1] = 0;
}
bool time_hook_matches(const TimeHook& h) { return h.want[0] == 0 || strcmp(h.want, h.cur) == 0; }
PathSwitches& path_switches() {
    static PathSwitches s = {1, 1, 1};
    return s;
}

}  // namespace pcl

extern "C" void pcl_time_next_launch(void* start_event, void* stop_event) {
    pcl::TimeHook& h = pcl::time_hook();
    h.start = static_cast<hipEvent_t>(start_event);
    h.stop = static_cast<hipEvent_t>(stop_event);
    h.want[0] = 0;
}

extern "C" void pcl_time_tagged_launch(void* start_event, void* stop_event, const char* tag) {
    pcl::TimeHook& h = pcl::time_hook();
    h.start = static_cast<hipEvent_t>(start_event);
    h.stop = static_cast<hipEvent_t>(stop_event);
    strncpy(h.want, tag ? tag : "", sizeof h.want - 1);
    h.want[sizeof h.want - 1] = 0;
}

extern "C" void pcl_set_kernel_paths(int fwd_resident, int narrow_stacks, int fused_backward) {
    pcl::PathSwitches& s = pcl::path_switches();
    if (fwd_resident >= 0) s.fwd_resident = fwd_resident != 0;
    if (narrow_stacks >= 0) s.narrow_stacks = narrow_stacks != 0;
    if (fused_backward >= 0) s.fused_backward = fused_backward != 0;
}
extern "C" const char* pcl_last_launch_kernel(void) {
    const char* k = pcl::time_hook().last_kernel;
    return k ? k : "";
}

extern "C" int pcl_version(void) { return 100; }   // 0.1.0
extern "C" const char* pcl_last_error(void) { return pcl::g_err; }

// misc/ops.py:110-111: 2 ** int(math.log(batch_size)) -- natural log, as written.
extern "C" int pcl_optimal_block(int batch_size) {
    if (batch_size < 1) return 1;
    return 1 << (int)log((double)batch_size);
}
