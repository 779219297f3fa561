// common.h -- shared helpers for the gfx950 kernels of libpcl_hip.so.
// Wave = 64 lanes everywhere (CDNA4); no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include "../../include/pcl_hip.h"

namespace pcl {

void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PCL_EHIP, "%s: %s", what, hipGetErrorString(e));
    return PCL_OK;
}

#define PCL_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::pcl::fail(PCL_EINVAL, __VA_ARGS__); } while (0)

// Every entry point converts its `void* stream` exactly once, before launching: we use that moment to drop a
// stale sticky error left by somebody else's HIP call in this thread, so check_launch() reports only ours.
static inline hipStream_t as_stream(void* s) {
    (void)hipGetLastError();
    return reinterpret_cast<hipStream_t>(s);
}

// Measurement hook (bench.py): pcl_time_next_launch(start, stop) arms two HIP events for the next GEMM-family kernel this
// thread launches; they receive the kernel's own begin / end timestamps (the dispatch packet's, what rocprofv3 reports)
// instead of the times two marker packets around the call would see.  Unarmed (normal operation) it is a plain launch.
// Inside a per-stack entry point (stack.hip) many such kernels are launched by ONE call: each carries a launch tag
// ("fb256x128", "fwd128x256", ...; set_launch_tag) and pcl_time_tagged_launch arms the events for the next launch whose tag
// matches (an empty wanted tag matches anything).
struct TimeHook { hipEvent_t start, stop; char want[32]; char cur[32]; const char* last_kernel; };
TimeHook& time_hook();
// Lab switches of the kernel selection (pcl_set_kernel_paths: a C call -- the library reads no environment variables):
// resident-weight forward, recompute-per-pass narrow stacks, fused dX + dW backward.  All on by default.
struct PathSwitches { int fwd_resident, narrow_stacks, fused_backward; };
PathSwitches& path_switches();
void set_launch_tag(const char* tag);
bool time_hook_matches(const TimeHook& h);
#define PCL_LAUNCH_TIMED(kernel, grid, blk, st, ...)                                                                   \
    do {                                                                                                                \
        ::pcl::TimeHook& h_ = ::pcl::time_hook();                                                                       \
        h_.last_kernel = #kernel;    /* pcl_last_launch_kernel(): which kernel an entry point chose (profiling tools) */ \
        if (h_.start && ::pcl::time_hook_matches(h_)) {                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, blk, 0, st, h_.start, h_.stop, 0, __VA_ARGS__);                         \
            h_.start = h_.stop = nullptr;                                                                               \
        } else hipLaunchKernelGGL(kernel, grid, blk, 0, st, __VA_ARGS__);                                               \
    } while (0)

// XCD-aware logical block id for a 1-D grid whose consecutive blocks walk the clouds one after another (`bpc` blocks per
// cloud, nblk = clouds * bpc).  The hardware deals consecutive workgroup ids to the 8 XCDs round-robin, each with its own
// 4 MB L2: with the plain id every XCD touches every cloud's table.  Here XCD j takes the clouds j, j+8, ... whole, one
// after the other, so a cloud's table is gathered by workgroups that share an L2.  Identity when the shape does not divide.
__device__ __forceinline__ unsigned xcd_cloud_block(unsigned lin, unsigned nblk, unsigned bpc) {
    if (bpc == 0 || nblk % (8u * bpc) != 0) return lin;
    const unsigned slot = lin >> 3;
    return ((lin & 7u) + 8u * (slot / bpc)) * bpc + slot % bpc;
}

// ---------------------------------------------------------------- device helpers
#define PCL_WAVE 64

// Single-rounded fp32 ops: index-producing kernels must evaluate distances exactly as the source
// text of the reference writes them (no FMA contraction), whatever the compiler flags.
__device__ __forceinline__ float sq_dist3(float ax, float ay, float az, float bx, float by, float bz) {
    // (ax-bx)*(ax-bx) + (ay-by)*(ay-by) + (az-bz)*(az-bz), left to right
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v, unsigned old) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, 0xf, false);
}

// Wave-wide max / min of a 32-bit unsigned value; result is wave-uniform (SGPR).
// 4 DPP steps reduce each row of 16 lanes, 4 readlanes + scalar ops combine the rows.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, dpp_u32<0xB1>(v, v));    // quad_perm [1,0,3,2]
    v = max(v, dpp_u32<0x4E>(v, v));    // quad_perm [2,3,0,1]
    v = max(v, dpp_u32<0x141>(v, v));   // row_half_mirror
    v = max(v, dpp_u32<0x140>(v, v));   // row_mirror
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, dpp_u32<0xB1>(v, v));
    v = min(v, dpp_u32<0x4E>(v, v));
    v = min(v, dpp_u32<0x141>(v, v));
    v = min(v, dpp_u32<0x140>(v, v));
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int mbcnt(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

}  // namespace pcl
