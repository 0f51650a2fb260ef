// Shared host/device helpers for libdfusion_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/dfusion_hip.h"

namespace dfh {

// ---- error plumbing: nothing throws across the C ABI --------------------------------
char *last_error_buf();               // thread-local, 512 bytes (dfh_core.hip)
int fail(int code, const char *fmt, ...);

#define DFH_HIP_CHECK(expr)                                                               \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return ::dfh::fail(DFH_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define DFH_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return ::dfh::fail(DFH_E_BADARG, __VA_ARGS__); \
    } while (0)

// ---- small fixed-size parameter blocks passed by value to kernels ---------------------
struct Mat3 { double m[9]; };
struct Mat34 { double m[12]; };
struct Vec3 { double v[3]; };
struct DQ { double q[8]; };

constexpr int kWave = 64;             // CDNA wavefront
// Samples per tile of the planned normal-equation build (dfh_gn_plan_* / dfh_gn_build_planned*): one workgroup of kGnTile
// threads per tile, a scratch row never spans two tiles.  128 (round 3; 256 before): a tile's life is a chain of dependent
// phases (association 4.6 us, Jacobian rows 2.9, run bookkeeping 1.2, Gram 3.7-10 by the longest run), the launch lasts two
// tile lives whatever the occupancy, so tiles are made shorter and more numerous (25.6 KB of LDS each: five per CU).
#ifndef DFH_GN_TILE                 // (experiment builds: tools/build_variant.sh <name> -DDFH_GN_TILE=64)
#define DFH_GN_TILE 128
#endif
constexpr int kGnTile = DFH_GN_TILE;

// ---- development switches (dfh_set_option / DFH_OPTIONS, include/dfusion_hip.h) -----------
// One table, filled once at the first call into the library; -1 = unset.  The call paths read this struct, never the
// environment.  k1_*: depth -> TSDF sweep; k2/k3: volume -> volume fusion; gn_* / plan_* / pcg_*: the solve; py_*: switches of
// the Python layer (dynamicfusion_body_amd/solve.py, pipeline.py, device.py), kept in the same table so that ONE place
// (DFH_OPTIONS / dfh_set_option) turns every A/B switch of the build: the Python call paths read no environment variable.
#define DFH_OPTION_LIST(X)                                                                                                \
    X(k1_strided) X(k1_late_loads) X(k1_planes_per_block) X(k1_force_scalar) X(k1_bricks_min) X(k1_no_bricks)            \
    X(k1_bricks_nocull) X(k1_no_multi) X(k1_cull) X(k1_nzi) X(k1_prefetch) X(k1_nt)                  \
    X(k2_exact) X(k2_no_strided) X(k2_nt) X(k3_no_cache) X(k3_exact) X(k3_no_lds) X(k3_wg_per_cu) X(k3_tpb) X(k3_skip) X(plan_radix) X(rigid_atomic) X(gn_reg_own_launch) X(gn_reg_own_gather) \
    X(dbg_gather_part) X(pcg_wpb) X(pcg_multilaunch) X(pcg_spin_limit) X(pcg_one_xcd) X(gn_iter_own_clear) X(gn_gather_full) X(gn_no_view_cull)                                                            \
    X(py_plan_torch) X(py_allreduce_full) X(py_gn_atomic) X(py_gn_no_fused_assoc) X(py_gn_no_fused_iter) X(py_gn_iter_per_call) X(py_no_side_stream) X(py_no_host_scalars)
struct Options {
#define DFH_X(n) long n;
    DFH_OPTION_LIST(DFH_X)
#undef DFH_X
};
const Options &opt();                 // dfh_core.hip

// ---- what the library remembers per device (a process may drive several GPUs) ----------
struct DeviceInfo {
    int n_cu = 0;                     // 0 = not queried yet
    int pcg_occ512 = -1, pcg_occ1024 = -1;   // workgroups of the persistent PCG kernels that fit one CU; -1 = not queried yet
};
DeviceInfo &device_info(int device);  // dfh_core.hip
inline bool on(long v) { return v > 0; }

}  // namespace dfh
