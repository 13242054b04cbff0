// ftgp_device.h -- device-side state layout and wave-level helpers (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ftgp.h"

#define FTGP_WAVE 64

// Per-car state, array-of-structs in HBM: one wave owns one car in the step kernel and pulls the whole
// record with a handful of wave-uniform loads (320 B = 2.5 cache lines) instead of ~30 scattered lines.
struct alignas(16) CarCore {
    double x, y, qw, qz;          // planar pose (yaw as quaternion (qw, 0, 0, qz))
    double vx, vy, wz;            // world-frame linear velocity, yaw rate
    double qs, qsd;               // virtual steering joint
    double w[4];                  // wheel spin fl, fr, bl, br
    double u_speed, u_steer;      // controls
    double last_steer;            // fast.py:12
    double dist2;                 // squared distance to the centre-line (custom.py:1343)
    int32_t completion, laps, start, offset;
    int32_t good_start, finished, off_track, delta;
    int32_t n_times, pad0, pad1, pad2;
};
// The lap-time list stays in HBM (written on lap crossings only); the step kernel keeps just CarCore in registers.
struct alignas(16) CarState : CarCore {
    double times[FTGP_MAX_LAP_TIMES];
};
static_assert(sizeof(CarCore) == 192, "CarCore layout");
static_assert(sizeof(CarState) == 320, "CarState layout");

struct DeviceParams {
    // sizes
    int32_t n_envs, cars_per_env, n_cars, n_rays;
    int32_t lap_target, spawn_mode, ranges_stride, env_base;
    uint64_t seed;
    double dt;
    // track
    int32_t width, height, words_per_row, pad1;
    double px_size_x, px_size_y, origin_x, origin_y, inv_px_x, inv_px_y;
    float inv_px_x_f, inv_px_y_f;
    // two-level wall grid over 8x8-pixel blocks (built on the host at create; staged into LDS by the step kernel)
    int32_t nbx, nby, nwpr, n_fine;    // blocks per row / column, 32-block words per row, non-empty blocks
    const uint8_t* coarse;        // 4 bits per block: chessboard distance in BLOCKS to the nearest non-empty block (0 = has walls), clamp 15
    const uint2* rank;            // [nby][nwpr] {non-empty bits of 32 blocks, number of non-empty blocks before this word}
    const uint8_t* fine;          // [n_fine][32] 4 bits per pixel of each non-empty block: chessboard distance in PIXELS to the nearest wall pixel
    // LDS layout of the step kernel (byte offsets, all 16-B aligned)
    int32_t off_params, off_veh, off_fine, off_rank, off_path, off_coarse, off_ray, off_state, off_next, off_scan, lds_bytes, pad5;
    int32_t eighth, scan_floats, ray_floats;   // int(n_rays / 8); floats per LDS scan = 1 + (n_rays - 2*eighth) padded to 4; padded ray table
    float snap_eps, pad3;         // the march re-checks a landing point with the exact comparisons within this distance of a pixel boundary
    const uint32_t* field;        // flat per-pixel OCTANT field, [height][width][2] dwords, HBM/L2.  Dword 0 serves rays whose dominant
                                  // axis is x, dword 1 those with dominant axis y; byte q of a dword belongs to the direction quadrant
                                  // (q&1 ? -x : +x, q&2 ? -y : +y).  A byte describes a wall-free rectangle of pixels with its corner
                                  // at the pixel, extending AHEAD of the ray: low 7 bits = h (0 = the pixel is a wall); bit 7 clear:
                                  // h x h square; bit 7 set: 2h along the dominant axis by h across it.
    int32_t use_field;            // 1: the march reads `field` (flat, from L2); 0: the two-level grid staged in LDS
    int32_t scan_full;            // 1: the LDS scan holds the whole row (flushed to HBM with coalesced 16-B stores); 0: only the driver's window
    const void* veh_dev;          // VehLds image (vehicle constants + wheel loads) in HBM
    const double* path;           // [100][2]
    const double* spawn;          // [100][4] x, y, qw, qz
    const float* ray_bx;          // body-frame ray directions (binary32)
    const float* ray_by;
    // state
    CarState* cars;
    float* ranges;                // [n_cars][ranges_stride]
    int64_t* steps;               // [n_envs]
    FtgpVehicle veh;              // host-side copy (kernels read the LDS image)
    double wheel_load[4];
};

// ---------------------------------------------------------------------------------------------
// Specified polynomials (DESIGN.md "arithmetic rules"): Taylor series in Horner form on x*x.
// Written operation by operation; the library is compiled with -ffp-contract=off.
__device__ __forceinline__ double spec_sin(double x)
{
    double z = x * x;
    double p = -1.0 / 51090942171709440000.0;
    p = p * z + 1.0 / 121645100408832000.0;
    p = p * z - 1.0 / 355687428096000.0;
    p = p * z + 1.0 / 1307674368000.0;
    p = p * z - 1.0 / 6227020800.0;
    p = p * z + 1.0 / 39916800.0;
    p = p * z - 1.0 / 362880.0;
    p = p * z + 1.0 / 5040.0;
    p = p * z - 1.0 / 120.0;
    p = p * z + 1.0 / 6.0;
    p = p * z;
    return x - x * p;
}
__device__ __forceinline__ double spec_cos(double x)
{
    double z = x * x;
    double p = 1.0 / 2432902008176640000.0;
    p = p * z - 1.0 / 6402373705728000.0;
    p = p * z + 1.0 / 20922789888000.0;
    p = p * z - 1.0 / 87178291200.0;
    p = p * z + 1.0 / 479001600.0;
    p = p * z - 1.0 / 3628800.0;
    p = p * z + 1.0 / 40320.0;
    p = p * z - 1.0 / 720.0;
    p = p * z + 1.0 / 24.0;
    p = p * z - 0.5;
    p = p * z;
    return 1.0 + p;
}

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

// ---------------------------------------------------------------------------------------------
// wave64 helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & (FTGP_WAVE - 1); }

__device__ __forceinline__ double shfl_xor_f64(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, FTGP_WAVE);
    hi = __shfl_xor(hi, m, FTGP_WAVE);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, FTGP_WAVE);
    hi = __shfl(hi, src, FTGP_WAVE);
    return __hiloint2double(hi, lo);
}

// Orders this wave's LDS stores before its later LDS loads (lanes exchange data through LDS without a
// workgroup barrier: DS operations of one wave execute in order, the fence keeps the compiler honest).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
