// ftgp_device.h -- device-side state layout and wave-level helpers (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ftgp.h"
#include "ftgp_march.h"

#define FTGP_WAVE 64
#define FTGP_MAX_GROUPS 256              // groups of 64 rays per car: n_rays <= 16384
#define FTGP_MAX_CARS_PER_BLOCK 16      // K1 / K3 run one car per lane group of a single wave: 16 cars x 4 lanes

// Per-car state, array-of-structs in HBM.  The step kernel keeps CarCore in LDS for the whole launch.
struct alignas(16) CarCore {
    double x, y, qw, qz;          // planar pose (yaw as quaternion (qw, 0, 0, qz))
    double vx, vy, wz;            // world-frame linear velocity, yaw rate
    double qs, qsd;               // virtual steering joint
    double w[4];                  // wheel spin fl, fr, bl, br
    double u_speed, u_steer;      // controls
    double last_steer;            // fast.py:12
    double dist2;                 // squared distance to the centre-line (custom.py:1343)
    int32_t completion, laps, offset, n_times;
    int32_t good_start, finished, off_track, delta;
    int64_t start;                // env step of the last counted line crossing (vehicle_state.start, custom.py:1362): 64 bits like self.steps
    int64_t finish_step;          // env step at which `finished` was set (custom.py:1367-1370); meaningful while finished != 0
};
// The lap-time list stays in HBM (written on lap crossings only).
struct alignas(16) CarState : CarCore {
    double times[FTGP_MAX_LAP_TIMES];
};
static_assert(sizeof(CarCore) == 192 && offsetof(CarCore, finish_step) == offsetof(CarCore, start) + 8, "CarCore layout (progress_update reaches finish_step through &start)");
static_assert(sizeof(CarState) == 192 + 8 * FTGP_MAX_LAP_TIMES, "CarState layout");
static_assert((FTGP_MAX_LAP_TIMES & (FTGP_MAX_LAP_TIMES - 1)) == 0, "the lap-time ring is indexed with a mask");

// LiDAR frame of one car, refreshed in LDS before every sweep: the sweep never reads the live state, so the dynamics
// of the same step can run beside it.  The second half is what OTHER cars of the env need to see this car.
struct alignas(16) LidarFrame {
    float u0, v0, chf, shf;       // LiDAR centre in pixels (binary32 of the binary64 value), heading (cos, sin) in binary32
    double lcx, lcy;              // LiDAR centre, world                         } FTGP_LIDAR_FAKELIDAR: i_x, i_y = the car's position in pixels
    double x, y, qw, qz;          // pre-step pose                               } (custom.py:1382-1384) and x, y = heading (cos, sin) instead
    int32_t finished;
    int32_t slot0;                // first car slot (inside the workgroup) of this car's env      } multi-car envs only:
    float fx, fy;                 // binary32 of x, y: what the inter-vehicle cull compares        } see frame_write()
};
static_assert(sizeof(LidarFrame) == 80, "LidarFrame layout");

// Multi-car envs: what a ray of car `me` needs to know about env-mate k to rule it out without looking at it.  B = mate's
// origin - my LiDAR centre; every visible part of the mate lies within `cull` of its origin, so a ray of direction d can only
// touch it if the mate sits in front and within `cull` of the ray's line, i.e. if  B . d >= sqrt(|B|^2 - cull^2) =: t.
// t = +inf: never (myself, a finished car -- invisible, custom.py:1441-1466 -- or my own rangefinders are off);
// t = -inf: always look (the mate is closer than 2 cull + ring radius, where the bound does not hold).
struct alignas(16) PairCull { float bx, by, t, pad; };
#define FTGP_PAIR_STRIDE 8              // FtgpConfig.cars_per_env <= 8

struct DeviceParams {
    // sizes
    int32_t n_envs, cars_per_env, n_cars, n_rays;
    int32_t lap_target, spawn_mode, ranges_stride, env_base;
    uint64_t seed;
    double dt;
    double rpp;                   // radians per LiDAR point, (2 pi) / n_rays (nidc.py:121), divided once on the host
    float two_over_rpp, pad_f0;   // 2 / rpp in binary32: seeds the cover-count search (cover_count), nothing else
    // track
    int32_t width, height, words_per_row, fstride;      // fstride = width + 2: cells per row of a field plane (one-pixel ring)
    double px_size_x, px_size_y, origin_x, origin_y, inv_px_x, inv_px_y;
    float inv_px_x_f, inv_px_y_f;
    float snap_eps;               // the march re-checks a landing point with the exact comparisons within this distance of a pixel boundary
    uint32_t reserved_m0;         // (rounds 2 - 4: the ray pool's division by multiplication)
    uint32_t plane256;            // bytes per (padded) sector plane of the box field / 256
    int32_t n_sectors;            // direction sectors of this handle's box field: 8 x (1, 2, 4 or 8 slope slices per octant)
    float slice_factor;           // FTGP_SLICE_FACTOR(FTGP_SLOPE_SLICES): a ray's sector is found among all FTGP_SECTORS, sector_tab maps it to planes
    int32_t n_planes, reserved4;  // planes of the field (= n_sectors)
    int32_t contact_reach;        // chessboard reach (pixels) of the wall-contact window; `nearbits` is the wall set dilated by it
    // step-kernel launch shape and LDS layout (byte offsets, all 16-B aligned)
    int32_t cars_per_block, waves_per_block, eighth, win_floats;   // win_floats: floats per LDS scan row: (eighth % 4) + (n_rays - 2*eighth) + 1, padded to 4
    int32_t off_params, off_veh, off_path, off_ray, off_cars, off_frame, off_steps, off_scan, off_list, off_pool, off_k1, off_cover, lds_bytes, cover_kmax;
    int32_t off_mmask, mmask_stride;      // multi-car envs: [2][cars_per_block][mmask_stride] bytes, which env-mates a ray group can see (mate_masks)
    int32_t bubble_wrap;          // custom.py:1041-1055: the four wheel softeners collide with the walls
    int32_t n_cu, pad_c;          // compute units of the device (sweep_priority)
    float edge_margin, pad_d;     // pixels: a LiDAR centre at least this far from every image edge has all its ray origins on the image (frame_write)
    // FTGP_LIDAR_FAKELIDAR (raycast.py:5-21 inside the step loop): the distance transform, the binary64 fan and the world size of the map
    const double* edt;            // [height][width] exact Euclidean distance to the nearest wall pixel (0 on walls), pixels; null in RANGEFINDER mode
    const double* fan_dirs;       // [n_rays][2] body-frame fan directions, binary64
    double map_size;              // 20 * scale = 40 (custom.py:1382)
    int32_t lidar_mode, pad_e;
    int32_t tasks_per_car;        // the sweep's work list per car (lidar_groups): groups of 64 consecutive rays, or pairs of opposite groups
    float group_cg, group_sg;     // cos / sin of the half-width of a ray group as mate_masks() needs it (cg < -1: no bound, e.g. a caller's fan)
    uint32_t reserved_m1;         // (rounds 4 - 5: the group draw's division by multiplication; a draw now reads its task descriptor, task_tab)
    const uint16_t* field;        // [n_sectors][height + 2][width + 2] box entries (ftgp_march.h), HBM/L2
    const uint32_t* bits;         // [height][words_per_row] wall bitmap
    const uint32_t* nearbits;     // [height][words_per_row] wall bitmap dilated by contact_reach (early-out of the wall contact)
    const void* veh_dev;          // VehLds image (vehicle constants + wheel loads) in HBM
    const double* path;           // [100][2]
    const double* spawn;          // [100][4] x, y, qw, qz
    const float* ray_dir;         // [n_rays][2] body-frame ray directions (sin phi, -cos phi), binary32
    const float* cover_thr;       // [2][cover_kmax + 1 (padded to 4)] cover-count thresholds of the nidc / fast drivers (cover_count)
    // state
    CarState* cars;
    float* ranges;                // [n_cars][ranges_stride]
    int64_t* steps;               // [n_envs]
    // end-of-launch metrics (ftgp_step_kernel's epilogue): one FTGP_METRIC_DOUBLES record per workgroup, an arrival counter, and
    // the two places the last workgroup to arrive writes the launch's record to (device memory for the RCCL all-gather, pinned
    // host memory for the caller); null: the step kernel leaves the metrics to ftgp_metrics_kernel
    double* wg_metrics;
    unsigned int* wg_ticket;
    double* metrics_dev;          // [2][FTGP_METRIC_DOUBLES]: the launch's slot (a kernel argument, alternating per launch) says which
    double* metrics_host;         // [2][FTGP_METRIC_DOUBLES]
    double* wg_metrics_host;      // [2][workgroups][FTGP_METRIC_DOUBLES] pinned host memory: the partial records of a launch whose record nothing on the device
                                  // needs (one rank, no communicator) -- the host adds them up
    alignas(16) int32_t sector_tab[FTGP_SECTORS][4];     // ftgp_sector_entry() of every sector (staged into LDS with the head of the block)
    FtgpVehicle veh;              // host-side copy (the step kernel reads the LDS image VehLds; from here on nothing is staged into LDS)
    double wheel_load[4];
    const unsigned char* stage_img;      // what every workgroup stages at the start of a launch, in ONE pass: the image of the LDS bytes [off_params, off_cars)
                                  // (head of this block | VehLds | path | fan), then the cover tables of nidc and of fast, stage_cover bytes each
    int32_t stage_cover, reserved3;
    int32_t car_policy[FTGP_MAX_CARS_PER_BLOCK];       // FTGP_POLICY_PER_CAR: the driver of the workgroup's car slot c (= the roster's entry c % cars_per_env: a workgroup
                                  // holds whole envs); 0 = not set.  Read with scalar loads.
    int32_t group_order[FTGP_MAX_GROUPS];      // the tasks of a car by expected march length, longest first (rays along the car's axis look down the
                                  // track, sideways rays hit the corridor wall at once): first ray | kind << 16, kind 0 = one group of 64
                                  // consecutive rays, 1 = that group and the opposite one (first ray + n/2), 2 = the short ends of both halves in
                                  // one group (lanes 0..31 / 32..63); read with scalar loads
    const int32_t* task_tab;      // [2][cars_per_block * tasks_per_car][4] the sweep's draws, one 16-byte scalar load each (lidar_groups; behind this block in device memory):
                                  //   [0] first ray (14 bits) | kind << 14 | car slot << 16 | window class of the first / second group << 20 / 22 | (the first group holds ray 0) << 24
                                  //       (window class: 0 = no ray of the group lies in the drivers' scan window, 1 = every ray does, 2 = test per ray)
                                  //   [1] byte offset of the car's LidarFrame in a frame buffer | rank of the task among the car's << 16
                                  //   [2] byte offset of the car's row of ranges from the workgroup's first row
                                  //   [3] byte offset, in a scan buffer, at which sample j of the car sits when 4 j is added: row + ((eighth & 3) - eighth) floats
                                  // the second table is for launches whose drivers do not read the scan: window classes 0, no ray 0
};

// vehicle constants as staged into LDS
struct VehLds {
    FtgpVehicle v;
    double wheel_load[4];
    float cull_radius;            // every part of a car that a ray can see lies within this distance of the car's origin
    int32_t puck_in_box;          // the LiDAR puck's disc lies inside the chassis box by a margin far above binary32 rounding: a ray then meets the box no later than
                                  // the puck, min(box, puck) is the box's time to the bit, and ray_vs_car() leaves the circle test (a square root) out
    // binary32 of the constants the inter-vehicle ray test uses (the conversions the specification makes per test, made once)
    float box_xmin_f, box_xmax_f, box_ymin_f, box_ymax_f, lidar_x_f, lidar_y_f, ring_radius_f, pad_f;
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
// (from the hardware's lane counters, not from threadIdx.x: the thread index then need not be kept in a register -- or in scratch
// memory -- for the whole life of a persistent kernel)
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ double shfl_xor_f64(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, FTGP_WAVE);
    hi = __shfl_xor(hi, m, FTGP_WAVE);
    return __hiloint2double(hi, lo);
}

// wave-uniform values pinned in SGPRs (values read from LDS arrive in VGPRs even when every lane reads the same address)
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float sgpr(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// wave-uniform copy of a pointer that was read from LDS (keeps it in SGPRs: global_load with an SGPR base)
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

// Orders this wave's LDS stores before its later LDS loads (lanes exchange data through LDS without a
// workgroup barrier: DS operations of one wave execute in order, the fence keeps the compiler honest).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
