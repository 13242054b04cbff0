// ftgp_api.hip -- implementation of the C-ABI of include/ftgp.h on top of the HIP kernels.
// Host code only owns resources and launches; there is no CPU compute path (no fallback).
#include <dlfcn.h>
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include <hip/hip_ext.h>
#include "ftgp_kernels.hip"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* a = "")
{
    snprintf(g_err, sizeof g_err, fmt, a);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            snprintf(g_err, sizeof g_err, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return FTGP_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)

// ---- RCCL, loaded lazily so that single-GPU use never touches it ------------------------------
struct Id128 { char internal[128]; };   // == ncclUniqueId (rccl.h:43)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclFloat64 = 8;          // ncclFloat64 / ncclDouble (rccl.h ncclDataType_t)

Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.lib) return 0;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(FTGP_ERR_COMM, "cannot load librccl.so: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy)
        return fail(FTGP_ERR_COMM, "librccl.so lacks an expected symbol%s");
    g_rccl.lib = h;
    return 0;
}

}  // namespace

struct FtgpEnv {
    DeviceParams P{};
    FtgpConfig cfg{};
    int device = 0;
    hipStream_t stream = nullptr, side = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop[2] = { nullptr, nullptr }, ev_metrics = nullptr, ev_gather = nullptr;      // ev_stop: one per metrics slot
    bool timed = false;
    bool ext_launch = true;      // FTGP_LAUNCH_PLAIN=1 switches it off (tools/launch_host.sh)
    bool last_roster = false;    // the newest launch ran the ROSTER instantiation
    // device buffers
    uint16_t* d_field = nullptr; uint32_t* d_bits = nullptr; uint32_t* d_nearbits = nullptr;
    double* d_path = nullptr; double* d_spawn = nullptr; float* d_ray = nullptr; float* d_cover = nullptr; void* d_veh = nullptr; unsigned char* d_stage = nullptr; DeviceParams* d_params = nullptr;
    CarState* d_cars = nullptr; float* d_ranges = nullptr; int64_t* d_steps = nullptr;
    uint8_t* d_env_mask = nullptr; uint8_t* d_car_mask = nullptr; double* d_ctrl = nullptr; double* d_pose = nullptr;
    double* d_metrics = nullptr; double* d_gather = nullptr; double* d_wg_metrics = nullptr; unsigned int* d_wg_ticket = nullptr;
    double* d_edt = nullptr; double* d_fan = nullptr;      // FTGP_LIDAR_FAKELIDAR: distance transform, binary64 fan
    // This rank's metrics record lives in two slots (device memory for RCCL, pinned host memory for the caller) that successive
    // launches alternate between, so that the exchange of launch k's record can run beside launch k + 1.
    int cur_slot = 0;                    // slot of the most recent step launch (or of the record ftgp_metrics_kernel refreshed)
    bool launch_metrics_valid = false;   // slot cur_slot of d_metrics / h_metrics still describes the state (no reset / set_pose / ... since)
    double* h_metrics = nullptr;      // pinned [2][FTGP_METRIC_DOUBLES]: THIS rank's records only (the gathered ones land in h_gather)
    double* h_metrics_dev = nullptr;  // the same buffer as the device sees it: the step kernel's epilogue writes straight into it
    double* h_gather = nullptr;       // pinned [world][FTGP_METRIC_DOUBLES]: landing buffer of the all-gather
    double* h_wg_metrics = nullptr;   // pinned [2][workgroups][FTGP_METRIC_DOUBLES]: the workgroups' partial records of a launch (one rank, no communicator:
                                      // nothing on the device needs the launch's record, the host adds the partial records up -- collect_slot())
    int n_blocks = 0;                 // workgroups of a step launch
    bool slot_partial[2] = { false, false };   // the record of that slot's launch is in h_wg_metrics (partial records), not in h_metrics
    bool gather_open = false;         // ftgp_metrics_allgather_begin without its _end
    int gather_slot = 0;              // the slot that exchange reads
    hipEvent_t gather_event = nullptr;   // what its _end waits for: ev_gather (side stream / metrics kernel) or the slot's own ev_stop
    bool gather_held = false;         // one rank: the record was copied to `held` because a later launch was about to reuse its slot
    double held[FTGP_METRIC_DOUBLES] = { 0 };
    int32_t* d_prog = nullptr; double* d_core = nullptr;
    std::vector<int32_t> h_prog; std::vector<double> h_core;
    bool rows_valid = false;          // h_prog / h_core mirror the device state (cleared by every call that changes it)
    bool multi = false;
    // comm
    void* comm = nullptr; int rank = 0, world = 1;
};

namespace {

constexpr int kCoreDoubles = 16;     // row of ftgp_pack_kernel

// exact chessboard distance transform of a W x H occupancy image (two raster sweeps over a padded image)
void chessboard_dt(const std::vector<uint8_t>& occ, int W, int H, std::vector<int>& out)
{
    const int S = W + 2;
    std::vector<int> d((size_t)S * (H + 2), 1 << 20);
    auto at = [&](int x, int y) -> int& { return d[(size_t)(y + 1) * S + (x + 1)]; };
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) if (occ[(size_t)y * W + x]) at(x, y) = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int n = std::min(std::min(at(x - 1, y), at(x - 1, y - 1)), std::min(at(x, y - 1), at(x + 1, y - 1))) + 1;
            if (n < at(x, y)) at(x, y) = n;
        }
    for (int y = H - 1; y >= 0; --y)
        for (int x = W - 1; x >= 0; --x) {
            const int n = std::min(std::min(at(x + 1, y), at(x + 1, y + 1)), std::min(at(x, y + 1), at(x - 1, y + 1))) + 1;
            if (n < at(x, y)) at(x, y) = n;
        }
    out.resize((size_t)W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) out[(size_t)y * W + x] = at(x, y);
}

// Host-side inputs of the sector box field (the boxes themselves are searched on the GPU, ftgp_box_field_kernel) and the
// bitmaps of the wall contact.
struct HostTables {
    std::vector<uint8_t> wall;        // [H][W] 1 = wall
    std::vector<uint16_t> runx, runy; // [2][H][W] wall-free run lengths along +x / -x and +y / -y (0 on walls, 65535 = beyond the image)
    std::vector<uint32_t> bits, nearbits;   // [H][wpr], padding bits clear
};

void build_tables(const FtgpTrack& t, int reach, HostTables& g)
{
    const int W = t.width, H = t.height, wpr = t.words_per_row;
    g.wall.assign((size_t)W * H, 0);
    g.bits.assign((size_t)H * wpr, 0u); g.nearbits.assign((size_t)H * wpr, 0u);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            if ((t.bits[(size_t)y * wpr + (x >> 5)] >> (x & 31)) & 1u) {
                g.wall[(size_t)y * W + x] = 1; g.bits[(size_t)y * wpr + (x >> 5)] |= 1u << (x & 31);
            }
    std::vector<int> dpx;
    chessboard_dt(g.wall, W, H, dpx);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            if (dpx[(size_t)y * W + x] <= reach) g.nearbits[(size_t)y * wpr + (x >> 5)] |= 1u << (x & 31);
    const size_t plane = (size_t)W * H;
    g.runx.assign(2 * plane, 0); g.runy.assign(2 * plane, 0);
    for (int y = 0; y < H; ++y) {
        int r = 65535;
        for (int x = W - 1; x >= 0; --x) { r = g.wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); g.runx[(size_t)y * W + x] = (uint16_t)r; }
        r = 65535;
        for (int x = 0; x < W; ++x) { r = g.wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); g.runx[plane + (size_t)y * W + x] = (uint16_t)r; }
    }
    for (int x = 0; x < W; ++x) {
        int r = 65535;
        for (int y = H - 1; y >= 0; --y) { r = g.wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); g.runy[(size_t)y * W + x] = (uint16_t)r; }
        r = 65535;
        for (int y = 0; y < H; ++y) { r = g.wall[(size_t)y * W + x] ? 0 : std::min(65535, r + 1); g.runy[plane + (size_t)y * W + x] = (uint16_t)r; }
    }
}

inline int pad16(size_t n) { return (int)((n + 15) & ~(size_t)15); }

// nidc.py:57,93-99 exactly as the driver evaluates it (binary64, libm): points covered by a disparity whose closer sample is d
int cover_count_host(double width, double rpp, double close_dist)
{
    const double angle = 2 * atan(width / (2 * close_dist));
    const double cnt = ceil(angle / rpp);
    return (cnt > 2147483000.0) ? 2147483000 : (cnt < -2147483000.0 ? -2147483000 : (int)cnt);
}

// thr[k], k = 1 .. kmax: the largest positive binary32 sample whose count is still >= k (the count never increases with the
// sample); thr[0] = the count of a sample of exactly 0.  Found by bisection over the bit patterns of the positive floats.
void build_cover_table(double car_width, int n_rays, int kmax, float* thr)
{
    const double width = (car_width / 2) * (1 + 300.0 / 100);       // nidc.py:93
    const double rpp = (2 * M_PI) / (double)n_rays;                 // nidc.py:121
    auto num = [&](uint32_t bits) { float f; memcpy(&f, &bits, 4); return cover_count_host(width, rpp, (double)f); };
    thr[0] = (float)cover_count_host(width, rpp, 0.0);
    for (int k = 1; k <= kmax; ++k) {
        uint32_t lo = 1u, hi = 0x7f7fffffu;                         // num(lo) >= k by the choice of kmax; num(hi) may be < k
        if (num(hi) >= k) { memcpy(&thr[k], &hi, 4); continue; }
        while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (num(mid) >= k) lo = mid; else hi = mid; }
        memcpy(&thr[k], &lo, 4);
    }
}

// LDS layout for `cpb` cars and `wpb` waves per workgroup; returns the total
int lds_layout(DeviceParams& P, int cpb, int wpb)
{
    int o = 0;
    P.off_params = o; o += pad16(offsetof(DeviceParams, veh));              // the head of the block: what the step kernel reads
    P.off_veh = o;    o += pad16(sizeof(VehLds));
    P.off_path = o;   o += pad16(sizeof(double) * 2 * FTGP_PATH_POINTS);
    P.off_ray = o;    o += pad16(sizeof(float) * 2 * (size_t)P.n_rays);
    P.off_cars = o;   o += cpb * (int)sizeof(CarCore);
    P.off_frame = o;  o += 2 * cpb * (int)sizeof(LidarFrame);      // double-buffered by step parity
    if (P.cars_per_env > 1) o += 2 * cpb * FTGP_PAIR_STRIDE * (int)sizeof(PairCull);   // env-mate records, right behind the frames
    P.off_steps = o;  o += pad16((size_t)cpb * sizeof(int64_t));
    P.off_scan = o;   o += 2 * cpb * P.win_floats * (int)sizeof(float);   // double-buffered by step parity
    P.off_list = o;   o += std::min(cpb, wpb) * FTGP_WAVE * (int)sizeof(int);                 // driver scratch: wave c runs the driver of car c
    P.off_pool = o;   o += 32;
    P.off_k1 = o;     o += cpb * (FTGP_FORCE_TERMS * 24 + 4 * 8 + 104 + (P.cars_per_env > 1 ? FTGP_PAIR_STRIDE * 24 : 0));   // K1 staging: force terms | new wheel spins | new state [| contact sums per env-mate]
    P.mmask_stride = pad16((size_t)2 * (size_t)((P.n_rays + FTGP_WAVE - 1) / FTGP_WAVE + 2));      // two groups per task, never more tasks than groups of 64 rays + 2
    P.off_mmask = o;  if (P.cars_per_env > 1) o += 2 * cpb * P.mmask_stride;              // env-mate visibility masks, double-buffered by step parity
    P.stage_cover = pad16(sizeof(float) * (size_t)(P.cover_kmax + 1));
    P.off_cover = o;  o += 2 * P.stage_cover;                                               // cover-count thresholds of the launch's driver (FTGP_POLICY_PER_CAR: of both, nidc's first)
    P.lds_bytes = o;
    P.cars_per_block = cpb; P.waves_per_block = wpb;
    return o;
}

// Waiting for an event of a launch is a blocked wait.  Polling hipEventQuery instead was measured (tools/launch_host.sh, round 4): 7 us
// SLOWER per launch -- every query takes the runtime's locks and walks the stream's command list.
hipError_t wait_event(const FtgpEnv*, hipEvent_t ev) { return hipEventSynchronize(ev); }

// nidc or fast, for every car or for some car of the roster
bool uses_disparity_driver(const FtgpEnv* e, int policy)
{
    if (policy == FTGP_POLICY_NIDC || policy == FTGP_POLICY_FAST) return true;
    if (policy != FTGP_POLICY_PER_CAR) return false;
    for (int k = 0; k < e->P.cars_per_env; ++k) if (e->P.car_policy[k] == FTGP_POLICY_NIDC || e->P.car_policy[k] == FTGP_POLICY_FAST) return true;
    return false;
}

// This rank's record of the launch (or metrics kernel) that wrote `slot`, once its event has been waited for: the record itself, or the
// sum of the workgroups' partial records (sums of integers, a minimum and a maximum: exact in any order -- bit-identical to what the
// step kernel's last workgroup or ftgp_metrics_kernel compute on the device).
void collect_slot(const FtgpEnv* e, int slot, double* out)
{
    if (!e->slot_partial[slot]) { memcpy(out, e->h_metrics + (size_t)slot * FTGP_METRIC_DOUBLES, sizeof(double) * FTGP_METRIC_DOUBLES); return; }
    double v[FTGP_METRIC_DOUBLES] = { 0, 0, 0, 0, 0, 0, INFINITY, -INFINITY };
    const double* r = e->h_wg_metrics + (size_t)slot * e->n_blocks * FTGP_METRIC_DOUBLES;
    for (int b = 0; b < e->n_blocks; ++b, r += FTGP_METRIC_DOUBLES) {
        for (int q = 0; q < 6; ++q) v[q] += r[q];
        v[6] = fmin(v[6], r[6]); v[7] = fmax(v[7], r[7]);
    }
    memcpy(out, v, sizeof v);
}

int launch_steps(FtgpEnv* e, int policy, int n_steps)
{
    e->rows_valid = false;
    if (n_steps < 0) return fail(FTGP_ERR_ARG, "n_steps < 0%s");
    if (policy == FTGP_POLICY_PER_CAR && !e->P.car_policy[0]) return fail(FTGP_ERR_STATE, "FTGP_POLICY_PER_CAR without ftgp_set_car_policies%s");
    if (uses_disparity_driver(e, policy)) {
        if (e->P.n_rays < 8) return fail(FTGP_ERR_ARG, "nidc/fast need n_rays >= 8 (they drop len/8 rays from each end)%s");
        if (e->P.n_rays - 2 * e->P.eighth > FTGP_WAVE * FTGP_WAVE) return fail(FTGP_ERR_ARG, "the device drivers handle at most 4096 samples in the front window%s");
    }
    HIP_TRY(hipSetDevice(e->device));
    const int cpb = e->P.cars_per_block;
    const int blocks = (e->P.n_cars + cpb - 1) / cpb;
    const int slot = e->cur_slot ^ 1;
    // an exchange that is still reading this launch's slot (begin without end, two launches ago) goes first: over RCCL on the device (the
    // side stream's event); with one rank the "exchange" is the record in pinned memory, which is put aside before the slot is reused
    if (n_steps > 0 && e->gather_open && e->gather_slot == slot) {
        if (e->comm) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_gather, 0));
        else if (!e->gather_held) {
            HIP_TRY(wait_event(e, e->gather_event));
            collect_slot(e, slot, e->held);
            e->gather_held = true;
        }
    }
    // The launch's two events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL): no marker packet before and after it,
    // and their difference is the kernel's time alone.  FTGP_LAUNCH_PLAIN=1: hipEventRecord on either side instead.
    const bool ext = e->ext_launch && n_steps > 0;
    if (!ext) HIP_TRY(hipEventRecord(e->ev_start, e->stream));
    if (n_steps > 0) {
        const dim3 grid(blocks), block(e->P.waves_per_block * FTGP_WAVE);
        const uint32_t lds = (uint32_t)e->P.lds_bytes;
        const bool fake = e->P.lidar_mode == FTGP_LIDAR_FAKELIDAR;
        hipEvent_t ev0 = ext ? e->ev_start : nullptr, ev1 = ext ? e->ev_stop[slot] : nullptr;
        const bool roster = policy == FTGP_POLICY_PER_CAR;
        e->last_roster = roster;
        // one rank and no communicator: the workgroups' partial records go straight to pinned host memory (bit 1 of the slot argument)
        const bool partial = e->h_wg_metrics != nullptr && e->comm == nullptr && e->d_wg_metrics != nullptr;
        const int slot_arg = slot | (partial ? 2 : 0);
        e->slot_partial[slot] = partial;
#define FTGP_LAUNCH(M, F, R) hipExtLaunchKernelGGL((ftgp_step_kernel<M, F, R>), grid, block, lds, e->stream, ev0, ev1, 0, e->d_params, policy, n_steps, slot_arg)
        if (e->multi) { if (fake) FTGP_LAUNCH(true, true, true); else if (roster) FTGP_LAUNCH(true, false, true); else FTGP_LAUNCH(true, false, false); }
        else          { if (fake) FTGP_LAUNCH(false, true, true); else if (roster) FTGP_LAUNCH(false, false, true); else FTGP_LAUNCH(false, false, false); }
#undef FTGP_LAUNCH
        HIP_TRY(hipGetLastError());
        e->cur_slot = slot;
        e->launch_metrics_valid = e->d_wg_metrics != nullptr;
    }
    if (!ext) HIP_TRY(hipEventRecord(e->ev_stop[e->cur_slot], e->stream));
    e->timed = true;
    return 0;
}

// packed read-back rows (one small kernel + two small copies instead of the whole state records)
int sync_rows_to_host(FtgpEnv* e)
{
    if (e->rows_valid) return 0;      // a host-driver step reads snapshot, progress and lap times: one pack, not three
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->P.n_cars;
    e->h_prog.resize(n * FTGP_PROGRESS_INTS); e->h_core.resize(n * kCoreDoubles);
    hipLaunchKernelGGL(ftgp_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, e->P, e->d_prog, e->d_core);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(e->h_prog.data(), e->d_prog, sizeof(int32_t) * e->h_prog.size(), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_core.data(), e->d_core, sizeof(double) * e->h_core.size(), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->rows_valid = true;
    return 0;
}

}  // namespace

extern "C" {

void ftgp_default_vehicle(FtgpVehicle* v)
{
    memset(v, 0, sizeof *v);
    // masses: chassis 3.542137 (mushr.em.xml:119) + 4 x 0.498952 (:69) + steering-wheel geom 0.01 (:122)
    // + LiDAR puck (cylinder r 0.03, half-height 0.015, default density 1000; :108) + softeners 4e-5 (:66)
    v->mass = 5.632768;
    v->izz = 0.0316994;              // yaw inertia of those parts about the body origin (chassis from the STL volume)
    const double s = 0.5;            // mushr_scale (:23)
    v->wheel_x[0] = s * 0.1385;  v->wheel_y[0] = s * 0.115;     // fl (:124)
    v->wheel_x[1] = s * 0.1385;  v->wheel_y[1] = s * -0.115;    // fr (:137)
    v->wheel_x[2] = s * -0.158;  v->wheel_y[2] = s * 0.115;     // bl (:150)
    v->wheel_x[3] = s * -0.158;  v->wheel_y[3] = s * -0.115;    // br (:162)
    v->wheel_radius = 0.03;
    v->wheel_inertia = 0.01 + 0.498952 / 5.0 * (0.03 * 0.03 + 0.03 * 0.03);   // armature + ellipsoid about its axle
    v->wheel_damping = 0.01;
    v->throttle_kv = 100.0; v->throttle_gear = 0.04; v->throttle_force_limit = 500.0;
    v->steer_kp = 20.0; v->steer_damping = 0.3;
    v->steer_inertia = 3 * 0.0002 + 2 * (0.498952 / 5.0 * (0.03 * 0.03 + 0.01 * 0.01)) + 0.01 / 5.0 * (0.03 * 0.03 + 0.01 * 0.01);
    v->steer_limit = 1.0;
    v->friction = 0.5; v->gravity = 9.81;
    v->tire_damping = (v->mass / 4.0) * (2.0 / (0.95 * 0.02));                 // solref 0.02, solimp dmax 0.95 (:69)
    v->contact_x[0] = 0.0385; v->contact_x[1] = 0.0; v->contact_x[2] = -0.0385;
    v->contact_radius = 0.0655;
    v->contact_stiffness = v->mass / (0.95 * 0.95 * 0.02 * 0.02);
    v->contact_damping = v->mass * (2.0 / (0.95 * 0.02));
    v->lidar_x = -0.0525; v->lidar_y = 0.0; v->lidar_ring_radius = 0.03;       // (:101-103)
    v->body_z = 0.0156;
    v->box_xmin = -0.1027; v->box_xmax = 0.1034; v->box_ymin = -0.0461; v->box_ymax = 0.0472;   // STL bbox x 0.5
    v->softener_radius = 0.65 * 0.0488;                                        // mushr_wheel.stl radius x (mushr_scale * 1.3) (:39,65-67)
}


void ftgp_tricycle_vehicle(FtgpVehicle* v)
{
    memset(v, 0, sizeof *v);
    v->kind = FTGP_VEHICLE_TRICYCLE;
    // masses: chassis mesh = convex hull of its 9 vertices at scale (0.01, 0.006, 0.0015), default density 1000 (car.em.xml:52,66): 0.4158
    // + LiDAR puck (density 2000, r 0.03, half-height 0.015; :78) 0.1696 + three wheels of 0.5 / 3 (:86,96,108,119)
    v->mass = 1.085446;
    v->izz = 0.005886;               // of those parts about the body origin
    v->wheel_x[0] = -0.07; v->wheel_y[0] = 0.06;      // left driven wheel (:97)
    v->wheel_x[1] = -0.07; v->wheel_y[1] = -0.06;     // right driven wheel (:110)
    v->wheel_x[2] = 0.08;  v->wheel_y[2] = 0.0;       // front caster: condim 1, frictionless (:96) -- carries load, transmits no force
    v->wheel_radius = 0.03;                           // cylinder size 0.03 0.01 (:24)
    v->wheel_inertia = 0.5 * (0.5 / 3.0) * 0.03 * 0.03;   // solid cylinder about its axle
    v->wheel_damping = 0.03;                          // default joint damping (:22)
    v->motor_forward_limit = 4.0; v->motor_turn_limit = 1.0;   // ctrlrange (:138-139)
    v->friction = 1.0; v->gravity = 9.81;             // MuJoCo default friction of wheel and plane
    v->tire_damping = (v->mass * (0.08 / 0.15) / 2.0) * (2.0 / (0.95 * 0.02));   // the load share of one driven wheel; solimp dmax 0.95 (:24), solref 0.02
    v->contact_x[0] = 0.045; v->contact_x[1] = 0.0; v->contact_x[2] = -0.045;    // chassis footprint 0.2 x 0.12 as three circles
    v->contact_radius = 0.06;
    v->contact_stiffness = v->mass / (0.95 * 0.95 * 0.02 * 0.02);
    v->contact_damping = v->mass * (2.0 / (0.95 * 0.02));
    v->lidar_x = -0.0525; v->lidar_y = 0.0; v->lidar_ring_radius = 0.03;         // (:72-76)
    v->body_z = 0.04;
    v->box_xmin = -0.10; v->box_xmax = 0.10; v->box_ymin = -0.06; v->box_ymax = 0.06;   // mesh bbox
    v->softener_radius = 0.035;                       // softener spheres (:93,104,116)
    v->steer_limit = 1.0; v->steer_inertia = 1.0;     // unused (no steering joint)
}

const char* ftgp_last_error(void) { return g_err; }

int ftgp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ftgp_destroy(FtgpEnv* e)
{
    if (!e) return 0;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->side) (void)hipStreamSynchronize(e->side);
    if (e->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm);
    void* bufs[] = { e->d_field, e->d_bits, e->d_nearbits, e->d_cover, e->d_stage, e->d_params, e->d_veh, e->d_path, e->d_spawn, e->d_ray, e->d_cars, e->d_ranges,
                     e->d_steps, e->d_env_mask, e->d_car_mask, e->d_ctrl, e->d_pose, e->d_metrics, e->d_gather, e->d_prog, e->d_core, e->d_wg_metrics, e->d_wg_ticket,
                     e->d_edt, e->d_fan };
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (e->h_metrics) (void)hipHostFree(e->h_metrics);
    if (e->h_gather) (void)hipHostFree(e->h_gather);
    if (e->h_wg_metrics) (void)hipHostFree(e->h_wg_metrics);
    if (e->ev_gather) (void)hipEventDestroy(e->ev_gather);
    if (e->ev_start) (void)hipEventDestroy(e->ev_start);
    for (hipEvent_t ev : e->ev_stop) if (ev) (void)hipEventDestroy(ev);
    if (e->ev_metrics) (void)hipEventDestroy(e->ev_metrics);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->side) (void)hipStreamDestroy(e->side);
    delete e;
    return 0;
}

int ftgp_create(const FtgpConfig* cfg, FtgpEnv** out)
{
    if (!cfg || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    *out = nullptr;
    if (cfg->abi_version != FTGP_ABI_VERSION) return fail(FTGP_ERR_ARG, "abi version mismatch%s");
    if (cfg->n_envs < 1 || cfg->cars_per_env < 1 || cfg->cars_per_env > 8 || cfg->n_rays < 1)
        return fail(FTGP_ERR_ARG, "bad n_envs / cars_per_env / n_rays%s");
    if (cfg->spawn_mode == 0 && (cfg->cars_per_env + 4) * 2 + 1 >= FTGP_PATH_POINTS)
        return fail(FTGP_ERR_ARG, "too many cars for the reference spawn rule%s");
    const FtgpTrack& t = cfg->track;
    if (t.width < 1 || t.height < 1 || !t.bits || !t.path || t.words_per_row < (t.width + 31) / 32)
        return fail(FTGP_ERR_ARG, "bad track%s");
    if (t.width > 8192 || t.height > 8192) return fail(FTGP_ERR_ARG, "images above 8192 pixels are not supported%s");
    // Direction sectors of the box field: more slope slices mean fewer march iterations and a larger field.  A large batch is bound by
    // throughput and by what of the field its cars keep in the 4-MiB L2s (16 sectors: 32 bytes per pixel); a small one by the latency of
    // its longest rays (64 sectors); 16384 cars (config 5) do best with 8.  Measured: profiles/round4/ab_sectors.log.  Results do not depend on the choice.
    const long cars_total = (long)cfg->n_envs * cfg->cars_per_env;
    int n_sectors = cars_total >= 8192 ? 8 : cars_total >= 2048 ? 16 : 64;
    if (const char* sv = getenv("FTGP_SECTORS_RT")) { const int c = atoi(sv); if (c == 8 || c == 16 || c == 32 || c == 64) n_sectors = c; }
    const int n_planes = n_sectors;
    // the march addresses the field with a 32-bit byte offset
    if ((uint64_t)ftgp_plane256(t.width, t.height) * 256u * (uint64_t)n_planes > 0xFFFFFFFFull)
        return fail(FTGP_ERR_ARG, "track image too large: the sector box field (2 bytes per pixel and direction sector) must stay below 4 GiB%s");
    if (cfg->env_base < 0) return fail(FTGP_ERR_ARG, "env_base < 0%s");
    if (cfg->lidar_mode != FTGP_LIDAR_RANGEFINDER && cfg->lidar_mode != FTGP_LIDAR_FAKELIDAR) return fail(FTGP_ERR_ARG, "unknown lidar_mode%s");
    if (!(cfg->dt > 0.0) || !(t.px_size_x > 0.0) || !(t.px_size_y > 0.0)) return fail(FTGP_ERR_ARG, "bad dt / pixel size%s");
    const FtgpVehicle& v = cfg->vehicle;
    if (!(v.contact_radius > 0.0) || !(v.mass > 0.0) || !(v.izz > 0.0) || (v.kind != FTGP_VEHICLE_MUSHR && v.kind != FTGP_VEHICLE_TRICYCLE))
        return fail(FTGP_ERR_ARG, "bad vehicle%s");
    if (cfg->bubble_wrap && !(v.softener_radius > 0.0)) return fail(FTGP_ERR_ARG, "bubble_wrap needs vehicle.softener_radius > 0%s");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(FTGP_ERR_NO_DEVICE, "no HIP device: this library has no CPU fallback%s");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(FTGP_ERR_ARG, "device_id out of range%s");

    FtgpEnv* e = new FtgpEnv();
    e->ext_launch = !getenv("FTGP_LAUNCH_PLAIN");
    e->cfg = *cfg;
    e->device = cfg->device_id;
#define CREATE_TRY(expr)                                                                           \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            snprintf(g_err, sizeof g_err, "%s failed: %s", #expr, hipGetErrorString(e_));          \
            ftgp_destroy(e);                                                                       \
            return FTGP_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)
    CREATE_TRY(hipSetDevice(e->device));
    // FTGP_WAIT_SPIN=1: the host waits for a launch by spinning instead of blocking on the interrupt (hipDeviceScheduleSpin: a CPU core per
    // waiting handle for a shorter wake-up; measured: tools/launch_host.sh).  Best effort: a device that is already active keeps its flags.
    if (const char* sv = getenv("FTGP_WAIT_SPIN")) { if (atoi(sv) == 1) (void)hipSetDeviceFlags(hipDeviceScheduleSpin); (void)hipGetLastError(); }
    CREATE_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreate(&e->ev_start));
    CREATE_TRY(hipEventCreate(&e->ev_stop[0]));
    CREATE_TRY(hipEventCreate(&e->ev_stop[1]));
    CREATE_TRY(hipEventCreateWithFlags(&e->ev_metrics, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&e->ev_gather, hipEventDisableTiming));

    DeviceParams& P = e->P;
    P.n_envs = cfg->n_envs; P.cars_per_env = cfg->cars_per_env; P.n_cars = cfg->n_envs * cfg->cars_per_env;
    P.n_rays = cfg->n_rays; P.lap_target = cfg->lap_target; P.spawn_mode = cfg->spawn_mode; P.env_base = cfg->env_base;
    P.ranges_stride = (cfg->n_rays + 31) & ~31;      // rows start on 128-B boundaries
    P.seed = cfg->seed; P.dt = cfg->dt;
    P.rpp = (2 * M_PI) / (double)cfg->n_rays;
    P.two_over_rpp = (float)(2.0 / P.rpp);
    P.bubble_wrap = cfg->bubble_wrap ? 1 : 0;        // cfg->naive_flatten: accepted, no effect on a planar model (custom.py:1338-1339)
    P.lidar_mode = cfg->lidar_mode;
    P.map_size = cfg->map_size > 0.0 ? cfg->map_size : 40.0;                     // 20 * scale, custom.py:1155,1382
    P.width = t.width; P.height = t.height; P.words_per_row = t.words_per_row; P.fstride = t.width + 2;
    P.plane256 = ftgp_plane256(t.width, t.height);
    // a ray's sector is always found among all FTGP_SECTORS; the table says which plane serves it
    P.n_sectors = n_sectors; P.slice_factor = FTGP_SLICE_FACTOR(FTGP_SLOPE_SLICES);
    P.n_planes = ftgp_sector_table(P.sector_tab, n_sectors, t.width + 2, P.plane256);
    P.px_size_x = t.px_size_x; P.px_size_y = t.px_size_y; P.origin_x = t.origin_x; P.origin_y = t.origin_y;
    P.inv_px_x = 1.0 / t.px_size_x; P.inv_px_y = 1.0 / t.px_size_y;
    P.inv_px_x_f = (float)P.inv_px_x; P.inv_px_y_f = (float)P.inv_px_y;
    P.veh = cfg->vehicle;
    {   // static wheel loads from the wheelbase split
        const double wtot = v.mass * v.gravity;
        if (v.kind == FTGP_VEHICLE_TRICYCLE) {       // two driven wheels behind the origin, the caster (wheel 2) in front
            const double a_f = v.wheel_x[2], a_r = -0.5 * (v.wheel_x[0] + v.wheel_x[1]);
            P.wheel_load[0] = P.wheel_load[1] = 0.5 * (wtot * (a_f / (a_f + a_r)));
            P.wheel_load[2] = wtot * (a_r / (a_f + a_r)); P.wheel_load[3] = 0.0;
        } else {
            const double a_f = 0.5 * (v.wheel_x[0] + v.wheel_x[1]), a_r = -0.5 * (v.wheel_x[2] + v.wheel_x[3]);
            P.wheel_load[0] = P.wheel_load[1] = 0.5 * (wtot * (a_r / (a_f + a_r)));
            P.wheel_load[2] = P.wheel_load[3] = 0.5 * (wtot * (a_f / (a_f + a_r)));
        }
    }
    e->multi = cfg->cars_per_env > 1;
    {   // chessboard reach of the largest wall-contact window
        const double rmax = std::max(v.contact_radius, cfg->bubble_wrap ? v.softener_radius : 0.0);
        P.contact_reach = std::max((int)ceil(rmax * P.inv_px_x), (int)ceil(rmax * P.inv_px_y));
    }
    P.eighth = (int)((double)cfg->n_rays / 8.0);                    // nidc.py:18
    {   // the largest cover count any positive sample can produce (that of the smallest positive float), over both drivers
        const double rpp = (2 * M_PI) / (double)cfg->n_rays;
        const float tiny = 1.401298464e-45f;
        P.cover_kmax = std::max(1, std::max(cover_count_host(0.24, rpp, (double)tiny), cover_count_host(0.12, rpp, (double)tiny)));
    }
    P.win_floats = ((P.eighth & 3) + (cfg->n_rays - 2 * P.eighth) + 1 + 3) & ~3;       // window at float (eighth % 4), ranges[0] in the last float
    P.snap_eps = ftgp_snap_eps(t.width, t.height);
    { hipDeviceProp_t prop; CREATE_TRY(hipGetDeviceProperties(&prop, e->device)); P.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }
    P.edge_margin = (float)(v.lidar_ring_radius * std::max(P.inv_px_x, P.inv_px_y) * 1.001 + 2.0);
    if ((cfg->n_rays + FTGP_WAVE - 1) / FTGP_WAVE > FTGP_MAX_GROUPS) { ftgp_destroy(e); return fail(FTGP_ERR_ARG, "n_rays above 16384 is not supported%s"); }

    // workgroup shape: whole envs, at most 16 cars (K1 / K3 run on the lanes of one wave), two workgroups per CU
    // (<= 80 KiB of LDS each) so that 8 waves per SIMD hide the latency of the field loads
    {
        const int unit = cfg->cars_per_env;
        int wpb = 16;
        if (const char* sv = getenv("FTGP_WAVES_PER_BLOCK")) { const int c = atoi(sv); if (c >= 1 && c <= 16) wpb = c; }
        int want = (FTGP_MAX_CARS_PER_BLOCK / unit) * unit;
        // small batches: fewer cars per workgroup so that every CU gets work.  Up to four envs per CU a batch runs best as ONE workgroup per CU
        // (its step is the driver -> dynamics latency chain plus one sweep task per wave: a second workgroup on the CU only competes for issue
        // slots -- 1024 envs: 9.1 us per step with 256 workgroups of 4, 9.8 with 512 of 2; 512 envs: 8.8 / 9.0); larger batches take two
        // workgroups per CU (1536 envs: 10.7 us with 512 workgroups of 3, 13.9 with 256 of 6) -- profiles/round5/config2_shapes.log
        const int n_units = P.n_cars / unit;
        const int n_cu = P.n_cu > 0 ? P.n_cu : 256;
        const int per_cu = (n_units + n_cu - 1) / n_cu;
        const int spread = (per_cu <= 4 ? std::max(1, per_cu) : std::max(1, n_units / (2 * n_cu))) * unit;
        int cpb = std::min(want, spread);
        if (const char* sv = getenv("FTGP_CARS_PER_BLOCK")) { const int c = atoi(sv); if (c >= unit && c <= FTGP_MAX_CARS_PER_BLOCK) cpb = (c / unit) * unit; }
        int lds_cap = 80 * 1024;
        if (const char* sv = getenv("FTGP_LDS_CAP_KB")) { const int c = atoi(sv); if (c >= 16 && c <= 160) lds_cap = c * 1024; }
        while (cpb > unit && lds_layout(P, cpb, wpb) > lds_cap) cpb -= unit;
        if (lds_layout(P, cpb, wpb) > 160 * 1024) {
            snprintf(g_err, sizeof g_err, "one env of %d car(s) with a %d-ray scan does not fit the 160 KiB LDS", unit, P.n_rays);
            ftgp_destroy(e);
            return FTGP_ERR_ARG;
        }
        if (getenv("FTGP_VERBOSE"))
            fprintf(stderr, "ftgp_create: %d cars x %d waves per workgroup, %d B of LDS (cap %d)\n", cpb, wpb, lds_layout(P, cpb, wpb), lds_cap);
    }
#define FTGP_BIG_LDS(M, F, R) CREATE_TRY(hipFuncSetAttribute((const void*)ftgp_step_kernel<M, F, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    FTGP_BIG_LDS(false, false, false); FTGP_BIG_LDS(true, false, false); FTGP_BIG_LDS(false, false, true); FTGP_BIG_LDS(true, false, true);
    FTGP_BIG_LDS(false, true, true); FTGP_BIG_LDS(true, true, true);
#undef FTGP_BIG_LDS

    // host-side tables
    HostTables tab;
    build_tables(t, P.contact_reach, tab);
    std::vector<float> ray(2 * (size_t)cfg->n_rays + 4, 0.0f);
    std::vector<double> fan(2 * (size_t)cfg->n_rays, 0.0);
    for (int j = 0; j < cfg->n_rays; ++j) {
        // mushr.em.xml:112-117: phi_j = radians(360/R*j - 90); the ray (+z of the site) is (sin phi, -cos phi, 0)
        const double phi = ((360.0 / (double)cfg->n_rays) * (double)j - 90.0) * (M_PI / 180.0);
        fan[2 * (size_t)j] = cfg->fan_dirs ? cfg->fan_dirs[2 * (size_t)j] : sin(phi);
        fan[2 * (size_t)j + 1] = cfg->fan_dirs ? cfg->fan_dirs[2 * (size_t)j + 1] : -cos(phi);
        ray[2 * (size_t)j] = (float)fan[2 * (size_t)j]; ray[2 * (size_t)j + 1] = (float)fan[2 * (size_t)j + 1];
        // The rangefinders' own fan is point-symmetric: site j + n/2 looks exactly opposite to site j.  The BINARY32 table says so to the last
        // bit (its second half is the negated first half -- the roundings of libm's sin / cos of phi + pi need not be), which is what lets the
        // sweep derive a ray from its opposite; a caller's fan_dirs is taken as it comes.  The binary64 fan of FAKELIDAR mode is libm's value
        // for every site, as include/ftgp.h says for fan_dirs == NULL (round 4 negated it too: a last-bit difference from the documented fan).
        if (!cfg->fan_dirs && cfg->n_rays % 2 == 0 && j >= cfg->n_rays / 2) {
            ray[2 * (size_t)j] = -ray[2 * (size_t)(j - cfg->n_rays / 2)]; ray[2 * (size_t)j + 1] = -ray[2 * (size_t)(j - cfg->n_rays / 2) + 1];
        }
    }
    {   // the sweep's work list (lidar_groups): draw g -> (kidx = g / cars_per_block, car slot = g % cars_per_block), task = group_order[kidx]
        const int R = cfg->n_rays, halfR = R / 2;
        bool sym = R % 2 == 0 && !getenv("FTGP_NO_PAIRS");
        for (int j = 0; sym && j < halfR; ++j)
            sym = ray[2 * (size_t)(j + halfR)] == -ray[2 * (size_t)j] && ray[2 * (size_t)(j + halfR) + 1] == -ray[2 * (size_t)j + 1] &&
                  std::signbit(ray[2 * (size_t)(j + halfR)]) != std::signbit(ray[2 * (size_t)j]) && std::signbit(ray[2 * (size_t)(j + halfR) + 1]) != std::signbit(ray[2 * (size_t)j + 1]);
        std::vector<int> tasks;
        if (!sym) for (int j0 = 0; j0 < R; j0 += FTGP_WAVE) tasks.push_back(j0);
        else {
            int j0 = 0;
            for (; j0 + FTGP_WAVE <= halfR; j0 += FTGP_WAVE) tasks.push_back(j0 | (1 << 16));
            if (j0 < halfR) tasks.push_back(j0 | ((halfR - j0 <= FTGP_WAVE / 2 ? 2 : 1) << 16));
        }
        P.tasks_per_car = (int)tasks.size();
        // expected march length of a task ~ how far its rays look along the car's axis: |cos| of the angle between the group's middle ray
        // and the axis (ray 0 looks backwards, ray n/2 ahead); ties keep index order
        std::vector<std::pair<double, int>> key;
        for (int t : tasks) {
            const double mid = std::min((double)R - 1.0, (double)(t & 0xffff) + 31.5);
            key.push_back({ getenv("FTGP_GROUP_ORDER_PLAIN") ? 0.0 : -fabs(cos(2.0 * M_PI * mid / (double)R)), t });
        }
        std::stable_sort(key.begin(), key.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first < b.first; });
        // the cheapest pairs -- the last tasks a sweep draws -- go out as two single groups each: the waves then end a sweep within ONE short
        // group of each other, not within two (the set-up shared inside a pair is worth less than that at the very end)
        int tail = 2;
        if (const char* sv = getenv("FTGP_PAIR_TAIL")) tail = atoi(sv);
        std::vector<int> order;
        for (size_t k = 0; k < key.size(); ++k) {
            const int t = key[k].second;
            if ((t >> 16) == 1 && (int)(key.size() - k) <= tail) { order.push_back(t & 0xffff); order.push_back((t & 0xffff) + halfR); }
            else order.push_back(t);
        }
        P.tasks_per_car = (int)order.size();
        for (size_t k = 0; k < order.size(); ++k) P.group_order[k] = order[k];
        // mate_masks(): a ray of a group lies within 32 spacings of the rangefinders' uniform fan of the group's middle ray; + 1.2 degrees for the
        // slack in the rays' own test (acos 0.9999 = 0.81 degrees) and rounding.  A caller's fan has no such bound: every mate is looked at.
        const double gamma = 32.0 * (2.0 * M_PI / (double)R) + 0.021;
        if (cfg->fan_dirs || gamma >= 1.5) { P.group_cg = -2.0f; P.group_sg = 0.0f; }
        else { P.group_cg = (float)cos(gamma); P.group_sg = (float)sin(gamma); }
        if (getenv("FTGP_VERBOSE")) fprintf(stderr, "ftgp_create: %d sweep tasks per car (%s)\n", P.tasks_per_car, sym ? "pairs of opposite ray groups" : "single groups");
    }
    std::vector<double> spawn(4 * FTGP_PATH_POINTS);
    for (int p = 0; p < FTGP_PATH_POINTS; ++p) {
        // position_vehicles (custom.py:1240-1245) + euler_to_quaternion([angle, 0, 0]) (custom.py:81-87)
        const int p1 = (p + 1) % FTGP_PATH_POINTS;
        const double ang = atan2(t.path[2 * p1 + 1] - t.path[2 * p + 1], t.path[2 * p1] - t.path[2 * p]);
        spawn[4 * p] = t.path[2 * p]; spawn[4 * p + 1] = t.path[2 * p + 1];
        spawn[4 * p + 2] = cos(ang / 2); spawn[4 * p + 3] = sin(ang / 2);
    }

    if (cfg->lidar_mode == FTGP_LIDAR_FAKELIDAR) {
        // The distance transform of custom.py:1149-1153 / raycast.py:24-27 (scipy.ndimage.distance_transform_edt of the non-wall
        // mask) without scipy, exact: the squared distance is the minimum over the columns x' of (x - x')^2 + g(x', y)^2 with g the
        // vertical distance to the nearest wall of column x' -- the run lengths above -- in integers; one correctly rounded sqrt.
        const size_t plane = (size_t)t.width * t.height;
        uint16_t* d_runy = nullptr;
        CREATE_TRY(hipMalloc(&d_runy, 2 * plane * sizeof(uint16_t)));
        CREATE_TRY(hipMalloc(&e->d_edt, plane * sizeof(double)));
        CREATE_TRY(hipMalloc(&e->d_fan, fan.size() * sizeof(double)));
        CREATE_TRY(hipMemcpy(d_runy, tab.runy.data(), 2 * plane * sizeof(uint16_t), hipMemcpyHostToDevice));
        CREATE_TRY(hipMemcpy(e->d_fan, fan.data(), fan.size() * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ftgp_edt_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, e->stream, d_runy, t.width, t.height, e->d_edt);
        CREATE_TRY(hipGetLastError());
        CREATE_TRY(hipStreamSynchronize(e->stream));
        (void)hipFree(d_runy);
        P.edt = e->d_edt; P.fan_dirs = e->d_fan;
    } else {   // sector box field: upload the run lengths, search the boxes on the device
        const size_t plane = (size_t)t.width * t.height;
        const size_t plane_cells = (size_t)P.plane256 * 128;
        const size_t cells = plane_cells * (size_t)P.n_planes;
        uint16_t* d_runx = nullptr; uint16_t* d_runy = nullptr;
        CREATE_TRY(hipMalloc(&e->d_field, cells * sizeof(uint16_t)));
        CREATE_TRY(hipMalloc(&d_runx, 2 * plane * sizeof(uint16_t)));
        CREATE_TRY(hipMalloc(&d_runy, 2 * plane * sizeof(uint16_t)));
        CREATE_TRY(hipMemcpy(d_runx, tab.runx.data(), 2 * plane * sizeof(uint16_t), hipMemcpyHostToDevice));
        CREATE_TRY(hipMemcpy(d_runy, tab.runy.data(), 2 * plane * sizeof(uint16_t), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ftgp_box_field_kernel, dim3((unsigned)((plane_cells * (size_t)P.n_sectors + 255) / 256)), dim3(256), 0, e->stream, d_runx, d_runy, t.width, t.height, P.n_sectors, e->d_field);
        CREATE_TRY(hipGetLastError());
        CREATE_TRY(hipStreamSynchronize(e->stream));
        (void)hipFree(d_runx); (void)hipFree(d_runy);
        P.field = e->d_field;
    }
    {
        std::vector<unsigned char> vimg((size_t)pad16(sizeof(VehLds)), 0);
        VehLds vl; memset(&vl, 0, sizeof vl);
        vl.v = P.veh; for (int i = 0; i < 4; ++i) vl.wheel_load[i] = P.wheel_load[i];
        // every part of a car that a ray can see (chassis box, LiDAR puck) lies within rmax of the car's origin; 10 % margin
        const double cx = std::max(fabs(v.box_xmin), fabs(v.box_xmax)), cy = std::max(fabs(v.box_ymin), fabs(v.box_ymax));
        const double rmax = std::max(sqrt(cx * cx + cy * cy), sqrt(v.lidar_x * v.lidar_x + v.lidar_y * v.lidar_y) + v.lidar_ring_radius);
        vl.cull_radius = (float)(1.1 * rmax);
        {   // the puck inside the box with at least 1e-3 to spare on every side (MuSHR: 0.016, tricycle: 0.0175): coordinates in a mate's frame are below the
            // map's 40 units, so binary32 rounding of the two tests is below 1e-5 -- the circle can never come out ahead of the box
            const double m = 1e-3, r = v.lidar_ring_radius;
            vl.puck_in_box = (v.lidar_x - r >= v.box_xmin + m && v.lidar_x + r <= v.box_xmax - m && v.lidar_y - r >= v.box_ymin + m && v.lidar_y + r <= v.box_ymax - m &&
                              !getenv("FTGP_PUCK_TEST")) ? 1 : 0;
        }
        vl.box_xmin_f = (float)v.box_xmin; vl.box_xmax_f = (float)v.box_xmax; vl.box_ymin_f = (float)v.box_ymin; vl.box_ymax_f = (float)v.box_ymax;
        vl.lidar_x_f = (float)v.lidar_x; vl.lidar_y_f = (float)v.lidar_y; vl.ring_radius_f = (float)v.lidar_ring_radius;
        memcpy(vimg.data(), &vl, sizeof vl);
        CREATE_TRY(hipMalloc(&e->d_veh, vimg.size()));
        CREATE_TRY(hipMemcpy(e->d_veh, vimg.data(), vimg.size(), hipMemcpyHostToDevice));
        P.veh_dev = e->d_veh;
    }
    const size_t sz_bits = sizeof(uint32_t) * (size_t)t.height * t.words_per_row;
    const size_t sz_path = (size_t)pad16(sizeof(double) * 2 * FTGP_PATH_POINTS), sz_ray = (size_t)pad16(sizeof(float) * 2 * (size_t)cfg->n_rays);
    const size_t n_cars = (size_t)P.n_cars;
    CREATE_TRY(hipMalloc(&e->d_bits, sz_bits));
    CREATE_TRY(hipMalloc(&e->d_nearbits, sz_bits));
    CREATE_TRY(hipMalloc(&e->d_path, sz_path));
    CREATE_TRY(hipMalloc(&e->d_ray, sz_ray));
    CREATE_TRY(hipMalloc(&e->d_spawn, sizeof(double) * 4 * FTGP_PATH_POINTS));
    {   // cover-count thresholds: nidc (car_width 0.12, nidc.py:5) then fast (0.06, fast.py:4), each padded to the staged size
        const size_t stride = (size_t)P.cover_kmax + 1, padded = (size_t)pad16(sizeof(float) * stride) / sizeof(float);
        std::vector<float> thr(stride + padded + 4, 0.0f);
        build_cover_table(0.12, cfg->n_rays, P.cover_kmax, thr.data());
        build_cover_table(0.06, cfg->n_rays, P.cover_kmax, thr.data() + stride);
        CREATE_TRY(hipMalloc(&e->d_cover, sizeof(float) * thr.size()));
        CREATE_TRY(hipMemcpy(e->d_cover, thr.data(), sizeof(float) * thr.size(), hipMemcpyHostToDevice));
        P.cover_thr = e->d_cover;
    }
    CREATE_TRY(hipMalloc(&e->d_cars, sizeof(CarState) * n_cars));
    CREATE_TRY(hipMalloc(&e->d_ranges, sizeof(float) * n_cars * P.ranges_stride));
    CREATE_TRY(hipMalloc(&e->d_steps, sizeof(int64_t) * (size_t)P.n_envs));
    CREATE_TRY(hipMalloc(&e->d_env_mask, (size_t)P.n_envs));
    CREATE_TRY(hipMalloc(&e->d_car_mask, n_cars));
    CREATE_TRY(hipMalloc(&e->d_ctrl, sizeof(double) * 2 * n_cars));
    CREATE_TRY(hipMalloc(&e->d_pose, sizeof(double) * FTGP_POSE_DOUBLES * n_cars));
    CREATE_TRY(hipMalloc(&e->d_metrics, sizeof(double) * FTGP_METRIC_DOUBLES * 2));
    CREATE_TRY(hipHostMalloc(&e->h_metrics, sizeof(double) * FTGP_METRIC_DOUBLES * 2, hipHostMallocMapped));
    CREATE_TRY(hipHostGetDevicePointer((void**)&e->h_metrics_dev, e->h_metrics, 0));
    if (!getenv("FTGP_NO_FUSED_METRICS")) {          // (diagnostic switch: tests compare the fused record with ftgp_metrics_kernel's)
        const size_t blocks = (n_cars + (size_t)P.cars_per_block - 1) / (size_t)P.cars_per_block;
        CREATE_TRY(hipMalloc(&e->d_wg_metrics, sizeof(double) * FTGP_METRIC_DOUBLES * blocks));
        CREATE_TRY(hipMalloc(&e->d_wg_ticket, sizeof(unsigned int)));
        CREATE_TRY(hipMemsetAsync(e->d_wg_ticket, 0, sizeof(unsigned int), e->stream));
        P.wg_metrics = e->d_wg_metrics; P.wg_ticket = e->d_wg_ticket; P.metrics_dev = e->d_metrics; P.metrics_host = e->h_metrics_dev;
        e->n_blocks = (int)blocks;
        if (!getenv("FTGP_NO_HOST_SUM")) {           // (diagnostic switch: the device-side hand-off of the record also with one rank)
            CREATE_TRY(hipHostMalloc(&e->h_wg_metrics, sizeof(double) * FTGP_METRIC_DOUBLES * 2 * blocks, hipHostMallocMapped));
            CREATE_TRY(hipHostGetDevicePointer((void**)&P.wg_metrics_host, e->h_wg_metrics, 0));
        }
    }
    CREATE_TRY(hipMalloc(&e->d_prog, sizeof(int32_t) * FTGP_PROGRESS_INTS * n_cars));
    CREATE_TRY(hipMalloc(&e->d_core, sizeof(double) * kCoreDoubles * n_cars));
    CREATE_TRY(hipMemcpy(e->d_bits, tab.bits.data(), sz_bits, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(e->d_nearbits, tab.nearbits.data(), sz_bits, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemsetAsync(e->d_path, 0, sz_path, e->stream));
    CREATE_TRY(hipStreamSynchronize(e->stream));
    CREATE_TRY(hipMemcpy(e->d_path, t.path, sizeof(double) * 2 * FTGP_PATH_POINTS, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(e->d_spawn, spawn.data(), sizeof(double) * spawn.size(), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemsetAsync(e->d_ray, 0, sz_ray, e->stream));
    CREATE_TRY(hipStreamSynchronize(e->stream));
    CREATE_TRY(hipMemcpy(e->d_ray, ray.data(), sizeof(float) * 2 * (size_t)cfg->n_rays, hipMemcpyHostToDevice));
    // on the handle's own stream: a non-blocking stream is not ordered against the null stream, and ftgp_reset() below runs on it
    CREATE_TRY(hipMemsetAsync(e->d_cars, 0, sizeof(CarState) * n_cars, e->stream));
    CREATE_TRY(hipMemsetAsync(e->d_ranges, 0, sizeof(float) * n_cars * P.ranges_stride, e->stream));
    CREATE_TRY(hipMemsetAsync(e->d_steps, 0, sizeof(int64_t) * (size_t)P.n_envs, e->stream));
    P.bits = e->d_bits; P.nearbits = e->d_nearbits; P.path = e->d_path; P.spawn = e->d_spawn;
    P.ray_dir = e->d_ray; P.cars = e->d_cars; P.ranges = e->d_ranges; P.steps = e->d_steps;
    {   // the staging image: the LDS bytes [off_params, off_cars) as every workgroup wants them, then both drivers' cover tables
        const size_t head = (size_t)(P.off_cars - P.off_params), cover = (size_t)P.stage_cover;
        std::vector<unsigned char> simg(head + 2 * cover, 0);
        CREATE_TRY(hipMalloc(&e->d_stage, simg.size()));
        P.stage_img = e->d_stage;
        memcpy(simg.data() + (P.off_params - P.off_params), &P, offsetof(DeviceParams, veh));
        CREATE_TRY(hipMemcpy(simg.data() + (P.off_veh - P.off_params), e->d_veh, (size_t)pad16(sizeof(VehLds)), hipMemcpyDeviceToHost));
        memcpy(simg.data() + (P.off_path - P.off_params), t.path, sizeof(double) * 2 * FTGP_PATH_POINTS);
        memcpy(simg.data() + (P.off_ray - P.off_params), ray.data(), sizeof(float) * 2 * (size_t)cfg->n_rays);
        const size_t stride = (size_t)P.cover_kmax + 1;
        CREATE_TRY(hipMemcpy(simg.data() + head, e->d_cover, std::min(cover, sizeof(float) * stride), hipMemcpyDeviceToHost));
        CREATE_TRY(hipMemcpy(simg.data() + head + cover, e->d_cover + stride, std::min(cover, sizeof(float) * stride), hipMemcpyDeviceToHost));
        CREATE_TRY(hipMemcpy(e->d_stage, simg.data(), simg.size(), hipMemcpyHostToDevice));
    }
    {   // device image of the parameter block, and behind it the sweep's task table (DeviceParams::task_tab): draw g is task g / cars_per_block of
        // car slot g % cars_per_block, with everything the draw and the delivery need precomputed
        const size_t head = (size_t)pad16(sizeof(DeviceParams));
        const int cpb = P.cars_per_block, ntasks = cpb * P.tasks_per_car, R = P.n_rays, halfR = R / 2;
        if (P.tasks_per_car > 256 || R > 0x4000) { ftgp_destroy(e); return fail(FTGP_ERR_ARG, "internal: the task table's fields are too narrow for this fan%s"); }
        std::vector<int32_t> tt(2 * 4 * (size_t)ntasks, 0);
        auto wclass = [&](int first, int lim) {          // rays first .. min(first + 63, lim - 1) against the window [eighth, R - eighth)
            const int last = std::min(first + FTGP_WAVE, lim) - 1, lo = P.eighth, hi = R - P.eighth;
            if (last < lo || first >= hi || lo >= hi) return 0;
            return (first >= lo && last < hi) ? 1 : 2;
        };
        for (int g = 0; g < ntasks; ++g) {
            const int kidx = g / cpb, c = g % cpb, ent = P.group_order[kidx], j0 = ent & 0xffff, kind = ent >> 16;
            const int w0 = kind == 2 ? 2 : wclass(j0, kind == 1 ? halfR : R), w1 = kind == 1 ? wclass(j0 + halfR, R) : 0;
            const uint32_t plain = (uint32_t)j0 | (uint32_t)kind << 14 | (uint32_t)c << 16;
            int32_t* a = &tt[4 * (size_t)g];
            int32_t* b = &tt[4 * (size_t)(ntasks + g)];
            a[0] = (int32_t)(plain | (uint32_t)w0 << 20 | (uint32_t)w1 << 22 | (uint32_t)(j0 == 0 ? 1 : 0) << 24);
            b[0] = (int32_t)plain;
            a[1] = b[1] = c * (int)sizeof(LidarFrame) | kidx << 16;
            a[2] = b[2] = c * P.ranges_stride * 4;
            a[3] = b[3] = 4 * (c * P.win_floats + (P.eighth & 3) - P.eighth);
        }
        std::vector<unsigned char> pimg(head + sizeof(int32_t) * tt.size() + 16, 0);
        CREATE_TRY(hipMalloc(&e->d_params, pimg.size()));
        P.task_tab = reinterpret_cast<const int32_t*>(reinterpret_cast<unsigned char*>(e->d_params) + head);
        memcpy(pimg.data(), &P, sizeof(DeviceParams));
        memcpy(pimg.data() + head, tt.data(), sizeof(int32_t) * tt.size());
        CREATE_TRY(hipMemcpy(e->d_params, pimg.data(), pimg.size(), hipMemcpyHostToDevice));
    }
#undef CREATE_TRY
    int rc = ftgp_reset(e, nullptr);
    if (rc != 0) { ftgp_destroy(e); return rc; }
    *out = e;
    return 0;
}

int ftgp_reset(FtgpEnv* e, const uint8_t* mask)
{
    if (!e) return fail(FTGP_ERR_ARG, "null handle%s");
    e->rows_valid = false; e->launch_metrics_valid = false;
    HIP_TRY(hipSetDevice(e->device));
    const uint8_t* dmask = nullptr;
    if (mask) {
        HIP_TRY(hipMemcpyAsync(e->d_env_mask, mask, (size_t)e->P.n_envs, hipMemcpyHostToDevice, e->stream));
        dmask = e->d_env_mask;
    }
    hipLaunchKernelGGL(ftgp_reset_kernel, dim3((e->P.n_cars + 63) / 64), dim3(64), 0, e->stream, e->P, dmask);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ftgp_zero_ranges_kernel, dim3(e->P.n_cars), dim3(256), 0, e->stream, e->P, dmask);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

int ftgp_set_ctrl(FtgpEnv* e, const double* ctrl, const uint8_t* car_mask)
{
    if (!e || !ctrl) return fail(FTGP_ERR_ARG, "null argument%s");
    e->rows_valid = false;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(e->d_ctrl, ctrl, sizeof(double) * 2 * (size_t)e->P.n_cars, hipMemcpyHostToDevice, e->stream));
    const uint8_t* dmask = nullptr;
    if (car_mask) {
        HIP_TRY(hipMemcpyAsync(e->d_car_mask, car_mask, (size_t)e->P.n_cars, hipMemcpyHostToDevice, e->stream));
        dmask = e->d_car_mask;
    }
    hipLaunchKernelGGL(ftgp_set_ctrl_kernel, dim3((e->P.n_cars + 255) / 256), dim3(256), 0, e->stream, e->P, e->d_ctrl, dmask);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));   // the caller's buffers may be reused as soon as we return
    return 0;
}

int ftgp_step(FtgpEnv* e, int n_steps)
{
    if (!e) return fail(FTGP_ERR_ARG, "null handle%s");
    return launch_steps(e, FTGP_POLICY_HOST, n_steps);
}

int ftgp_rollout(FtgpEnv* e, int policy, int n_steps)
{
    if (!e) return fail(FTGP_ERR_ARG, "null handle%s");
    if (policy < FTGP_POLICY_HOST || policy > FTGP_POLICY_PER_CAR) return fail(FTGP_ERR_ARG, "unknown policy%s");
    return launch_steps(e, policy, n_steps);
}

int ftgp_set_car_policies(FtgpEnv* e, const int32_t* policies)
{
    if (!e || !policies) return fail(FTGP_ERR_ARG, "null argument%s");
    for (int k = 0; k < e->P.cars_per_env; ++k)
        if (policies[k] < FTGP_POLICY_LOBOTOMY || policies[k] > FTGP_POLICY_RANDOM) return fail(FTGP_ERR_ARG, "set_car_policies: lobotomy / nidc / fast / random only%s");
    // a workgroup holds whole envs, so its car slot c runs the roster's entry c % cars_per_env
    for (int c = 0; c < FTGP_MAX_CARS_PER_BLOCK; ++c) e->P.car_policy[c] = policies[c % e->P.cars_per_env];
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));           // no launch is reading the block while it changes
    HIP_TRY(hipMemcpy(reinterpret_cast<unsigned char*>(e->d_params) + offsetof(DeviceParams, car_policy), e->P.car_policy, sizeof e->P.car_policy, hipMemcpyHostToDevice));
    return 0;
}

int ftgp_get_lidar(FtgpEnv* e, float* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpy2DAsync(out, sizeof(float) * (size_t)e->P.n_rays, e->d_ranges, sizeof(float) * (size_t)e->P.ranges_stride,
                             sizeof(float) * (size_t)e->P.n_rays, (size_t)e->P.n_cars, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

int ftgp_get_snapshot(FtgpEnv* e, double* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    if (int rc = sync_rows_to_host(e)) return rc;
    for (int i = 0; i < e->P.n_cars; ++i) {
        const double* a = e->h_core.data() + (size_t)i * kCoreDoubles;
        double* o = out + (size_t)i * FTGP_SNAPSHOT_DOUBLES;
        // quaternion_to_euler(w, 0, 0, z), custom.py:62-76
        const double w = a[2], x = 0.0, y = 0.0, z = a[3];
        const double roll = atan2(+2.0 * (w * x + y * z), +1.0 - 2.0 * (x * x + y * y));
        double t2 = +2.0 * (w * y - z * x);
        t2 = t2 > +1.0 ? +1.0 : t2; t2 = t2 < -1.0 ? -1.0 : t2;
        const double pitch = asin(t2);
        const double yaw = atan2(+2.0 * (w * z + x * y), +1.0 - 2.0 * (y * y + z * z));
        o[0] = a[9]; o[1] = a[4]; o[2] = a[5]; o[3] = 0.0; o[4] = yaw; o[5] = pitch; o[6] = roll;
        o[7] = a[10]; o[8] = a[11];
        o[9] = a[12] / e->P.dt;   // time = steps / timestep, custom.py:1397 (sic)
    }
    return 0;
}

int ftgp_get_pose(FtgpEnv* e, double* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    if (int rc = sync_rows_to_host(e)) return rc;
    for (int i = 0; i < e->P.n_cars; ++i) {
        const double* a = e->h_core.data() + (size_t)i * kCoreDoubles;
        double* o = out + (size_t)i * FTGP_POSE_DOUBLES;
        o[0] = a[0]; o[1] = a[1]; o[2] = e->P.veh.body_z; o[3] = a[2]; o[4] = 0; o[5] = 0; o[6] = a[3];
        o[7] = a[4]; o[8] = a[5]; o[9] = 0; o[10] = 0; o[11] = 0; o[12] = a[6];
    }
    return 0;
}

int ftgp_set_pose(FtgpEnv* e, const double* pose)
{
    if (!e || !pose) return fail(FTGP_ERR_ARG, "null argument%s");
    e->rows_valid = false; e->launch_metrics_valid = false;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(e->d_pose, pose, sizeof(double) * FTGP_POSE_DOUBLES * (size_t)e->P.n_cars, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(ftgp_set_pose_kernel, dim3((e->P.n_cars + 255) / 256), dim3(256), 0, e->stream, e->P, e->d_pose);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

int ftgp_policy_eval(FtgpEnv* e, int policy, const float* ranges, double* ctrl_out)
{
    if (!e || !ranges) return fail(FTGP_ERR_ARG, "null argument%s");
    e->rows_valid = false;
    if (policy < FTGP_POLICY_LOBOTOMY || policy > FTGP_POLICY_PER_CAR) return fail(FTGP_ERR_ARG, "policy_eval: device policies only%s");
    if (policy == FTGP_POLICY_PER_CAR && !e->P.car_policy[0]) return fail(FTGP_ERR_STATE, "FTGP_POLICY_PER_CAR without ftgp_set_car_policies%s");
    if (uses_disparity_driver(e, policy)) {
        if (e->P.n_rays < 8) return fail(FTGP_ERR_ARG, "nidc/fast need n_rays >= 8%s");
        if (e->P.n_rays - 2 * e->P.eighth > FTGP_WAVE * FTGP_WAVE) return fail(FTGP_ERR_ARG, "the device drivers handle at most 4096 samples in the front window%s");
    }
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpy2DAsync(e->d_ranges, sizeof(float) * (size_t)e->P.ranges_stride, ranges, sizeof(float) * (size_t)e->P.n_rays,
                             sizeof(float) * (size_t)e->P.n_rays, (size_t)e->P.n_cars, hipMemcpyHostToDevice, e->stream));
    const size_t lds = 4 * ((size_t)e->P.win_floats * sizeof(float) + sizeof(CarCore) + FTGP_WAVE * sizeof(int));
    if (lds > 64 * 1024) return fail(FTGP_ERR_ARG, "scan does not fit LDS%s");
    hipLaunchKernelGGL(ftgp_policy_kernel, dim3((e->P.n_cars + 3) / 4), dim3(256), lds, e->stream, e->P, policy, ctrl_out ? e->d_ctrl : nullptr);
    HIP_TRY(hipGetLastError());
    if (ctrl_out) HIP_TRY(hipMemcpyAsync(ctrl_out, e->d_ctrl, sizeof(double) * 2 * (size_t)e->P.n_cars, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

int ftgp_eval_progress(FtgpEnv* e)
{
    if (!e) return fail(FTGP_ERR_ARG, "null handle%s");
    e->rows_valid = false; e->launch_metrics_valid = false;
    HIP_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(ftgp_progress_kernel, dim3((e->P.n_cars + 63) / 64), dim3(64), 0, e->stream, e->P);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ftgp_get_progress(FtgpEnv* e, int32_t* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    if (int rc = sync_rows_to_host(e)) return rc;
    memcpy(out, e->h_prog.data(), sizeof(int32_t) * e->h_prog.size());
    return 0;
}

int ftgp_get_winners(FtgpEnv* e, int32_t* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    if (int rc = sync_rows_to_host(e)) return rc;
    // places in the order cars reached lap_target; cars that got there in the same step rank in car order, the order of the
    // reference's per-car loop (custom.py:1337,1367-1369)
    const int cpe = e->P.cars_per_env;
    for (int env = 0; env < e->P.n_envs; ++env) {
        const int32_t* p = e->h_prog.data() + (size_t)env * cpe * FTGP_PROGRESS_INTS;
        for (int i = 0; i < cpe; ++i) {
            int place = 0;
            if (p[i * FTGP_PROGRESS_INTS + 4]) {
                place = 1;
                auto fin64 = [&](int k) { int64_t v; memcpy(&v, e->h_core.data() + ((size_t)env * cpe + k) * kCoreDoubles + 15, sizeof v); return v; };      // all 64 bits
                const int64_t mine = fin64(i);
                for (int k = 0; k < cpe; ++k)
                    if (k != i && p[k * FTGP_PROGRESS_INTS + 4]) {
                        const int64_t theirs = fin64(k);
                        if (theirs < mine || (theirs == mine && k < i)) ++place;
                    }
            }
            out[(size_t)env * cpe + i] = place;
        }
    }
    return 0;
}

int ftgp_get_lap_times(FtgpEnv* e, int32_t* counts, double* times)
{
    if (!e || !counts || !times) return fail(FTGP_ERR_ARG, "null argument%s");
    if (int rc = sync_rows_to_host(e)) return rc;
    for (int i = 0; i < e->P.n_cars; ++i) counts[i] = (int32_t)e->h_core[(size_t)i * kCoreDoubles + 13];
    HIP_TRY(hipMemcpy2DAsync(times, sizeof(double) * FTGP_MAX_LAP_TIMES, reinterpret_cast<const char*>(e->d_cars) + sizeof(CarCore), sizeof(CarState),   // times[] follows the CarCore head
                            
                             sizeof(double) * FTGP_MAX_LAP_TIMES, (size_t)e->P.n_cars, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

int ftgp_get_race_steps(FtgpEnv* e, int64_t* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    if (int rc = sync_rows_to_host(e)) return rc;
    for (int i = 0; i < e->P.n_cars; ++i) {          // the pack kernel ships the two int64 as bit patterns in the row's last two doubles
        memcpy(out + 2 * (size_t)i, e->h_core.data() + (size_t)i * kCoreDoubles + 14, sizeof(int64_t));
        memcpy(out + 2 * (size_t)i + 1, e->h_core.data() + (size_t)i * kCoreDoubles + 15, sizeof(int64_t));
    }
    return 0;
}

int ftgp_get_ctrl(FtgpEnv* e, double* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    if (int rc = sync_rows_to_host(e)) return rc;
    for (int i = 0; i < e->P.n_cars; ++i) { out[2 * i] = e->h_core[(size_t)i * kCoreDoubles + 7]; out[2 * i + 1] = e->h_core[(size_t)i * kCoreDoubles + 8]; }
    return 0;
}

int ftgp_get_steps(FtgpEnv* e, int64_t* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(out, e->d_steps, sizeof(int64_t) * (size_t)e->P.n_envs, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

// the record of this GPU: the step kernel's last workgroup has written it into pinned memory (slot cur_slot), or ftgp_metrics_kernel does now
static int metrics_to_host(FtgpEnv* e, double* out)
{
    if (!e->launch_metrics_valid) {
        // An exchange that was begun on this very slot and not ended yet (one rank: its "exchange" IS the slot in pinned memory) promised the
        // record of the state at its begin: put that aside before the slot is refreshed with the present state's.
        if (e->gather_open && e->gather_slot == e->cur_slot && !e->comm && !e->gather_held) {
            HIP_TRY(wait_event(e, e->gather_event));
            collect_slot(e, e->cur_slot, e->held);
            e->gather_held = true;
        }
        hipLaunchKernelGGL(ftgp_metrics_kernel, dim3(1), dim3(FTGP_METRIC_THREADS), 0, e->stream, e->P, e->h_metrics_dev + (size_t)e->cur_slot * FTGP_METRIC_DOUBLES);
        HIP_TRY(hipGetLastError());
        e->slot_partial[e->cur_slot] = false;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    collect_slot(e, e->cur_slot, out);
    return 0;
}

int ftgp_metrics_local(FtgpEnv* e, double* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    HIP_TRY(hipSetDevice(e->device));
    return metrics_to_host(e, out);
}

int ftgp_comm_unique_id(uint8_t id_out[128])
{
    if (!id_out) return fail(FTGP_ERR_ARG, "null argument%s");
    if (int rc = load_rccl()) return rc;
    Id128 id;
    int r = g_rccl.GetUniqueId(&id);
    if (r != 0) return fail(FTGP_ERR_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    memcpy(id_out, id.internal, 128);
    return 0;
}

int ftgp_comm_init(FtgpEnv* e, const uint8_t id[128], int rank, int world_size)
{
    if (!e || !id || world_size < 1 || rank < 0 || rank >= world_size) return fail(FTGP_ERR_ARG, "bad comm arguments%s");
    if (e->comm) return fail(FTGP_ERR_STATE, "the handle already has a communicator%s");
    if (int rc = load_rccl()) return rc;
    HIP_TRY(hipSetDevice(e->device));
    Id128 uid; memcpy(uid.internal, id, 128);
    // the gathered records get buffers of their own (this rank's record slots, h_metrics, written by the step kernel, stay untouched) -- allocated
    // BEFORE the communicator: a handle never holds a communicator without them
    double* d_gather = nullptr; double* h_gather = nullptr;
    if (hipMalloc(&d_gather, sizeof(double) * FTGP_METRIC_DOUBLES * (size_t)world_size) != hipSuccess ||
        hipHostMalloc(&h_gather, sizeof(double) * FTGP_METRIC_DOUBLES * (size_t)world_size, hipHostMallocDefault) != hipSuccess) {
        if (d_gather) (void)hipFree(d_gather);
        return fail(FTGP_ERR_HIP, "ftgp_comm_init: no memory for the gathered records%s");
    }
    void* comm = nullptr;
    int r = g_rccl.CommInitRank(&comm, world_size, uid, rank);
    if (r != 0) {
        (void)hipFree(d_gather); (void)hipHostFree(h_gather);
        return fail(FTGP_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    }
    // launches from here on leave their record on the device (no partial records to the host); one that is still in flight finishes first
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->comm = comm; e->d_gather = d_gather; e->h_gather = h_gather;
    e->rank = rank; e->world = world_size;
    return 0;
}

int ftgp_metrics_allgather_begin(FtgpEnv* e)
{
    if (!e) return fail(FTGP_ERR_ARG, "null handle%s");
    if (e->gather_open) return fail(FTGP_ERR_STATE, "ftgp_metrics_allgather_begin: the previous exchange has not been ended%s");
    HIP_TRY(hipSetDevice(e->device));
    const int slot = e->cur_slot;
    const size_t so = (size_t)slot * FTGP_METRIC_DOUBLES;
    const bool rccl = e->comm != nullptr;            // a one-rank communicator goes through RCCL too (that is how one GPU tests the path)
    // a communicator that arrived after a launch whose record went to the host as partial records: the device has no copy of that record
    const bool refresh = !e->launch_metrics_valid || (rccl && e->slot_partial[slot]);
    if (refresh) {                       // otherwise the slot already holds this state's record (step kernel epilogue), on the device and in pinned memory
        hipLaunchKernelGGL(ftgp_metrics_kernel, dim3(1), dim3(FTGP_METRIC_THREADS), 0, e->stream, e->P, rccl ? e->d_metrics + so : e->h_metrics_dev + so);
        HIP_TRY(hipGetLastError());
        e->slot_partial[slot] = false;
    }
    e->gather_held = false;
    if (!rccl) {                         // one rank: the "exchange" is the record's arrival in pinned memory
        if (!refresh && e->timed) e->gather_event = e->ev_stop[slot];      // ... with the launch that wrote it: nothing to enqueue
        else { HIP_TRY(hipEventRecord(e->ev_gather, e->stream)); e->gather_event = e->ev_gather; }
    } else {
        // the record is produced on the compute stream; everything else happens on the side stream, beside the next launch
        // (the launch's own stop event -- it rides on the kernel's dispatch packet -- says when the record is there.  An event of its own, recorded behind the
        // launch, is a barrier packet with a system-scope fence between this launch and the next: measured with a one-rank communicator, the next launch then runs
        // 20 - 25 us longer -- it finds the L2s flushed -- whatever the exchange itself does: profiles/round5/exchange_overlap_one_rank.log)
        hipEvent_t ready = (!refresh && e->timed) ? e->ev_stop[slot] : e->ev_metrics;
        if (ready == e->ev_metrics) HIP_TRY(hipEventRecord(e->ev_metrics, e->stream));
        HIP_TRY(hipStreamWaitEvent(e->side, ready, 0));
        int r = g_rccl.AllGather(e->d_metrics + so, e->d_gather, FTGP_METRIC_DOUBLES, kNcclFloat64, e->comm, e->side);
        if (r != 0) return fail(FTGP_ERR_COMM, "ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
        HIP_TRY(hipMemcpyAsync(e->h_gather, e->d_gather, sizeof(double) * FTGP_METRIC_DOUBLES * (size_t)e->world, hipMemcpyDeviceToHost, e->side));
        HIP_TRY(hipEventRecord(e->ev_gather, e->side));
        e->gather_event = e->ev_gather;
    }
    e->gather_open = true; e->gather_slot = slot;
    return 0;
}

int ftgp_metrics_allgather_end(FtgpEnv* e, double* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    if (!e->gather_open) return fail(FTGP_ERR_STATE, "ftgp_metrics_allgather_end without _begin%s");
    HIP_TRY(hipSetDevice(e->device));
    if (!e->gather_held) HIP_TRY(wait_event(e, e->gather_event));      // this exchange only: a later launch on the compute stream is not waited for
    e->gather_open = false;
    if (e->comm) memcpy(out, e->h_gather, sizeof(double) * FTGP_METRIC_DOUBLES * (size_t)e->world);
    else if (e->gather_held) memcpy(out, e->held, sizeof e->held);
    else collect_slot(e, e->gather_slot, out);
    return 0;
}

int ftgp_metrics_allgather(FtgpEnv* e, double* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    if (int rc = ftgp_metrics_allgather_begin(e)) return rc;
    return ftgp_metrics_allgather_end(e, out);
}

int ftgp_get_distance_field(FtgpEnv* e, double* out)
{
    if (!e || !out) return fail(FTGP_ERR_ARG, "null argument%s");
    if (!e->d_edt) return fail(FTGP_ERR_STATE, "no distance field: the handle was not created with lidar_mode = FTGP_LIDAR_FAKELIDAR%s");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpy(out, e->d_edt, sizeof(double) * (size_t)e->P.width * e->P.height, hipMemcpyDeviceToHost));
    return 0;
}

int ftgp_fakelidar(int device_id, const double* dt, int H, int W, int n_origins, const double* origins, int rangefinders,
                   const double* cosines, const double* sines, double eps, double* scan, double* points)
{
    if (!dt || !origins || !cosines || !sines || !scan || !points || H < 1 || W < 1 || n_origins < 1 || rangefinders < 1)
        return fail(FTGP_ERR_ARG, "fakelidar: bad argument%s");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FTGP_ERR_NO_DEVICE, "no HIP device: this library has no CPU fallback%s");
    if (device_id < 0 || device_id >= ndev) return fail(FTGP_ERR_ARG, "device_id out of range%s");
    HIP_TRY(hipSetDevice(device_id));
    const size_t n = (size_t)n_origins * rangefinders, ndt = (size_t)H * W;
    double *d_dt = nullptr, *d_o = nullptr, *d_c = nullptr, *d_s = nullptr, *d_scan = nullptr, *d_pts = nullptr; int* d_err = nullptr;
    int rc = 0, herr = 0;
    auto cleanup = [&]() { void* b[] = { d_dt, d_o, d_c, d_s, d_scan, d_pts, d_err }; for (void* p : b) if (p) (void)hipFree(p); };
#define FL_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(g_err, sizeof g_err, "%s failed: %s", #expr, hipGetErrorString(e_)); cleanup(); return FTGP_ERR_HIP; } } while (0)
    FL_TRY(hipMalloc(&d_dt, ndt * 8)); FL_TRY(hipMalloc(&d_o, (size_t)n_origins * 16)); FL_TRY(hipMalloc(&d_c, n * 8)); FL_TRY(hipMalloc(&d_s, n * 8));
    FL_TRY(hipMalloc(&d_scan, n * 8)); FL_TRY(hipMalloc(&d_pts, n * 16)); FL_TRY(hipMalloc(&d_err, 4));
    FL_TRY(hipMemcpy(d_dt, dt, ndt * 8, hipMemcpyHostToDevice)); FL_TRY(hipMemcpy(d_o, origins, (size_t)n_origins * 16, hipMemcpyHostToDevice));
    FL_TRY(hipMemcpy(d_c, cosines, n * 8, hipMemcpyHostToDevice)); FL_TRY(hipMemcpy(d_s, sines, n * 8, hipMemcpyHostToDevice));
    FL_TRY(hipMemset(d_err, 0, 4));
    hipLaunchKernelGGL(ftgp_fakelidar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_dt, H, W, (int)n, rangefinders, d_o, d_c, d_s, eps, d_scan, d_pts, d_err);
    FL_TRY(hipGetLastError());
    FL_TRY(hipMemcpy(scan, d_scan, n * 8, hipMemcpyDeviceToHost)); FL_TRY(hipMemcpy(points, d_pts, n * 16, hipMemcpyDeviceToHost));
    FL_TRY(hipMemcpy(&herr, d_err, 4, hipMemcpyDeviceToHost));
#undef FL_TRY
    cleanup();
    if (herr) rc = fail(FTGP_ERR_ARG, "fakelidar: IndexError (a ray left the image through the right or bottom edge)%s");
    return rc;
}

int ftgp_last_kernel_ms(FtgpEnv* e, float* ms)
{
    if (!e || !ms) return fail(FTGP_ERR_ARG, "null argument%s");
    if (!e->timed) return fail(FTGP_ERR_STATE, "no step/rollout has been launched yet%s");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(wait_event(e, e->ev_stop[e->cur_slot]));
    HIP_TRY(hipEventElapsedTime(ms, e->ev_start, e->ev_stop[e->cur_slot]));
    return 0;
}

// diagnostic libraries only (-DFTGP_DIAG, csrc/diag/ftgp_diag.inc): read and clear the phase stamps, workgroup tables
#if defined(FTGP_DIAG) && defined(FTGP_STAMPS)
int ftgp_debug_stamps(unsigned long long* out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ftgp_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    unsigned long long z[16] = { 0 };
    return hipMemcpyToSymbol(HIP_SYMBOL(ftgp_stamps), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
#if defined(FTGP_DIAG) && defined(FTGP_STAMPS)
int ftgp_debug_substamps(unsigned long long* out)          // read and clear
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ftgp_substamps), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    unsigned long long z[32] = { 0 };
    return hipMemcpyToSymbol(HIP_SYMBOL(ftgp_substamps), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
#if defined(FTGP_DIAG) && defined(FTGP_WG_TIMES)
int ftgp_debug_set_wg_groups(const int* groups, int n_blocks)      // groups == nullptr: identity
{
    std::vector<int> g(8192, 0);
    if (groups) { for (int i = 0; i < n_blocks && i < 8191; ++i) g[i] = groups[i]; g[8191] = 1; }
    return hipMemcpyToSymbol(HIP_SYMBOL(ftgp_wg_group), g.data(), sizeof(int) * 8192) == hipSuccess ? 0 : -1;
}
int ftgp_debug_wg_times(unsigned long long* out, int n_blocks)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ftgp_wg_times), sizeof(unsigned long long) * 4 * (size_t)n_blocks) == hipSuccess ? 0 : -1;
}
#endif

// What this library was built from and with: "abi=<n> sources=<hash> diag=<switches|none> fair_shift=<n> waves_per_eu=<n>".  The hash is
// tools/evidence.py's hash of the kernel sources, handed in by the build (__graft_entry__.build: -DFTGP_BUILD_SOURCES=...); a library
// built by hand without it says "unstamped".  Touches no device.
#ifndef FTGP_BUILD_SOURCES
#define FTGP_BUILD_SOURCES unstamped
#endif
#define FTGP_STR2_(x) #x
#define FTGP_STR_(x) FTGP_STR2_(x)
const char* ftgp_build_info(void)
{
    return "abi=" FTGP_STR_(FTGP_ABI_VERSION) " sources=" FTGP_STR_(FTGP_BUILD_SOURCES) " diag=" FTGP_DIAG_FLAGS
           " fair_shift=" FTGP_STR_(FTGP_FAIR_SHIFT) " waves_per_eu=" FTGP_STR_(FTGP_WAVES_PER_EU);
}

int ftgp_selftest(int device_id, int64_t* mismatches)
{
    if (!mismatches) return fail(FTGP_ERR_ARG, "null argument%s");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FTGP_ERR_NO_DEVICE, "no HIP device: this library has no CPU fallback%s");
    if (device_id < 0 || device_id >= ndev) return fail(FTGP_ERR_ARG, "device_id out of range%s");
    HIP_TRY(hipSetDevice(device_id));
    unsigned long long* d = nullptr; unsigned long long h = 0;
    HIP_TRY(hipMalloc(&d, sizeof h));
    hipError_t e1 = hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(ftgp_selftest_rcp_kernel, dim3(1u << 12), dim3(256), 0, 0, d);
    hipError_t e2 = hipGetLastError();
    hipError_t e3 = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(FTGP_ERR_HIP, "selftest: HIP error%s");
    *mismatches = (int64_t)h;
    return 0;
}

const char* ftgp_kernel_name(FtgpEnv* e)
{
    if (!e) return "ftgp_step_kernel";
    // <MULTI, FAKE, ROSTER>: the instantiation of the newest launch (before any: the single-driver one)
    if (e->P.lidar_mode == FTGP_LIDAR_FAKELIDAR) return e->multi ? "ftgp_step_kernel<true, true, true>" : "ftgp_step_kernel<false, true, true>";
    if (e->last_roster) return e->multi ? "ftgp_step_kernel<true, false, true>" : "ftgp_step_kernel<false, false, true>";
    return e->multi ? "ftgp_step_kernel<true, false, false>" : "ftgp_step_kernel<false, false, false>";
}

}  // extern "C"
