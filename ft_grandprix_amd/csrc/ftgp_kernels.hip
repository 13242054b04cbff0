// ftgp_kernels.hip -- HIP kernels of the ft_grandprix hot path for gfx950 (CDNA4, wave64).
//
//   ftgp_step_kernel<MULTI, FAKE, ROSTER>   n_steps per launch; one workgroup = whole envs, up to 16 cars on up to 16 waves (the headline
//       shape is 8 cars on 16 waves, two workgroups per CU), persistent over all steps.  ONE workgroup barrier per step.  Inside a
//       step, concurrently:
//         K5       waves 0 .. cars-1, one car each: the driver on the car's PREVIOUS scan (the other LDS scan buffer) -> controls
//         K1 + K3  the wave whose driver delivers last: integrate + lap progress for ALL cars of the workgroup on four lanes per
//                  car (dynamics_lanes), then the LiDAR frames of the NEXT step into the other frame buffer
//         K2       every wave, as soon as its driver work is done: the sweep of THIS step in groups of 64 neighbouring rays of one car
//                  (lidar_groups): a wave draws a task -- a group, or a group and the same rays turned round --, sets the rays up,
//                  marches them until all have finished (march_all, hand-written) and delivers the 64 ranges: one 256-byte segment of
//                  the car's row in HBM, the drivers' window also into the LDS scan buffer.
//       The sweep reads the LiDAR frames, never the live state, and the scan a driver sees lags the pose by one step
//       (custom.py:1395-1425): that is what makes the overlap legal.
//       Staged into LDS once per launch, in one pass, from an image laid out as the LDS is: parameter block, vehicle constants,
//       centre-line, ray table, cover tables; then the cars' state records and the scan windows the drivers read.  The march reads the
//       sector box field from L2 (ftgp_march.h).  FAKE: K2 is the reference's own 2-D LiDAR (lidar_fake).  ROSTER: every car slot its own driver.
//       The end-of-launch metrics record is part of the kernel (launch_metrics).
//   ftgp_policy_kernel    K5 alone (ftgp_policy_eval).
//   ftgp_reset_kernel     K4 reset / spawn (+ K3 at the spawn pose), one car per lane.
//   ftgp_progress_kernel  K3 alone (after ftgp_set_pose), one car per lane.
//   ftgp_fakelidar_kernel raycast.fakelidar restated, one ray per lane.
//   ftgp_box_field_kernel the sector box field at create; ftgp_edt_kernel the distance transform of FAKELIDAR mode.
//   ftgp_metrics_kernel   per-GPU metrics record.
//
// Reference behaviour restated by each block is cited inline (paths relative to the reference repo).
// Arithmetic follows DESIGN.md operation by operation (-ffp-contract=off; fmaf only where the specification
// says "fma"), so results are bit-identical to the CPU oracle.
#include "ftgp_device.h"

// K1 staging types
struct Force { double fx, fy, tz; };
struct Dyn { double x, y, qw, qz, vx, vy, wz, qs, qsd, w[4]; };
static_assert(sizeof(Dyn) == 104 && offsetof(CarCore, w) == offsetof(Dyn, w) && offsetof(CarCore, qsd) == offsetof(Dyn, qsd), "Dyn must mirror the head of CarCore");
#define FTGP_FORCE_TERMS 11      // 4 wheels, 3 chassis circles, 4 softeners: summed in this order

struct Lds {
    const VehLds* veh;
    const double* path;
    const float2* ray;        // body-frame ray directions
    CarCore* cars;
    LidarFrame* frame;        // [2][cars_per_block], double-buffered by step parity
    PairCull* pairs;          // [2][cars_per_block][FTGP_PAIR_STRIDE], right behind the frames (multi-car envs only)
    int64_t* steps;
    float* scan;              // [2][cars_per_block][win_floats] scan windows (layout: scan_window_* below), double-buffered by step parity
    int* list;                // [waves_per_block][64] driver scratch
    int* pool;                // [2] next ray group of the sweep (lidar_groups), [2] drivers finished, [2] "some LiDAR frame is
                              // not known to lie well inside the image" (frame_write) -- all double-buffered by step parity
    Force* terms;             // [cars_per_block][FTGP_FORCE_TERMS] K1 staging: force terms in the order they are summed
    double* wnew;             // [cars_per_block][4] K1 staging: new wheel spins
    Dyn* next;                // [cars_per_block] K1 staging: new dynamic state before the commit
    Force* mates;             // [cars_per_block][FTGP_PAIR_STRIDE] K1 staging, multi-car envs: the contact forces with each env-mate, summed (car_contact_mate)
    const float* cover;       // [cover_kmax + 1] cover-count thresholds of the launch's driver (see cover_count)
    unsigned char* mmask;     // [2][cars_per_block][mmask_stride] multi-car envs: which env-mates each ray group of a car can see (mate_masks)
};

// LDS scan window of one car: the samples the on-device drivers read.  Sample ranges[eighth + i] sits at float index
// (eighth % 4) + i, so that LDS and HBM addresses of a sample agree modulo 16 bytes (whole float4 groups move with one
// ds_read_b128 + one global_store_dwordx4); ranges[0] (fast.py:135) sits in the last float of the row.
__device__ __forceinline__ int scan_window_first(int eighth) { return eighth & 3; }

// A wave-uniform value the optimiser cannot see through.  The step loop rebuilds its LDS pointers from such offsets every
// step, so loads of constants (vehicle parameters, centre-line, ...) are never hoisted out of the loop -- hoisted, they
// would stay live across the sweep and end up in scratch memory under the 64-VGPR budget.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+s"(v)); return v; }
// The parameter block as the scalar unit sees it: the kernel argument's copy in device memory, read through the constant address
// space, i.e. with scalar loads -- wave-uniform values then reach their SGPRs without a vector instruction (out of LDS each one
// costs a read and a v_readfirstlane, out of a spilled SGPR a v_readlane: the vector pipe is what this kernel is short of).
// The pointer is made opaque once per step, so the loads stay inside the step instead of being hoisted and spilled.
typedef const __attribute__((address_space(4))) DeviceParams* ScalarParams;
__device__ __forceinline__ ScalarParams scalar_view(const DeviceParams* p)
{
    uint64_t a = (uint64_t)p; asm volatile("" : "+s"(a));
    return (ScalarParams)a;
}
// the lane index, recomputed where it is used: values derived from it (per-lane addresses, selects) then stay local to the phase
// that needs them instead of being hoisted in front of the step loop and carried -- spilled -- across the sweep
__device__ __forceinline__ int lane_here() { int l = lane_id(); asm volatile("" : "+v"(l)); return l; }

struct LdsOffsets { int params, veh, path, ray, cars, frame, steps, scan, list, pool, k1, cover, mmask, cpb; };

__device__ __forceinline__ LdsOffsets lds_offsets(const DeviceParams& P)
{
    LdsOffsets o;
    o.params = sgpr(P.off_params); o.veh = sgpr(P.off_veh); o.path = sgpr(P.off_path); o.ray = sgpr(P.off_ray); o.cars = sgpr(P.off_cars);
    o.frame = sgpr(P.off_frame); o.steps = sgpr(P.off_steps); o.scan = sgpr(P.off_scan); o.list = sgpr(P.off_list); o.pool = sgpr(P.off_pool);
    o.k1 = sgpr(P.off_k1); o.cover = sgpr(P.off_cover); o.mmask = sgpr(P.off_mmask); o.cpb = sgpr(P.cars_per_block);
    return o;
}

__device__ __forceinline__ LdsOffsets lds_offsets(ScalarParams G)
{
    LdsOffsets o;
    o.params = G->off_params; o.veh = G->off_veh; o.path = G->off_path; o.ray = G->off_ray; o.cars = G->off_cars;
    o.frame = G->off_frame; o.steps = G->off_steps; o.scan = G->off_scan; o.list = G->off_list; o.pool = G->off_pool;
    o.k1 = G->off_k1; o.cover = G->off_cover; o.mmask = G->off_mmask; o.cpb = G->cars_per_block;
    return o;
}

// what the on-device drivers need to know about the scan (wave-uniform)
struct DriverShape { int n_rays, eighth, win_floats, cover_kmax; double rpp; float two_over_rpp; };
template <class PP> __device__ __forceinline__ DriverShape driver_shape(PP p)
{
    DriverShape d; d.n_rays = p->n_rays; d.eighth = p->eighth; d.win_floats = p->win_floats; d.cover_kmax = p->cover_kmax; d.rpp = p->rpp; d.two_over_rpp = p->two_over_rpp;
    return d;
}

__device__ __forceinline__ Lds lds_view(const LdsOffsets& o, unsigned char* lds)
{
    Lds L;
    L.veh = reinterpret_cast<const VehLds*>(lds + opaque(o.veh));
    L.path = reinterpret_cast<const double*>(lds + opaque(o.path));
    L.ray = reinterpret_cast<const float2*>(lds + opaque(o.ray));
    L.cars = reinterpret_cast<CarCore*>(lds + opaque(o.cars));
    L.frame = reinterpret_cast<LidarFrame*>(lds + opaque(o.frame));
    L.pairs = reinterpret_cast<PairCull*>(L.frame + 2 * o.cpb);
    L.steps = reinterpret_cast<int64_t*>(lds + opaque(o.steps));
    L.scan = reinterpret_cast<float*>(lds + opaque(o.scan));
    L.list = reinterpret_cast<int*>(lds + opaque(o.list));
    L.pool = reinterpret_cast<int*>(lds + opaque(o.pool));
    unsigned char* k1 = lds + opaque(o.k1);     // one block: terms | wnew | next (sizes follow from cars_per_block)
    L.terms = reinterpret_cast<Force*>(k1);
    L.wnew = reinterpret_cast<double*>(k1 + o.cpb * (int)(FTGP_FORCE_TERMS * sizeof(Force)));
    L.next = reinterpret_cast<Dyn*>(k1 + o.cpb * (int)(FTGP_FORCE_TERMS * sizeof(Force) + 4 * sizeof(double)));
    L.mates = reinterpret_cast<Force*>(k1 + o.cpb * (int)(FTGP_FORCE_TERMS * sizeof(Force) + 4 * sizeof(double) + sizeof(Dyn)));
    L.cover = reinterpret_cast<const float*>(lds + opaque(o.cover));
    L.mmask = lds + opaque(o.mmask);
    return L;
}

// =============================================================================================
// K2: LiDAR
// =============================================================================================
// The sweep of one step for all cars of the workgroup, executed by every wave.  Rangefinder geometry:
// template/mushr.em.xml:98-117 -- ray j leaves the ring at centre - 0.03*dir_j, dir_j = R(yaw) * (sin phi_j, -cos phi_j);
// j = 0 is the rear, CCW.  Values replace data.sensordata[vehicle_state.sensors] (custom.py:1395).
//
// Scheduling: lidar_groups() below -- groups of 64 neighbouring rays, march_all().  Which lane marches which ray has no influence on any result.
// number of set bits of `mask` below this lane
__device__ __forceinline__ int rank_below(uint64_t mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0)); }

// |1 / x| and |1 / y| correctly rounded, as the specification's IEEE divisions give them.  One Newton step on v_rcp_f32
// (1 ulp) with fused residuals is correctly rounded for every binary32 input with 2^-100 < |x| < 2^100 (checked exhaustively
// over all 2^32 bit patterns by ftgp_selftest / tests/test_gpu_parity.py); zeros, denormals, huge values and NaNs take the
// division.  The range test is done on the bit patterns of both operands at once: b - (lo + 1) is below hi - lo - 1 (unsigned)
// exactly for lo < b < hi, and the OR of two such differences can only be below that if both are -- an OR that is not sends
// the wave through the per-lane test and the division, which is then simply not needed.
// CLAMP: +inf (a zero, or so small that its reciprocal overflows) comes back as FTGP_IV_MAX, as the ray specification wants it.
template <bool CLAMP = false>
__device__ __forceinline__ void rcp_abs2(float x, float y, float& rx, float& ry)
{
    const float ax = fabsf(x), ay = fabsf(y);
    float p = __builtin_amdgcn_rcpf(ax), q = __builtin_amdgcn_rcpf(ay);
    const float ep = fmaf(-ax, p, 1.0f), eq = fmaf(-ay, q, 1.0f);
    p = fmaf(p, ep, p); q = fmaf(q, eq, q);
    const uint32_t lo1 = 0x0D800001u, span = 0x71800000u - 0x0D800001u;             // bit patterns of 2^-100 (+ 1) and 2^100
    if (__any(((__float_as_uint(ax) - lo1) | (__float_as_uint(ay) - lo1)) >= span)) {
        const bool oddx = !(ax > 0x1p-100f && ax < 0x1p100f), oddy = !(ay > 0x1p-100f && ay < 0x1p100f);      // also true for a NaN
        float zx = fabsf(1.0f / x), zy = fabsf(1.0f / y);
        if (CLAMP) { zx = zx < FTGP_IV_MAX ? zx : FTGP_IV_MAX; zy = zy < FTGP_IV_MAX ? zy : FTGP_IV_MAX; }
        p = oddx ? zx : p; q = oddy ? zy : q;
    }
    rx = p; ry = q;
}

// Ray against another car: chassis box (slab test) and LiDAR puck (circle; left out where it cannot win, VehLds::puck_in_box), binary32 -- the specification's arithmetic with
// everything that does not depend on the ray taken from where it already exists: (bx, by) = mate's origin - my LiDAR centre is
// the pair record's (its negation is the specification's (float)(centre - origin): binary64 subtraction and the conversion
// are odd functions), the mate's heading in binary32 is its LiDAR frame's, the vehicle constants in binary32 are VehLds',
// and the two IEEE reciprocals are rcp_abs2's with the sign put back.
__device__ __forceinline__ float ray_vs_car(const VehLds* V, const LidarFrame* b, float bx, float by, float dxw, float dyw)
{
    const float r0 = V->ring_radius_f;
    float best = INFINITY;
    const float relx = -bx, rely = -by;
    const float ox = fmaf(dxw, -r0, relx), oy = fmaf(dyw, -r0, rely);
    const float cbf = b->chf, sbf = b->shf;
    const float lx = fmaf(cbf, ox, sbf * oy), ly = fmaf(cbf, oy, -(sbf * ox));
    const float ldx = fmaf(cbf, dxw, sbf * dyw), ldy = fmaf(cbf, dyw, -(sbf * dxw));
    {
        const float xmin = V->box_xmin_f, xmax = V->box_xmax_f, ymin = V->box_ymin_f, ymax = V->box_ymax_f;
        float tmin = -INFINITY, tmax = INFINITY; bool miss = false;
        float ax, ay;
        rcp_abs2(ldx, ldy, ax, ay);
        if (ldx != 0.0f) {
            const float inv = copysignf(ax, ldx); const float t1 = (xmin - lx) * inv, t2 = (xmax - lx) * inv;
            tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
        } else if (lx < xmin || lx > xmax) miss = true;
        if (ldy != 0.0f) {
            const float inv = copysignf(ay, ldy); const float t1 = (ymin - ly) * inv, t2 = (ymax - ly) * inv;
            tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
        } else if (ly < ymin || ly > ymax) miss = true;
        if (!miss && tmax >= fmaxf(tmin, 0.0f)) {
            const float t = tmin > 0.0f ? tmin : 0.0f;
            if (t < best) best = t;
        }
    }
    if (!sgpr(V->puck_in_box)) {            // (wave-uniform; with the puck inside the box -- both bundled vehicles -- the box's time is the minimum already: VehLds)
        const float px = lx - V->lidar_x_f, py = ly - V->lidar_y_f;
        const float bq = fmaf(px, ldx, py * ldy);
        const float cq = fmaf(px, px, py * py) - r0 * r0;
        const float disc = fmaf(bq, bq, -cq);
        if (disc >= 0.0f) {
            float t = -bq - sqrtf(disc);
            if (t < 0.0f) t = (cq < 0.0f) ? 0.0f : INFINITY;
            if (t < best) best = t;
        }
    }
    return best;
}

// Measurement hooks (phase stamps, workgroup entry / exit times, phases compiled out): all of them live in diag/ftgp_diag.inc and exist only
// in libraries built with -DFTGP_DIAG by tools/*.sh on the GPU box.  The product build defines none of them (ftgp_build_info() says so,
// tests/test_capi.py checks it).
#include "diag/ftgp_diag.inc"

// Two workgroups share a CU, and the hardware arbitrates vector issue by priority first and age second: left alone, the
// workgroup that was dispatched first wins every contested cycle, finishes a launch ~20 % ahead of its partner and leaves the CU
// half empty (four waves per SIMD, too few to keep the vector pipe full) for the rest of it.  So the two take turns: a 100-MHz
// wall-clock bit that both see picks which of them sweeps at the raised priority.  "Workgroups b and b + (number of CUs) are
// partners" is only what the dispatcher is observed to do (the hardware promises no order); a wrong guess costs fairness,
// never correctness.
#ifndef FTGP_FAIR_SHIFT
#define FTGP_FAIR_SHIFT 14        // turns of 2^14 ticks = 164 us (round 3: 2^12 .. 2^15 equally good; on round 5's kernel 2^14 is 1.2 % ahead of 2^13 in sixteen same-box rows,
                                  // 2^11, 2^12 and 2^15 behind: profiles/round5/ab_fair_turns.log)
#endif
__device__ __forceinline__ void sweep_priority(bool second_half)
{
    if (!FTGP_DIAG_FAIR) return;
    const bool mine = (((uint32_t)__builtin_amdgcn_s_memrealtime() >> FTGP_FAIR_SHIFT) & 1u) != (second_half ? 1u : 0u);
    if (mine) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
}

// The march of one wave's rays until ALL of them sit on their terminal cell: hand-written, for two reasons that round 5 measured one after the other
// (DESIGN.md section 6).  (1) Instruction count: scalar, branch and wait instructions cost a SIMD about what a cheap vector instruction costs
// (profiles/round4/salu_cost.log), and the compiler's loop spends 13 scalar-side instructions per iteration on mask bookkeeping.  Here a lane leaves the
// loop by dropping out of exec (v_cmpx on "the cell's entry is a box", i.e. kx != 0), so the body needs no live mask, finished rays hold their crossing
// times for free (and burn no vector lanes), and an iteration is 20 vector (the look-up's six included) + 3 scalar-side instructions, + the near-boundary path;
// the crossing time into the terminal cell is the smaller of the last jump's two, taken once after the loop.
// (2) The dependent chain from one look-up to the next, which is what the waves spend most of their life on: the next look-up is issued as soon as the
// landing estimate's floor is there; the near-boundary test (fract, recentre, compare, branch) runs in the load's shadow, and a
// lane that turns out to be within eps of a pixel boundary recomputes its cell (ftgp_ray_fix) and looks up again -- answers return in order, the second
// overwrites the first.  For the fix path to find the cell the ray came from after the move has been made, the loop runs two iterations per trip with the
// cell alternating between (mx, my) and (nx, ny).  The first look-up and jump are peeled (the start cell is (0, 0): no zeroes to set, a mask and a shift
// for the byte-select adds), and the tail of ftgp_ray_place -- g, the origin's offset inside its start cell seen in the direction of travel, and c = g * iv --
// is worked out in the shadow of that first load.  (-5.4 % and -1.1 % of the headline's cycles: profiles/round5/ab_speculative_lookup.log,
// ab_setup_latency_experiments.log.)
// In: exec = the lanes that hold a ray; (pu, pv) = the ray's origin in pixels (on the image, or the ray is parked: base = 0, ax = ay = 0), du, dv signed
// (the body reads their magnitudes through the operand modifier), ivx, ivy = |1 / du|, |1 / dv| as the specification wants them, base = the byte offset
// of the start cell's entry, ax, ay = the strides.
// Out: w = the terminal cell's entry (0 = wall, FTGP_FIELD_OUT = ring), s = the crossing time into it; exec as on entry.
// The arithmetic is ftgp_ray_place's tail and ftgp_ray_step / ftgp_ray_fix / ftgp_ray_commit of ftgp_march.h, instruction for instruction: a crossing
// time is fma((float)boundary, iv, -c) -- boundaries counted from the ray's own start cell.
__device__ __forceinline__ void march_all(float& s, uint32_t& w, float pu, float pv, float ivx, float ivy, float du, float dv,
                                          int base, int ax, int ay, float thr /* 0.5f - eps */, const void* field)
{
    int a, b, c, d, e, f, h, i, mx, my, nx, ny;
    float gu, gv, cx, cy;
    uint64_t stepx, sv, sq, ex0;
    asm volatile(
        "s_mov_b64 %[ex0], exec\n\t"
        // crossing times of the box's far edges, the axis that is reached first, the landing estimate and its floor
#define FTGP_MARCH_BODY \
        "v_cvt_f32_i32_e32 %[a], %[c]\n\t" \
        "v_cvt_f32_i32_e32 %[b], %[d]\n\t" \
        "v_fma_f32 %[a], %[a], %[ivx], -%[cx]\n\t"                       /* sX = fma((float)xe, ivx, -cx): the box's far edge is the xe-th boundary */ \
        "v_fma_f32 %[b], %[b], %[ivy], -%[cy]\n\t"                       /* sY */ \
        "v_cmp_lt_f32_e64 %[stepx], %[a], %[b]\n\t"                      /* the x edge of the box is reached first (a tie steps in y) */ \
        "v_fma_f32 %[e], |%[dv]|, %[a], %[gv]\n\t"                       /* landing estimate after an x-jump ... */ \
        "v_fma_f32 %[f], |%[du]|, %[b], %[gu]\n\t"                       /* ... after a y-jump */ \
        "v_cndmask_b32_e64 %[e], %[f], %[e], %[stepx]\n\t"               /* v */ \
        "v_cvt_flr_i32_f32_e32 %[h], %[e]\n\t"                           /* t = floor(v) */
        // the new cell (NX, NY) and its look-up, issued on the landing estimate; THEN the estimate's distance from a pixel boundary, in the shadow of
        // the load: within eps the specification's comparisons decide (fix) and the look-up is issued again for those lanes
#define FTGP_MARCH_MOVE(NX, NY, fix) \
        "v_cndmask_b32_e64 " NX ", %[h], %[c], %[stepx]\n\t" \
        "v_cndmask_b32_e64 " NY ", %[d], %[h], %[stepx]\n\t" \
        "v_mad_i32_i24 %[i], " NY ", %[ay], %[base]\n\t"                 /* entry offset = ftgp_ray_offset() */ \
        "v_mad_i32_i24 %[i], " NX ", %[ax], %[i]\n\t" \
        "global_load_ushort %[w], %[i], %[field]\n\t" \
        "v_fract_f32_e32 %[f], %[e]\n\t" \
        "v_add_f32_e32 %[f], -0.5, %[f]\n\t" \
        "v_cmp_gt_f32_e64 vcc, |%[f]|, %[thr]\n\t" \
        "s_cbranch_vccnz " fix "\n"
        // the entry of the cell the ray stands on: far corner of its box; kx == 0 -- a wall or ring cell -- and the lane is done
#define FTGP_MARCH_ARRIVE(NX, NY) \
        "s_waitcnt vmcnt(0)\n\t" \
        "v_add_u32_sdwa %[c], " NX ", %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"     /* xe = mx + kx */ \
        "v_add_u32_sdwa %[d], " NY ", %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"     /* ye = my + ky */ \
        "v_cmpx_ne_u32_e32 vcc, %[c], " NX "\n\t"
        // ftgp_ray_fix() for the lanes in vcc; cur = the select that yields the transverse coordinate of the cell the ray stands on
#define FTGP_MARCH_FIX(cur, NX, NY, back) \
        "s_and_saveexec_b64 %[sv], vcc\n\t" \
        "v_cndmask_b32_e64 %[s], %[cx], %[cy], %[stepx]\n\t"             /* transverse c ... (s, f, i, h: scratch here -- sX, sY stay: the crossing time is their minimum, taken after the loop) */ \
        "v_cndmask_b32_e64 %[f], %[ivx], %[ivy], %[stepx]\n\t"           /* ... reciprocal */ \
        "v_rndne_f32_e32 %[e], %[e]\n\t"                                 /* the boundary in doubt */ \
        "v_fma_f32 %[s], %[e], %[f], -%[s]\n\t"                          /* its crossing time, the specification's way */ \
        "v_cvt_i32_f32_e32 %[f], %[e]\n\t" \
        "v_min_f32_e32 %[h], %[a], %[b]\n\t"                             /* sn, the crossing time of the jump */ \
        "v_cmp_lt_f32_e64 vcc, %[s], %[h]\n\t" \
        "v_cmp_le_f32_e64 %[sq], %[s], %[h]\n\t" \
        "s_and_b64 %[sq], %[sq], %[stepx]\n\t"                           /* crossed: S <= sn after an x-jump, S < sn after a y-jump */ \
        "s_or_b64 vcc, vcc, %[sq]\n\t" \
        "v_cndmask_b32_e64 %[s], -1, 0, vcc\n\t" \
        "v_add_u32_e32 %[h], %[f], %[s]\n\t"                             /* the cell beyond the boundary if crossed, else the one before */ \
        cur                                                              /* the cell the ray stands on (transverse coordinate, into f) ... */ \
        "v_cndmask_b32_e64 %[i], %[c], %[d], %[stepx]\n\t" \
        "v_add_u32_e32 %[i], -1, %[i]\n\t"                               /* ... and the last cell of the box's span */ \
        "v_med3_i32 %[h], %[h], %[f], %[i]\n\t"                          /* inside the box's span */ \
        "v_cndmask_b32_e64 " NX ", %[h], %[c], %[stepx]\n\t"             /* the cell and its look-up again (the first look-up's answer arrives first and is overwritten) */ \
        "v_cndmask_b32_e64 " NY ", %[d], %[h], %[stepx]\n\t" \
        "v_mad_i32_i24 %[i], " NY ", %[ay], %[base]\n\t" \
        "v_mad_i32_i24 %[i], " NX ", %[ax], %[i]\n\t" \
        "global_load_ushort %[w], %[i], %[field]\n\t" \
        "s_mov_b64 exec, %[sv]\n\t" \
        "s_branch " back "\n"
        // The start cell and the first jump, peeled: the ray stands on cell (0, 0), so the box's far corner is the entry's two bytes as they
        // come (a mask and a shift for two byte-select adds) and nobody has to set the cell to zero first.
        "global_load_ushort %[w], %[base], %[field]\n\t"
        // ... and in the shadow of that load what the jumps need of the ray's origin (the tail of ftgp_ray_place): its offset inside the start cell seen in
        // the direction of travel, g, and c = g * iv; the crossing time into the start cell is 0
        "v_fract_f32_e32 %[gu], %[pu]\n\t"                               // p - floor(p): exact for every p >= 0 (an origin off the image is parked: it ends on this look-up)
        "v_fract_f32_e32 %[gv], %[pv]\n\t"
        "v_sub_f32_e32 %[a], 1.0, %[gu]\n\t"
        "v_sub_f32_e32 %[b], 1.0, %[gv]\n\t"
        "v_cmp_gt_f32_e32 vcc, 0, %[du]\n\t"                             // (a component of -0 counts as not negative: it never steps)
        "v_cndmask_b32_e32 %[gu], %[gu], %[a], vcc\n\t"
        "v_cmp_gt_f32_e32 vcc, 0, %[dv]\n\t"
        "v_cndmask_b32_e32 %[gv], %[gv], %[b], vcc\n\t"
        "v_mul_f32_e32 %[cx], %[gu], %[ivx]\n\t"
        "v_mul_f32_e32 %[cy], %[gv], %[ivy]\n\t"
        "v_mov_b32_e32 %[a], 0\n\t"                                      // (a ray that ends on its start cell: its crossing time is min(0, 0))
        "v_mov_b32_e32 %[b], 0\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_and_b32_e32 %[c], 0xff, %[w]\n\t"                              // xe = kx
        "v_lshrrev_b32_e32 %[d], 8, %[w]\n\t"                             // ye = ky
        "v_cmpx_ne_u32_e32 vcc, 0, %[c]\n\t"                              // kx == 0: the ray starts in a wall (or was parked on a ring cell)
        "s_cbranch_execz L_march_done_%=\n\t"
        FTGP_MARCH_BODY
        FTGP_MARCH_MOVE("%[mx]", "%[my]", "L_march_fix0_%=")
        "L_march_arrive0_%=:\n\t"
        FTGP_MARCH_ARRIVE("%[mx]", "%[my]")
        "s_cbranch_execz L_march_done_%=\n"
        // the loop, two iterations per trip: the cell alternates between (mx, my) and (nx, ny), so that the near-boundary path still finds the
        // cell the ray came from after the move has been made
        "L_march_loop_%=:\n\t"
        FTGP_MARCH_BODY
        FTGP_MARCH_MOVE("%[nx]", "%[ny]", "L_march_fix1_%=")
        "L_march_arrive1_%=:\n\t"
        FTGP_MARCH_ARRIVE("%[nx]", "%[ny]")
        "s_cbranch_execz L_march_done_%=\n\t"
        FTGP_MARCH_BODY
        FTGP_MARCH_MOVE("%[mx]", "%[my]", "L_march_fix2_%=")
        "L_march_arrive2_%=:\n\t"
        FTGP_MARCH_ARRIVE("%[mx]", "%[my]")
        "s_cbranch_execnz L_march_loop_%=\n\t"
        "s_branch L_march_done_%=\n"
        "L_march_fix0_%=:\n\t"
        FTGP_MARCH_FIX("v_mov_b32_e32 %[f], 0\n\t", "%[mx]", "%[my]", "L_march_arrive0_%=")
        "L_march_fix1_%=:\n\t"
        FTGP_MARCH_FIX("v_cndmask_b32_e64 %[f], %[mx], %[my], %[stepx]\n\t", "%[nx]", "%[ny]", "L_march_arrive1_%=")
        "L_march_fix2_%=:\n\t"
        FTGP_MARCH_FIX("v_cndmask_b32_e64 %[f], %[nx], %[ny], %[stepx]\n\t", "%[mx]", "%[my]", "L_march_arrive2_%=")
        "L_march_done_%=:\n\t"
        "s_mov_b64 exec, %[ex0]\n\t"
        "v_min_f32_e32 %[s], %[a], %[b]"                                  // the crossing time into the terminal cell: the smaller of the last jump's two (a finished lane's stay as they were)
        : [s] "=&v"(s), [w] "=&v"(w), [mx] "=&v"(mx), [my] "=&v"(my), [nx] "=&v"(nx), [ny] "=&v"(ny),
          [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [d] "=&v"(d), [e] "=&v"(e), [f] "=&v"(f), [h] "=&v"(h), [i] "=&v"(i),
          [gu] "=&v"(gu), [gv] "=&v"(gv), [cx] "=&v"(cx), [cy] "=&v"(cy),
          [stepx] "=&s"(stepx), [sv] "=&s"(sv), [sq] "=&s"(sq), [ex0] "=&s"(ex0)
        : [pu] "v"(pu), [pv] "v"(pv), [ivx] "v"(ivx), [ivy] "v"(ivy), [du] "v"(du), [dv] "v"(dv),
          [base] "v"(base), [ax] "v"(ax), [ay] "v"(ay), [thr] "s"(thr), [field] "s"(field)
        : "vcc", "scc", "memory");
#undef FTGP_MARCH_BODY
#undef FTGP_MARCH_MOVE
#undef FTGP_MARCH_ARRIVE
#undef FTGP_MARCH_FIX
}

// The sweep of one step for all cars of the workgroup, by every wave, in GROUPS of 64 consecutive rays of one car: a wave draws the
// next task (one LDS atomic), sets its 64 rays up, marches them until all have finished (march_all), stores the 64 ranges, and draws
// again.  Nothing is handed out ray by ray: no ranks, no per-lane bookkeeping, no partial refills -- the price is that a group
// lasts as long as its slowest ray (lane utilisation 0.61 against 0.78 with batched refills, tools/sweep_model.cpp), paid in lanes
// that sit masked out, not in instructions: a set-up costs about 60 instructions per 64 rays against 150 per 52, an iteration 24
// against 35.  Rays of a group are neighbours: they share field lines, and their ranges leave as one 256-byte row segment.
// With the rangefinders' own fan -- ray j + n/2 points exactly opposite to ray j (ftgp_create builds the table that way) -- a task is
// a PAIR of groups, j0 .. j0 + 63 and the same rays turned round: the second group keeps the first one's |direction|, reciprocals and
// slope slice (the two divisions and the sector search are half of a set-up) and only places its rays anew (ftgp_ray_place with
// sector ^ 3).  Tasks are drawn long-first across the workgroup's cars (group_order), so that the waves end a sweep together.
template <bool MULTI>
__device__ __forceinline__ void lidar_groups(const DeviceParams& P, ScalarParams G, const Lds& L, const LidarFrame* frames, const PairCull* pairs, const unsigned char* mmask,
                                             float* scan_rows, int* pool, int ncars_here, int ci0, bool scan_lds, bool second_half STAMP_ARG)
{
    typedef __attribute__((address_space(1))) float* global_f32;
    typedef __attribute__((address_space(1))) unsigned char* global_u8w;
    struct alignas(16) Task { int32_t x, y, z, w; };
    typedef const __attribute__((address_space(4))) Task* ScalarTasks;
    const int R = G->n_rays, half = R >> 1, ntasks = G->cars_per_block * G->tasks_per_car;
    const int W = G->width, H = G->height, fstride = G->fstride;
    const uint32_t plane256 = G->plane256;
    const int eighth = G->eighth, win_floats = G->win_floats;
    const float isx = G->inv_px_x_f, isy = G->inv_px_y_f, thr = 0.5f - G->snap_eps, nsf = G->slice_factor;
    const float r0 = sgpr(L.veh->ring_radius_f);
    const void* field = G->field;
    const ScalarTasks tasks = (ScalarTasks)G->task_tab + (scan_lds ? 0 : ntasks);      // (the second table: no ray goes to a scan window)
    const global_u8w ranges = (global_u8w)((global_f32)G->ranges + (size_t)ci0 * G->ranges_stride);      // the workgroup's first row
    const int mmask_stride = G->mmask_stride;
    const int lane = lane_here();
    const bool all_safe = sgpr(pool[4]) == 0;        // every ray of this sweep starts on the image (frame_write): no test per ray
    sweep_priority(second_half);
    for (;;) {                   // (ends: every draw moves the counter on)
        STAMP(ta);
        // (a task is drawn when the wave is ready for it, not earlier: drawing the next one before the march -- to hide the LDS round
        // trip -- reserves work behind a wave that may be on a long group, and cost 6 % on the headline and 12 % on config 2; drawing it
        // between the last march and its delivery is within noise: profiles/round5/ab_setup_latency_experiments.log)
        int g = 0;
        if (lane == 0) g = atomicAdd(pool, 1);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= ntasks) break;
        // Draw g = the (g / cars_per_block)-th most expensive task (rays along the car's axis march longest, see group_order) of car slot
        // g % cars_per_block: the long tasks of every car go first, so that the waves of a workgroup end a sweep within a short group of each
        // other.  Everything about the draw that ftgp_create can know -- first ray, kind, car slot, where the car's LiDAR frame, its row of
        // ranges and its scan window start, whether the group's rays lie in the drivers' window -- comes as ONE 16-byte scalar load
        // (ftgp_task_entry; round 5: a division by multiplication, a table read and three 32- and 64-bit address products per group used to
        // stand here and in the delivery: a third of the kernel's scalar instructions).
        Task task; { const uint32_t gu = (uint32_t)g; task.x = tasks[gu].x; task.y = tasks[gu].y; task.z = tasks[gu].z; task.w = tasks[gu].w; }
        const int j0 = task.x & 0x3fff, kind = (task.x >> 14) & 3, c = (task.x >> 16) & 15;     // kind 0: one group; 1: a group and its opposite; 2: both halves in one group
        int j; bool mine;
        if (kind == 2) { j = j0 + (lane & 31) + (lane >= 32 ? half : 0); mine = (lane & 31) < half - j0; }
        else { j = j0 + lane; mine = j < (kind == 1 ? half : R); }
        mine = mine && c < ncars_here;               // (a ragged last workgroup draws tasks of cars it does not have)
        const LidarFrame* frame = reinterpret_cast<const LidarFrame*>(reinterpret_cast<const unsigned char*>(frames) + (task.y & 0xffff));
        FtgpRay ray;
        float du, dv, dxw, dyw, opu, opv;              // opu, opv: the ray's origin in pixels
        uint32_t sector;
        asm volatile("" : "=v"(du), "=v"(dv), "=v"(dxw), "=v"(dyw), "=v"(opu), "=v"(opv), "=v"(sector));      // (lanes that hold no ray never look at these: no moves to define them)
        // march the lanes' rays and deliver their ranges: ftgp_ray_range(), the inter-vehicle test, the stores
        auto finish = [&](bool active, int pass) {
            if (!active) return;
            uint32_t w;
            STAMP(tb);
            march_all(ray.s, w, opu, opv, ray.ivx, ray.ivy, du, dv, ray.base, ray.ax, ray.ay, thr, field);
            STAMP(tc); STAMP_ADD(10, tc - tb); STAMP_ADD(9, 1);
            float r = (w == 0u) ? fabsf(ray.s) : ray.result;
            if (MULTI && FTGP_DIAG_RUN_MATES) {
                // Rays also see the other cars of the env (a9).  One record per env-mate (PairCull, written with the frames) rules a
                // mate out with a dot product: it can only be touched if it lies in front of the ray and within `cull` of its line.
                const PairCull* mates = pairs + c * FTGP_PAIR_STRIDE;
                const int slot0 = sgpr(frame->slot0);
                // ... and one byte per (car, group) says which mates the group's rays can see at all (mate_masks): mostly none
                const int kidx = task.y >> 16;
                uint32_t mm = (uint32_t)sgpr((int)mmask[c * mmask_stride + 2 * kidx + pass]);
                while (mm) {
                    const int k = __builtin_ctz(mm); mm &= mm - 1u;
                    const float4 q = *reinterpret_cast<const float4*>(mates + k);
                    const float al = fmaf(q.x, dxw, q.y * dyw);
                    if (al >= q.z) {
                        const float cull = L.veh->cull_radius;
                        if (r >= 0.0f && (al + r0) - cull > r) continue;             // the mate lies beyond the wall hit
                        const float rc = ray_vs_car(L.veh, frames + slot0 + k, q.x, q.y, dxw, dyw);
                        if (rc < INFINITY && (r < 0.0f || rc < r)) r = rc;
                    }
                }
            }
            // every range leaves for HBM now, 64 consecutive floats per wave (one 256-byte row segment); the window the on-device drivers
            // read next step -- ranges[0] and ranges[eighth : n - eighth] -- is kept in LDS as well
            const uint32_t j4 = (uint32_t)j << 2;
            *(global_f32)(ranges + (j4 + (uint32_t)task.z)) = r;
            // task.w: byte offset of the car's scan row + ((eighth & 3) - eighth) floats, so that sample j sits at task.w + 4 j.  Whether the
            // group's rays lie inside the window, outside it or across one of its ends is the task's to know (two bits per pass; all zero in
            // the table of a launch whose drivers do not read the scan).  The rare cases sit behind scalar branches that the optimiser is kept
            // from turning into selects.
            const int wclass = (task.x >> (20 + 2 * pass)) & 3;
            if (wclass) {                        // (a group across an end of the window sends its outside rays to a spare word: one store site, no masks)
                float* dst = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(scan_rows) + task.w + j4);
                if (wclass == 2) { asm volatile(""); dst = (unsigned)(j - eighth) < (unsigned)(R - 2 * eighth) ? dst : reinterpret_cast<float*>(pool + 6); }
                *dst = r;
            }
            if ((task.x & (1 << 24)) && pass == 0) {          // the group holds ray 0: ranges[0] (fast.py:135) sits in the last float of the row
                asm volatile("");
                if (j == 0) *reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(scan_rows) + task.w + 4 * (win_floats - 1 - (eighth & 3) + eighth)) = r;
            }
        };
        if (mine) {
            const float4 f4 = *reinterpret_cast<const float4*>(frame);      // u0, v0, chf, shf (one address for the wave)
            const float2 bd = L.ray[j];
            dxw = fmaf(f4.z, bd.x, -(f4.w * bd.y));
            dyw = fmaf(f4.w, bd.x, f4.z * bd.y);
            du = dxw * isx;
            dv = -(dyw * isy);
            const float pu = fmaf(du, -r0, f4.x);
            const float pv = fmaf(dv, -r0, f4.y);
            opu = pu; opv = pv;
            float ivx, ivy;
            rcp_abs2<true>(du, dv, ivx, ivy);
            ray.result = -1.0f;
            sector = ftgp_ray_sector(du, dv, ivx, ivy, nsf);
            ftgp_ray_place(ray, pu, pv, du, dv, ivx, ivy, sector, W, H, fstride, plane256, true, &P.sector_tab[0][0]);
            if (!all_safe) {             // wave-uniform, rare: some car of the workgroup is near the image edge, off it, or has finished
                ftgp_ray_park_if_outside(ray, pu, pv, W, H);
                // a finished car's rangefinders are switched off (custom.py:1436-1439): its frame carries u0 = -inf, so the ray is
                // parked like any ray that starts off the image, and reads 0 instead of -1
                ray.result = (f4.x == -INFINITY) ? 0.0f : -1.0f;
            }
        }
        STAMP(td); STAMP_ADD(8, td - ta);
        finish(mine, 0);
        if (kind == 1) {                 // the same rays turned round: ray j + n/2 = -(ray j), exactly
            if (mine) {
                j += half;
                const float2 f2 = *reinterpret_cast<const float2*>(frame);
                du = -du; dv = -dv; dxw = -dxw; dyw = -dyw;
                const float pu = fmaf(du, -r0, f2.x);
                const float pv = fmaf(dv, -r0, f2.y);
                opu = pu; opv = pv;
                ftgp_ray_place(ray, pu, pv, du, dv, ray.ivx, ray.ivy, sector ^ 3u, W, H, fstride, plane256, true, &P.sector_tab[0][0]);
                if (!all_safe) {
                    ftgp_ray_park_if_outside(ray, pu, pv, W, H);
                    ray.result = (f2.x == -INFINITY) ? 0.0f : -1.0f;
                }
            }
            finish(mine, 1);
        }
    }
}

// K2 in FTGP_LIDAR_FAKELIDAR mode: the reference's own 2-D LiDAR inside the step loop (option use_simulated_simulation_lidar,
// custom.py:987,1381-1393).  Per ray, in binary64 and in the order of the Python statements:
//   origin   i_x = (x / s) * W, i_y = -(y / s) * H of the car's position, s = 20 * scale (custom.py:1382-1384)
//   ray j    image-frame direction (dxw, -dyw), (dxw, dyw) = R(yaw) * fan[j]: the rangefinders' order
//   march    raycast.py:9-20: while dt[int(y), int(x)] > 2 and 0 <= x <= W and 0 <= y <= H: advance by dt
//   range    (distance / W) * s (custom.py:1392-1393), stored as binary32
// int() truncates toward zero and a negative index wraps like numpy's; an index past the end -- the reference's IndexError --
// ends the ray with range -1.
//
// march_fake(): the loop of raycast.py:12-17 for the 64 rays of a wave, hand-written like march_all() -- a ray leaves the loop by dropping
// out of exec -- with the statement order kept and the tests reduced to what can differ:
//   * numpy raises IndexError before the loop condition is looked at: an index is valid iff -n <= int(v) < n, i.e. (unsigned)(int(v) + n) < 2 n
//     (`bad`: the ray reads -1);
//   * a ray with x < 0 or y < 0 ends (its look-up, valid or wrapped, no longer matters: no load is issued for it); x <= W and y <= H need no
//     test of their own: a ray that is still here has int(x) < W, int(y) < H;
//   * what is left has 0 <= int(x) < W, 0 <= int(y) < H: one 24-bit multiply-add for the index, one 8-byte load (the table is
//     L2-resident: hit rate 0.998, profiles/round5/fakelidar_r5base.log), `nearest > 2` by v_cmpx.
// Binary64 throughout, one rounding per written operation (x += dx * nearest is a product and a sum).  `guard`: iterations after which a ray
// is given up with what it has accumulated (a caller's fan may hold a zero direction: the reference would loop for ever).
__device__ __forceinline__ void march_fake(double& x, double& y, double& dist, int& bad, double dx, double dy, int W, int H, const double* dt)
{
    double n = 0.0, t, u;
    int xi, yi, a, b;
    uint64_t sq, ex0;
    int guard = 1 << 16;
    // The look-up of an iteration is ISSUED before its tests are made (round 5, like march_all's): on an index clamped to the table -- for a ray
    // that passes the tests the clamp changes nothing, for one that fails them the answer is never looked at -- so that the dependent chain from
    // one look-up to the next is convert, clamp, index, load instead of convert and a dozen tests first; the tests run in the load's shadow.
    asm volatile(
        "s_mov_b64 %[ex0], exec\n"
        "L_fake_loop_%=:\n\t"
        "v_cvt_i32_f64_e32 %[xi], %[x]\n\t"                              // int(x): truncation toward zero
        "v_cvt_i32_f64_e32 %[yi], %[y]\n\t"
        "v_med3_i32 %[a], %[xi], 0, %[Wm1]\n\t"
        "v_med3_i32 %[b], %[yi], 0, %[Hm1]\n\t"
        "v_mad_u32_u24 %[a], %[b], %[W], %[a]\n\t"
        "v_lshlrev_b32_e32 %[a], 3, %[a]\n\t"
        "global_load_dwordx2 %[n], %[a], %[dt]\n\t"
        "v_add_u32_e32 %[a], %[W], %[xi]\n\t"
        "v_add_u32_e32 %[b], %[H], %[yi]\n\t"
        "v_cmp_le_u32_e32 vcc, %[W2], %[a]\n\t"                          // IndexError: not (-W <= int(x) < W) ...
        "v_cmp_le_u32_e64 %[sq], %[H2], %[b]\n\t"                        // ... or not (-H <= int(y) < H)
        "s_or_b64 vcc, vcc, %[sq]\n\t"
        "v_cndmask_b32_e64 %[bad], %[bad], -1, vcc\n\t"
        "v_cmp_gt_f64_e64 %[sq], 0, %[x]\n\t"                            // 0 <= x fails
        "s_or_b64 vcc, vcc, %[sq]\n\t"
        "v_cmp_gt_f64_e64 %[sq], 0, %[y]\n\t"
        "s_or_b64 vcc, vcc, %[sq]\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"                                // those rays are done
        "s_cbranch_execz L_fake_done_%=\n\t"
        "s_sub_u32 %[guard], %[guard], 1\n\t"
        "s_cbranch_scc1 L_fake_done_%=\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_cmpx_lt_f64_e32 vcc, 2.0, %[n]\n\t"                           // while nearest > eps
        "s_cbranch_execz L_fake_done_%=\n\t"
        "v_mul_f64 %[t], %[dx], %[n]\n\t"
        "v_mul_f64 %[u], %[dy], %[n]\n\t"
        "v_add_f64 %[x], %[x], %[t]\n\t"                                 // x += dx * nearest
        "v_add_f64 %[y], %[y], %[u]\n\t"                                 // y += dy * nearest
        "v_add_f64 %[dist], %[dist], %[n]\n\t"                           // distance += nearest
        "s_branch L_fake_loop_%=\n"
        "L_fake_done_%=:\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_mov_b64 exec, %[ex0]"
        : [x] "+v"(x), [y] "+v"(y), [dist] "+v"(dist), [bad] "+v"(bad), [n] "+v"(n), [t] "=&v"(t), [u] "=&v"(u), [xi] "=&v"(xi), [yi] "=&v"(yi), [a] "=&v"(a), [b] "=&v"(b),
          [sq] "=&s"(sq), [ex0] "=&s"(ex0), [guard] "+s"(guard)
        : [dx] "v"(dx), [dy] "v"(dy), [W] "s"(W), [H] "s"(H), [W2] "s"(2 * W), [H2] "s"(2 * H), [Wm1] "s"(W - 1), [Hm1] "s"(H - 1), [dt] "s"(dt)
        : "vcc", "scc", "memory");
}

// The FAKELIDAR sweep of one step for all cars of the workgroup: the task list, the draw and the delivery of lidar_groups() -- groups of 64
// neighbouring rays of one car, long-first across the workgroup's cars, a pair's two groups one after the other -- around march_fake().
// What depends on the car alone -- i_x, i_y (two binary64 divisions) and the heading -- comes with its LiDAR frame (frame_write).
__device__ __forceinline__ void lidar_fake(ScalarParams G, const LidarFrame* frames, float* scan_rows, int* pool, int ncars_here, int ci0, bool scan_lds)
{
    typedef __attribute__((address_space(1))) float* global_f32;
    typedef __attribute__((address_space(1))) unsigned char* global_u8w;
    struct alignas(16) Task { int32_t x, y, z, w; };             // DeviceParams::task_tab (see lidar_groups)
    typedef const __attribute__((address_space(4))) Task* ScalarTasks;
    const int R = G->n_rays, half = R >> 1, ntasks = G->cars_per_block * G->tasks_per_car;
    const int W = G->width, H = G->height;
    const int eighth = G->eighth, win_floats = G->win_floats;
    const double s = G->map_size;
    const double* dt = G->edt;
    const double* __restrict__ fan = G->fan_dirs;
    const ScalarTasks tasks = (ScalarTasks)G->task_tab + (scan_lds ? 0 : ntasks);
    const global_u8w ranges = (global_u8w)((global_f32)G->ranges + (size_t)ci0 * G->ranges_stride);
    const int lane = lane_here();
    for (;;) {
        int g = 0;
        if (lane == 0) g = atomicAdd(pool, 1);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= ntasks) break;
        Task task; { const uint32_t gu = (uint32_t)g; task.x = tasks[gu].x; task.y = tasks[gu].y; task.z = tasks[gu].z; task.w = tasks[gu].w; }
        const int j0 = task.x & 0x3fff, kind = (task.x >> 14) & 3, c = (task.x >> 16) & 15;
        int j; bool mine;
        if (kind == 2) { j = j0 + (lane & 31) + (lane >= 32 ? half : 0); mine = (lane & 31) < half - j0; }
        else { j = j0 + lane; mine = j < (kind == 1 ? half : R); }
        mine = mine && c < ncars_here;
        const LidarFrame* f = reinterpret_cast<const LidarFrame*>(reinterpret_cast<const unsigned char*>(frames) + (task.y & 0xffff));
        for (int pass = 0; pass < (kind == 1 ? 2 : 1); ++pass, j += half) {
            if (!mine) continue;
            float r = 0.0f;                                       // a finished car's rangefinders are switched off (custom.py:1436-1439): its scan reads 0
            if (!f->finished) {
                const double ch = f->x, sh = f->y;                // FAKELIDAR frames: heading (cos, sin) in binary64 (frame_write)
                const double bx = fan[2 * j], by = fan[2 * j + 1];
                const double dxw = ch * bx - sh * by, dyw = sh * bx + ch * by;
                double x = f->lcx, y = f->lcy, distance = 0.0;    // FAKELIDAR frames: i_x, i_y (custom.py:1383-1384)
                int bad = 0;
                march_fake(x, y, distance, bad, dxw, -dyw, W, H, dt);      // image rows grow downwards
                r = bad ? -1.0f : (float)((distance / (double)W) * s);     // ranges /= original_width; ranges *= s (custom.py:1392-1393)
            }
            const uint32_t j4 = (uint32_t)j << 2;
            *(global_f32)(ranges + (j4 + (uint32_t)task.z)) = r;
            const int wclass = (task.x >> (20 + 2 * pass)) & 3;   // the delivery of lidar_groups(): window classes from the task
            if (wclass) {
                float* dst = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(scan_rows) + task.w + j4);
                if (wclass == 2) { asm volatile(""); dst = (unsigned)(j - eighth) < (unsigned)(R - 2 * eighth) ? dst : reinterpret_cast<float*>(pool + 6); }
                *dst = r;
            }
            if ((task.x & (1 << 24)) && pass == 0) {
                asm volatile("");
                if (j == 0) *reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(scan_rows) + task.w + 4 * (win_floats - 1 - (eighth & 3) + eighth)) = r;
            }
        }
    }
}

// device arithmetic the kernels rely on, checked over every binary32 bit pattern: rcp_abs2() == |1 / x| (IEEE division) in either slot
__global__ void ftgp_selftest_rcp_kernel(unsigned long long* __restrict__ mismatches)
{
    unsigned long long bad = 0;
    const uint32_t per = 1u << 12;                                     // 2^20 threads x 2^12 patterns
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per; ++k) {
        const float x = __uint_as_float(t * per + k);
        float a, a2;
        rcp_abs2(x, 3.0f, a, a2);
        const float b = fabsf(1.0f / x);
        float c, c2;
        rcp_abs2(0.7f, x, c2, c);                              // second operand too (and the first in range: the slow path only if x needs it)
        const bool same = (__float_as_uint(a) == __float_as_uint(b) || (a != a && b != b)) && (__float_as_uint(c) == __float_as_uint(b) || (c != c && b != b));
        bad += same ? 0 : 1;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// =============================================================================================
// K3: lap progress (custom.py:1340-1372)
// =============================================================================================
// (start and finish_step -- 64 bits each, touched on line crossings only -- stay in the car's record and are reached through `steps64`:
// carried in registers they would push the K1 + K3 wave over the kernel's 64-register budget)
struct Race { int32_t completion, laps, offset, good_start, finished, off_track, delta, n_times; double dist2; };

__device__ __forceinline__ void progress_update(const DeviceParams& P, Race& s, int64_t steps, int closest, double best, double* __restrict__ times, int64_t* steps64 /* &start, &finish_step */)
{
    s.dist2 = best;                                       // custom.py:1343 (squared)
    s.off_track = best > 1.0;                             // custom.py:1344
    if (s.off_track) return;                              // custom.py:1345: progress frozen off-track
    const int completion = ((closest - s.offset) % 100 + 100) % 100;
    const int delta = completion - s.completion;
    s.delta = (((completion - s.completion + 50) % 100) + 100) % 100 - 50;
    if (abs(delta) > 90) {
        // one store site for the ring (an appended lap time, or the mark of an emptied slot): two would not fit the K1 + K3 wave's registers
        double val = (double)(steps - steps64[0]) * P.dt;      // lap_time
        int slot = -1;
        if (s.delta < 0) {                                // backwards across the line, custom.py:1352-1356
            s.laps -= 1;
            s.good_start = 0;
            if (s.n_times != 0) {                             // times.pop()
                // the ring keeps the newest FTGP_MAX_LAP_TIMES: beyond that the popped entry sits in the slot of the oldest one the list would
                // still show (it was overwritten when the popped lap was appended) -- that slot is marked empty (NaN: skipped by every reader)
                if (s.n_times > FTGP_MAX_LAP_TIMES) { slot = s.n_times - 1; val = val * __builtin_nan(""); }
                s.n_times -= 1;
            }
        } else if (s.delta > 0) {                         // custom.py:1357-1366
            if (s.good_start) {
                slot = s.n_times;                             // times.append(lap_time): a ring of the newest FTGP_MAX_LAP_TIMES
                s.n_times += 1;
                steps64[0] = steps;
            }
            s.laps += 1;
            s.good_start = 1;
        }
        if (slot >= 0) times[slot & (FTGP_MAX_LAP_TIMES - 1)] = val;
    }
    if (s.laps >= P.lap_target) {                         // custom.py:1367-1370; the step of the first time orders the winners (custom.py:1368-1369)
        if (!s.finished) steps64[1] = steps;
        s.finished = 1;
    }
    s.completion = completion;
}

__device__ __forceinline__ void race_load(Race& r, const CarCore* st)
{
    r.completion = st->completion; r.laps = st->laps; r.offset = st->offset;
    r.good_start = st->good_start; r.finished = st->finished; r.off_track = st->off_track; r.delta = st->delta;
    r.n_times = st->n_times; r.dist2 = st->dist2;
}
__device__ __forceinline__ void race_store(const Race& r, CarCore* st)
{
    st->completion = r.completion; st->laps = r.laps;
    st->good_start = r.good_start; st->finished = r.finished; st->off_track = r.off_track; st->delta = r.delta;
    st->n_times = r.n_times; st->dist2 = r.dist2;
}

// =============================================================================================
// K1: integrate one dt (reduced planar model of template/mushr.em.xml stepped by mj_step, custom.py:1425)
// =============================================================================================

// 32 wall bits of bitmap row `row` starting at column x0 (bit i = column x0 + i; columns outside the image read 0)
__device__ __forceinline__ uint32_t wall_window(const uint32_t* __restrict__ row, int wpr, int x0)
{
    const int wi = x0 >> 5, sh = x0 & 31;
    const uint32_t lo = (wi >= 0 && wi < wpr) ? row[wi] : 0u;
    const uint32_t hi = (wi + 1 >= 0 && wi + 1 < wpr) ? row[wi + 1] : 0u;
    return sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
}

// One circle (centre (px, py), radius r) against the wall pixels, in two parts so that neither keeps the other's values in
// registers (the kernel runs under a 64-VGPR budget): the search returns the pixel of deepest penetration -- ties go to the
// first pixel in raster order -- and the force part turns it into the penalty spring/damper term along the contact normal.
struct WallHit { double pen; int cx, cy; bool found; };

__device__ __forceinline__ WallHit wall_search(const DeviceParams& P, const uint32_t* __restrict__ bits, const uint32_t* __restrict__ nearbits,
                                               double px, double py, double r)
{
    WallHit h; h.pen = 0.0; h.cx = 0; h.cy = 0; h.found = false;
    const int W = P.width, H = P.height, wpr = P.words_per_row;
    const double sx = P.px_size_x, sy = P.px_size_y;
    const int nx = (int)ceil(r * P.inv_px_x), ny = (int)ceil(r * P.inv_px_y);
    const double u = (px - P.origin_x) * P.inv_px_x, w = (P.origin_y - py) * P.inv_px_y;
    const int ix = (int)floor(u), iy = (int)floor(w);
    if (ix < 0 || ix >= W || iy < 0 || iy >= H) return h;
    if (!((nearbits[(size_t)iy * wpr + (ix >> 5)] >> (ix & 31)) & 1u)) return h;      // no wall pixel within the window
    const int wx = 2 * nx + 1;
    for (int dy = -ny; dy <= ny; ++dy) {
        const int cy = iy + dy;
        if (cy < 0 || cy >= H) continue;
        const uint32_t* row = bits + (size_t)cy * wpr;
        for (int cb = 0; cb < wx; cb += 32) {
            const int x0 = ix - nx + cb, nb = wx - cb < 32 ? wx - cb : 32;
            uint32_t win = wall_window(row, wpr, x0);
            if (nb < 32) win &= (1u << nb) - 1u;
            while (win) {
                const int cx = x0 + __builtin_ctz(win);
                win &= win - 1u;
                const double x0w = P.origin_x + (double)cx * sx, x1w = x0w + sx;
                const double y1w = P.origin_y - (double)cy * sy, y0w = y1w - sy;
                const double qx = px < x0w ? x0w : (px > x1w ? x1w : px);
                const double qy = py < y0w ? y0w : (py > y1w ? y1w : py);
                const double ex = px - qx, ey = py - qy;
                const double d2 = ex * ex + ey * ey;
                if (d2 >= r * r) continue;
                const double pen = r - sqrt(d2);
                if (!h.found || pen > h.pen) { h.pen = pen; h.cx = cx; h.cy = cy; h.found = true; }
            }
        }
    }
    return h;
}

__device__ __forceinline__ Force wall_force(const DeviceParams& P, const WallHit& h, double px, double py, double vx, double vy, double wz,
                                            double rxw, double ryw, double stiffness, double damping)
{
    Force out = { 0.0, 0.0, 0.0 };
    const double sx = P.px_size_x, sy = P.px_size_y;
    const double x0w = P.origin_x + (double)h.cx * sx, x1w = x0w + sx;
    const double y1w = P.origin_y - (double)h.cy * sy, y0w = y1w - sy;
    const double qx = px < x0w ? x0w : (px > x1w ? x1w : px);
    const double qy = py < y0w ? y0w : (py > y1w ? y1w : py);
    const double ex = px - qx, ey = py - qy;
    const double d = sqrt(ex * ex + ey * ey);
    double nxv, nyv;
    if (d > 0.0) { nxv = ex / d; nyv = ey / d; }
    else {
        const double mx = px - (x0w + 0.5 * sx), my = py - (y0w + 0.5 * sy);
        const double mm = sqrt(mx * mx + my * my);
        if (mm > 0.0) { nxv = mx / mm; nyv = my / mm; } else { nxv = 0.0; nyv = 0.0; }
    }
    const double vcx = vx - wz * ryw, vcy = vy + wz * rxw;
    const double vn = vcx * nxv + vcy * nyv;
    const double mag = stiffness * h.pen - damping * vn;
    if (mag <= 0.0) return out;
    out.fx = mag * nxv; out.fy = mag * nyv; out.tz = rxw * out.fy - ryw * out.fx;
    return out;
}

// everything a later block needs is re-read from LDS after this point instead of being carried in registers
#define FTGP_FORGET_REGISTERS() asm volatile("" ::: "memory")

// chassis circle k (body (contact_x[k], 0)) or wheel softener k (body (wheel_x[k], wheel_y[k])) of car `st` against the walls
__device__ __forceinline__ Force wall_term(const DeviceParams& P, const FtgpVehicle& v, const CarCore* st, int k, bool softener)
{
    WallHit h; double px, py;
    {
        const double qw = st->qw, qz = st->qz;
        const double ch = 1.0 - 2.0 * (qz * qz), sh = 2.0 * (qw * qz);
        const double rxw = softener ? ch * v.wheel_x[k] - sh * v.wheel_y[k] : ch * v.contact_x[k];
        const double ryw = softener ? sh * v.wheel_x[k] + ch * v.wheel_y[k] : sh * v.contact_x[k];
        px = st->x + rxw; py = st->y + ryw;
        // (the two bitmap pointers are wave-uniform: in scalar registers they do not count against the 64 vector registers)
        h = wall_search(P, uniform_ptr(P.bits), uniform_ptr(P.nearbits), px, py, softener ? v.softener_radius : v.contact_radius);
    }
    FTGP_FORGET_REGISTERS();
    Force t = { 0.0, 0.0, 0.0 };
    if (h.found) {
        const double qw = st->qw, qz = st->qz;
        const double ch = 1.0 - 2.0 * (qz * qz), sh = 2.0 * (qw * qz);
        const double rxw = softener ? ch * v.wheel_x[k] - sh * v.wheel_y[k] : ch * v.contact_x[k];
        const double ryw = softener ? sh * v.wheel_x[k] + ch * v.wheel_y[k] : sh * v.contact_x[k];
        t = wall_force(P, h, px, py, st->vx, st->vy, st->wz, rxw, ryw, v.contact_stiffness, v.contact_damping);
    }
    return t;
}

// Circles of this car against the circles of ONE other car of the env (penalty spring/damper, pre-step states): the sum of the nine
// circle pairs' forces, from +0, in the order i, j -- the specification adds the contacts with each env-mate up on their own and lets the
// mates' sums join the car's force in the order of the mates (round 5: that is what lets a car's four lanes take a mate each; rounds 1-4
// ran one sum over k, i, j on one lane, 27 dependent rounds of this loop body on the step's critical path).  The loop stays rolled and every
// operand is re-read from LDS where it is used, so that the only values carried around it are the running sum and a few indices.
__device__ __forceinline__ Force car_contact_mate(const FtgpVehicle& v, const CarCore* me, const CarCore* b)
{
    Force f = { 0.0, 0.0, 0.0 };
    #pragma unroll 1
    for (int ij = 0; ij < 9; ++ij) {
        FTGP_FORGET_REGISTERS();
        const int i = ij / 3, j = ij - 3 * i;
        const double r2 = 2.0 * v.contact_radius;
        double rxw, ryw, sxw, syw, ex, ey;
        {
            const double qw = me->qw, qz = me->qz;
            const double ch = 1.0 - 2.0 * (qz * qz), sh = 2.0 * (qw * qz);
            rxw = ch * v.contact_x[i]; ryw = sh * v.contact_x[i];
            const double bqw = b->qw, bqz = b->qz;
            const double cb = 1.0 - 2.0 * (bqz * bqz), sb = 2.0 * (bqw * bqz);
            sxw = cb * v.contact_x[j]; syw = sb * v.contact_x[j];
            const double px = me->x + rxw, py = me->y + ryw;
            const double qx = b->x + sxw, qy = b->y + syw;
            ex = px - qx; ey = py - qy;
        }
        const double d2 = ex * ex + ey * ey;
        if (d2 >= r2 * r2 || d2 <= 0.0) continue;
        const double d = sqrt(d2);
        const double nxv = ex / d, nyv = ey / d;
        const double swz = me->wz, bwz = b->wz;
        const double vax = me->vx - swz * ryw, vay = me->vy + swz * rxw;
        const double vbx = b->vx - bwz * syw, vby = b->vy + bwz * sxw;
        const double vn = (vax - vbx) * nxv + (vay - vby) * nyv;
        const double mag = v.contact_stiffness * (r2 - d) - v.contact_damping * vn;
        if (mag <= 0.0) continue;
        const double fx = mag * nxv, fy = mag * nyv;
        f.fx += fx; f.fy += fy; f.tz += rxw * fy - ryw * fx;
    }
    return f;
}

// LiDAR frame of a car at its current pose (lidar_car() of the oracle: centre, heading, binary32 pixel coordinates).
// Returns whether every ray of this car is KNOWN to start on the image: the car races, and its LiDAR centre is finite and at
// least edge_margin (ring radius in pixels + 2) away from every image edge.  When that holds for all cars of the workgroup the
// sweep initialises its rays without the on-image test (ftgp_ray_init, assume_inside).
__device__ __forceinline__ bool frame_write(const DeviceParams& P, const FtgpVehicle& v, const CarCore* st, LidarFrame* fr, int slot)
{
    const double qw = st->qw, qz = st->qz;
    const double ch = 1.0 - 2.0 * (qz * qz), sh = 2.0 * (qw * qz);
    const double lcx = st->x + (ch * v.lidar_x - sh * v.lidar_y);
    const double lcy = st->y + (sh * v.lidar_x + ch * v.lidar_y);
    const int finished = st->finished;
    const float u0 = (float)((lcx - P.origin_x) * P.inv_px_x), v0 = (float)((P.origin_y - lcy) * P.inv_px_y);
    fr->u0 = finished ? -INFINITY : u0;                       // -inf: rangefinders switched off (see lidar_groups)
    fr->v0 = v0;
    fr->chf = (float)ch; fr->shf = (float)sh;
    fr->lcx = lcx; fr->lcy = lcy;
    fr->x = st->x; fr->y = st->y; fr->qw = qw; fr->qz = qz;
    if (sgpr(P.lidar_mode) == FTGP_LIDAR_FAKELIDAR) {          // what lidar_fake() needs of the car: i_x, i_y (custom.py:1382-1384) and the heading, binary64
        fr->lcx = (st->x / P.map_size) * (double)P.width; fr->lcy = -(st->y / P.map_size) * (double)P.height;
        fr->x = ch; fr->y = sh;
    }
    fr->finished = finished;
    if (sgpr(P.cars_per_env) > 1) {       // what the inter-vehicle cull reads (this function sits on the driver -> dynamics chain: nothing it does not need)
        fr->slot0 = slot - slot % P.cars_per_env;
        fr->fx = (float)st->x; fr->fy = (float)st->y;
    }
    const float m = P.edge_margin;
    return !finished && u0 >= m && u0 <= (float)P.width - m && v0 >= m && v0 <= (float)P.height - m;      // false for a NaN
}

// The env-mate records of car c for the frames just written (see PairCull): one lane per (car, mate) pair.
__device__ __forceinline__ void pair_cull_write(const LidarFrame* frames, PairCull* pairs, int c, int k, int cars_per_env, float cull, float r0)
{
    const LidarFrame* me = frames + c;
    const int mate = me->slot0 + k;
    PairCull p; p.bx = 0.0f; p.by = 0.0f; p.t = INFINITY; p.pad = 0.0f;
    if (k < cars_per_env && mate != c && !me->finished && !frames[mate].finished) {
        p.bx = (float)(frames[mate].x - me->lcx); p.by = (float)(frames[mate].y - me->lcy);
        const float d2 = p.bx * p.bx + p.by * p.by, close = 2.0f * cull + r0;
        p.t = d2 <= close * close ? -INFINITY : sqrtf(d2 - cull * cull) * 0.9999f;
    }
    pairs[c * FTGP_PAIR_STRIDE + k] = p;
}

// Multi-car envs: which env-mates can the rays of a GROUP see?  The sweep tests a ray against a mate only if the mate lies inside the
// ray's cone (PairCull); most groups of 64 neighbouring rays look nowhere near any mate, and finding that out ray by ray -- a record
// read, a dot product, a compare and a wave-wide vote per mate and group -- was a third of the multi-car sweep.  So it is decided once
// per (car, group, mate), lane-parallel, by the wave that has just written the frames: a ray of the group lies within 32 ray
// spacings (10.7 degrees at 1080 rays; the bound below is computed for the actual fan spacing) of the group's middle ray d_g, so it can
// only pass its own test  B . d >= t  if  B . d_g >= cos(gamma) t - sin(gamma) cull  (cos of a sum of angles), gamma = that half-width
// plus what the 0.9999 in t and binary32 rounding can add.  One byte per (car, group): bit k = env-mate k may be visible.  A mask that
// says "may" too often costs time, never a result: the exact tests follow.
typedef const __attribute__((address_space(4))) int32_t* ScalarInts;
__device__ __forceinline__ void mate_masks(const LidarFrame* frames, const PairCull* pairs, unsigned char* mmask, int mmask_stride, ScalarInts order,
                                           const float2* __restrict__ ray_tab, int R, int tasks, int cars_per_env, float cull, float cg, float sg,
                                           int ncars_here, int first, int step)
{
    const int slots = 2 * tasks;
    for (int idx = first; idx < ncars_here * slots; idx += step) {
        const int c = idx / slots, s = idx - c * slots;
        const int ent = order[s >> 1], j0 = ent & 0xffff, kind = ent >> 16;
        uint32_t m = 0;
        if (kind == 2 || cg < -1.0f) m = (s & 1) ? 0u : 0xffu;                // both half-fans in one group, or a fan the bound does not cover: look at every mate
        else if (!(s & 1) || kind == 1) {
            const int jc = min(j0 + 32, (kind == 1 ? (R >> 1) : R) - 1);
            const float2 bd = ray_tab[jc];
            const LidarFrame* me = frames + c;
            float dxw = fmaf(me->chf, bd.x, -(me->shf * bd.y)), dyw = fmaf(me->shf, bd.x, me->chf * bd.y);
            if (s & 1) { dxw = -dxw; dyw = -dyw; }                                   // the opposite group
            for (int k = 0; k < cars_per_env; ++k) {
                const float4 q = *reinterpret_cast<const float4*>(pairs + c * FTGP_PAIR_STRIDE + k);
                const float al = fmaf(q.x, dxw, q.y * dyw);
                if (al >= fmaf(cg, q.z, -(sg * cull)) - 1e-4f) m |= 1u << k;          // (t = +inf: never -- myself, a finished car; t = -inf: always)
            }
        }
        mmask[c * mmask_stride + s] = (unsigned char)m;
    }
}

// One exchange of K3's argmin inside a quad of lanes: every lane sees the (distance, index) of the lane QUAD_PERM names and keeps the better pair
template <int QUAD_PERM>
__device__ __forceinline__ void quad_argmin_step(double& best, int& idx)
{
    const int lo = __double2loint(best), hi = __double2hiint(best);
    const double ob = __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, QUAD_PERM, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, QUAD_PERM, 0xf, 0xf, false));
    const int oi = __builtin_amdgcn_update_dpp(idx, idx, QUAD_PERM, 0xf, 0xf, false);
    if (ob < best || (ob == best && oi < idx)) { best = ob; idx = oi; }
}

// K1 + K3 for every car of the workgroup by ONE wave, four lanes per car (lane = 4 * car + r).
//   K1  lane r evaluates wheel r (fl, fr, bl, br), then wall-contact circle r (r < 3), then wheel softener r (bubble_wrap);
//       the force terms go to an LDS staging row and lane 0 of the car adds them up in the specification's order
//       (wheels 0..3, circles 0..2, softeners 0..3 -- a term that touches nothing is +0 and changes nothing, because the
//       running sum starts at +0 and can never become -0), adds the car-car contacts and integrates.  All lanes read the
//       pre-step states before any lane commits, so multi-car envs need no staging buffer for the states.
//   K3  25 centre-line points per lane, first minimum wins; steps += 1 happens between K1 and K3, as in custom.py:1425-1426
//       followed by the head of the next loop iteration.
// Then the LiDAR frames of the next step are written.
template <bool MULTI>
__device__ __forceinline__ void dynamics_lanes(const DeviceParams& P, const Lds& L, LidarFrame* next_frames, PairCull* next_pairs, unsigned char* next_mmask, ScalarInts order,
                                               int* unsafe_next, int ncars_here, int ci0)
{
    const int lane = lane_here();
    const int c = lane >> 2, r = lane & 3;
    const bool on = c < ncars_here;
    CarCore* st = L.cars + (on ? c : 0);
    Force* terms = L.terms + (on ? c : 0) * FTGP_FORCE_TERMS;
    const FtgpVehicle& v = L.veh->v;
    const bool finished = st->finished != 0;
    SUBSTAMP_BEGIN();
    {   // ---- wheel r
        const double qw = st->qw, qz = st->qz;
        const double ch = 1.0 - 2.0 * (qz * qz), sh = 2.0 * (qw * qz);
        const double vx = st->vx, vy = st->vy, wz = st->wz;
        // Ackermann coupling, mushr.em.xml:185-186: q + 0.375 q^2 + 0.140625 q^3 - 0.0722656 q^4 (fl), the odd signs flipped for fr
        const double q = st->qs;
        // (the odd coefficients by sign flips of the products: a negation is exact, and no 64-bit constant has to be selected per lane)
        const double c1 = (r == 0) ? 0.375 : -0.375;
        const double q4 = q * 0.0722656, t3 = (r == 0) ? -q4 : q4;
        const double steer = q * (1.0 + q * (c1 + q * (0.140625 + t3)));
        const bool tri = sgpr(v.kind) == FTGP_VEHICLE_TRICYCLE;         // legacy differential-drive car (car.em.xml): no steering, two driven wheels
        const double ang = (!tri && r < 2) ? steer : 0.0;               // wheels that do not steer: the polynomials give exactly (1, 0) at 0
        const double cwi = spec_cos(ang), swi = spec_sin(ang);
        double torque;
        if (!tri) {
            // velocity servo on the tendon = mean wheel spin, mushr.em.xml:180,191-196
            const double wbar = 0.25 * (((st->w[0] + st->w[1]) + st->w[2]) + st->w[3]);
            double fa = v.throttle_kv * (st->u_speed - v.throttle_gear * wbar);
            if (fa > v.throttle_force_limit) fa = v.throttle_force_limit;
            if (fa < -v.throttle_force_limit) fa = -v.throttle_force_limit;
            torque = (v.throttle_gear * 0.25) * fa;
        } else {
            // two torque motors on the tendons 0.5 (l + r) and 0.5 (r - l), controls clamped to their ctrlrange (car.em.xml:126-139)
            double uf = st->u_speed, ut = st->u_steer;
            if (uf > v.motor_forward_limit) uf = v.motor_forward_limit;
            if (uf < -v.motor_forward_limit) uf = -v.motor_forward_limit;
            if (ut > v.motor_turn_limit) ut = v.motor_turn_limit;
            if (ut < -v.motor_turn_limit) ut = -v.motor_turn_limit;
            torque = (r == 0) ? 0.5 * uf - 0.5 * ut : 0.5 * uf + 0.5 * ut;
        }
        const bool rolling = !(tri && r >= 2);                          // the caster is frictionless and there is no fourth wheel
        const double wi = st->w[r];
        const double wx_ = v.wheel_x[r], wy_ = v.wheel_y[r];
        const double rxw = ch * wx_ - sh * wy_;
        const double ryw = sh * wx_ + ch * wy_;
        const double vpx = vx - wz * ryw, vpy = vy + wz * rxw;
        const double fdx = ch * cwi - sh * swi, fdy = sh * cwi + ch * swi;
        const double vlong = (vpx * fdx + vpy * fdy) - v.wheel_radius * wi;
        const double vlat = vpy * fdx - vpx * fdy;
        double flong = -(v.tire_damping * vlong), flat = -(v.tire_damping * vlat);
        const double lim = v.friction * L.veh->wheel_load[r];
        const double m2 = flong * flong + flat * flat;
        if (m2 > lim * lim) { const double sc = lim / sqrt(m2); flong = flong * sc; flat = flat * sc; }
        Force t;
        t.fx = flong * fdx - flat * fdy; t.fy = flong * fdy + flat * fdx; t.tz = rxw * t.fy - ryw * t.fx;
        const double dt = P.dt;
        const double wn = (v.wheel_inertia * wi + dt * (torque - v.wheel_radius * flong)) / (v.wheel_inertia + dt * v.wheel_damping);
        if (!rolling) { t.fx = 0.0; t.fy = 0.0; t.tz = 0.0; }
        if (on) { terms[r] = t; L.wnew[(on ? c : 0) * 4 + r] = rolling ? wn : wi; }
    }
    FTGP_FORGET_REGISTERS();
    SUBSTAMP(0);
    {   // ---- wall-contact circle r and wheel softener r (a shadowed car collides with nothing, custom.py:1452-1457)
        Force t = { 0.0, 0.0, 0.0 };
        if (on && !finished && r < 3) t = wall_term(P, v, st, r, false);
        if (on && r < 3) terms[4 + r] = t;
        FTGP_FORGET_REGISTERS();
        Force u = { 0.0, 0.0, 0.0 };
        if (on && !finished && P.bubble_wrap) u = wall_term(P, v, st, r, true);        // custom.py:1041-1055, mushr.em.xml:65-67,126-129
        if (on) terms[7 + r] = u;
    }
    FTGP_FORGET_REGISTERS();
    SUBSTAMP(1);
    if (MULTI) {                     // ---- car-car contacts: lane r takes env-mates r and r + 4 (a shadowed car collides with nothing, custom.py:1452-1457)
        #pragma unroll 1
        for (int k = r; k < sgpr(P.cars_per_env); k += 4) {
            Force t = { 0.0, 0.0, 0.0 };
            const CarCore* b = st + (k - c % P.cars_per_env);          // env-mate k of this car's env (cars of an env are neighbours in the workgroup)
            if (on && !finished && b != st && !b->finished) t = car_contact_mate(v, st, b);
            if (on) L.mates[c * FTGP_PAIR_STRIDE + k] = t;
        }
        FTGP_FORGET_REGISTERS();
    }
    wave_lds_sync();                 // every lane has read the pre-step states; the force terms are staged
    SUBSTAMP(2);
    if (on && r == 0) {
        Force f = { 0.0, 0.0, 0.0 };
        #pragma unroll 1
        for (int k = 0; k < FTGP_FORCE_TERMS; ++k) { const Force t = terms[k]; f.fx += t.fx; f.fy += t.fy; f.tz += t.tz; }
        const double qw = st->qw, qz = st->qz;
        if (MULTI) {                 // the env-mates' contact sums, in the order of the mates (zeros where nothing touches)
            #pragma unroll 1
            for (int k = 0; k < P.cars_per_env; ++k) { const Force t = L.mates[c * FTGP_PAIR_STRIDE + k]; f.fx += t.fx; f.fy += t.fy; f.tz += t.tz; }
        }
        const double dt = P.dt;
        Dyn o;
        o.vx = st->vx + dt * (f.fx / v.mass);
        o.vy = st->vy + dt * (f.fy / v.mass);
        o.wz = st->wz + dt * (f.tz / v.izz);
        // position servo on the steering joint, implicit damping (mushr.em.xml:78,179)
        o.qsd = (v.steer_inertia * st->qsd + dt * (v.steer_kp * (st->u_steer - st->qs))) / (v.steer_inertia + dt * v.steer_damping);
        o.qs = st->qs + dt * o.qsd;
        if (o.qs > v.steer_limit) { o.qs = v.steer_limit; if (o.qsd > 0.0) o.qsd = 0.0; }
        if (o.qs < -v.steer_limit) { o.qs = -v.steer_limit; if (o.qsd < 0.0) o.qsd = 0.0; }
        if (v.kind == FTGP_VEHICLE_TRICYCLE) { o.qs = st->qs; o.qsd = st->qsd; }       // no steering joint
        // semi-implicit Euler: positions with the new velocities
        const double h = (0.5 * dt) * o.wz;
        const double chh = spec_cos(h), shh = spec_sin(h);
        const double nw = qw * chh - qz * shh, nz = qz * chh + qw * shh;
        const double n = sqrt(nw * nw + nz * nz);
        o.x = st->x + dt * o.vx;
        o.y = st->y + dt * o.vy;
        o.qw = nw / n; o.qz = nz / n;
        o.w[0] = L.wnew[c * 4]; o.w[1] = L.wnew[c * 4 + 1]; o.w[2] = L.wnew[c * 4 + 2]; o.w[3] = L.wnew[c * 4 + 3];
        L.next[c] = o;               // committed below, after every car of the workgroup has read its neighbours' pre-step states
    }
    wave_lds_sync();
    SUBSTAMP(3);
    {   // commit: the 13 doubles of the new dynamic state, spread over the car's four lanes
        if (on) {
            const double* src = reinterpret_cast<const double*>(L.next + c);
            double* dst = reinterpret_cast<double*>(st);
            for (int k = r; k < (int)(sizeof(Dyn) / sizeof(double)); k += 4) dst[k] = src[k];
            if (r == 0) L.steps[c] += 1;
        }
    }
    wave_lds_sync();
    SUBSTAMP(4);
    // ---- K3: distances = ((path - xpos)**2).sum(1); closest = distances.argmin() (first minimum), then the race-state update
    int idx; double best;
    bool upd = on && r == 0;             // the lane that updates the car's race state, its car slot and record
    int upd_c = c; CarCore* upd_st = st;
    if (ncars_here * 16 <= FTGP_WAVE) {
        // A small batch's workgroup holds at most four cars and its step IS the driver -> dynamics chain: sixteen lanes per car then walk seven
        // points each instead of four lanes twenty-five (lanes 14 and 15: two and none), and two more exchanges -- rotations by 8 and 4 inside the
        // row of 16 lanes -- bring every lane the car's first minimum.
        const int c16 = lane >> 4, q = lane & 15;
        const bool on16 = c16 < ncars_here;
        CarCore* st16 = L.cars + (on16 ? c16 : 0);
        const double x = st16->x, y = st16->y;
        const int first = q * 7, last = first + 7 < FTGP_PATH_POINTS ? first + 7 : FTGP_PATH_POINTS;
        idx = first < FTGP_PATH_POINTS ? first : 0x7fffffff; best = INFINITY;
        #pragma unroll 1
        for (int i = first; i < last; ++i) {
            const double dx = L.path[2 * i] - x, dy = L.path[2 * i + 1] - y;
            const double d = dx * dx + dy * dy;
            if (i == first || d < best) { best = d; idx = i; }
        }
        quad_argmin_step<0x128>(best, idx);      // row_ror:8
        quad_argmin_step<0x124>(best, idx);      // row_ror:4
        quad_argmin_step<0x4E>(best, idx);       // quad_perm [2,3,0,1]
        quad_argmin_step<0xB1>(best, idx);       // quad_perm [1,0,3,2]
        upd = on16 && q == 0; upd_c = c16; upd_st = st16;
    } else {
        const double x = st->x, y = st->y;
        idx = r * (FTGP_PATH_POINTS / 4);
        {
            const double dx = L.path[2 * idx] - x, dy = L.path[2 * idx + 1] - y;
            best = dx * dx + dy * dy;
        }
        #pragma unroll 1
        for (int i = idx + 1; i < (r + 1) * (FTGP_PATH_POINTS / 4); ++i) {
            const double dx = L.path[2 * i] - x, dy = L.path[2 * i + 1] - y;
            const double d = dx * dx + dy * dy;
            if (d < best) { best = d; idx = i; }
        }
        // the car's four lanes: exchanges inside the quad by DPP (quad_perm [1,0,3,2], then [2,3,0,1]), no LDS crossbar trip.
        // The lower index wins ties; a NaN never wins (the oracle's `d < best` is false for it too)
        quad_argmin_step<0xB1>(best, idx);
        quad_argmin_step<0x4E>(best, idx);
    }
    SUBSTAMP(5);
    if (upd) {
        Race rc; race_load(rc, upd_st);
        progress_update(P, rc, L.steps[upd_c], idx, best, P.cars[ci0 + upd_c].times, &upd_st->start);
        race_store(rc, upd_st);
    }
    wave_lds_sync();
    SUBSTAMP(6);
    // the LiDAR frames of the next step (its sweep starts after the workgroup barrier that ends this step)
    bool safe = true;
    if (on && r == 0) safe = frame_write(P, v, st, next_frames + c, c);
    const bool any_unsafe = __any(!safe);
    if (lane == 0) *unsafe_next = any_unsafe;
    SUBSTAMP(7);
    if (MULTI) {                     // env-mate records of the new frames: lane r of a car writes mates r and r + 4
        wave_lds_sync();
        const float cull = L.veh->cull_radius, r0f = (float)v.lidar_ring_radius;
        if (on) {
            pair_cull_write(next_frames, next_pairs, c, r, P.cars_per_env, cull, r0f);
            pair_cull_write(next_frames, next_pairs, c, r + 4, P.cars_per_env, cull, r0f);
        }
        wave_lds_sync();
        mate_masks(next_frames, next_pairs, next_mmask, P.mmask_stride, order, L.ray, P.n_rays, P.tasks_per_car, P.cars_per_env, cull, P.group_cg, P.group_sg, ncars_here, lane, FTGP_WAVE);
    }
}

// Inclusive prefix sum over the 64 lanes of a wave with data-parallel-primitive moves (no LDS round trips): three shifts
// inside each row of 16 lanes, two more across its banks, then the row totals broadcast down (lane 15 of rows 0 and 2 into
// rows 1 and 3, lane 31 into rows 2 and 3).  A lane that a step does not address adds 0.
__device__ __forceinline__ int wave_inclusive_sum(int x)
{
    const int s1 = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      // row_shr:1
    const int s2 = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);      // row_shr:2
    const int s3 = __builtin_amdgcn_update_dpp(0, x, 0x113, 0xf, 0xf, true);      // row_shr:3
    int v = x + s1 + s2 + s3;
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, true);                // row_shr:4, banks 1..3
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, true);                // row_shr:8, banks 2..3
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);                // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);                // row_bcast:31 into rows 2 and 3
    return v;
}

// Maximum / minimum over the 64 lanes of a wave, same moves, fused into the arithmetic: shifts by 1, 2, 4, 8 inside each row of
// 16 lanes leave a row's result in its lane 15, the two broadcasts carry it on; lane 63 holds the wave's.  A lane that a step
// does not address keeps its value.  (One asm statement each: the compiler emits move + DPP move + max per step; the waits
// are the two states a DPP read needs after the write of its source.)
#define FTGP_DPP_REDUCE(op) \
    "s_nop 1\n\t" op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_nop 1\n\t" op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
    "s_nop 1\n\t" op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
    "s_nop 1\n\t" op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
    "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
    "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
    "s_nop 1"
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x)
{
    asm volatile(FTGP_DPP_REDUCE("v_max_u32_dpp") : "+v"(x));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, FTGP_WAVE - 1);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x)
{
    asm volatile(FTGP_DPP_REDUCE("v_min_u32_dpp") : "+v"(x));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, FTGP_WAVE - 1);
}

// =============================================================================================
// K5: on-device drivers.  nidc.py:12-131 / fast.py:11-139 restated for one wave; the previous scan sits in LDS.
// =============================================================================================
// Number of points one disparity covers: ceil(2 * atan(width / (2 * close_dist)) / radians_per_point), nidc.py:57,93-99.
// close_dist is a binary32 sample and the count is a non-increasing step function of it, so the host tabulates -- with the
// very expression above, in binary64 with libm -- the largest sample thr[k] that still yields a count >= k (ftgp_create,
// build_cover_table).  The device then needs no atan: the count is the number of thresholds >= the sample.
// thr[0] holds the count for a sample of exactly 0 (the division by zero of the reference), as a float.
__device__ __forceinline__ int cover_count(const float* __restrict__ thr, int kmax, float close_dist, float half_width, float two_over_rpp)
{
    if (!(close_dist > 0.0f)) return close_dist == 0.0f ? (int)thr[0] : 0;      // negative (no hit) or NaN: nothing to cover
    // The answer is the largest k with thr[k] >= d (k = 0: none), and the table is the truth.  Seed k with the expression itself in
    // binary32 -- atan(x) ~ x / (1 + 0.28125 x^2), good to 0.005 rad for |x| <= 1 -- and walk to the answer: at most a step or two for
    // every sample a car can measure (a bisection of the table took eight rounds).  Any seed gives the same result.
    const float x = half_width * __builtin_amdgcn_rcpf(close_dist);
    const float a = x < 1.0f ? x * __builtin_amdgcn_rcpf(fmaf(0.28125f * x, x, 1.0f)) : 1.5707964f - __builtin_amdgcn_rcpf(x + 0.28125f * __builtin_amdgcn_rcpf(x));
    int k = (int)(a * two_over_rpp + 1.0f);
    k = k < 0 ? 0 : (k > kmax ? kmax : k);
    while (k < kmax && thr[k + 1] >= close_dist) ++k;
    while (k > 0 && thr[k] < close_dist) --k;
    return k;
}

// scan: [0] = ranges[0], [1 ...] = ranges[eighth : n - eighth] (the copy the reference makes, nidc.py:19, is this LDS row).
// Disparities are found on the unmodified scan (nidc.py:26-40) and extended in index order (nidc.py:86-105).  The cover
// counts (one atan each) are prepared for up to 64 disparities at once, one per lane; the ordered pass re-reads the two
// samples of each disparity and only recomputes the count when an earlier extension has changed them.
__device__ __forceinline__ void policy_disparity(const DriverShape& D, float* __restrict__ scan, CarCore* st, bool fast, int* __restrict__ list, const float* __restrict__ thr)
{
    const int lane = lane_here();
    const int n = D.n_rays;
    const double rpp = D.rpp;                                       // (2 pi) / n, nidc.py:121
    const int eighth = D.eighth;                                    // int(n / 8), nidc.py:18
    const int m = n - 2 * eighth;
    float* __restrict__ proc = scan + scan_window_first(eighth);
    const float range0 = scan[D.win_floats - 1];                    // ranges[0], fast.py:135
    const int kmax = D.cover_kmax;                                  // thr = the cover-count thresholds of this driver (width = (car_width / 2) * (1 + 300 / 100), nidc.py:93)
    const float half_width = fast ? 0.06f : 0.12f;                  // width / 2 = car_width (fast.py:4: 0.06, nidc.py:5: 0.12): seeds cover_count(), nothing else
    // Disparity flags (nidc.py:26-40, on the unmodified scan): lane l looks at samples l, 64 + l, 128 + l, ... (launch_steps()
    // guarantees m <= 64 * 64, i.e. K <= 64), so index order is (k, lane) order and a disparity's place in the ordered list is
    // the running total plus its rank among the flagged lanes of its k.  Disparities are rare: the common path of a k is two
    // LDS reads, a subtraction and ONE comparison (is anything at or above the threshold?); flags, ranks and list entries
    // are worked out only where that says yes.  Lanes past the end of the window (and sample 0, which has no predecessor)
    // read whatever lies there and are masked where it matters.
    // The reference compares |cur - prev| in binary64 with 0.6 (nidc.py:33).  In binary32, with T the float next above 0.6
    // (the nearest one): rounding is monotone, so |fl(cur - prev)| > T implies the exact difference is > 0.6 and < T implies it
    // is not; only a difference that rounds to exactly T needs the binary64 comparison (NaNs fail every test, as there).
    const int K = (m + FTGP_WAVE - 1) / FTGP_WAVE;
    SUBSTAMP_BEGIN();
    uint64_t mymask = 0;             // bit k: sample 64 k + lane is a disparity (needed again only beyond 64 disparities)
    int total = 0;                   // wave-uniform
    {
        const float T = 0.60000002384185791015625f;
        const float* __restrict__ p = proc + lane;                  // sample 64 k + lane is p[64 k]
        auto flagged = [&](int k, float cur, float prev, float ad) {    // some lane of group k is at or above the threshold
            const int i = k * FTGP_WAVE + lane;
            const bool valid = i >= 1 && i < m;
            bool fl = valid && ad > T;
            if (valid && ad == T) fl = fabs((double)cur - (double)prev) > 0.6;
            const uint64_t f = __builtin_amdgcn_ballot_w64(fl);
            if (fl) {
                const int r = total + rank_below(f);
                if (r < FTGP_WAVE) list[r] = i;
                mymask |= 1ull << k;
            }
            total += __popcll(f);
        };
        for (int k = 0; k < K; k += 2) {                            // two groups per trip: one address, two offsets
            const float cur0 = p[k * FTGP_WAVE], prev0 = p[k * FTGP_WAVE - 1];
            const float ad0 = fabsf(cur0 - prev0);
            if (__any(ad0 >= T)) flagged(k, cur0, prev0, ad0);
            if (k + 1 < K) {
                const float cur1 = p[(k + 1) * FTGP_WAVE], prev1 = p[(k + 1) * FTGP_WAVE - 1];
                const float ad1 = fabsf(cur1 - prev1);
                if (__any(ad1 >= T)) flagged(k + 1, cur1, prev1, ad1);
            }
        }
    }
    SUBSTAMP(8);
    for (int c0 = 0; c0 < total; c0 += FTGP_WAVE) {
        if (c0 > 0) {                // disparities c0 .. c0 + 63 of more than 64: their indices again, from the kept flags
            int run = 0;
            for (int k = 0; k < K; ++k) {
                const bool fl = ((mymask >> k) & 1ull) != 0;
                const uint64_t f = __builtin_amdgcn_ballot_w64(fl);
                if (fl) {
                    const int r = run + rank_below(f) - c0;
                    if (r >= 0 && r < FTGP_WAVE) list[r] = k * FTGP_WAVE + lane;
                }
                run += __popcll(f);
            }
        }
        wave_lds_sync();
        const int nchunk = total - c0 < FTGP_WAVE ? total - c0 : FTGP_WAVE;
        int index = 1; float q0 = 0.0f, q1 = 0.0f; int num = 0;
        if (lane < nchunk) {
            index = list[lane];
            q0 = proc[index - 1]; q1 = proc[index];
            num = cover_count(thr, kmax, (q1 < q0) ? q1 : q0, half_width, D.two_over_rpp);
        }
        wave_lds_sync();
        for (int d = 0; d < nchunk; ++d) {
            const int first = __builtin_amdgcn_readlane(index, d) - 1;
            const float p0v = proc[first], p1v = proc[first + 1];
            int nn = __builtin_amdgcn_readlane(num, d);
            const bool same = __float_as_uint(p0v) == (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(q0), d) &&
                              __float_as_uint(p1v) == (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(q1), d);
            const int close_idx = first + ((p1v < p0v) ? 1 : 0);   // argmin: first minimum
            const int far_idx = first + ((p1v > p0v) ? 1 : 0);     // argmax: first maximum
            const float ndf = (p1v < p0v) ? p1v : p0v;
            if (!__all(same)) nn = cover_count(thr, kmax, ndf, half_width, D.two_over_rpp);
            const bool cover_right = close_idx < far_idx;
            for (int i = lane; i < nn; i += FTGP_WAVE) {          // nidc.py:72-83, one target per lane
                const int idx = cover_right ? close_idx + 1 + i : close_idx - 1 - i;
                if (idx < 0 || idx >= m) break;
                if (proc[idx] > ndf) proc[idx] = ndf;
            }
            wave_lds_sync();
        }
    }
    wave_lds_sync();
    SUBSTAMP(9);
    // argmax, first maximum (nidc.py:127) = what the sequential loop "x > best" finds: every lane scans its own samples (l, 64 + l,
    // ...) in index order and remembers the GROUP of its best one (a wave-uniform number: no index arithmetic per sample); then
    // the wave's maximum, and the lowest index among the lanes that hold it.  Sample 0 is the running maximum to begin with,
    // whatever it is: a NaN there stays (like numpy's loop), NaNs elsewhere never win.
    float bv = -INFINITY; int bk = -1;
    if (lane == 0) { bv = proc[0]; bk = 0; }
    {
        const float* __restrict__ p = proc + lane;
        int k = 0;
        for (; k + 1 < K - 1; k += 2) {                             // groups before the last lie inside the window
            const float x0 = p[k * FTGP_WAVE], x1 = p[(k + 1) * FTGP_WAVE];
            if (x0 > bv) { bv = x0; bk = k; }
            if (x1 > bv) { bv = x1; bk = k + 1; }
        }
        if (k < K - 1) { const float x = p[k * FTGP_WAVE]; if (x > bv) { bv = x; bk = k; } }
        const float x = p[(K - 1) * FTGP_WAVE];
        if ((K - 1) * FTGP_WAVE + lane < m && x > bv) { bv = x; bk = K - 1; }
    }
    int bi;
    {
        // order-preserving integer image of the lane's maximum (-0 counts as +0, as it does for ">")
        const uint32_t b = __float_as_uint(bv + 0.0f);
        const uint32_t key = b ^ ((uint32_t)((int)b >> 31) | 0x80000000u);
        const uint32_t top = wave_max_u32(key);
        const uint32_t mine = (key == top && bk >= 0) ? (uint32_t)(bk * FTGP_WAVE + lane) : 0x7fffffffu;
        bi = (int)wave_min_u32(mine);
        const float b0 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bv)));
        if (b0 != b0) bi = 0;                                       // sample 0 is a NaN: nothing is greater
    }
    SUBSTAMP(10);
    double ang = ((double)bi - ((double)m / 2)) * rpp;              // nidc.py:112
    const double lim = 90.0 * (M_PI / 180.0);
    if (ang < -lim) ang = -lim;
    if (ang > lim) ang = lim;
    double speed, last = st->last_steer;
    if (!fast) {
        speed = 0.5 * 5 * (1 - fabs(ang) / (1.57 * 2));             // nidc.py:130
    } else {
        const double old = 0.0;                                     // fast.py:131-133
        ang = st->last_steer * old + ang * (1 - old);
        last = ang;
        if (fabs(ang) < 0.1 && (double)range0 > 0.5) speed = 7.0;   // fast.py:135-138
        else { const double sp = 0.5 * 5 * (1 - fabs(ang) / M_PI); speed = sp < 2.0 ? sp : 2.0; }
    }
    if (lane_id() == 0) { st->u_speed = speed; st->u_steer = ang; st->last_steer = last; }
    SUBSTAMP(11);
}

// evaluates the driver of car ci and stores the controls into its state record
__device__ __forceinline__ void policy_apply(const DeviceParams& P, const DriverShape& D, int policy, float* scan, CarCore* st, int ci, int64_t steps, int* list, const float* thr)
{
    const bool lane0 = lane_id() == 0;
    if (st->finished) {                                             // finished cars get the null driver (custom.py:1446)
        if (lane0) { st->u_speed = 0.0; st->u_steer = 0.0; }
        return;
    }
    switch (policy) {
    case FTGP_POLICY_LOBOTOMY: if (lane0) { st->u_speed = 0.0; st->u_steer = 0.0; } break;   // lobotomy.py:2-3
    case FTGP_POLICY_NIDC:
    case FTGP_POLICY_FAST: policy_disparity(D, scan, st, policy == FTGP_POLICY_FAST, list, thr); break;
    case FTGP_POLICY_RANDOM: {
        uint64_t h = splitmix64(P.seed + (uint64_t)((long)P.env_base * P.cars_per_env + ci) * 0x9E3779B97F4A7C15ull);
        h = splitmix64(h ^ (uint64_t)steps);
        if (lane0) { st->u_speed = 3.0 * u01(h); st->u_steer = 2.0 * u01(splitmix64(h)) - 1.0; }
        break; }
    default: break;
    }
}

// =============================================================================================
// End-of-launch metrics record (FTGP_METRIC_DOUBLES; the record ftgp_metrics_kernel computes from the state in HBM), folded into
// the step kernel so that a launch needs no second kernel and no second synchronisation: every workgroup reduces its own cars
// and publishes the partial record; the workgroup that arrives last adds the partials up.  The sums are integers, exact in
// binary64 in any order, so the result is bit-identical to the separate kernel's.  Hand-off between workgroups in the counter
// form of the CDNA guide with write-through stores: ONE lane stores its workgroup's record with agent-scope (sc1) stores, waits
// for them and draws a ticket (relaxed agent-scope fetch_add); the workgroup that draws the last ticket takes an agent-scope
// acquire fence, waits, passes a barrier and reads the records.  (A release fence per workgroup instead -- an L2 write-back
// each -- cost 14 us per launch, as much as the separate kernel.  Tickets in two levels -- eight per-XCD counters, then a top one --
// changed nothing: what this costs per launch -- 5 us since the reductions are DPP moves and the last workgroup reads the records with one
// thread each -- is three dependent trips to memory (the record's stores, the ticket, the last workgroup's read), not the 512 atomics on one word.)
__device__ __forceinline__ double wave_sum_f64(double v)
{
    #pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
    return v;
}

// The same reductions with data-parallel-primitive moves (no LDS round trips; the wave's result comes out of lane 63): the end-of-launch
// record is on every launch's critical tail, and six sums + a min + a max by __shfl_xor are 96 LDS crossbar trips.
// one step: every lane receives the value of the lane CTRL names; a lane the step does not address (or whose source lies outside the
// row) sees the identity: +0 for the sums (old = 0, bound_ctrl), its own value for min / max (old = itself)
template <int CTRL, int ROW_MASK, int BANK_MASK, bool KEEP_OWN>
__device__ __forceinline__ double dpp_moved_f64(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int mlo, mhi;
    if (KEEP_OWN) { mlo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, BANK_MASK, false); mhi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, BANK_MASK, false); }
    else          { mlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, true);   mhi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, true); }
    return __hiloint2double(mhi, mlo);
}
template <bool KEEP_OWN, class Op>
__device__ __forceinline__ double wave_reduce_f64_dpp(double x, Op op)
{
    double v = x;
    v = op(v, dpp_moved_f64<0x111, 0xf, 0xf, KEEP_OWN>(v));      // row_shr:1   -- after the four shifts lane 15 of every row holds the row's result
    v = op(v, dpp_moved_f64<0x112, 0xf, 0xf, KEEP_OWN>(v));      // row_shr:2
    v = op(v, dpp_moved_f64<0x114, 0xf, 0xf, KEEP_OWN>(v));      // row_shr:4
    v = op(v, dpp_moved_f64<0x118, 0xf, 0xf, KEEP_OWN>(v));      // row_shr:8
    v = op(v, dpp_moved_f64<0x142, 0xa, 0xf, KEEP_OWN>(v));      // row_bcast:15 into rows 1 and 3
    v = op(v, dpp_moved_f64<0x143, 0xc, 0xf, KEEP_OWN>(v));      // row_bcast:31 into rows 2 and 3
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), FTGP_WAVE - 1), __builtin_amdgcn_readlane(__double2loint(v), FTGP_WAVE - 1));
}
__device__ __forceinline__ double wave_total_f64(double x) { return wave_reduce_f64_dpp<false>(x, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ double wave_least_f64(double x) { return wave_reduce_f64_dpp<true>(x, [](double a, double b) { return fmin(a, b); }); }
__device__ __forceinline__ double wave_most_f64(double x) { return wave_reduce_f64_dpp<true>(x, [](double a, double b) { return fmax(a, b); }); }

// One rank and no communicator (HOST_SUM: bit 1 of the launch's slot argument): nothing on the device needs the launch's record, so every
// workgroup writes its partial record straight to pinned host memory -- [slot][workgroup][FTGP_METRIC_DOUBLES] -- and is done: no ticket, no
// last workgroup, no barrier; the host adds the partial records up when the record is asked for (collect_slot() in ftgp_api.hip: integers,
// exact in any order, bit-identical again).  That takes the hand-off's three dependent trips to memory out of every launch's tail.
// called by ALL threads of the workgroup, after the state records have gone back to HBM; `scratch`: one int of LDS
template <bool HOST_SUM>
__device__ __forceinline__ void launch_metrics(const DeviceParams& P, const CarCore* cars_lds, const int64_t* steps_lds, int ncars_here, int ci0,
                                               unsigned char* scratch, double* partial /* [waves][FTGP_METRIC_DOUBLES] of LDS */, int slot, int wave)
{
    int* last_flag = reinterpret_cast<int*>(scratch);
    const int lane = lane_id();
    if (HOST_SUM && wave != 0) return;
    if (wave == 0) {                       // wave 0: one car per lane (a workgroup holds at most 16)
        double v[7] = { 0, 0, 0, 0, 0, INFINITY, -INFINITY };      // steps, laps, absolute completion, finished, off track, min / max lap time
        if (lane < ncars_here) {
            const CarCore* a = cars_lds + lane;
            const int ci = ci0 + lane;
            const int lc = a->good_start ? a->completion : -(100 - a->completion);     // custom.py:132-143
            if (ci % P.cars_per_env == 0) v[0] = (double)steps_lds[lane];
            v[1] = a->laps; v[2] = a->laps * 100 + lc; v[3] = a->finished; v[4] = a->off_track;
            const int n = a->n_times < FTGP_MAX_LAP_TIMES ? a->n_times : FTGP_MAX_LAP_TIMES;      // the ring's occupied slots
            const double* t = P.cars[ci].times;
            for (int k = 0; k < n; ++k) { v[5] = fmin(v[5], t[k]); v[6] = fmax(v[6], t[k]); }
        }
        #pragma unroll
        for (int q = 0; q < 5; ++q) v[q] = wave_total_f64(v[q]);
        v[5] = wave_least_f64(v[5]); v[6] = wave_most_f64(v[6]);
        if (HOST_SUM) {
            if (lane == 0) {
                double* mine = P.wg_metrics_host + ((size_t)slot * gridDim.x + blockIdx.x) * FTGP_METRIC_DOUBLES;
                const double rec[FTGP_METRIC_DOUBLES] = { v[0], (double)ncars_here, v[1], v[2], v[3], v[4], v[5], v[6] };
                #pragma unroll
                for (int q = 0; q < FTGP_METRIC_DOUBLES; ++q) mine[q] = rec[q];
            }
            return;
        }
        if (lane == 0) {
            // the one lane that publishes also counts: write-through (agent-scope) stores of the record, wait for them, then the ticket
            double* mine = P.wg_metrics + (size_t)blockIdx.x * FTGP_METRIC_DOUBLES;
            const double rec[FTGP_METRIC_DOUBLES] = { v[0], (double)ncars_here, v[1], v[2], v[3], v[4], v[5], v[6] };
            #pragma unroll
            for (int q = 0; q < FTGP_METRIC_DOUBLES; ++q) __hip_atomic_store(mine + q, rec[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int ticket = __hip_atomic_fetch_add(P.wg_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *last_flag = ticket == gridDim.x - 1;
        }
    }
    __syncthreads();
    if (!*last_flag) return;                              // workgroup-uniform
    if (wave == 0 && lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // every thread of the last workgroup takes records (one thread per workgroup of the grid: the loads of a record are in flight together and
    // the whole read costs one memory latency), each wave reduces its lanes by DPP, wave 0 reduces the waves' results
    static_assert(sizeof(double) * 2 * FTGP_PATH_POINTS >= 16 * FTGP_METRIC_DOUBLES * sizeof(double), "the centre line's LDS bytes hold the waves' partial records");
    const int nwaves = (int)(blockDim.x >> 6);
    {
        double v[FTGP_METRIC_DOUBLES] = { 0, 0, 0, 0, 0, 0, INFINITY, -INFINITY };
        for (unsigned int b = (unsigned int)(wave * FTGP_WAVE + lane); b < gridDim.x; b += blockDim.x) {      // (threadIdx.x itself would be one more register held from the kernel's first instruction to here)
            const double* r = P.wg_metrics + (size_t)b * FTGP_METRIC_DOUBLES;
            #pragma unroll
            for (int q = 0; q < 6; ++q) v[q] += r[q];
            v[6] = fmin(v[6], r[6]); v[7] = fmax(v[7], r[7]);
        }
        if ((unsigned int)wave * FTGP_WAVE < gridDim.x) {                 // waves without a record have nothing to add (workgroup-uniform per wave)
            #pragma unroll
            for (int q = 0; q < 6; ++q) v[q] = wave_total_f64(v[q]);
            v[6] = wave_least_f64(v[6]); v[7] = wave_most_f64(v[7]);
        }
        if (lane == 0) {
            #pragma unroll
            for (int q = 0; q < FTGP_METRIC_DOUBLES; ++q) partial[wave * FTGP_METRIC_DOUBLES + q] = v[q];
        }
    }
    __syncthreads();
    if (wave == 0) {
        double v[FTGP_METRIC_DOUBLES] = { 0, 0, 0, 0, 0, 0, INFINITY, -INFINITY };
        if (lane < nwaves) {
            #pragma unroll
            for (int q = 0; q < FTGP_METRIC_DOUBLES; ++q) v[q] = partial[lane * FTGP_METRIC_DOUBLES + q];
        }
        #pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = wave_total_f64(v[q]);
        v[6] = wave_least_f64(v[6]); v[7] = wave_most_f64(v[7]);
        if (lane == 0) {
            #pragma unroll
            for (int q = 0; q < FTGP_METRIC_DOUBLES; ++q) { P.metrics_dev[slot * FTGP_METRIC_DOUBLES + q] = v[q]; P.metrics_host[slot * FTGP_METRIC_DOUBLES + q] = v[q]; }
            __hip_atomic_store(P.wg_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next launch counts from zero again
        }
    }
}

// =============================================================================================
// The step kernel.  Per step and car, in the order of the reference loop (custom.py:1337-1426):
//   driver(previous scan) -> ctrl -> [mj_step: sensors at the current pose, integrate] -> steps += 1 ->
//   progress at the new pose (= the head of the next loop iteration).
// =============================================================================================
__device__ __forceinline__ void stage16(void* dst, const void* src, int bytes)
{
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    const int n = bytes >> 4;
    for (int i = threadIdx.x; i < n; i += blockDim.x) d4[i] = s4[i];
}

#ifndef FTGP_WAVES_PER_EU
#define FTGP_WAVES_PER_EU 8       // two 16-wave workgroups per CU: at most 64 VGPRs per lane
#endif
// MULTI: several cars per env (inter-vehicle rays and contacts).  FAKE: FTGP_LIDAR_FAKELIDAR -- the sweep is lidar_fake() (its own
// instantiation, so that the rangefinder kernels' code generation does not move).  metrics_slot: which of the two record slots this
// launch's metrics go to (ftgp_metrics_allgather_begin / _end); bit 1: the workgroups' partial records go to pinned host memory (launch_metrics).
// ROSTER: `policy` may be FTGP_POLICY_PER_CAR (every car slot its own driver, ftgp_set_car_policies) -- again its own instantiation, so
// that the single-driver kernels stay exactly what they were (the multi-car one lost 3 % to the run-time form of this test); the
// FAKELIDAR kernels, which are for parity and not for throughput, exist in the ROSTER form only.
template <bool MULTI, bool FAKE, bool ROSTER>
__global__ void __launch_bounds__(1024, FTGP_WAVES_PER_EU) ftgp_step_kernel(const DeviceParams* __restrict__ Pg, int policy, int n_steps, int metrics_slot)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const DeviceParams& P0 = *reinterpret_cast<const DeviceParams*>(lds + Pg->off_params);   // valid after staging + barrier
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
    FTGP_DIAG_WG_ENTER();
    // The parameter block itself goes to LDS (later reads come from there, not from ~70 pinned SGPRs), with the vehicle constants, the centre
    // line, the fan and the driver's cover table: ONE image in HBM laid out as the LDS is (ftgp_create), so that a thread's loads are all in
    // flight together -- the launch pays one memory latency here, not one per table.
    {
        // (the cover table of the launch's driver; FTGP_POLICY_PER_CAR: both, nidc's first)
        const int head = Pg->off_cars - Pg->off_params;
        const int cover = (policy == FTGP_POLICY_NIDC || policy == FTGP_POLICY_FAST) ? Pg->stage_cover : (ROSTER && policy == FTGP_POLICY_PER_CAR) ? 2 * Pg->stage_cover : 0;
        const uint4* img = reinterpret_cast<const uint4*>(Pg->stage_img);
        const uint4* cov = img + ((head + (policy == FTGP_POLICY_FAST ? Pg->stage_cover : 0)) >> 4);
        uint4* d_head = reinterpret_cast<uint4*>(lds + Pg->off_params);
        uint4* d_cov = reinterpret_cast<uint4*>(lds + Pg->off_cover);
        const int nh = head >> 4, n = nh + (cover >> 4);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            if (i < nh) d_head[i] = img[i];
            else d_cov[i - nh] = cov[i - nh];
        }
    }
    if (threadIdx.x < 8) reinterpret_cast<int*>(lds + Pg->off_pool)[threadIdx.x] = 0;
    __syncthreads();
    const LdsOffsets off = lds_offsets(P0);
    const int cpb = sgpr(P0.cars_per_block);
    const int ci0 = FTGP_DIAG_WG_GROUP((int)blockIdx.x) * cpb;
    const int ncars_here = min(cpb, sgpr(P0.n_cars) - ci0);
    const bool second_half = (((int)blockIdx.x / max(1, sgpr(P0.n_cu))) & 1) != 0;      // see sweep_priority(): workgroups b and b + n_cu share a CU in the first dispatch wave
    const bool need_scan = (policy == FTGP_POLICY_NIDC || policy == FTGP_POLICY_FAST || (ROSTER && policy == FTGP_POLICY_PER_CAR));
    {
    const DeviceParams& P = P0;
    const Lds L = lds_view(off, lds);
    const int R = P.n_rays, eighth = P.eighth, win_floats = P.win_floats;
    if (need_scan) {     // the scan of the previous launch is what the first driver call sees: buffer 1 = parity of step -1.  All waves fetch it,
                         // a thread's loads (one per car) in flight together; the first driver call is on the launch's critical path
        const int nwin = R - 2 * eighth;
        for (int j = (int)threadIdx.x; j < nwin; j += (int)blockDim.x) {
            #pragma unroll 8
            for (int c = 0; c < ncars_here; ++c)
                L.scan[(cpb + c) * win_floats + scan_window_first(eighth) + j] = P.ranges[(size_t)(ci0 + c) * P.ranges_stride + eighth + j];
        }
        if ((int)threadIdx.x < ncars_here) L.scan[(cpb + (int)threadIdx.x) * win_floats + win_floats - 1] = P.ranges[(size_t)(ci0 + (int)threadIdx.x) * P.ranges_stride];
    }
    for (int c = wave; c < ncars_here; c += nwaves) {
        const int ci = ci0 + c;
        if (lane < (int)(sizeof(CarCore) / 4))
            reinterpret_cast<uint32_t*>(L.cars + c)[lane] = reinterpret_cast<const uint32_t*>(static_cast<const CarCore*>(&P.cars[ci]))[lane];
        if (lane == 0) L.steps[c] = P.steps[ci / P.cars_per_env];
        wave_lds_sync();
        if (lane == 0 && !frame_write(P, L.veh->v, L.cars + c, L.frame + c, c)) atomicOr(L.pool + 4, 1);
    }
    if (MULTI) {                         // env-mate records of the first frames
        __syncthreads();
        const int c = (int)threadIdx.x / FTGP_PAIR_STRIDE, k = (int)threadIdx.x % FTGP_PAIR_STRIDE;
        if (c < ncars_here) pair_cull_write(L.frame, L.pairs, c, k, P.cars_per_env, L.veh->cull_radius, (float)L.veh->v.lidar_ring_radius);
        __syncthreads();
        if (wave == 0)
            mate_masks(L.frame, L.pairs, L.mmask, P.mmask_stride, &scalar_view(Pg)->group_order[0], L.ray, P.n_rays, P.tasks_per_car, P.cars_per_env, L.veh->cull_radius, P.group_cg, P.group_sg,
                       ncars_here, lane_here(), FTGP_WAVE);
    }
    }
    __syncthreads();

    // One workgroup barrier per step.  Inside a step, concurrently:
    //   waves 0 .. cars-1  K5: driver of one car each on the scan of the PREVIOUS step (the other scan buffer) -> controls;
    //                      the wave whose driver finishes last runs K1 + K3 for all cars (one car per lane) and writes the
    //                      LiDAR frames of the NEXT step into the other frame buffer
    //   every wave         K2: the sweep of THIS step from the current frame buffer (as soon as its driver work is done)
    // The scan a driver sees lags the pose by one step (custom.py:1395-1425), which is what makes this legal: the sweep of
    // step t needs only the pose of step t, and that depends on the controls of step t-1.
    STAMP_DECLARE();
    for (int it = 0; it < n_steps; ++it) {
        STAMP(t0);
        // Each phase of a step fetches what it needs of the parameter block with scalar loads (scalar_view) and sees the LDS arrays
        // through fresh offsets: nothing of this is carried -- spilled -- from step to step or across the other phase.
        const int par = it & 1;
        if (wave < ncars_here) {         // drivers (and, for the wave that delivers last, dynamics)
            const ScalarParams G = scalar_view(Pg);
            const LdsOffsets off = lds_offsets(G);
            const DeviceParams& P = *reinterpret_cast<const DeviceParams*>(lds + off.params);
            const Lds L = lds_view(off, lds);
            const int win_floats = G->win_floats;
            LidarFrame* next_frames = L.frame + (par ^ 1) * cpb;
            float* scan_prev = L.scan + (par ^ 1) * cpb * win_floats;
            // the driver -> dynamics chain is the latency-critical path of a step and a small share of its instructions: let it issue first
            if (FTGP_DIAG_PRIO) __builtin_amdgcn_s_setprio(3);
            for (int c = wave; c < ncars_here; c += nwaves) {
                if (ROSTER && policy == FTGP_POLICY_PER_CAR) {
                    // the roster's drivers: car slot c of the workgroup has its own (wave-uniform: a scalar load); both cover tables are staged
                    const int pol = G->car_policy[c];
                    policy_apply(P, driver_shape(G), pol, scan_prev + c * win_floats, L.cars + c, ci0 + c, L.steps[c], L.list + wave * FTGP_WAVE,
                                 L.cover + (pol == FTGP_POLICY_FAST ? (G->stage_cover >> 2) : 0));
                }
                else if (policy != FTGP_POLICY_HOST) policy_apply(P, driver_shape(G), policy, scan_prev + c * win_floats, L.cars + c, ci0 + c, L.steps[c], L.list + wave * FTGP_WAVE, L.cover);
                wave_lds_sync();
                // "my controls are in LDS" -> the wave that counts last reads every car's controls: release on this side (the
                // fence of wave_lds_sync() + the RMW), acquire on the reader's (the RMW + the fence below), workgroup scope
                int n = 0;
                if (lane_here() == 0) n = __hip_atomic_fetch_add(L.pool + 2 + par, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                n = __builtin_amdgcn_readfirstlane(n);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                STAMP(t1); STAMP_ADD(0, t1 - t0);
                if (n == ncars_here - 1) {            // every driver of the workgroup has delivered its controls
                    if (lane_here() == 0) { L.pool[par ^ 1] = 0; L.pool[2 + (par ^ 1)] = 0; }     // the next step's counters (idle during this step)
                    if (FTGP_DIAG_RUN_K1)
                        dynamics_lanes<MULTI>(P, L, next_frames, L.pairs + (par ^ 1) * cpb * FTGP_PAIR_STRIDE, L.mmask + (par ^ 1) * cpb * G->mmask_stride, &G->group_order[0],
                                              L.pool + 4 + (par ^ 1), ncars_here, ci0);
                    STAMP(t2); STAMP_ADD(2, t2 - t1); STAMP_ADD(7, 1);
                }
            }
            if (FTGP_DIAG_PRIO) __builtin_amdgcn_s_setprio(0);
        }
        STAMP(t3);
        if (FTGP_DIAG_RUN_K2) {
            const ScalarParams G = scalar_view(Pg);
            const LdsOffsets off = lds_offsets(G);
            const DeviceParams& P = *reinterpret_cast<const DeviceParams*>(lds + off.params);
            const Lds L = lds_view(off, lds);
            if (FAKE) lidar_fake(G, L.frame + par * cpb, L.scan + par * cpb * G->win_floats, L.pool + par, ncars_here, ci0, need_scan);
            else lidar_groups<MULTI>(P, G, L, L.frame + par * cpb, L.pairs + par * cpb * FTGP_PAIR_STRIDE, L.mmask + par * cpb * G->mmask_stride, L.scan + par * cpb * G->win_floats,
                                     L.pool + par, ncars_here, ci0, need_scan, second_half STAMP_PASS);
        }
        STAMP(t4);
        __syncthreads();
        STAMP(t5);
        STAMP_ADD(3, t4 - t3); STAMP_ADD(4, t5 - t4); STAMP_ADD(5, 1); STAMP_ADD(6, t5 - t0);
    }

    STAMP_FLUSH();
    const DeviceParams& P = P0;
    const LdsOffsets off_end = lds_offsets(scalar_view(Pg));
    const Lds L = lds_view(off_end, lds);
    for (int c = wave; c < ncars_here; c += nwaves) {
        const int ci = ci0 + c;
        if (lane_here() < (int)(sizeof(CarCore) / 4))
            reinterpret_cast<uint32_t*>(static_cast<CarCore*>(&P.cars[ci]))[lane_here()] = reinterpret_cast<const uint32_t*>(L.cars + c)[lane_here()];
        if (lane_here() == 0 && ci % P.cars_per_env == 0) P.steps[ci / P.cars_per_env] = L.steps[c];
    }
    if (P.wg_metrics) {                  // the scan windows and the centre line are dead now: the reduction's scratch
        if (metrics_slot & 2) launch_metrics<true>(P, L.cars, L.steps, ncars_here, ci0, nullptr, nullptr, metrics_slot & 1, wave);      // (LDS state records: final since the last step's barrier)
        else {
            __syncthreads();
            launch_metrics<false>(P, L.cars, L.steps, ncars_here, ci0, lds + P.off_scan, reinterpret_cast<double*>(lds + P.off_path), metrics_slot, wave);
        }
    }
    FTGP_DIAG_WG_EXIT();
}

template __global__ void ftgp_step_kernel<false, false, false>(const DeviceParams*, int, int, int);
template __global__ void ftgp_step_kernel<true, false, false>(const DeviceParams*, int, int, int);
template __global__ void ftgp_step_kernel<false, false, true>(const DeviceParams*, int, int, int);
template __global__ void ftgp_step_kernel<true, false, true>(const DeviceParams*, int, int, int);
template __global__ void ftgp_step_kernel<false, true, true>(const DeviceParams*, int, int, int);
template __global__ void ftgp_step_kernel<true, true, true>(const DeviceParams*, int, int, int);

// K5 alone: one wave per car evaluates the driver on the scan stored in P.ranges (ftgp_policy_eval).
__global__ void __launch_bounds__(256) ftgp_policy_kernel(DeviceParams P, int policy, double* __restrict__ ctrl_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ci = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
    if (ci >= P.n_cars) return;
    const int win_floats = P.win_floats;
    CarCore* st = reinterpret_cast<CarCore*>(lds) + wave;
    int* list = reinterpret_cast<int*>(lds + 4 * sizeof(CarCore)) + wave * FTGP_WAVE;
    float* scan = reinterpret_cast<float*>(lds + 4 * sizeof(CarCore) + 4 * FTGP_WAVE * sizeof(int)) + wave * win_floats;
    const float* my_ranges = P.ranges + (size_t)ci * P.ranges_stride;
    if (lane == 0) scan[win_floats - 1] = my_ranges[0];
    for (int j = P.eighth + lane; j < P.n_rays - P.eighth; j += FTGP_WAVE) scan[scan_window_first(P.eighth) + j - P.eighth] = my_ranges[j];
    if (lane < (int)(sizeof(CarCore) / 4))
        reinterpret_cast<uint32_t*>(st)[lane] = reinterpret_cast<const uint32_t*>(static_cast<const CarCore*>(&P.cars[ci]))[lane];
    wave_lds_sync();
    const int pol = policy == FTGP_POLICY_PER_CAR ? P.car_policy[ci % P.cars_per_env] : policy;
    policy_apply(P, driver_shape(&P), pol, scan, st, ci, P.steps[ci / P.cars_per_env], list, P.cover_thr + (pol == FTGP_POLICY_FAST ? P.cover_kmax + 1 : 0));
    wave_lds_sync();
    if (lane == 0) {
        P.cars[ci].u_speed = st->u_speed; P.cars[ci].u_steer = st->u_steer; P.cars[ci].last_steer = st->last_steer;
        if (ctrl_out) { ctrl_out[2 * ci] = st->u_speed; ctrl_out[2 * ci + 1] = st->u_steer; }
    }
}

// =============================================================================================
// K4: reset / spawn (custom.py:1089-1128, 1232-1245, 81-87), one car per lane; then K3 at the spawn pose.
// =============================================================================================
__device__ __forceinline__ void progress_lane(const DeviceParams& P, CarCore& s, int64_t steps, double* __restrict__ times)
{
    double best = 0.0; int closest = 0;
    for (int i = 0; i < FTGP_PATH_POINTS; ++i) {
        const double dx = P.path[2 * i] - s.x, dy = P.path[2 * i + 1] - s.y;
        const double d = dx * dx + dy * dy;
        if (i == 0 || d < best) { best = d; closest = i; }
    }
    Race r; race_load(r, &s);
    progress_update(P, r, steps, closest, best, times, &s.start);
    race_store(r, &s);
}

__global__ void ftgp_reset_kernel(DeviceParams P, const uint8_t* __restrict__ env_mask)
{
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= P.n_cars) return;
    const int env = ci / P.cars_per_env, car = ci % P.cars_per_env;
    if (env_mask && !env_mask[env]) return;
    CarCore s;
    memset(&s, 0, sizeof s);
    const int p = (P.spawn_mode == 0) ? (car + 5) * 2 : (int)((10 + 7 * (long)(P.env_base + env) + 2 * car) % 98);   // custom.py:1112
    s.offset = p;
    s.good_start = 1;
    s.x = P.spawn[4 * p]; s.y = P.spawn[4 * p + 1];
    double qw = P.spawn[4 * p + 2], qz = P.spawn[4 * p + 3];
    if (P.spawn_mode == 1) {
        const uint64_t h = splitmix64(P.seed ^ (0xA0761D6478BD642Full + (uint64_t)((long)P.env_base * P.cars_per_env + ci)));
        const double j = 0.2 * u01(h) - 0.1;
        const double cj = spec_cos(0.5 * j), sj = spec_sin(0.5 * j);
        const double nw = qw * cj - qz * sj, nz = qz * cj + qw * sj;
        const double n = sqrt(nw * nw + nz * nz);
        qw = nw / n; qz = nz / n;
    }
    s.qw = qw; s.qz = qz;
    if (car == 0) P.steps[env] = 0;
    for (int k = 0; k < FTGP_MAX_LAP_TIMES; ++k) P.cars[ci].times[k] = 0.0;
    progress_lane(P, s, 0, P.cars[ci].times);
    static_cast<CarCore&>(P.cars[ci]) = s;
}

// sensordata = 0 after mj_resetData (custom.py:1092): coalesced zero fill of the reset envs' scans
__global__ void ftgp_zero_ranges_kernel(DeviceParams P, const uint8_t* __restrict__ env_mask)
{
    const int ci = blockIdx.x;
    if (env_mask && !env_mask[ci / P.cars_per_env]) return;
    float* r = P.ranges + (size_t)ci * P.ranges_stride;
    for (int j = threadIdx.x; j < P.ranges_stride; j += blockDim.x) r[j] = 0.0f;
}

__global__ void ftgp_progress_kernel(DeviceParams P)
{
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= P.n_cars) return;
    CarCore s = static_cast<const CarCore&>(P.cars[ci]);
    progress_lane(P, s, P.steps[ci / P.cars_per_env], P.cars[ci].times);
    static_cast<CarCore&>(P.cars[ci]) = s;
}

__global__ void ftgp_set_ctrl_kernel(DeviceParams P, const double* __restrict__ ctrl, const uint8_t* __restrict__ mask)
{
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= P.n_cars) return;
    if (mask && !mask[ci]) return;
    P.cars[ci].u_speed = ctrl[2 * ci];
    P.cars[ci].u_steer = ctrl[2 * ci + 1];
}

__global__ void ftgp_set_pose_kernel(DeviceParams P, const double* __restrict__ pose)
{
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= P.n_cars) return;
    const double* o = pose + (size_t)ci * FTGP_POSE_DOUBLES;
    const double n = sqrt(o[3] * o[3] + o[6] * o[6]);
    CarState& s = P.cars[ci];
    s.x = o[0]; s.y = o[1]; s.qw = o[3] / n; s.qz = o[6] / n; s.vx = o[7]; s.vy = o[8]; s.wz = o[12];
}

// Packed read-back rows, one car per lane: the host copies 3 small arrays instead of the whole state records.
//   prog  int32[n_cars][FTGP_PROGRESS_INTS]   (custom.py:91-143: lap_completion / absolute_completion folded in)
//   core  double[n_cars][16]: x y qw qz vx vy wz u_speed u_steer laps lap_completion absolute_completion steps n_times start finish_step (the last two: int64 bit patterns)
__global__ void ftgp_pack_kernel(DeviceParams P, int32_t* __restrict__ prog, double* __restrict__ core)
{
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= P.n_cars) return;
    const CarState& a = P.cars[ci];
    const int lc = a.good_start ? a.completion : -(100 - a.completion);         // custom.py:132-140
    int32_t* o = prog + (size_t)ci * FTGP_PROGRESS_INTS;
    o[0] = a.laps; o[1] = a.completion; o[2] = lc; o[3] = a.laps * 100 + lc; o[4] = a.finished;   // custom.py:142-143
    // the row is int32 while steps are int64: start and finish_step saturate at 2^31 - 1 here (ftgp_get_race_steps has all 64 bits)
    const int64_t top = 0x7fffffffll;
    o[5] = a.off_track; o[6] = (int32_t)(a.start > top ? top : a.start); o[7] = a.good_start; o[8] = a.delta;
    o[9] = a.finished ? (int32_t)(a.finish_step > top ? top : a.finish_step) : -1;
    double* d = core + (size_t)ci * 16;
    d[0] = a.x; d[1] = a.y; d[2] = a.qw; d[3] = a.qz; d[4] = a.vx; d[5] = a.vy; d[6] = a.wz; d[7] = a.u_speed; d[8] = a.u_steer;
    d[9] = a.laps; d[10] = lc; d[11] = a.laps * 100 + lc; d[12] = (double)P.steps[ci / P.cars_per_env]; d[13] = a.n_times;
    d[14] = __longlong_as_double(a.start); d[15] = __longlong_as_double(a.finished ? a.finish_step : -1ll);      // bit patterns of the two int64 (ftgp_get_race_steps)
}

// fakelidar-compat (raycast.py:5-21): one ray per lane, binary64, same operation order as the Python loop.
__global__ void ftgp_fakelidar_kernel(const double* __restrict__ dt, int H, int W, int n_rays_total, int R,
                                      const double* __restrict__ origins, const double* __restrict__ cosines, const double* __restrict__ sines,
                                      double eps, double* __restrict__ scan, double* __restrict__ points, int* __restrict__ index_error)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays_total) return;
    const int o = i / R;
    double x = origins[2 * o], y = origins[2 * o + 1];
    const double dx = cosines[i], dy = sines[i];
    double distance = 0.0;
    bool bad = false;
    long yi = (long)y, xi = (long)x;                          // int(): truncation toward zero
    if (yi < 0) yi += H;
    if (xi < 0) xi += W;                                      // numpy negative-index wrap
    double nearest = 0.0;
    if (yi < 0 || yi >= H || xi < 0 || xi >= W) bad = true; else nearest = dt[(size_t)yi * W + xi];
    for (int guard = 0; guard < (1 << 20) && !bad && nearest > eps && 0 <= x && x <= W && 0 <= y && y <= H; ++guard) {
        distance += nearest;
        x += dx * nearest;
        y += dy * nearest;
        yi = (long)y; xi = (long)x;
        if (yi < 0) yi += H;
        if (xi < 0) xi += W;
        if (yi < 0 || yi >= H || xi < 0 || xi >= W) { bad = true; break; }
        nearest = dt[(size_t)yi * W + xi];
    }
    if (bad) atomicOr(index_error, 1);
    scan[i] = distance; points[2 * i] = x; points[2 * i + 1] = y;
}

// Sector box field build (ftgp_create): one cell of one plane per lane, ring included.
__global__ void ftgp_box_field_kernel(const uint16_t* __restrict__ runx, const uint16_t* __restrict__ runy, int W, int H, int n_sectors, uint16_t* __restrict__ out)
{
    const size_t cells = (size_t)ftgp_plane256(W, H) * 128;          // 16-bit entries per (padded) plane
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells * (size_t)n_sectors) return;
    const int sector = (int)(i / cells);
    const size_t c = i - (size_t)sector * cells;
    const int X = (int)(c % (size_t)(W + 2)), Y = (int)(c / (size_t)(W + 2));
    uint32_t e = FTGP_FIELD_OUT;
    if (X >= 1 && X <= W && Y >= 1 && Y <= H) e = ftgp_box_entry(runx, runy, W, H, X - 1, Y - 1, sector, n_sectors / 8);
    out[i] = (uint16_t)e;
}

// Exact Euclidean distance transform (FTGP_LIDAR_FAKELIDAR): out[y][x] = distance from pixel (x, y) to the nearest wall pixel,
// centre to centre, 0 on walls -- scipy.ndimage.distance_transform_edt(~wall), the recipe of custom.py:1149-1153 / raycast.py:24-27.
// runy[d][y][x] = wall-free run from the pixel along +y (d = 0) / -y (d = 1), 0 on walls, 65535 = to the image edge, so
// g(x', y) = min of the two is the vertical distance to the nearest wall of column x'.  One pixel per lane walks its row outwards
// from x until (x - x')^2 alone reaches the best squared distance: integers throughout, then one correctly rounded sqrt.
__global__ void ftgp_edt_kernel(const uint16_t* __restrict__ runy, int W, int H, double* __restrict__ out)
{
    const size_t plane = (size_t)W * H;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    const int x = (int)(i % (size_t)W);
    const uint16_t* up = runy + (i - x), * down = runy + plane + (i - x);       // the pixel's row in both planes
    const long long INF = 1ll << 40;
    long long best = INF;
    for (int k = 0; k < W; ++k) {
        const long long k2 = (long long)k * k;
        if (k2 >= best) break;
        for (int sgn = 0; sgn < (k ? 2 : 1); ++sgn) {
            const int xp = sgn ? x + k : x - k;
            if (xp < 0 || xp >= W) continue;
            const int g = min((int)up[xp], (int)down[xp]);
            if (g == 65535) continue;                                            // no wall in that column
            const long long d2 = k2 + (long long)g * g;
            best = d2 < best ? d2 : best;
        }
    }
    out[i] = sqrt((double)best);
}

// Metrics record (FTGP_METRIC_DOUBLES): one block of 1024 threads (a handful of cars each: the records are 448 B apart, so the
// loads want to be in flight together); reduction inside each wave by shuffles, then across the 16 waves through LDS.  The
// sums are integers, exact in binary64 in any order.
#define FTGP_METRIC_THREADS 1024
__global__ void __launch_bounds__(FTGP_METRIC_THREADS) ftgp_metrics_kernel(DeviceParams P, double* __restrict__ out)
{
    __shared__ double part[7][FTGP_METRIC_THREADS / FTGP_WAVE];
    double steps = 0, laps = 0, absc = 0, fin = 0, off = 0, tmin = INFINITY, tmax = -INFINITY;
    for (int e = threadIdx.x; e < P.n_envs; e += blockDim.x) steps += (double)P.steps[e];
    for (int i = threadIdx.x; i < P.n_cars; i += blockDim.x) {
        const CarState& a = P.cars[i];
        const int lc = a.good_start ? a.completion : -(100 - a.completion);     // custom.py:132-143
        laps += a.laps; absc += a.laps * 100 + lc; fin += a.finished; off += a.off_track;
        const int n = a.n_times < FTGP_MAX_LAP_TIMES ? a.n_times : FTGP_MAX_LAP_TIMES;
        for (int k = 0; k < n; ++k) { tmin = fmin(tmin, a.times[k]); tmax = fmax(tmax, a.times[k]); }
    }
    const int lane = threadIdx.x & (FTGP_WAVE - 1), wave = threadIdx.x >> 6;
    steps = wave_sum_f64(steps); laps = wave_sum_f64(laps); absc = wave_sum_f64(absc); fin = wave_sum_f64(fin); off = wave_sum_f64(off);
    #pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { tmin = fmin(tmin, shfl_xor_f64(tmin, m)); tmax = fmax(tmax, shfl_xor_f64(tmax, m)); }
    if (lane == 0) { part[0][wave] = steps; part[1][wave] = laps; part[2][wave] = absc; part[3][wave] = fin; part[4][wave] = off; part[5][wave] = tmin; part[6][wave] = tmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r[7] = { 0, 0, 0, 0, 0, INFINITY, -INFINITY };
        for (int w = 0; w < FTGP_METRIC_THREADS / FTGP_WAVE; ++w) {
            for (int q = 0; q < 5; ++q) r[q] += part[q][w];
            r[5] = fmin(r[5], part[5][w]); r[6] = fmax(r[6], part[6][w]);
        }
        out[0] = r[0]; out[1] = (double)P.n_cars; out[2] = r[1]; out[3] = r[2];
        out[4] = r[3]; out[5] = r[4]; out[6] = r[5]; out[7] = r[6];
    }
}
