// ftgp_kernels.hip -- HIP kernels of the ft_grandprix hot path for gfx950 (CDNA4, wave64).
//
//   ftgp_step_kernel<MULTI, GF>   K5 driver -> K2 LiDAR sweep -> K1 integrate -> K3 lap progress, n_steps per launch.
//       One wave per car, up to 16 waves per workgroup, persistent over all steps of the launch.  Staged into LDS once
//       per launch with coalesced 16-B loads: the parameter block, vehicle constants, centre-line, ray table, each car's
//       state record and the scan window the driver reads.  The 64 lanes of a wave stride the car's rays (aligned 256-B
//       range stores, the only per-step HBM traffic besides scratch); contact candidates and the centre-line argmin are
//       spread over lanes and resolved with wave-level min/max reductions; the driver's disparity masks come from wave
//       ballots.  The march skips wall-free cells with one of two interchangeable structures (template flag GF):
//         GF = true  : flat per-pixel octant field read from L2 (default): per direction octant the wall-free square or 2:1
//                      rectangle of pixels AHEAD of the cell -- walls beside or behind a ray never shorten its jumps
//         GF = false : two-level grid over 8x8-pixel blocks staged in LDS (4-bit block distances, 4-bit pixel distances
//                      of non-empty blocks behind a rank table)
//       Both return the bits of the plain-DDA specification.
//   ftgp_policy_kernel    K5 alone (ftgp_policy_eval).
//   ftgp_reset_kernel     K4 reset / spawn (+ K3 at the spawn pose), one car per lane.
//   ftgp_progress_kernel  K3 alone (after ftgp_set_pose), one car per lane.
//   ftgp_fakelidar_kernel raycast.fakelidar restated, one ray per lane.
//   ftgp_metrics_kernel   per-GPU metrics record.
//
// Reference behaviour restated by each block is cited inline (paths relative to the reference repo).
// Arithmetic follows DESIGN.md operation by operation (-ffp-contract=off; fmaf only where the specification
// says "fma"), so results are bit-identical to the CPU oracle.
#include "ftgp_device.h"

// vehicle constants as staged into LDS (kept out of SGPRs: the step loop would otherwise pin ~80 of them)
struct VehLds { FtgpVehicle v; double wheel_load[4]; };

struct LdsView {
    const VehLds* veh;
    const uint8_t* fine;      // 32 B per non-empty block: 4-bit chessboard distance to the nearest wall pixel (0 = wall)
    const uint2* rank;        // per 32 blocks {non-empty bits, non-empty blocks before this word}
    const double* path;
    const uint8_t* coarse;    // 4 bits per block: chessboard distance in blocks to the nearest non-empty block (0 = non-empty)
    const float* ray_bx;
    const float* ray_by;
};

__host__ __device__ __forceinline__ int coarse_at(const DeviceParams& P, const LdsView& L, int bx, int by)
{
    const int q = by * P.nbx + bx;
    return (L.coarse[q >> 1] >> ((q & 1) << 2)) & 15;
}
// per-pixel distance nibble of pixel (cx, cy) inside the NON-EMPTY block (bx, by)
__host__ __device__ __forceinline__ int fine_at(const DeviceParams& P, const LdsView& L, int bx, int by, int cx, int cy)
{
    const uint2 r = L.rank[by * P.nwpr + (bx >> 5)];
    const int idx = (int)r.y + __builtin_popcount(r.x & ((1u << (bx & 31)) - 1u));
    const int n = ((cy & 7) << 3) | (cx & 7);
    return (L.fine[(idx << 5) + (n >> 1)] >> ((n & 1) << 2)) & 15;
}
// wall bit of pixel (cx, cy), which must lie inside the image
template <bool GF>
__device__ __forceinline__ bool wall_px(const DeviceParams& P, const LdsView& L, int cx, int cy);
__host__ __device__ __forceinline__ bool grid_wall(const DeviceParams& P, const LdsView& L, int cx, int cy)
{
    const int bx = cx >> 3, by = cy >> 3;
    if (coarse_at(P, L, bx, by) != 0) return false;
    return fine_at(P, L, bx, by, cx, cy) == 0;
}

template <> __device__ __forceinline__ bool wall_px<false>(const DeviceParams& P, const LdsView& L, int cx, int cy) { return grid_wall(P, L, cx, cy); }
template <> __device__ __forceinline__ bool wall_px<true>(const DeviceParams& P, const LdsView& L, int cx, int cy) { return (P.field[2 * (cy * P.width + cx)] & 127u) == 0; }
// no wall pixel within `reach` pixels (chessboard) of pixel (ix, iy)?
template <bool GF>
__device__ __forceinline__ bool far_from_walls(const DeviceParams& P, const LdsView& L, int ix, int iy, int reach)
{
    if (GF) {   // every stored rectangle contains the h x h forward square; four such squares cover the (2h-1)^2 pixels around the cell
        const uint32_t q = P.field[2 * (iy * P.width + ix)];
        const uint32_t m = min(min(q & 127u, (q >> 8) & 127u), min((q >> 16) & 127u, (q >> 24) & 127u));
        return (int)m > reach;
    }
    return coarse_at(P, L, ix >> 3, iy >> 3) >= ((reach + 7) >> 3) + 1;
}

// =============================================================================================
// K2: LiDAR
// =============================================================================================
// The specification of a ray is the plain cell-by-cell DDA (DESIGN.md "K2"): crossing times
// sX(b) = ((float)b - pu) * (1/du), sY(b) = ((float)b - pv) * (1/dv); x-step iff sX < sY (a tie steps in y);
// the range is |crossing time| of the step that enters the first wall pixel, 0 in a wall, -1 off the image.
// The march below returns the same bits while skipping wall-free cells:
//   * an empty 8x8 block with block distance c  -> the (2c-1)^2 blocks around it hold no wall;
//   * a pixel of a non-empty block with distance k -> the (2k-1)^2 pixels around it hold no wall;
//   in both cases the ray jumps to the far edge of that rectangle, and the coordinate of the other axis is the
//   number of its boundaries the specification says were crossed by then (sY(b) <= s after an x-jump,
//   sX(b) < s after a y-jump).  That count is floor(p + d*s) unless the landing point is within 2^-9 pixel of
//   a boundary; only then are the specification's comparisons evaluated (rounding errors are < 6e-4 pixel for
//   images up to 8192 pixels).
// Both axes are mirrored so that the ray always travels towards +x', +y' (x' = -x is exact in IEEE arithmetic and
// maps cell i to ~i, block b to ~b), which removes every direction-dependent select from the loop.
// ray_step() is one generic iteration written with selects only, so a wave can run several independent rays per
// lane inside one wave-uniform loop (instruction-level parallelism hides the LDS latency of the lookups).
struct Ray {
    float pum, pvm, dum, dvm, ivx, ivy;   // mirrored origin / direction / inverse direction (0 where the direction is 0)
    float s, result;
    int ix, iy, mx, my;                   // mirrored cell; mirror masks (0 or -1)
    int qshift;                           // bit offset of this ray's direction quadrant in an octant-field dword
    int dom;                              // 0: |du| >= |dv| (x-dominant), 1: y-dominant
    bool active;
};

__host__ __device__ __forceinline__ void ray_init(const DeviceParams& P, Ray& r, float pu, float pv, float du, float dv, bool valid)
{
    const int ix0 = (int)floorf(pu), iy0 = (int)floorf(pv);
    r.mx = du < 0.0f ? -1 : 0; r.my = dv < 0.0f ? -1 : 0;
    r.qshift = ((r.mx & 1) | ((r.my & 1) << 1)) << 3;
    r.dom = fabsf(du) >= fabsf(dv) ? 0 : 1;
    r.pum = r.mx ? -pu : pu; r.pvm = r.my ? -pv : pv;
    r.dum = fabsf(du); r.dvm = fabsf(dv);
    r.ivx = (du != 0.0f) ? fabsf(1.0f / du) : 0.0f;
    r.ivy = (dv != 0.0f) ? fabsf(1.0f / dv) : 0.0f;
    r.s = 0.0f; r.result = -1.0f;
    const bool inside = ix0 >= 0 && ix0 < P.width && iy0 >= 0 && iy0 < P.height;
    r.active = valid && inside;
    r.ix = (inside ? ix0 : 0) ^ r.mx; r.iy = (inside ? iy0 : 0) ^ r.my;
}

// Lookup of one generic iteration, split so that a wave can put the LDS reads of several rays in flight together:
//   stage 1: the rank word of the ray's block (non-empty bit + running count)            -> 1 LDS read
//   stage 2: EITHER the block-distance nibble (empty block) OR the pixel-distance nibble  -> 1 LDS read
struct Probe { int addr2, shift2; bool nonempty; };
// wall-free rectangle ahead of the cell, in cells: [ix, ix + kx) x [iy, iy + ky) in the mirrored frame; hit: the cell is a wall
struct Ahead { int kx, ky; bool hit; };

__host__ __device__ __forceinline__ int ray_rank_addr(const DeviceParams& P, const Ray& r)
{
    const int tx = r.ix ^ r.mx, ty = r.iy ^ r.my;                  // true pixel (always inside the image)
    return (ty >> 3) * P.nwpr + (tx >> 8);
}
__host__ __device__ __forceinline__ Probe ray_probe(const DeviceParams& P, const Ray& r, uint2 rk)
{
    const int tx = r.ix ^ r.mx, ty = r.iy ^ r.my;
    const int bx = tx >> 3, by = ty >> 3;
    Probe p;
    p.nonempty = (rk.x >> (bx & 31)) & 1u;
    const int idx = (int)rk.y + __builtin_popcount(rk.x & ((1u << (bx & 31)) - 1u));
    const int n = ((ty & 7) << 3) | (tx & 7);
    const int q = by * P.nbx + bx;
    // byte offsets relative to the start of the fine table / coarse table (both live in one LDS allocation)
    p.addr2 = p.nonempty ? (P.off_fine + (idx << 5) + (n >> 1)) : (P.off_coarse + (q >> 1));
    p.shift2 = ((p.nonempty ? n : q) & 1) << 2;
    return p;
}

// decode of the two-level grid lookup (nibble = pixel distance in a non-empty block, block distance in an empty one)
__host__ __device__ __forceinline__ Ahead ahead_from_grid(const Ray& r, const Probe& pb, unsigned byte2)
{
    const int k = (int)(byte2 >> pb.shift2) & 15;
    Ahead a;
    a.hit = pb.nonempty & (k == 0);
    a.kx = pb.nonempty ? k : ((((r.ix >> 3) + k) << 3) - r.ix);
    a.ky = pb.nonempty ? k : ((((r.iy >> 3) + k) << 3) - r.iy);
    return a;
}
// decode of one octant-field byte
__host__ __device__ __forceinline__ Ahead ahead_from_octant(const Ray& r, unsigned dword)
{
    const unsigned b = (dword >> r.qshift) & 255u;
    const int h = (int)(b & 127u);
    const int wide = (int)(b >> 7);                      // 1: 2h along the dominant axis
    Ahead a;
    a.hit = h == 0;
    a.kx = h << (wide & (r.dom ^ 1));
    a.ky = h << (wide & r.dom);
    return a;
}

// returns true when the landing point was too close to a pixel boundary to trust floor(): the caller then runs ray_fix()
__host__ __device__ __forceinline__ bool ray_step(const DeviceParams& P, Ray& r, const Ahead& ah,
                                                  int& t_out, int& cur_out, int& hi_out, bool& stepx_out, int& xhi_out, int& yhi_out)
{
    const bool hit = r.active & ah.hit;
    r.result = hit ? fabsf(r.s) : r.result;
    r.active = r.active & !hit;
    const int xhi = r.ix + ah.kx - 1;
    const int yhi = r.iy + ah.ky - 1;
    const float sX = (r.dum != 0.0f) ? ((float)(xhi + 1) - r.pum) * r.ivx : INFINITY;
    const float sY = (r.dvm != 0.0f) ? ((float)(yhi + 1) - r.pvm) * r.ivy : INFINITY;
    const bool stepx = sX < sY;
    const float s = stepx ? sX : sY;
    r.s = r.active ? s : r.s;
    const float tp = stepx ? r.pvm : r.pum, td = stepx ? r.dvm : r.dum;
    const int cur = stepx ? r.iy : r.ix, hi = stepx ? yhi : xhi;
    const float v = fmaf(td, s, tp);
    const float fl = floorf(v);
    int t = (int)fl;
    t = t < cur ? cur : (t > hi ? hi : t);
    const float frac = v - fl;
    t_out = t; cur_out = cur; hi_out = hi; stepx_out = stepx; xhi_out = xhi; yhi_out = yhi;
    return r.active & !(frac >= P.snap_eps && frac <= 1.0f - P.snap_eps);
}

// the specification's comparisons for a landing point within snap_eps of a boundary
__host__ __device__ __forceinline__ int ray_fix(const Ray& r, int t, int cur, int hi, bool stepx)
{
    const float tp = stepx ? r.pvm : r.pum, tinv = stepx ? r.ivy : r.ivx;
    const float Sa = ((float)t - tp) * tinv, Sb = ((float)(t + 1) - tp) * tinv;
    const bool ca = stepx ? (Sa <= r.s) : (Sa < r.s), cb = stepx ? (Sb <= r.s) : (Sb < r.s);
    const bool dec = (t > cur) & !ca;
    const bool inc = !dec & (t < hi) & cb;
    return t + (inc ? 1 : 0) - (dec ? 1 : 0);
}

__host__ __device__ __forceinline__ void ray_commit(const DeviceParams& P, Ray& r, int t, int cur, bool stepx, int xhi, int yhi)
{
    const bool tnz = stepx ? (r.dvm != 0.0f) : (r.dum != 0.0f);
    t = tnz ? t : cur;
    const int nix = stepx ? xhi + 1 : t, niy = stepx ? t : yhi + 1;
    const int nx = nix ^ r.mx, ny = niy ^ r.my;
    const bool inside = ((unsigned)nx < (unsigned)P.width) & ((unsigned)ny < (unsigned)P.height);
    const bool go = r.active & inside;
    r.ix = go ? nix : r.ix; r.iy = go ? niy : r.iy;               // an inactive ray keeps its last in-image cell (lookups stay in range)
    r.active = go;                                                 // leaving the image: result stays -1
}

// single ray (host harness, contact-free uses); lds = base of the staged LDS image
__host__ __device__ __forceinline__ float march_grid(const DeviceParams& P, const unsigned char* lds, float pu, float pv, float du, float dv)
{
    const uint2* rank = reinterpret_cast<const uint2*>(lds + P.off_rank);
    Ray r; ray_init(P, r, pu, pv, du, dv, true);
    for (int guard = 0; guard < 8192 && r.active; ++guard) {
        const uint2 rk = rank[ray_rank_addr(P, r)];
        const Probe pb = ray_probe(P, r, rk);
        int t, cur, hi, xhi, yhi; bool stepx;
        const bool near = ray_step(P, r, ahead_from_grid(r, pb, lds[pb.addr2]), t, cur, hi, stepx, xhi, yhi);
        if (near) t = ray_fix(r, t, cur, hi, stepx);
        ray_commit(P, r, t, cur, stepx, xhi, yhi);
    }
    return r.result;
}

// Ray against another car: chassis box (slab test) and LiDAR puck (circle), binary32.
__device__ __forceinline__ float ray_vs_car(const FtgpVehicle& v, const CarCore* b, double lcx, double lcy, float dxw, float dyw)
{
    const float r0 = (float)v.lidar_ring_radius;
    float best = INFINITY;
    const double bqw = b->qw, bqz = b->qz;
    const double cb = 1.0 - 2.0 * (bqz * bqz), sb = 2.0 * (bqw * bqz);
    const float relx = (float)(lcx - b->x), rely = (float)(lcy - b->y);
    const float ox = fmaf(dxw, -r0, relx), oy = fmaf(dyw, -r0, rely);
    const float cbf = (float)cb, sbf = (float)sb;
    const float lx = fmaf(cbf, ox, sbf * oy), ly = fmaf(cbf, oy, -(sbf * ox));
    const float ldx = fmaf(cbf, dxw, sbf * dyw), ldy = fmaf(cbf, dyw, -(sbf * dxw));
    {
        const float xmin = (float)v.box_xmin, xmax = (float)v.box_xmax, ymin = (float)v.box_ymin, ymax = (float)v.box_ymax;
        float tmin = -INFINITY, tmax = INFINITY; bool miss = false;
        if (ldx != 0.0f) {
            const float inv = 1.0f / ldx; const float t1 = (xmin - lx) * inv, t2 = (xmax - lx) * inv;
            tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
        } else if (lx < xmin || lx > xmax) miss = true;
        if (ldy != 0.0f) {
            const float inv = 1.0f / ldy; const float t1 = (ymin - ly) * inv, t2 = (ymax - ly) * inv;
            tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
        } else if (ly < ymin || ly > ymax) miss = true;
        if (!miss && tmax >= fmaxf(tmin, 0.0f)) {
            const float t = tmin > 0.0f ? tmin : 0.0f;
            if (t < best) best = t;
        }
    }
    {
        const float px = lx - (float)v.lidar_x, py = ly - (float)v.lidar_y;
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

// Full sweep of one car by one wave.  Rangefinder geometry: template/mushr.em.xml:98-117 -- ray j leaves
// the ring at centre - 0.03*dir_j, dir_j = R(yaw) * (sin phi_j, -cos phi_j); j = 0 is the rear, CCW.
// Values replace data.sensordata[vehicle_state.sensors] (custom.py:1395).
//
// Scheduling: the 64 lanes work through the car's rays in index order, but a "pass" does not wait for its slowest
// ray: as soon as FTGP_REFILL lanes are free, they take the next rays (rank among the free lanes via ballot/popcount,
// one batched ray_init for all of them) while the unfinished rays simply carry on.  On the headline workload this cuts
// the wave-iterations per car from 245 (sum of per-pass maxima) to ~170 (ideal 135); which lane marches which ray has no
// influence on any result.
// measured on MI355X (tools/sweep_refill.sh): flat optimum around 36-48 free lanes; 40 for single-car envs, 48 for multi-car
// envs (their refill block also runs the inter-vehicle tests); 64 would be the classic "wait for the slowest ray" pass
#ifndef FTGP_REFILL
#define FTGP_REFILL (MULTI ? 48 : 40)
#endif
#ifndef FTGP_SLOTS
#define FTGP_SLOTS 1
#endif
template <bool MULTI, bool GF>
__device__ __forceinline__ void lidar_sweep(const DeviceParams& P, const LdsView& L, const CarCore* st, float* __restrict__ out_global,
                                            float* __restrict__ out_lds, const CarCore* env_cars, int my_slot)
{
    const FtgpVehicle& v = L.veh->v;
    const double qw = st->qw, qz = st->qz;
    const double ch = 1.0 - 2.0 * (qz * qz), sh = 2.0 * (qw * qz);
    const double lcx = st->x + (ch * v.lidar_x - sh * v.lidar_y);
    const double lcy = st->y + (sh * v.lidar_x + ch * v.lidar_y);
    const float u0 = (float)((lcx - P.origin_x) * P.inv_px_x);
    const float v0 = (float)((P.origin_y - lcy) * P.inv_px_y);
    const float chf = (float)ch, shf = (float)sh;
    const float isx = P.inv_px_x_f, isy = P.inv_px_y_f;
    const float r0 = (float)v.lidar_ring_radius;
    const int R = P.n_rays;
    const int lane = lane_id();
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    const unsigned char* lds_base = reinterpret_cast<const unsigned char*>(L.veh) - P.off_veh;
    const uint2* rank = L.rank;
    // the pointers come out of the LDS parameter block: tell the compiler they are global (global_load / global_store, not flat)
    typedef const __attribute__((address_space(1))) uint32_t* global_u32;
    typedef __attribute__((address_space(1))) float* global_f32;
    const global_u32 field = (global_u32)P.field;
    const global_f32 out_g = (global_f32)out_global;
    const int W = P.width;

    // FTGP_SLOTS independent rays per lane (instruction-level parallelism: that many field loads in flight per wave)
    Ray ray[FTGP_SLOTS]; float dxw[FTGP_SLOTS], dyw[FTGP_SLOTS]; int j[FTGP_SLOTS];
    #pragma unroll
    for (int q = 0; q < FTGP_SLOTS; ++q) {
        ray[q].active = false; ray[q].result = -1.0f; ray[q].s = 0.0f; ray[q].qshift = 0; ray[q].dom = 0;
        ray[q].pum = ray[q].pvm = ray[q].dum = ray[q].dvm = ray[q].ivx = ray[q].ivy = 0.0f;
        ray[q].ix = ray[q].iy = ray[q].mx = ray[q].my = 0;
        dxw[q] = dyw[q] = 0.0f;
        j[q] = -1;           // the ray this slot is marching (or has just finished); -1: none
    }
    int next = 0;            // first ray not handed out yet (wave-uniform)
    for (int round = 0; round < 4 * 8192; ++round) {
        // ---- slots whose ray is finished: store its range, then take the next ray in index order
        int n_idle = 0; bool any_pending = false;
        #pragma unroll
        for (int q = 0; q < FTGP_SLOTS; ++q) {
            const bool idle = !ray[q].active;
            if (idle && j[q] >= 0) {
                float r = ray[q].result;
                if (MULTI) {
                    // conservative cull before the exact box / puck tests: every visible part of a car lies within 0.114 of
                    // its origin (chassis corner 0.1137, puck 0.0825), so a car whose origin is farther than 0.125 from the
                    // ray, behind its start, or beyond the wall hit cannot change the range.  Results are unaffected.
                    const float ox = (float)lcx - r0 * dxw[q], oy = (float)lcy - r0 * dyw[q];
                    for (int k = 0; k < P.cars_per_env; ++k) {
                        if (k == my_slot || env_cars[k].finished) continue;      // shadowed cars are invisible (custom.py:1441-1466)
                        const float wx = (float)env_cars[k].x - ox, wy = (float)env_cars[k].y - oy;
                        const float along = wx * dxw[q] + wy * dyw[q];
                        const float perp2 = (wx * wx + wy * wy) - along * along;
                        if (perp2 > 0.125f * 0.125f || along < -0.125f || (r >= 0.0f && along - 0.125f > r)) continue;
                        const float rc = ray_vs_car(v, env_cars + k, lcx, lcy, dxw[q], dyw[q]);
                        if (rc < INFINITY && (r < 0.0f || rc < r)) r = rc;
                    }
                }
                if (P.scan_full) {
                    out_lds[j[q]] = r;                                // whole row staged in LDS, flushed below
                } else {
                    out_g[j[q]] = r;                                  // no LDS room for the row: 4-byte stores, merged in L2
                    if (out_lds) {  // the on-device driver only reads ranges[0] and ranges[eighth : n - eighth]
                        if (j[q] == 0) out_lds[0] = r;
                        if (j[q] >= P.eighth && j[q] < R - P.eighth) out_lds[1 + j[q] - P.eighth] = r;
                    }
                }
                j[q] = -1;
            }
            const uint64_t idle_mask = __ballot(idle);
            if (next < R) {
                const int mine = next + __popcll(idle_mask & lanes_below);
                if (idle && mine < R) {
                    j[q] = mine;
                    const float bx = L.ray_bx[mine], by = L.ray_by[mine];
                    dxw[q] = fmaf(chf, bx, -(shf * by));
                    dyw[q] = fmaf(shf, bx, chf * by);
                    const float du = dxw[q] * isx;
                    const float dv = -(dyw[q] * isy);
                    const float pu = fmaf(du, -r0, u0);
                    const float pv = fmaf(dv, -r0, v0);
                    ray_init(P, ray[q], pu, pv, du, dv, true);
                }
                next += __popcll(idle_mask);
            }
            any_pending |= j[q] >= 0;
        }
        bool any_active = false;
        #pragma unroll
        for (int q = 0; q < FTGP_SLOTS; ++q) any_active |= ray[q].active;
        if (!__any(any_active)) {
            if (next >= R && !__any(any_pending)) break;     // nothing marching, nothing to store, nothing left to hand out
            continue;                                        // e.g. rays that started outside the image: store them and refill
        }
        // ---- march until enough slots are free to make a batched refill worthwhile (or, at the end, until all are done)
        const int want_free = (next < R) ? FTGP_REFILL * FTGP_SLOTS : FTGP_WAVE * FTGP_SLOTS;
        for (int guard = 0; guard < 8192; ++guard) {
            Ahead ah[FTGP_SLOTS];
            if (GF) {
                // flat octant field from L2: one dword per (pixel, dominant axis), no indirection; the byte of the ray's own
                // quadrant is the wall-free rectangle AHEAD of the cell (walls beside or behind the ray do not shorten the jump)
                unsigned w[FTGP_SLOTS];
                #pragma unroll
                for (int q = 0; q < FTGP_SLOTS; ++q) {
                    const int tx = ray[q].ix ^ ray[q].mx, ty = ray[q].iy ^ ray[q].my;
                    w[q] = field[2 * (ty * W + tx) + ray[q].dom];
                }
                #pragma unroll
                for (int q = 0; q < FTGP_SLOTS; ++q) ah[q] = ahead_from_octant(ray[q], w[q]);
            } else {
                #pragma unroll
                for (int q = 0; q < FTGP_SLOTS; ++q) {
                    const uint2 rk = rank[ray_rank_addr(P, ray[q])];
                    const Probe pb = ray_probe(P, ray[q], rk);
                    ah[q] = ahead_from_grid(ray[q], pb, lds_base[pb.addr2]);
                }
            }
            int n_free = 0;
            #pragma unroll
            for (int q = 0; q < FTGP_SLOTS; ++q) {
                int t, cur, hi, xhi, yhi; bool stepx;
                const bool near = ray_step(P, ray[q], ah[q], t, cur, hi, stepx, xhi, yhi);
                if (__any(near)) { const int tf = ray_fix(ray[q], t, cur, hi, stepx); t = near ? tf : t; }
                ray_commit(P, ray[q], t, cur, stepx, xhi, yhi);
                n_free += __popcll(__ballot(!ray[q].active));
            }
            if (n_free >= want_free) break;
        }
        (void)n_idle;
    }
    if (P.scan_full) {
        // the row goes to HBM as aligned 16-B-per-lane stores (rows start on 256-B boundaries)
        wave_lds_sync();
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(1))) f32x4* global_f32x4;
        const global_f32x4 dst4 = (global_f32x4)out_global;
        const f32x4* src4 = reinterpret_cast<const f32x4*>(out_lds);
        const int n4 = R >> 2;
        for (int i = lane; i < n4; i += FTGP_WAVE) dst4[i] = src4[i];
        for (int i = (n4 << 2) + lane; i < R; i += FTGP_WAVE) out_g[i] = out_lds[i];
    }
}

// =============================================================================================
// K3: lap progress (custom.py:1340-1372)
// =============================================================================================
struct Race { int32_t completion, laps, start, offset, good_start, finished, off_track, delta, n_times; double dist2; };

__device__ __forceinline__ void progress_update(const DeviceParams& P, Race& s, int64_t steps, int closest, double best, double* __restrict__ times)
{
    s.dist2 = best;                                       // custom.py:1343 (squared)
    s.off_track = best > 1.0;                             // custom.py:1344
    if (s.off_track) return;                              // custom.py:1345: progress frozen off-track
    const int completion = ((closest - s.offset) % 100 + 100) % 100;
    const int delta = completion - s.completion;
    s.delta = (((completion - s.completion + 50) % 100) + 100) % 100 - 50;
    if (abs(delta) > 90) {
        const double lap_time = (double)(steps - (int64_t)s.start) * P.dt;
        if (s.delta < 0) {                                // backwards across the line, custom.py:1352-1356
            s.laps -= 1;
            s.good_start = 0;
            if (s.n_times != 0) s.n_times -= 1;
        } else if (s.delta > 0) {                         // custom.py:1357-1366
            if (s.good_start) {
                if (s.n_times < FTGP_MAX_LAP_TIMES) times[s.n_times] = lap_time;
                s.n_times += 1;
                s.start = (int32_t)steps;
            }
            s.laps += 1;
            s.good_start = 1;
        }
    }
    if (s.laps >= P.lap_target) s.finished = 1;           // custom.py:1367-1370
    s.completion = completion;
}

__device__ __forceinline__ void race_load(Race& r, const CarCore* st)
{
    r.completion = st->completion; r.laps = st->laps; r.start = st->start; r.offset = st->offset;
    r.good_start = st->good_start; r.finished = st->finished; r.off_track = st->off_track; r.delta = st->delta;
    r.n_times = st->n_times; r.dist2 = st->dist2;
}
__device__ __forceinline__ void race_store(const Race& r, CarCore* st)
{
    st->completion = r.completion; st->laps = r.laps; st->start = r.start;
    st->good_start = r.good_start; st->finished = r.finished; st->off_track = r.off_track; st->delta = r.delta;
    st->n_times = r.n_times; st->dist2 = r.dist2;
}

// wave-cooperative: argmin over the 100 centre-line points (first minimum), then the race-state update
__device__ __forceinline__ void progress_wave(const DeviceParams& P, const double* __restrict__ path, CarCore* st, int64_t steps, double* __restrict__ times)
{
    const int lane = lane_id();
    const double x = st->x, y = st->y;
    // distances = ((path - xpos)**2).sum(1); closest = distances.argmin()
    double best = INFINITY; int idx = 0x7fffffff;
    #pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = lane + h * FTGP_WAVE;
        if (i < FTGP_PATH_POINTS) {
            const double dx = path[2 * i] - x, dy = path[2 * i + 1] - y;
            const double d = dx * dx + dy * dy;
            if (d < best) { best = d; idx = i; }
        }
    }
    #pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ob = shfl_xor_f64(best, m);
        const int oi = __shfl_xor(idx, m, FTGP_WAVE);
        if (ob < best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    Race r; race_load(r, st);
    progress_update(P, r, steps, idx, best, times);
    if (lane == 0) race_store(r, st);
}

// =============================================================================================
// K1: integrate one dt (reduced planar model of template/mushr.em.xml stepped by mj_step, custom.py:1425)
// =============================================================================================
struct Force { double fx, fy, tz; };
struct Dyn { double x, y, qw, qz, vx, vy, wz, qs, qsd, w[4]; };
static_assert(sizeof(Dyn) == 104 && offsetof(CarCore, w) == offsetof(Dyn, w) && offsetof(CarCore, qsd) == offsetof(Dyn, qsd), "Dyn must mirror the head of CarCore");

// Chassis circles against wall pixels: the (2nx+1) x (2ny+1) candidate cells of each circle are tested
// one per lane; the deepest penetration (ties: first in raster order) is picked by a wave reduction.
template <bool GF>
__device__ __forceinline__ void wall_contact(const DeviceParams& P, const LdsView& L, const Dyn& s, double ch, double sh, Force& f)
{
    const FtgpVehicle& v = L.veh->v;
    const int W = P.width, H = P.height;
    const double sx = P.px_size_x, sy = P.px_size_y;
    const double r = v.contact_radius;
    const int nx = (int)ceil(r * P.inv_px_x), ny = (int)ceil(r * P.inv_px_y);
    const int reach = (nx > ny ? nx : ny) + 1;
    const int wx = 2 * nx + 1, ncell = wx * (2 * ny + 1);
    const int lane = lane_id();
    #pragma unroll 1
    for (int k = 0; k < 3; ++k) {
        const double rxw = ch * v.contact_x[k], ryw = sh * v.contact_x[k];
        const double px = s.x + rxw, py = s.y + ryw;
        const double u = (px - P.origin_x) * P.inv_px_x, w = (P.origin_y - py) * P.inv_px_y;
        const int ix = (int)floor(u), iy = (int)floor(w);
        if (ix < 0 || ix >= W || iy < 0 || iy >= H) continue;
        if (far_from_walls<GF>(P, L, ix, iy, reach)) continue;
        double mypen = 0.0; int myc = 0x7fffffff;
        for (int base = 0; base < ncell; base += FTGP_WAVE) {
            const int c = base + lane;
            if (c < ncell) {
                const int cy = iy + (c / wx - ny), cx = ix + (c % wx - nx);
                if (cx >= 0 && cx < W && cy >= 0 && cy < H && wall_px<GF>(P, L, cx, cy)) {
                    const double x0 = P.origin_x + (double)cx * sx, x1 = x0 + sx;
                    const double y1 = P.origin_y - (double)cy * sy, y0 = y1 - sy;
                    const double qx = px < x0 ? x0 : (px > x1 ? x1 : px);
                    const double qy = py < y0 ? y0 : (py > y1 ? y1 : py);
                    const double ex = px - qx, ey = py - qy;
                    const double d2 = ex * ex + ey * ey;
                    if (d2 < r * r) {
                        const double pen = r - sqrt(d2);
                        if (pen > mypen) { mypen = pen; myc = c; }
                    }
                }
            }
        }
        #pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double op = shfl_xor_f64(mypen, m);
            const int oc = __shfl_xor(myc, m, FTGP_WAVE);
            if (op > mypen || (op == mypen && oc < myc)) { mypen = op; myc = oc; }
        }
        if (!(mypen > 0.0)) continue;
        // winner, recomputed wave-uniformly
        const int cy = iy + (myc / wx - ny), cx = ix + (myc % wx - nx);
        const double x0 = P.origin_x + (double)cx * sx, x1 = x0 + sx;
        const double y1 = P.origin_y - (double)cy * sy, y0 = y1 - sy;
        const double qx = px < x0 ? x0 : (px > x1 ? x1 : px);
        const double qy = py < y0 ? y0 : (py > y1 ? y1 : py);
        const double ex = px - qx, ey = py - qy;
        const double d = sqrt(ex * ex + ey * ey);
        double nxv, nyv;
        if (d > 0.0) { nxv = ex / d; nyv = ey / d; }
        else {
            const double mx = px - (x0 + 0.5 * sx), my = py - (y0 + 0.5 * sy);
            const double mm = sqrt(mx * mx + my * my);
            if (mm > 0.0) { nxv = mx / mm; nyv = my / mm; } else { nxv = 0.0; nyv = 0.0; }
        }
        const double vcx = s.vx - s.wz * ryw, vcy = s.vy + s.wz * rxw;
        const double vn = vcx * nxv + vcy * nyv;
        const double mag = v.contact_stiffness * mypen - v.contact_damping * vn;
        if (mag <= 0.0) continue;
        const double fx = mag * nxv, fy = mag * nyv;
        f.fx += fx; f.fy += fy; f.tz += rxw * fy - ryw * fx;
    }
}

// Circles of this car against the circles of the other cars of the env (penalty spring/damper).
__device__ __forceinline__ void car_contact(const DeviceParams& P, const LdsView& L, const Dyn& s, double ch, double sh,
                                            const CarCore* env_cars, int my_slot, Force& f)
{
    const FtgpVehicle& v = L.veh->v;
    const double r2 = 2.0 * v.contact_radius;
    if (env_cars[my_slot].finished) return;          // a shadowed car collides with nothing (custom.py:1452-1457)
    for (int k = 0; k < P.cars_per_env; ++k) {
        if (k == my_slot || env_cars[k].finished) continue;
        const CarCore* b = env_cars + k;
        const double bx = b->x, by = b->y, bqw = b->qw, bqz = b->qz, bvx = b->vx, bvy = b->vy, bwz = b->wz;
        const double cb = 1.0 - 2.0 * (bqz * bqz), sb = 2.0 * (bqw * bqz);
        for (int i = 0; i < 3; ++i) {
            const double rxw = ch * v.contact_x[i], ryw = sh * v.contact_x[i];
            const double px = s.x + rxw, py = s.y + ryw;
            const double vax = s.vx - s.wz * ryw, vay = s.vy + s.wz * rxw;
            for (int j = 0; j < 3; ++j) {
                const double sxw = cb * v.contact_x[j], syw = sb * v.contact_x[j];
                const double qx = bx + sxw, qy = by + syw;
                const double ex = px - qx, ey = py - qy;
                const double d2 = ex * ex + ey * ey;
                if (d2 >= r2 * r2 || d2 <= 0.0) continue;
                const double d = sqrt(d2);
                const double nxv = ex / d, nyv = ey / d;
                const double vbx = bvx - bwz * syw, vby = bvy + bwz * sxw;
                const double vn = (vax - vbx) * nxv + (vay - vby) * nyv;
                const double mag = v.contact_stiffness * (r2 - d) - v.contact_damping * vn;
                if (mag <= 0.0) continue;
                const double fx = mag * nxv, fy = mag * nyv;
                f.fx += fx; f.fy += fy; f.tz += rxw * fy - ryw * fx;
            }
        }
    }
}

// new dynamic state from the pre-step state in LDS (other cars of the env are read pre-step too)
// The result goes to `out` in LDS (written by lane 0): a by-value return of a non-inlined function would travel through
// scratch memory for all 64 lanes.  Dyn is layout-compatible with the head of CarCore.
template <bool MULTI, bool GF>
__device__ __attribute__((noinline)) void integrate(const DeviceParams& P, const LdsView& L, const CarCore* st, const CarCore* env_cars, int my_slot, Dyn* out)
{
    const FtgpVehicle& v = L.veh->v;
    const double dt = P.dt;
    Dyn s;
    s.x = st->x; s.y = st->y; s.qw = st->qw; s.qz = st->qz; s.vx = st->vx; s.vy = st->vy; s.wz = st->wz;
    s.qs = st->qs; s.qsd = st->qsd; s.w[0] = st->w[0]; s.w[1] = st->w[1]; s.w[2] = st->w[2]; s.w[3] = st->w[3];
    const double u_speed = st->u_speed, u_steer = st->u_steer;
    const double ch = 1.0 - 2.0 * (s.qz * s.qz), sh = 2.0 * (s.qw * s.qz);
    // Ackermann coupling, mushr.em.xml:185-186
    const double q = s.qs;
    const double dfl = q * (1.0 + q * (0.375 + q * (0.140625 + q * -0.0722656)));
    const double dfr = q * (1.0 + q * (-0.375 + q * (0.140625 + q * 0.0722656)));
    // velocity servo on the tendon = mean wheel spin, mushr.em.xml:180,191-196
    const double wbar = 0.25 * (((s.w[0] + s.w[1]) + s.w[2]) + s.w[3]);
    double fa = v.throttle_kv * (u_speed - v.throttle_gear * wbar);
    if (fa > v.throttle_force_limit) fa = v.throttle_force_limit;
    if (fa < -v.throttle_force_limit) fa = -v.throttle_force_limit;
    const double ta = (v.throttle_gear * 0.25) * fa;
    Force f = { 0.0, 0.0, 0.0 };
    Dyn o;
    // rolled on purpose (register pressure): the rear wheels evaluate the polynomials at 0, which gives exactly (1, 0)
    #pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        const double ang = (i == 0) ? dfl : ((i == 1) ? dfr : 0.0);
        const double cwi = spec_cos(ang), swi = spec_sin(ang);
        const double wi = (i == 0) ? s.w[0] : ((i == 1) ? s.w[1] : ((i == 2) ? s.w[2] : s.w[3]));
        const double wx_ = v.wheel_x[i], wy_ = v.wheel_y[i];
        const double rxw = ch * wx_ - sh * wy_;
        const double ryw = sh * wx_ + ch * wy_;
        const double vpx = s.vx - s.wz * ryw, vpy = s.vy + s.wz * rxw;
        const double fdx = ch * cwi - sh * swi, fdy = sh * cwi + ch * swi;
        const double vlong = (vpx * fdx + vpy * fdy) - v.wheel_radius * wi;
        const double vlat = vpy * fdx - vpx * fdy;
        double flong = -(v.tire_damping * vlong), flat = -(v.tire_damping * vlat);
        const double lim = v.friction * L.veh->wheel_load[i];
        const double m2 = flong * flong + flat * flat;
        if (m2 > lim * lim) { const double sc = lim / sqrt(m2); flong = flong * sc; flat = flat * sc; }
        const double fx = flong * fdx - flat * fdy, fy = flong * fdy + flat * fdx;
        f.fx += fx; f.fy += fy; f.tz += rxw * fy - ryw * fx;
        const double wn = (v.wheel_inertia * wi + dt * (ta - v.wheel_radius * flong)) / (v.wheel_inertia + dt * v.wheel_damping);
        if (i == 0) o.w[0] = wn; else if (i == 1) o.w[1] = wn; else if (i == 2) o.w[2] = wn; else o.w[3] = wn;
    }
    if (!st->finished) wall_contact<GF>(P, L, s, ch, sh, f);
    if (MULTI) car_contact(P, L, s, ch, sh, env_cars, my_slot, f);
    o.vx = s.vx + dt * (f.fx / v.mass);
    o.vy = s.vy + dt * (f.fy / v.mass);
    o.wz = s.wz + dt * (f.tz / v.izz);
    // position servo on the steering joint, implicit damping (mushr.em.xml:78,179)
    o.qsd = (v.steer_inertia * s.qsd + dt * (v.steer_kp * (u_steer - s.qs))) / (v.steer_inertia + dt * v.steer_damping);
    o.qs = s.qs + dt * o.qsd;
    if (o.qs > v.steer_limit) { o.qs = v.steer_limit; if (o.qsd > 0.0) o.qsd = 0.0; }
    if (o.qs < -v.steer_limit) { o.qs = -v.steer_limit; if (o.qsd < 0.0) o.qsd = 0.0; }
    // semi-implicit Euler: positions with the new velocities
    const double h = (0.5 * dt) * o.wz;
    const double chh = spec_cos(h), shh = spec_sin(h);
    const double nw = s.qw * chh - s.qz * shh, nz = s.qz * chh + s.qw * shh;
    const double n = sqrt(nw * nw + nz * nz);
    o.x = s.x + dt * o.vx;
    o.y = s.y + dt * o.vy;
    o.qw = nw / n; o.qz = nz / n;
    if (lane_id() == 0) *out = o;
}

// =============================================================================================
// K5: on-device drivers.  nidc.py:12-131 / fast.py:11-139 restated for one wave; the previous scan sits in LDS.
// =============================================================================================
__device__ __attribute__((noinline)) void policy_disparity(const DeviceParams& P, float* __restrict__ scan, CarCore* st, bool fast)
{
    const int lane = lane_id();
    const int n = P.n_rays;
    const double car_width = fast ? 0.06 : 0.12;                    // fast.py:4 / nidc.py:5
    const double rpp = (2 * M_PI) / (double)n;                      // nidc.py:121
    const int eighth = (int)((double)n / 8.0);                      // nidc.py:18
    const int m = n - 2 * eighth;
    float* __restrict__ proc = scan + (P.scan_full ? eighth : 1);    // nidc.py:19: ranges[eighth:-eighth] (the copy is the LDS image)
    const float range0 = scan[0];                                   // ranges[0], fast.py:135
    const double width = (car_width / 2) * (1 + 300.0 / 100);       // nidc.py:93
    // disparities on the UNMODIFIED scan (nidc.py:26-40): one ballot per 64 elements, parked in lane (pass)
    uint64_t mymask = 0;
    const int npass = (m + FTGP_WAVE - 1) / FTGP_WAVE;
    for (int p = 0; p < npass; ++p) {
        const int i = p * FTGP_WAVE + lane;
        bool flag = false;
        if (i >= 1 && i < m) flag = fabs((double)proc[i] - (double)proc[i - 1]) > 0.6;
        const uint64_t b = __ballot(flag);
        if (lane == (p & 63)) mymask = b;
        if ((p & 63) == 63 || p == npass - 1) {
            // flush this group of up to 64 ballots: extend the disparities in index order (nidc.py:86-105)
            const int p0 = p & ~63;
            for (int pp = p0; pp <= p; ++pp) {
                const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)mymask, pp & 63);
                const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(mymask >> 32), pp & 63);
                uint64_t mask = ((uint64_t)hi << 32) | lo;
                while (mask) {
                    const int bit = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int index = pp * FTGP_WAVE + bit;
                    const int first = index - 1;
                    wave_lds_sync();
                    const float p0v = proc[first], p1v = proc[first + 1];
                    const int close_idx = first + ((p1v < p0v) ? 1 : 0);   // argmin: first minimum
                    const int far_idx = first + ((p1v > p0v) ? 1 : 0);     // argmax: first maximum
                    const float ndf = (p1v < p0v) ? p1v : p0v;
                    const double close_dist = (double)ndf;
                    const double angle = 2 * atan(width / (2 * close_dist));   // nidc.py:57
                    const double cnt = ceil(angle / rpp);
                    const int num = (cnt > 2147483000.0) ? 2147483000 : (cnt < -2147483000.0 ? -2147483000 : (int)cnt);
                    const bool cover_right = close_idx < far_idx;
                    for (int i = lane; i < num; i += FTGP_WAVE) {          // nidc.py:72-83, one target per lane
                        const int idx = cover_right ? close_idx + 1 + i : close_idx - 1 - i;
                        if (idx < 0 || idx >= m) break;
                        if (proc[idx] > ndf) proc[idx] = ndf;
                    }
                }
            }
            mymask = 0;
        }
    }
    wave_lds_sync();
    // argmax, first maximum (nidc.py:127)
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < m; i += FTGP_WAVE) { const float x = proc[i]; if (bi == 0x7fffffff || x > bv) { bv = x; bi = i; } }
    #pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1) {
        const float ov = __shfl_xor(bv, mm, FTGP_WAVE);
        const int oi = __shfl_xor(bi, mm, FTGP_WAVE);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
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
}

// evaluates the driver of car ci and stores the controls into its state record
__device__ __forceinline__ void policy_apply(const DeviceParams& P, int policy, float* scan, CarCore* st, int ci, int64_t steps)
{
    const bool lane0 = lane_id() == 0;
    if (st->finished) {                                             // finished cars get the null driver (custom.py:1446)
        if (lane0) { st->u_speed = 0.0; st->u_steer = 0.0; }
        return;
    }
    switch (policy) {
    case FTGP_POLICY_LOBOTOMY: if (lane0) { st->u_speed = 0.0; st->u_steer = 0.0; } break;   // lobotomy.py:2-3
    case FTGP_POLICY_NIDC: policy_disparity(P, scan, st, false); break;
    case FTGP_POLICY_FAST: policy_disparity(P, scan, st, true); break;
    case FTGP_POLICY_RANDOM: {
        uint64_t h = splitmix64(P.seed + (uint64_t)((long)P.env_base * P.cars_per_env + ci) * 0x9E3779B97F4A7C15ull);
        h = splitmix64(h ^ (uint64_t)steps);
        if (lane0) { st->u_speed = 3.0 * u01(h); st->u_steer = 2.0 * u01(splitmix64(h)) - 1.0; }
        break; }
    default: break;
    }
}

// =============================================================================================
// LDS staging
// =============================================================================================
__device__ __forceinline__ void stage16(void* dst, const void* src, int bytes)
{
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    const int n = bytes >> 4;
    for (int i = threadIdx.x; i < n; i += blockDim.x) d4[i] = s4[i];
}

__device__ __forceinline__ LdsView stage_track(const DeviceParams& P, unsigned char* lds)
{
    stage16(lds + P.off_params, &P, P.off_veh - P.off_params);   // the parameter block itself: later reads come from LDS, not from ~70 pinned SGPRs
    stage16(lds + P.off_veh, P.veh_dev, P.off_fine - P.off_veh);
    if (!P.use_field) {
        stage16(lds + P.off_fine, P.fine, P.off_rank - P.off_fine);
        stage16(lds + P.off_rank, P.rank, P.off_path - P.off_rank);
        stage16(lds + P.off_coarse, P.coarse, P.off_ray - P.off_coarse);
    }
    stage16(lds + P.off_path, P.path, P.off_coarse - P.off_path);
    stage16(lds + P.off_ray, P.ray_bx, P.off_state - P.off_ray);    // ray_bx and ray_by are one allocation
    LdsView L;
    L.veh = reinterpret_cast<const VehLds*>(lds + P.off_veh);
    L.fine = lds + P.off_fine;
    L.rank = reinterpret_cast<const uint2*>(lds + P.off_rank);
    L.path = reinterpret_cast<const double*>(lds + P.off_path);
    L.coarse = lds + P.off_coarse;
    L.ray_bx = reinterpret_cast<const float*>(lds + P.off_ray);
    L.ray_by = L.ray_bx + P.ray_floats;
    return L;
}

// =============================================================================================
// The step kernel.  Per step and car, in the order of the reference loop (custom.py:1337-1426):
//   driver(previous scan) -> ctrl -> [mj_step: sensors at the current pose, integrate] -> steps += 1 ->
//   progress at the new pose (= the head of the next loop iteration).
// =============================================================================================
#ifndef FTGP_MAX_THREADS
#define FTGP_MAX_THREADS 1024     // 16 waves per workgroup -> at most 128 VGPRs per lane
#endif
#ifndef FTGP_MIN_WAVES
#define FTGP_MIN_WAVES 1
#endif
template <bool MULTI, bool GF>
__global__ void __launch_bounds__(FTGP_MAX_THREADS, FTGP_MIN_WAVES) ftgp_step_kernel(const DeviceParams* __restrict__ Pg, int policy, int n_steps, int cars_per_block)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const DeviceParams& P = *reinterpret_cast<const DeviceParams*>(lds + Pg->off_params);   // valid after stage_track + barrier
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ci = __builtin_amdgcn_readfirstlane((int)blockIdx.x * cars_per_block + wave);
    const LdsView L = stage_track(*Pg, lds);
    __syncthreads();
    const bool live = ci < P.n_cars;
    const int env = live ? ci / P.cars_per_env : 0;
    const int my_slot = MULTI ? wave % P.cars_per_env : 0;
    const bool need_scan = (policy == FTGP_POLICY_NIDC || policy == FTGP_POLICY_FAST);
    const int scan_floats = P.scan_floats;
    CarCore* states = reinterpret_cast<CarCore*>(lds + P.off_state);
    CarCore* st = states + wave;
    const CarCore* env_cars = states + (wave - my_slot);
    float* scan = (need_scan || P.scan_full) ? reinterpret_cast<float*>(lds + P.off_scan) + wave * scan_floats : nullptr;

    int64_t steps = 0;
    float* my_ranges = nullptr;
    if (live) {
        if (lane < (int)(sizeof(CarCore) / 4))
            reinterpret_cast<uint32_t*>(st)[lane] = reinterpret_cast<const uint32_t*>(static_cast<const CarCore*>(&P.cars[ci]))[lane];
        steps = P.steps[env];
        my_ranges = P.ranges + (size_t)ci * P.ranges_stride;
        if (need_scan) {
            if (P.scan_full) {
                for (int j = lane; j < P.n_rays; j += FTGP_WAVE) scan[j] = my_ranges[j];
            } else {
                if (lane == 0) scan[0] = my_ranges[0];
                for (int j = P.eighth + lane; j < P.n_rays - P.eighth; j += FTGP_WAVE) scan[1 + j - P.eighth] = my_ranges[j];
            }
        }
    }
    __syncthreads();

    for (int it = 0; it < n_steps; ++it) {
        if (live) {
            if (policy != FTGP_POLICY_HOST) {
                policy_apply(P, policy, scan, st, ci, steps);
                wave_lds_sync();
            }
#ifndef FTGP_ABLATE_K2
            lidar_sweep<MULTI, GF>(P, L, st, my_ranges, scan, env_cars, my_slot);   // sensors at the pre-integration pose
#endif
        }
        // single-car envs: nobody else reads this record, K1 commits in place; multi-car: into a staging slot, committed
        // after every car of the env has read the pre-step states
        Dyn* nxt = MULTI ? reinterpret_cast<Dyn*>(lds + P.off_next) + wave : reinterpret_cast<Dyn*>(st);
#ifndef FTGP_ABLATE_K1
        if (live) integrate<MULTI, GF>(P, L, st, env_cars, my_slot, nxt);
#endif
        if (MULTI) __syncthreads();          // every car of the env has read the pre-step states
        if (live) {
            if (MULTI) {
                wave_lds_sync();
                if (lane < (int)(sizeof(Dyn) / 4)) reinterpret_cast<uint32_t*>(st)[lane] = reinterpret_cast<const uint32_t*>(nxt)[lane];
            }
            wave_lds_sync();
            steps += 1;
#ifndef FTGP_ABLATE_K3
            progress_wave(P, L.path, st, steps, P.cars[ci].times);
#endif
            wave_lds_sync();
        }
        if (MULTI) __syncthreads();          // new states visible before the next step reads them
    }
    if (live) {
        wave_lds_sync();
        if (lane < (int)(sizeof(CarCore) / 4))
            reinterpret_cast<uint32_t*>(static_cast<CarCore*>(&P.cars[ci]))[lane] = reinterpret_cast<const uint32_t*>(st)[lane];
        if (lane == 0 && ci % P.cars_per_env == 0) P.steps[env] = steps;
    }
}

// K5 alone: one wave per car evaluates the driver on the scan stored in P.ranges (ftgp_policy_eval).
__global__ void __launch_bounds__(256) ftgp_policy_kernel(DeviceParams P, int policy, double* __restrict__ ctrl_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ci = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
    if (ci >= P.n_cars) return;
    const int scan_floats = P.scan_floats;
    CarCore* st = reinterpret_cast<CarCore*>(lds) + wave;
    float* scan = reinterpret_cast<float*>(lds + 4 * sizeof(CarCore)) + wave * scan_floats;
    const float* my_ranges = P.ranges + (size_t)ci * P.ranges_stride;
    if (P.scan_full) {
        for (int j = lane; j < P.n_rays; j += FTGP_WAVE) scan[j] = my_ranges[j];
    } else {
        if (lane == 0) scan[0] = my_ranges[0];
        for (int j = P.eighth + lane; j < P.n_rays - P.eighth; j += FTGP_WAVE) scan[1 + j - P.eighth] = my_ranges[j];
    }
    if (lane < (int)(sizeof(CarCore) / 4))
        reinterpret_cast<uint32_t*>(st)[lane] = reinterpret_cast<const uint32_t*>(static_cast<const CarCore*>(&P.cars[ci]))[lane];
    wave_lds_sync();
    policy_apply(P, policy, scan, st, ci, P.steps[ci / P.cars_per_env]);
    wave_lds_sync();
    if (lane == 0) {
        P.cars[ci].u_speed = st->u_speed; P.cars[ci].u_steer = st->u_steer; P.cars[ci].last_steer = st->last_steer;
        if (ctrl_out) { ctrl_out[2 * ci] = st->u_speed; ctrl_out[2 * ci + 1] = st->u_steer; }
    }
}

template __global__ void ftgp_step_kernel<false, false>(const DeviceParams*, int, int, int);
template __global__ void ftgp_step_kernel<true, false>(const DeviceParams*, int, int, int);
template __global__ void ftgp_step_kernel<false, true>(const DeviceParams*, int, int, int);
template __global__ void ftgp_step_kernel<true, true>(const DeviceParams*, int, int, int);

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
    progress_update(P, r, steps, closest, best, times);
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

// Octant field build (ftgp_create): one pixel per lane.  ksq[q] = side of the largest wall-free square ahead of the pixel in
// quadrant q (host recurrence); run*[d] = wall-free run length starting at the pixel along +x, -x, +y, -y (65535 = to the edge
// and beyond).  For each quadrant and dominant axis the largest h with a wall-free (2h along the axis) x (h across) rectangle is
// found by walking the h rows (columns) with a running minimum of the run lengths; the rectangle is stored when it reaches
// farther along the dominant axis than the square.
__global__ void ftgp_octant_field_kernel(const uint8_t* __restrict__ ksq, const uint16_t* __restrict__ runx, const uint16_t* __restrict__ runy,
                                         int W, int H, uint32_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const int x = i % W, y = i / W;
    const size_t plane = (size_t)W * H;
    uint32_t word[2] = { 0u, 0u };
    for (int q = 0; q < 4; ++q) {
        const int k = ksq[(size_t)q * plane + i];
        if (k == 0) continue;                                     // wall: all bytes stay 0
        const int sx = (q & 1) ? -1 : 1, sy = (q & 2) ? -1 : 1;
        const uint16_t* rx = runx + ((q & 1) ? plane : 0);        // runs along the quadrant's x direction
        const uint16_t* ry = runy + ((q & 2) ? plane : 0);
        for (int dom = 0; dom < 2; ++dom) {
            int h = 0, m = 65535;
            for (; h < 127; ++h) {
                int r;
                if (dom == 0) { const int yy = y + sy * h; r = (yy >= 0 && yy < H) ? (int)rx[(size_t)yy * W + x] : 65535; }
                else          { const int xx = x + sx * h; r = (xx >= 0 && xx < W) ? (int)ry[(size_t)y * W + xx] : 65535; }
                m = r < m ? r : m;
                if (m < 2 * (h + 1)) break;
            }
            const int ks = k < 127 ? k : 127;
            const uint32_t byte = (2 * h > ks) ? (0x80u | (uint32_t)h) : (uint32_t)ks;
            word[dom] |= byte << (8 * q);
        }
    }
    out[2 * (size_t)i] = word[0];
    out[2 * (size_t)i + 1] = word[1];
}

// Metrics record (FTGP_METRIC_DOUBLES): one block, deterministic tree reduction (integers are exact in f64).
__global__ void __launch_bounds__(256) ftgp_metrics_kernel(DeviceParams P, double* __restrict__ out)
{
    __shared__ double red[5][256];
    __shared__ double rmin[256], rmax[256];
    double steps = 0, laps = 0, absc = 0, fin = 0, off = 0, tmin = INFINITY, tmax = -INFINITY;
    for (int e = threadIdx.x; e < P.n_envs; e += blockDim.x) steps += (double)P.steps[e];
    for (int i = threadIdx.x; i < P.n_cars; i += blockDim.x) {
        const CarState& a = P.cars[i];
        const int lc = a.good_start ? a.completion : -(100 - a.completion);     // custom.py:132-143
        laps += a.laps; absc += a.laps * 100 + lc; fin += a.finished; off += a.off_track;
        const int n = a.n_times < FTGP_MAX_LAP_TIMES ? a.n_times : FTGP_MAX_LAP_TIMES;
        for (int k = 0; k < n; ++k) { tmin = fmin(tmin, a.times[k]); tmax = fmax(tmax, a.times[k]); }
    }
    red[0][threadIdx.x] = steps; red[1][threadIdx.x] = laps; red[2][threadIdx.x] = absc;
    red[3][threadIdx.x] = fin; red[4][threadIdx.x] = off; rmin[threadIdx.x] = tmin; rmax[threadIdx.x] = tmax;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            for (int q = 0; q < 5; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + w];
            rmin[threadIdx.x] = fmin(rmin[threadIdx.x], rmin[threadIdx.x + w]);
            rmax[threadIdx.x] = fmax(rmax[threadIdx.x], rmax[threadIdx.x + w]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = red[0][0]; out[1] = (double)P.n_cars; out[2] = red[1][0]; out[3] = red[2][0];
        out[4] = red[3][0]; out[5] = red[4][0]; out[6] = rmin[0]; out[7] = rmax[0];
    }
}
