/*
 * ftgp_oracle.c -- CPU oracle for the ft_grandprix hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (ft_grandprix_amd/, libftgp.so) never does and has no CPU fallback.
 *
 * What is restated here, and from where (paths relative to the reference repo):
 *   step order ................ ft_grandprix/custom.py:1337-1426 (progress -> driver -> ctrl -> mj_step -> steps += 1)
 *   lap progress .............. ft_grandprix/custom.py:1340-1372, 132-143        [pinned: tests/golden G5 + G4]
 *   snapshot / euler .......... ft_grandprix/custom.py:149-160, 62-76, vehicle.py:3-12   [pinned: G4]
 *   reset / spawn ............. ft_grandprix/custom.py:1089-1128, 1232-1245, 81-87
 *   2-D sphere-traced LiDAR ... ft_grandprix/raycast.py:5-21                     [pinned: G2]
 *   ... inside the step loop .. ft_grandprix/custom.py:1381-1393 (FtgpConfig.lidar_mode = FTGP_LIDAR_FAKELIDAR)   [pinned: G2 + G8]
 *   distance transform ........ custom.py:1149-1153 / raycast.py:24-27 (scipy.ndimage.distance_transform_edt), restated
 *                               without scipy: tests/test_oracle_golden.py compares it with scipy's on the four tracks
 *   drivers nidc / fast / lobotomy ... ft_grandprix/nidc.py:12-131, fast.py:11-139, lobotomy.py:1-3 [pinned: G1]
 *   vehicle + rangefinder model ..... template/mushr.em.xml:28-218 stepped by mujoco.mj_step
 *       (custom.py:1425).  MuJoCo (pinned 3.2.2 / 3.3.2 by requirements.txt:4 / uv.lock:104) is NOT
 *       installed and is not in the reference tree: PARITY UNPINNED for integrate, contacts and
 *       rangefinder values.  What is implemented is the reduced planar model specified in DESIGN.md
 *       ("K1", "K2"); the HIP kernels implement the same specification independently.
 *
 * Arithmetic rules of the specification (so that an independent implementation can be bit-identical):
 *   - dynamics, progress, policies: IEEE binary64, one rounding per written operation, no contraction
 *     (build with -ffp-contract=off), no libm calls on the per-step path except sqrt (correctly rounded)
 *     and, in the nidc/fast policies only, atan/ceil.
 *   - LiDAR march: IEEE binary32 with explicit fmaf where written.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/ftgp.h"

#define NPATH FTGP_PATH_POINTS

typedef struct Car {
    /* pose / velocity (planar) */
    double x, y, qw, qz;
    double vx, vy, wz;
    /* steering servo, wheel spin */
    double qs, qsd;
    double w[4];
    /* controls */
    double u_speed, u_steer;
    /* race state (custom.py:91-143) */
    int32_t completion, laps, offset;
    int32_t good_start, finished, off_track, delta;
    int32_t n_times;
    int64_t start;           /* vehicle_state.start (custom.py:1362): 64 bits like self.steps */
    int64_t finish_step;     /* self.steps when `finished` was set (custom.py:1367-1370) */
    double times[FTGP_MAX_LAP_TIMES];
    double dist2;            /* distance_from_track (squared, custom.py:1343) */
    /* fast.py:12 */
    double last_steer;
} Car;

struct OracleEnv {
    FtgpConfig cfg;
    uint32_t *bits;
    double path[NPATH][2];
    uint8_t *field;          /* Chebyshev distance to the nearest wall cell, 0 on walls, clamped to 255 */
    float *ray_bx, *ray_by;  /* body-frame ray directions, binary32 */
    double *ray_bxd, *ray_byd;
    double *edt;             /* FTGP_LIDAR_FAKELIDAR: Euclidean distance transform of the wall image (what the reference calls self.dt) */
    double spawn[NPATH][4];  /* x, y, qw, qz for a car spawned at path index p */
    int n_cars;
    Car *cars;
    float *ranges;           /* [n_cars][n_rays] */
    int64_t *steps;          /* per env */
    int32_t *place;          /* per car: Mujoco.winners[id] (custom.py:1125,1368-1369), 0 = not a winner yet */
    int32_t *n_winners;      /* per env: len(self.winners) */
    double wheel_load[4];
    int lidar_mode;          /* 0 = f32 field-accelerated march (== 2 bit for bit), 1 = binary64 plain DDA, 2 = THE SPEC: f32 plain DDA */
    int threads;
    double last_ms;
    int32_t car_policy[8];   /* FTGP_POLICY_PER_CAR: the driver of car slot k of every env (the roster, custom.py:1097-1104); 0 = not set */
};
typedef struct OracleEnv OracleEnv;

static __thread char g_err[256];
static int fail(int code, const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return code; }
const char *oracle_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ vehicle constants */
void oracle_default_vehicle(FtgpVehicle *v)
{
    memset(v, 0, sizeof *v);
    /* chassis 3.542137 + 4 wheels 0.498952 + steering-wheel geom 0.01 + lidar puck (density 1000) + softeners */
    v->mass = 5.632768;
    v->izz = 0.0316994;
    const double s = 0.5;  /* mushr_scale */
    v->wheel_x[0] = s * 0.1385;  v->wheel_y[0] = s * 0.115;
    v->wheel_x[1] = s * 0.1385;  v->wheel_y[1] = s * -0.115;
    v->wheel_x[2] = s * -0.158;  v->wheel_y[2] = s * 0.115;
    v->wheel_x[3] = s * -0.158;  v->wheel_y[3] = s * -0.115;
    v->wheel_radius = 0.03;
    v->wheel_inertia = 0.01 + 0.498952 / 5.0 * (0.03 * 0.03 + 0.03 * 0.03);
    v->wheel_damping = 0.01;
    v->throttle_kv = 100.0; v->throttle_gear = 0.04; v->throttle_force_limit = 500.0;
    v->steer_kp = 20.0; v->steer_damping = 0.3;
    v->steer_inertia = 3 * 0.0002 + 2 * (0.498952 / 5.0 * (0.03 * 0.03 + 0.01 * 0.01)) + 0.01 / 5.0 * (0.03 * 0.03 + 0.01 * 0.01);
    v->steer_limit = 1.0;
    v->friction = 0.5; v->gravity = 9.81;
    v->tire_damping = (v->mass / 4.0) * (2.0 / (0.95 * 0.02));
    v->contact_x[0] = 0.0385; v->contact_x[1] = 0.0; v->contact_x[2] = -0.0385;
    v->contact_radius = 0.0655;
    v->contact_stiffness = v->mass / (0.95 * 0.95 * 0.02 * 0.02);
    v->contact_damping = v->mass * (2.0 / (0.95 * 0.02));
    v->lidar_x = -0.0525; v->lidar_y = 0.0; v->lidar_ring_radius = 0.03;
    v->body_z = 0.0156;
    v->box_xmin = -0.1027; v->box_xmax = 0.1034; v->box_ymin = -0.0461; v->box_ymax = 0.0472;
    v->softener_radius = 0.65 * 0.0488;   /* mushr_wheel.stl radius at the scale of mushr.em.xml:39 */
}


void oracle_tricycle_vehicle(FtgpVehicle* v)
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

/* ------------------------------------------------------------------ small math (specified polynomials) */
/* sin/cos by Taylor series in Horner form on x*x; accurate to < 1e-15 for |x| <= 1.7 (the only range used). */
static double spec_sin(double x)
{
    double z = x * x;
    double p = -1.0 / 51090942171709440000.0;          /* -1/21! */
    p = p * z + 1.0 / 121645100408832000.0;            /*  1/19! */
    p = p * z - 1.0 / 355687428096000.0;               /* -1/17! */
    p = p * z + 1.0 / 1307674368000.0;                 /*  1/15! */
    p = p * z - 1.0 / 6227020800.0;                    /* -1/13! */
    p = p * z + 1.0 / 39916800.0;                      /*  1/11! */
    p = p * z - 1.0 / 362880.0;                        /* -1/9!  */
    p = p * z + 1.0 / 5040.0;                          /*  1/7!  */
    p = p * z - 1.0 / 120.0;                           /* -1/5!  */
    p = p * z + 1.0 / 6.0;                             /*  1/3!  */
    p = p * z;
    return x - x * p;
}
static double spec_cos(double x)
{
    double z = x * x;
    double p = 1.0 / 2432902008176640000.0;            /*  1/20! */
    p = p * z - 1.0 / 6402373705728000.0;              /* -1/18! */
    p = p * z + 1.0 / 20922789888000.0;                /*  1/16! */
    p = p * z - 1.0 / 87178291200.0;                   /* -1/14! */
    p = p * z + 1.0 / 479001600.0;                     /*  1/12! */
    p = p * z - 1.0 / 3628800.0;                       /* -1/10! */
    p = p * z + 1.0 / 40320.0;                         /*  1/8!  */
    p = p * z - 1.0 / 720.0;                           /* -1/6!  */
    p = p * z + 1.0 / 24.0;                            /*  1/4!  */
    p = p * z - 0.5;                                   /* -1/2!  */
    p = p * z;
    return 1.0 + p;
}

static uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

/* ------------------------------------------------------------------ track helpers */
static inline int wall_at(const OracleEnv *e, int ix, int iy)
{
    return (e->bits[(size_t)iy * e->cfg.track.words_per_row + (ix >> 5)] >> (ix & 31)) & 1u;
}

/* exact chessboard distance transform, two raster passes */
static void build_field(OracleEnv *e)
{
    const int W = e->cfg.track.width, H = e->cfg.track.height;
    int *d = (int *)malloc(sizeof(int) * (size_t)W * H);
    const int BIG = 1 << 20;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) d[(size_t)y * W + x] = wall_at(e, x, y) ? 0 : BIG;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int v = d[(size_t)y * W + x];
            if (x > 0 && d[(size_t)y * W + x - 1] + 1 < v) v = d[(size_t)y * W + x - 1] + 1;
            if (y > 0) {
                if (d[(size_t)(y - 1) * W + x] + 1 < v) v = d[(size_t)(y - 1) * W + x] + 1;
                if (x > 0 && d[(size_t)(y - 1) * W + x - 1] + 1 < v) v = d[(size_t)(y - 1) * W + x - 1] + 1;
                if (x < W - 1 && d[(size_t)(y - 1) * W + x + 1] + 1 < v) v = d[(size_t)(y - 1) * W + x + 1] + 1;
            }
            d[(size_t)y * W + x] = v;
        }
    for (int y = H - 1; y >= 0; --y)
        for (int x = W - 1; x >= 0; --x) {
            int v = d[(size_t)y * W + x];
            if (x < W - 1 && d[(size_t)y * W + x + 1] + 1 < v) v = d[(size_t)y * W + x + 1] + 1;
            if (y < H - 1) {
                if (d[(size_t)(y + 1) * W + x] + 1 < v) v = d[(size_t)(y + 1) * W + x] + 1;
                if (x > 0 && d[(size_t)(y + 1) * W + x - 1] + 1 < v) v = d[(size_t)(y + 1) * W + x - 1] + 1;
                if (x < W - 1 && d[(size_t)(y + 1) * W + x + 1] + 1 < v) v = d[(size_t)(y + 1) * W + x + 1] + 1;
            }
            d[(size_t)y * W + x] = v;
        }
    for (size_t i = 0; i < (size_t)W * H; ++i) e->field[i] = (uint8_t)(d[i] > 255 ? 255 : d[i]);
    free(d);
}

/* ------------------------------------------------------------------ exact Euclidean distance transform */
/* custom.py:1149-1153 / raycast.py:24-27: dt = scipy.ndimage.distance_transform_edt(non-wall mask): for every pixel the distance,
 * centre to centre, to the nearest wall pixel; 0 on walls.  Two separable passes on SQUARED distances: down every column the 1-D
 * distance to the nearest wall of that column, then along every row the lower envelope of the parabolas (x - q)^2 + g(q)^2.  The
 * envelope is built with exact integer arithmetic (parabola q takes over from p where x > ((g_q^2 + q^2) - (g_p^2 + p^2)) / (2 (q - p)),
 * compared by cross-multiplication), so the squared distances are exact integers; dt = sqrt of them, correctly rounded. */
static void build_edt(OracleEnv *e)
{
    const int W = e->cfg.track.width, H = e->cfg.track.height;
    const int64_t INF = (int64_t)1 << 40;
    int64_t *g2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)W * H);
    for (int x = 0; x < W; ++x) {
        int64_t d = INF;                                   /* distance to the last wall seen above */
        for (int y = 0; y < H; ++y) {
            d = wall_at(e, x, y) ? 0 : (d >= INF ? INF : d + 1);
            g2[(size_t)y * W + x] = d;
        }
        d = INF;
        for (int y = H - 1; y >= 0; --y) {
            d = wall_at(e, x, y) ? 0 : (d >= INF ? INF : d + 1);
            if (d < g2[(size_t)y * W + x]) g2[(size_t)y * W + x] = d;
        }
    }
    int *v = (int *)malloc(sizeof(int) * (size_t)W);       /* parabola sites of the envelope */
    int64_t *zn = (int64_t *)malloc(sizeof(int64_t) * (size_t)(W + 1)), *zd = (int64_t *)malloc(sizeof(int64_t) * (size_t)(W + 1));
    int64_t *f = (int64_t *)malloc(sizeof(int64_t) * (size_t)W);
    for (int y = 0; y < H; ++y) {
        int n = 0;
        for (int q = 0; q < W; ++q) {
            int64_t gq = g2[(size_t)y * W + q];
            if (gq >= INF) continue;                       /* no wall in this column */
            f[q] = gq * gq;
            /* intersection of parabola q with the envelope's last one p: s = ((f_q + q^2) - (f_p + p^2)) / (2 (q - p)), kept as a fraction */
            while (n > 0) {
                int p = v[n - 1];
                int64_t num = (f[q] + (int64_t)q * q) - (f[p] + (int64_t)p * p), den = 2 * (int64_t)(q - p);
                /* s <= z[n-1]  <=>  num * zd <= zn * den  (all denominators positive) */
                if (n > 1 && num * zd[n - 1] <= zn[n - 1] * den) { --n; continue; }
                zn[n] = num; zd[n] = den;
                break;
            }
            v[n] = q;
            if (n == 0) { zn[0] = -INF; zd[0] = 1; }
            ++n;
        }
        int k = 0;
        for (int x = 0; x < W; ++x) {
            double out;
            if (n == 0) out = sqrt((double)INF);
            else {
                /* the parabola whose range contains x: advance while x > z[k + 1], i.e. x * zd > zn */
                while (k + 1 < n && (int64_t)x * zd[k + 1] > zn[k + 1]) ++k;
                int q = v[k];
                out = sqrt((double)((int64_t)(x - q) * (x - q) + f[q]));
            }
            e->edt[(size_t)y * W + x] = out;
        }
    }
    free(g2); free(v); free(zn); free(zd); free(f);
}

/* ------------------------------------------------------------------ K2: LiDAR */
/* Ray against the other cars of the same env: chassis box and LiDAR puck (binary32). Returns +INF when nothing is hit. */
static float ray_vs_cars(const OracleEnv *e, int car_index, double lcx, double lcy, float dxw, float dyw)
{
    const FtgpVehicle *v = &e->cfg.vehicle;
    const int cpe = e->cfg.cars_per_env;
    const int env = car_index / cpe;
    const float r0 = (float)v->lidar_ring_radius;
    float best = INFINITY;
    for (int k = 0; k < cpe; ++k) {
        int other = env * cpe + k;
        if (other == car_index) continue;
        const Car *b = &e->cars[other];
        if (b->finished) continue;                    /* shadowed cars are invisible (custom.py:1441-1466) */
        double cb = 1.0 - 2.0 * (b->qz * b->qz), sb = 2.0 * (b->qw * b->qz);
        float relx = (float)(lcx - b->x), rely = (float)(lcy - b->y);
        float ox = fmaf(dxw, -r0, relx), oy = fmaf(dyw, -r0, rely);
        float cbf = (float)cb, sbf = (float)sb;
        float lx = fmaf(cbf, ox, sbf * oy), ly = fmaf(cbf, oy, -(sbf * ox));
        float ldx = fmaf(cbf, dxw, sbf * dyw), ldy = fmaf(cbf, dyw, -(sbf * dxw));
        /* chassis box, slab test */
        {
            float xmin = (float)v->box_xmin, xmax = (float)v->box_xmax, ymin = (float)v->box_ymin, ymax = (float)v->box_ymax;
            float tmin = -INFINITY, tmax = INFINITY; int miss = 0;
            if (ldx != 0.0f) {
                float inv = 1.0f / ldx; float t1 = (xmin - lx) * inv, t2 = (xmax - lx) * inv;
                tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
            } else if (lx < xmin || lx > xmax) miss = 1;
            if (ldy != 0.0f) {
                float inv = 1.0f / ldy; float t1 = (ymin - ly) * inv, t2 = (ymax - ly) * inv;
                tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
            } else if (ly < ymin || ly > ymax) miss = 1;
            if (!miss && tmax >= fmaxf(tmin, 0.0f)) {
                float t = tmin > 0.0f ? tmin : 0.0f;
                if (t < best) best = t;
            }
        }
        /* LiDAR puck: circle of radius lidar_ring_radius at (lidar_x, lidar_y) */
        {
            float px = lx - (float)v->lidar_x, py = ly - (float)v->lidar_y;
            float bq = fmaf(px, ldx, py * ldy);
            float cq = fmaf(px, px, py * py) - r0 * r0;
            float disc = fmaf(bq, bq, -cq);
            if (disc >= 0.0f) {
                float t = -bq - sqrtf(disc);
                if (t < 0.0f) t = (cq < 0.0f) ? 0.0f : INFINITY;
                if (t < best) best = t;
            }
        }
    }
    return best;
}

/*
 * THE SPECIFICATION of a ray (binary32): plain cell-by-cell DDA with crossing times in coordinates relative to the start cell.
 * Start cell (ix0, iy0) = floor of the origin.  Per axis (p = origin coordinate, i0 = its floor, d = direction component):
 *   f = p - (float)i0 (exact);  g = d < 0 ? 1.0f - f : f  (the offset inside the start cell, seen in the direction of travel);
 *   iv = |1 / d| (IEEE division), +inf replaced by FLT_MAX (d = 0: the axis is never stepped, and 0 * iv stays 0);  c = g * iv;
 *   S(k) = fma((float)k, iv, -c) = crossing time of the k-th cell boundary the ray meets on this axis, k = 1, 2, ...
 * After mx steps in x and my steps in y the next step is the x-neighbour iff Sx(mx + 1) < Sy(my + 1) (a tie steps in y); the range is
 * the crossing time of the step that enters the first wall cell (0 if the start cell is a wall, -1 if the ray leaves the image).
 * Every quantity is a function of (cell, ray) alone -- nothing accumulates -- so any implementation that skips wall-free cells and
 * re-synchronises with these comparisons gives the same bits.  (Rounds 1-4: S = ((float)b - p) * (1 / d) on absolute boundary
 * coordinates b; DESIGN.md section 4 says why it changed.)
 */
typedef struct { float ivx, ivy, cx, cy; int sx, sy; } RaySpec;      /* sx, sy = +1 / -1: direction of travel in the image */

static float spec_iv(float d) { float z = fabsf(1.0f / d); return z < FLT_MAX ? z : FLT_MAX; }

static void ray_spec(RaySpec *q, float pu, float pv, float du, float dv, int ix0, int iy0)
{
    const float fu = pu - (float)ix0, fv = pv - (float)iy0;
    const float gu = (du < 0.0f) ? 1.0f - fu : fu, gv = (dv < 0.0f) ? 1.0f - fv : fv;
    q->ivx = spec_iv(du); q->ivy = spec_iv(dv);
    q->cx = gu * q->ivx; q->cy = gv * q->ivy;
    q->sx = (du < 0.0f) ? -1 : 1; q->sy = (dv < 0.0f) ? -1 : 1;
}
#define SPEC_SX(q, k) fmaf((float)(k), (q).ivx, -(q).cx)
#define SPEC_SY(q, k) fmaf((float)(k), (q).ivy, -(q).cy)

static float march_plain_f32(const OracleEnv *e, float pu, float pv, float du, float dv)
{
    const int W = e->cfg.track.width, H = e->cfg.track.height;
    const int ix0 = (int)floorf(pu), iy0 = (int)floorf(pv);
    if (!(pu == pu) || !(pv == pv) || ix0 < 0 || ix0 >= W || iy0 < 0 || iy0 >= H) return -1.0f;
    RaySpec q; ray_spec(&q, pu, pv, du, dv, ix0, iy0);
    int mx = 0, my = 0;
    float s = 0.0f;
    for (;;) {
        const int ix = ix0 + q.sx * mx, iy = iy0 + q.sy * my;
        if (ix < 0 || ix >= W || iy < 0 || iy >= H) return -1.0f;
        if (wall_at(e, ix, iy)) return fabsf(s);   /* |s|: a ray that starts on a boundary can produce -0 */
        const float sX = SPEC_SX(q, mx + 1), sY = SPEC_SY(q, my + 1);
        if (sX < sY) { s = sX; mx += 1; }
        else         { s = sY; my += 1; }
    }
}

/*
 * The oracle's own fast path (used for the CPU baseline and the long closed-loop tests): skip over the wall-free
 * (2k-1)^2 block of cells given by the chessboard distance field, then re-synchronise with the specification's
 * comparisons.  tests/test_oracle_lidar.py checks it against march_plain_f32 bit for bit.
 */
static float march_f32(const OracleEnv *e, float pu, float pv, float du, float dv)
{
    const int W = e->cfg.track.width, H = e->cfg.track.height;
    const int ix0 = (int)floorf(pu), iy0 = (int)floorf(pv);
    if (!(pu == pu) || !(pv == pv) || ix0 < 0 || ix0 >= W || iy0 < 0 || iy0 >= H) return -1.0f;
    RaySpec q; ray_spec(&q, pu, pv, du, dv, ix0, iy0);
    const float adu = fabsf(du), adv = fabsf(dv);
    const float fu = pu - (float)ix0, fv = pv - (float)iy0;
    const float gu = (du < 0.0f) ? 1.0f - fu : fu, gv = (dv < 0.0f) ? 1.0f - fv : fv;
    int mx = 0, my = 0;          /* cells travelled along each axis */
    float s = 0.0f;
    for (int it = 0; it < 1 << 20; ++it) {
        const int ix = ix0 + q.sx * mx, iy = iy0 + q.sy * my;
        if (ix < 0 || ix >= W || iy < 0 || iy >= H) return -1.0f;
        const int k = e->field[(size_t)iy * W + ix];            /* the (2k-1)^2 block around the cell is wall-free */
        if (k == 0) return fabsf(s);
        const float sX = SPEC_SX(q, mx + k), sY = SPEC_SY(q, my + k);      /* leaving the block: k boundaries further on */
        if (sX < sY) {
            s = sX;
            /* y boundaries already crossed at time s: those with Sy(b) <= s; estimate, then settle with the comparisons */
            int t = (int)floorf(fmaf(adv, s, gv));
            const int hi = my + k - 1;
            if (t < my) t = my;
            if (t > hi) t = hi;
            while (t > my && !(SPEC_SY(q, t) <= s)) t -= 1;
            while (t < hi && (SPEC_SY(q, t + 1) <= s)) t += 1;
            mx += k; my = t;
        } else {
            s = sY;
            /* x boundaries already crossed at time s: those with Sx(b) < s (a tie steps in y first) */
            int t = (int)floorf(fmaf(adu, s, gu));
            const int hi = mx + k - 1;
            if (t < mx) t = mx;
            if (t > hi) t = hi;
            while (t > mx && !(SPEC_SX(q, t) < s)) t -= 1;
            while (t < hi && (SPEC_SX(q, t + 1) < s)) t += 1;
            my += k; mx = t;
        }
    }
    return -1.0f;
}

/* Truth: plain cell-by-cell DDA in binary64 (no acceleration structure). */
static double march_f64(const OracleEnv *e, double pu, double pv, double du, double dv)
{
    const int W = e->cfg.track.width, H = e->cfg.track.height;
    int ix = (int)floor(pu), iy = (int)floor(pv);
    if (ix < 0 || ix >= W || iy < 0 || iy >= H) return -1.0;
    double s = 0.0;
    for (;;) {
        if (wall_at(e, ix, iy)) return s;
        double sX = (du != 0.0) ? ((double)((du > 0.0) ? ix + 1 : ix) - pu) / du : INFINITY;
        double sY = (dv != 0.0) ? ((double)((dv > 0.0) ? iy + 1 : iy) - pv) / dv : INFINITY;
        if (sX < sY) { s = sX; ix += (du > 0.0) ? 1 : -1; }
        else         { s = sY; iy += (dv > 0.0) ? 1 : -1; }
        if (ix < 0 || ix >= W || iy < 0 || iy >= H) return -1.0;
    }
}

/* The reference's own 2-D LiDAR inside the step loop (option use_simulated_simulation_lidar, custom.py:987,1381-1393), statement by
 * statement; the fan is the rangefinders' (ray order per SURVEY.md 8a-3: the branch's own linspace, custom.py:1387, is dead code). */
static void fakelidar_car(OracleEnv *e, int ci)
{
    const FtgpConfig *c = &e->cfg;
    const Car *a = &e->cars[ci];
    const int R = c->n_rays, W = c->track.width, H = c->track.height;
    float *out = e->ranges + (size_t)ci * R;
    const double s = c->map_size > 0.0 ? c->map_size : 40.0;       /* s = 20 * self.map_metadata["scale"], custom.py:1382 */
    const double i_x = (a->x / s) * (double)W;                      /* custom.py:1383 */
    const double i_y = -(a->y / s) * (double)H;                     /* custom.py:1384 */
    const double ch = 1.0 - 2.0 * (a->qz * a->qz), sh = 2.0 * (a->qw * a->qz);
    for (int j = 0; j < R; ++j) {
        const double dxw = ch * e->ray_bxd[j] - sh * e->ray_byd[j], dyw = sh * e->ray_bxd[j] + ch * e->ray_byd[j];
        const double dx = dxw, dy = -dyw;                           /* image rows grow downwards */
        /* raycast.py:9-20 */
        double x = i_x, y = i_y, distance = 0;
        int bad = 0;
        long yi = (long)y, xi = (long)x;
        if (yi < 0) yi += H;
        if (xi < 0) xi += W;
        double nearest = 0;
        if (yi < 0 || yi >= H || xi < 0 || xi >= W) bad = 1; else nearest = e->edt[(size_t)yi * W + xi];
        while (!bad && nearest > 2 && 0 <= x && x <= W && 0 <= y && y <= H) {
            distance += nearest;
            x += dx * nearest;
            y += dy * nearest;
            yi = (long)y; xi = (long)x;
            if (yi < 0) yi += H;
            if (xi < 0) xi += W;
            if (yi < 0 || yi >= H || xi < 0 || xi >= W) { bad = 1; break; }    /* IndexError in the reference: the ray reads -1 here */
            nearest = e->edt[(size_t)yi * W + xi];
        }
        out[j] = bad ? -1.0f : (float)((distance / (double)W) * s);             /* ranges /= original_width; ranges *= s (custom.py:1392-1393) */
    }
}

static void lidar_car(OracleEnv *e, int ci)
{
    const FtgpConfig *c = &e->cfg;
    const FtgpVehicle *v = &c->vehicle;
    const Car *a = &e->cars[ci];
    const int R = c->n_rays;
    float *out = e->ranges + (size_t)ci * R;
    if (a->finished) {                                /* shadow_rangefinders: a finished car's sensors are switched off (custom.py:1436-1439) */
        for (int j = 0; j < R; ++j) out[j] = 0.0f;
        return;
    }
    if (c->lidar_mode == FTGP_LIDAR_FAKELIDAR) { fakelidar_car(e, ci); return; }
    const double ch = 1.0 - 2.0 * (a->qz * a->qz), sh = 2.0 * (a->qw * a->qz);
    const double lcx = a->x + (ch * v->lidar_x - sh * v->lidar_y);
    const double lcy = a->y + (sh * v->lidar_x + ch * v->lidar_y);
    const double inv_sx = 1.0 / c->track.px_size_x, inv_sy = 1.0 / c->track.px_size_y;
    if (e->lidar_mode == 1) {
        const double u0 = (lcx - c->track.origin_x) * inv_sx, v0 = (c->track.origin_y - lcy) * inv_sy;
        for (int j = 0; j < R; ++j) {
            double dxw = ch * e->ray_bxd[j] - sh * e->ray_byd[j], dyw = sh * e->ray_bxd[j] + ch * e->ray_byd[j];
            double du = dxw * inv_sx, dv = -(dyw * inv_sy);
            double r0 = v->lidar_ring_radius;
            double r = march_f64(e, u0 - du * r0, v0 - dv * r0, du, dv);
            if (c->cars_per_env > 1) {
                float rc = ray_vs_cars(e, ci, lcx, lcy, (float)dxw, (float)dyw);
                if (rc < INFINITY && (r < 0.0 || (double)rc < r)) r = rc;
            }
            out[j] = (float)r;
        }
        return;
    }
    const float u0 = (float)((lcx - c->track.origin_x) * inv_sx);
    const float v0 = (float)((c->track.origin_y - lcy) * inv_sy);
    const float chf = (float)ch, shf = (float)sh;
    const float isx = (float)inv_sx, isy = (float)inv_sy;
    const float r0 = (float)v->lidar_ring_radius;
    for (int j = 0; j < R; ++j) {
        float bx = e->ray_bx[j], by = e->ray_by[j];
        float dxw = fmaf(chf, bx, -(shf * by));
        float dyw = fmaf(shf, bx, chf * by);
        float du = dxw * isx;
        float dv = -(dyw * isy);
        float pu = fmaf(du, -r0, u0);
        float pv = fmaf(dv, -r0, v0);
        float r = (e->lidar_mode == 2) ? march_plain_f32(e, pu, pv, du, dv) : march_f32(e, pu, pv, du, dv);
        if (c->cars_per_env > 1) {
            float rc = ray_vs_cars(e, ci, lcx, lcy, dxw, dyw);
            if (rc < INFINITY && (r < 0.0f || rc < r)) r = rc;
        }
        out[j] = r;
    }
}

/* ------------------------------------------------------------------ K3: lap progress (custom.py:1340-1372) */
static void progress_car(OracleEnv *e, int ci)
{
    Car *a = &e->cars[ci];
    const int env = ci / e->cfg.cars_per_env;
    const int64_t steps = e->steps[env];
    /* distances = ((path - xpos)**2).sum(1); closest = distances.argmin()  -- first minimum */
    int closest = 0; double best = 0.0;
    for (int i = 0; i < NPATH; ++i) {
        double dx = e->path[i][0] - a->x, dy = e->path[i][1] - a->y;
        double d = dx * dx + dy * dy;
        if (i == 0 || d < best) { best = d; closest = i; }
    }
    a->dist2 = best;
    a->off_track = best > 1.0;
    if (a->off_track) return;
    int completion = ((closest - a->offset) % 100 + 100) % 100;
    int delta = completion - a->completion;
    a->delta = (((completion - a->completion + 50) % 100) + 100) % 100 - 50;
    if (abs(delta) > 90) {
        double lap_time = (double)(steps - a->start) * e->cfg.dt;
        if (a->delta < 0) {
            a->laps -= 1;
            a->good_start = 0;
            if (a->n_times != 0) {                     /* times.pop() */
                /* beyond FTGP_MAX_LAP_TIMES counted laps the popped entry sits where the oldest entry the list would still show used to be:
                 * that slot holds no entry now (NaN), see include/ftgp.h */
                if (a->n_times > FTGP_MAX_LAP_TIMES) a->times[(a->n_times - 1) % FTGP_MAX_LAP_TIMES] = NAN;
                a->n_times -= 1;
            }
        } else if (a->delta > 0) {
            if (a->good_start) {
                a->times[a->n_times % FTGP_MAX_LAP_TIMES] = lap_time;   /* times.append(lap_time); the newest FTGP_MAX_LAP_TIMES are kept */
                a->n_times += 1;
                a->start = steps;
            }
            a->laps += 1;
            a->good_start = 1;
        }
    }
    if (a->laps >= e->cfg.lap_target) {              /* custom.py:1367-1370: winners[id] = len(winners) + 1 the first time */
        if (!a->finished) a->finish_step = steps;
        if (e->place[ci] == 0) e->place[ci] = ++e->n_winners[env];     /* kept the reference's way: a dict filled inside the per-car loop */
        a->finished = 1;
    }
    a->completion = completion;
}

/* ------------------------------------------------------------------ K1: integrate one dt */
typedef struct Force { double fx, fy, tz; } Force;

/* One circle (centre = car position + (rxw, ryw), radius r) against the wall pixels: deepest penetration, ties -> first in raster order. */
static void wall_circle(const OracleEnv *e, const Car *a, double rxw, double ryw, double r, Force *f)
{
    const FtgpConfig *c = &e->cfg;
    const FtgpVehicle *v = &c->vehicle;
    const int W = c->track.width, H = c->track.height;
    const double sx = c->track.px_size_x, sy = c->track.px_size_y;
    const double inv_sx = 1.0 / sx, inv_sy = 1.0 / sy;
    const int nx = (int)ceil(r * inv_sx), ny = (int)ceil(r * inv_sy);
    const int reach = (nx > ny ? nx : ny) + 1;
    double px = a->x + rxw, py = a->y + ryw;
    double u = (px - c->track.origin_x) * inv_sx, w = (c->track.origin_y - py) * inv_sy;
    int ix = (int)floor(u), iy = (int)floor(w);
    if (ix < 0 || ix >= W || iy < 0 || iy >= H) return;
    if (e->field[(size_t)iy * W + ix] > reach) return;
    double best_pen = 0.0, bnx = 0.0, bny = 0.0; int found = 0;
    for (int dy = -ny; dy <= ny; ++dy) {
        int cy = iy + dy; if (cy < 0 || cy >= H) continue;
        for (int dx = -nx; dx <= nx; ++dx) {
            int cx = ix + dx; if (cx < 0 || cx >= W) continue;
            if (!wall_at(e, cx, cy)) continue;
            /* wall cell rectangle in world coordinates */
            double x0 = c->track.origin_x + (double)cx * sx, x1 = x0 + sx;
            double y1 = c->track.origin_y - (double)cy * sy, y0 = y1 - sy;
            double qx = px < x0 ? x0 : (px > x1 ? x1 : px);
            double qy = py < y0 ? y0 : (py > y1 ? y1 : py);
            double ex = px - qx, ey = py - qy;
            double d2 = ex * ex + ey * ey;
            if (d2 >= r * r) continue;
            double d = sqrt(d2);
            double pen = r - d;
            if (!found || pen > best_pen) {
                double nxv, nyv;
                if (d > 0.0) { nxv = ex / d; nyv = ey / d; }
                else {
                    double mx = px - (x0 + 0.5 * sx), my = py - (y0 + 0.5 * sy);
                    double m = sqrt(mx * mx + my * my);
                    if (m > 0.0) { nxv = mx / m; nyv = my / m; } else { nxv = 0.0; nyv = 0.0; }
                }
                best_pen = pen; bnx = nxv; bny = nyv; found = 1;
            }
        }
    }
    if (!found) return;
    double vcx = a->vx - a->wz * ryw, vcy = a->vy + a->wz * rxw;
    double vn = vcx * bnx + vcy * bny;
    double mag = v->contact_stiffness * best_pen - v->contact_damping * vn;
    if (mag <= 0.0) return;
    double fx = mag * bnx, fy = mag * bny;
    f->fx += fx; f->fy += fy; f->tz += rxw * fy - ryw * fx;
}

static void wall_contact(const OracleEnv *e, const Car *a, double ch, double sh, Force *f)
{
    const FtgpVehicle *v = &e->cfg.vehicle;
    for (int k = 0; k < 3; ++k)                                           /* chassis circles, body (cx, 0) rotated */
        wall_circle(e, a, ch * v->contact_x[k], sh * v->contact_x[k], v->contact_radius, f);
    if (e->cfg.bubble_wrap)                                               /* wheel softeners (custom.py:1041-1055; mushr.em.xml:65-67,126-129) */
        for (int k = 0; k < 4; ++k)
            wall_circle(e, a, ch * v->wheel_x[k] - sh * v->wheel_y[k], sh * v->wheel_x[k] + ch * v->wheel_y[k], v->softener_radius, f);
}

static void car_contact(const OracleEnv *e, int ci, double ch, double sh, Force *f)
{
    const FtgpConfig *c = &e->cfg;
    const FtgpVehicle *v = &c->vehicle;
    const int cpe = c->cars_per_env, env = ci / cpe;
    const Car *a = &e->cars[ci];
    const double r2 = 2.0 * v->contact_radius;
    if (a->finished) return;                          /* a shadowed car collides with nothing (custom.py:1452-1457) */
    for (int k = 0; k < cpe; ++k) {
        int other = env * cpe + k;
        if (other == ci) continue;
        const Car *b = &e->cars[other];
        if (b->finished) continue;
        double cb = 1.0 - 2.0 * (b->qz * b->qz), sb = 2.0 * (b->qw * b->qz);
        /* the contacts with one env-mate are added up on their own (from +0, in the order i, j) and the mates' sums join the car's force in
         * the order k (round 5; rounds 1-4: one running sum over k, i, j -- the same bits unless two contacts act on a car at once) */
        Force pk = { 0.0, 0.0, 0.0 };
        for (int i = 0; i < 3; ++i) {
            double rxw = ch * v->contact_x[i], ryw = sh * v->contact_x[i];
            double px = a->x + rxw, py = a->y + ryw;
            double vax = a->vx - a->wz * ryw, vay = a->vy + a->wz * rxw;
            for (int j = 0; j < 3; ++j) {
                double sxw = cb * v->contact_x[j], syw = sb * v->contact_x[j];
                double qx = b->x + sxw, qy = b->y + syw;
                double ex = px - qx, ey = py - qy;
                double d2 = ex * ex + ey * ey;
                if (d2 >= r2 * r2 || d2 <= 0.0) continue;
                double d = sqrt(d2);
                double nxv = ex / d, nyv = ey / d;
                double vbx = b->vx - b->wz * syw, vby = b->vy + b->wz * sxw;
                double vn = (vax - vbx) * nxv + (vay - vby) * nyv;
                double mag = v->contact_stiffness * (r2 - d) - v->contact_damping * vn;
                if (mag <= 0.0) continue;
                double fx = mag * nxv, fy = mag * nyv;
                pk.fx += fx; pk.fy += fy; pk.tz += rxw * fy - ryw * fx;
            }
        }
        f->fx += pk.fx; f->fy += pk.fy; f->tz += pk.tz;
    }
}

/* new state of car ci from the pre-step states of all cars (Jacobi over cars of an env) */
static void integrate_car(const OracleEnv *e, int ci, Car *out)
{
    const FtgpConfig *c = &e->cfg;
    const FtgpVehicle *v = &c->vehicle;
    const Car *a = &e->cars[ci];
    const double dt = c->dt;
    *out = *a;
    const double ch = 1.0 - 2.0 * (a->qz * a->qz), sh = 2.0 * (a->qw * a->qz);
    /* Ackermann polynomials, mushr.em.xml:185-186 */
    const double q = a->qs;
    const double dfl = q * (1.0 + q * (0.375 + q * (0.140625 + q * -0.0722656)));
    const double dfr = q * (1.0 + q * (-0.375 + q * (0.140625 + q * 0.0722656)));
    const double cw[4] = { spec_cos(dfl), spec_cos(dfr), 1.0, 1.0 };
    const double sw[4] = { spec_sin(dfl), spec_sin(dfr), 0.0, 0.0 };
    /* velocity servo on the mean wheel speed, mushr.em.xml:180,191-196 */
    const double wbar = 0.25 * (((a->w[0] + a->w[1]) + a->w[2]) + a->w[3]);
    double fa = v->throttle_kv * (a->u_speed - v->throttle_gear * wbar);
    if (fa > v->throttle_force_limit) fa = v->throttle_force_limit;
    if (fa < -v->throttle_force_limit) fa = -v->throttle_force_limit;
    const double ta = (v->throttle_gear * 0.25) * fa;
    /* legacy tricycle (car.em.xml:126-139): two torque motors on the tendons 0.5 (l + r) and 0.5 (r - l), ctrl clamped to ctrlrange */
    const int tri = v->kind == FTGP_VEHICLE_TRICYCLE;
    double uf = a->u_speed, ut = a->u_steer;
    if (uf > v->motor_forward_limit) uf = v->motor_forward_limit;
    if (uf < -v->motor_forward_limit) uf = -v->motor_forward_limit;
    if (ut > v->motor_turn_limit) ut = v->motor_turn_limit;
    if (ut < -v->motor_turn_limit) ut = -v->motor_turn_limit;
    const double tl = 0.5 * uf - 0.5 * ut, tr = 0.5 * uf + 0.5 * ut;
    Force f = { 0.0, 0.0, 0.0 };
    for (int i = 0; i < 4; ++i) {
        if (tri && i >= 2) continue;                  /* front caster: frictionless; there is no fourth wheel */
        const double torque = tri ? (i == 0 ? tl : tr) : ta;
        const double cwi = tri ? 1.0 : cw[i], swi = tri ? 0.0 : sw[i];
        double rxw = ch * v->wheel_x[i] - sh * v->wheel_y[i];
        double ryw = sh * v->wheel_x[i] + ch * v->wheel_y[i];
        double vpx = a->vx - a->wz * ryw, vpy = a->vy + a->wz * rxw;
        double fdx = ch * cwi - sh * swi, fdy = sh * cwi + ch * swi;             /* wheel heading, world */
        double vlong = (vpx * fdx + vpy * fdy) - v->wheel_radius * a->w[i];
        double vlat = vpy * fdx - vpx * fdy;                                    /* along (-fdy, fdx) */
        double flong = -(v->tire_damping * vlong), flat = -(v->tire_damping * vlat);
        double lim = v->friction * e->wheel_load[i];
        double m2 = flong * flong + flat * flat;
        if (m2 > lim * lim) { double sc = lim / sqrt(m2); flong = flong * sc; flat = flat * sc; }
        double fx = flong * fdx - flat * fdy, fy = flong * fdy + flat * fdx;
        f.fx += fx; f.fy += fy; f.tz += rxw * fy - ryw * fx;
        out->w[i] = (v->wheel_inertia * a->w[i] + dt * (torque - v->wheel_radius * flong)) / (v->wheel_inertia + dt * v->wheel_damping);
    }
    if (!a->finished) wall_contact(e, a, ch, sh, &f);
    if (c->cars_per_env > 1) car_contact(e, ci, ch, sh, &f);
    out->vx = a->vx + dt * (f.fx / v->mass);
    out->vy = a->vy + dt * (f.fy / v->mass);
    out->wz = a->wz + dt * (f.tz / v->izz);
    /* steering servo, implicit damping (the tricycle has no steering joint) */
    if (!tri) {
        out->qsd = (v->steer_inertia * a->qsd + dt * (v->steer_kp * (a->u_steer - a->qs))) / (v->steer_inertia + dt * v->steer_damping);
        out->qs = a->qs + dt * out->qsd;
        if (out->qs > v->steer_limit) { out->qs = v->steer_limit; if (out->qsd > 0.0) out->qsd = 0.0; }
        if (out->qs < -v->steer_limit) { out->qs = -v->steer_limit; if (out->qsd < 0.0) out->qsd = 0.0; }
    }
    /* positions with the new velocities (semi-implicit Euler) */
    out->x = a->x + dt * out->vx;
    out->y = a->y + dt * out->vy;
    double h = (0.5 * dt) * out->wz;
    double chh = spec_cos(h), shh = spec_sin(h);
    double nw = a->qw * chh - a->qz * shh, nz = a->qz * chh + a->qw * shh;
    double n = sqrt(nw * nw + nz * nz);
    out->qw = nw / n; out->qz = nz / n;
}

/* ------------------------------------------------------------------ policies (K5 oracle) */
static void policy_disparity(const OracleEnv *e, const float *ranges, Car *a, int fast, double *speed, double *steer)
{
    const int n = e->cfg.n_rays;
    const double car_width = fast ? 0.06 : 0.12;
    const double rpp = (2 * M_PI) / (double)n;                    /* nidc.py:121 */
    const int eighth = (int)((double)n / 8.0);                    /* nidc.py:18 */
    const int m = n - 2 * eighth;
    double *proc = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    int *disp = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    for (int i = 0; i < m; ++i) proc[i] = (double)ranges[eighth + i];
    int nd = 0;
    for (int i = 1; i < m; ++i)
        if (fabs(proc[i] - proc[i - 1]) > 0.6) disp[nd++] = i;    /* nidc.py:26-40 */
    const double width = (car_width / 2) * (1 + 300.0 / 100);      /* nidc.py:93 */
    for (int t = 0; t < nd; ++t) {                                /* nidc.py:94-105 */
        int first = disp[t] - 1;
        double p0 = proc[first], p1 = proc[first + 1];
        int close_idx = first + ((p1 < p0) ? 1 : 0);             /* argmin: first minimum */
        int far_idx = first + ((p1 > p0) ? 1 : 0);               /* argmax: first maximum */
        double close_dist = proc[close_idx];
        double angle = 2 * atan(width / (2 * close_dist));        /* nidc.py:57 */
        double cnt = ceil(angle / rpp);
        int num = (cnt > 2147483000.0) ? 2147483000 : (cnt < -2147483000.0 ? -2147483000 : (int)cnt);
        int cover_right = close_idx < far_idx;
        double nd_ = proc[close_idx];
        if (cover_right) {
            for (int i = 0; i < num; ++i) { int idx = close_idx + 1 + i; if (idx >= m) break; if (proc[idx] > nd_) proc[idx] = nd_; }
        } else {
            for (int i = 0; i < num; ++i) { int idx = close_idx - 1 - i; if (idx < 0) break; if (proc[idx] > nd_) proc[idx] = nd_; }
        }
    }
    int arg = 0;
    for (int i = 1; i < m; ++i) if (proc[i] > proc[arg]) arg = i;  /* argmax: first maximum */
    double ang = ((double)arg - ((double)m / 2)) * rpp;           /* nidc.py:112 */
    const double lim = 90.0 * (M_PI / 180.0);                     /* np.radians(90) */
    if (ang < -lim) ang = -lim;
    if (ang > lim) ang = lim;
    if (!fast) {
        *speed = 0.5 * 5 * (1 - fabs(ang) / (1.57 * 2));          /* nidc.py:130 */
    } else {
        const double old = 0.0;                                   /* fast.py:131-133 */
        ang = a->last_steer * old + ang * (1 - old);
        a->last_steer = ang;
        if (fabs(ang) < 0.1 && (double)ranges[0] > 0.5) *speed = 7.0;  /* fast.py:135-138 */
        else { double s = 0.5 * 5 * (1 - fabs(ang) / M_PI); *speed = s < 2.0 ? s : 2.0; }
    }
    *steer = ang;
    free(proc); free(disp);
}

static void policy_car(OracleEnv *e, int policy, int ci)
{
    Car *a = &e->cars[ci];
    const int env = ci / e->cfg.cars_per_env;
    const float *r = e->ranges + (size_t)ci * e->cfg.n_rays;
    if (a->finished) { a->u_speed = 0.0; a->u_steer = 0.0; return; }    /* finished cars get the Lobotomy driver, custom.py:1446 */
    if (policy == FTGP_POLICY_PER_CAR) policy = e->car_policy[ci % e->cfg.cars_per_env];      /* one Driver per vehicle, custom.py:1398-1411 */
    switch (policy) {
    case FTGP_POLICY_LOBOTOMY: a->u_speed = 0.0; a->u_steer = 0.0; break;
    case FTGP_POLICY_NIDC: policy_disparity(e, r, a, 0, &a->u_speed, &a->u_steer); break;
    case FTGP_POLICY_FAST: policy_disparity(e, r, a, 1, &a->u_speed, &a->u_steer); break;
    case FTGP_POLICY_RANDOM: {
        uint64_t h = splitmix64(e->cfg.seed + (uint64_t)((long)e->cfg.env_base * e->cfg.cars_per_env + ci) * 0x9E3779B97F4A7C15ull);
        h = splitmix64(h ^ (uint64_t)e->steps[env]);
        a->u_speed = 3.0 * u01(h);
        a->u_steer = 2.0 * u01(splitmix64(h)) - 1.0;
        break; }
    default: break;
    }
}

/* standalone driver evaluation for the G1 golden vectors: ranges float[n], state = last_steering_angle in/out */
int oracle_policy_eval1(int policy, int n_rays, const float *ranges, double *last_steer, double *speed, double *steer)
{
    OracleEnv tmp; memset(&tmp, 0, sizeof tmp); tmp.cfg.n_rays = n_rays;
    Car c; memset(&c, 0, sizeof c); c.last_steer = *last_steer;
    if (policy == FTGP_POLICY_LOBOTOMY) { *speed = 0; *steer = 0; return 0; }
    if (policy != FTGP_POLICY_NIDC && policy != FTGP_POLICY_FAST) return fail(FTGP_ERR_ARG, "policy_eval: nidc/fast/lobotomy only");
    policy_disparity(&tmp, ranges, &c, policy == FTGP_POLICY_FAST, speed, steer);
    *last_steer = c.last_steer;
    return 0;
}

/* ------------------------------------------------------------------ K4: reset / spawn */
static int spawn_index(const OracleEnv *e, int env, int car)
{
    if (e->cfg.spawn_mode == 0) return (car + 5) * 2;             /* custom.py:1112 */
    return (int)((10 + 7 * (long)(e->cfg.env_base + env) + 2 * car) % 98);
}

static void reset_car(OracleEnv *e, int ci)
{
    const int cpe = e->cfg.cars_per_env;
    const int env = ci / cpe, car = ci % cpe;
    Car *a = &e->cars[ci];
    memset(a, 0, sizeof *a);
    int p = spawn_index(e, env, car);
    a->offset = p;
    a->good_start = 1;
    a->x = e->spawn[p][0]; a->y = e->spawn[p][1];
    double qw = e->spawn[p][2], qz = e->spawn[p][3];
    if (e->cfg.spawn_mode == 1) {
        /* yaw jitter U(-0.1, 0.1) rad, keyed (seed, car index) */
        uint64_t h = splitmix64(e->cfg.seed ^ (0xA0761D6478BD642Full + (uint64_t)((long)e->cfg.env_base * cpe + ci)));
        double j = 0.2 * u01(h) - 0.1;
        double cj = spec_cos(0.5 * j), sj = spec_sin(0.5 * j);
        double nw = qw * cj - qz * sj, nz = qz * cj + qw * sj;
        double n = sqrt(nw * nw + nz * nz);
        qw = nw / n; qz = nz / n;
    }
    a->qw = qw; a->qz = qz;
    memset(e->ranges + (size_t)ci * e->cfg.n_rays, 0, sizeof(float) * (size_t)e->cfg.n_rays);
}

/* ------------------------------------------------------------------ public API (mirrors include/ftgp.h) */
int oracle_create(const FtgpConfig *cfg, OracleEnv **out)
{
    if (!cfg || !out) return fail(FTGP_ERR_ARG, "null argument");
    if (cfg->abi_version != FTGP_ABI_VERSION) return fail(FTGP_ERR_ARG, "abi version mismatch");
    if (cfg->n_envs < 1 || cfg->cars_per_env < 1 || cfg->cars_per_env > 8 || cfg->n_rays < 1)
        return fail(FTGP_ERR_ARG, "bad n_envs / cars_per_env / n_rays");
    if (cfg->spawn_mode == 0 && (cfg->cars_per_env + 4) * 2 + 1 >= NPATH) return fail(FTGP_ERR_ARG, "too many cars for reference spawn");
    const FtgpTrack *t = &cfg->track;
    if (t->width < 1 || t->height < 1 || !t->bits || !t->path || t->words_per_row < (t->width + 31) / 32)
        return fail(FTGP_ERR_ARG, "bad track");
    OracleEnv *e = (OracleEnv *)calloc(1, sizeof *e);
    e->cfg = *cfg;
    size_t nw = (size_t)t->height * t->words_per_row;
    e->bits = (uint32_t *)malloc(nw * 4); memcpy(e->bits, t->bits, nw * 4);
    memcpy(e->path, t->path, sizeof e->path);
    e->cfg.track.bits = e->bits; e->cfg.track.path = &e->path[0][0];
    e->field = (uint8_t *)malloc((size_t)t->width * t->height);
    build_field(e);
    e->cfg.fan_dirs = NULL;                               /* copied into ray_bxd / ray_byd below: the caller's buffer is not kept */
    if (cfg->lidar_mode == FTGP_LIDAR_FAKELIDAR) {
        e->edt = (double *)malloc(sizeof(double) * (size_t)t->width * t->height);
        build_edt(e);
    }
    const int R = cfg->n_rays;
    e->ray_bx = (float *)malloc(sizeof(float) * R); e->ray_by = (float *)malloc(sizeof(float) * R);
    e->ray_bxd = (double *)malloc(sizeof(double) * R); e->ray_byd = (double *)malloc(sizeof(double) * R);
    for (int j = 0; j < R; ++j) {
        /* mushr.em.xml:112-117: phi = radians(360/R*j - 90); ray = (sin phi, -cos phi) */
        double phi = ((360.0 / (double)R) * (double)j - 90.0) * (M_PI / 180.0);
        e->ray_bxd[j] = cfg->fan_dirs ? cfg->fan_dirs[2 * j] : sin(phi);
        e->ray_byd[j] = cfg->fan_dirs ? cfg->fan_dirs[2 * j + 1] : -cos(phi);
        e->ray_bx[j] = (float)e->ray_bxd[j]; e->ray_by[j] = (float)e->ray_byd[j];
        /* the rangefinders' own fan is point-symmetric, and the BINARY32 table says so to the last bit: site j + R/2 = -(site j) (DESIGN.md
         * section 4).  The binary64 fan (FAKELIDAR mode) stays libm's sin / cos of every phi_j: what include/ftgp.h documents for fan_dirs == NULL */
        if (!cfg->fan_dirs && R % 2 == 0 && j >= R / 2) { e->ray_bx[j] = -e->ray_bx[j - R / 2]; e->ray_by[j] = -e->ray_by[j - R / 2]; }
    }
    for (int p = 0; p < NPATH; ++p) {
        /* custom.py:1240-1245 + 81-87 with pitch = roll = 0 */
        int p1 = (p + 1) % NPATH;
        double ang = atan2(e->path[p1][1] - e->path[p][1], e->path[p1][0] - e->path[p][0]);
        e->spawn[p][0] = e->path[p][0]; e->spawn[p][1] = e->path[p][1];
        e->spawn[p][2] = cos(ang / 2); e->spawn[p][3] = sin(ang / 2);
    }
    const FtgpVehicle *v = &cfg->vehicle;
    double wtot = v->mass * v->gravity;
    if (v->kind == FTGP_VEHICLE_TRICYCLE) {           /* two driven wheels behind the origin, the caster (wheel 2) in front */
        double a_f = v->wheel_x[2], a_r = -0.5 * (v->wheel_x[0] + v->wheel_x[1]);
        e->wheel_load[0] = e->wheel_load[1] = 0.5 * (wtot * (a_f / (a_f + a_r)));
        e->wheel_load[2] = wtot * (a_r / (a_f + a_r)); e->wheel_load[3] = 0.0;
    } else {
        double a_f = 0.5 * (v->wheel_x[0] + v->wheel_x[1]), a_r = -0.5 * (v->wheel_x[2] + v->wheel_x[3]);
        e->wheel_load[0] = e->wheel_load[1] = 0.5 * (wtot * (a_r / (a_f + a_r)));
        e->wheel_load[2] = e->wheel_load[3] = 0.5 * (wtot * (a_f / (a_f + a_r)));
    }
    e->n_cars = cfg->n_envs * cfg->cars_per_env;
    e->cars = (Car *)calloc((size_t)e->n_cars, sizeof(Car));
    e->ranges = (float *)calloc((size_t)e->n_cars * R, sizeof(float));
    e->steps = (int64_t *)calloc((size_t)cfg->n_envs, sizeof(int64_t));
    e->place = (int32_t *)calloc((size_t)e->n_cars, sizeof(int32_t));
    e->n_winners = (int32_t *)calloc((size_t)cfg->n_envs, sizeof(int32_t));
    e->threads = 1;
    *out = e;
    extern int oracle_reset(OracleEnv *, const uint8_t *);
    return oracle_reset(e, NULL);
}

int oracle_destroy(OracleEnv *e)
{
    if (!e) return 0;
    free(e->edt); free(e->bits); free(e->field); free(e->ray_bx); free(e->ray_by); free(e->ray_bxd); free(e->ray_byd);
    free(e->cars); free(e->ranges); free(e->steps); free(e->place); free(e->n_winners); free(e);
    return 0;
}

int oracle_set_threads(OracleEnv *e, int n) { e->threads = n < 1 ? 1 : n; return 0; }
int oracle_set_lidar_mode(OracleEnv *e, int mode) { e->lidar_mode = mode; return 0; }

int oracle_reset(OracleEnv *e, const uint8_t *mask)
{
    const int cpe = e->cfg.cars_per_env;
    for (int env = 0; env < e->cfg.n_envs; ++env) {
        if (mask && !mask[env]) continue;
        e->steps[env] = 0;
        e->n_winners[env] = 0;                                           /* self.winners = {} (custom.py:1125) */
        for (int k = 0; k < cpe; ++k) e->place[env * cpe + k] = 0;
        for (int k = 0; k < cpe; ++k) reset_car(e, env * cpe + k);
        for (int k = 0; k < cpe; ++k) progress_car(e, env * cpe + k);
    }
    return 0;
}

int oracle_set_ctrl(OracleEnv *e, const double *ctrl, const uint8_t *car_mask)
{
    for (int i = 0; i < e->n_cars; ++i) {
        if (car_mask && !car_mask[i]) continue;
        e->cars[i].u_speed = ctrl[2 * i]; e->cars[i].u_steer = ctrl[2 * i + 1];
    }
    return 0;
}

static void step_env(OracleEnv *e, int env, int policy)
{
    const int cpe = e->cfg.cars_per_env;
    Car next[8];
    if (policy != FTGP_POLICY_HOST)
        for (int k = 0; k < cpe; ++k) policy_car(e, policy, env * cpe + k);
    for (int k = 0; k < cpe; ++k) lidar_car(e, env * cpe + k);          /* sensors at the pre-integration pose */
    for (int k = 0; k < cpe; ++k) integrate_car(e, env * cpe + k, &next[k]);
    for (int k = 0; k < cpe; ++k) e->cars[env * cpe + k] = next[k];
    e->steps[env] += 1;
    for (int k = 0; k < cpe; ++k) progress_car(e, env * cpe + k);
}

static int run(OracleEnv *e, int policy, int n_steps)
{
    if (n_steps < 0) return fail(FTGP_ERR_ARG, "n_steps < 0");
#ifdef _OPENMP
    #pragma omp parallel for schedule(dynamic, 4) num_threads(e->threads)
#endif
    for (int env = 0; env < e->cfg.n_envs; ++env)
        for (int s = 0; s < n_steps; ++s) step_env(e, env, policy);
    return 0;
}
int oracle_step(OracleEnv *e, int n_steps) { return run(e, FTGP_POLICY_HOST, n_steps); }
int oracle_rollout(OracleEnv *e, int policy, int n_steps)
{
    if (policy < 0 || policy > FTGP_POLICY_PER_CAR) return fail(FTGP_ERR_ARG, "unknown policy");
    if (policy == FTGP_POLICY_PER_CAR && !e->car_policy[0]) return fail(FTGP_ERR_STATE, "FTGP_POLICY_PER_CAR without ftgp_set_car_policies");
    return run(e, policy, n_steps);
}

int oracle_get_lidar(OracleEnv *e, float *out) { memcpy(out, e->ranges, sizeof(float) * (size_t)e->n_cars * e->cfg.n_rays); return 0; }

/* custom.py:62-76 */
static void quat_to_euler(double w, double x, double y, double z, double *yaw, double *pitch, double *roll)
{
    double t0 = +2.0 * (w * x + y * z), t1 = +1.0 - 2.0 * (x * x + y * y);
    *roll = atan2(t0, t1);
    double t2 = +2.0 * (w * y - z * x);
    t2 = t2 > +1.0 ? +1.0 : t2; t2 = t2 < -1.0 ? -1.0 : t2;
    *pitch = asin(t2);
    double t3 = +2.0 * (w * z + x * y), t4 = +1.0 - 2.0 * (y * y + z * z);
    *yaw = atan2(t3, t4);
}
static int lap_completion(const Car *a) { return a->good_start ? a->completion : -(100 - a->completion); }  /* custom.py:132-140 */

int oracle_get_snapshot(OracleEnv *e, double *out)
{
    for (int i = 0; i < e->n_cars; ++i) {
        const Car *a = &e->cars[i]; double *o = out + (size_t)i * FTGP_SNAPSHOT_DOUBLES;
        double yaw, pitch, roll; quat_to_euler(a->qw, 0.0, 0.0, a->qz, &yaw, &pitch, &roll);
        int lc = lap_completion(a);
        o[0] = a->laps; o[1] = a->vx; o[2] = a->vy; o[3] = 0.0; o[4] = yaw; o[5] = pitch; o[6] = roll;
        o[7] = lc; o[8] = a->laps * 100 + lc;
        o[9] = (double)e->steps[i / e->cfg.cars_per_env] / e->cfg.dt;   /* custom.py:1397 (sic) */
    }
    return 0;
}
int oracle_get_pose(OracleEnv *e, double *out)
{
    for (int i = 0; i < e->n_cars; ++i) {
        const Car *a = &e->cars[i]; double *o = out + (size_t)i * FTGP_POSE_DOUBLES;
        o[0] = a->x; o[1] = a->y; o[2] = e->cfg.vehicle.body_z; o[3] = a->qw; o[4] = 0; o[5] = 0; o[6] = a->qz;
        o[7] = a->vx; o[8] = a->vy; o[9] = 0; o[10] = 0; o[11] = 0; o[12] = a->wz;
    }
    return 0;
}
int oracle_set_pose(OracleEnv *e, const double *pose)
{
    for (int i = 0; i < e->n_cars; ++i) {
        Car *a = &e->cars[i]; const double *o = pose + (size_t)i * FTGP_POSE_DOUBLES;
        double n = sqrt(o[3] * o[3] + o[6] * o[6]);
        a->x = o[0]; a->y = o[1]; a->qw = o[3] / n; a->qz = o[6] / n; a->vx = o[7]; a->vy = o[8]; a->wz = o[12];
    }
    return 0;
}
int oracle_get_progress(OracleEnv *e, int32_t *out)
{
    for (int i = 0; i < e->n_cars; ++i) {
        const Car *a = &e->cars[i]; int32_t *o = out + (size_t)i * FTGP_PROGRESS_INTS;
        int lc = lap_completion(a);
        o[0] = a->laps; o[1] = a->completion; o[2] = lc; o[3] = a->laps * 100 + lc; o[4] = a->finished;
        const int64_t top = 0x7fffffffll;                 /* the row is int32: start and finish_step saturate (oracle_get_race_steps has all 64 bits) */
        o[5] = a->off_track; o[6] = (int32_t)(a->start > top ? top : a->start); o[7] = a->good_start; o[8] = a->delta;
        o[9] = a->finished ? (int32_t)(a->finish_step > top ? top : a->finish_step) : -1;
    }
    return 0;
}
int oracle_get_winners(OracleEnv *e, int32_t *out) { memcpy(out, e->place, sizeof(int32_t) * (size_t)e->n_cars); return 0; }
int oracle_get_lap_times(OracleEnv *e, int32_t *counts, double *times)
{
    for (int i = 0; i < e->n_cars; ++i) {
        counts[i] = e->cars[i].n_times;
        memcpy(times + (size_t)i * FTGP_MAX_LAP_TIMES, e->cars[i].times, sizeof(double) * FTGP_MAX_LAP_TIMES);
    }
    return 0;
}
int oracle_get_ctrl(OracleEnv *e, double *out)
{
    for (int i = 0; i < e->n_cars; ++i) { out[2 * i] = e->cars[i].u_speed; out[2 * i + 1] = e->cars[i].u_steer; }
    return 0;
}
int oracle_get_steps(OracleEnv *e, int64_t *out) { memcpy(out, e->steps, sizeof(int64_t) * (size_t)e->cfg.n_envs); return 0; }
int oracle_policy_eval(OracleEnv *e, int policy, const float *ranges, double *ctrl_out)
{
    if (policy < FTGP_POLICY_LOBOTOMY || policy > FTGP_POLICY_PER_CAR) return fail(FTGP_ERR_ARG, "policy_eval: device policies only");
    if (policy == FTGP_POLICY_PER_CAR && !e->car_policy[0]) return fail(FTGP_ERR_STATE, "FTGP_POLICY_PER_CAR without ftgp_set_car_policies");
    memcpy(e->ranges, ranges, sizeof(float) * (size_t)e->n_cars * e->cfg.n_rays);
    for (int i = 0; i < e->n_cars; ++i) {
        policy_car(e, policy, i);
        if (ctrl_out) { ctrl_out[2 * i] = e->cars[i].u_speed; ctrl_out[2 * i + 1] = e->cars[i].u_steer; }
    }
    return 0;
}
int oracle_set_car_policies(OracleEnv *e, const int32_t *policies)
{
    if (!e || !policies) return fail(FTGP_ERR_ARG, "null argument");
    for (int k = 0; k < e->cfg.cars_per_env; ++k)
        if (policies[k] < FTGP_POLICY_LOBOTOMY || policies[k] > FTGP_POLICY_RANDOM) return fail(FTGP_ERR_ARG, "set_car_policies: lobotomy / nidc / fast / random only");
    for (int k = 0; k < e->cfg.cars_per_env; ++k) e->car_policy[k] = policies[k];
    return 0;
}
int oracle_eval_progress(OracleEnv *e) { for (int i = 0; i < e->n_cars; ++i) progress_car(e, i); return 0; }
int oracle_get_distance_field(OracleEnv *e, double *out)
{
    if (!e->edt) return fail(FTGP_ERR_STATE, "no distance field: lidar_mode is not FTGP_LIDAR_FAKELIDAR");
    memcpy(out, e->edt, sizeof(double) * (size_t)e->cfg.track.width * e->cfg.track.height);
    return 0;
}
int oracle_get_field(OracleEnv *e, uint8_t *out) { memcpy(out, e->field, (size_t)e->cfg.track.width * e->cfg.track.height); return 0; }

int oracle_get_race_steps(OracleEnv *e, int64_t *out)
{
    for (int i = 0; i < e->n_cars; ++i) { out[2 * i] = e->cars[i].start; out[2 * i + 1] = e->cars[i].finished ? e->cars[i].finish_step : -1; }
    return 0;
}

int oracle_metrics_local(OracleEnv *e, double *out)
{
    double steps = 0, laps = 0, absc = 0, fin = 0, off = 0, tmin = INFINITY, tmax = -INFINITY;
    for (int env = 0; env < e->cfg.n_envs; ++env) steps += (double)e->steps[env];
    for (int i = 0; i < e->n_cars; ++i) {
        const Car *a = &e->cars[i];
        laps += a->laps; absc += a->laps * 100 + lap_completion(a); fin += a->finished; off += a->off_track;
        int n = a->n_times < FTGP_MAX_LAP_TIMES ? a->n_times : FTGP_MAX_LAP_TIMES;
        for (int k = 0; k < n; ++k) { if (a->times[k] < tmin) tmin = a->times[k]; if (a->times[k] > tmax) tmax = a->times[k]; }
    }
    out[0] = steps; out[1] = e->n_cars; out[2] = laps; out[3] = absc; out[4] = fin; out[5] = off; out[6] = tmin; out[7] = tmax;
    return 0;
}

/* ------------------------------------------------------------------ fakelidar-compat (raycast.py:5-21) */
int oracle_fakelidar(double orig_x, double orig_y, const double *dt, int H, int W, int rangefinders,
                     const double *cosines, const double *sines, double eps, double *scan, double *points)
{
    for (int i = 0; i < rangefinders; ++i) {
        double x = orig_x, y = orig_y, dx = cosines[i], dy = sines[i];
        double distance = 0;
        /* Python int() truncates toward zero; negative indices wrap (numpy) */
        long yi = (long)y, xi = (long)x;
        if (yi < 0) yi += H; if (xi < 0) xi += W;
        if (yi < 0 || yi >= H || xi < 0 || xi >= W) return fail(FTGP_ERR_ARG, "fakelidar: IndexError");
        double nearest = dt[(size_t)yi * W + xi];
        while (nearest > eps && 0 <= x && x <= W && 0 <= y && y <= H) {
            distance += nearest;
            x += dx * nearest;
            y += dy * nearest;
            yi = (long)y; xi = (long)x;
            if (yi < 0) yi += H; if (xi < 0) xi += W;
            if (yi < 0 || yi >= H || xi < 0 || xi >= W) return fail(FTGP_ERR_ARG, "fakelidar: IndexError");
            nearest = dt[(size_t)yi * W + xi];
        }
        scan[i] = distance; points[2 * i] = x; points[2 * i + 1] = y;
    }
    return 0;
}

/* custom.py:62-76 / 81-87 exposed for the G4 vectors */
int oracle_quaternion_to_euler(double w, double x, double y, double z, double *out3)
{
    quat_to_euler(w, x, y, z, &out3[0], &out3[1], &out3[2]);
    return 0;
}
int oracle_euler_to_quaternion(const double *r, double *out4)
{
    double yaw = r[0], pitch = r[1], roll = r[2];
    out4[0] = cos(roll / 2) * cos(pitch / 2) * cos(yaw / 2) + sin(roll / 2) * sin(pitch / 2) * sin(yaw / 2);
    out4[1] = sin(roll / 2) * cos(pitch / 2) * cos(yaw / 2) - cos(roll / 2) * sin(pitch / 2) * sin(yaw / 2);
    out4[2] = cos(roll / 2) * sin(pitch / 2) * cos(yaw / 2) + sin(roll / 2) * cos(pitch / 2) * sin(yaw / 2);
    out4[3] = cos(roll / 2) * cos(pitch / 2) * sin(yaw / 2) - sin(roll / 2) * sin(pitch / 2) * cos(yaw / 2);
    return 0;
}

/* Drive the lap logic alone with a prescribed sequence of (closest index, off_track) observations (G5 traces). */
int oracle_progress_trace(int offset, int lap_target, double dt, int n, const int32_t *closest, const uint8_t *off_track,
                          int32_t *out /* [n][6]: laps, completion, good_start, n_times, finished, delta */, double *times_out)
{
    Car a; memset(&a, 0, sizeof a); a.offset = offset; a.good_start = 1;
    for (int s = 0; s < n; ++s) {
        a.off_track = off_track[s];
        if (!a.off_track) {
            int completion = ((closest[s] - a.offset) % 100 + 100) % 100;
            int delta = completion - a.completion;
            a.delta = (((completion - a.completion + 50) % 100) + 100) % 100 - 50;
            if (abs(delta) > 90) {
                double lap_time = (double)(s - a.start) * dt;
                if (a.delta < 0) { a.laps -= 1; a.good_start = 0; if (a.n_times != 0) { if (a.n_times > FTGP_MAX_LAP_TIMES) a.times[(a.n_times - 1) % FTGP_MAX_LAP_TIMES] = NAN; a.n_times -= 1; } }
                else if (a.delta > 0) {
                    if (a.good_start) { a.times[a.n_times % FTGP_MAX_LAP_TIMES] = lap_time; a.n_times += 1; a.start = s; }
                    a.laps += 1; a.good_start = 1;
                }
            }
            if (a.laps >= lap_target) a.finished = 1;
            a.completion = completion;
        }
        int32_t *o = out + (size_t)s * 6;
        o[0] = a.laps; o[1] = a.completion; o[2] = a.good_start; o[3] = a.n_times; o[4] = a.finished; o[5] = a.delta;
    }
    if (times_out) memcpy(times_out, a.times, sizeof a.times);
    return 0;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
