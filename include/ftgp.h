/*
 * ftgp.h -- C-ABI of the MI355X-native ft_grandprix hot path
 *           (vehicle integrate + LiDAR sweep + lap progress, batched over envs).
 *
 * This is the drop-in boundary.  The reference has no FFI:
 * its hot loop talks to MuJoCo's Python bindings.  Each entry point below names
 * the reference call sites it replaces (paths relative to the reference repo).
 *
 * Conventions: extern "C", opaque handle, int status (0 = ok, negative = error,
 * text via ftgp_last_error()), plain pointers and sizes only.  The caller owns
 * all host buffers; device buffers are owned by the handle.  A handle is not
 * thread-safe; independent handles are.
 *
 * Layout conventions for per-car arrays: index = (env * cars_per_env + car).
 */
#ifndef FTGP_H
#define FTGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTGP_ABI_VERSION 5

/* status codes */
#define FTGP_OK              0
#define FTGP_ERR_ARG        -1   /* bad argument / config */
#define FTGP_ERR_NO_DEVICE  -2   /* no HIP device (the product path has no CPU fallback) */
#define FTGP_ERR_HIP        -3   /* HIP runtime error */
#define FTGP_ERR_STATE      -4   /* call not valid in the handle's current state */
#define FTGP_ERR_COMM       -5   /* RCCL error */

/* device-side policies for ftgp_rollout; FTGP_POLICY_HOST = ctrl comes from ftgp_set_ctrl */
#define FTGP_POLICY_HOST      0
#define FTGP_POLICY_LOBOTOMY  1  /* ft_grandprix/lobotomy.py:2-3 : (0, 0)                    */
#define FTGP_POLICY_NIDC      2  /* ft_grandprix/nidc.py:116-131 : disparity extender         */
#define FTGP_POLICY_FAST      3  /* ft_grandprix/fast.py:118-139 : same + straight-line boost */
#define FTGP_POLICY_RANDOM    4  /* counter-based RNG keyed (seed, car, step): speed~U(0,3), steer~U(-1,1) */
#define FTGP_POLICY_PER_CAR   5  /* every car slot of an env its own driver, as set by ftgp_set_car_policies (the roster) */

#define FTGP_VEHICLE_MUSHR     0
#define FTGP_VEHICLE_TRICYCLE  1

/* what the rangefinders are (FtgpConfig.lidar_mode) */
#define FTGP_LIDAR_RANGEFINDER 0  /* exact 2-D ray against the wall pixels and the other cars (template/mushr.em.xml:112-117,204-206 as read at
                                     custom.py:1395) */
#define FTGP_LIDAR_FAKELIDAR   1  /* the reference's own 2-D LiDAR: sphere tracing over the Euclidean distance transform of the track image,
                                     ft_grandprix/raycast.py:5-21, wired as custom.py:1381-1393 (option use_simulated_simulation_lidar) */

#define FTGP_PATH_POINTS   100   /* ft_grandprix/curve.py:8 */
#define FTGP_MAX_LAP_TIMES  32   /* lap times kept per car: a ring of the NEWEST 32 -- lap time number k (0-based, in the order VehicleState.times
                                    lists them, custom.py:124,1351-1363) sits in slot k % 32; the true count is kept beside it.  VehicleState.times
                                    is unbounded; lap_target defaults to 10 (custom.py:961).  A backward crossing pops the newest entry (custom.py:1355-1356):
                                    while more than 32 lap times have been counted, the popped entry had overwritten the oldest one the list would
                                    still show -- that slot then reads NaN (= no entry; skipped by the metrics record's min / max) until it is filled again */

/* number of doubles / ints per car in the packed read-back rows */
#define FTGP_SNAPSHOT_DOUBLES 10 /* laps, vel[3], yaw, pitch, roll, lap_completion, absolute_completion, time */
#define FTGP_POSE_DOUBLES     13 /* qpos[7] = x y z qw qx qy qz ; qvel[6] = vx vy vz wx wy wz */
#define FTGP_PROGRESS_INTS    10 /* laps, completion, lap_completion, absolute_completion, finished, off_track, start, good_start, delta,
                                    finish_step: the env step at which `finished` was set (custom.py:1367-1370), -1 while racing;
                                    the row is int32 while steps are int64: saturates at 2^31 - 1 (99 days of simulated time) */
#define FTGP_METRIC_DOUBLES    8 /* steps, n_cars, sum_laps, sum_abs_completion, n_finished, n_off_track, min_lap_time, max_lap_time */

/*
 * Track geometry (built by ft_grandprix_amd/track.py from <track>.png + <track>-path.svg).
 *   wall bitmap : ft_grandprix/chunk.py:39-43 threshold (wall iff pure white)
 *   wall frame  : template/mushr.em.xml:17-20,55,92 (pixel (px,py) covers
 *                 x in [origin_x + px*px_size_x, +px_size_x), y in (origin_y - (py+1)*px_size_y, origin_y - py*px_size_y])
 *   path        : ft_grandprix/curve.py:6-18 + ft_grandprix/custom.py:1184-1186 (100 x (x, y), float64)
 */
typedef struct FtgpTrack {
    int32_t width, height;          /* pixels */
    int32_t words_per_row;          /* uint32 words per bitmap row = ceil(width/32) */
    int32_t reserved0;
    const uint32_t *bits;           /* [height][words_per_row]; bit (x & 31) of word (x >> 5), 1 = wall */
    double px_size_x, px_size_y;    /* world units per pixel */
    double origin_x, origin_y;      /* world position of the top-left corner of pixel (0, 0) */
    const double *path;             /* [FTGP_PATH_POINTS][2] world coordinates */
} FtgpTrack;

/*
 * Vehicle parameters: the reduced planar model of the MuSHR car of
 * template/mushr.em.xml:61-89,95-198 (see DESIGN.md "K1").  ftgp_default_vehicle() fills
 * the values lifted from that file.
 */
typedef struct FtgpVehicle {
    double mass, izz;               /* total mass, yaw inertia about the CoM */
    double wheel_x[4], wheel_y[4];  /* fl, fr, bl, br contact points, body frame (mushr.em.xml:124,137,150,162) */
    double wheel_radius;            /* 0.03 (mushr.em.xml:24,69) */
    double wheel_inertia;           /* spin inertia incl. armature 0.01 (mushr.em.xml:81) */
    double wheel_damping;           /* 0.01 (mushr.em.xml:81) */
    double throttle_kv, throttle_gear, throttle_force_limit; /* 100, 0.04, 500 (mushr.em.xml:180) */
    double steer_kp, steer_damping, steer_inertia, steer_limit; /* 20, 3*0.1, ~8e-4, 1 rad (mushr.em.xml:78,179) */
    double friction, gravity;       /* 0.5 = max(wheel 0.3, plane 0.5) (mushr.em.xml:69,94); 9.81 */
    double tire_damping;            /* slip-velocity coupling per wheel, N s/m (from solref 0.02/solimp 0.95, mushr.em.xml:69) */
    double contact_x[3];            /* body-frame x of the 3 wall/car contact circles */
    double contact_radius;
    double contact_stiffness, contact_damping; /* penalty spring/damper against walls and other cars */
    double lidar_x, lidar_y;        /* LiDAR centre, body frame: (-0.0525, 0) (mushr.em.xml:101) */
    double lidar_ring_radius;       /* 0.03: ray j starts at centre - 0.03*dir_j (mushr.em.xml:103,115) */
    double body_z;                  /* constant ride height reported in qpos[2] */
    double box_xmin, box_xmax, box_ymin, box_ymax; /* chassis bbox, body frame, seen by other cars' rays */
    double softener_radius;         /* bubble_wrap: wall-contact circles at the four wheel positions; 0.65 * 0.0488 = radius of
                                       meshes/mushr_wheel.stl at the scale of mushr.em.xml:39 (softener geoms, mushr.em.xml:65-67) */
    double motor_forward_limit, motor_turn_limit; /* FTGP_VEHICLE_TRICYCLE: ctrlrange of the two torque motors (car.em.xml:138-139) */
    int32_t kind;                   /* FTGP_VEHICLE_MUSHR: Ackermann car with a velocity servo and a steering servo (mushr.em.xml);
                                       FTGP_VEHICLE_TRICYCLE: the legacy differential-drive car of template/car.em.xml (option tricycle_mode,
                                       custom.py:1154-1170): wheels 0 / 1 = left / right driven wheels, wheel 2 = frictionless front caster,
                                       ctrl = (forward torque, turn torque) on the tendons 0.5 (l + r) and 0.5 (r - l) (car.em.xml:126-139) */
    int32_t reserved1;
} FtgpVehicle;

typedef struct FtgpConfig {
    int32_t abi_version;            /* FTGP_ABI_VERSION */
    int32_t n_envs;
    int32_t cars_per_env;           /* 1..8; cars of one env share a world (template/cars/cars.json) */
    int32_t n_rays;                 /* rangefinders per car (custom.py:1158 uses 90; BASELINE uses 1080) */
    int32_t lap_target;             /* custom.py:961, used custom.py:1367 */
    int32_t device_id;              /* HIP device ordinal */
    int32_t spawn_mode;             /* 0 = reference: car i at path[(i+5)*2] (custom.py:1112,1232-1245);
                                       1 = benchmark spread: global env e, car i at path[(10 + 7*e + 2*i) % 98] with seeded yaw jitter */
    int32_t env_base;               /* global index of this handle's env 0: a shard [env_base, env_base + n_envs) of a larger batch spawns,
                                       jitters and draws random controls exactly like the same slice of the monolithic batch */
    uint64_t seed;                  /* spawn jitter and FTGP_POLICY_RANDOM */
    double dt;                      /* 0.004 (mushr.em.xml:30) */
    int32_t bubble_wrap;            /* option "bubble_wrap" (custom.py:970,1041-1055): the four wheel softeners (mushr.em.xml:65-67,126-129)
                                       collide with the walls -- here: four more wall-contact circles at the wheel positions */
    int32_t naive_flatten;          /* option "naive_flatten" (custom.py:981,1338-1339): re-projects the body quaternion onto pure yaw every
                                       step; the planar model has no pitch / roll, so this is accepted and changes nothing */
    int32_t lidar_mode;             /* FTGP_LIDAR_RANGEFINDER (default) or FTGP_LIDAR_FAKELIDAR (option "use_simulated_simulation_lidar",
                                       custom.py:987,1381-1393).  FAKELIDAR, per car and step, in binary64:
                                         origin   i_x = (x / map_size) * width, i_y = -(y / map_size) * height of the car's position (custom.py:1382-1384)
                                         ray j    image-frame direction (dxw, -dyw), (dxw, dyw) = R(yaw) * fan_dirs[j] -- ray order and orientation
                                                  as the rangefinders' (index 0 = rear, counter-clockwise; the dead branch's own linspace,
                                                  custom.py:1387, was never exercised: SURVEY.md 8a-3)
                                         march    raycast.py:5-21 on the exact Euclidean distance transform of the wall image (the recipe of
                                                  custom.py:1149-1153 / raycast.py:24-27, built at ftgp_create: integer squared distances, one sqrt)
                                         range    (scan / width) * map_size (custom.py:1392-1393), stored as binary32
                                       int() truncates toward zero and negative indices wrap like numpy's; a lookup past the right / bottom edge -- the
                                       reference's IndexError -- ends the ray with range -1.  The rays see walls only (no other cars), as there. */
    int32_t reserved2;
    double map_size;                /* FAKELIDAR: world size of the map, 20 * scale = 40 (custom.py:1155,1382; mushr.em.xml:16-18); <= 0 means 40 */
    const double *fan_dirs;         /* optional [n_rays][2]: body-frame unit directions of the rangefinder fan; NULL = the sites of
                                       template/mushr.em.xml:112-117, (sin phi_j, -cos phi_j) with phi_j = radians(360 / n_rays * j - 90).
                                       FAKELIDAR mode uses the binary64 values as they are.  RANGEFINDER mode uses their binary32 roundings -- for
                                       fan_dirs == NULL and an even n_rays with the second half of the table written as the exact negation of the
                                       first (site j + n/2 looks exactly opposite to site j: the sweep derives a ray from its opposite); a caller's
                                       fan is rounded entry by entry. */
    FtgpTrack track;
    FtgpVehicle vehicle;
} FtgpConfig;

typedef struct FtgpEnv FtgpEnv;

/* Fill *v with the MuSHR constants of template/mushr.em.xml. */
void ftgp_default_vehicle(FtgpVehicle *v);

/* Fill *v with the constants of the legacy tricycle of template/car.em.xml (use FtgpConfig.dt = 0.0075, car.em.xml:11). */
void ftgp_tricycle_vehicle(FtgpVehicle *v);

/* Text of the last error raised on the calling thread. */
const char *ftgp_last_error(void);

/* Number of visible HIP devices (<= 0 when there is none). */
int ftgp_device_count(void);

/*
 * Build the world.  Replaces Mujoco.stage(): chunk() + produce_mjcf() + MjModel.from_xml_path +
 * MjData + path load (ft_grandprix/custom.py:1133-1194; drive.py:21-46).  Uploads the track,
 * builds the ray-march acceleration grid, allocates per-car state and calls ftgp_reset(NULL).
 */
int ftgp_create(const FtgpConfig *cfg, FtgpEnv **out);
int ftgp_destroy(FtgpEnv *env);

/*
 * Reset.  Replaces Mujoco.reload() = mj_resetData + VehicleState rebuild + position_vehicles
 * (custom.py:1089-1128,1232-1245,81-87).  mask: NULL = all envs, else uint8[n_envs], non-zero = reset.
 * After reset: qvel = 0, ctrl = 0, LiDAR ranges = 0 (custom.py:1092), steps of the
 * env = 0, race state cleared, progress evaluated once at the spawn pose.
 */
int ftgp_reset(FtgpEnv *env, const uint8_t *mask);

/*
 * Controls.  Replaces data.ctrl[forward] = speed; data.ctrl[turn] = steering_angle
 * (custom.py:1421-1423; drive.py:82-83).  ctrl: double[n_envs*cars_per_env][2] = (speed, steering_angle).
 * car_mask (may be NULL): uint8 per car, 0 = leave that car's ctrl unchanged -- the reference's
 * behaviour when a driver raises (custom.py:1409-1411).
 */
int ftgp_set_ctrl(FtgpEnv *env, const double *ctrl, const uint8_t *car_mask);

/*
 * n_steps iterations of: sensors at the current pose -> integrate one dt -> steps += 1 ->
 * lap progress at the new pose.  Replaces mujoco.mj_step + steps += 1 (custom.py:1425-1426;
 * drive.py:89) followed by the progress block of the next loop iteration (custom.py:1340-1372).
 * Controls stay at their last ftgp_set_ctrl value.
 */
int ftgp_step(FtgpEnv *env, int n_steps);

/*
 * Same loop with the driver evaluated on the device between progress and integrate:
 * per step: policy(ranges of the previous step) -> ctrl -> sensors -> integrate -> progress.
 * This is the throughput path; one launch covers all n_steps.
 */
int ftgp_rollout(FtgpEnv *env, int policy, int n_steps);

/*
 * The roster on the device: policies = int32[cars_per_env], the bundled driver (FTGP_POLICY_LOBOTOMY / NIDC / FAST / RANDOM) of
 * car slot k of every env; ftgp_rollout(FTGP_POLICY_PER_CAR, n) and ftgp_policy_eval(FTGP_POLICY_PER_CAR, ...) then evaluate each
 * car with its own driver.  Replaces the per-vehicle Driver() instances the reference builds from the roster's "driver" strings
 * and calls one by one (custom.py:1097-1104,1398-1411; template/cars/cars.json: nidc, fast, nidc).  Survives ftgp_reset.
 */
int ftgp_set_car_policies(FtgpEnv *env, const int32_t *policies);

/* Read-backs (host buffers).  All are synchronous with respect to earlier calls on the handle. */

/* float[n_cars][n_rays]; replaces data.sensordata[vehicle_state.sensors] (custom.py:1395; drive.py:81).
 * Index 0 = rear, counter-clockwise; world units; -1 = no hit; all 0 right after reset, and all 0 for a car that has
 * finished (its rangefinders are switched off when it is sent to the shadow realm, custom.py:1436-1439). */
int ftgp_get_lidar(FtgpEnv *env, float *out);

/* double[n_cars][FTGP_SNAPSHOT_DOUBLES]; replaces VehicleState.snapshot (custom.py:149-160,62-76;
 * vehicle.py:3-12) incl. the reference's time = steps / timestep (custom.py:1397). */
int ftgp_get_snapshot(FtgpEnv *env, double *out);

/* double[n_cars][FTGP_POSE_DOUBLES]; replaces joint.qpos / joint.qvel reads (custom.py:1340; 149-152). */
int ftgp_get_pose(FtgpEnv *env, double *out);

/* int32[n_cars][FTGP_PROGRESS_INTS]; replaces the VehicleState race fields (custom.py:91-143,1340-1372).
 * finish_step orders the finishers of an env: the reference hands out places in the order cars reach lap_target, and within
 * one step in car order (winners[id] = len(winners) + 1 inside the per-car loop, custom.py:1337,1367-1369) -- i.e. by
 * (finish_step, car index).  It survives a multi-step ftgp_rollout, so one launch to the end of a race still says who won. */
int ftgp_get_progress(FtgpEnv *env, int32_t *out);

/* int32[n_cars]: place of each car among the finishers of its env, 1 = winner, 0 = still racing (Mujoco.winners, custom.py:1125,1367-1369). */
int ftgp_get_winners(FtgpEnv *env, int32_t *out);

/* counts: int32[n_cars] = len(VehicleState.times), the TRUE count; times: double[n_cars][FTGP_MAX_LAP_TIMES] = the ring of the newest
 * 32 (lap time k in slot k % 32, see FTGP_MAX_LAP_TIMES); replaces VehicleState.times (custom.py:124,1351-1363). */
int ftgp_get_lap_times(FtgpEnv *env, int32_t *counts, double *times);

/* int64[n_cars][2] = (start, finish_step): the env step of the car's last counted line crossing (vehicle_state.start, custom.py:1362) and
 * the env step at which `finished` was set (-1 while racing) with all 64 bits of self.steps; columns 6 and 9 of ftgp_get_progress hold the
 * same two saturated at 2^31 - 1. */
int ftgp_get_race_steps(FtgpEnv *env, int64_t *out);

/* double[n_cars][2] current controls. */
int ftgp_get_ctrl(FtgpEnv *env, double *out);

/* int64[n_envs] physics steps since each env's last reset (self.steps, custom.py:1124,1426). */
int ftgp_get_steps(FtgpEnv *env, int64_t *out);

/* Overwrite poses (testing / curriculum): double[n_cars][FTGP_POSE_DOUBLES]; only x, y, yaw (from qw, qz), vx, vy, wz are used. */
int ftgp_set_pose(FtgpEnv *env, const double *pose);

/* Evaluate a device policy once on caller-supplied scans: ranges float[n_cars][n_rays] replaces the stored scan,
 * the policy writes each car's controls (and fast.py's last_steering_angle); ctrl_out double[n_cars][2] (may be NULL).
 * Replaces one vehicle_state.driver.process_lidar(ranges) call per car (custom.py:1404). */
int ftgp_policy_eval(FtgpEnv *env, int policy, const float *ranges, double *ctrl_out);

/* Re-evaluate the lap-progress block (custom.py:1340-1372) at the current poses and step counts, without integrating.
 * ftgp_reset ends with this; use it after ftgp_set_pose. */
int ftgp_eval_progress(FtgpEnv *env);

/* Local metrics record (double[FTGP_METRIC_DOUBLES]) reduced on the device. */
int ftgp_metrics_local(FtgpEnv *env, double *out);

/*
 * Multi-GPU: one handle per rank, envs sharded, no data-path collective.
 * The only exchange is the end-of-step metrics all-gather over RCCL.
 *   ftgp_comm_unique_id : rank 0 creates the 128-byte RCCL id; the host ships it to the other ranks.
 *   ftgp_comm_init      : ncclCommInitRank on the handle's device.
 *   ftgp_metrics_allgather : out = double[world_size][FTGP_METRIC_DOUBLES]; runs on a side stream.
 * Nothing in the reference to mirror (it has no collective call sites).
 */
int ftgp_comm_unique_id(uint8_t id_out[128]);
int ftgp_comm_init(FtgpEnv *env, const uint8_t id[128], int rank, int world_size);
int ftgp_metrics_allgather(FtgpEnv *env, double *out);
/*
 * The same exchange in two halves, so that it overlaps the next launch ("side stream, overlapped with the
 * next step kernel"):
 *   ftgp_metrics_allgather_begin : enqueues, behind the most recent ftgp_step / ftgp_rollout launch, the all-gather of the
 *                                  record that launch leaves and the copy to pinned host memory -- on the side stream -- and
 *                                  returns at once.  The caller may launch the next steps right away.
 *   ftgp_metrics_allgather_end   : waits for that exchange only (never for a later launch) and copies the records out.
 * The step kernel writes its record into one of two slots, alternating per launch; a launch that would reuse the slot of an
 * exchange still in flight waits for it on the device.  At most one exchange may be open per handle (a second begin before
 * the end is FTGP_ERR_STATE); ftgp_metrics_allgather() == begin + end.
 */
int ftgp_metrics_allgather_begin(FtgpEnv *env);
int ftgp_metrics_allgather_end(FtgpEnv *env, double *out);

/*
 * fakelidar-compatible 2-D sphere tracing (ft_grandprix/raycast.py:5-21), batched over origins, one ray per lane.
 *   dt        double[H][W]   distance transform of the track image in pixels (the caller computes it, as the
 *                            reference does with scipy: custom.py:1149-1153, raycast.py:24-27)
 *   origins   double[n_origins][2]   (orig_x, orig_y) in pixels
 *   cosines / sines  double[n_origins][rangefinders]
 *   scan      double[n_origins][rangefinders]        accumulated distance per ray (pixels)
 *   points    double[n_origins][rangefinders][2]     end point per ray
 * Same loop as the reference: while dt[int(y), int(x)] > eps and 0 <= x <= W and 0 <= y <= H: advance by dt.
 * int() truncates toward zero and negative indices wrap like numpy's; an index past the end is the reference's
 * IndexError and is reported as FTGP_ERR_ARG.  Standalone: needs no FtgpEnv.
 */
int ftgp_fakelidar(int device_id, const double *dt, int H, int W, int n_origins, const double *origins, int rangefinders,
                   const double *cosines, const double *sines, double eps, double *scan, double *points);

/* FAKELIDAR mode: the distance transform ftgp_create built, double[height][width] in pixels (what the reference calls self.dt,
 * custom.py:1152-1153); FTGP_ERR_STATE in RANGEFINDER mode. */
int ftgp_get_distance_field(FtgpEnv *env, double *out);

/* Self-test of device arithmetic the kernels rely on (no reference counterpart): the fast reciprocal of the ray set-up against
 * the IEEE division of the specification over all 2^32 binary32 bit patterns.  *mismatches = number of differing results. */
int ftgp_selftest(int device_id, int64_t *mismatches);

/* Timing of the most recent ftgp_step / ftgp_rollout launch sequence, measured with HIP events on the handle's stream (ms). */
int ftgp_last_kernel_ms(FtgpEnv *env, float *ms);

/* Name of the kernel that ftgp_rollout/ftgp_step launches for the current configuration (for rocprof matching). */
const char *ftgp_kernel_name(FtgpEnv *env);

/* What the loaded library was built from and with (no reference counterpart; touches no device):
 * "abi=<n> sources=<hash of the kernel sources, tools/evidence.py sha> diag=<diagnostic switches, "none" in the product> fair_shift=<n>
 * waves_per_eu=<n>".  __graft_entry__.build() rebuilds a library whose hash is not the tree's; tests/test_capi.py checks both fields. */
const char *ftgp_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* FTGP_H */
