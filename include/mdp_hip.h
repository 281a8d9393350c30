/*
 * mdp_hip.h -- C ABI of libmdp_hip.so: the MI355X (gfx950) implementation of Isaac Lab's
 * per-step vectorised MDP compute (actuators, sensors, manager terms, math, ray casting).
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer borrowed for the duration of the call (never retained,
 *     never freed) unless the function is documented as a host-side builder;
 *   - tensors are fp32 row-major with the row stride ("ld", in elements) passed explicitly, masks
 *     are uint8 (torch.bool storage), ids are int32 unless noted (torch int64 buffers that the
 *     reference owns -- episode_length_buf, command_counter -- are read as int64);
 *   - every launch is asynchronous on the `stream` argument (a hipStream_t passed as void*);
 *     nothing synchronises, nothing allocates after the create calls;
 *   - return value: 0 = ok, <0 = error class (MDP_ERR_*); mdp_last_error() gives the message.
 *     The Python shim raises the matching exception type (ValueError/RuntimeError).
 *
 * Each entry point cites the reference interface it replaces as `LAB/...:line`
 * (LAB = source/isaaclab/isaaclab, TASKS = source/isaaclab_tasks/isaaclab_tasks).
 */
#ifndef MDP_HIP_H
#define MDP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDP_OK 0
#define MDP_ERR_INVALID (-1)   /* bad argument / shape  -> ValueError   */
#define MDP_ERR_RUNTIME (-2)   /* HIP runtime failure   -> RuntimeError */
#define MDP_ERR_UNSUPPORTED (-3)

int mdp_version(void);
const char* mdp_last_error(void);
/* number of visible devices / name of device 0 (host only, used by smoke tests) */
int mdp_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Actuators  (LAB/actuators/actuator_pd.py, actuator_net.py)
 * ------------------------------------------------------------------------------------------ */
enum {
    MDP_ACT_IMPLICIT = 0, /* ImplicitActuator.compute   actuator_pd.py:62-87   */
    MDP_ACT_IDEAL_PD = 1, /* IdealPDActuator.compute    actuator_pd.py:131-145 */
    MDP_ACT_DC_MOTOR = 2  /* DCMotor.compute/_clip_effort actuator_pd.py:211-233 */
};

/* One actuator group, one physics sub-step.  All matrices are (N, J) with row stride ld_*.
 * q_des/qd_des/tau_ff may be NULL (treated as zeros, which is what Articulation feeds when a
 * target was never set).  computed/applied are the group's `computed_effort` / `applied_effort`.
 * saturation_effort is only read for MDP_ACT_DC_MOTOR (may be +inf, actuator_pd.py:195-198). */
int mdp_actuator_pd_compute(int kind, int num_envs, int num_joints,
                            const float* q_des, int ld_q_des,
                            const float* qd_des, int ld_qd_des,
                            const float* tau_ff, int ld_tau_ff,
                            const float* joint_pos, int ld_pos,
                            const float* joint_vel, int ld_vel,
                            const float* stiffness, const float* damping,
                            const float* effort_limit, const float* velocity_limit,
                            float saturation_effort,
                            float* computed_effort, float* applied_effort,
                            void* stream);

/* CircularBuffer / DelayBuffer (LAB/utils/buffers/circular_buffer.py:107-166,
 * delay_buffer.py:159-177).  buffer is (max_len, N, D); num_pushes int64 (N); pointer is the
 * *new* head slot (host-tracked python int, circular_buffer.py:126).  append back-fills every
 * slot of rows whose num_pushes == 0 (circular_buffer.py:130-133) without a host sync, then
 * increments num_pushes.  If out != NULL it also performs the delayed read
 * buffer[(pointer - min(lag, pushes-1)) mod max_len, env]  (delay_buffer.py:176). */
int mdp_circular_buffer_append(float* buffer, int64_t* num_pushes, int max_len, int num_envs, int dim,
                               int pointer, const float* data, int ld_data,
                               const int32_t* time_lags, float* out, void* stream);
/* buffer property: oldest -> newest, output (N, max_len, D)  (circular_buffer.py:78-86) */
int mdp_circular_buffer_read(const float* buffer, int max_len, int num_envs, int dim, int pointer,
                             float* out, void* stream);
/* reset rows flagged in env_mask (NULL = all): num_pushes = 0, buffer rows = 0 (circular_buffer.py:92-105) */
int mdp_circular_buffer_reset(float* buffer, int64_t* num_pushes, int max_len, int num_envs, int dim,
                              const uint8_t* env_mask, void* stream);

/* LinearInterpolation.compute (LAB/utils/interpolation/linear_interpolation.py:54-85) followed,
 * when efforts != NULL, by RemotizedPDActuator's clamp (actuator_pd.py:357-362):
 * efforts = clamp(efforts, -f(q), +f(q)) in place.  fq (optional) receives f(q). */
int mdp_linear_interp(const float* x, const float* y, int num_samples,
                      const float* q, int64_t count, float* fq, float* efforts, void* stream);

/* ActuatorNet weights.  A network is an (optional) LSTM stack followed by an MLP, packed on the
 * host into MFMA A-fragment order by mdp_net_pack (host pointers in, host blob out) and uploaded
 * by the caller.  Activation ids: 0 none, 1 elu, 2 relu, 3 tanh, 4 softsign, 5 sigmoid.  */
typedef struct {
    int32_t lstm_layers;   /* 0 for ActuatorNetMLP */
    int32_t lstm_hidden;   /* multiple of 4, <= 64 */
    int32_t lstm_input;    /* 2 for the SEA net (actuator_net.py:56) */
    int32_t mlp_layers;    /* >= 1 (decoder / MLP) */
    int32_t mlp_dims[9];   /* dims[0] = input width, dims[i+1] = output width of layer i; last must be 1 */
    int32_t mlp_act[8];    /* activation after layer i (last one normally 0) */
} mdp_net_desc_t;

/* size in floats of the packed blob for `desc` (the fragment tables; for the ANYdrive shape -- hidden 8,
 * <= 2 layers, Linear decoder -- followed by the same fragments laid out four per lane for the
 * register-resident kernel).  The blob is opaque to the caller: upload it 16-byte aligned, pass it back. */
int64_t mdp_net_packed_size(const mdp_net_desc_t* desc);
/* weights: host array of pointers in PyTorch state_dict order:
 * for each LSTM layer l: weight_ih_l (4H,in), weight_hh_l (4H,H), bias_ih_l (4H), bias_hh_l (4H);
 * then for each MLP layer: weight (out,in), bias (out).  All row-major fp32 host memory. */
int mdp_net_pack(const mdp_net_desc_t* desc, const float* const* weights, float* packed_out);

/* ActuatorNetLSTM.compute (actuator_net.py:76-99): rows = N*J; x = [q_des - q, qd];
 * (h, c) are (L, rows, H) updated in place; torque -> computed (N*J); DC-motor clip -> applied. */
int mdp_actuator_lstm_compute(const mdp_net_desc_t* desc, const float* packed_weights,
                              int num_envs, int num_joints,
                              const float* q_des, int ld_q_des,
                              const float* joint_pos, int ld_pos,
                              const float* joint_vel, int ld_vel,
                              float* hidden_state, float* cell_state,
                              const float* effort_limit, const float* velocity_limit, float saturation_effort,
                              float* computed_effort, float* applied_effort, void* stream);
/* ActuatorNetLSTM.reset (actuator_net.py:70-74): zero (h, c)[:, env] for rows flagged in env_mask (NULL = all). */
int mdp_actuator_lstm_reset(int layers, int hidden, int num_envs, int num_joints,
                            float* hidden_state, float* cell_state, const uint8_t* env_mask, void* stream);

/* ActuatorNetMLP.compute (actuator_net.py:147-188).  Histories are (N, Hlen, J); instead of
 * rolling the whole history every sub-step (actuator_net.py:152-156) the kernel shifts in place.
 * input_idx (device int32, n_idx taps), input_order 0 = pos_vel, 1 = vel_pos. */
int mdp_actuator_mlp_compute(const mdp_net_desc_t* desc, const float* packed_weights,
                             int num_envs, int num_joints, int history_len,
                             const float* q_des, int ld_q_des,
                             const float* joint_pos, int ld_pos,
                             const float* joint_vel, int ld_vel,
                             float* pos_err_history, float* vel_history,
                             const int32_t* input_idx, int n_idx, int input_order,
                             float pos_scale, float vel_scale, float torque_scale,
                             const float* effort_limit, const float* velocity_limit, float saturation_effort,
                             float* computed_effort, float* applied_effort, void* stream);

/* ------------------------------------------------------------------------------------------
 * Scene update (state container + sensors)
 * ------------------------------------------------------------------------------------------ */

/* ArticulationData.update -> joint_acc finite difference + previous-velocity side effect
 * (LAB/assets/articulation/articulation_data.py:77-82,446-455).  inv_dt = 1/elapsed. */
int mdp_joint_acc_update(int num_envs, int num_joints, const float* joint_vel, int ld_vel,
                         float* prev_joint_vel, float* joint_acc, float elapsed, void* stream);

/* root_state_w / body_state_w assembly (articulation_data.py:264-279,329-346): pose (count,7) in the
 * physics engine's xyzw convention + velocity (count,6) -> state (count,13) with wxyz quaternions. */
int mdp_state_pack(int64_t count, const float* pose_xyzw, const float* velocity, float* state_out, void* stream);
/* The same for the rows of `num_ids` bodies (device ids) of every env of a (N, num_bodies, .) layout: a term that reads two
 * of twenty bodies (feet_slide) does not need the other eighteen assembled; the other rows are left untouched. */
int mdp_state_pack_rows(int64_t num_envs, int num_bodies, const int32_t* body_ids, int num_ids, const float* pose_xyzw,
                        const float* velocity, float* state_out, void* stream);

/* Derived root quantities (articulation_data.py:412-425,503-518): any output may be NULL.
 * root_state is (N,13) [pos, quat wxyz, lin_vel, ang_vel]; gravity_dir is the unit gravity vector. */
int mdp_root_derive(int num_envs, const float* root_state, int ld_state,
                    float gx, float gy, float gz,
                    float* root_lin_vel_b, float* root_ang_vel_b, float* projected_gravity_b,
                    float* heading_w, void* stream);

/* root_link_state_w COM->link velocity correction (articulation_data.py:288-302), works for
 * bodies too: count = N or N*B poses (ld 13 in / out), com_pos_b (count,3). */
int mdp_link_state_from_com(int64_t count, const float* state_in, const float* com_pos_b, float* state_out,
                            void* stream);

/* SensorBase.update + ContactSensor._update_buffers_impl
 * (LAB/sensors/sensor_base.py:197-205,284-293; contact_sensor/contact_sensor.py:314-373).
 * timestamp += dt; outdated |= (timestamp - last_update + 1e-6 >= update_period); then for
 * outdated envs: net_forces_w <- forces, history shift, air/contact-time state machine,
 * last_update = timestamp, outdated = 0.  history may be NULL when history_length == 0
 * (then net_forces_w itself is the (N,1,B,3) history view).  timers may be NULL
 * (track_air_time False). */
int mdp_contact_sensor_update(int num_envs, int num_bodies, int history_len,
                              const float* net_contact_forces, /* (N,B,3) from the physics view */
                              float* net_forces_w, float* net_forces_w_history,
                              float* last_air_time, float* current_air_time,
                              float* last_contact_time, float* current_contact_time,
                              float* timestamp, float* timestamp_last_update, uint8_t* is_outdated,
                              float dt, float update_period, float force_threshold, int eager,
                              void* stream);
/* ContactSensor.reset / SensorBase.reset for rows flagged in env_mask (contact_sensor.py:136-155). */
int mdp_contact_sensor_reset(int num_envs, int num_bodies, int history_len,
                             float* net_forces_w, float* net_forces_w_history,
                             float* last_air_time, float* current_air_time,
                             float* last_contact_time, float* current_contact_time,
                             float* timestamp, float* timestamp_last_update, uint8_t* is_outdated,
                             const uint8_t* env_mask, void* stream);
/* plain SensorBase.update for lazily evaluated sensors (RayCaster): timestamps + outdated flag only. */
int mdp_sensor_tick(int num_envs, float* timestamp, const float* timestamp_last_update, uint8_t* is_outdated,
                    float dt, float update_period, void* stream);

/* InteractiveScene.update (LAB/scene/interactive_scene.py:496-513) as ONE launch: everything the scene does
 * between a physics step and the next reader -- ArticulationData.update (articulation_data.py:77-82: root_state_w
 * assembly :264-279, joint state refresh :428-444, joint_acc finite difference :446-455), ContactSensor.update
 * (sensor_base.py:197-205 + contact_sensor.py:314-373) and the timestamp / outdated bookkeeping of lazily evaluated
 * sensors such as the RayCaster (sensor_base.py:197-205).  Every group is optional (NULL pointers skip it); the
 * arithmetic of each group is exactly that of mdp_state_pack / mdp_joint_acc_update / mdp_contact_sensor_update /
 * mdp_sensor_tick.  The *_src pointers are the physics view's buffers of the CURRENT frame (borrowed for the call:
 * a replayed ring of frames is read in place, no staging copy); joint_pos / joint_vel are the state container's own
 * buffers that later kernels (actuators, manager programs) bind to. */
typedef struct {
    int32_t num_envs, num_joints, num_contact_bodies, history_len;
    const float* root_transforms;  /* (N,7) pos + quat xyzw */
    const float* root_velocities;  /* (N,6) */
    float* root_state_w;           /* (N,13) pos, quat wxyz, lin vel, ang vel */
    const float* joint_pos_src;    /* (N,J) row stride ld_joint_src */
    const float* joint_vel_src;
    int32_t ld_joint_src;
    float acc_elapsed;             /* time since joint_acc was last evaluated */
    float* joint_pos;              /* (N,J) dense; may be NULL when the container aliases the physics buffer */
    float* joint_vel;
    float* prev_joint_vel;
    float* joint_acc;
    const float* net_contact_forces; /* (N,B,3) */
    float* net_forces_w;
    float* net_forces_w_history;
    float* last_air_time;
    float* current_air_time;
    float* last_contact_time;
    float* current_contact_time;
    float* contact_timestamp;
    float* contact_timestamp_last_update;
    uint8_t* contact_is_outdated;
    float contact_update_period, force_threshold;
    int32_t contact_eager;
    int32_t num_ticks;             /* lazily evaluated sensors (<= 4): timestamps + outdated flags only */
    float* tick_timestamp[4];
    const float* tick_timestamp_last_update[4];
    uint8_t* tick_is_outdated[4];
    float tick_update_period[4];
} mdp_scene_update_t;
int mdp_scene_update(const mdp_scene_update_t* desc, float dt, void* stream);

/* scene.update(k) and write_data_to_sim(k+1) have no physics step between them (manager_based_rl_env.py:184-195:
 * the loop body is apply_action, write_data_to_sim, sim.step, scene.update), so the actuator launch of sub-step k+1
 * can carry the scene update of sub-step k along: extra blocks of the SAME launch run mdp_scene_update(update, dt)
 * while the actuator blocks run mdp_actuator_{pd,lstm}_compute (articulation.py:1399-1436).  joint_pos / joint_vel
 * must then be the physics buffers of the current frame (update->joint_pos_src / joint_vel_src), not the state
 * container's copies, which the update blocks are writing.  update == NULL is the plain actuator call. */
int mdp_actuator_pd_compute_after_update(const mdp_scene_update_t* update, float dt, int kind, int num_envs,
                                         int num_joints, const float* q_des, int ld_q_des,
                                         const float* qd_des, int ld_qd_des, const float* tau_ff, int ld_tau_ff,
                                         const float* joint_pos, int ld_pos, const float* joint_vel, int ld_vel,
                                         const float* stiffness, const float* damping,
                                         const float* effort_limit, const float* velocity_limit,
                                         float saturation_effort, float* computed_effort, float* applied_effort,
                                         void* stream);
int mdp_actuator_lstm_compute_after_update(const mdp_scene_update_t* update, float dt,
                                           const mdp_net_desc_t* desc, const float* packed_weights,
                                           int num_envs, int num_joints, const float* q_des, int ld_q_des,
                                           const float* joint_pos, int ld_pos, const float* joint_vel, int ld_vel,
                                           float* hidden_state, float* cell_state,
                                           const float* effort_limit, const float* velocity_limit,
                                           float saturation_effort, float* computed_effort, float* applied_effort,
                                           void* stream);

/* ------------------------------------------------------------------------------------------
 * Ray casting (LAB/utils/warp/ops.py:24-145, kernels.py:13-75; LAB/sensors/ray_caster/ray_caster.py:220-260)
 * ------------------------------------------------------------------------------------------ */
typedef struct mdp_mesh mdp_mesh_t;

/* convert_to_warp_mesh (ops.py:130-145): HOST points (V,3) fp32 and indices (M,3) int32; builds the
 * acceleration structures (uniform xy cell grid + LBVH) on the host and uploads them to `device`.
 * accel: 0 = auto, 1 = grid, 2 = bvh. */
int mdp_mesh_create(const float* points, int num_points, const int32_t* indices, int num_tris,
                    int device, int accel, mdp_mesh_t** out);
int mdp_mesh_destroy(mdp_mesh_t* mesh);
/* introspection used by tests / DESIGN.md numbers */
int mdp_mesh_info(const mdp_mesh_t* mesh, int64_t* num_tris, int64_t* num_cells, int64_t* num_cell_refs,
                  int64_t* num_bvh_nodes, int64_t* device_bytes, int* accel);

/* raycast_mesh (ops.py:24-127): closest hit of each ray; outputs pre-filled by the kernel with
 * +inf / -1 on a miss (ops.py:70,78,85,92).  dist/normal/face_id may be NULL.  accel: 0 = mesh default. */
int mdp_raycast(const mdp_mesh_t* mesh, int64_t num_rays,
                const float* ray_starts, const float* ray_dirs, float max_dist,
                float* ray_hits, float* ray_dist, float* ray_normal, int32_t* ray_face_id,
                int accel, void* stream);

/* RayCaster._update_buffers_impl fused with the lazy-update bookkeeping
 * (ray_caster.py:220-260, sensor_base.py:284-293): for every env whose is_outdated flag is set
 * (or all when is_outdated == NULL): pos_w = root_pos + drift, quat_w = root quat, rays from the
 * (R,3) local pattern (yaw-only or full attach), cast, write ray_hits_w (N,R,3); then
 * last_update = timestamp, outdated = 0.  root_pose: (N, >=7) [pos, quat wxyz] with row stride ld.
 * pattern_vertical_down != 0 is the caller's promise that every local direction is (0, 0, -c), c > 0 (the height-scan
 * grid pattern, patterns.py:42-56): with yaw-only attachment the rays are then vertical in the world and a
 * specialised kernel (same arithmetic, half the registers, one grid cell per ray) is used.
 * pose_xyzw != 0: root_pose rows are the physics engine's (N,7) [pos, quat xyzw] transforms read in place
 * (ray_caster.py:226-232) instead of (w,x,y,z) root-state rows.  tick_dt > 0: the kernel first performs this sensor's
 * SensorBase.update(tick_dt) bookkeeping (timestamp += dt, outdated when update_period elapsed; sensor_base.py:197-205),
 * so the cast of the last physics sub-step does not depend on the launch that updates the rest of the scene. */
struct mdp_height_scan_obs;
int mdp_raycaster_update(const mdp_mesh_t* mesh, int num_envs, int num_rays,
                         const float* root_pose, int ld_pose, const float* drift,
                         const float* ray_starts_local, const float* ray_dirs_local,
                         int attach_yaw_only, int pattern_vertical_down, float max_dist,
                         float* pos_w, float* quat_w, float* ray_hits_w,
                         float* timestamp, float* timestamp_last_update, uint8_t* is_outdated,
                         const struct mdp_height_scan_obs* scan, int pose_xyzw, float tick_dt, float update_period,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Manager term programs (LAB/managers/{observation,reward,termination}_manager.py,
 * LAB/envs/mdp/{observations,rewards,terminations}.py, TASKS/.../velocity/mdp/ *.py)
 * ------------------------------------------------------------------------------------------ */

/* One lowered term.  Pointer slots and scalar slots are interpreted per opcode (see the tables in
 * isaaclab_amd/csrc/programs.hip and isaaclab_amd/_terms.py).  The struct is plain data and is
 * copied to the device by mdp_program_create. */
typedef struct {
    int32_t op;
    int32_t out_offset;   /* obs: first output column; reward/termination: term slot */
    int32_t dim;          /* obs: number of columns */
    int32_t n_idx;        /* -1 = all (0..count-1), else number of entries at idx_offset in the index pool */
    int32_t idx_offset;
    int32_t n_idx2;
    int32_t idx2_offset;
    int32_t i0, i1, i2, i3;
    int64_t l0;
    float f0, f1, f2, f3, f4, f5;
    const void* p0;
    const void* p1;
    const void* p2;
    const void* p3;
    const void* p4;
    const void* p5;
    /* observation post-processing (observation_manager.py:313-318) */
    int32_t noise_kind;   /* 0 none, 1 uniform, 2 gaussian, 3 constant */
    int32_t noise_op;     /* 0 add, 1 scale, 2 abs  (noise_model.py:35-42) */
    float n0, n1;         /* uniform: n_min, n_max; gaussian: mean, std; constant: bias */
    const float* noise_u; /* optional injected random numbers (N, dim) used instead of the generator */
    int32_t has_clip;
    float clip_lo, clip_hi;
    int32_t has_scale;    /* 1 scalar `scale`, 2 per-column scale_vec */
    float scale;
    const float* scale_vec;
    /* reward: weight (reward_manager.py:148) ; termination: time_out flag in i3 */
    float weight;
} mdp_term_t;

/* The observation's consumer folded into the kernels that produce it: next to the raw value v of
 * observation column c the producing thread also stores clamp((v - mean32[c]) / denom[c], -clip, clip) -- the forward
 * map of mdp_scaler_apply, bit for bit -- into out[env * ld_out + c], e.g. the current slot of the RL library's
 * rollout storage.  mean32 / denom: the tables mdp_scaler_prepare / mdp_scaler_update maintain, one entry per
 * observation column.  out == NULL: off. */
typedef struct {
    const float* mean32;
    const float* denom;
    float* out;
    int32_t ld_out;
    float clip;
} mdp_obs_normalize_t;

/* height_scan observation (LAB/envs/mdp/observations.py:165-173: sensor.pos_w.z - ray_hits_w.z - offset) and its
 * post-processing (observation_manager.py:313-318: noise -> clip -> scale, described by the post-processing fields of
 * `post`) written by the CAST kernel itself into columns [col_offset, col_offset + R) of the observation tensor: the
 * optional last argument of mdp_raycaster_update.  Envs that are not outdated get their columns from the stored hits.
 * rng: device uint64 {seed, one epoch counter per launch block ...} (1 + num_envs words), advanced by the kernel; only
 * read when `post` asks for generated noise.  The observation program marks these columns MDP_OBS_SKIP. */
typedef struct mdp_height_scan_obs {
    float* obs;
    int32_t ld_obs;
    int32_t col_offset;
    float offset;
    int32_t pad0;
    uint64_t* rng;
    mdp_term_t post;
    mdp_obs_normalize_t normalize;   /* tables indexed by the observation column (col_offset + ray) */
} mdp_height_scan_obs_t;

/* Ray-caster binding for a fused height-scan observation term (op MDP_OBS_HEIGHT_SCAN_CAST): the
 * term's p4 holds a HOST pointer to this struct at mdp_program_create time; the builder copies it
 * (and the mesh view) to the device.  Field meaning as in mdp_raycaster_update. */
typedef struct {
    const mdp_mesh_t* mesh;
    int32_t num_rays;
    int32_t attach_yaw_only;
    float max_dist;
    int32_t ld_pose;
    const float* root_pose;
    const float* drift;
    const float* ray_starts_local;
    const float* ray_dirs_local;
    float* pos_w;
    float* quat_w;
    float* ray_hits_w;
    const float* timestamp;
    float* timestamp_last_update;
    uint8_t* is_outdated;
} mdp_raycaster_desc_t;

typedef struct mdp_program mdp_program_t;

enum { MDP_PROG_OBS = 0, MDP_PROG_REWARD = 1, MDP_PROG_TERMINATION = 2 };

/* host-side builder: copies `terms` (host) and the index pool (host int32) to `device`. */
int mdp_program_create(int kind, int device, int num_envs, const mdp_term_t* terms, int num_terms,
                       const int32_t* index_pool, int index_pool_len, uint64_t seed, mdp_program_t** out);
int mdp_program_destroy(mdp_program_t* prog);

/* ObservationManager.compute_group (observation_manager.py:260-335): obs is (N, ld_obs) with the
 * group's terms concatenated at their out_offset.  Terms with a fused height scan also update the
 * ray caster buffers exactly like mdp_raycaster_update. */
int mdp_obs_compute(mdp_program_t* prog, float* obs, int ld_obs, void* stream);

/* RewardManager.compute (reward_manager.py:128-157): reward_buf (N); episode_sums (T,N) row per
 * term; step_reward (N,T).  Terms with weight == 0 are skipped exactly like the reference. */
int mdp_reward_compute(mdp_program_t* prog, float dt, float* reward_buf, float* episode_sums,
                       float* step_reward, void* stream);

/* TerminationManager.compute (termination_manager.py:151-174): term_dones (T,N) uint8,
 * truncated/terminated (N) uint8, dones (N) = truncated | terminated (optional). */
int mdp_termination_compute(mdp_program_t* prog, uint8_t* term_dones, uint8_t* truncated,
                            uint8_t* terminated, uint8_t* dones, void* stream);

/* The post-physics prologue of ManagerBasedRLEnv.step in ONE launch (manager_based_rl_env.py:199-206):
 * episode_length_buf += 1 (optional, int64 (N)), then TerminationManager.compute, then RewardManager.compute -- a block
 * owns 64 envs, so the reward terms that read the termination buffers (rewards.py:31-38) see this step's flags.
 * Output arguments as in mdp_termination_compute / mdp_reward_compute. */
int mdp_termination_reward_compute(mdp_program_t* termination_prog, mdp_program_t* reward_prog, float dt,
                                   int64_t* episode_length_buf_inc,
                                   uint8_t* term_dones, uint8_t* truncated, uint8_t* terminated, uint8_t* dones,
                                   float* reward_buf, float* episode_sums, float* step_reward, void* stream);

/* Stand-alone evaluation of ONE lowered term (the stock `func(env, **params)` term functions call
 * this when used outside a manager; LAB/managers/manager_base.py:250-267 contract).  `term` is a
 * HOST pointer and travels as a kernel argument; index_pool is a DEVICE int32 array (or NULL).
 * out: obs -> float (N, ld_out); reward -> float (N); termination -> uint8 (N). */
int mdp_term_eval(int kind, const mdp_term_t* term, const int32_t* index_pool, int num_envs, void* out,
                  int ld_out, uint64_t seed, uint64_t epoch, void* stream);

/* generic post-processing for the unfused path: in place on data (N, dim) (observation_manager.py:313-318),
 * described by the post-processing fields of `term` (a HOST pointer); seed/offset key the generator. */
int mdp_obs_postprocess(const mdp_term_t* term, int num_envs, float* data, int ld, uint64_t seed,
                        uint64_t offset, void* stream);

/* Class modifiers (LAB/utils/modifiers/modifier.py:67-259).  count = number of data elements.
 * DigitalFilter.__call__ (:170-191): x_hist (count, nb) / y_hist (count, na) are rolled by one in place,
 * y = x_hist . B - y_hist . A;  A, B device vectors.  Integrator.__call__ (:245-259): integral +=
 * (data + y_prev)/2 * dt; y_prev = data; out = integral.  Reset zeroes rows of flagged envs
 * (per_env elements each). */
int mdp_modifier_digital_filter(int64_t count, const float* data, float* x_hist, int nb, float* y_hist, int na,
                                const float* B, const float* A, float* out, void* stream);
int mdp_modifier_integrator(int64_t count, const float* data, float* integral, float* y_prev, float dt,
                            float* out, void* stream);
int mdp_rows_zero(float* buf, int num_envs, int64_t per_env, const uint8_t* env_mask, void* stream);

/* ------------------------------------------------------------------------------------------
 * UniformVelocityCommand (LAB/envs/mdp/commands/velocity_command.py:111-160,
 * LAB/managers/command_manager.py:151-187)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float resampling_time_lo, resampling_time_hi;
    float lin_vel_x_lo, lin_vel_x_hi, lin_vel_y_lo, lin_vel_y_hi, ang_vel_z_lo, ang_vel_z_hi;
    float heading_lo, heading_hi;
    float rel_standing_envs, rel_heading_envs;
    float heading_control_stiffness;
    int32_t heading_command;
    float max_command_step;   /* resampling_time_range[1] / step_dt (velocity_command.py:113-114) */
} mdp_velocity_command_cfg_t;

/* CommandTerm.compute: metrics, time_left -= dt, resample where <= 0 (command_counter += 1), update.
 * rng_state: device uint64[2] {seed, epoch}; epoch is advanced by the kernel (graph-replay safe).
 * injected_u: optional (N,7) uniforms [time, vx, vy, wz, heading, is_heading, is_standing] used instead
 * of the generator (parity tests).  force_mask: optional (N) uint8 -> resample exactly those envs
 * regardless of time_left (CommandTerm.reset path, command_manager.py:120-149; metrics are then
 * zeroed for them and command_counter set to 1). */
int mdp_velocity_command_compute(const mdp_velocity_command_cfg_t* cfg, int num_envs,
                                 const float* root_state, int ld_state,
                                 float* vel_command_b, float* heading_target,
                                 uint8_t* is_heading_env, uint8_t* is_standing_env,
                                 float* time_left, int64_t* command_counter,
                                 float* metric_error_vel_xy, float* metric_error_vel_yaw,
                                 float dt, uint64_t* rng_state, const float* injected_u,
                                 const uint8_t* force_mask, void* stream);

/* JointAction.process_actions (LAB/envs/mdp/actions/joint_actions.py:130-139) +
 * ActionManager.process_action bookkeeping (LAB/managers/action_manager.py:331-332):
 * prev_action = action; action = in; processed = raw*scale + offset [clip]. */
int mdp_joint_action_process(int num_envs, int action_dim, const float* actions_in, int ld_in,
                             float* action, float* prev_action, float* processed,
                             const float* scale_vec, float scale, const float* offset /* (N,A) or NULL */,
                             const float* clip /* (N,A,2) or NULL */, void* stream);

/* The arguments of mdp_joint_action_process as one record, for the actuator entry points below that process the action
 * of the step inside the first actuator launch of the step. */
typedef struct {
    const float* actions_in;
    int32_t ld_in;
    int32_t pad0;
    float* action;
    float* prev_action;
    float* processed;       /* (N,A), required: it is the joint position target the actuator reads */
    const float* scale_vec;
    float scale;
    int32_t pad1;
    const float* offset;
    const float* clip;
} mdp_joint_action_t;

/* ------------------------------------------------------------------------------------------
 * isaaclab.utils.math kernels (LAB/utils/math.py) -- op ids in enum, generic batched launcher.
 * a/b/c are input rows with widths given per op; out width given per op (see mathops.hip table).
 * ------------------------------------------------------------------------------------------ */
enum {
    MDP_MATH_QUAT_MUL = 0,            /* math.py:463-499 */
    MDP_MATH_QUAT_CONJUGATE = 1,      /* :223-235 */
    MDP_MATH_QUAT_INV = 2,            /* :238-248 */
    MDP_MATH_QUAT_UNIQUE = 3,         /* :447-460 */
    MDP_MATH_QUAT_ROTATE = 4,         /* :582-602 */
    MDP_MATH_QUAT_ROTATE_INVERSE = 5, /* :605-625 */
    MDP_MATH_QUAT_APPLY = 6,          /* :545-564 */
    MDP_MATH_QUAT_APPLY_YAW = 7,      /* :567-579 */
    MDP_MATH_YAW_QUAT = 8,            /* :521-542 */
    MDP_MATH_QUAT_FROM_EULER_XYZ = 9, /* :251-278  a=(n,3) roll,pitch,yaw */
    MDP_MATH_EULER_XYZ_FROM_QUAT = 10,/* :413-444  out=(n,3) */
    MDP_MATH_AXIS_ANGLE_FROM_QUAT = 11,/* :645-674 (f0 = eps) */
    MDP_MATH_QUAT_ERROR_MAGNITUDE = 12,/* :677-689 */
    MDP_MATH_QUAT_BOX_MINUS = 13,     /* :502-518 */
    MDP_MATH_QUAT_FROM_ANGLE_AXIS = 14,/* :628-642 a=(n,1) angle, b=(n,3) axis */
    MDP_MATH_MATRIX_FROM_QUAT = 15,   /* :143-174 out=(n,9) */
    MDP_MATH_QUAT_FROM_MATRIX = 16,   /* :294-349 a=(n,9) */
    MDP_MATH_WRAP_TO_PI = 17,         /* :95-117  width 1 */
    MDP_MATH_NORMALIZE = 18,          /* :81-92   width = i0, f0 = eps */
    MDP_MATH_SCALE_TRANSFORM = 19,    /* :22-40   width 1, a=x b=lower c=upper */
    MDP_MATH_UNSCALE_TRANSFORM = 20,  /* :43-61 */
    MDP_MATH_SATURATE = 21,           /* :64-78 */
    MDP_MATH_COMBINE_FRAME = 22,      /* :749-781 a=t01 b=q01 c=t12 d=q12 out=(n,7) */
    MDP_MATH_SUBTRACT_FRAME = 23,     /* :785-816 out=(n,7) */
    MDP_MATH_COMPUTE_POSE_ERROR = 24, /* :820-869 out=(n,6) axis-angle or (n,7) quat (i0=1) */
    MDP_MATH_APPLY_DELTA_POSE = 25,   /* :872-910 a=pos b=quat c=delta(6) out=(n,7), f0 = eps */
    MDP_MATH_SKEW = 26,               /* :692-720 out=(n,9) */
    MDP_MATH_COPYSIGN = 27,           /* :120-135 f0 = mag */
    MDP_MATH_NUM_OPS = 28
};
/* Broadcast: a row stride of 0 repeats row 0 (e.g. a (dims,) lower/upper bound). */
int mdp_math_op(int op, int64_t n,
                const float* a, int lda, const float* b, int ldb,
                const float* c, int ldc, const float* d, int ldd,
                float* out, int ldo, int i0, float f0, void* stream);

/* Samplers of LAB/utils/math.py:1250-1421 on the device generator (Philox4x32-10 keyed by {seed, epoch, kind, sample}):
 * random_orientation :1250-1267, random_yaw_orientation :1270-1285, sample_triangle :1288-1310, sample_uniform
 * :1313-1331, sample_log_uniform :1334-1362, sample_gaussian :1365-1384, sample_cylinder :1387-1421
 * (default_orientation :1233-1247 is a constant fill).  `a` / `b` = per-sample lower / upper (mean / std) arrays of n
 * floats, or NULL -> a_scalar / b_scalar; cylinder: a_scalar = radius, b_scalar..c_scalar = height range.
 * out: n floats; (n,3) cylinder; (n,4) wxyz for the two orientation samplers.  `state` = device {seed, epoch}: the epoch
 * is advanced on the device in stream order (a captured call keeps drawing fresh numbers).  `inject` (optional) replaces
 * the generator: 2 uniforms per sample (the first is the standard normal for MDP_SAMPLE_GAUSSIAN), 4 standard normals
 * per sample for MDP_SAMPLE_QUAT -- parity with the reference's transformation of a recorded draw. */
enum {
    MDP_SAMPLE_UNIFORM = 0, MDP_SAMPLE_LOG_UNIFORM = 1, MDP_SAMPLE_GAUSSIAN = 2, MDP_SAMPLE_TRIANGLE = 3,
    MDP_SAMPLE_CYLINDER = 4, MDP_SAMPLE_YAW_QUAT = 5, MDP_SAMPLE_QUAT = 6, MDP_SAMPLE_NUM = 7
};
int mdp_sample(int kind, int64_t n, const float* a, const float* b, float a_scalar, float b_scalar, float c_scalar,
               const float* inject, float* out, uint64_t* state, void* stream);

/* ------------------------------------------------------------------------------------------
 * Shadow-hand in-hand manipulation, direct workflow
 * (TASKS/direct/inhand_manipulation/inhand_manipulation_env.py:97-117,179-199,262-433)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t num_envs, num_dofs, num_actuated, num_fingertips, num_bodies;
    float act_moving_average, vel_obs_scale;
    float dist_reward_scale, rot_reward_scale, rot_eps, action_penalty_scale, success_tolerance,
          reach_goal_bonus, fall_dist, fall_penalty, av_factor;
    int32_t max_consecutive_success;
    int64_t max_episode_length;
} mdp_inhand_cfg_t;

/* _reset_target_pose (:248-260) for the envs of a byte mask instead of the id list the reference takes from nonzero()
 * (:173): new goal_rot = randomize_rotation(u0, u1, x_unit, y_unit) (:363-367) with u = 2 * rand - 1 (sample_uniform,
 * math.py:1331), reset_goal_buf cleared.  Draws: `injected_u` (N, ld_injected) -- columns [0, 2) -- or the device
 * generator rng_state = {seed, one epoch word per env} (an env's word advances when it draws: hipGraph replays keep
 * drawing).  The goal markers (:257-258) are visualisation and out of scope. */
typedef struct {
    const uint8_t* env_mask;      /* (N) or NULL = every env; ignored by mdp_inhand_dones_rewards_step (own goal_resets) */
    const float* injected_u;      /* or NULL */
    int32_t ld_injected, pad0;
    uint64_t* rng_state;          /* 1 + N words; required when injected_u is NULL */
    const float* x_unit;          /* (N,3) */
    const float* y_unit;          /* (N,3) */
    float* goal_rot;              /* (N,4) wxyz */
    uint8_t* reset_goal_buf;      /* (N) */
} mdp_inhand_goal_reset_t;

/* _apply_action (inhand_manipulation_env.py:97-117): scale -> moving average -> saturate, in place
 * on cur_targets / prev_targets (N, num_dofs) for the actuated columns (device int32 indices). */
int mdp_inhand_apply_action(const mdp_inhand_cfg_t* cfg, const float* actions, const int32_t* actuated_idx,
                            const float* lower, const float* upper, float* cur_targets, float* prev_targets,
                            void* stream);
/* _get_dones (:179-199): out_of_reach, time_out (uint8); episode_length_buf (int64) reset on success
 * when max_consecutive_success > 0. */
int mdp_inhand_dones(const mdp_inhand_cfg_t* cfg, const float* object_root_state /* (N,13) */,
                     const float* env_origins, const float* in_hand_pos, const float* goal_rot,
                     const float* successes, int64_t* episode_length_buf,
                     uint8_t* out_of_reach, uint8_t* time_out, void* stream);
/* compute_rewards (:377-433).  reset_buf / reset_goal_buf uint8 in; goal_resets uint8 out (may alias
 * reset_goal_buf); successes in/out; partial sums (2 doubles: num_resets, sum(successes*resets)) are
 * accumulated into `partials` (device double[2], zeroed by the kernel's first phase) and
 * consecutive_successes (1) is updated by mdp_inhand_finish_rewards after an optional cross-rank
 * all-reduce of `partials`. */
int mdp_inhand_rewards(const mdp_inhand_cfg_t* cfg, const float* object_root_state, const float* env_origins,
                       const float* in_hand_pos, const float* goal_rot, const float* actions,
                       const uint8_t* reset_buf, const uint8_t* reset_goal_buf, float* successes,
                       float* reward, uint8_t* goal_resets, double* partials, void* stream);
int mdp_inhand_finish_rewards(const mdp_inhand_cfg_t* cfg, const double* partials,
                              float* consecutive_successes, void* stream);
/* compute_full_observations (:298-322): obs (N, 2*dofs + 13 + 11 + 13*tips + actions). */
int mdp_inhand_full_obs(const mdp_inhand_cfg_t* cfg, const float* joint_pos, const float* joint_vel,
                        const float* lower, const float* upper,
                        const float* object_root_state, const float* env_origins,
                        const float* in_hand_pos, const float* goal_rot,
                        const float* body_state_w /* (N,B,13) */, const int32_t* finger_bodies,
                        const float* actions, int action_dim, float* obs, int ld_obs, void* stream);

/* The same five methods with the host statements that surround them in DirectRLEnv.step folded in (direct_rl_env.py:
 * 293-380, inhand_manipulation_env.py:97-199): one launch each instead of a launch plus two to four small torch kernels.
 * Same arithmetic, same outputs as the calls above.
 *  _to        : also writes the actuated columns of the articulation's joint_pos_target (N, ld_target) --
 *               `self.hand.set_joint_position_target(self.cur_targets[:, actuated], joint_ids=actuated)` (:115-117);
 *               actions_copy (N, num_actuated; NULL = none): `self.actions = actions.clone()` of _pre_physics_step
 *               (:94-95) written by the same launch (the first sub-step's call passes the caller's tensor as `actions`).
 *  dones_step : `episode_length_buf += 1` (direct_rl_env.py:343) first, then _get_dones written straight into
 *               reset_terminated / reset_time_outs, and reset_buf = terminated | time_outs (:344-346).
 *               With object_transforms / object_velocities (the object's physics view buffers) the launch first
 *               assembles object_root_state (N,13) from them -- RigidObjectData.root_state_w, the cat of
 *               articulation_data.py:264-279 that mdp_state_pack does -- stores it for the launches behind it and works
 *               from registers; without them object_root_state is an input as in mdp_inhand_dones.
 *  rewards_step / finish_rewards_step : reset_goal_buf updated in place (:172); `partials` must be zero on entry and is
 *               left zeroed by finish_rewards_step (no memset node per step); an all-reduce of `partials` may sit
 *               between the two calls as above.
 *  full_obs_from_view : the fingertip rows are read from the physics view's link buffers ((N,B,7) xyzw transforms,
 *               (N,B,6) velocities: what body_state_w is assembled from, articulation_data.py:323-337) instead of a
 *               packed (N,B,13) body_state_w -- 5 of 26 bodies are used, the pack of all 26 was a third of the step.
 *               With partials / consecutive_successes the launch also does what finish_rewards_step does (the reward
 *               launch of the step, and an all-reduce of `partials` if there is one, are in front of it in the stream). */
int mdp_inhand_apply_action_to(const mdp_inhand_cfg_t* cfg, const float* actions, float* actions_copy, const int32_t* actuated_idx,
                               const float* lower, const float* upper, float* cur_targets, float* prev_targets,
                               float* joint_pos_target, int ld_target, void* stream);
/* apply_action_to with the PREVIOUS sub-step's `self.hand.data.update(dt)` (direct_rl_env.py:330-339, articulation_data.py:
 * 77-82: the mdp_scene_update descriptor) on blocks of the same launch -- the new targets do not depend on it. */
int mdp_inhand_apply_action_to_with_update(const mdp_inhand_cfg_t* cfg, const float* actions, float* actions_copy,
                                           const int32_t* actuated_idx, const float* lower, const float* upper,
                                           float* cur_targets, float* prev_targets, float* joint_pos_target, int ld_target,
                                           const mdp_scene_update_t* update, float dt, void* stream);
/* direct_rl_env.py:343-349 as ONE launch: dones_step followed, per env and in the same thread, by rewards_step (which takes
 * the env's reset flag and the object's state from registers); `update` (or NULL): the last sub-step's
 * `self.hand.data.update(dt)` on blocks of the same launch (neither the dones nor the rewards read the hand).  Same
 * buffers and values as mdp_inhand_dones_step + mdp_inhand_rewards_step; `partials` as for rewards_step. */
int mdp_inhand_dones_rewards_step(const mdp_inhand_cfg_t* cfg, float* object_root_state, const float* env_origins,
                                  const float* in_hand_pos, const float* goal_rot, float* successes,
                                  int64_t* episode_length_buf, uint8_t* reset_terminated, uint8_t* reset_time_outs,
                                  uint8_t* reset_buf, const float* object_transforms /* (N,7) xyzw or NULL */,
                                  const float* object_velocities /* (N,6) or NULL */, const float* actions,
                                  uint8_t* reset_goal_buf, float* reward, double* partials,
                                  const mdp_scene_update_t* update /* or NULL */, float dt,
                                  const mdp_inhand_goal_reset_t* goal_reset /* or NULL: the goal reset of _get_rewards
                                  (:172-175) for the envs whose goal_resets this launch sets */,
                                  void* stream);
int mdp_inhand_dones_step(const mdp_inhand_cfg_t* cfg, float* object_root_state, const float* env_origins,
                          const float* in_hand_pos, const float* goal_rot, const float* successes,
                          int64_t* episode_length_buf, uint8_t* reset_terminated, uint8_t* reset_time_outs,
                          uint8_t* reset_buf, const float* object_transforms /* (N,7) xyzw or NULL */,
                          const float* object_velocities /* (N,6) or NULL */, void* stream);
int mdp_inhand_rewards_step(const mdp_inhand_cfg_t* cfg, const float* object_root_state, const float* env_origins,
                            const float* in_hand_pos, const float* goal_rot, const float* actions,
                            const uint8_t* reset_buf, uint8_t* reset_goal_buf, float* successes, float* reward,
                            double* partials, void* stream);
int mdp_inhand_finish_rewards_step(const mdp_inhand_cfg_t* cfg, double* partials, float* consecutive_successes,
                                   void* stream);
int mdp_inhand_full_obs_from_view(const mdp_inhand_cfg_t* cfg, const float* joint_pos, const float* joint_vel,
                                  const float* lower, const float* upper, const float* object_root_state,
                                  const float* env_origins, const float* in_hand_pos, const float* goal_rot,
                                  const float* link_transforms /* (N,B,7) xyzw */,
                                  const float* link_velocities /* (N,B,6) */, const int32_t* finger_bodies,
                                  const float* actions, int action_dim, float* obs, int ld_obs,
                                  double* partials /* or NULL */, float* consecutive_successes /* or NULL */,
                                  void* stream);

/* _reset_target_pose (:248-260) as a masked launch: mdp_inhand_goal_reset_t above. */
int mdp_inhand_reset_target_pose(const mdp_inhand_cfg_t* cfg, const mdp_inhand_goal_reset_t* goal, void* stream);

/* InHandManipulationEnv._reset_idx (:201-246) + DirectRLEnv._reset_idx (direct_rl_env.py:560-581) for the envs of a byte
 * mask, one launch, nothing synchronises with the host: goal (as above), object root state = default + position noise +
 * env origin, randomize_rotation of two more draws, zero velocity, written like RigidObject.write_root_pose_to_sim /
 * write_root_velocity_to_sim (rigid_object.py:192-213, 268-290: data buffer and the physics view's xyzw transforms /
 * velocities); hand joint state = default + noise scaled into the joint range (:228-236), written like
 * Articulation.write_joint_state_to_sim (articulation.py:472-508: joint_pos, joint_vel, previous velocity, joint_acc = 0
 * and the view's dof buffers) and into prev_targets / cur_targets / hand_dof_targets / joint_pos_target (:238-242);
 * successes = 0, episode_length_buf = 0.  Draw order per env as the reference's torch.rand calls: goal 2, position 3,
 * rotation 2, dof positions D, dof velocities D (injected_u columns in that order, ld_injected >= 7 + 2 D). */
typedef struct {
    mdp_inhand_goal_reset_t goal;                 /* env_mask, draws, goal buffers */
    float reset_position_noise, reset_dof_pos_noise, reset_dof_vel_noise, pad0;
    const float* object_default_root_state;       /* (N,13) */
    const float* env_origins;                     /* (N,3) */
    float* object_root_state_w;                   /* (N,13) */
    float* sim_object_transforms;                 /* (N,7) xyzw */
    float* sim_object_velocities;                 /* (N,6) */
    const float* default_joint_pos; const float* default_joint_vel; const float* lower; const float* upper; /* (N,D) */
    float* prev_targets; float* cur_targets; float* hand_dof_targets;   /* (N,D) */
    float* joint_pos_target; int32_t ld_target, ld_sim;
    float* joint_pos; float* joint_vel; float* prev_joint_vel; float* joint_acc;   /* (N,D) */
    float* sim_dof_positions; float* sim_dof_velocities;                          /* (N, ld_sim) */
    float* successes;                             /* (N) */
    int64_t* episode_length_buf;                  /* (N) */
} mdp_inhand_reset_t;
int mdp_inhand_reset_idx(const mdp_inhand_cfg_t* cfg, const mdp_inhand_reset_t* args, void* stream);

/* The three observation layouts of the env in ONE launch (:119-138): `policy` = compute_full_observations (:298-322, 157
 * columns for the Shadow hand) or compute_reduced_observations (:281-296: fingertip positions, object position,
 * quat_mul(object_rot, conj(goal_rot)), actions = 42), and -- asymmetric_obs -- `critic` = compute_full_state (:324-350:
 * the full layout with force_torque_obs_scale * fingertip force sensors in front of the actions, 187).  Fingertip rows
 * come from body_state_w (N,B,13) or, when that is NULL, from the physics view's link buffers as in
 * mdp_inhand_full_obs_from_view; link_incoming_joint_force (N,B,6) is the view's get_link_incoming_joint_force()
 * (:121-123), read for the critic only.  A kind < 0 skips that output.  partials / consecutive_successes as for
 * mdp_inhand_full_obs_from_view. */
enum { MDP_INHAND_OBS_FULL = 0, MDP_INHAND_OBS_REDUCED = 1, MDP_INHAND_OBS_FULL_STATE = 2 };
typedef struct {
    const float* joint_pos; const float* joint_vel; const float* lower; const float* upper;
    const float* object_root_state; const float* env_origins; const float* in_hand_pos; const float* goal_rot;
    const float* body_state_w;              /* (N,B,13) or NULL */
    const float* link_transforms;           /* (N,B,7) xyzw, used when body_state_w is NULL */
    const float* link_velocities;           /* (N,B,6) */
    const float* link_incoming_joint_force; /* (N,B,6); required for MDP_INHAND_OBS_FULL_STATE */
    const int32_t* finger_bodies;
    const float* actions;
    int32_t action_dim;
    float force_torque_obs_scale;
    int32_t policy_kind, ld_policy;
    float* policy;
    int32_t critic_kind, ld_critic;
    float* critic;
    double* partials;                       /* or NULL */
    float* consecutive_successes;           /* or NULL */
    const mdp_inhand_reset_t* reset;        /* or NULL: mdp_inhand_reset_idx of the envs of reset->goal.env_mask by the block
                                               that owns them, in front of their observations (direct_rl_env.py:355-372:
                                               every buffer both touch is the env's own row) -- no launch of its own */
} mdp_inhand_obs_args_t;
int mdp_inhand_observations(const mdp_inhand_cfg_t* cfg, const mdp_inhand_obs_args_t* args, void* stream);
/* number of columns of a layout for this cfg (0 for an unknown kind) */
int mdp_inhand_obs_dim(const mdp_inhand_cfg_t* cfg, int kind, int action_dim);


/* ------------------------------------------------------------------------------------------
 * Statistics for the cross-GPU reductions: per-rank partials in fp64.
 * ------------------------------------------------------------------------------------------ */
/* obs-normaliser partials of x (N,D): out[0]=N, out[1..D]=sum x, out[1+D..2D]=sum (x-mean_r)^2
 * (Chan-mergeable; my_code/models/my_running_standard_scaler.py:57-80). */
int mdp_stats_moments(const float* x, int num_envs, int dim, int ld, double* out, void* stream);
/* masked sum/count of per-env scalars (episode statistics at reset; reward_manager.py:115-121):
 * out[2*k] = sum_{mask} values[k][env], out[2*k+1] = count, for k rows of a (K,N) matrix. */
int mdp_stats_masked_sums(const float* values, int rows, int num_envs, const uint8_t* mask, double* out,
                          void* stream);
/* The same over rows of another element type -- elem_kind 0: float, 1: uint8 (per-term done flags:
 * `Episode_Termination/<term>` = count_nonzero over the reset envs, termination_manager.py:142-144), 2: int64 (terrain
 * levels: `Curriculum/terrain_levels` = mean over ALL envs, TASKS/.../velocity/mdp/curriculums.py:55, mask NULL). */
int mdp_stats_masked_sums_typed(const void* values, int elem_kind, int rows, int num_envs, const uint8_t* mask,
                                double* out, void* stream);
/* All of a rollout's statistics families in ONE launch: source i is a (rows_i, N) block of element kind elem_kind_i
 * (as above), summed over the mask when `masked` is non-zero and over all envs otherwise; out receives the
 * [sum, count] pairs of the sources' rows back to back, i.e. exactly what one mdp_stats_masked_sums_typed call per
 * source writes (same order of additions: bit-identical) -- the packed vector of EpisodeStats
 * (reward_manager.py:115-121, command_manager.py:136-142, termination_manager.py:142-144, curriculums.py:55). */
#define MDP_STATS_MAX_SOURCES 8
typedef struct {
    const void* values;   /* (rows, N) row-major */
    int32_t elem_kind;    /* 0 float, 1 uint8, 2 int64 */
    int32_t rows;
    int32_t masked;       /* != 0: follows `mask` */
    int32_t _pad;
} mdp_stats_source_t;
int mdp_stats_masked_sums_multi(const mdp_stats_source_t* sources, int num_sources, int num_envs, const uint8_t* mask,
                                double* out, void* stream);

/* Reset of per-env state by MASK instead of by id list, many buffers in ONE launch (the `buf[env_ids] = value` lines of
 * ManagerBasedRLEnv._reset_idx's callees: sensor_base.py:192-195, contact_sensor.py:143-155, action_manager.py reset,
 * reward_manager.py:121, command_manager.py:137-142, actuator_net.py:61-64, manager_based_rl_env.py:391).  With the
 * mask = the step's reset_buf no `nonzero()` host sync is needed and the reset can sit inside a captured hipGraph.
 * Each record describes `count` elements per env of `elem_bytes` (1, 4 or 8) bytes: element k of env e is at
 * ptr + (e * env_stride + k * elem_stride) * elem_bytes -- (N, count) buffers have env_stride = count, elem_stride = 1;
 * term-major (count, N) blocks env_stride = 1, elem_stride = N.  value_bits holds the fill value's bit pattern. */
typedef struct {
    void* ptr;
    int32_t elem_bytes;
    int32_t count;
    int64_t env_stride;
    int64_t elem_stride;
    uint64_t value_bits;
} mdp_fill_t;
#define MDP_MAX_FILLS 48
int mdp_masked_fill(const mdp_fill_t* fills, int num_fills, const uint8_t* env_mask, int num_envs, void* stream);

/* ManagerBasedRLEnv.step lines 168-186 for a joint-position action term that drives every joint of an articulation with
 * one actuator group: action_manager.process_action(action) (mdp_joint_action_process), apply_action() (the processed
 * action IS the joint position target) and the first scene.write_data_to_sim() of the step
 * (mdp_actuator_lstm_compute / mdp_actuator_pd_compute with q_des = action->processed, ld = num_joints) as ONE launch:
 * each (env, joint) row processes its own action element before it evaluates the actuator.  action_dim == num_joints.
 * Outputs are bit-identical to the two separate calls. */
int mdp_actuator_lstm_compute_with_action(const mdp_joint_action_t* action, const mdp_net_desc_t* desc,
                                          const float* packed_weights, int num_envs, int num_joints,
                                          const float* joint_pos, int ld_pos, const float* joint_vel, int ld_vel,
                                          float* hidden_state, float* cell_state, const float* effort_limit,
                                          const float* velocity_limit, float saturation_effort, float* computed_effort,
                                          float* applied_effort, void* stream);
int mdp_actuator_pd_compute_with_action(const mdp_joint_action_t* action, int kind, int num_envs, int num_joints,
                                        const float* qd_des, int ld_qd_des, const float* tau_ff, int ld_tau_ff,
                                        const float* joint_pos, int ld_pos, const float* joint_vel, int ld_vel,
                                        const float* stiffness, const float* damping, const float* effort_limit,
                                        const float* velocity_limit, float saturation_effort, float* computed_effort,
                                        float* applied_effort, void* stream);

/* Everything ManagerBasedRLEnv.step does after its last physics sub-step, for a height-scan locomotion task, as ONE
 * launch (manager_based_rl_env.py:195-238 without the reset branch):
 *   scene.update(physics_dt)                                  -> mdp_scene_update          (update, may be NULL)
 *   episode_length_buf += 1; terminations; rewards            -> mdp_termination_reward_compute
 *   command_manager.compute(step_dt)                          -> mdp_velocity_command_compute (command, may be NULL)
 *   observation_manager.compute() incl. the lazy ray cast     -> mdp_obs_compute + mdp_raycaster_update (+ scan)
 * Every stage is independent per env, so a block of the launch owns a few envs (24) and walks them through the stages in the
 * reference order (block barriers in between), while other blocks of the same launch cast the height-scan rays (the cast
 * only needs the physics transforms: pose_xyzw + tick_dt as in mdp_raycaster_update).  Outputs are bit-identical to the
 * separate calls (generated noise and command samples use their own device-side epoch counters).  Restrictions (MDP_ERR_UNSUPPORTED otherwise; callers fall back to the separate calls): the cast must
 * qualify for the vertical-ray kernel (yaw-only attachment, pattern_vertical_down, grid-accelerated mesh), a single
 * command term, no lazily evaluated sensor read by a termination / reward term.  The structs only bundle the arguments
 * of the calls named above. */
typedef struct {
    mdp_program_t* termination;
    mdp_program_t* reward;
    float dt;
    int32_t pad0;
    int64_t* episode_length_buf_inc;
    uint8_t* term_dones;
    uint8_t* truncated;
    uint8_t* terminated;
    uint8_t* dones;
    float* reward_buf;
    float* episode_sums;
    float* step_reward;
} mdp_termination_reward_args_t;

typedef struct {
    mdp_velocity_command_cfg_t cfg;
    int32_t ld_state;
    const float* root_state;
    float* vel_command_b;
    float* heading_target;
    uint8_t* is_heading_env;
    uint8_t* is_standing_env;
    float* time_left;
    int64_t* command_counter;
    float* metric_error_vel_xy;
    float* metric_error_vel_yaw;
    float dt;
    int32_t pad0;
    uint64_t* rng_state;      /* {seed, -, -, -, one epoch counter per manager block ...}: 4 + N words */
    const float* injected_u;
} mdp_velocity_command_args_t;

typedef struct {
    mdp_program_t* program;
    float* obs;
    int32_t ld_obs;
    int32_t pad0;
    mdp_obs_normalize_t normalize;   /* mdp_post_physics_step only (the cast member carries its own copy in `scan`) */
} mdp_obs_args_t;

typedef struct {
    const mdp_mesh_t* mesh;
    int32_t num_rays, ld_pose, pose_xyzw, attach_yaw_only, pattern_vertical_down, pad0;
    const float* root_pose;
    const float* drift;
    const float* ray_starts_local;
    const float* ray_dirs_local;
    float max_dist, tick_dt, update_period, pad1;
    float* pos_w;
    float* quat_w;
    float* ray_hits_w;
    float* timestamp;
    float* timestamp_last_update;
    uint8_t* is_outdated;
    const struct mdp_height_scan_obs* scan;
    /* optional workspace of 8 floats per env owned by this sensor (mdp_post_physics_step only; NULL = none).  With it,
     * launches of many envs compute the per-env part of the update -- SensorBase.update bookkeeping, sensor pose, yaw
     * quaternion (ray_caster.py:220-240, sensor_base.py:197-205) -- in a small pass of one thread per env in front of
     * the post-physics kernel, whose cast blocks then start with their rays instead of waiting on one lane's chain of
     * cold loads; same values, same buffers written. */
    float* pose_ws;
} mdp_raycaster_update_args_t;

int mdp_post_physics_step(int num_envs, const mdp_scene_update_t* update, float physics_dt,
                          const mdp_termination_reward_args_t* termination_reward,
                          const mdp_velocity_command_args_t* command, const mdp_obs_args_t* observation,
                          const mdp_raycaster_update_args_t* cast, void* stream);

/* RunningStandardScaler._compute (my_code/models/my_running_standard_scaler.py:82-119), SURVEY.md 8f N2: the consumer
 * of the observation.  mdp_scaler_prepare casts the fp64 running statistics once per update (mean32 = float(mean),
 * std32 = sqrt(float(var)), denom = std32 + epsilon); mdp_scaler_apply is one pass over (rows, dim):
 *   forward: out = clamp((x - mean32) / denom, -clip, clip)      inverse: out = std32 * clamp(x, -clip, clip) + mean32
 * `out` has its own row stride, so the result can land directly in a slot of the RL library's (T, N, O) rollout
 * storage (or in place, out == x).  The statistics themselves come from mdp_stats_moments (+ the cross-rank merge). */
/* Training-mode update (my_running_standard_scaler.py:57-80, 94-105) in two launches: mdp_stats_moments_chunked writes
 * `num_chunks` Chan-mergeable partial rows [n, sum x (dim), M2 (dim)] of the (rows, dim) batch (coalesced reads, fp64
 * accumulation); mdp_scaler_update merges `num_partials` such rows -- this rank's, or every rank's after one all-gather
 * -- into the batch mean and torch.var (Bessel-corrected), applies the parallel-variance update to the fp64 running
 * statistics (current_count: one double) and refreshes the fp32 casts. */
int mdp_stats_moments_chunked(const float* x, int64_t rows, int dim, int64_t ld, int num_chunks, double* partials,
                              void* stream);
int mdp_scaler_update(int num_partials, int dim, const double* partials, double* running_mean,
                      double* running_variance, double* current_count, float epsilon, float* mean32, float* std32,
                      float* denom, void* stream);
int mdp_scaler_prepare(int dim, const double* running_mean, const double* running_variance, float epsilon,
                       float* mean32, float* std32, float* denom, void* stream);
int mdp_scaler_apply(int64_t rows, int dim, const float* x, int64_t ld, const float* mean32, const float* std32,
                     const float* denom, float clip_threshold, int inverse, float* out, int64_t ld_out, void* stream);

/* The statistics the managers' reset(env_ids) return for extras["log"] (reward_manager.py:115-121,
 * command_manager.py:131-136, termination_manager.py:149-150) for the envs flagged in `dones`, in one launch and
 * without an id list: log[0] = number of flagged envs, log[1..Tr] = mean(episode sum) / max_episode_length_s per
 * reward term (episode_sums is (Tr, N), term-major), then the means of two per-env command metrics (either may be
 * NULL -> 0), then one count per termination term (term_dones is (Tt, N)).  All 0 when no env is flagged.  Same
 * layout as mdp_post_reset_args_t.log. */
int mdp_reset_log(int num_envs, const uint8_t* dones, const float* episode_sums, int num_reward_terms,
                  const float* metric_a, const float* metric_b, const uint8_t* term_dones, int num_termination_terms,
                  float max_episode_length_s, float* log, void* stream);

/* The same launch with the reset branch of the step in it (manager_based_rl_env.py:214-229, 346-391 without events and
 * curriculum): between its rewards and its command update a manager block resets the envs whose `dones` flag the
 * launch has just written -- the buffers of `fills` (DEVICE array of mdp_fill_t records, as for mdp_masked_fill) and
 * the command term (CommandTerm.reset: metrics, counter, resample) -- and adds the episode statistics of those envs to
 * `log_sums`; the last manager block to finish turns the sums into the values the managers' reset() log
 * (reward_manager.py:115-121, command_manager.py:131-136, termination_manager.py:149-150) in `log`: log[0] = number of
 * reset envs, then per reward term mean(episode sum) / max_episode_length_s, the means of the two command metrics, and
 * one count per termination term (all 0 when no env was done).  Blocks keep their partial sums in slots of their own
 * and the last one adds them in a fixed order (no floating-point atomics: the log is reproducible); word 0 of
 * `log_sums` is the block ticket, which the caller zeroes once, when it allocates the workspace.  The ray caster's own
 * reset (timestamps zeroed, flagged outdated, cast again by the observation) cannot ride in this launch -- its cast
 * blocks run concurrently with the blocks that find out which envs are done -- so the caller follows it with
 * mdp_raycaster_update_flagged(dones). */
typedef struct {
    const mdp_fill_t* fills;   /* device pointer */
    int32_t num_fills;
    float max_episode_length_s;
    double* log_sums;          /* device workspace, 1 + (1 + Tr + 2 + Tt) * ceil(num_envs / 8) doubles */
    float* log;                /* device, 1 + Tr + 2 + Tt floats (out) */
} mdp_post_reset_args_t;

int mdp_post_physics_step_with_reset(int num_envs, const mdp_scene_update_t* update, float physics_dt,
                                     const mdp_termination_reward_args_t* termination_reward,
                                     const mdp_velocity_command_args_t* command, const mdp_obs_args_t* observation,
                                     const mdp_raycaster_update_args_t* cast, const mdp_post_reset_args_t* reset,
                                     void* stream);

/* Allocation-free variant of mdp_post_physics_step_with_reset.  It has NO counterpart in the reference: it exists for
 * callers that cannot allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  Same arguments, same kernels and bit-identical outputs, except that reset->log_sums MUST be NULL: the
 * block-statistics workspace is a file-scope device array of the library (MDP_POST_NOALLOC_SCRATCH_DOUBLES doubles =
 * 1 MiB + 8 B, held by every GPU the library runs on until the process ends).  The call allocates nothing and frees
 * nothing (no hipMalloc / hipFree, no torch); its first call on a device looks the array's address up
 * (hipGetSymbolAddress), so make that first call outside a capture (a warm-up step, as captures need anyway).
 * Limit: 1 + (1 + Tr + 2 + Tt) * ceil(num_envs / 8) <= MDP_POST_NOALLOC_SCRATCH_DOUBLES (e.g. 65,536 envs with 16
 * statistics rows), else MDP_ERR_INVALID with a message naming both numbers.  Every call on one GPU shares that GPU's
 * array: the ticket word is cleared on `stream` before the launch and the slots are rewritten by the launch itself,
 * so calls in stream order are safe, but two calls that OVERLAP on one GPU (different streams / processes' threads)
 * race on it -- use the workspace variant there. */
#define MDP_POST_NOALLOC_SCRATCH_DOUBLES 131073
int mdp_post_physics_step_with_reset_noalloc(int num_envs, const mdp_scene_update_t* update, float physics_dt,
                                             const mdp_termination_reward_args_t* termination_reward,
                                             const mdp_velocity_command_args_t* command,
                                             const mdp_obs_args_t* observation,
                                             const mdp_raycaster_update_args_t* cast,
                                             const mdp_post_reset_args_t* reset, void* stream);

/* RayCaster.reset(env_ids) + the lazy re-evaluation the next data access triggers (sensor_base.py:182-195, 284-293;
 * ray_caster.py:106-113, 220-260), for the envs flagged in `reset_mask` (device, num_envs bytes, may be null): their
 * timestamps go to 0 and they count as outdated; then every outdated env -- and only those -- is cast again and, with
 * cast->scan, gets its height_scan observation columns rewritten.  No update(dt) bookkeeping (cast->tick_dt is
 * ignored).  Same coverage as the cast member of mdp_post_physics_step (vertical rays, yaw-only, grid mesh), else
 * MDP_ERR_UNSUPPORTED. */
int mdp_raycaster_update_flagged(int num_envs, const mdp_raycaster_update_args_t* cast, const uint8_t* reset_mask,
                                 void* stream);

/* The actuator step the reference runs right after a reset -- `self.scene.write_data_to_sim()` inside
 * `if len(reset_env_ids) > 0:` (manager_based_rl_env.py:214-222): one more actuator evaluation for EVERY env (it advances
 * the LSTM state of all of them) in the steps where at least one env was reset.  The condition lives on the device
 * here: the launch does nothing unless *gate > 0 (`gate` = device float, e.g. the number of reset envs the step's
 * statistics hold).  Otherwise identical to mdp_actuator_lstm_compute / mdp_actuator_pd_compute. */
int mdp_actuator_lstm_compute_gated(const float* gate, const mdp_net_desc_t* desc, const float* packed_weights,
                                    int num_envs, int num_joints, const float* q_des, int ld_q_des,
                                    const float* joint_pos, int ld_pos, const float* joint_vel, int ld_vel,
                                    float* hidden_state, float* cell_state, const float* effort_limit,
                                    const float* velocity_limit, float saturation_effort, float* computed_effort,
                                    float* applied_effort, void* stream);
/* mdp_raycaster_update_flagged(reset_mask) and mdp_actuator_lstm_compute_gated(gate) -- the two launches that follow a
 * post-physics launch with masked resets (sensor_base.py:182-195 + ray_caster.py:220-260 for the reset envs;
 * manager_based_rl_env.py:214-222) -- as ONE launch: recast blocks in front, actuator blocks behind.  The actuator
 * blocks do nothing unless *gate > 0; the recast blocks follow reset_mask / is_outdated exactly like
 * mdp_raycaster_update_flagged (no env reset and none outdated: nothing is cast).  Covers the register-resident LSTM
 * (hidden 8, <= 2 layers, linear decoder) and a cast that mdp_raycaster_update_flagged accepts, else
 * MDP_ERR_UNSUPPORTED -- callers then issue the two launches. */
int mdp_actuator_lstm_compute_gated_with_recast(const float* gate, const mdp_net_desc_t* desc, const float* packed_weights,
                                                int num_envs, int num_joints, const float* q_des, int ld_q_des,
                                                const float* joint_pos, int ld_pos, const float* joint_vel, int ld_vel,
                                                float* hidden_state, float* cell_state, const float* effort_limit,
                                                const float* velocity_limit, float saturation_effort,
                                                float* computed_effort, float* applied_effort,
                                                const mdp_raycaster_update_args_t* cast, const uint8_t* reset_mask,
                                                void* stream);
int mdp_actuator_pd_compute_gated(const float* gate, int kind, int num_envs, int num_joints, const float* q_des,
                                  int ld_q_des, const float* qd_des, int ld_qd_des, const float* tau_ff, int ld_tau_ff,
                                  const float* joint_pos, int ld_pos, const float* joint_vel, int ld_vel,
                                  const float* stiffness, const float* damping, const float* effort_limit,
                                  const float* velocity_limit, float saturation_effort, float* computed_effort,
                                  float* applied_effort, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Event terms of the "reset" and "interval" modes.  The reference calls them with an id list taken
 * from nonzero() on the host (manager_based_rl_env.py:214, event_manager.py:218); here the selection is `env_mask`
 * (device, num_envs bytes, NULL = every env, i.e. env_ids=None) and nothing synchronises.  Each call writes what the
 * reference's asset.write_*_to_sim writes: the data buffers of the asset AND the physics buffers (set_root_transforms
 * / set_root_velocities / set_dof_positions / set_dof_velocities with indices).
 * Random numbers: `injected_u` (device, uniforms in [0,1), row e = the numbers env e consumes, in the order the
 * reference draws them) or `rng_state` (device, 1 + num_envs words: seed, then one epoch per env, advanced when that
 * env draws).  Ranges travel as lo / scale with sample = u * scale + lo (sample_uniform, math.py:1331): scale is
 * fp32(hi) - fp32(lo) for the dict ranges (tensors in the reference) and fp32(hi - lo) for the tuple ranges (Python
 * floats in the reference).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {                       /* reset_root_state_uniform, events.py:638-680 */
    const uint8_t* env_mask;
    const float* default_root_state;   /* (N,13) */
    const float* env_origins;          /* (N,3) */
    float pose_lo[6], pose_scale[6];   /* x, y, z, roll, pitch, yaw */
    float vel_lo[6], vel_scale[6];
    const float* injected_u;           /* (N,12): 6 pose then 6 velocity uniforms */
    uint64_t* rng_state;
    float* root_state_w;               /* (N,13) data buffer (articulation.py:332, 417) */
    float* sim_root_transforms;        /* (N,7) x, y, z, qx, qy, qz, qw (articulation.py:334-341) */
    float* sim_root_velocities;        /* (N,6) (articulation.py:420) */
    float* body_acc_w;                 /* (N,B,6), zeroed for the envs (articulation.py:418), may be NULL */
    int32_t num_bodies;
    int32_t pad0;
} mdp_event_root_state_t;
int mdp_event_reset_root_state_uniform(int num_envs, const mdp_event_root_state_t* args, void* stream);

typedef struct {                       /* reset_joints_by_scale / _by_offset, events.py:802-866 */
    const uint8_t* env_mask;
    int32_t num_joints;
    int32_t mode;                      /* 0 = by scale, 1 = by offset */
    const float* default_joint_pos;    /* (N,J) */
    const float* default_joint_vel;    /* (N,J) */
    float pos_lo, pos_scale, vel_lo, vel_scale;
    const float* soft_joint_pos_limits;/* (N,J,2) */
    const float* soft_joint_vel_limits;/* (N,J) */
    const float* injected_u;           /* (N,2J): J position then J velocity uniforms */
    uint64_t* rng_state;
    float* joint_pos;                  /* (N,J) data buffers (articulation.py:498-501) */
    float* joint_vel;
    float* prev_joint_vel;
    float* joint_acc;
    float* sim_dof_positions;          /* (N, ld_sim) physics buffers (articulation.py:507-508) */
    float* sim_dof_velocities;
    int32_t ld_sim;
    int32_t pad0;
} mdp_event_joints_t;
int mdp_event_reset_joints(int num_envs, const mdp_event_joints_t* args, void* stream);

typedef struct {                       /* push_by_setting_velocity, events.py:610-635 */
    const uint8_t* env_mask;
    float vel_lo[6], vel_scale[6];
    const float* injected_u;           /* (N,6) */
    uint64_t* rng_state;
    float* root_state_w;               /* (N,13): columns 7..12 are read and written */
    float* sim_root_velocities;        /* (N,6) */
    float* body_acc_w;                 /* (N,B,6) or NULL */
    int32_t num_bodies;
    int32_t pad0;
} mdp_event_push_t;
int mdp_event_push_by_setting_velocity(int num_envs, const mdp_event_push_t* args, void* stream);

/* EventManager.apply(mode="reset") bookkeeping of the terms with min_step_count_between_reset == 0 (event_manager.py:
 * 235-240: `last_triggered_step_id[env_ids] = global_env_step_count; last_triggered_once[env_ids] = True`) for the envs of
 * env_mask (NULL = all), every such term in ONE launch instead of two index-put kernels per term. */
#define MDP_EVENT_MARK_MAX_TERMS 8
typedef struct {
    int32_t num_terms;
    int32_t step_count;
    int32_t* last_triggered_step[MDP_EVENT_MARK_MAX_TERMS];   /* (N) int32 each */
    uint8_t* triggered_once[MDP_EVENT_MARK_MAX_TERMS];        /* (N) bool each */
} mdp_event_reset_mark_t;
int mdp_event_reset_mark(int num_envs, const uint8_t* env_mask, const mdp_event_reset_mark_t* args, void* stream);

/* EventManager.apply(mode="interval") for one term (event_manager.py:199-225): time_left -= dt; where it drops below
 * 1e-6 the timer is resampled (u * scale + lo) and trigger_mask = 1, else 0.  is_global_time: `time_left` is ONE
 * word shared by all envs (and injected_u / rng_state hold one uniform / one epoch). */
int mdp_event_interval_tick(int num_envs, float* time_left, float dt, float lo, float scale, int is_global_time,
                            const float* injected_u, uint64_t* rng_state, uint8_t* trigger_mask, void* stream);

/* terrain_levels_vel (TASKS/manager_based/locomotion/velocity/mdp/curriculums.py:26-55) + TerrainImporter.
 * update_env_origins (terrain_importer.py:307-324) for the envs of `env_mask`: the distance walked from the env origin
 * decides whether the env moves one terrain level up or down, envs beyond the last level get a random one, and the env
 * origin becomes terrain_origins[level, type].  terrain_levels / env_origins are updated in place; the term's return
 * value (mean level over ALL envs, curriculums.py:55) is a plain reduction the caller does, or -- with mean_level and its
 * two zero-initialised workspace words mean_ws -- written by the launch itself (integer block sums, last block writes
 * sum * (1 / N) and clears the words).  injected_levels: the randint draws (tests). */
typedef struct {
    const uint8_t* env_mask;
    const float* root_state_w;         /* (N, ld_state), columns 0..1 read */
    int32_t ld_state;
    int32_t max_terrain_level;         /* number of terrain rows */
    const float* command;              /* (N,3) base_velocity command */
    float half_tile_size;              /* terrain_generator.size[0] / 2 */
    float max_episode_length_s;
    int64_t* terrain_levels;           /* (N) */
    const int64_t* terrain_types;      /* (N) */
    const float* terrain_origins;      /* (rows, num_cols, 3) */
    int32_t num_cols;
    int32_t pad0;
    float* env_origins;                /* (N,3) */
    const int64_t* injected_levels;    /* (N) or NULL */
    uint64_t* rng_state;               /* 1 + N words (see the event terms) */
    float* mean_level;                 /* (1) or NULL */
    unsigned long long* mean_ws;       /* 2 words, zero when first handed over; NULL iff mean_level is NULL */
} mdp_curriculum_terrain_t;
int mdp_curriculum_terrain_levels_vel(int num_envs, const mdp_curriculum_terrain_t* args, void* stream);

/* mdp_post_physics_step_with_reset with the task's terrain curriculum and event terms INSIDE the launch
 * (manager_based_rl_env.py:214-235, 346-391 complete; event_manager.py:199-240): a manager block runs, for the envs its
 * termination phase has just flagged, curriculum -> episode statistics -> buffer fills -> reset event terms (root state,
 * then joints; EventManager's reset bookkeeping) -> CommandTerm.reset, then for all its envs the command update and the
 * interval term (per-env timer tick; push_by_setting_velocity where it fired), then the observations -- the order of the
 * reference, with the arithmetic of the stand-alone mdp_event_* / mdp_curriculum_* launches (same device routines, same
 * generator words: one epoch per env and term).  Members that are NULL are skipped; the env_mask members of the term
 * structs are ignored (the launch's own `dones` / the timers decide), and so are curriculum->mean_level / mean_ws: with a
 * curriculum the statistics get one more row, log[1 + Tr + 2 + Tt] = mean terrain level over ALL envs
 * (curriculums.py:55), and log_sums holds 1 + (2 + Tr + 2 + Tt) * ceil(num_envs / 8) doubles.  The event terms move the
 * robot under the cast blocks of the same launch, so -- as without events -- the caller follows the launch with
 * mdp_raycaster_update_flagged(dones), which casts the reset envs again from their new pose.  Interval timers shared by
 * all envs (is_global_time) are not covered: use the stand-alone launches. */
typedef struct {
    const mdp_curriculum_terrain_t* curriculum;       /* terrain_levels_vel, or NULL */
    const mdp_event_root_state_t* reset_root_state;   /* mode "reset": reset_root_state_uniform, or NULL */
    const mdp_event_joints_t* reset_joints;           /* mode "reset": reset_joints_by_scale / _by_offset (after the root state), or NULL */
    const mdp_event_reset_mark_t* reset_mark;         /* EventManager bookkeeping of the reset terms, or NULL */
    const mdp_event_push_t* push;                     /* mode "interval": push_by_setting_velocity, or NULL */
    float* push_time_left;                            /* (N) timers of the interval term */
    float push_lo, push_scale;                        /* resampled interval = u * scale + lo */
    const float* push_tick_u;                         /* (N) injected uniforms of the resampling, or NULL */
    uint64_t* push_tick_rng;                          /* 1 + N generator words (see the event terms), or NULL with push_tick_u */
    uint8_t* push_trigger;                            /* (N) out: 1 where the timer fired, or NULL */
    float interval_dt;                                /* env step dt (event_manager.py:203) */
    int32_t pad0;
} mdp_post_events_t;
int mdp_post_physics_step_with_reset_events(int num_envs, const mdp_scene_update_t* update, float physics_dt,
                                            const mdp_termination_reward_args_t* termination_reward,
                                            const mdp_velocity_command_args_t* command,
                                            const mdp_obs_args_t* observation,
                                            const mdp_raycaster_update_args_t* cast,
                                            const mdp_post_reset_args_t* reset, const mdp_post_events_t* events,
                                            void* stream);

/* Imu._update_buffers_impl (LAB/sensors/imu/imu.py:142-182; SURVEY.md 8f N4: a sensor that is pure math over the body
 * state) for the envs flagged in is_outdated (NULL = all; the flags are cleared and timestamp_last_update = timestamp,
 * sensor_base.py:284-293): pose of the sensor frame (offset applied), velocity of the link's centre of mass moved to
 * the sensor, finite-difference accelerations (+ gravity bias) -- all rotated into the sensor frame. */
typedef struct {
    uint8_t* is_outdated;            /* (N) or NULL */
    const float* timestamp;          /* (N) or NULL */
    float* timestamp_last_update;    /* (N) or NULL */
    const float* transforms;         /* (N,7) x, y, z, qx, qy, qz, qw of the body (RigidBodyView.get_transforms) */
    const float* velocities;         /* (N,6) (get_velocities) */
    const float* coms;               /* (N, ld_coms): centre-of-mass position in the body frame first (get_coms) */
    int32_t ld_coms;
    float dt;                        /* time since the previous update (imu.py:135-140) */
    const float* offset_pos_b;       /* (N,3) */
    const float* offset_quat_b;      /* (N,4) wxyz */
    const float* gravity_bias_w;     /* (N,3) */
    float* prev_lin_vel_w;           /* (N,3) in/out */
    float* prev_ang_vel_w;           /* (N,3) in/out */
    float* pos_w;                    /* outputs: (N,3), (N,4), 4 x (N,3) */
    float* quat_w;
    float* lin_vel_b;
    float* ang_vel_b;
    float* lin_acc_b;
    float* ang_acc_b;
} mdp_imu_args_t;
int mdp_imu_update(int num_envs, const mdp_imu_args_t* args, void* stream);

/* FrameTransformer._update_buffers_impl (LAB/sensors/frame_transformer/frame_transformer.py:322-385): poses of
 * num_frames target frames per env, in the world and relative to the env's source frame.  The index tables are the
 * ones the reference resolves from the USD stage at initialisation (:133-320): rows of the rigid-body view reordered
 * per env (per_env_indices), the source body row per env, the target body rows, and for every (env, frame) which
 * target body it sits on (duplicate_frame_indices).  Offsets: (N,3)+(N,4) for the source, (N*F,3)+(N*F,4) for the
 * targets, or NULL.  gate (device float, e.g. the number of outdated envs) == 0 -> the launch does nothing, like the
 * reference, which recomputes every env as soon as one is outdated and none otherwise. */
typedef struct {
    const float* gate;                   /* or NULL: always run */
    uint8_t* is_outdated;                /* (N) or NULL */
    const float* timestamp;
    float* timestamp_last_update;
    const float* transforms;             /* (M,7) x, y, z, qx, qy, qz, qw: frame_physx_view.get_transforms() */
    const int64_t* per_env_indices;      /* (M) */
    const int64_t* source_frame_body_ids;/* (N) */
    const int64_t* target_frame_body_ids;/* (N * unique target bodies) */
    const int64_t* duplicate_frame_indices; /* (N * F) */
    const float* source_offset_pos;      /* (N,3) or NULL */
    const float* source_offset_quat;     /* (N,4) wxyz */
    const float* target_offset_pos;      /* (N*F,3) or NULL */
    const float* target_offset_quat;     /* (N*F,4) */
    float* source_pos_w;                 /* (N,3) */
    float* source_quat_w;                /* (N,4) */
    float* target_pos_w;                 /* (N,F,3) */
    float* target_quat_w;                /* (N,F,4) */
    float* target_pos_source;            /* (N,F,3) */
    float* target_quat_source;           /* (N,F,4) */
} mdp_frame_transformer_args_t;
int mdp_frame_transformer_update(int num_envs, int num_frames, const mdp_frame_transformer_args_t* args, void* stream);

/* RayCasterCamera._update_buffers_impl (LAB/sensors/ray_caster/ray_caster_camera.py:257-318) for the outdated envs:
 * camera pose = view pose combined with the offset (:415-430), rays of the pinhole pattern rotated by the full camera
 * orientation and cast (max_dist 1e6), and the depth post-processing of the requested data types:
 * distance_to_image_plane = x of the hit vector in the camera frame, distance_to_camera = ray length, both clipped per
 * depth_clipping_behavior (0 "none", 1 "max": beyond max_distance or NaN -> max_distance, 2 "zero": -> 0), normals of
 * the hit faces.  Output pointers of data types that were not requested are NULL.  frame (N, int64) counts updates. */
typedef struct {
    const mdp_mesh_t* mesh;
    uint8_t* is_outdated;              /* (N) or NULL = every env */
    const float* timestamp;
    float* timestamp_last_update;
    const float* view_pose;            /* (N, ld_pose): position + quaternion of the view the camera is attached to */
    int32_t ld_pose;
    int32_t pose_xyzw;                 /* 1: physics order (x, y, z, w); 0: (w, x, y, z) */
    const float* offset_pos;           /* (N,3) */
    const float* offset_quat;          /* (N,4) wxyz, world convention */
    int32_t num_rays;
    int32_t depth_clipping;
    float max_distance;
    int32_t pad0;
    const float* ray_starts;           /* (N,R,3) camera frame */
    const float* ray_directions;       /* (N,R,3) */
    float* pos_w;                      /* (N,3) out */
    float* quat_w_world;               /* (N,4) out */
    float* ray_hits_w;                 /* (N,R,3) out */
    float* distance_to_image_plane;    /* (N,R) or NULL */
    float* distance_to_camera;         /* (N,R) or NULL */
    float* normals;                    /* (N,R,3) or NULL */
    int64_t* frame;                    /* (N) or NULL */
} mdp_raycaster_camera_args_t;
int mdp_raycaster_camera_update(int num_envs, const mdp_raycaster_camera_args_t* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDP_HIP_H */
