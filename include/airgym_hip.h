/*
 * airgym_hip.h - C ABI of libairgym_hip.so, the MI355X (gfx950) hot path of the
 * AirGym vectorised quadrotor environments.
 *
 * One handle owns the SoA state of `num_envs` environments on one HIP device and
 * advances all of them with ONE fused kernel launch per env step:
 *
 *   action map -> control cascade -> rotor wrench -> RK4 rigid body -> progress++
 *   -> observation (+noise) -> reward + termination -> in-place reset of done envs
 *
 * What each entry point replaces in the reference (emNavi/AirGym, paths relative
 * to the reference root):
 *
 *   ag_create            BaseTask.__init__ buffer allocation + create_sim/_create_envs
 *                        (airgym/envs/base/base_task.py:40-95, airgym/envs/base/hovering.py:42-152,173-201)
 *                        and ParallelRate/Atti/Vel/PosControl(num_envs) (hovering.py:93-123)
 *   ag_reset_all         BaseTask.reset()'s reset_idx(all) half (base_task.py:107-111, hovering.py:310-335)
 *   ag_reset_envs        reset_idx(env_ids) called from the host on a subset (hovering.py:310-335)
 *   ag_step / _into      Hovering.step (hovering.py:286-308) incl. pre_physics_step (:203-281),
 *                        gym.simulate + refresh (PhysX, :290,:283-284), compute_observations (:337-358),
 *                        compute_reward (:360-459), reset_idx (:310-335); Tracking overrides
 *                        (airgym/envs/task/tracking.py:159-296); the rlPx4Controller calls
 *                        set_q_world/set_status/update (hovering.py:235-250)
 *   ag_step_rollout      the same step in the form A2CBase.play_steps consumes (lib/agent/a2c_base.py:662-695)
 *   ag_step_rollout_fused  that step with the policy sampling in front of it and the reward / episode accounting behind it
 *   ag_step_with_inputs  same, random numbers supplied by the caller (parity mode)
 *   ag_get_buffers       the tensors the env exposes: obs_buf, rew_buf, reset_buf, time_out_buf,
 *                        extras["item_reward_info"] (base_task.py:72-76, hovering.py:304-308,448-457)
 *   ag_get_state / ag_set_state
 *                        root_states view of gym.acquire_actor_root_state_tensor (hovering.py:60-77)
 *                        and gym.set_actor_root_state_tensor (:331); progress_buf (:164-165);
 *                        pre_actions (:138); controller memory (inside rlPx4Controller objects)
 *   ag_compact_reset_ids reset_buf.nonzero(as_tuple=False).squeeze(-1) (hovering.py:209,300)
 *   ag_set_target_state  Hovering.callback (hovering.py:154-156)
 *
 * Conventions
 *   - every function returns 0 on success, a negative ag_status on failure;
 *     ag_last_error() returns a human-readable message for the calling thread.
 *   - all pointers named *_dev are DEVICE pointers on the handle's device; the
 *     library never synchronises the host inside ag_step*; work is enqueued on the
 *     `stream` argument (a hipStream_t passed as void*, NULL = the null stream).
 *   - one handle is not thread-safe; distinct handles are independent.
 *   - memory: if `arena_dev` passed to ag_create is NULL the library hipMalloc()s
 *     ag_arena_bytes(cfg) bytes and frees them in ag_destroy; otherwise the caller
 *     owns the arena (>= ag_arena_bytes(cfg), 256-byte aligned) and must keep it
 *     alive until ag_destroy.  Pointers returned by ag_get_buffers point into the
 *     arena and stay valid until ag_destroy.
 */
#ifndef AIRGYM_HIP_H
#define AIRGYM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AG_VERSION 100 /* 0.1.0 */

typedef struct ag_env* ag_handle;

typedef enum ag_status {
    AG_OK = 0,
    AG_ERR_INVALID_ARG = -1,   /* NULL pointer, bad enum, num_envs <= 0, ... */
    AG_ERR_UNKNOWN_TASK = -2,  /* task_registry.py:78-79 raises ValueError */
    AG_ERR_UNKNOWN_CTL = -3,   /* hovering.py:122-123 only prints "Mode Error!"; here it is an error */
    AG_ERR_HIP = -4,           /* a HIP runtime call failed; message has hipGetErrorString */
    AG_ERR_NO_DEVICE = -5,     /* no usable gfx950 device */
    AG_ERR_UNSUPPORTED = -6    /* valid request this build has no kernel for */
} ag_status;

typedef enum ag_task {
    AG_TASK_HOVERING = 0, /* airgym/envs/base/hovering.py, 18 obs, 24 s episodes */
    AG_TASK_TRACKING = 1, /* airgym/envs/task/tracking.py, 48 obs, 36 s episodes */
    AG_TASK_PLANNING = 2, /* airgym/envs/task/planning.py (on base/customized.py): 16 obs + 212x120 depth image,
                             16 s episodes, 40 cylinder obstacles + goal per env; ctl_mode pos|vel|rate|prop */
    AG_TASK_BALLOON = 3,  /* airgym/envs/task/balloon.py (on base/customized.py): 18 noisy obs relative to a static target
                             ball, 8 s episodes, no camera; all five ctl modes */
    AG_TASK_AVOID = 4     /* airgym/envs/task/avoid.py (on base/customized.py): 16 obs + 212x120 depth image, 6 s episodes,
                             one thrown 0.3 m cube per env (avoid.py:58-90); ctl_mode pos|vel|rate|prop */
} ag_task;

/* --ctl_mode pos|vel|atti|rate|prop  (README: PY / LV / CTA / CTBR / SRT) */
typedef enum ag_ctl_mode {
    AG_CTL_POS = 0,
    AG_CTL_VEL = 1,
    AG_CTL_ATTI = 2,
    AG_CTL_RATE = 3,
    AG_CTL_PROP = 4
} ag_ctl_mode;

enum {
    AG_FLAG_REWARD_TERMS = 1u << 0, /* emit the 9 item_reward_info arrays + cmd_thrusts each step */
    AG_FLAG_OBS_NOISE_OFF = 1u << 1, /* testing aid: skip add_noise (reference: always on, hovering.py:343) */
    AG_FLAG_STAGGER_PHASE = 1u << 3, /* opt-in (Hovering): a FULL reset (ag_create, ag_reset_all) starts env i at
                                        progress ~ U{0 .. max_episode_length - 2}, drawn from the counter RNG (stream 2) keyed
                                        by the GLOBAL env id, instead of 0 - so that the 2 400-step time limit does not end every
                                        episode of the shard in the same rollout.  Only the first episode of an env is
                                        shorter; in-step resets (hovering.py:300-302) still start at progress 0 = the reference
                                        (hovering.py:310-335 sets progress_buf[env_ids] = 0 for all envs at once) */
    AG_FLAG_FIX_TIME_OUTS = 1u << 2  /* opt-in (Hovering / Tracking): time_out_buf flags the envs whose episode reached the time
                                        limit this step (progress >= max_episode_length - 1 before the reset).  Default off =
                                        the reference: hovering.py:304 evaluates `progress_buf > max_episode_length` AFTER
                                        reset_idx zeroed the progress (:300-302,:435), so extras["time_outs"] is never true and
                                        the PPO loop's bootstrap (lib/agent/a2c_base.py:672-673) never fires */
};

#define AG_NUM_REWARD_TERMS 11 /* Hovering/Tracking use the first 9, Planning all 11 */
#define AG_MAX_ACTIONS 5
#define AG_STATE_DIM 13     /* pos3 quat_xyzw4 linvel3 angvel3 (world frame), hovering.py:73-77 */
#define AG_CTL_STATE_DIM 12 /* rate_int3 prev_body_rate3 vel_int3 prev_vel3 */
#define AG_NOISE_DIM 18
#define AG_RESET_UNIFORMS 12

typedef struct ag_config {
    uint32_t struct_size;       /* = sizeof(ag_config), ABI guard */
    int32_t task;               /* ag_task */
    int32_t ctl_mode;           /* ag_ctl_mode */
    int32_t num_envs;           /* envs owned by this handle (one GPU's shard) */
    int32_t device;             /* HIP device ordinal */
    uint32_t flags;             /* AG_FLAG_* */
    uint64_t seed;              /* Philox key */
    uint32_t env_id_offset;     /* global id of local env 0: results do not depend on the sharding */
    double dt;                  /* sim.dt, 0.01 in every shipped config (double: the oracle rounds dt/6 etc. once) */
    int32_t max_episode_length; /* int(episode_length_s / dt); <= 0 selects the task default */
    float target_state[18];     /* cfg.env.target_state: flattened 3x3 attitude, pos3, linvel3, angvel3 */
} ag_config;

/* Reward-term order in ag_buffers.reward_terms[k] (each float[num_envs]).
 * Hovering (hovering.py:448-457): continous_action, effort, thrust, pos, vel_direction, ups, spin, yaw, reward
 * Tracking (tracking.py:285-294): dist_norm, dist_reward, yaw, spin, continous_action, thrust, effort, ups, reward
 * Planning (planning.py:294-305): continous_action, heading, speed, forward, alive, ups, z, esdf, thrust, reach_goal, reward
 * Balloon (balloon.py:228-235): guidance, hit, action_smoothness, effort, ups, reward
 * Avoid (avoid.py:289-298): pose, ups, spin, effort, action_smoothness, thrust, alive, reward */
typedef struct ag_buffers {
    int32_t num_envs;
    int32_t num_obs;
    int32_t num_actions;
    int32_t max_episode_length;
    float* obs_dev;           /* [num_envs, num_obs] row-major f32 */
    float* rew_dev;           /* [num_envs] f32 */
    int64_t* reset_dev;       /* [num_envs] int64 0/1 (base_task.py:75) */
    uint8_t* timeout_dev;     /* [num_envs] u8  progress > max_episode_length (hovering.py:304; see AG_FLAG_FIX_TIME_OUTS) */
    uint64_t* reset_mask_dev; /* [ceil(num_envs/64)] one ballot word per wavefront, bit l = env 64*w+l done */
    int32_t* reset_ids_dev;   /* [num_envs] ascending ids, valid after ag_compact_reset_ids */
    int32_t* reset_count_dev; /* [1] */
    float* reward_terms_dev[AG_NUM_REWARD_TERMS]; /* NULL unless AG_FLAG_REWARD_TERMS */
    float* cmd_thrusts_dev;   /* [num_envs,4] NULL unless AG_FLAG_REWARD_TERMS */
} ag_buffers;

/* Host-visible view of the env state for checkpoints and tests; every member is a
 * DEVICE pointer supplied by the caller, any of them may be NULL (skipped). */
typedef struct ag_state_view {
    float* root_states_dev;   /* [num_envs, 13] */
    float* ctl_state_dev;     /* [num_envs, 12] */
    float* pre_actions_dev;   /* [num_envs, num_actions] */
    int32_t* progress_dev;    /* [num_envs] */
    int32_t* was_reset_dev;   /* [num_envs] 1 = reset at the end of the previous step (thrust zeroed next step) */
} ag_state_view;

int ag_version(void);
const char* ag_last_error(void);

int ag_num_obs(int task);                      /* 18 / 48 / 16 / 18 / 16, <0 on unknown task */
int ag_num_actions(int ctl_mode);              /* 5 for atti else 4 (hovering.py:47) */
int ag_default_episode_length(int task, double dt);
size_t ag_arena_bytes(const ag_config* cfg);   /* 0 on invalid cfg */

int ag_create(const ag_config* cfg, void* arena_dev, ag_handle* out);
int ag_destroy(ag_handle h);

int ag_reset_all(ag_handle h, void* stream);
/* reset_idx(env_ids) as a host call for a SUBSET of the envs (hovering.py:310-335, tracking.py:159-192, planning.py:63-136,
 * balloon.py:57-99, avoid.py:91-163): env_ids_dev [count] int32 device array (ids outside [0, num_envs) are ignored,
 * duplicates are harmless).  The listed envs are re-randomised from the counter RNG, reset_buf = 1 and the ballot-mask bit
 * are set, progress / pre_actions / controller memory cleared; the others are untouched. */
int ag_reset_envs(ag_handle h, const int32_t* env_ids_dev, int count, void* stream);
int ag_step(ag_handle h, const float* actions_dev, void* stream);
/* Same as ag_step but obs / reward / done flags are written to caller buffers (e.g. slot t of a
 * rollout buffer) instead of the handle's own; any of the three may be NULL = use the handle's. */
int ag_step_into(ag_handle h, const float* actions_dev, float* obs_out_dev, float* rew_out_dev,
                 int64_t* reset_out_dev, void* stream);
/* Rollout form of ag_step_into, what A2CBase.play_steps needs from Hovering.step (lib/agent/a2c_base.py:662-695):
 * obs / reward written into rollout slots, done flags as u8 (the width ExperienceBuffer stores dones in,
 * lib/core/experience.py:329) and - instead of the nine per-env item_reward_info arrays - per-tile sums of the reward
 * terms for the Episode/<term> means (lib/utils/isaacgym_utils.py:66-99): term_sums_dev [ag_term_sum_tiles(num_envs), 12]
 * f32 (NULL = none), row b = sums over envs 64b .. 64b+63 of reward_terms[0..8].  The handle's int64 reset buffer, term
 * arrays and cmd_thrusts are NOT written by this call; reset_mask / timeout are.  Hovering / Tracking handles. */
int ag_term_sum_tiles(int num_envs);
int ag_step_rollout(ag_handle h, const float* actions_dev, float* obs_out_dev, float* rew_out_dev, uint8_t* done_out_dev,
                    float* term_sums_dev, void* stream);
/* num_steps consecutive env steps in ONE launch: the loop `for t in range(K): env.step(actions[t])` of a caller that
 * already holds the K actions (replay of recorded actions, open-loop / scripted evaluation, the env-only throughput line of
 * SURVEY.md 8(d) whose actions are pre-generated on the device; reference loop: Hovering.step, hovering.py:286-308, under
 * A2CBase.play_steps, lib/agent/a2c_base.py:651-711).  Results are IDENTICAL, bit for bit, to num_steps calls of
 * ag_step_rollout(actions_dev + t * num_envs * A, obs_out_dev + t * num_envs * num_obs, rew_out_dev + t * num_envs,
 * done_out_dev + t * num_envs, term_sums_dev + t * tiles * 12) - same kernel, same arithmetic, Philox ticks tick0 + t, in-step
 * resets included - but the env state, the controller memory and the previous action stay in registers between the steps:
 * per env-step only the action (read) and observation / reward / done (written) move through HBM; the state is read once and
 * written once per launch.  timeout_out_dev: [num_steps, num_envs] u8 or NULL (the handle's time_out buffer keeps the last
 * step's flags).  actions_dev [num_steps, num_envs, A]; obs_out_dev [num_steps, num_envs, num_obs]; rew_out_dev
 * [num_steps, num_envs]; done_out_dev [num_steps, num_envs] u8; term_sums_dev [num_steps, ag_term_sum_tiles(n), 12] or NULL.
 * num_steps > 1 needs num_envs * num_obs % 4 == 0 (16-byte aligned slices).  Hovering / Tracking handles. */
int ag_step_multi(ag_handle h, const float* actions_dev, int num_steps, float* obs_out_dev, float* rew_out_dev,
                  uint8_t* done_out_dev, uint8_t* timeout_out_dev, float* term_sums_dev, void* stream);
/* One whole step of A2CBase.play_steps behind the policy GEMMs (lib/agent/a2c_base.py:651-695) as ONE launch: what
 * ag_policy_sample does in front of ag_step_rollout (get_action_values' sampling, a2c_continuous_logstd_model.py:159-167;
 * preprocess_actions, a2c_base.py:229-236) and what ag_rollout_account does behind it (rewards_shaper + time-out bootstrap,
 * :668-673; current_rewards / current_lengths and the sums over the episodes that ended, :678-695) run inside the env-step
 * kernel: the action goes from the sampler to the integrator through LDS, reward / done flags reach the accounting in LDS.
 * Same arithmetic as the three separate calls (bit-identical outputs; the episode sums are per 64-env tile instead of per
 * 256-env block).  partials_dev: [ag_term_sum_tiles(num_envs), 4] double.  Other members: as the arguments of the same
 * name of ag_policy_sample / ag_rollout_account; bootstrap_timeouts != 0 adds gamma * value where the step's time-out flag
 * is set.  Hovering / Tracking handles. */
typedef struct ag_rollout_tail {
    uint32_t struct_size;            /* = sizeof(ag_rollout_tail), ABI guard */
    const float* heads_dev;          /* [n, A+1] mu | normalised value */
    const float* logstd_dev;         /* [A] */
    const double* vmean_dev;         /* [1] or NULL */
    const double* vvar_dev;          /* [1] or NULL */
    float veps;
    unsigned long long seed;         /* Philox key of the action noise */
    const long long* counter_dev;    /* [1] rollout counter */
    int horizon, slot;
    long long id_offset;
    float* actions_dev;              /* [n, A] */
    float* neglogp_dev;              /* [n] */
    float* values_dev;               /* [n] */
    float* mus_dev;                  /* [n, A] */
    float* sigmas_dev;               /* [n, A] */
    float scale, shift, min_val, max_val;
    int log_val;
    float gamma;
    int bootstrap_timeouts;
    float* shaped_dev;               /* [n] */
    float* cur_rew_dev;              /* [n] read-modify-write */
    float* cur_shaped_dev;           /* [n] read-modify-write */
    float* cur_len_dev;              /* [n] read-modify-write */
    double* partials_dev;            /* [ag_term_sum_tiles(n), 4] {episodes ended, sum reward, sum shaped, sum length} */
} ag_rollout_tail;
int ag_step_rollout_fused(ag_handle h, const ag_rollout_tail* tail, float* obs_out_dev, float* rew_out_dev,
                          uint8_t* done_out_dev, float* term_sums_dev, void* stream);
/* Parity mode: noise_dev [num_envs,18] standard normals, reset_uniforms_dev [num_envs,12] U[0,1). */
int ag_step_with_inputs(ag_handle h, const float* actions_dev, const float* noise_dev,
                        const float* reset_uniforms_dev, void* stream);

/* Parity / inspection mode: compute_observations + compute_quadcopter_reward (hovering.py:337-459, tracking.py:202-296)
 * evaluated on the handle's CURRENT state (as left by ag_set_state: root state, progress, pre_actions), with the
 * processed action (self.actions after hovering.py:212-216) and the controller output (self.cmd_thrusts, :235-254)
 * supplied by the caller; noise_dev [num_envs,18] standard normals or NULL (no noise).  No integration, no reset, the
 * state is untouched; obs / reward / reset(int64) / reward terms land in the handle's own buffers.  This is how the
 * golden vectors recorded from the reference's own methods are replayed on the HIP code object (tests/test_gpu_golden.py). */
int ag_eval_obs_reward(ag_handle h, const float* processed_actions_dev, const float* cmd_thrusts_dev, const float* noise_dev,
                       void* stream);

int ag_get_buffers(ag_handle h, ag_buffers* out);
int ag_get_state(ag_handle h, const ag_state_view* view, void* stream);
int ag_set_state(ag_handle h, const ag_state_view* view, void* stream);
int ag_compact_reset_ids(ag_handle h, void* stream);
int ag_set_target_state(ag_handle h, const float* target_state18);
uint64_t ag_get_tick(ag_handle h);
int ag_set_tick(ag_handle h, uint64_t tick);

/* ---- PPO minibatch loss, fused (airgym_amd/csrc/ppo_kernels.hip) -------------------------------------
 * Replaces the eager op chain of ContinuousA2CBase.calc_gradients (lib/agent/a2c_continuous.py:299-369):
 * neglogp (lib/model/a2c_continuous_logstd_model.py:195-198), actor_loss / critic_loss
 * (lib/core/common_losses.py:10-20,39-48), bound_loss (a2c_continuous.py:382-390), policy_kl
 * (lib/core/torch_ext.py:27-36) and PPODataset.update_mu_sigma (lib/core/datasets.py:20-24).
 *   heads_dev [M, A+1]: mu columns then the value column (one GEMM output); logstd_dev [A].
 *   d_heads_dev [M, A+1]: d(a_loss.mean + 0.5*critic_coef*c_loss.mean + bounds_coef*b_loss.mean)/d heads.
 *   partials_dev [ag_ppo_loss_max_blocks(), ag_ppo_loss_num_sums()]: per-block sums of
 *       {a_loss, c_loss, b_loss, kl, d(sum_i a_i)/d logstd_0..AG_MAX_ACTIONS-1, column sums of d_heads 0..AG_MAX_ACTIONS,
 *        number of rows whose probability ratio left [1 - e_clip, 1 + e_clip]};
 *       *num_blocks_out rows are valid; the caller reduces them (deterministic) and divides by M, or hands them to
 *       ag_ppo_loss_finalize.  bound_type: 0 none, 1 'bound', 2 'regularisation'.
 *   new_mu_dev / new_sigma_dev [M, A]: optional write-back of the current policy rows (both or neither). */
int ag_ppo_loss_num_sums(void);
int ag_ppo_loss_max_blocks(void);
int ag_ppo_loss(const float* heads_dev, const float* logstd_dev, const float* actions_dev,
                const float* old_neglogp_dev, const float* advantages_dev, const float* returns_dev,
                const float* old_values_dev, const float* old_mu_dev, const float* old_sigma_dev, int M, int A,
                float e_clip, float critic_coef, float bounds_loss_coef, int clip_value, int bound_type,
                float* d_heads_dev, float* new_mu_dev, float* new_sigma_dev, float* partials_dev,
                int* num_blocks_out, void* stream);

/* Reduce the partials of ag_ppo_loss in one workgroup and write what the optimizer step consumes:
 *   grad_logstd_dev [A] = d loss / d logstd (entropy term included), grad_head_bias_dev [A+1] = bias gradient of the fused
 *   mu|value head, *kl_out_dev = minibatch KL, stats_dev [8] = {a_loss, c_loss, entropy, b_loss, kl, total loss,
 *   clip fraction (PpoDiagnostics, lib/core/dignostics.py:49-59; torch_ext.policy_clip_fraction :168-178), 0}
 *   (a2c_continuous.py:340-369). */
int ag_ppo_loss_finalize(const float* partials_dev, int num_blocks, int M, int A, const float* logstd_dev,
                         float entropy_coef, float critic_coef, float bounds_loss_coef, float* grad_logstd_dev,
                         float* grad_head_bias_dev, float* kl_out_dev, float* stats_dev, void* stream);

/* RunningMeanStd.update (lib/core/running_mean_std.py:31-62) on a [rows, D] float batch: batch mean / unbiased variance in
 * float64, merged in place into mean_dev / var_dev [D] and *count_dev with the reference's parallel-variance formula.
 * scratch_dev: ag_rms_scratch_doubles(D) doubles.  D <= 256, rows >= 2. */
long long ag_rms_scratch_doubles(int D);
int ag_rms_update(const float* x_dev, long long rows, int D, double* mean_dev, double* var_dev, double* count_dev,
                  double* scratch_dev, void* stream);

/* out = clamp((x - mean) / sqrt(var + eps), -clip, clip) over [rows, D] with float64 running statistics
 * (lib/core/running_mean_std.py:64-79, RunningMeanStd.forward in eval mode). */
int ag_normalize_rows(const float* x_dev, const double* mean_dev, const double* var_dev, float* out_dev, long long rows,
                      int D, float eps, float clip, void* stream);

/* Rollout bookkeeping of A2CBase.play_steps (lib/agent/a2c_base.py:651-695) and GAE (a2c_base.py:463-478).
 *   ag_policy_sample: actions = mu + sigma * N(0,1) from heads [n, A+1] (mu | value) and logstd [A]; the noise is
 *       Philox4x32-10 keyed by `seed` with counter (id_offset + env, (*counter_dev) * horizon + slot, stream 16, block), so a
 *       captured hipGraph draws fresh noise on every replay once the caller bumps *counter_dev (int64, device).
 *       Writes actions / mus / sigmas [n, A], neglogp [n], values [n] (de-normalised with vmean/vvar when given,
 *       base_model.py:29-35) and, optionally, env_actions = clamp(actions, -1, 1) (a2c_base.py:229-236).
 *   ag_rollout_account: shaped = clamp((r + shift) * scale, min, max) [log] (+ gamma * value on time-outs), running episode
 *       reward / shaped reward / length, and per-block partial sums {episodes ended, sum reward, sum shaped, sum length}
 *       in partials_dev [ag_rollout_account_blocks(n), 4] (double); running sums are cleared where dones != 0.
 *       dones_dev is u8 [n], the width the rollout buffer stores (lib/core/experience.py:329) and ag_step_rollout writes.
 *   ag_gae: dones_dev u8 [H+1, n] (dones[t] = done entering step t), rewards / values / advs / returns [H, n]. */
int ag_policy_sample(const float* heads_dev, const float* logstd_dev, const double* vmean_dev, const double* vvar_dev,
                     float veps, unsigned long long seed, const long long* counter_dev, int horizon, int slot,
                     long long id_offset, float* actions_dev, float* neglogp_dev, float* values_dev, float* mus_dev,
                     float* sigmas_dev, float* env_actions_dev, int n, int A, void* stream);
int ag_rollout_account_blocks(int n);
int ag_rollout_account(const float* raw_reward_dev, const unsigned char* dones_dev, const unsigned char* timeouts_dev,
                       const float* values_dev, float scale, float shift, float min_val, float max_val, int log_val,
                       float gamma, float* shaped_dev, float* cur_rew_dev, float* cur_shaped_dev, float* cur_len_dev,
                       double* partials_dev, int n, void* stream);
int ag_gae(const float* rewards_dev, const float* values_dev, const unsigned char* dones_dev, const float* last_values_dev,
           float gamma, float tau, float* advs_dev, float* returns_dev, int H, int n, void* stream);

/* Fused edges of the MLP trunk (lib/network/mlp.py:36-39, a2c_continuous_logstd_model.py:126-146); the wide GEMMs in
 * between stay with hipBLASLt.
 *   ag_mlp_input_layer: xn = clamp((obs - mean)/sqrt(var + eps), +-clip) [M, D] (skipped when mean/var/xn are all NULL,
 *       then obs is used as is), h = ELU(xn W^T + bias) [M, C]; W [C, D] row-major; D*C + 64*D floats must fit 64 KB.
 *   ag_elu_heads: heads [M, A1] = ELU(zh) Wh^T + bh; with write_back zh [M, C] <- ELU(zh) in place, without it zh keeps
 *       the pre-activation (then pass h_is_preactivation = 1 to ag_heads_bwd_elu_wgrad); zbias_dev [C] (optional) is added
 *       to zh before the ELU - for a producing GEMM run WITHOUT its bias epilogue (pass the same pointer to the backward); Wh [A1, C]; C a power of two
 *       64..256, A1 = A + 1 in {5, 6}.
 */
int ag_mlp_input_layer(const float* obs_dev, const double* mean_dev, const double* var_dev, const float* W_dev,
                       const float* bias_dev, float* xn_dev, float* h_dev, int M, int D, int C, float eps, float clip,
                       void* stream);
int ag_elu_heads(float* zh_dev, const float* Wh_dev, const float* bh_dev, float* heads_dev, int M, int C, int A1,
                 int write_back, const float* zbias_dev, void* stream);

/* float32-accurate GEMM on the bf16 matrix cores (airgym_amd/csrc/split_gemm.hip) for the 256 x 256 hidden layer
 * (lib/network/mlp.py:36-39: forward X W^T; autograd's dX = dZ W): every f32 operand is split EXACTLY into three bf16 pieces
 * and six of the nine cross products are accumulated in f32 - error <= one f32 rounding per product, at 6/16 of the cost of
 * the f32-input MFMA path (gfx950 has no TF32 form).  n = k = 256 only (AG_ERR_UNSUPPORTED otherwise).
 *   ag_split_gemm_prepare: W_dev [256, 256] f32 row-major -> planes_dev (ag_split_gemm_plane_bytes() bytes, 16-byte aligned);
 *       transpose = 0: B = W (C = A W^T), 1: B = W^T (C = A W).  Run once per weight update.
 *   ag_split_gemm_prepare_pair: both images of one weight in one launch (planes_dev: transpose 0, planes_t_dev: transpose 1).
 *   ag_split_gemm: C_dev [M, 256] = A_dev [M, 256] B^T (+ bias_dev [256] if not NULL).
 *   ag_split_gemm_elu_heads: the last hidden layer and the actor/critic heads in one launch (mlp.py:36-39 + the mu / value
 *       Linear): Z_dev [M, 256] = A B^T WITHOUT the bias (what ag_heads_bwd_elu_wgrad(zbias) reads back), heads_dev [M, A1] =
 *       ELU(Z + bias_dev) Wh_dev^T + bh_dev formed from the accumulators (Wh_dev [A1, 256], A1 in {5, 6}) - the same
 *       result as ag_split_gemm followed by ag_elu_heads(write_back = 0, zbias = bias_dev) without re-reading Z.
 *   ag_split_gemm_input_wgrad: the backward dX GEMM of the second layer with the first layer's whole backward in its epilogue
 *       (autograd of mlp.py:36-39 for a [D -> 256 -> 256] trunk; what ag_split_gemm(dZ, W^T planes) followed by
 *       ag_elu_bwd_input_wgrad computes): dh1 = dZ_dev W stays in registers, is multiplied by ELU'(h1_dev) and reduced against
 *       x_dev [M, D] over each tile of ag_split_gemm_input_wgrad_rows() rows: dw_partials_dev [tiles, 256, D], db_partials_dev
 *       [tiles, 256] (tiles = ceil(M / rows); the caller sums over dim 0, e.g. ag_sum_rows_multi).  Neither dh1 nor dz1 is
 *       written.  D in {16, 18, 20} (Hovering: 18) or 48 (Tracking, tracking.py:202-214): ag_split_gemm_input_wgrad_supported. */
/*   ag_split_wgrad: the weight gradient of the same layer, dW [256, 256] = dZ_dev^T X_dev (dZ_dev, X_dev: [M, 256] f32 row-major;
 *       autograd's grad_weight of mlp.py:36-39), both operands split three ways on the fly, the contraction running over the
 *       rows.  K = M is cut into `slices` contiguous row ranges, one workgroup each (ag_split_wgrad_slices(M) = one per CU, at
 *       most one per 16 rows); slice s writes partials_dev [s, 256, 256] and the caller sums over dim 0 in a fixed order
 *       (ag_sum_rows_multi).  Each operand is read from HBM once. */
int ag_split_wgrad_slices(int M);
int ag_split_wgrad(const float* dZ_dev, const float* X_dev, float* partials_dev, int M, int n, int k, int slices, void* stream);
long long ag_split_gemm_plane_bytes(void);
int ag_split_gemm_prepare(const float* W_dev, void* planes_dev, int n, int k, int transpose, void* stream);
int ag_split_gemm_prepare_pair(const float* W_dev, void* planes_dev, void* planes_t_dev, int n, int k, void* stream);
int ag_split_gemm(const float* A_dev, const void* planes_dev, const float* bias_dev, float* C_dev, int M, int n, int k,
                  void* stream);
int ag_split_gemm_input_wgrad_rows(void);
int ag_split_gemm_input_wgrad_supported(int D);   /* 1 for D in {16, 18, 20, 48} */
int ag_split_gemm_input_wgrad(const float* dZ_dev, const void* planes_dev, const float* h1_dev, const float* x_dev,
                              float* dw_partials_dev, float* db_partials_dev, int M, int n, int k, int D, void* stream);
/* Round 5 - the first layer's activations h1 = ELU(x W1^T + b1) are RECOMPUTED on the matrix cores wherever the backward of a
 * [D -> 256 -> 256] trunk needs them, instead of stored by the forward and read back twice (autograd keeps them as saved tensors,
 * lib/network/mlp.py:36-39 under lib/agent/a2c_continuous.py:299-369; 201 MB written + 402 MB read per 196 608-row minibatch):
 *   ag_split_gemm_input_wgrad_recompute: ag_split_gemm_input_wgrad with ELU'(h1) formed from a recomputation of the
 *       pre-activation z1 = x_ext W1ext^T (exact 3-way split, K = 32, natural orientation: it lands in the accumulator layout of
 *       dh1) - image_dev = the image of ag_split_gemm_input_prepare (its first-layer part), x_dev [M, D] the (normalised) network
 *       inputs.  D in {16, 18} (ag_split_gemm_input_wgrad_recompute_supported).  Same outputs as ag_split_gemm_input_wgrad.
 *   ag_split_wgrad_input: ag_split_wgrad for dW2 = dZ^T h1 with the X operand h1 produced from x_dev [M, D] and the same image
 *       (8 waves as 1 x 8; a wave's h1 tile IS its B fragment: the X operand never touches LDS or HBM).  M a multiple of 32,
 *       D in {16, 18, 20} (ag_split_wgrad_input_supported); slices = ag_split_wgrad_input_slices(M) (one workgroup per CU, at
 *       most one per 32 rows); partials_dev [slices, 256, 256], summed over dim 0 by the caller like ag_split_wgrad's.
 * With both, ag_split_gemm_input_loss_heads_bwd may be given h1_dev = NULL: it then does not write h1 at all. */
int ag_split_gemm_input_wgrad_recompute_supported(int D);
int ag_split_gemm_input_wgrad_recompute(const float* dZ_dev, const void* planes_dev, const void* image_dev, const float* x_dev,
                                        float* dw_partials_dev, float* db_partials_dev, int M, int n, int k, int D, int tile_rows,
                                        void* stream);      /* tile_rows: 0 / 256 or 128 (partials per that many rows) */
int ag_split_wgrad_input_supported(int D);
int ag_split_wgrad_input_slices(int M);
int ag_split_wgrad_input(const float* dZ_dev, const float* x_dev, const void* image_dev, float* partials_dev, int M, int n, int k,
                         int D, int slices, void* stream);
/* The whole actor-critic MLP [D -> 256 -> 256 -> (A + 1)] as ONE launch with the activations in registers
 * (airgym_amd/csrc/mlp_chain.hip): input normaliser -> Linear + ELU -> Linear + ELU -> mu | value heads
 * (ModelA2CContinuousLogStd.forward, lib/model/a2c_continuous_logstd_model.py:80-193; MLP, lib/network/mlp.py:36-39; fixed
 * sigma, shared trunk).  Same float32-accurate arithmetic as ag_split_gemm for all three products.  Replaces
 * ag_mlp_input_layer + ag_split_gemm_elu_heads in the rollout: neither h1 nor z2 goes through HBM.
 *   ag_mlp_chain_supported: 1 for C = 256, D + 1 <= 64, A1 in {5, 6}.
 *   ag_mlp_chain_image_bytes(D): size of the prepared weight image (16-byte aligned buffer).
 *   ag_mlp_chain_prepare: W1 [256, D], b1 [256], W2 [256, 256], Wh [A1, 256] (row-major f32) -> image.  Once per policy version.
 *   ag_mlp_chain_forward: heads_dev [M, A1] = Wh ELU(W2 ELU(W1 xn + b1) + b2) + bh with xn = clamp((obs - mean) /
 *       sqrt(var + eps), +-clip) (mean_dev / var_dev NULL: xn = obs).  Optional outputs (NULL = not written): xn_dev [M, D],
 *       h1_dev / h2_dev [M, 256] = the two layers' activations (after ELU). */
int ag_mlp_chain_supported(int D, int C, int A1);
long long ag_mlp_chain_image_bytes(int D);
int ag_mlp_chain_prepare(const float* W1_dev, const float* b1_dev, int D, const float* W2_dev, const float* Wh_dev, int A1,
                         void* image_dev, void* stream);
int ag_mlp_chain_forward(const float* obs_dev, const double* mean_dev, const double* var_dev, float eps, float clip,
                         const void* image_dev, const float* b2_dev, const float* bh_dev, float* heads_dev, float* xn_dev,
                         float* h1_dev, float* h2_dev, int M, int D, int A1, void* stream);
/* The last hidden layer's forward, the PPO loss and the head layer's backward in ONE launch (calc_gradients,
 * lib/agent/a2c_continuous.py:299-369, around lib/network/mlp.py:36-39): the GEMM of ag_split_gemm_elu_heads, then - in its
 * epilogue, with the activations h = ELU(z + bias) still in the accumulators and the heads in LDS - ag_ppo_loss's per-row
 * arithmetic (csrc/ppo_loss_math.hpp: the same expressions) and ag_heads_bwd_elu_wgrad's: dZ_dev [M, 256] = (d_heads Wh) *
 * ELU'(h) is written where the pre-activation would have gone, and one partial per row tile of the head weight gradient
 * dwh_partials [tiles, A1, 256], of this layer's bias gradient db_partials [tiles, 256] (column sums of dZ) and of the loss
 * sums loss_partials [tiles, ag_ppo_loss_num_sums()] (feed ag_ppo_loss_finalize with num_blocks = tiles).  tiles =
 * ceil(M / ag_split_gemm_loss_rows()).  Neither the pre-activation nor the heads nor d_heads goes through HBM (heads_dev, if not
 * NULL, receives the heads anyway); new_mu_dev / new_sigma_dev: the rows' (mu, sigma) written back (PPODataset.update_mu_sigma,
 * lib/core/datasets.py:20-24) or NULL.  A1 = 5 (four actions + value). */
typedef struct ag_loss_epilogue {
    uint32_t struct_size;              /* = sizeof(ag_loss_epilogue), ABI guard */
    const float* logstd_dev;           /* [A] */
    const float* actions_dev;          /* [M, A] */
    const float* old_neglogp_dev;      /* [M] */
    const float* advantages_dev;       /* [M] */
    const float* returns_dev;          /* [M] */
    const float* old_values_dev;       /* [M] */
    const float* old_mu_dev;           /* [M, A] */
    const float* old_sigma_dev;        /* [M, A] */
    float* new_mu_dev;                 /* [M, A] or NULL (may alias old_mu_dev) */
    float* new_sigma_dev;              /* [M, A] or NULL (may alias old_sigma_dev) */
    float* heads_dev;                  /* [M, A1] or NULL */
    float* loss_partials_dev;          /* [tiles, ag_ppo_loss_num_sums()] */
    float* dwh_partials_dev;           /* [tiles, A1, 256] */
    float* db_partials_dev;            /* [tiles, 256] */
    float e_clip, critic_coef, bounds_loss_coef;
    int clip_value, bound_type;        /* as ag_ppo_loss */
    int tile_rows;                     /* 0 / 256: one partial per 256 rows (8-wave workgroups); 128: per 128 rows (4-wave workgroups;
                                          for minibatches with fewer 256-row tiles than CUs - ag_split_gemm_pick_tile_rows) */
    int partial_tiles;                 /* capacity of loss_partials / dwh_partials / db_partials in tiles; the launch writes
                                          M / tile_rows of them and returns AG_ERR_INVALID_ARG when that is more (0: not checked) */
} ag_loss_epilogue;
int ag_split_gemm_pick_tile_rows(int M);   /* 128 when ceil(M / 256) < CUs and M % 128 == 0, else 256 */
int ag_split_gemm_loss_rows(void);
int ag_split_gemm_loss_heads_bwd(const float* A_dev, const void* planes_dev, const float* bias_dev, const float* Wh_dev,
                                 const float* bh_dev, float* dZ_dev, const ag_loss_epilogue* loss, int M, int n, int k, int A1,
                                 void* stream);
int ag_split_gemm_elu_heads(const float* A_dev, const void* planes_dev, const float* bias_dev, const float* Wh_dev,
                            const float* bh_dev, float* Z_dev, float* heads_dev, int M, int n, int k, int A1, void* stream);

/* The same launch with the FIRST layer of a [D -> 256 -> 256] trunk formed inside it (lib/network/mlp.py:36-39, first Linear + ELU,
 * behind the input normaliser of lib/core/running_mean_std.py:78-79): replaces ag_mlp_input_layer + ag_split_gemm_loss_heads_bwd in
 * the update.  h1 is produced on the matrix cores (the same exact 3-way split; bias through an all-ones input column) straight into
 * the GEMM's A operand - it is not read back - and written to h1_dev as a by-product (the backward reads it); xn_dev receives the
 * normalised inputs (the first layer's weight gradient needs them).  Weights come as ONE image made once per optimizer step:
 * ag_split_gemm_input_prepare(W1 [256, D], b1 [256], D, W2 [256, 256], image) - ag_split_gemm_input_image_bytes() bytes.
 * D in {16, 18, 20} (ag_split_gemm_input_fwd_supported), M a multiple of 256, A1 = 5.  Float32-accurate like the other split
 * products (not bit-identical to ag_mlp_input_layer's FMA chain). */
typedef struct ag_input_layer_args {
    uint32_t struct_size;              /* = sizeof(ag_input_layer_args), ABI guard */
    int D;                             /* input width */
    const float* obs_dev;              /* [M, D] observations (raw when mean/var are given) */
    const double* mean_dev;            /* [D] running mean or NULL (no normaliser: obs is used as it is) */
    const double* var_dev;             /* [D] running variance or NULL */
    float* xn_dev;                     /* [M, D] out: clamp((obs - mean) / sqrt(var + eps), +-clip); NULL iff mean_dev is NULL */
    float* h1_dev;                     /* [M, 256] out: ELU(xn W1^T + b1), or NULL: not written (the backward recomputes it) */
    float eps, clip;
} ag_input_layer_args;
int ag_split_gemm_input_fwd_supported(int D);
long long ag_split_gemm_input_image_bytes(void);
int ag_split_gemm_input_prepare(const float* W1_dev, const float* b1_dev, int D, const float* W2_dev, void* image_dev, void* stream);
/* ... the same plus the backward planes of W2 (= ag_split_gemm_prepare(W2, planes_t_dev, 256, 256, transpose = 1)) in one launch */
int ag_split_gemm_input_prepare_pair(const float* W1_dev, const float* b1_dev, int D, const float* W2_dev, void* image_dev,
                                     void* planes_t_dev, void* stream);
int ag_split_gemm_input_loss_heads_bwd(const ag_input_layer_args* in, const void* image_dev, const float* bias_dev,
                                       const float* Wh_dev, const float* bh_dev, float* dZ_dev, const ag_loss_epilogue* loss, int M,
                                       int n, int k, int A1, void* stream);

/* The first layer of the trunk as a launch of its own on the matrix cores (airgym_amd/csrc/first_layer.hip): xn = clamp((obs - mean) /
 * sqrt(var + eps), +-clip) (lib/core/running_mean_std.py:78-79; mean_dev / var_dev NULL: xn = obs and xn_dev MUST be NULL too; given:
 * xn_dev MUST be given - AG_ERR_INVALID_ARG otherwise), h1 = ELU(xn W1^T +
 * b1) (lib/network/mlp.py:36-39) - what ag_mlp_input_layer computes, with the product as an exact 3-way bf16 split on the MFMA (float32-
 * accurate, not bit-identical to the FMA chain).  For input widths the forward GEMM cannot produce itself (D + 1 <= 64, e.g. Tracking's
 * 48, tracking.py:202-214): 256-wide layer only.  prepare: W1 [256, D], b1 [256] -> image (ag_mlp_first_layer_image_bytes(D) bytes,
 * 16-byte aligned), once per optimizer step. */
int ag_mlp_first_layer_supported(int D, int C);
long long ag_mlp_first_layer_image_bytes(int D);
int ag_mlp_first_layer_prepare(const float* W1_dev, const float* b1_dev, int D, void* image_dev, void* stream);
int ag_mlp_first_layer(const float* obs_dev, const double* mean_dev, const double* var_dev, float eps, float clip, const void* image_dev,
                       float* xn_dev, float* h1_dev, int M, int D, void* stream);

/* mixed_precision (the reference's torch.cuda.amp switch, lib/agent/a2c_base.py:236-237,566,582: autocast around the model forward
 * of the rollout and of calc_gradients): every matrix-core entry point above exists a second time with the suffix _bf16 and the
 * SAME arguments - ONE bf16 MFMA per product instead of six (operands rounded to bf16, round to nearest; float32 accumulate; the
 * float32 master weights and the same prepared weight images, of which only the leading plane is read).  Everything around the
 * products (normaliser, ELU, PPO loss, reductions, Adam) stays float32.  Error per product <= 2^-7 |a||b| (both operands rounded). */
int ag_split_gemm_bf16(const float* A_dev, const void* planes_dev, const float* bias_dev, float* C_dev, int M, int n, int k, void* stream);
int ag_split_gemm_elu_heads_bf16(const float* A_dev, const void* planes_dev, const float* bias_dev, const float* Wh_dev,
                                 const float* bh_dev, float* Z_dev, float* heads_dev, int M, int n, int k, int A1, void* stream);
int ag_split_gemm_loss_heads_bwd_bf16(const float* A_dev, const void* planes_dev, const float* bias_dev, const float* Wh_dev,
                                      const float* bh_dev, float* dZ_dev, const ag_loss_epilogue* loss, int M, int n, int k, int A1,
                                      void* stream);
int ag_split_gemm_input_loss_heads_bwd_bf16(const ag_input_layer_args* in, const void* image_dev, const float* bias_dev,
                                            const float* Wh_dev, const float* bh_dev, float* dZ_dev, const ag_loss_epilogue* loss, int M,
                                            int n, int k, int A1, void* stream);
int ag_split_gemm_input_wgrad_bf16(const float* dZ_dev, const void* planes_dev, const float* h1_dev, const float* x_dev,
                                   float* dw_partials_dev, float* db_partials_dev, int M, int n, int k, int D, void* stream);
int ag_split_gemm_input_wgrad_recompute_bf16(const float* dZ_dev, const void* planes_dev, const void* image_dev, const float* x_dev,
                                             float* dw_partials_dev, float* db_partials_dev, int M, int n, int k, int D, int tile_rows,
                                             void* stream);
int ag_split_wgrad_bf16(const float* dZ_dev, const float* X_dev, float* partials_dev, int M, int n, int k, int slices, void* stream);
int ag_split_wgrad_input_bf16(const float* dZ_dev, const float* x_dev, const void* image_dev, float* partials_dev, int M, int n, int k,
                              int D, int slices, void* stream);
int ag_mlp_first_layer_bf16(const float* obs_dev, const double* mean_dev, const double* var_dev, float eps, float clip,
                            const void* image_dev, float* xn_dev, float* h1_dev, int M, int D, void* stream);
int ag_mlp_chain_forward_bf16(const float* obs_dev, const double* mean_dev, const double* var_dev, float eps, float clip,
                              const void* image_dev, const float* b2_dev, const float* bh_dev, float* heads_dev, float* xn_dev,
                              float* h1_dev, float* h2_dev, int M, int D, int A1, void* stream);

/* ReLU followed by BatchNorm2d on [N, C, H, W] float32 (NCHW, C <= 64) for the depth-image feature extractor (reference:
 * lib/network/cnn.py:3-33: Conv2d -> ReLU -> BatchNorm2d, three times) - airgym_amd/csrc/cnn_kernels.hip.  The ReLU output is
 * never materialised; x is the CONVOLUTION output.  HW = H * W.  blocks = ceil(N * C / ag_relu_bn_planes_per_block()).
 *   ag_relu_bn_stats     : partials_dev [blocks, C, 2] = per-channel (sum, sum of squares) of relu(x); the caller sums over
 *                          dim 0 and forms mean / biased variance (nn.BatchNorm2d training statistics).
 *   ag_relu_bn_apply     : y = relu(x) * scale[c] + shift[c]   (scale = gamma invstd, shift = beta - mean scale; with the
 *                          running statistics this is the eval-mode forward).
 *   ag_relu_bn_bwd_reduce: partials_dev [blocks, C, 2] = per-channel (sum dy, sum dy * xhat), xhat = (relu(x) - mean) invstd.
 *   ag_relu_bn_bwd_dx    : dx = [x > 0] coef[c][2] (dy - sums[c][0] coef[c][3] - xhat sums[c][1] coef[c][3]);
 *                          coef_dev [C, 4] = {mean, invstd, gamma invstd, 1 / (N HW)}, sums_dev [C, 2] = {dbeta, dgamma}.
 *   ag_relu_bn_stats_weighted / ag_relu_bn_bwd_dx_weighted: the same with per-image multiplicities weights_dev [N] (NULL = 1):
 *                          a minibatch that holds image i m_i times - the depth camera runs every 4th env step
 *                          (planning.py:153-156), so consecutive rollout samples of an env share their image - has the batch
 *                          statistics of its DISTINCT images weighted by m_i; in the backward dy is the gradient summed over the
 *                          copies and the two mean terms are scaled by m_i (coef[c][3] = 1 / (sum_i m_i HW)).  Same result as
 *                          the reference's computation on the full minibatch, on 1/4 of the images.
 *                          plane_sums_dev (NULL = off) [N * C]: the sum of dx over each plane - summed over images this is the bias
 *                          gradient of the convolution that produced x, which ag_cnn_conv_wgrad(with_bias = 0) then need not form.
 *                          border_sums_dev (NULL = off; W = plane width) [N * C][5]: what ag_plane_border_sums(dx) would return, formed
 *                          in passing. */
/*   ag_relu_bn_bwd_dx_plane: ag_relu_bn_bwd_dx_weighted with dy constant over each plane, dyp_dev [N * C]: the backward of the global
 *                          average pool that follows the extractor's last BatchNorm (cnn.py:14).  Not in place (dx_dev != x_dev: the
 *                          border sums read x again). */
int ag_relu_bn_planes_per_block(void);
int ag_relu_bn_stats(const float* x_dev, float* partials_dev, int N, int C, int HW, void* stream);
int ag_relu_bn_apply(const float* x_dev, const float* scale_dev, const float* shift_dev, float* y_dev, int N, int C, int HW,
                     void* stream);
int ag_relu_bn_bwd_reduce(const float* dy_dev, const float* x_dev, const float* mean_dev, const float* invstd_dev,
                          float* partials_dev, int N, int C, int HW, void* stream);
int ag_relu_bn_bwd_dx(const float* dy_dev, const float* x_dev, const float* coef_dev, const float* sums_dev, float* dx_dev,
                      int N, int C, int HW, void* stream);
int ag_relu_bn_stats_weighted(const float* x_dev, const float* weights_dev, float* partials_dev, int N, int C, int HW,
                              void* stream);
int ag_relu_bn_bwd_dx_weighted(const float* dy_dev, const float* x_dev, const float* coef_dev, const float* sums_dev,
                               const float* weights_dev, float* dx_dev, float* plane_sums_dev, float* border_sums_dev, int W, int N,
                               int C, int HW, void* stream);
int ag_relu_bn_bwd_dx_plane(const float* dyp_dev, const float* x_dev, const float* coef_dev, const float* sums_dev,
                            const float* weights_dev, float* dx_dev, float* plane_sums_dev, float* border_sums_dev, int W, int N, int C,
                            int HW, void* stream);

/* The per-channel arithmetic of ReLU + BatchNorm between the big passes (two-stage float64 column sums in a fixed order, then one
 * workgroup; each call replaces 10 - 30 launches of [C]-sized tensor operations per layer and step) - airgym_amd/csrc/cnn_kernels.hip.
 *   ag_bn_finalize     : forward.  stats_dev [n][G][C][2] = per (image, band) sums of relu(y), relu(y)^2 (ag_cnn_conv*_fwd's stats),
 *                        weights_dev [n] image multiplicities (NULL = 1), m = (sum of multiplicities) * HW.  training != 0: batch
 *                        statistics, running_mean / running_var / num_batches updated as nn.BatchNorm2d does (momentum, unbiased
 *                        variance); training == 0: the running statistics.  coef_dev [4][C] = {mean, invstd, scale = gamma invstd,
 *                        shift = beta - mean scale}.  With pooled_dev / plane1_dev [n][C] (the extractor's last layer, followed by
 *                        AdaptiveAvgPool2d((1, 1)), cnn.py:14): plane1 = per-image sum of relu(y), pooled = scale plane1 / HW + shift.
 *   ag_bn_bwd_prep     : backward.  partials_dev [blocks][C][2] of ag_relu_bn_bwd_reduce -> sums_dev [C][2] = {dbeta, dgamma} and
 *                        tab_dev [C][4]: mode 0 {mean, invstd, gamma invstd, 1 / m} (coef of ag_relu_bn_bwd_dx), mode 1 {A, B, C, 0}
 *                        (bn_tab of ag_cnn_conv1_wgrad).  coef_fwd_dev = ag_bn_finalize's coef.  dgamma_out_dev / dbeta_out_dev (NULL =
 *                        off) [C]: the two parameter gradients also written there (e.g. straight into the optimizer's gradient buffer).
 *   ag_bn_pool_bwd_prep: backward of the last layer from dpool_dev [n][C] (gradient of the pooled features) and plane1_dev:
 *                        sums {sum_n dpool, sum_n dpool (plane1 / HW - mean) invstd}, tab (mode 0), dyp_dev [n][C] = dpool / HW
 *                        (input of ag_relu_bn_bwd_dx_plane). */
/*   ag_plane_border_sums: out_dev [N * C][5] = per plane of dz_dev [N,C,H,W] {sum of row 0, sum of row H-1, sum of column 0, dz[0][0],
 *                        dz[H-1][0]}.  With them (and the plane totals of ag_relu_bn_bwd_dx*) the two reductions of the ReLU + BatchNorm
 *                        backward of the layer in FRONT of a convolution follow from that convolution's weights and weight gradient
 *                        (sum dy y = sum w dw, sum dy = sum w S; airgym_amd/lib/network/fused_cnn.py) - no pass over dy. */
int ag_plane_border_sums(const float* dz_dev, float* out_dev, int N, int C, int H, int W, void* stream);
/*   ag_bn_sums_from_conv: those identities in one launch.  w_dev / dw_dev [cout][cin][3][3] of the 3x3 / stride-2 / pad-1 convolution
 *                        BEHIND the ReLU + BatchNorm (input height hin), total_dev [cout] = sum of its output gradient per channel (its
 *                        bias gradient), border_dev [cout][5] = ag_plane_border_sums' rows summed over the images, gamma / beta [cin]
 *                        of the BatchNorm: part_dev [cin][2] = {sum dy, sum dy xhat} - one "partial" row for ag_bn_bwd_prep (blocks = 1).
 *                        gamma = 0 is clamped to 1e-30 (xhat is not recoverable from a constant y: use ag_relu_bn_bwd_reduce there). */
int ag_bn_sums_from_conv(const float* w_dev, const float* dw_dev, const float* total_dev, const float* border_dev, int cout, int cin,
                         int hin, const float* gamma_dev, const float* beta_dev, float* part_dev, void* stream);
long long ag_bn_scratch_doubles(void);      /* scratch_dev: this many doubles (stage-1 partial sums), reusable between calls on a stream */
int ag_bn_finalize(const float* stats_dev, const float* weights_dev, long long n, int G, int C, double m, const float* gamma_dev,
                   const float* beta_dev, float* running_mean_dev, float* running_var_dev, long long* num_batches_dev,
                   float momentum, double eps, int training, float* coef_dev, float* plane1_dev, float* pooled_dev, int HW,
                   double* scratch_dev, void* stream);
int ag_bn_bwd_prep(const float* partials_dev, long long blocks, int C, const float* coef_fwd_dev, const float* gamma_dev, double m,
                   int mode, float* sums_dev, float* tab_dev, float* dgamma_out_dev, float* dbeta_out_dev, double* scratch_dev,
                   void* stream);
int ag_bn_pool_bwd_prep(const float* dpool_dev, const float* plane1_dev, long long n, int C, const float* coef_fwd_dev,
                        const float* gamma_dev, double m, int HW, float* sums_dev, float* tab_dev, float* dyp_dev,
                        float* dgamma_out_dev, float* dbeta_out_dev, double* scratch_dev, void* stream);

/* Weighted per-element moments of a batch of wide rows (depth images) for the input normaliser (reference: RunningMeanStd.update,
 * running_mean_std.py:34-60, on the 'image' observation) - airgym_amd/csrc/cnn_kernels.hip.  partial_dev
 * [ag_weighted_moments_chunks()][2][D] (double) = (sum w x, sum w x^2) over each chunk's rows; the caller adds the chunks.  Row r is
 * x_dev[index_dev[r]] (NULL = r), w = weights_dev[r] (NULL = 1).  One pass over the rows. */
int ag_weighted_moments_chunks(void);
int ag_weighted_moments(const float* x_dev, const long long* index_dev, const float* weights_dev, long long rows, long long D,
                        double* partial_dev, void* stream);

/* The three stride-2 convolutions of the same feature extractor (reference: lib/network/cnn.py:11-13 - nn.Conv2d(1, 16, 5, 2, 2),
 * nn.Conv2d(16, 32, 3, 2, 1), nn.Conv2d(32, 64, 3, 2, 1) on (1, 212, 120) images; they replace torch's conv2d / MIOpen for exactly
 * these shapes) - airgym_amd/csrc/conv_kernels.hip.  All tensors NCHW float32 as torch holds them, weights [COUT][CIN][k][k], exact
 * float32 arithmetic (f32-input MFMA / fmaf).  n = images.  AG_ERR_UNSUPPORTED for any other shape.
 *   ag_cnn_conv1_fwd      : y [n,16,106,60] = conv(in, w [16,1,5,5]) + b with in = x [n,1,212,120], or, with norm_mean_dev /
 *                           norm_std_dev [212*120] given, in = clamp((x - mean) / std, -5, 5): the policy's image normaliser
 *                           (running_mean_std.py:78-79, per-pixel statistics) applied while the image is staged.
 *                           stats_dev (NULL = off) [n][16][2]: per image the sums of relu(y) and relu(y)^2 of every channel.
 *                           index_dev (NULL = identity) [n] int64: image i is row index[i] of x_dev - the minibatch's distinct frames
 *                           are read straight out of the rollout's frame store, not gathered into a tensor of their own first.
 *   ag_cnn_conv1_wgrad    : partials_dev [ag_cnn_conv1_wgrad_partials(n)][16][32]: columns 0-24 = dw[co][tap], column 25 = db[co]
 *                           (26-31 zero); the caller sums over dim 0 (fixed order -> deterministic).  x / norm_* as in the forward.
 *                           With bn_x_dev (the layer's own output x1 [n,16,106,60]) and bn_tab_dev [16][4] = {A, B, C, 0} per channel,
 *                           dz_dev is the gradient of the layer's ReLU + BatchNorm OUTPUT and the ReLU + BatchNorm backward is
 *                           folded in: dz = [x1 > 0] (A dy + m_i (B x1 + C)), m_i = weights_dev[i] (NULL = 1) - ag_relu_bn_bwd_dx's
 *                           arithmetic with A = gamma invstd, B = -A invstd dgamma / m, C = A (invstd mean dgamma - dbeta) / m.
 *   ag_cnn_conv_supported : 1 for (cin, cout, hin, win) in {(16, 32, 106, 60), (32, 64, 53, 30)} (3x3, stride 2, pad 1).
 *   ag_cnn_conv_fwd       : y [n,cout,ho,wo] = conv(in, w) + b with in = x, or, with scale_dev / shift_dev [cin] given,
 *                           in = relu(x) * scale[c] + shift[c] - the previous layer's ReLU + BatchNorm applied while the input is
 *                           staged, so that activation is never written to memory.
 *                           stats_dev (NULL = off) [n * ag_cnn_conv_fwd_bands(...)][cout][2]: per (image, band of output rows)
 *                           the sums of relu(y) and relu(y)^2 of every output channel - the following ReLU + BatchNorm's batch
 *                           statistics (sum over images and bands) and, for the last layer, the plane sums the global average pool
 *                           needs (sum over bands), with no extra pass over y.
 *   ag_cnn_conv_dgrad     : dx [n,cin,hin,win] = gradient of the layer's input (w.r.t. `in` above) from dz [n,cout,ho,wo].
 *   ag_cnn_conv_dgrad_bn  : (32, 64, 53, 30) only.  ag_cnn_conv_dgrad followed, in the kernel's epilogue, by the backward of the
 *                           nn.ReLU() + nn.BatchNorm2d in front of the layer (cnn.py:12): dx <- [bn_x > 0] (A g + m_i (B bn_x + C)),
 *                           g = the input gradient, bn_x = the previous convolution's output, bn_tab [cin][4] = {A, B, C, 0} as
 *                           ag_bn_bwd_prep (mode 1) writes it, weights [n] = image multiplicities m_i (NULL = 1).  The coefficients
 *                           are known before this kernel runs when they come from ag_bn_sums_from_conv.  sums_dev (NULL = off)
 *                           [ag_cnn_conv_dgrad_bn_rows(...)][cin][6]: per workgroup {total, row 0, last row, column 0, (0,0),
 *                           (last row, 0)} of dx - summed over the rows: the previous convolution's bias gradient and the border
 *                           sums ag_bn_sums_from_conv needs for the layer below.
 *   ag_cnn_conv_wgrad     : partials_dev [ag_cnn_conv_wgrad_partials(...)][cout*cin*9 + cout]: dw [cout][cin][3][3] then db [cout];
 *                           x / scale / shift as in ag_cnn_conv_fwd.  with_bias = 0: the db part is left unwritten (taken from
 *                           ag_relu_bn_bwd_dx*'s plane sums instead; the weight gradient alone is 10 - 16 % faster).
 * workspace_dev: ag_cnn_conv_workspace_floats(cin, cout) floats (the weights re-laid out for the kernel, rebuilt every call). */
int ag_cnn_conv_workspace_floats(int cin, int cout);
int ag_cnn_conv1_fwd(const float* x_dev, const long long* index_dev, const float* norm_mean_dev, const float* norm_std_dev,
                     const float* w_dev, const float* b_dev, float* y_dev, float* stats_dev, int n, float* workspace_dev,
                     void* stream);
int ag_cnn_conv1_wgrad_partials(int n);
int ag_cnn_conv1_wgrad(const float* dz_dev, const float* bn_x_dev, const float* bn_tab_dev, const float* weights_dev,
                       const float* x_dev, const long long* index_dev, const float* norm_mean_dev, const float* norm_std_dev,
                       float* partials_dev, int n, void* stream);
int ag_cnn_conv_supported(int cin, int cout, int hin, int win);
int ag_cnn_conv_fwd_bands(int cin, int cout, int hin, int win);
int ag_cnn_conv_fwd(const float* x_dev, const float* scale_dev, const float* shift_dev, const float* w_dev, const float* b_dev,
                    float* y_dev, float* stats_dev, int n, int cin, int cout, int hin, int win, float* workspace_dev, void* stream);
/* ag_cnn_conv_fwd on the BF16 matrix cores at float32 accuracy (round 6; conv_s2_fwd_split_kernel): every operand as an exact
 * three-way bf16 split, six bf16 MFMAs per product block - 6/16 of the f32-input MFMA's time, results within float32 rounding of
 * the f32 kernel's (not bit-identical: another summation order).  Same arguments and outputs; stats_dev has
 * ag_cnn_conv_fwd_split_bands(...) rows per image (bands of 4 output rows); workspace_dev (ag_cnn_conv_workspace_floats) 16-byte aligned. */
int ag_cnn_conv_fwd_split_bands(int cin, int cout, int hin, int win);
int ag_cnn_conv_fwd_split(const float* x_dev, const float* scale_dev, const float* shift_dev, const float* w_dev, const float* b_dev,
                          float* y_dev, float* stats_dev, int n, int cin, int cout, int hin, int win, float* workspace_dev,
                          void* stream);
int ag_cnn_conv_dgrad(const float* dz_dev, const float* w_dev, float* dx_dev, int n, int cin, int cout, int hin, int win,
                      float* workspace_dev, void* stream);
int ag_cnn_conv_dgrad_bn_rows(int n, int cin, int cout, int hin, int win);
int ag_cnn_conv_dgrad_bn(const float* dz_dev, const float* w_dev, const float* bn_x_dev, const float* bn_tab_dev,
                         const float* weights_dev, float* dx_dev, float* sums_dev, int n, int cin, int cout, int hin, int win,
                         float* workspace_dev, void* stream);
/*   ag_cnn_conv_dgrad_conv1_wgrad: ag_cnn_conv_dgrad(16, 32, 106, 60) + ag_cnn_conv1_wgrad(bn_x, bn_tab, ...) as ONE kernel: the second
 *                           convolution's input gradient is turned into the gradient of the first convolution's output in the
 *                           epilogue (as ag_cnn_conv_dgrad_bn) and consumed from LDS by the first convolution's weight gradient; the
 *                           1.9 GB tensor between the two never exists.  partials_dev [ag_cnn_conv_dgrad_conv1_wgrad_partials(n)][16][32]
 *                           (columns 0-24 dw [16][5][5], column 25 db; the caller sums the rows in order). */
int ag_cnn_conv_dgrad_conv1_wgrad_partials(int n);
int ag_cnn_conv_dgrad_conv1_wgrad(const float* dz_dev, const float* w_dev, const float* bn_x_dev, const float* bn_tab_dev,
                                  const float* weights_dev, const float* x_dev, const long long* index_dev, const float* norm_mean_dev,
                                  const float* norm_std_dev, float* partials_dev, int n, float* workspace_dev, void* stream);
int ag_cnn_conv_wgrad_partials(int n, int cin, int cout, int hin, int win);
int ag_cnn_conv_wgrad(const float* dz_dev, const float* x_dev, const float* scale_dev, const float* shift_dev,
                      float* partials_dev, int with_bias, int n, int cin, int cout, int hin, int win, void* stream);

/* Backward edges with the small weight gradients folded in.  Partials are per block of ag_wgrad_rows_per_block(which) rows
 * (ceil(M / rows) blocks); the caller reduces them over dim 0.
 *   ag_heads_bwd_elu_wgrad: dz = (d_heads Wh) * ELU'(h) (the head's dX formed inside the ELU' pass), plus dwh_partials_dev [blocks, A1, C] of dWh[a,c] = sum_m d_heads[m,a] h[m,c];
 *       db_partials_dev [blocks, C].
 *   ag_elu_bwd_input_wgrad: first layer; dz = dh * ELU'(h) is consumed on the fly and never stored:
 *       dw_partials_dev [blocks, C, D] of dW[c,d] = sum_m dz[m,c] x[m,d], db_partials_dev [blocks, C].
 *       D in {16, 18, 20}; blocks = ceil(M / ag_input_wgrad_rows(D)). */
int ag_wgrad_rows_per_block(int which);   /* which: 0 = ag_heads_bwd_elu_wgrad, 1 = ag_elu_bwd_input_wgrad (D <= 20) */
int ag_input_wgrad_rows(int D);           /* rows per block of ag_elu_bwd_input_wgrad for input width D; 0 = unsupported */
int ag_heads_bwd_elu_wgrad(const float* d_heads_dev, const float* Wh_dev, const float* h_dev, float* dz_dev,
                           float* db_partials_dev, float* dwh_partials_dev, int M, int C, int A1, int h_is_preactivation,
                           const float* zbias_dev, void* stream);
int ag_elu_bwd_input_wgrad(const float* dh_dev, const float* h_dev, const float* x_dev, float* dw_partials_dev,
                           float* db_partials_dev, int M, int C, int D, void* stream);

/* Every partial-sum reduction of one minibatch in two launches.  Job j: out_dev[0..n) = sum over `rows` rows of
 * partials_dev [rows, n] (n % 4 == 0, partials 16-byte aligned).  scratch_dev holds up to ag_sum_rows_groups() * sum_j n floats.
 * Fixed summation order (deterministic); at most AG_MAX_SUM_JOBS jobs per call. */
#define AG_MAX_SUM_JOBS 12
typedef struct ag_sum_job {
    const float* partials_dev;
    float* out_dev;
    int rows;
    int n;
} ag_sum_job;
int ag_sum_rows_groups(void);
int ag_sum_rows_multi(const ag_sum_job* jobs_host, int njobs, float* scratch_dev, long long scratch_floats, void* stream);
/* The same two launches with ag_ppo_loss_finalize's work (its arguments, in its order) done by one more workgroup of the first
 * launch: the end of `calc_gradients`' loss assembly (lib/agent/a2c_continuous.py:330-369) rides along with the reductions that
 * autograd's accumulation does for the weight gradients - one launch less per optimizer step.  Results bit-identical to the two
 * separate calls. */
int ag_sum_rows_multi_finalize(const ag_sum_job* jobs_host, int njobs, float* scratch_dev, long long scratch_floats,
                               const float* loss_partials_dev, int num_blocks, int M, int A, const float* logstd_dev,
                               float entropy_coef, float critic_coef, float bounds_loss_coef, float* grad_logstd_dev,
                               float* grad_head_bias_dev, float* kl_out_dev, float* stats_dev, void* stream);

/* ELU backward fused with the bias gradient of the producing Linear (lib/network/mlp.py:36-39 under autograd):
 * dz = dh * ELU'(z) computed from h = ELU(z); db_partials_dev [ceil(M / rows_per_block), C] per-block column sums of dz
 * (caller reduces).  C % 4 == 0 and 256 % (C/4) == 0 (C = 64, 128, 256, 512, 1024 ...). */
int ag_elu_bwd_bias_rows_per_block(void);
int ag_elu_bwd_bias(const float* dh_dev, const float* h_dev, float* dz_dev, float* db_partials_dev, int M, int C,
                    void* stream);

/* Clip-by-norm + Adam + KL-adaptive LR over a flat parameter buffer in one launch
 * (trancate_gradients_and_step, lib/agent/a2c_base.py:293-316; AdaptiveScheduler, lib/core/schedulers.py:19-32).
 * grad_dev has n + 1 elements: element n is the (already all-reduced) KL of this minibatch.
 * state_dev: ag_adam_state_bytes() bytes whose first double[2] = {learning rate, step count}, updated in place (the rest
 * is scratch).  max_grad_norm <= 0 disables clipping, kl_threshold <= 0 disables the LR adaptation. */
int ag_adam_state_bytes(void);

int ag_adam_clip_step(float* param_dev, float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, double* state_dev,
                      int n, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                      float kl_threshold, float min_lr, float max_lr, void* stream);

/* ---- Planning task (AG_TASK_PLANNING) -------------------------------------------------------------------
 * Replaces, on top of the entries above: Planning.reset_idx / step / compute_observations /
 * compute_quadcopter_reward (airgym/envs/task/planning.py:63-307), Customized.pre_physics_step, check_collisions,
 * render_cameras + dump_images (airgym/envs/base/customized.py:216-298,386-435) and the IsaacGym camera /
 * contact-force tensors behind them (customized.py:51-55,138,387-390).
 * ag_step* on a planning handle = physics kernel, then (every 4th step, cam_dt/dt) the render kernel, then the
 * observation/reward/reset kernel.  Before the first ag_reset_all/ag_step the obstacle variant table must be set. */
#define AG_PLANNING_NUM_OBSTACLES 40
#define AG_PLANNING_CAM_W 212
#define AG_PLANNING_CAM_H 120
#define AG_PLANNING_RESET_UNIFORMS 121

typedef struct ag_planning_buffers {
    float* image_dev;       /* [num_envs, 1, 212, 120] f32 == full_camera_array (customized.py:144) */
    float* collisions_dev;  /* [num_envs] f32 0/1 (customized.py:141,393-397) */
} ag_planning_buffers;

typedef struct ag_planning_state_view {   /* all DEVICE pointers, any may be NULL */
    float* obstacles_dev;   /* planning only: [num_envs, 40, 4]: root x, y, yaw, variant index (as float) */
    float* goal_dev;        /* [num_envs, 3]: planning goal / balloon position (balloon.py:37-39) / cube position (avoid.py:38-40) */
    float* extra_dev;       /* [num_envs, 5]: pre_root_positions xyz, esdf_dist, prev_related_dist */
    float* object_vel_dev;  /* avoid only: [num_envs, 3] linear velocity of the thrown cube (avoid.py:41) */
} ag_planning_state_view;

/* table_host: [n_variants <= 100, 8] = centre xyz, unit axis xyz, radius, half length of each obstacle variant in
 * its own frame (airgym_amd/assets/thin_trees.json).  Host pointer; copied. */
int ag_planning_set_obstacle_table(ag_handle h, const float* table_host, int n_variants);
int ag_planning_get_buffers(ag_handle h, ag_planning_buffers* out);
int ag_planning_get_state(ag_handle h, const ag_planning_state_view* view, void* stream);
int ag_planning_set_state(ag_handle h, const ag_planning_state_view* view, void* stream);
/* The ag_planning_* entry points below also serve Balloon and Avoid handles (the same Customized family).
 * Parity mode: per-env reset uniforms supplied by the caller (NULL = counter RNG): planning [num_envs, 121];
 * balloon [num_envs, 15] = balloon xyz | root xy | root z | euler xyz | linvel | angvel (balloon.py:57-99);
 * avoid [num_envs, 11] = throw mask | theta | aim xyz | root xy | root z | euler xy | euler z (avoid.py:58-163). */
int ag_planning_step_with_uniforms(ag_handle h, const float* actions_dev, const float* reset_uniforms_dev, void* stream);
/* Parity / inspection mode: the post-physics half of Planning.step (planning.py:158-183: progress++, compute_observations,
 * compute_quadcopter_reward, reset of done envs) on the handle's CURRENT state, with the collision flags
 * (Customized.check_collisions, customized.py:393-397) supplied by the caller [num_envs] f32 0/1 instead of the geometric
 * test - how the vectors recorded from the reference's own Planning methods are replayed on the HIP kernel.  actions_dev is
 * the RAW action (the thrust channel is remapped in rate mode exactly as in a step).  noise_dev: balloon handles only,
 * [num_envs, 18] standard normals of Customized.add_noise (customized.py:450-459) or NULL (counter RNG / noise off). */
int ag_planning_eval_post(ag_handle h, const float* actions_dev, const float* collisions_dev, const float* noise_dev,
                          void* stream);
/* Render the camera on the NEXT step regardless of the every-4th-step schedule (planning.py:153-156). */
int ag_planning_render_now(ag_handle h);
/* 1 if the most recent step (or reset) wrote a new depth image, 0 if the image buffer still holds the previous one (the camera
 * runs every 4th step), -1 for a handle without a camera.  Lets a caller that derives features from the image with a FROZEN
 * encoder (lib/network/vae_image_encoder.py:34-53) re-encode only when the image changed. */
int ag_planning_last_step_rendered(ag_handle h);

/* Benchmark / diagnostic knobs are not part of this interface nor of the shipped library: see airgym_hip_debug.h
 * (experiments build, `python airgym_amd/csrc/build.py --experiments`). */

#ifdef __cplusplus
}
#endif
#endif /* AIRGYM_HIP_H */
