/*
 * msk_physx.h — C ABI of the MI355X-native batched rigid-body backend.
 *
 * This is the drop-in boundary for the ManiSkill hot path `PhysxGpuSystem.step()`
 * (reference: mani_skill/envs/scene.py:379-380) and the buffer / apply / fetch /
 * contact-query contract around it (mani_skill/envs/scene.py:902-986, 741-801;
 * mani_skill/utils/structs/base.py:103-144; structs/articulation.py:723-921).
 *
 * The reference reaches this functionality through the `sapien.physx` Python module
 * (pybind11 over PhysX 5, not in the reference tree).  Every entry point below names
 * the `sapien.physx` call it stands in for.  Signatures use only plain pointers and
 * sizes; device pointers returned by msk_buffer() are wrapped zero-copy by the host
 * language (torch.from_blob-style).
 *
 * One msk_ctx == one `PhysxGpuSystem` == all sub-scenes ("envs") of one process/GPU.
 * All envs share ONE template (same bodies / shapes / joints); state differs per env.
 * Thread model: single host thread (as in the reference); kernels are enqueued on the
 * hipStream_t handed to each call (torch's current stream).
 *
 * Return convention: int functions return >= 0 on success (often an id) and a negative
 * msk_status on failure; msk_last_error() gives the text.
 */
#ifndef MSK_PHYSX_H
#define MSK_PHYSX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msk_ctx msk_ctx;

enum msk_status {
  MSK_OK = 0,
  MSK_ERR_INVALID = -1,   /* bad argument / wrong phase (e.g. add_* after finalize)   */
  MSK_ERR_CAPACITY = -2,  /* template exceeds a compile-time capacity (MSK_MAX_*)     */
  MSK_ERR_HIP = -3,       /* a HIP runtime call failed                                */
  MSK_ERR_OVERFLOW = -4   /* reserved: step() never fails on contact overflow, it drops
                           * the contacts past the per-env capacity in (pair, point)
                           * order and raises the sticky flag msk_get_sizes()[7]       */
};

enum msk_joint_type { MSK_JOINT_FIXED = 0, MSK_JOINT_REVOLUTE = 1, MSK_JOINT_PRISMATIC = 2 };
enum msk_body_kind { MSK_BODY_KINEMATIC = 1, MSK_BODY_DYNAMIC = 2, MSK_BODY_LINK = 3 };
enum msk_shape_type { MSK_SHAPE_PLANE = 0, MSK_SHAPE_BOX = 1, MSK_SHAPE_SPHERE = 2, MSK_SHAPE_CONVEX = 3, MSK_SHAPE_CAPSULE = 4,
                      MSK_SHAPE_CYLINDER = 5 };

/* Capacities of one env template (compile-time, shared by oracle and HIP library). */
#define MSK_MAX_BODIES 64
#define MSK_MAX_SHAPES 64
#define MSK_MAX_DOF 64        /* articulation DoF per env (all articulations, six per floating root included): <= MSK_MAX_DOF - 1 */
#define MSK_MAX_NV 64         /* generalized velocity size: art DoF + 6 per free body (the solver runs in 16-, 32- and 64-coordinate forms) */
#define MSK_MAX_PAIRS 2048    /* candidate shape pairs after static filtering         */
#define MSK_MAX_CONTACTS 48   /* contact points per env per step (msk_config.contact_capacity = 0) */
#define MSK_MAX_BLOCKS 64     /* ... and solver blocks per env: a joint with a drive / limit row, a joint with friction, a contact point, a torsional row */
#define MSK_MAX_CONTACTS_WIDE 128   /* msk_config.contact_capacity = 1: the wide solver class (two blocks per lane) behind the others */
#define MSK_MAX_BLOCKS_WIDE 128
#define MSK_MAX_HULL_VERTS 64
#define MSK_MAX_TENDONS 4

/* Scene / solver configuration.  Mirrors what ManiSkill pushes through
 * physx.set_scene_config / set_body_config / set_shape_config / PhysxGpuSystem.timestep
 * (mani_skill/envs/sapien_env.py:1173-1180,1227; structs/types.py:35-90). */
typedef struct msk_config {
  float timestep;             /* 1 / sim_freq                                          */
  float gravity[3];
  int32_t solver_position_iterations;
  int32_t solver_velocity_iterations;
  float contact_offset;       /* per shape; a pair generates contacts below offA+offB  */
  float rest_offset;
  float bounce_threshold;     /* a normal row approaching faster than this rebounds with the pair's restitution */
  float sleep_threshold;      /* accepted, NOT modelled (bodies never sleep): noted in msk_warnings()          */
  int32_t enable_tgs;         /* must be 1                                             */
  int32_t enable_pcm;         /* accepted, no effect (manifolds are generated one-shot every step): 0 is noted in msk_warnings() */
  int32_t contact_capacity;   /* 0: MSK_MAX_CONTACTS points / MSK_MAX_BLOCKS solver blocks per env, points past them are dropped in (pair, point)
                               * order and flagged (msk_get_sizes out[7]); 1: MSK_MAX_CONTACTS_WIDE / MSK_MAX_BLOCKS_WIDE (PhysX sizes its contact
                               * buffers for the whole scene: mani_skill/utils/structs/types.py:18-23) */
  int32_t reserved[5];
} msk_config;

/* ---- lifetime -------------------------------------------------------------------- */
/* sapien.physx.PhysxGpuSystem(device)  (sapien_env.py:1187) */
msk_ctx* msk_create(int hip_device, const msk_config* cfg);
/* scene teardown (sapien_env.py:1232-1243) */
void msk_destroy(msk_ctx* ctx);
const char* msk_last_error(msk_ctx* ctx);
/* Parameters this backend accepted without modelling them, one per line ("" if none): nothing handed over the ABI is dropped
 * silently.  Today: sleep_threshold > 0 (SAPIEN puts bodies slower than it to sleep; here every body is simulated every step) and
 * enable_pcm = 0.  Everything else the ABI takes -- static / dynamic friction, restitution, patch radii, joint friction and
 * armature, drive modes and force limits, dampings -- is part of the simulation. */
const char* msk_warnings(msk_ctx* ctx);

/* ---- template build phase (host only; before msk_finalize) ------------------------ */
/* PhysxArticulation creation via ArticulationBuilder.build (building/articulation_builder.py:114).
 * root_pose = [px py pz qw qx qy qz]; fixed base.  Returns articulation index. */
int msk_add_articulation(msk_ctx* ctx, const float root_pose[7]);
/* PhysxArticulationLinkComponent(parent) + joint record (articulation_builder.py:65-112).
 * parent_body = -1 for the root link.  Joint axis is +x of the joint frame (SAPIEN
 * convention): child pose = parent pose * pose_in_parent * J_x(q) * inv(pose_in_child).
 * inertia6 = [ixx iyy izz ixy ixz iyz] about the COM, in link axes.  Returns body id. */
int msk_add_link(msk_ctx* ctx, int art, int parent_body, int joint_type,
                 const float pose_in_parent[7], const float pose_in_child[7],
                 float limit_lo, float limit_hi, float mass, const float com[3],
                 const float inertia6[6], int disable_gravity, float armature,
                 float joint_friction);
/* fix_root_link = False (utils/building/articulation_builder.py:212; agents with a free base: mani_skill/agents/robots/anymal, unitree_*,
 * the MJCF ant / humanoid of envs/tasks/control): the articulation's root link is not held by the world.  It gets six coordinates of
 * its own -- those of a free body: velocity of its centre of mass, then angular velocity -- that enter the joint-space inertia, the
 * solver and the integration like joint coordinates do; qpos / qvel keep SAPIEN's layout (joints only), the root's pose
 * and velocity travel in its row of rigid_body_data (msk_apply with MSK_APPLY_ART_ROOT_POSE / MSK_APPLY_ART_ROOT_VELOCITY).
 * Call before msk_finalize; joint dofs + 6 per floating root <= MSK_MAX_DOF - 1. */
int msk_set_articulation_floating(msk_ctx* ctx, int articulation);
/* PhysxArticulationJoint.set_drive_properties (agents/controllers/pd_joint_pos.py:38-52).  Also after msk_finalize (BaseAgent.set_control_mode
 * re-programs the drives of a running simulation: agents/base_agent.py, examples/benchmarking/gpu_sim.py): the drive belongs to the template,
 * every sub-scene takes the new values; synchronises the device when something changes. */
int msk_set_drive(msk_ctx* ctx, int link_body, float stiffness, float damping,
                  float force_limit, int mode_acceleration);
/* PhysxArticulation.create_fixed_tendon for URDF mimic joints
 * (articulation_builder.py:161-200): soft constraint
 * coef_a*q(link_a) + coef_b*q(link_b) = rest, with the given stiffness/damping. */
int msk_add_tendon(msk_ctx* ctx, int link_a, int link_b, float coef_a, float coef_b,
                   float rest, float stiffness, float damping);
/* PhysxRigidDynamicComponent / kinematic actors (building/actor_builder.py:193-261).
 * Returns body id. */
int msk_add_actor(msk_ctx* ctx, int kind, const float pose[7], float mass,
                  const float com[3], const float inertia6[6], float linear_damping,
                  float angular_damping, int disable_gravity);
/* PhysxRigidDynamicComponent.set_locked_motion_axes([lin x, y, z, ang x, y, z]) (utils/structs/base.py:340-354, 455-468): bit k of `mask`
 * locks world axis k of the dynamic actor `body` -- the body keeps no velocity along / about it, constraints find infinite mass there
 * (the corresponding rows and columns of its inverse mass matrix are zero).  Before msk_finalize. */
int msk_set_locked_axes(msk_ctx* ctx, int body, uint32_t mask);
/* PhysxCollisionShape* + body.attach(shape) (actor_builder.py:57-164).  body = -1
 * attaches to the static world (PhysxRigidStaticComponent).  params: box half sizes /
 * sphere {radius} / capsule and cylinder {radius, half length} (axis = +x of the shape frame, as in
 * SAPIEN / PhysX) / convex {rounding radius, normally 0}.  verts (convex only): nverts*3 floats in
 * shape-local coordinates, scale already applied, nverts <= MSK_MAX_HULL_VERTS.  Plane normal is
 * +x of the shape frame (SAPIEN convention, building/ground.py:38-40).
 * Internally every non-box, non-plane shape is a "rounded hull": a convex vertex set swept by a ball.
 * Sphere = 1 vertex + radius, capsule = 2 vertices + radius (exact); a cylinder is cooked to a 16-sided
 * prism hull (PhysX has no cylinder primitive either: SAPIEN hands it a cooked convex mesh [ext]). */
int msk_add_shape(msk_ctx* ctx, int body, int type, const float local_pose[7],
                  const float params[3], const float* verts, int nverts,
                  float static_friction, float dynamic_friction, float restitution,
                  const uint32_t groups[4], float patch_radius, float min_patch_radius);
/* SRDF <disable_collisions> (panda_v2.srdf) */
int msk_disable_collision(msk_ctx* ctx, int body_a, int body_b);
/* PhysxGpuSystem.gpu_init() (envs/scene.py:910): freeze the template, replicate it for
 * num_envs sub-scenes, allocate and upload everything. */
int msk_finalize(msk_ctx* ctx, int num_envs);
/* PhysxGpuSystem.set_scene_offset (sapien_env.py:1202): offsets = num_envs*3 host floats,
 * added to positions on fetch and subtracted on apply. */
int msk_set_scene_offsets(msk_ctx* ctx, const float* offsets);

/* ---- buffers (device pointers, valid until msk_destroy) --------------------------- */
enum msk_buffer_id {
  MSK_BUF_RIGID_BODY_DATA = 0, /* px.cuda_rigid_body_data: (num_envs*NB, 13) f32
                                  [p(3) q(wxyz) v(3) w(3)], row = env*NB + body id       */
  MSK_BUF_ART_QPOS = 1,        /* px.cuda_articulation_qpos: (num_envs*NA, max_dof) f32 */
  MSK_BUF_ART_QVEL = 2,
  MSK_BUF_ART_QACC = 3,
  MSK_BUF_ART_QF = 4,
  MSK_BUF_ART_TARGET_QPOS = 5, /* px.cuda_articulation_target_qpos                      */
  MSK_BUF_ART_TARGET_QVEL = 6,
  MSK_BUF_RIGID_BODY_FORCE = 7,  /* px.cuda_rigid_body_force (structs/actor.py:316-322): (num_envs*NB, 4) f32
                                  * [fx fy fz -], world frame, at the centre of mass; same rows as buffer 0 */
  MSK_BUF_RIGID_BODY_TORQUE = 8, /* px.cuda_rigid_body_torque: (num_envs*NB, 4) f32 [tx ty tz -]            */
  MSK_BUF_ART_LINK_JOINT_FORCES = 9, /* px.cuda_articulation_link_incoming_joint_forces (structs/articulation.py:596-620):
                                  * (num_envs*NA*max_links, 6) f32, row = (env*NA + articulation)*max_links + link index
                                  * (links of an articulation in build order), [fx fy fz tx ty tz]: the wrench the parent
                                  * transmits through the link's inbound joint, at the origin and in the axes of the
                                  * joint's child frame (x = joint axis); root link: the wrench that holds the fixed base,
                                  * in the root link's frame.  msk_get_sizes()[7] is not involved; max_links = shape[0] /
                                  * (num_envs*NA). */
  MSK_BUF_COUNT = 10
};
void* msk_buffer(msk_ctx* ctx, int buffer_id, int64_t shape[2]);

/* ---- apply / fetch / step (async on `stream`; a hipStream_t passed as void*) ------- */
enum msk_apply_mask {
  MSK_APPLY_RIGID_DATA = 1 << 0,   /* gpu_apply_rigid_dynamic_data                       */
  MSK_APPLY_ART_QPOS = 1 << 1,     /* gpu_apply_articulation_qpos                        */
  MSK_APPLY_ART_QVEL = 1 << 2,     /* gpu_apply_articulation_qvel                        */
  MSK_APPLY_ART_QF = 1 << 3,       /* gpu_apply_articulation_qf                          */
  MSK_APPLY_ART_TARGET_QPOS = 1 << 4, /* gpu_apply_articulation_target_position          */
  MSK_APPLY_ART_TARGET_QVEL = 1 << 5, /* gpu_apply_articulation_target_velocity          */
  MSK_APPLY_ART_ROOT_POSE = 1 << 6,   /* gpu_apply_articulation_root_pose (root link row) */
  /* gpu_apply_rigid_dynamic_force / _torque (Actor.apply_force, structs/actor.py:316-322): the buffer's rows of the
   * dynamic actors and of the articulation links (force at the centre of mass, world frame) become the external wrench of
   * the NEXT msk_step only (PhysX clears forces after every simulate()); a second apply before that step replaces the
   * rows, it does not add.  Rows of kinematic actors are ignored. */
  MSK_APPLY_RIGID_FORCE = 1 << 7,
  MSK_APPLY_RIGID_TORQUE = 1 << 8,
  MSK_APPLY_ART_ROOT_VELOCITY = 1 << 9 /* gpu_apply_articulation_root_velocity: linear (centre of mass) and angular velocity of a floating root,
                                        * columns 7..12 of its rigid_body_data row */
};
enum msk_fetch_mask {
  MSK_FETCH_RIGID_DATA = 1 << 0,   /* gpu_fetch_rigid_dynamic_data + link_pose/velocity  */
  MSK_FETCH_ART_QPOS = 1 << 1,
  MSK_FETCH_ART_QVEL = 1 << 2,
  MSK_FETCH_ART_QACC = 1 << 3,
  MSK_FETCH_ART_TARGETS = 1 << 4,
  MSK_FETCH_ART_LINK_FORCES = 1 << 5 /* gpu_fetch_articulation_link_incoming_joint_forces: inverse dynamics of the state the
                                      * last msk_step left (qacc, contact impulses / dt); not part of gpu_fetch_all        */
};
int msk_apply(msk_ctx* ctx, uint32_t mask, void* stream);
int msk_fetch(msk_ctx* ctx, uint32_t mask, void* stream);

/* ---- partial reset without the host ---------------------------------------------------------------------------------------
 * ManiSkillVectorEnv.step resets the sub-scenes that finished inside the same step (vector/wrappers/gymnasium.py:164-176), and once episodes drift out of
 * phase that is almost every step.  The reference's reset is host work: BaseEnv.reset -> _clear_sim_state, _initialize_episode (numpy / torch
 * RNG), controller.reset, masked writes through the torch views, _gpu_apply_all, kinematics, _gpu_fetch_all (envs/sapien_env.py:857-978,1023-1036;
 * utils/scene_builder/table/scene_builder.py:67-103).  msk_reset_masked is that apply for the envs a DEVICE-side mask names, from episode images prepared
 * ahead of time: nothing on the step path waits for the host.
 *   mask      device [num_envs] u8: the envs to reset (the step's `terminated | truncated`)
 *   image     device [num_envs][slots][nent] f32: per env a ring of prepared episodes; episode k of an env lives in slot k % slots
 *   ent       device [nent] i32: where entry t of an image goes -- (buffer << 24) | word, buffer 0 = rigid_body_data (word = body * 13 + column),
 *             1 = qpos, 2 = qvel, 3 = target_qpos, 4 = target_qvel (word = articulation * pitch + coordinate): the sapien buffers' own layout per env
 *   episode   device [num_envs] i32, read-modify-write: the env's next episode number; a reset consumes image slot episode % slots and adds one
 *   elapsed   device [num_envs] i32 or NULL: the env's step counter, zeroed by a reset
 * Per named env: msk_fetch's rows of that env (so that entries the image does not name keep their values), the image's entries over them, then
 * msk_apply's work for that env with MSK_APPLY_RIGID_DATA | ART_ROOT_POSE | ART_QPOS | ART_QVEL | ART_QF | ART_TARGET_QPOS | ART_TARGET_QVEL -- the same
 * comparisons, the same normalisation, the contact cache of a teleported env dropped: bit for bit what the host-side path leaves.  Envs the mask does not
 * name are not touched at all.  Link frames are stale afterwards as after msk_step (the next fetch / observe refreshes them).  Asynchronous on `stream`. */
enum { MSK_RESET_RIGID_BODY_DATA = 0, MSK_RESET_ART_QPOS = 1, MSK_RESET_ART_QVEL = 2, MSK_RESET_ART_TARGET_QPOS = 3, MSK_RESET_ART_TARGET_QVEL = 4 };
int msk_reset_masked(msk_ctx* ctx, const uint8_t* mask, const float* image, int slots, const int32_t* ent, int nent, int32_t* episode, int32_t* elapsed,
                     void* stream);
/* ---- the vector wrapper's episode book-keeping in one launch --------------------------------------------------------------
 * ManiSkillVectorEnv.step (vector/wrappers/gymnasium.py:127-176) keeps, per sub-scene, the running return and the success_once / fail_once flags of the
 * episode in progress, reports them (`infos["episode"]`: return, episode_len, reward = return / episode_len, success_once, fail_once, with
 * ignore_terminations also success_at_end / fail_at_end), clears the termination flags when it ignores terminations, forms `dones = terminated | truncated`,
 * asks `dones.any()` and clears the book of the envs it then resets (:104-125) -- a dozen elementwise torch launches per step.  msk_episode_book_step is that
 * sequence for all sub-scenes: same values, same order (accumulate, report, clear).  Plain device pointers, [n] each; the u8 inputs come with the number of bytes
 * between consecutive envs (the fused task kernels hand their flags out as columns of one [n][6] block).  Asynchronous on `stream`. */
typedef struct msk_episode_book {
  const float* reward;             /* the step's reward */
  const int32_t* elapsed;          /* BaseEnv.elapsed_steps after the step (episode_len) */
  const uint8_t* success;          /* infos["success"], NULL when the task reports none */
  const uint8_t* fail;             /* infos["fail"], NULL when the task reports none */
  const uint8_t* terminated;
  const uint8_t* truncated;
  int32_t success_stride, fail_stride, terminated_stride, truncated_stride;   /* bytes between consecutive envs (1 = a plain [n] array) */
  int32_t record_metrics;          /* 0: only out_terminated / out_done / any_done are produced */
  int32_t ignore_terminations;     /* out_terminated = 0 everywhere; success / fail of the step are reported as *_at_end */
  int32_t clear_done;              /* auto_reset: the book of the envs that are done is cleared after it was reported (the reset that follows does that) */
  float* returns;                  /* state, read-modify-write (record_metrics) */
  uint8_t* success_once;
  uint8_t* fail_once;
  float* out_return;               /* outputs (record_metrics; the *_once / *_at_end ones only where the input exists) */
  int32_t* out_episode_len;
  float* out_reward;
  uint8_t* out_success_once;
  uint8_t* out_fail_once;
  uint8_t* out_success_at_end;
  uint8_t* out_fail_at_end;
  uint8_t* out_terminated;         /* always */
  uint8_t* out_done;               /* always: out_terminated | truncated */
  int32_t* any_done;               /* always, [1]: 1 when some env is done, else 0 (`dones.any()`: the one value the host reads back) */
} msk_episode_book;
int msk_episode_book_step(msk_ctx* ctx, int n, const msk_episode_book* book, void* stream);
/* PhysxGpuSystem.gpu_update_articulation_kinematics (sapien_env.py:959,1304) */
int msk_update_kinematics(msk_ctx* ctx, void* stream);
/* PhysxGpuSystem.step() (envs/scene.py:379-380): one substep of `timestep` for all envs. */
int msk_step(msk_ctx* ctx, void* stream);
/* `count` consecutive msk_step calls with nothing in between -- the reference's `for _ in range(sim_freq // control_freq): px.step()`
 * (envs/scene.py:379-380, sapien_env.py:1073-1086) when no controller acts between the substeps.  Same results as the loop, env for env.
 * The envs of a context never interact, so the library may run the substeps of contiguous env PARTITIONS (whole 64-env chunks; at most
 * MSK_STEP_PARTS_MAX, msk_get_step_parts() says how many: 1 unless msk_set_step_parts or the environment variable MSK_STEP_PARTS ask for
 * more -- measured slower on MI355X for the benchmarked tasks, see msk_physx.hip step_part_count) as
 * independent kernel chains on streams of its own, forked from and joined to `stream`: with msk_step_n the chains run `count` substeps
 * between one fork and one join (inside a captured graph: parallel branches), with msk_step one. */
#define MSK_STEP_PARTS_MAX 8
int msk_step_n(msk_ctx* ctx, int count, void* stream);
int msk_get_step_parts(msk_ctx* ctx);
/* re-partitions a finalized context (tuning and tests; synchronises the device): -> the partition count in force (whole 64-env chunks:
 * a request that does not divide the env set that way is lowered), < 0 on error */
int msk_set_step_parts(msk_ctx* ctx, int parts);

/* ---- several contexts behind one set of sapien tensors ------------------------------------------------------------ */
/* ManiSkill sees ONE px.cuda_rigid_body_data / px.cuda_articulation_* per process, over all sub-scenes (utils/structs/actor.py:352,
 * articulation.py:726-797), also when the sub-scenes differ in structure (a different cabinet per sub-scene:
 * envs/tasks/mobile_manipulation/open_cabinet_drawer.py:128-177) and therefore run as one context per structural group.
 * msk_bind_buffers: the caller owns the storage of the nine apply / fetch buffers (ids MSK_BUF_RIGID_BODY_DATA ..
 * MSK_BUF_RIGID_BODY_TORQUE, ptrs[id]) and hands each context ITS ROW RANGE of the shared tensors: no copies between a context's
 * buffers and the tensor ManiSkill reads.  Articulation rows are `art_pitch` floats apart (>= the context's max_dof: SAPIEN pads
 * every articulation to the largest one of the scene).  Contents: whatever the memory holds (fetch to fill the outputs; qf / force
 * / torque are inputs).  msk_buffer reports the bound pointers afterwards, shape[1] = art_pitch for the articulation buffers.
 * The memory must outlive the context. */
int msk_bind_buffers(msk_ctx* ctx, void* const ptrs[9], int64_t art_pitch);
/* The same boundary call on n contexts of one device, in one native call: op 0 = msk_step, 1 = msk_apply(mask), 2 = msk_fetch(mask),
 * 3 = msk_update_kinematics.  The contexts are independent, so their kernels are issued round-robin on internal streams forked
 * from and joined into `stream` (what a host loop over the contexts with per-call stream switches would do, without its cost:
 * 25 groups x 28 calls per control step were 23 ms of Python in OpenCabinetDrawer-v1). */
enum msk_batch_op { MSK_BATCH_STEP = 0, MSK_BATCH_APPLY = 1, MSK_BATCH_FETCH = 2, MSK_BATCH_UPDATE_KINEMATICS = 3 };
int msk_batch(msk_ctx* const* ctxs, int n, int op, uint32_t mask, void* stream);

/* ---- per-env instances of the template (heterogeneous sub-scenes) -------------------------------------------- */
/* ManiSkill builds some tasks with a different actor per sub-scene and merges them (Actor.merge: e.g. PegInsertionSide-v1's
 * peg and box-with-hole, one size per env, envs/tasks/tabletop/peg_insertion_side.py:133-187).  Here the template stays one
 * description; a box shape / a dynamic actor can be declared to take its numbers from the env's own record instead:
 *   msk_declare_env_box(shape)   before msk_finalize: half sizes and the position part of the local pose vary per env
 *   msk_declare_env_mass(body)   before msk_finalize: mass and principal inertia vary per env (the body's centre of mass must be
 *                                its origin and its inertia diagonal)
 *   msk_set_env_boxes(shape, half_sizes[num_envs][3], local_pos[num_envs][3] or NULL)      after msk_finalize
 *   msk_set_env_masses(body, mass[num_envs], principal_inertia[num_envs][3])               after msk_finalize
 * Until set, every env uses the template's values.  Collision filtering and the candidate-pair table stay those of the template. */
int msk_declare_env_box(msk_ctx* ctx, int shape);
int msk_declare_env_mass(msk_ctx* ctx, int body);
int msk_set_env_boxes(msk_ctx* ctx, int shape, const float* half_sizes, const float* local_pos);
int msk_set_env_masses(msk_ctx* ctx, int body, const float* mass, const float* principal_inertia);

/* ---- contact impulse queries (envs/scene.py:771-781) ------------------------------ */
/* gpu_create_contact_pair_impulse_query(body_pairs): body ids are template body ids, the
 * same pair is queried in every env.  Output buffer (num_envs*npairs, 3) f32, row =
 * env*npairs + pair; value = sum of contact impulses of the last step() applied ON
 * body_a BY body_b.  Returns query id. */
int msk_query_create_pairs(msk_ctx* ctx, const int32_t* body_pairs, int npairs);
/* gpu_create_contact_body_impulse_query(bodies) (utils/structs/base.py:116-136, articulation.py:447-462): the net contact
 * impulse of the last step() on each listed body, from every other body and the static scene.  Same buffer / run calls;
 * output (num_envs*nbodies, 3), row = env*nbodies + i.  (A pair query whose second id is MSK_ANY_BODY.) */
#define MSK_ANY_BODY (-2)
int msk_query_create_bodies(msk_ctx* ctx, const int32_t* bodies, int nbodies);
void* msk_query_buffer(msk_ctx* ctx, int query, int64_t shape[2]);
/* gpu_query_contact_pair_impulses(query) */
int msk_query_run(msk_ctx* ctx, int query, void* stream);

/* ---- inspection (parity tests; synchronous, host output) --------------------------- */
/* Template sizes: out[0]=NB bodies, out[1]=NA articulations, out[2]=max_dof, out[3]=nv,
 * out[4]=number of shapes, out[5]=number of candidate pairs, out[6]=num_envs,
 * out[7]: sticky flags since init -- bit 0: an env exceeded its contact capacity; bits 1, 2: solver scheduling errors (never set in a
 * correct build); bit 3: a camera's record / list capacities ran over (its pictures may miss triangles). */
int msk_get_sizes(msk_ctx* ctx, int32_t out[8]);
/* Contacts generated by the last step() in env `env`: for each contact point
 * ids[3*i..] = {shape_a, shape_b, body-pair slot}, vals[8*i..] = {pos(3), normal(3), separation,
 * normal impulse}.  Returns the number of points (<= max_points written). */
int msk_get_contacts(msk_ctx* ctx, int env, int32_t* ids, float* vals, int max_points);

/* Number of contact points each env solved in the last step(): out[num_envs]. */
int msk_get_env_contact_counts(msk_ctx* ctx, int32_t* out);

/* Solver scheduling (no counterpart in the reference; results do not depend on it).  Envs are sorted by their number
 * of constraint blocks (joints near a limit + contact points) into four capacity classes: class 0 is solved four
 * (two for > 16 coordinates) envs to a wavefront, classes 1..3 one env per wavefront.  caps[k] = largest block count
 * of class k (k = 0..2; class 3 takes the rest, up to 64); caps[0] may not exceed what the packed LDS pool holds and
 * caps[2] may not exceed 32 (both are clamped).  A negative caps[0..2] empties that class: {-1,-1,-1} sends every env
 * through class 3.  msk_get_solver_class_counts returns the five list lengths of the last step() (the fifth: the wide class of
 * msk_config.contact_capacity = 1, envs of more than MSK_MAX_BLOCKS blocks). */
int msk_set_solver_classes(msk_ctx* ctx, const int32_t caps[3]);
int msk_get_solver_class_counts(msk_ctx* ctx, int32_t out[5]);

/* ---- measurement (bench.py: roofline.achieved) --------------------------------------- */
/* Per-kernel HIP-event timing of msk_step(), on the stream the kernels are launched on.
 * The reference's harness only has wall-clock (examples/benchmarking/profiling.py:96-113);
 * this is the per-kernel counterpart.  msk_timing_enable(ctx, max_steps) arms the timer for
 * the next max_steps calls of msk_step (0 disarms and drops the samples): an armed step launches
 * its kernels with a (begin, end) event pair each (hipExtLaunchKernelGGL: the time stamps of the
 * dispatch itself, the duration rocprofv3's kernel trace reports -- no dispatch gap inside).
 * msk_timing_read waits for the recorded events and returns, for kernel slot `slot`
 * (enum msk_kernel_slot), the summed duration in milliseconds and the number of launches:
 * MSK_K_DYNAMICS = k_dynamics (joint-space dynamics + broadphase), MSK_K_COLLIDE = k_narrowphase
 * (k_classify for a scene without candidate pairs), MSK_K_SOLVE = k_csolve (the TGS solver),
 * MSK_K_SUBSTEP = begin of the first to end of the last of the three (launch gaps included). */
enum msk_kernel_slot { MSK_K_DYNAMICS = 0, MSK_K_COLLIDE = 1, MSK_K_SOLVE = 2, MSK_K_KERNELS = 3, MSK_K_SUBSTEP = 3, MSK_K_SLOTS = 4 };
int msk_timing_enable(msk_ctx* ctx, int max_steps);
int msk_timing_read(msk_ctx* ctx, int slot, double* total_ms, int32_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* MSK_PHYSX_H */
