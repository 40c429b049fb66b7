/*
 * msk_physx.hip — C-ABI host side of the MI355X-native rigid-body backend (include/msk_physx.h).
 *
 * Host code only records the env template, uploads it once, owns the device arrays and
 * enqueues the kernels of msk_kernels.h on the caller's HIP stream.  No CPU physics here:
 * every entry point that computes anything launches a kernel.
 */
#include "msk_kernels.h"
#include "msk_task.h"
#include "msk_render.h"

#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define MSK_API extern "C" __attribute__((visibility("default")))

/* a launch whose own begin / end time stamps (the dispatch's completion signal: what rocprofv3's kernel trace reports) land in two events */
#define LAUNCH_TIMED(ev, kern, grid, block, lds, s, ...)                                                       \
  do {                                                                                                         \
    if (ev) hipExtLaunchKernelGGL(kern, grid, block, lds, s, (ev)[0], (ev)[1], 0, __VA_ARGS__);                \
    else hipLaunchKernelGGL(kern, grid, block, lds, s, __VA_ARGS__);                                           \
  } while (0)

static void merged_forget(struct msk_ctx* c);   /* the merged-batch tables that hold a context's pointers go with it (below, at msk_batch) */
static unsigned long long g_bind_epoch = 1;   /* msk_bind_buffers calls so far: invalidates the merged-batch table */

struct HostQuery { int npairs; int* d_pairs; float* d_out; };

/* wave-per-env kernels: two envs share a wavefront when the template is small enough */
static int lanes_per_env(const DModel& m) { return (m.nb <= 32 && m.nd < 16) ? 32 : 64; }
/* capacity of k_dynamics' per-lane joint-space rows: lane md computes the unconstrained velocity, so nd <= md (and md < lanes) */
#define MSK_WIDE_WORKERS 256   /* solver workgroups that take envs of the wide class */
static size_t wide_scratch_words(int G) { return G == 16 ? CsWide<16>::SCRATCH : (G == 32 ? CsWide<32>::SCRATCH : CsWide<64>::SCRATCH); }
static int dyn_md(const DModel& m) { return m.nd <= 16 ? 16 : (m.nd <= 32 ? 32 : 64); }
static void launch_kinematics(const DModel& m, const DModel* d_model, const DState& st, hipStream_t s) {
  const int lpe = lanes_per_env(m), epb = 64 / lpe;
  const size_t lds = (size_t)DynLds(m.nb, 0).total * sizeof(float) * epb;
  if (lpe == 32) hipLaunchKernelGGL(k_kinematics<32>, dim3((m.N + 1) / 2), dim3(64), lds, s, d_model, st);
  else hipLaunchKernelGGL(k_kinematics<64>, dim3(m.N), dim3(64), lds, s, d_model, st);
}
/* threads per workgroup of the dynamics launch: 64 (DW + 1) -- DW wavefronts run the dynamics of DW env blocks, one more their broadphase
 * beside it (msk_dynamics.h) -- in the widest form whose wavefronts are all resident at once WITH ROOM TO SPARE (3 per SIMD at the kernel's
 * register count: 3072 on the 256 CUs): 128 threads up to 1280 env blocks, 256 (three dynamics wavefronts + one: a wavefront per SIMD per
 * workgroup) up to 2304 -- 4096 envs at two per wavefront are 683 workgroups x 4 = 89 % of the slots --, else 64 (broadphase as the tail of the
 * one wavefront).  Round 6 (profiles/r06_launch_position_probe.log): 1024 workgroups x 3 filled the slots exactly and workgroups that found their
 * CU full started 21-23 us late.  MI355X, round 2: 512 envs 44 -> 38 us per launch with 128 threads. */
static int dyn_threads(int env_blocks, int lpe = 32) {
  static const int e = getenv("MSK_DYN_THREADS") ? atoi(getenv("MSK_DYN_THREADS")) : 0;   /* tuning aid */
  if (e == 64 || e == 128 || e == 192 || (e == 256 && lpe == 32)) return e;
  if (lpe != 32) { /* a wavefront per env (more registers, up to 36 KB of LDS per env): the forms of round 2 */
    if (2 * env_blocks <= 3072) return 128;
    if (3 * ((env_blocks + 1) / 2) <= 3072) return 192;
    return 64;
  }
  if (2 * env_blocks <= 2560) return 128;
  if (4 * ((env_blocks + 2) / 3) <= 3072) return 256;
  return 64;
}
static void launch_dynamics(const DModel& m, int N, const DModel* d_model, const DState& st, hipStream_t s, hipEvent_t* ev = nullptr) {
  const int lpe = lanes_per_env(m), epb = 64 / lpe;
  const int md = dyn_md(m);
  const int blocks = lpe == 32 ? (N + 1) / 2 : N;      /* env blocks: the envs of one dynamics wavefront */
  const int th = dyn_threads(blocks, lpe);
  const int dw = th > 64 ? th / 64 - 1 : 1;            /* dynamics wavefronts per workgroup */
  const size_t lds = (size_t)DynLds(m.nb, md).total * sizeof(float) * epb * dw;
  const int wgs = (blocks + dw - 1) / dw;
  if (lpe == 32) LAUNCH_TIMED(ev, (k_dynamics<32, 16>), dim3(wgs), dim3(th), lds, s, d_model, st);
  else if (md == 16) LAUNCH_TIMED(ev, (k_dynamics<64, 16>), dim3(wgs), dim3(th), lds, s, d_model, st);
  else if (md == 32) LAUNCH_TIMED(ev, (k_dynamics<64, 32>), dim3(wgs), dim3(th), lds, s, d_model, st);
  else LAUNCH_TIMED(ev, (k_dynamics<64, 64>), dim3(wgs), dim3(th), lds, s, d_model, st);
}

struct msk_ctx {
  int device;
  bool finalized;
  DModel model;          /* host copy of the template */
  DModel* d_model;
  DState st;
  DBuffers bufs;
  int art_root[8], art_dof0[8], art_ndof[8], art_floating[8];
  int *d_art_dof0, *d_art_ndof;
  int max_dof;
  int ndisabled;
  int plane_pairs;   /* candidate pairs with a plane: 0 = the narrowphase launch needs no plane blocks */
  int disabled[256][2];
  pose init_pose[MSK_MAX_BODIES];
  pose pending_root;
  int nverts_total;
  std::vector<float> h_xshape, h_xbody;   /* host mirrors of the per-env instance records [N][nxs | nxb][8] */
  float* d_wrench = nullptr;    /* [N][nb][8] external wrench of the next step (msk_apply FORCE / TORQUE) */
  LinkSlots link_slots;         /* row of a link within its articulation (link incoming joint forces) */
  size_t lds_solve = 0;  /* dynamic LDS of the solver launch */
  int solve_workers = 0; /* its one-env-per-wave workgroups */
  RModel* rmodel;            /* host copy of the render geometry (include/msk_render.h) */
  std::vector<unsigned> texels;  /* host copy of the textures (uploaded by msk_render_finalize) */
  RModel* d_rmodel;
  bool render_finalized;
  int ncams;
  bool cam_no_color[MSK_MAX_CAMERAS] = {false, false, false, false};   /* msk_camera_set_outputs: MSK_CAM_OUT_NO_COLOR */
  RCamera cams[MSK_MAX_CAMERAS];
  msk_pickcube_desc pickcube; /* fused task kernels (include/msk_task.h) */
  bool has_pickcube;
  msk_pusht_desc pusht;
  PushTTables pusht_tb;
  bool has_pusht = false;
  PegTables peg_tb;
  bool has_peg = false;
  bool kin_dirty;        /* link frames in st.bpose are older than (q, qd): run k_kinematics before reading them */
  PairSel pick_lsel, pick_rsel;   /* pickcube task: shape pairs finger <-> object */
  uint32_t groups[MSK_MAX_SHAPES][4]; /* collision groups: only the static pair filter needs them */
  float rest[MSK_MAX_SHAPES];         /* restitution per shape (pair value = average, DPairInfo::rest) */
  float sfric[MSK_MAX_SHAPES], patch_r[MSK_MAX_SHAPES], min_patch_r[MSK_MAX_SHAPES];   /* static friction, torsional patch radii per shape */
  std::vector<void*> allocs;
  std::vector<HostQuery> queries;
  /* env partitions of msk_step (msk_step_n): partition p runs the substep's kernel chain for its contiguous env range on a stream of its
   * own.  A partition is a VIEW: per-env arrays are the context's own memory at the partition's first env, launch-wide structures (solver
   * class lists, the narrowphase's hull queue and sign-off counters, class-3 scratch) are its own; its template copy differs in N only. */
  struct StepPart { DModel* d_model; DState st; int e0, n, solve_workers; };
  std::vector<StepPart> parts;
  /* per-kernel event timing (msk_timing_*): a (begin, end) pair of events per kernel launch of an armed step */
  std::vector<hipEvent_t> tev;
  int t_cap, t_n;
  char err[256];
  char warn[512];
};

static int fail(msk_ctx* c, int code, const char* msg) {
  snprintf(c->err, sizeof(c->err), "%s", msg);
  return code;
}
static int hip_fail(msk_ctx* c, hipError_t e, const char* what) {
  snprintf(c->err, sizeof(c->err), "%s: %s", what, hipGetErrorString(e));
  return MSK_ERR_HIP;
}
#define HIP_TRY(call)                                        \
  do {                                                       \
    hipError_t _e = (call);                                  \
    if (_e != hipSuccess) return hip_fail(c, _e, #call);     \
  } while (0)

static pose pose_from7(const float* p) {
  pose r;
  r.p.x = p[0]; r.p.y = p[1]; r.p.z = p[2];
  float w = p[3], x = p[4], y = p[5], z = p[6];
  float n2 = fmaf(w, w, fmaf(x, x, fmaf(y, y, z * z)));
  float inv = 1.0f / sqrtf(n2);
  r.q.w = w * inv; r.q.x = x * inv; r.q.y = y * inv; r.q.z = z * inv;
  return r;
}
static v3 h_cross(v3 a, v3 b) {
  v3 r = {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
  return r;
}
static v3 h_rotate(quat q, v3 v) {
  v3 u = {q.x, q.y, q.z};
  v3 t = h_cross(u, v);
  t.x += t.x; t.y += t.y; t.z += t.z;
  v3 c2 = h_cross(u, t);
  v3 r = {fmaf(t.x, q.w, v.x) + c2.x, fmaf(t.y, q.w, v.y) + c2.y, fmaf(t.z, q.w, v.z) + c2.z};
  return r;
}
static pose h_pose_inv(pose a) {
  pose r;
  r.q.w = a.q.w; r.q.x = -a.q.x; r.q.y = -a.q.y; r.q.z = -a.q.z;
  v3 t = h_rotate(r.q, a.p);
  r.p.x = -t.x; r.p.y = -t.y; r.p.z = -t.z;
  return r;
}

MSK_API msk_ctx* msk_create(int hip_device, const msk_config* cfg) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= hip_device) {
    fprintf(stderr, "msk_create: no HIP device %d (found %d); this backend has no CPU path\n", hip_device, count);
    return nullptr;
  }
  msk_ctx* c = new msk_ctx();
  memset(&c->model, 0, sizeof(c->model));
  memset(c->model.xs_slot, 0xFF, sizeof(c->model.xs_slot));   /* -1: no per-env instance */
  memset(c->model.xb_slot, 0xFF, sizeof(c->model.xb_slot));
  memset(&c->st, 0, sizeof(c->st));
  memset(&c->bufs, 0, sizeof(c->bufs));
  c->device = hip_device;
  c->finalized = false;
  c->model.cfg = *cfg;
  c->model.cap_contacts = cfg->contact_capacity ? MSK_MAX_CONTACTS_WIDE : MSK_MAX_CONTACTS;
  c->model.cap_blocks = cfg->contact_capacity ? MSK_MAX_BLOCKS_WIDE : MSK_MAX_BLOCKS;
  c->d_model = nullptr;
  c->ndisabled = 0;
  c->nverts_total = 0;
  c->max_dof = 0;
  c->t_cap = 0; c->t_n = 0;
  c->has_pickcube = false;
  c->rmodel = nullptr; c->d_rmodel = nullptr; c->render_finalized = false; c->ncams = 0;
  c->err[0] = 0;
  c->warn[0] = 0;
  if (cfg->sleep_threshold > 0.0f)
    snprintf(c->warn + strlen(c->warn), sizeof(c->warn) - strlen(c->warn), "sleep_threshold=%g accepted, not modelled: bodies never sleep\n", cfg->sleep_threshold);
  if (!cfg->enable_pcm)
    snprintf(c->warn + strlen(c->warn), sizeof(c->warn) - strlen(c->warn), "enable_pcm=0 accepted, no effect: contact manifolds are generated one-shot every step\n");
  return c;
}

MSK_API void msk_destroy(msk_ctx* c) {
  if (!c) return;
  g_bind_epoch++;   /* a merged-batch table may hold this context's pointers */
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  merged_forget(c);
  for (void* p : c->allocs) hipFree(p);
  for (hipEvent_t e : c->tev) hipEventDestroy(e);
  delete c->rmodel;
  delete c;
}

MSK_API const char* msk_last_error(msk_ctx* c) { return c ? c->err : "null context"; }
MSK_API const char* msk_warnings(msk_ctx* c) { return c ? c->warn : ""; }

MSK_API int msk_add_articulation(msk_ctx* c, const float root_pose[7]) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "add_articulation after finalize");
  DModel& m = c->model;
  if (m.na >= 8) return fail(c, MSK_ERR_CAPACITY, "too many articulations");
  c->art_root[m.na] = -1;
  c->art_dof0[m.na] = m.nd;
  c->art_ndof[m.na] = 0;
  c->art_floating[m.na] = 0;
  c->pending_root = pose_from7(root_pose);
  return m.na++;
}

MSK_API int msk_set_articulation_floating(msk_ctx* c, int art) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "set_articulation_floating after finalize");
  if (art < 0 || art >= c->model.na) return fail(c, MSK_ERR_INVALID, "set_articulation_floating: no such articulation");
  c->art_floating[art] = 1;
  return MSK_OK;
}

MSK_API int msk_set_locked_axes(msk_ctx* c, int body, uint32_t mask) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "set_locked_axes after finalize");
  if (body < 0 || body >= c->model.nb || c->model.bodies[body].kind != MSK_BODY_DYNAMIC) return fail(c, MSK_ERR_INVALID, "set_locked_axes: not a dynamic actor");
  c->model.bodies[body].lock = mask & 63u;
  return MSK_OK;
}

MSK_API int msk_add_link(msk_ctx* c, int art, int parent_body, int joint_type, const float pose_in_parent[7],
                         const float pose_in_child[7], float limit_lo, float limit_hi, float mass, const float com[3],
                         const float inertia6[6], int disable_gravity, float armature, float joint_friction) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "add_link after finalize");
  DModel& m = c->model;
  if (m.nb >= MSK_MAX_BODIES - 1) return fail(c, MSK_ERR_CAPACITY, "too many bodies");
  if (art != m.na - 1) return fail(c, MSK_ERR_INVALID, "links must be added to the most recent articulation");
  DBody* b = &m.bodies[m.nb];
  memset(b, 0, sizeof(*b));
  b->kind = MSK_BODY_LINK;
  b->art = art;
  b->parent = parent_body;
  b->jtype = (parent_body < 0) ? MSK_JOINT_FIXED : joint_type;
  b->Xp = pose_from7(pose_in_parent);
  b->XcInv = h_pose_inv(pose_from7(pose_in_child));
  b->lim_lo = limit_lo; b->lim_hi = limit_hi;
  b->mass = mass;
  b->com.x = com[0]; b->com.y = com[1]; b->com.z = com[2];
  memcpy(b->I6, inertia6, sizeof(b->I6));
  b->nograv = disable_gravity;
  b->armature = armature;
  b->jfriction = joint_friction > 0.0f ? joint_friction : 0.0f;
  b->dof = -1; b->vofs = -1; b->root_dof = -1;
  memset(&c->init_pose[m.nb], 0, sizeof(pose));
  c->init_pose[m.nb].q.w = 1.0f;
  if (parent_body < 0) {
    c->init_pose[m.nb] = c->pending_root;
    c->art_root[art] = m.nb;
  } else {
    if (parent_body >= m.nb || m.bodies[parent_body].art != art) return fail(c, MSK_ERR_INVALID, "bad parent link");
    if (b->jtype != MSK_JOINT_FIXED) {
      if (m.nd >= MSK_MAX_DOF - 1) return fail(c, MSK_ERR_CAPACITY, "too many dofs (63 per sub-scene)");
      b->dof = m.nd++;
      c->art_ndof[art]++;
    }
    b->movable = (b->dof >= 0) || m.bodies[parent_body].movable;
  }
  return m.nb++;
}

MSK_API int msk_set_drive(msk_ctx* c, int link_body, float K, float D, float force_limit, int mode_acc) {
  DModel& m = c->model;
  if (link_body < 0 || link_body >= m.nb || m.bodies[link_body].dof < 0) return fail(c, MSK_ERR_INVALID, "set_drive: not an active joint");
  DBody* b = &m.bodies[link_body];
  if (c->finalized && b->K == K && b->D == D && b->fmax == force_limit && b->drive_accel == (mode_acc != 0)) return MSK_OK;
  b->K = K; b->D = D; b->fmax = force_limit; b->drive_accel = mode_acc != 0;
  if (c->finalized) { /* a control-mode switch after gpu_init (agent.set_control_mode): the drive lives in the template, every sub-scene takes it */
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&c->d_model->bodies[link_body], b, sizeof(DBody), hipMemcpyHostToDevice));
    for (auto& p : c->parts)
      if (p.d_model != c->d_model) HIP_TRY(hipMemcpy(&p.d_model->bodies[link_body], b, sizeof(DBody), hipMemcpyHostToDevice));
  }
  return MSK_OK;
}

MSK_API int msk_add_tendon(msk_ctx* c, int link_a, int link_b, float ca, float cb, float rest, float K, float D) {
  DModel& m = c->model;
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "add_tendon after finalize");
  if (m.nt >= MSK_MAX_TENDONS) return fail(c, MSK_ERR_CAPACITY, "too many tendons");
  if (link_a < 0 || link_b < 0 || link_a >= m.nb || link_b >= m.nb) return fail(c, MSK_ERR_INVALID, "bad tendon link");
  if (m.bodies[link_a].dof < 0 || m.bodies[link_b].dof < 0) return fail(c, MSK_ERR_INVALID, "tendon on a fixed joint");
  DTendon* t = &m.tendons[m.nt];
  t->dof_a = m.bodies[link_a].dof; t->dof_b = m.bodies[link_b].dof;
  t->ca = ca; t->cb = cb; t->rest = rest; t->K = K; t->D = D;
  return m.nt++;
}

static void sym6_inverse(const float I[6], float out[6]) {
  float a = I[0], b = I[3], cc = I[4], d = I[1], e = I[5], f = I[2];
  float det = a * (d * f - e * e) - b * (b * f - e * cc) + cc * (b * e - d * cc);
  float inv = 1.0f / det;
  out[0] = (d * f - e * e) * inv;
  out[1] = (a * f - cc * cc) * inv;
  out[2] = (a * d - b * b) * inv;
  out[3] = (cc * e - b * f) * inv;
  out[4] = (b * e - cc * d) * inv;
  out[5] = (b * cc - a * e) * inv;
}

MSK_API int msk_add_actor(msk_ctx* c, int kind, const float pose7[7], float mass, const float com[3],
                          const float inertia6[6], float lin_damp, float ang_damp, int disable_gravity) {
  DModel& m = c->model;
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "add_actor after finalize");
  if (m.nb >= MSK_MAX_BODIES - 1) return fail(c, MSK_ERR_CAPACITY, "too many bodies");
  if (kind != MSK_BODY_KINEMATIC && kind != MSK_BODY_DYNAMIC) return fail(c, MSK_ERR_INVALID, "bad actor kind");
  DBody* b = &m.bodies[m.nb];
  memset(b, 0, sizeof(*b));
  b->kind = kind; b->art = -1; b->parent = -1; b->dof = -1; b->vofs = -1; b->root_dof = -1;
  c->init_pose[m.nb] = pose_from7(pose7);
  b->mass = mass;
  b->com.x = com[0]; b->com.y = com[1]; b->com.z = com[2];
  memcpy(b->I6, inertia6, sizeof(b->I6));
  if (kind == MSK_BODY_DYNAMIC) {
    if (!(mass > 0.0f)) return fail(c, MSK_ERR_INVALID, "dynamic actor needs positive mass");
    sym6_inverse(b->I6, b->Iinv6);
  }
  b->lin_damp = lin_damp; b->ang_damp = ang_damp; b->nograv = disable_gravity;
  b->movable = kind == MSK_BODY_DYNAMIC;
  return m.nb++;
}

MSK_API int msk_add_shape(msk_ctx* c, int body, int type, const float local_pose[7], const float params[3],
                          const float* verts, int nverts, float sf, float df, float rest, const uint32_t groups[4],
                          float patch_radius, float min_patch_radius) {
  DModel& m = c->model;
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "add_shape after finalize");
  if (m.ns >= MSK_MAX_SHAPES) return fail(c, MSK_ERR_CAPACITY, "too many shapes");
  if (body >= m.nb) return fail(c, MSK_ERR_INVALID, "bad body");
  DShape* s = &m.shapes[m.ns];
  memset(s, 0, sizeof(*s));
  s->body = body; s->type = type;
  s->local = pose_from7(local_pose);
  s->par[0] = params[0]; s->par[1] = params[1]; s->par[2] = params[2];
  s->df = df;
  /* sphere / capsule / cylinder become rounded hulls (include/msk_physx.h): core vertices + rounding radius in par[0] */
  float gen[32 * 3];
  if (type == MSK_SHAPE_SPHERE) {
    if (!(params[0] > 0.0f)) return fail(c, MSK_ERR_INVALID, "sphere: radius must be positive");
    gen[0] = gen[1] = gen[2] = 0.0f;
    verts = gen; nverts = 1;
    s->type = type = MSK_SHAPE_CONVEX;
    s->par[0] = params[0]; s->par[1] = s->par[2] = 0.0f;
  } else if (type == MSK_SHAPE_CAPSULE) {
    if (!(params[0] > 0.0f) || !(params[1] >= 0.0f)) return fail(c, MSK_ERR_INVALID, "capsule: radius > 0, half length >= 0");
    gen[0] = -params[1]; gen[1] = gen[2] = 0.0f; gen[3] = params[1]; gen[4] = gen[5] = 0.0f;
    verts = gen; nverts = 2;
    s->type = type = MSK_SHAPE_CONVEX;
    s->par[0] = params[0]; s->par[1] = s->par[2] = 0.0f;
  } else if (type == MSK_SHAPE_CYLINDER) {
    if (!(params[0] > 0.0f) || !(params[1] > 0.0f)) return fail(c, MSK_ERR_INVALID, "cylinder: radius and half length must be positive");
    for (int k = 0; k < 16; ++k) {
      const float a = (float)k * (6.28318530717958647692f / 16.0f);
      const float y = params[0] * cosf(a), z = params[0] * sinf(a);
      gen[3 * k] = -params[1]; gen[3 * k + 1] = y; gen[3 * k + 2] = z;
      gen[3 * (16 + k)] = params[1]; gen[3 * (16 + k) + 1] = y; gen[3 * (16 + k) + 2] = z;
    }
    verts = gen; nverts = 32;
    s->type = type = MSK_SHAPE_CONVEX;
    s->par[0] = s->par[1] = s->par[2] = 0.0f;
  } else if (type == MSK_SHAPE_CONVEX) {
    if (nverts < 4 || nverts > MSK_MAX_HULL_VERTS) return fail(c, MSK_ERR_CAPACITY, "convex: 4..64 vertices");
    if (!(params[0] >= 0.0f)) return fail(c, MSK_ERR_INVALID, "convex: negative rounding radius");
  }
  if (type == MSK_SHAPE_CONVEX) {
    s->nverts = nverts;
    s->vbase = c->nverts_total;
    v3 lo = {3e38f, 3e38f, 3e38f}, hi = {-3e38f, -3e38f, -3e38f};
    for (int i = 0; i < nverts; ++i) {
      v3 p = {verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]};
      m.verts[s->vbase + i] = p;
      lo.x = fminf(lo.x, p.x); lo.y = fminf(lo.y, p.y); lo.z = fminf(lo.z, p.z);
      hi.x = fmaxf(hi.x, p.x); hi.y = fmaxf(hi.y, p.y); hi.z = fmaxf(hi.z, p.z);
    }
    c->nverts_total += nverts;
    s->aabb_c.x = (lo.x + hi.x) * 0.5f; s->aabb_c.y = (lo.y + hi.y) * 0.5f; s->aabb_c.z = (lo.z + hi.z) * 0.5f;
    s->aabb_h.x = (hi.x - lo.x) * 0.5f + s->par[0]; s->aabb_h.y = (hi.y - lo.y) * 0.5f + s->par[0];   /* core box + rounding radius */
    s->aabb_h.z = (hi.z - lo.z) * 0.5f + s->par[0];
  } else if (type == MSK_SHAPE_BOX) {
    s->aabb_h.x = params[0]; s->aabb_h.y = params[1]; s->aabb_h.z = params[2];
  } else if (type == MSK_SHAPE_PLANE) {
    if (body >= 0) return fail(c, MSK_ERR_INVALID, "planes must be static");
  } else {
    return fail(c, MSK_ERR_INVALID, "shape type not supported");
  }
  memcpy(c->groups[m.ns], groups, 4 * sizeof(uint32_t));
  c->rest[m.ns] = rest;
  c->sfric[m.ns] = sf;
  c->patch_r[m.ns] = patch_radius > 0.0f ? patch_radius : 0.0f;
  c->min_patch_r[m.ns] = min_patch_radius > 0.0f ? min_patch_radius : 0.0f;
  return m.ns++;
}

MSK_API int msk_disable_collision(msk_ctx* c, int a, int b) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "disable_collision after finalize");
  if (c->ndisabled >= 256) return fail(c, MSK_ERR_CAPACITY, "too many disabled pairs");
  c->disabled[c->ndisabled][0] = a; c->disabled[c->ndisabled][1] = b;
  c->ndisabled++;
  return MSK_OK;
}

static bool pair_enabled(msk_ctx* c, int i, int j) {
  const DModel& m = c->model;
  const DShape* A = &m.shapes[i];
  const DShape* B = &m.shapes[j];
  const uint32_t* ga = c->groups[i];
  const uint32_t* gb = c->groups[j];
  auto movable = [&](int b) { return b >= 0 && m.bodies[b].movable; };
  if (A->body == B->body) return false;
  if (!movable(A->body) && !movable(B->body)) return false;
  if (ga[2] & gb[2]) return false;
  if (!((ga[0] & gb[1]) || (ga[1] & gb[0]))) return false;
  if (A->body >= 0 && B->body >= 0) {
    const DBody* ba = &m.bodies[A->body];
    const DBody* bb = &m.bodies[B->body];
    if (ba->kind == MSK_BODY_LINK && bb->kind == MSK_BODY_LINK && ba->art == bb->art)
      if (ba->parent == B->body || bb->parent == A->body) return false;
    for (int k = 0; k < c->ndisabled; ++k)
      if ((c->disabled[k][0] == A->body && c->disabled[k][1] == B->body) ||
          (c->disabled[k][0] == B->body && c->disabled[k][1] == A->body))
        return false;
  }
  return true;
}

template <typename T>
static int dev_alloc(msk_ctx* c, T** out, size_t count) {
  void* p = nullptr;
  size_t bytes = (count > 0 ? count : 1) * sizeof(T);
  HIP_TRY(hipMalloc(&p, bytes));
  HIP_TRY(hipMemset(p, 0, bytes));
  c->allocs.push_back(p);
  *out = (T*)p;
  return MSK_OK;
}
#define ALLOC(ptr, count)                                   \
  do {                                                      \
    int _r = dev_alloc(c, &(ptr), (size_t)(count));         \
    if (_r < 0) return _r;                                  \
  } while (0)

/* How many env partitions msk_step runs side by side: ONE unless MSK_STEP_PARTS / msk_set_step_parts ask for more.  The idea -- the substep
 * is three launches whose length is one wave's dependent chain on a chip that is 4-18 % occupied, so half-size chains on two streams should
 * overlap almost freely -- was measured in round 4 (tools/gpu_parts_probe.py, graph replay, one MI355X) and does not hold: PickCube-v1 4096 envs
 * 4.70 M env-steps/s with 1 partition, 4.23 M with 2, 3.75 M with 4 (512 envs: 0.79 / 0.71 / 0.59 M; PegInsertionSide 2.84 / 2.70 / 2.37 M).
 * A half-size launch is barely shorter (k_dynamics 54 -> 52 us, k_csolve 54 -> 50 us: the chain, not the env count, sets its length) and two
 * chains in flight do not overlap enough to pay for the doubled launch count.  Kept as a tuning facility; partitions are whole 64-env
 * classification chunks. */
static int step_part_count(int N) {
  static const int e = getenv("MSK_STEP_PARTS") ? atoi(getenv("MSK_STEP_PARTS")) : 0;
  int P = e > 0 ? e : 1;
  if (P > MSK_STEP_PARTS_MAX) P = MSK_STEP_PARTS_MAX;
  while (P > 1 && (N % (64 * P)) != 0) --P;
  return P;
}

static int build_step_parts(msk_ctx* c, int want = 0) {
  const DModel& m = c->model;
  const int N = m.N;
  int P = want > 0 ? (want > MSK_STEP_PARTS_MAX ? MSK_STEP_PARTS_MAX : want) : step_part_count(N);
  while (P > 1 && (N % (64 * P)) != 0) --P;
  c->parts.clear();
  if (P <= 1) {
    msk_ctx::StepPart p; p.d_model = c->d_model; p.st = c->st; p.e0 = 0; p.n = N; p.solve_workers = c->solve_workers;
    c->parts.push_back(p);
    return MSK_OK;
  }
  const size_t G = (size_t)m.G, npp = (size_t)m.npp, np1 = (size_t)(m.np > 0 ? m.np : 1), npp1 = npp > 0 ? npp : 1;
  const int n = N / P;
  for (int k = 0; k < P; ++k) {
    msk_ctx::StepPart p;
    p.e0 = k * n; p.n = n;
    p.solve_workers = n < 768 ? n : 768;
    const size_t e0 = (size_t)p.e0;
    DModel mp = m;
    mp.N = n;
    ALLOC(p.d_model, 1);
    HIP_TRY(hipMemcpy(p.d_model, &mp, sizeof(DModel), hipMemcpyHostToDevice));
    DState v = c->st;   /* per-env arrays: the context's own memory from env e0 on */
    v.env += e0 * (size_t)m.lay.stride;
    v.Scol += e0 * G * 8; v.W += e0 * G * G; v.vfree += e0 * G;
    v.ct_cnt += e0 * npp; v.ct_rec += e0 * npp * MSK_CT_REC;
    v.np_count += e0 * 4; v.np_items += e0 * NP_TYPES * np1;
    v.ext_wrench += e0 * (size_t)m.nb * 8;
    v.env_ncontacts += e0; v.ct_total += e0;
    v.drv_mask += e0; v.drv += e0 * G * 4;
    if (v.ct_slip) v.ct_slip += e0 * npp;
    v.gjk_cache += e0 * npp1;
    if (v.jforce) v.jforce += e0 * (size_t)m.nb * 6;
    /* launch-wide structures: the partition's own */
    ALLOC(v.cls_list, MSK_SOLVE_CLASSES * (size_t)n); ALLOC(v.cls_count, MSK_SOLVE_CLASSES); ALLOC(v.np_done, (size_t)n);
    ALLOC(v.a_scratch, (size_t)p.solve_workers * 9 * MSK_CLASS3_BLOCKS * (MSK_CLASS3_BLOCKS + 4));
    if (v.wide_scratch) { /* (the context's own slices belong to the unpartitioned step) */
      v.wide_workers = p.solve_workers < MSK_WIDE_WORKERS ? p.solve_workers : MSK_WIDE_WORKERS;
      ALLOC(v.wide_scratch, (size_t)v.wide_workers * wide_scratch_words(m.G));
    }
    ALLOC(v.hq_items, (size_t)n * np1); ALLOC(v.hq_count, 1);
    ALLOC(v.dbg, (size_t)n * 16 + 64 + MSK_DBG_NP_BLOCKS);
    p.st = v;
    c->parts.push_back(p);
  }
  return MSK_OK;
}

MSK_API int msk_finalize(msk_ctx* c, int num_envs) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "finalize twice");
  DModel& m = c->model;
  if (!m.cfg.enable_tgs) return fail(c, MSK_ERR_INVALID, "only the TGS solver is implemented");
  if (num_envs <= 0) return fail(c, MSK_ERR_INVALID, "num_envs must be positive");
  HIP_TRY(hipSetDevice(c->device));
  /* floating roots: six coordinates each, behind the joint dofs (qpos / qvel keep the joints-only layout) */
  for (int a = 0; a < m.na; ++a) {
    if (!c->art_floating[a] || c->art_root[a] < 0) continue;
    if (m.nd + 6 > MSK_MAX_DOF - 1) return fail(c, MSK_ERR_CAPACITY, "too many dofs (63 per sub-scene, 6 per floating root)");
    m.bodies[c->art_root[a]].root_dof = m.nd;
    m.nd += 6;
  }
  for (int i = 0; i < m.nb; ++i) { /* links below a floating root move even behind fixed joints */
    DBody* b = &m.bodies[i];
    if (b->kind != MSK_BODY_LINK) continue;
    if (b->parent < 0) b->movable = b->root_dof >= 0;
    else b->movable = (b->dof >= 0) || m.bodies[b->parent].movable;
  }
  m.nv = m.nd;
  for (int i = 0; i < m.nb; ++i)
    if (m.bodies[i].kind == MSK_BODY_DYNAMIC) { m.bodies[i].vofs = m.nv; m.nv += 6; }
  if (m.nv > MSK_MAX_NV) return fail(c, MSK_ERR_CAPACITY, "generalized velocity too large");
  c->max_dof = 0;
  for (int a = 0; a < m.na; ++a) {
    if (c->art_root[a] < 0) return fail(c, MSK_ERR_INVALID, "articulation without links");
    if (c->art_ndof[a] > c->max_dof) c->max_dof = c->art_ndof[a];
  }
  m.np = 0;
  c->plane_pairs = 0;
  for (int i = 0; i < m.ns; ++i)
    for (int j = i + 1; j < m.ns; ++j)
      if (pair_enabled(c, i, j)) {
        if (m.np >= MSK_MAX_PAIRS) return fail(c, MSK_ERR_CAPACITY, "too many candidate pairs");
        m.pairs[m.np].sa = i; m.pairs[m.np].sb = j;
        if (m.shapes[i].type == MSK_SHAPE_PLANE || m.shapes[j].type == MSK_SHAPE_PLANE) c->plane_pairs++;
        m.np++;
      }
  m.N = num_envs;
  m.nverts_total = c->nverts_total;
  /* lane-group solver tables */
  m.G = (m.nv <= 16) ? 16 : ((m.nv <= 32) ? 32 : 64);
  m.npp = ((m.np + m.G - 1) / m.G) * m.G;
  if (m.npp == 0) m.npp = m.G;
  for (int k = 0; k < MSK_MAX_NV; ++k) { m.coord_moves[k] = 0; m.coord_body[k] = -1; m.coord_root[k] = -1; }
  for (int k = 0; k < MSK_MAX_DOF; ++k) m.dof_body_is_root[k] = 0;
  for (int i = 0; i < m.nb; ++i) {
    const DBody* b = &m.bodies[i];
    if (b->kind == MSK_BODY_LINK) {
      if (b->dof >= 0) { m.dof_lo[b->dof] = b->lim_lo; m.dof_hi[b->dof] = b->lim_hi; }
      if (b->root_dof >= 0) {
        m.coord_root[b->root_dof] = i;
        for (int a = 0; a < 6; ++a) { m.dof_lo[b->root_dof + a] = -3.0e38f; m.dof_hi[b->root_dof + a] = 3.0e38f; m.dof_body[b->root_dof + a] = i; m.dof_body_is_root[b->root_dof + a] = 1; }
      }
      for (int j = i; j >= 0; j = m.bodies[j].parent) {
        if (m.bodies[j].dof >= 0) m.coord_moves[m.bodies[j].dof] |= 1ull << i;
        if (m.bodies[j].root_dof >= 0)
          for (int a = 0; a < 6; ++a) m.coord_moves[m.bodies[j].root_dof + a] |= 1ull << i;
      }
    } else if (b->kind == MSK_BODY_DYNAMIC) {
      for (int a = 0; a < 6; ++a) m.coord_moves[b->vofs + a] = 1ull << i;
      m.coord_body[b->vofs] = i;
    }
  }
  m.maxdepth = 0;
  for (int i = 0; i < m.nb; ++i) {
    const DBody* b = &m.bodies[i];
    m.depth[i] = (b->kind == MSK_BODY_LINK && b->parent >= 0) ? m.depth[b->parent] + 1 : 0;
    if (m.depth[i] > m.maxdepth) m.maxdepth = m.depth[i];
    if (b->kind == MSK_BODY_LINK && b->dof >= 0) m.dof_body[b->dof] = i;
    /* the body's path from its root (entry 0) down to itself (entry depth[i]): what the velocity sums of the forward pass walk */
    for (int k = 0; k < MSK_MAX_BODIES; ++k) m.path[i][k] = (unsigned char)i;
    if (m.depth[i] > 0) {
      for (int k = 0; k < m.depth[i]; ++k) m.path[i][k] = m.path[b->parent][k];
      m.path[i][m.depth[i]] = (unsigned char)i;
    }
  }
  for (int i = 0; i < m.nb; ++i) { /* a link's descendants, in descending body index: what its lane sums up on the way back (the oracle's order) */
    int k = 0;
    memset(m.desc[i], 0, sizeof(m.desc[i]));
    if (m.bodies[i].kind == MSK_BODY_LINK)
      for (int d = m.nb - 1; d > i; --d) {
        if (m.bodies[d].kind != MSK_BODY_LINK) continue;
        int a = m.bodies[d].parent;
        while (a > i) a = m.bodies[a].parent;
        if (a == i) m.desc[i][k++] = (unsigned char)d;
      }
    m.ndesc[i] = (unsigned char)k;
  }
  {
    int o = 0;
    for (int i = 0; i < m.nb; ++i) {
      m.child_off[i] = o;
      for (int j = m.nb - 1; j > i; --j)
        if (m.bodies[j].kind == MSK_BODY_LINK && m.bodies[j].parent == i) m.child_idx[o++] = j;
    }
    m.child_off[m.nb] = o;
  }
  for (int b = 0; b < m.nb; ++b) {
    m.body_coords[b] = 0;
    for (int k = 0; k < m.nv; ++k)
      if ((m.coord_moves[k] >> b) & 1ull) m.body_coords[b] |= 1ull << k;
  }
  for (int p = 0; p < m.np; ++p) {
    const DShape* A = &m.shapes[m.pairs[p].sa];
    const DShape* B = &m.shapes[m.pairs[p].sb];
    m.pinfo[p].ba = A->body; m.pinfo[p].bb = B->body;
    const unsigned long long mca = A->body >= 0 ? m.body_coords[A->body] : 0ull, mcb = B->body >= 0 ? m.body_coords[B->body] : 0ull;
    m.pinfo[p].ca = (unsigned)mca; m.pinfo[p].cb = (unsigned)mcb;
    m.pinfo[p].ca_hi = (unsigned)(mca >> 32); m.pinfo[p].cb_hi = (unsigned)(mcb >> 32);
    const int ia = m.pairs[p].sa, ib = m.pairs[p].sb;
    m.pinfo[p].mu = 0.5f * (A->df + B->df);
    const float mu_s = 0.5f * (c->sfric[ia] + c->sfric[ib]);
    m.pinfo[p].mu_s = mu_s < m.pinfo[p].mu ? m.pinfo[p].mu : mu_s;   /* static below dynamic makes no sense: PhysX raises it */
    m.pinfo[p].rest = 0.5f * (c->rest[ia] + c->rest[ib]);
    m.pinfo[p].patch_r = fmaxf(c->patch_r[ia], c->patch_r[ib]);
    m.pinfo[p].min_patch_r = fmaxf(c->min_patch_r[ia], c->min_patch_r[ib]);
    if (m.pinfo[p].mu_s != m.pinfo[p].mu) m.has_static = 1;
    if (m.pinfo[p].patch_r > 0.0f || m.pinfo[p].min_patch_r > 0.0f) m.has_tors = 1;
  }
  m.jfric_mask = 0ull; m.njfric = 0;
  for (int i = 0; i < m.nb; ++i)
    if (m.bodies[i].kind == MSK_BODY_LINK && m.bodies[i].dof >= 0 && m.bodies[i].jfriction > 0.0f) { m.jfric_mask |= 1ull << m.bodies[i].dof; m.njfric++; }
  {
    EnvLayout& L = m.lay;
    int o = 0;
    const int dpad = m.nd <= 16 ? 16 : (m.nd <= 32 ? 32 : 64);   /* the six joint vectors: 16 floats each as long as that holds the template's dofs */
    L.q = o; o += dpad; L.qd = o; o += dpad; L.qacc = o; o += dpad;
    L.qf = o; o += dpad; L.qt = o; o += dpad; L.qdt = o; o += dpad;
    L.off = o; o += 4;
    L.bpose = o; o += m.nb * 8;
    L.blin = o; o += m.nb * 4; L.bang = o; o += m.nb * 4; L.comw = o; o += m.nb * 4;
    L.xshape = o; o += m.nxs * 8;
    L.xbody = o; o += m.nxb * 8;
    L.stride = o;
  }
  const size_t N = (size_t)num_envs;
  m.cls_cap[0] = (m.G == 16) ? CsLds<16, 16, 16>::fit() : ((m.G == 32) ? CsLds<32, 32, 32>::fit() : CsLds<64, 64, 64>::fit());   /* solver capacity classes (msk_solve.h) */
  m.cls_cap[1] = MSK_CLASS1_BLOCKS;
  m.cls_cap[2] = MSK_CLASS2_BLOCKS;
  m.cls_cap[3] = MSK_CLASS3_BLOCKS;
  ALLOC(c->d_model, 1);
  HIP_TRY(hipMemcpy(c->d_model, &m, sizeof(DModel), hipMemcpyHostToDevice));
  DState& st = c->st;
  ALLOC(st.env, N * (size_t)m.lay.stride);
  /* per-env instances start at the template's values */
  c->h_xshape.assign(N * (size_t)m.nxs * 8, 0.0f);
  c->h_xbody.assign(N * (size_t)m.nxb * 8, 0.0f);
  for (size_t e = 0; e < N; ++e) {
    for (int si = 0; si < m.ns; ++si) {
      if (m.xs_slot[si] < 0) continue;
      float* x = &c->h_xshape[(e * m.nxs + m.xs_slot[si]) * 8];
      const DShape& sh = m.shapes[si];
      x[0] = sh.par[0]; x[1] = sh.par[1]; x[2] = sh.par[2];
      x[4] = sh.local.p.x; x[5] = sh.local.p.y; x[6] = sh.local.p.z;
    }
    for (int bi = 0; bi < m.nb; ++bi) {
      if (m.xb_slot[bi] < 0) continue;
      float* x = &c->h_xbody[(e * m.nxb + m.xb_slot[bi]) * 8];
      const DBody& b = m.bodies[bi];
      x[0] = b.mass; x[1] = b.Iinv6[0]; x[2] = b.Iinv6[1]; x[3] = b.Iinv6[2];
    }
  }
  if (m.nxs > 0)
    HIP_TRY(hipMemcpy2D(st.env + m.lay.xshape, sizeof(float) * m.lay.stride, c->h_xshape.data(), sizeof(float) * m.nxs * 8,
                        sizeof(float) * m.nxs * 8, N, hipMemcpyHostToDevice));
  if (m.nxb > 0)
    HIP_TRY(hipMemcpy2D(st.env + m.lay.xbody, sizeof(float) * m.lay.stride, c->h_xbody.data(), sizeof(float) * m.nxb * 8,
                        sizeof(float) * m.nxb * 8, N, hipMemcpyHostToDevice));
  const size_t G = (size_t)m.G;
  ALLOC(st.Scol, N * G * 8); ALLOC(st.W, N * G * G); ALLOC(st.vfree, N * G);
  ALLOC(st.ct_cnt, N * m.npp); ALLOC(st.ct_rec, N * m.npp * MSK_CT_REC);
  ALLOC(st.cls_list, MSK_SOLVE_CLASSES * N); ALLOC(st.cls_count, MSK_SOLVE_CLASSES); ALLOC(st.np_done, N);
  ALLOC(st.dbg, N * 16 + 64 + MSK_DBG_NP_BLOCKS);
  /* the solver launch (msk_solve.h): one LDS size for every kind of workgroup; one-env-per-wave workers */
  c->solve_workers = num_envs < 768 ? num_envs : 768;
  ALLOC(st.a_scratch, (size_t)c->solve_workers * 9 * MSK_CLASS3_BLOCKS * (MSK_CLASS3_BLOCKS + 4));   /* + prefetch slack */
  st.wide_scratch = nullptr; st.wide_workers = 0;
  if (m.cap_blocks > MSK_MAX_BLOCKS) { /* the wide solver class (msk_solve_wide.h): 0.6 - 0.7 MB of Y and A per worker */
    st.wide_workers = c->solve_workers < MSK_WIDE_WORKERS ? c->solve_workers : MSK_WIDE_WORKERS;
    ALLOC(st.wide_scratch, (size_t)st.wide_workers * wide_scratch_words(m.G));
  }
  if (m.G == 16) {
    auto k0 = k_csolve<16, 16>;
    auto k1 = k_multi_csolve<16, 16>;
    c->lds_solve = CsLds<16, 16, 16>::TOTAL * sizeof(float);
    HIP_TRY(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_solve));
    HIP_TRY(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_solve));
  } else if (m.G == 32) {
    auto k0 = k_csolve<32, 32>;
    auto k1 = k_multi_csolve<32, 32>;
    c->lds_solve = CsLds<32, 32, 32>::TOTAL * sizeof(float);
    HIP_TRY(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_solve));
    HIP_TRY(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_solve));
  } else {
    auto k0 = k_csolve<64, 64>;
    auto k1 = k_multi_csolve<64, 64>;
    c->lds_solve = CsLds<64, 64, 64>::TOTAL * sizeof(float);
    HIP_TRY(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_solve));
    HIP_TRY(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_solve));
  }
  if (dyn_md(m) == 64) { /* the 64-row dynamics keeps two 65 x 64 matrices per env in LDS: above the default dynamic LDS limit with two dynamics wavefronts */
    auto d0 = k_dynamics<64, 64>;
    auto d1 = k_multi_dynamics<64, 64>;
    const int lds_dyn64 = (int)((size_t)DynLds(m.nb, 64).total * sizeof(float) * 2);
    HIP_TRY(hipFuncSetAttribute((const void*)d0, hipFuncAttributeMaxDynamicSharedMemorySize, lds_dyn64));
    HIP_TRY(hipFuncSetAttribute((const void*)d1, hipFuncAttributeMaxDynamicSharedMemorySize, lds_dyn64));
  }
  ALLOC(st.env_ncontacts, N); ALLOC(st.env_overflow, 1); ALLOC(st.ct_total, N);
  ALLOC(st.drv_mask, N); ALLOC(st.drv, N * G * 4);
  if (m.has_static) ALLOC(st.ct_slip, N * m.npp);
  ALLOC(st.gjk_cache, N * (size_t)(m.npp > 0 ? m.npp : 1));
  if (m.njfric > 0) ALLOC(st.jforce, N * (size_t)m.nb * 6);
  ALLOC(st.np_count, N * 4); ALLOC(st.np_items, N * NP_TYPES * (size_t)(m.np > 0 ? m.np : 1));
  ALLOC(st.hq_items, N * (size_t)(m.np > 0 ? m.np : 1)); ALLOC(st.hq_count, 1);
  ALLOC(c->d_art_dof0, 8); ALLOC(c->d_art_ndof, 8);
  HIP_TRY(hipMemcpy(c->d_art_dof0, c->art_dof0, sizeof(int) * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_art_ndof, c->art_ndof, sizeof(int) * 8, hipMemcpyHostToDevice));
  /* external (torch-visible) AoS buffers */
  const size_t nrb = N * (size_t)m.nb * 13;
  const size_t nart = N * (size_t)(m.na > 0 ? m.na : 1) * (size_t)(c->max_dof > 0 ? c->max_dof : 1);
  ALLOC(c->bufs.buf[MSK_BUF_RIGID_BODY_DATA], nrb);
  for (int b = MSK_BUF_ART_QPOS; b <= MSK_BUF_ART_TARGET_QVEL; ++b) ALLOC(c->bufs.buf[b], nart);
  ALLOC(c->bufs.buf[MSK_BUF_RIGID_BODY_FORCE], N * (size_t)m.nb * 4);
  ALLOC(c->bufs.buf[MSK_BUF_RIGID_BODY_TORQUE], N * (size_t)m.nb * 4);
  {
    int count[8] = {0};
    c->link_slots.max_links = 0;
    for (int i = 0; i < m.nb; ++i) {
      c->link_slots.slot[i] = -1;
      if (m.bodies[i].kind != MSK_BODY_LINK) continue;
      c->link_slots.slot[i] = (signed char)count[m.bodies[i].art]++;
      if (count[m.bodies[i].art] > c->link_slots.max_links) c->link_slots.max_links = count[m.bodies[i].art];
    }
  }
  ALLOC(c->bufs.buf[MSK_BUF_ART_LINK_JOINT_FORCES], N * (size_t)(m.na > 0 ? m.na : 1) * (size_t)(c->link_slots.max_links > 0 ? c->link_slots.max_links : 1) * 6);
  ALLOC(c->d_wrench, N * (size_t)m.nb * 8);
  c->st.ext_wrench = c->d_wrench;
  { const int r = build_step_parts(c); if (r < 0) return r; }
  c->bufs.max_dof = c->max_dof;
  c->bufs.pitch = c->max_dof > 0 ? c->max_dof : 1;
  /* initial poses (template replicated) */
  std::vector<float> h(N * (size_t)m.lay.stride, 0.0f);
  for (size_t e = 0; e < N; ++e)
    for (int i = 0; i < m.nb; ++i) {
      const pose& p = c->init_pose[i];
      const float vals[7] = {p.p.x, p.p.y, p.p.z, p.q.w, p.q.x, p.q.y, p.q.z};
      memcpy(&h[e * m.lay.stride + m.lay.bpose + i * 8], vals, sizeof(vals));
    }
  HIP_TRY(hipMemcpy(st.env, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  c->finalized = true;
  c->kin_dirty = false;
  launch_kinematics(m, c->d_model, c->st, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return MSK_OK;
}

MSK_API int msk_set_scene_offsets(msk_ctx* c, const float* offsets) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "set_scene_offsets before finalize");
  const size_t N = (size_t)c->model.N;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy2D(c->st.env + c->model.lay.off, sizeof(float) * c->model.lay.stride, offsets, 3 * sizeof(float),
                      3 * sizeof(float), N, hipMemcpyHostToDevice));
  return MSK_OK;
}

MSK_API void* msk_buffer(msk_ctx* c, int id, int64_t shape[2]) {
  if (!c->finalized || id < 0 || id >= MSK_BUF_COUNT) return nullptr;
  if (id == MSK_BUF_RIGID_BODY_DATA) { shape[0] = (int64_t)c->model.N * c->model.nb; shape[1] = 13; }
  else if (id == MSK_BUF_RIGID_BODY_FORCE || id == MSK_BUF_RIGID_BODY_TORQUE) { shape[0] = (int64_t)c->model.N * c->model.nb; shape[1] = 4; }
  else if (id == MSK_BUF_ART_LINK_JOINT_FORCES) { shape[0] = (int64_t)c->model.N * c->model.na * c->link_slots.max_links; shape[1] = 6; }
  else { shape[0] = (int64_t)c->model.N * c->model.na; shape[1] = c->bufs.pitch; }
  return c->bufs.buf[id];
}

MSK_API int msk_bind_buffers(msk_ctx* c, void* const ptrs[9], int64_t art_pitch) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "bind_buffers before finalize");
  if (art_pitch < (c->max_dof > 0 ? c->max_dof : 1)) return fail(c, MSK_ERR_INVALID, "bind_buffers: art_pitch below max_dof");
  for (int id = 0; id <= MSK_BUF_RIGID_BODY_TORQUE; ++id)
    if (!ptrs[id]) return fail(c, MSK_ERR_INVALID, "bind_buffers: null pointer");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());   /* nothing in flight may still use the old storage (which stays allocated until msk_destroy) */
  for (int id = 0; id <= MSK_BUF_RIGID_BODY_TORQUE; ++id) c->bufs.buf[id] = (float*)ptrs[id];
  c->bufs.pitch = (int)art_pitch;
  g_bind_epoch++;
  return MSK_OK;
}

MSK_API int msk_apply(msk_ctx* c, uint32_t mask, void* stream) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "apply before finalize");
  const int N = c->model.N;
  if (mask & (MSK_APPLY_RIGID_FORCE | MSK_APPLY_RIGID_TORQUE)) { /* external wrench of the next step */
    const int rows = N * c->model.nb;
    hipLaunchKernelGGL(k_apply_wrench, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->d_wrench,
                       c->bufs.buf[MSK_BUF_RIGID_BODY_FORCE], c->bufs.buf[MSK_BUF_RIGID_BODY_TORQUE], rows, mask);
    if (!(mask & ~(uint32_t)(MSK_APPLY_RIGID_FORCE | MSK_APPLY_RIGID_TORQUE))) { HIP_TRY(hipGetLastError()); return MSK_OK; }
  }
  hipLaunchKernelGGL(k_apply, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->d_model, c->st, c->bufs, mask,
                     c->d_art_dof0, c->d_art_ndof);
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

MSK_API int msk_fetch(msk_ctx* c, uint32_t mask, void* stream) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "fetch before finalize");
  const int N = c->model.N;
  if (c->kin_dirty && (mask & MSK_FETCH_RIGID_DATA)) { /* link frames of the post-step (q, qd) */
    launch_kinematics(c->model, c->d_model, c->st, (hipStream_t)stream);
    c->kin_dirty = false;
  }
  if ((mask & MSK_FETCH_ART_LINK_FORCES) && c->model.na > 0) {
    hipLaunchKernelGGL(k_link_forces, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, c->d_model, c->st, c->link_slots,
                       c->bufs.buf[MSK_BUF_ART_LINK_JOINT_FORCES], (float*)nullptr);
    if (!(mask & ~(uint32_t)MSK_FETCH_ART_LINK_FORCES)) { HIP_TRY(hipGetLastError()); return MSK_OK; }
  }
  hipLaunchKernelGGL(k_fetch, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->d_model, c->st, c->bufs, mask,
                     c->d_art_dof0, c->d_art_ndof);
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

MSK_API int msk_reset_masked(msk_ctx* c, const uint8_t* mask, const float* image, int slots, const int32_t* ent, int nent, int32_t* episode, int32_t* elapsed,
                             void* stream) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "reset_masked before finalize");
  if (!mask || !image || !ent || !episode || slots < 1 || nent < 0) return fail(c, MSK_ERR_INVALID, "reset_masked: null argument or empty ring");
  const int N = c->model.N;
  if (c->kin_dirty) { /* the fetch of a named env publishes its link frames: those of the state the last step left */
    launch_kinematics(c->model, c->d_model, c->st, (hipStream_t)stream);
    c->kin_dirty = false;
  }
  ResetPlan rp;
  rp.image = image; rp.ent = ent; rp.nent = nent; rp.slots = slots; rp.mask = mask; rp.episode = episode; rp.elapsed = elapsed;
  const unsigned fetch_mask = MSK_FETCH_RIGID_DATA | MSK_FETCH_ART_QPOS | MSK_FETCH_ART_QVEL | MSK_FETCH_ART_QACC | MSK_FETCH_ART_TARGETS;
  const unsigned apply_mask = MSK_APPLY_RIGID_DATA | MSK_APPLY_ART_ROOT_POSE | MSK_APPLY_ART_QPOS | MSK_APPLY_ART_QVEL | MSK_APPLY_ART_QF | MSK_APPLY_ART_TARGET_QPOS | MSK_APPLY_ART_TARGET_QVEL;
  hipLaunchKernelGGL(k_reset_masked, dim3(N), dim3(64), 0, (hipStream_t)stream, c->d_model, c->st, c->bufs, rp, fetch_mask, apply_mask,
                     c->d_art_dof0, c->d_art_ndof);
  HIP_TRY(hipGetLastError());
  c->kin_dirty = true;   /* joint positions of the named envs changed: their link frames follow at the next fetch / observe */
  return MSK_OK;
}

static const char* episode_book_problem(int n, const msk_episode_book* b) {
  if (n < 1 || !b) return "episode_book_step: no sub-scenes or no book";
  if (!b->terminated || !b->truncated || !b->out_terminated || !b->out_done || !b->any_done) return "episode_book_step: terminated / truncated / out_terminated / out_done / any_done are required";
  if (b->record_metrics) {
    if (!b->reward || !b->elapsed || !b->returns || !b->out_return || !b->out_episode_len || !b->out_reward) return "episode_book_step: record_metrics needs reward, elapsed, returns and their outputs";
    if (b->success && (!b->success_once || !b->out_success_once || (b->ignore_terminations && !b->out_success_at_end))) return "episode_book_step: success without its state / outputs";
    if (b->fail && (!b->fail_once || !b->out_fail_once || (b->ignore_terminations && !b->out_fail_at_end))) return "episode_book_step: fail without its state / outputs";
  }
  return nullptr;
}
MSK_API int msk_episode_book_step(msk_ctx* c, int n, const msk_episode_book* book, void* stream) {
  if (const char* why = episode_book_problem(n, book)) return fail(c, MSK_ERR_INVALID, why);
  hipLaunchKernelGGL(k_episode_book, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, *book);
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

MSK_API int msk_update_kinematics(msk_ctx* c, void* stream) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "update_kinematics before finalize");
  const int N = c->model.N;
  launch_kinematics(c->model, c->d_model, c->st, (hipStream_t)stream);
  c->kin_dirty = false;
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

/* env-group size and block kinds of the narrowphase launch for N envs */
static void np_launch_shape(int N, int plane_pairs, int nverts, int* group_out, NpCfg* cfg) {
  static const int e_group = getenv("MSK_NP_GROUP") ? atoi(getenv("MSK_NP_GROUP")) : 0;
  int group = e_group > 0 ? e_group : N / 256;   /* ~256 x 6 waves whatever the env count */
  group = group < 1 ? 1 : (group > NP_GROUP_MAX ? NP_GROUP_MAX : group);
  while (group & (group - 1)) group &= group - 1;   /* a power of two: groups never straddle the 64-env classification chunks */
  static const int e_nbox = getenv("MSK_NP_NBOX") ? atoi(getenv("MSK_NP_NBOX")) : 2;       /* tuning aids */
  static const int e_nhull = getenv("MSK_NP_NHULL") ? atoi(getenv("MSK_NP_NHULL")) : 4;
  /* Block kinds per env group.  Two box-box blocks: a group whose list needs a second pass of 32 pairs (arms lying on the table) ran them
   * one after the other and ended the launch (one block: k_narrowphase 56 / 59 us, two: 48 / 52; three: 49 / 54).  Dealing the pairs
   * out item by item instead of pass by pass (fewer divergent manifold cases per wave) was measured too and is no better with two
   * blocks and much worse with three or four (63 / 72 us). */
  cfg->nplane = plane_pairs > 0 ? 1 : 0;
  cfg->nbox = e_nbox;
  cfg->nhull = e_nhull;
  cfg->lds_words = np_lds_words(nverts);
  *group_out = group;
}

/* one substep of one env partition (msk_ctx::StepPart) on stream s; ev: the partition's [kernel][begin, end] events of an armed step, or null */
static void step_part(msk_ctx* c, const msk_ctx::StepPart& p, hipStream_t s, hipEvent_t* ev) {
  const int N = p.n;
  const int nblk = (N + 63) / 64;
  hipEvent_t* const ev_dyn = ev ? ev + 2 * MSK_K_DYNAMICS : nullptr;
  hipEvent_t* const ev_np = ev ? ev + 2 * MSK_K_COLLIDE : nullptr;
  hipEvent_t* const ev_cs = ev ? ev + 2 * MSK_K_SOLVE : nullptr;
  /* k_dynamics: joint-space inertia, drives, unconstrained velocities; consumes and clears pending external wrenches (data-driven,
   * graph-safe); its tail is the broadphase of the same envs.  (Running it as a second branch of the captured graph next to the
   * collision kernels OF THE SAME ENVS was measured slower than the serial order: 1.64 against 1.58 ms per control step -- removed.) */
  if (c->model.njfric > 0)   /* joint friction: the wrenches the joints transmitted in the last substep size this one's friction rows */
    hipLaunchKernelGGL(k_link_forces, dim3((N + 63) / 64), dim3(64), 0, s, p.d_model, p.st, c->link_slots, (float*)nullptr, p.st.jforce);
  launch_dynamics(c->model, N, p.d_model, p.st, s, ev_dyn);
  if (c->model.np > 0) {
    int group;
    NpCfg cfg;
    np_launch_shape(N, c->plane_pairs, c->nverts_total, &group, &cfg);
    static const int e_w2 = getenv("MSK_NP_W2") ? atoi(getenv("MSK_NP_W2")) : -1;   /* measurement knob: 0 / 1 = never / always the two-per-SIMD form */
    const dim3 npgrid((N + group - 1) / group, cfg.nplane + cfg.nbox + cfg.nhull);
    if (e_w2 >= 0 ? e_w2 != 0 : N >= 8192)   /* several times more workgroups than one-per-SIMD slots: msk_kernels.h, k_narrowphase_w2 */
      LAUNCH_TIMED(ev_np, k_narrowphase_w2, npgrid, dim3(64), (size_t)cfg.lds_words * sizeof(float), s, p.d_model, p.st, group, cfg);
    else
      LAUNCH_TIMED(ev_np, k_narrowphase, npgrid, dim3(64), (size_t)cfg.lds_words * sizeof(float), s, p.d_model, p.st, group, cfg);
  } else {
    hipMemsetAsync(p.st.cls_count, 0, sizeof(int) * MSK_SOLVE_CLASSES, s);
    LAUNCH_TIMED(ev_np, k_classify, dim3(nblk), dim3(64), 0, s, p.d_model, p.st);
  }
  const int gm = p.solve_workers;
  if (c->model.G == 16) { /* (the wide class's workers, msk_config.contact_capacity = 1, are the first workgroups of this launch: msk_solve.h) */
    auto k0 = k_csolve<16, 16>;
    static const int e_merge = getenv("MSK_WIDE_IN_CSOLVE") ? atoi(getenv("MSK_WIDE_IN_CSOLVE")) : 1;   /* measurement knob: 0 = the launch of its own */
    const int ww = e_merge ? p.st.wide_workers : 0;
    LAUNCH_TIMED(ev_cs, k0, dim3(ww + gm + (N + 3) / 4), dim3(64), c->lds_solve, s, p.d_model, p.st, gm, ww);
    if (!e_merge && p.st.wide_workers > 0) hipLaunchKernelGGL(k_csolve_wide<16>, dim3(p.st.wide_workers), dim3(64), CsWide<16>::TOTAL * sizeof(float), s, p.d_model, p.st);
    return;
  } else if (c->model.G == 32) {
    auto k0 = k_csolve<32, 32>;
    LAUNCH_TIMED(ev_cs, k0, dim3(gm + (N + 1) / 2), dim3(64), c->lds_solve, s, p.d_model, p.st, gm, 0);
  } else {
    auto k0 = k_csolve<64, 64>;
    LAUNCH_TIMED(ev_cs, k0, dim3(gm + N), dim3(64), c->lds_solve, s, p.d_model, p.st, gm, 0);
  }
  if (p.st.wide_workers > 0) { /* msk_config.contact_capacity = 1: the envs of more than MSK_MAX_BLOCKS blocks (msk_solve_wide.h) */
    if (c->model.G == 32) hipLaunchKernelGGL(k_csolve_wide<32>, dim3(p.st.wide_workers), dim3(64), CsWide<32>::TOTAL * sizeof(float), s, p.d_model, p.st);
    else hipLaunchKernelGGL(k_csolve_wide<64>, dim3(p.st.wide_workers), dim3(64), CsWide<64>::TOTAL * sizeof(float), s, p.d_model, p.st);
  }
}

/* side streams of the env partitions, per process and device: partition 0 runs on the caller's stream */
static hipStream_t g_part_stream[MSK_STEP_PARTS_MAX];
static hipEvent_t g_part_fork, g_part_join[MSK_STEP_PARTS_MAX];
static int g_part_dev = -1;

MSK_API int msk_step_n(msk_ctx* c, int count, void* stream) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "step before finalize");
  if (count <= 0) return MSK_OK;
  hipStream_t s = (hipStream_t)stream;
  const int P = (int)c->parts.size();
  const size_t per_step = (size_t)P * 2 * MSK_K_KERNELS;     /* events of one armed step: [partition][kernel][begin, end] */
  if (P > 1 && g_part_dev != c->device) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventCreateWithFlags(&g_part_fork, hipEventDisableTiming));
    for (int k = 1; k < MSK_STEP_PARTS_MAX; ++k) {
      HIP_TRY(hipStreamCreateWithFlags(&g_part_stream[k], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&g_part_join[k], hipEventDisableTiming));
    }
    g_part_dev = c->device;
  }
  /* fork: every partition's chain of `count` substeps is independent of the others' (envs never interact), so the chains only meet again
   * at the join -- inside a captured step graph these are parallel branches */
  if (P > 1) {
    HIP_TRY(hipEventRecord(g_part_fork, s));
    for (int k = 1; k < P; ++k) HIP_TRY(hipStreamWaitEvent(g_part_stream[k], g_part_fork, 0));
  }
  const int t0 = c->t_n;
  for (int k = 0; k < P; ++k) {
    hipStream_t sk = k == 0 ? s : g_part_stream[k];
    for (int i = 0; i < count; ++i) {
      const bool timed = t0 + i < c->t_cap;
      step_part(c, c->parts[k], sk, timed ? &c->tev[(size_t)(t0 + i) * per_step + (size_t)k * 2 * MSK_K_KERNELS] : nullptr);
    }
  }
  for (int i = 0; i < count; ++i) if (c->t_n < c->t_cap) c->t_n++;
  if (P > 1) {
    for (int k = 1; k < P; ++k) {
      HIP_TRY(hipEventRecord(g_part_join[k], g_part_stream[k]));
      HIP_TRY(hipStreamWaitEvent(s, g_part_join[k], 0));
    }
  }
  c->kin_dirty = true;
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

MSK_API int msk_step(msk_ctx* c, void* stream) { return msk_step_n(c, 1, stream); }

MSK_API int msk_get_step_parts(msk_ctx* c) { return c->finalized ? (int)c->parts.size() : 1; }

MSK_API int msk_set_step_parts(msk_ctx* c, int parts) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "set_step_parts before finalize");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  c->t_cap = c->t_n = 0;      /* armed timing was laid out for the old partition count */
  { const int r = build_step_parts(c, parts); if (r < 0) return r; }   /* (the old partitions' launch-wide arrays stay allocated until msk_destroy) */
  /* the partitions' template copies follow the context's (drives and solver classes may have been changed since finalize) */
  for (auto& p : c->parts) {
    if (p.d_model == c->d_model) continue;
    DModel mp = c->model; mp.N = p.n;
    HIP_TRY(hipMemcpy(p.d_model, &mp, sizeof(DModel), hipMemcpyHostToDevice));
  }
  return (int)c->parts.size();
}

/* msk_batch, merged form: the contexts share every launch (k_multi_*, msk_kernels.h).  Possible when they are instances of the same
 * kernel variants (lanes per env, dof padding, solver padding) and nothing per-context is armed (kernel timing).  The table of the
 * contexts (GroupRef) lives in device memory and is rebuilt when the list of contexts or a buffer binding changes. */
struct MergedCache {
  std::vector<msk_ctx*> key;
  unsigned long long epoch = 0;
  GroupRef* d_refs = nullptr;
  int t_dyn = 0, t_np = 0, t_cs = 0, t_af = 0;
  int wide_workers = 0;   /* largest DState::wide_workers of the contexts (msk_config.contact_capacity = 1), 0: no wide launch */
  size_t lds_dyn = 0, lds_kin = 0, lds_np = 0;
  bool any_np = false;
  unsigned long long used = 0;   /* last use, for eviction */
};
/* one table per list of contexts, kept for as long as its contexts live: a captured step graph holds the table's device address, so a table is
 * never freed or moved behind a graph's back -- when a buffer binding changes it is rewritten IN PLACE, and only msk_destroy of one of its
 * contexts releases it.  (Round 4 kept four slots, least recently used evicted and freed: two envs of 17 contexts each in one process evicted each
 * other's tables, and a replayed graph then read whatever the allocator had put at the old address -- first seen on hardware in round 5,
 * tests/test_fused_step.py::test_open_cabinet_drawer_step_as_one_hip_graph.) */
static std::vector<MergedCache*> g_merged;
static unsigned long long g_merged_clock = 0;
static void merged_forget(msk_ctx* c) { /* msk_destroy: the device is idle (synchronised by the caller) */
  for (size_t k = 0; k < g_merged.size();) {
    MergedCache* q = g_merged[k];
    bool has = false;
    for (msk_ctx* x : q->key) has = has || x == c;
    if (has) { if (q->d_refs) hipFree(q->d_refs); delete q; g_merged.erase(g_merged.begin() + (long)k); } else ++k;
  }
}

static int batch_merged(msk_ctx* const* ctxs, int n, int op, uint32_t mask, hipStream_t s) {
  msk_ctx* c = ctxs[0];
  if (n < 2) return 1;
  static const int e_off = getenv("MSK_BATCH_MERGED") ? atoi(getenv("MSK_BATCH_MERGED")) : 1;
  if (!e_off) return 1;
  const int lpe = lanes_per_env(c->model), md = dyn_md(c->model), G = c->model.G;
  /* mergeable?  decided before any cached table is touched (a list that is not keeps its streams path and costs no synchronisation) */
  for (int i = 0; i < n; ++i) {
    msk_ctx* x = ctxs[i];
    if (!x->finalized || x->device != c->device || x->t_n < x->t_cap) return 1;
    if (lanes_per_env(x->model) != lpe || dyn_md(x->model) != md || x->model.G != G || x->lds_solve != c->lds_solve) return 1;
    if (x->model.njfric > 0) return 1;   /* joint friction: an extra launch per context in front of the dynamics */
    if ((x->model.np > 0) != (c->model.np > 0)) return 1;   /* mixed: the pair-less contexts classify in a kernel of their own */
  }
  if (op == MSK_BATCH_APPLY && (mask & (MSK_APPLY_RIGID_FORCE | MSK_APPLY_RIGID_TORQUE))) return 1;   /* wrench staging: per context */
  if (op == MSK_BATCH_FETCH && (mask & MSK_FETCH_ART_LINK_FORCES)) return 1;
  if (op == MSK_BATCH_STEP && c->model.np == 0) return 1;
  MergedCache* hit = nullptr;
  for (size_t k = 0; k < g_merged.size() && !hit; ++k) {
    MergedCache* q = g_merged[k];
    bool same = (int)q->key.size() == n;
    for (int i = 0; same && i < n; ++i) same = q->key[i] == ctxs[i];
    if (same) hit = q;
  }
  if (!hit || hit->epoch != g_bind_epoch) { /* a new list of contexts, or a binding changed since the table was written: (re)write it, at the same address */
    if (!hit) { hit = new MergedCache(); hit->key.assign(ctxs, ctxs + n); g_merged.push_back(hit); }
    MergedCache& mc = *hit;
    mc.epoch = 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());   /* a launch in flight may still read the table that is rewritten */
    std::vector<GroupRef> refs((size_t)n);
    mc.t_dyn = mc.t_np = mc.t_cs = mc.t_af = 0;
    mc.lds_dyn = mc.lds_kin = mc.lds_np = 0;
    mc.any_np = false;
    mc.wide_workers = 0;
    const int epb = 64 / lpe, epw = 64 / G;       /* class-0 envs per solver workgroup: 4, 2 or 1 */
    for (int i = 0; i < n; ++i) {
      msk_ctx* x = ctxs[i];
      GroupRef& r = refs[(size_t)i];
      const int N = x->model.N;
      r.m = x->d_model; r.st = x->st; r.bufs = x->bufs; r.art_dof0 = x->d_art_dof0; r.art_ndof = x->d_art_ndof;
      r.b_dyn = mc.t_dyn; mc.t_dyn += (N + epb - 1) / epb;
      r.b_af = mc.t_af; mc.t_af += (N + 255) / 256;
      r.gm = x->solve_workers;
      mc.wide_workers = std::max(mc.wide_workers, x->st.wide_workers);
      r.b_cs = mc.t_cs; mc.t_cs += r.gm + (N + epw - 1) / epw;
      np_launch_shape(N, x->plane_pairs, x->nverts_total, &r.np_group, &r.np_cfg);
      r.np_gx = (N + r.np_group - 1) / r.np_group;
      r.np_gy = r.np_cfg.nplane + r.np_cfg.nbox + r.np_cfg.nhull;
      r.b_np = mc.t_np;
      mc.lds_np = std::max(mc.lds_np, (size_t)r.np_cfg.lds_words * sizeof(float));
      if (x->model.np > 0) { mc.t_np += r.np_gx * r.np_gy; mc.any_np = true; } else { r.np_gx = 0; }
      mc.lds_dyn = std::max(mc.lds_dyn, (size_t)DynLds(x->model.nb, md).total * sizeof(float) * epb);
      mc.lds_kin = std::max(mc.lds_kin, (size_t)DynLds(x->model.nb, 0).total * sizeof(float) * epb);
    }
    if (!mc.d_refs) HIP_TRY(hipMalloc(&mc.d_refs, sizeof(GroupRef) * (size_t)n));
    HIP_TRY(hipMemcpy(mc.d_refs, refs.data(), sizeof(GroupRef) * (size_t)n, hipMemcpyHostToDevice));
    mc.epoch = g_bind_epoch;
  }
  MergedCache& mc = *hit;
  mc.used = ++g_merged_clock;
  auto kinematics = [&]() {
    if (lpe == 32) hipLaunchKernelGGL(k_multi_kinematics<32>, dim3(mc.t_dyn), dim3(64), mc.lds_kin, s, mc.d_refs, n);
    else hipLaunchKernelGGL(k_multi_kinematics<64>, dim3(mc.t_dyn), dim3(64), mc.lds_kin, s, mc.d_refs, n);
    for (int i = 0; i < n; ++i) ctxs[i]->kin_dirty = false;
  };
  switch (op) {
    case MSK_BATCH_STEP: {
      if (lpe == 32) hipLaunchKernelGGL((k_multi_dynamics<32, 16>), dim3(mc.t_dyn), dim3(dyn_threads(mc.t_dyn, 64) == 128 ? 128 : 64), mc.lds_dyn, s, mc.d_refs, n);
      else if (md == 16) hipLaunchKernelGGL((k_multi_dynamics<64, 16>), dim3(mc.t_dyn), dim3(dyn_threads(mc.t_dyn, 64) == 128 ? 128 : 64), mc.lds_dyn, s, mc.d_refs, n);
      else if (md == 32) hipLaunchKernelGGL((k_multi_dynamics<64, 32>), dim3(mc.t_dyn), dim3(dyn_threads(mc.t_dyn, 64) == 128 ? 128 : 64), mc.lds_dyn, s, mc.d_refs, n);
      else hipLaunchKernelGGL((k_multi_dynamics<64, 64>), dim3(mc.t_dyn), dim3(dyn_threads(mc.t_dyn, 64) == 128 ? 128 : 64), mc.lds_dyn, s, mc.d_refs, n);
      hipLaunchKernelGGL(k_multi_narrowphase, dim3(mc.t_np), dim3(64), mc.lds_np, s, mc.d_refs, n);
      if (G == 16) { auto k0 = k_multi_csolve<16, 16>; hipLaunchKernelGGL(k0, dim3(mc.t_cs), dim3(64), c->lds_solve, s, mc.d_refs, n); }
      else if (G == 32) { auto k0 = k_multi_csolve<32, 32>; hipLaunchKernelGGL(k0, dim3(mc.t_cs), dim3(64), c->lds_solve, s, mc.d_refs, n); }
      else { auto k0 = k_multi_csolve<64, 64>; hipLaunchKernelGGL(k0, dim3(mc.t_cs), dim3(64), c->lds_solve, s, mc.d_refs, n); }
      if (mc.wide_workers > 0) { /* msk_config.contact_capacity = 1 */
        if (G == 16) hipLaunchKernelGGL(k_multi_csolve_wide<16>, dim3(n * mc.wide_workers), dim3(64), CsWide<16>::TOTAL * sizeof(float), s, mc.d_refs, n, mc.wide_workers);
        else if (G == 32) hipLaunchKernelGGL(k_multi_csolve_wide<32>, dim3(n * mc.wide_workers), dim3(64), CsWide<32>::TOTAL * sizeof(float), s, mc.d_refs, n, mc.wide_workers);
        else hipLaunchKernelGGL(k_multi_csolve_wide<64>, dim3(n * mc.wide_workers), dim3(64), CsWide<64>::TOTAL * sizeof(float), s, mc.d_refs, n, mc.wide_workers);
      }
      for (int i = 0; i < n; ++i) ctxs[i]->kin_dirty = true;
      break;
    }
    case MSK_BATCH_APPLY:
      hipLaunchKernelGGL(k_multi_apply, dim3(mc.t_af), dim3(256), 0, s, mc.d_refs, n, mask);
      break;
    case MSK_BATCH_FETCH: {
      bool dirty = false;
      for (int i = 0; i < n; ++i) dirty = dirty || ctxs[i]->kin_dirty;
      if (dirty && (mask & MSK_FETCH_RIGID_DATA)) kinematics();
      hipLaunchKernelGGL(k_multi_fetch, dim3(mc.t_af), dim3(256), 0, s, mc.d_refs, n, mask);
      break;
    }
    case MSK_BATCH_UPDATE_KINEMATICS:
      kinematics();
      break;
    default:
      return fail(c, MSK_ERR_INVALID, "batch: unknown op");
  }
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

/* fork / join streams of msk_batch, per process (the contexts of a batch live on one device) */
#define MSK_BATCH_STREAMS 32
static hipStream_t g_side[MSK_BATCH_STREAMS];
static hipEvent_t g_fork, g_join[MSK_BATCH_STREAMS];
static int g_side_dev = -1;

MSK_API int msk_batch(msk_ctx* const* ctxs, int n, int op, uint32_t mask, void* stream) {
  if (n <= 0) return MSK_OK;
  msk_ctx* c = ctxs[0];   /* errors of the fork / join are reported on the first context */
  hipStream_t s = (hipStream_t)stream;
  auto one = [&](msk_ctx* c, void* st) -> int {
    switch (op) {
      case MSK_BATCH_STEP: return msk_step(c, st);
      case MSK_BATCH_APPLY: return msk_apply(c, mask, st);
      case MSK_BATCH_FETCH: return msk_fetch(c, mask, st);
      case MSK_BATCH_UPDATE_KINEMATICS: return msk_update_kinematics(c, st);
      default: return fail(c, MSK_ERR_INVALID, "batch: unknown op");
    }
  };
  { const int r = batch_merged(ctxs, n, op, mask, s); if (r != 1) return r; }   /* 1: not mergeable, go on below */
  if (n < 3) { /* not worth a fork */
    for (int i = 0; i < n; ++i) { const int r = one(ctxs[i], stream); if (r < 0) return r; }
    return MSK_OK;
  }
  if (g_side_dev != c->device) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventCreateWithFlags(&g_fork, hipEventDisableTiming));
    for (int k = 0; k < MSK_BATCH_STREAMS; ++k) {
      HIP_TRY(hipStreamCreateWithFlags(&g_side[k], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&g_join[k], hipEventDisableTiming));
    }
    g_side_dev = c->device;
  }
  static const int e_ns = getenv("MSK_BATCH_STREAMS") ? atoi(getenv("MSK_BATCH_STREAMS")) : 8;   /* tuning aid */
  const int cap = e_ns < 1 ? 1 : (e_ns > MSK_BATCH_STREAMS ? MSK_BATCH_STREAMS : e_ns);
  const int ns = n < cap ? n : cap;
  HIP_TRY(hipEventRecord(g_fork, s));
  for (int k = 0; k < ns; ++k) HIP_TRY(hipStreamWaitEvent(g_side[k], g_fork, 0));
  int rc = MSK_OK;
  for (int i = 0; i < n && rc >= 0; ++i) rc = one(ctxs[i], (void*)g_side[i % ns]);
  for (int k = 0; k < ns; ++k) {
    HIP_TRY(hipEventRecord(g_join[k], g_side[k]));
    HIP_TRY(hipStreamWaitEvent(s, g_join[k], 0));
  }
  return rc;
}

MSK_API int msk_timing_enable(msk_ctx* c, int max_steps) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "timing before finalize");
  HIP_TRY(hipSetDevice(c->device));
  const size_t need = (size_t)(max_steps > 0 ? max_steps : 0) * c->parts.size() * (2 * MSK_K_KERNELS);
  while (c->tev.size() < need) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    c->tev.push_back(e);
  }
  c->t_cap = max_steps > 0 ? max_steps : 0;
  c->t_n = 0;
  return MSK_OK;
}

MSK_API int msk_timing_read(msk_ctx* c, int slot, double* total_ms, int32_t* launches) {
  if (slot < 0 || slot >= MSK_K_SLOTS) return fail(c, MSK_ERR_INVALID, "bad kernel slot");
  /* a kernel slot: the kernel's own begin -> end; MSK_K_SUBSTEP: begin of the first kernel -> end of the last (the gaps between the
   * launches included) */
  const int first = slot == MSK_K_SUBSTEP ? 0 : 2 * slot, last = slot == MSK_K_SUBSTEP ? 2 * MSK_K_KERNELS - 1 : 2 * slot + 1;
  const int P = (int)c->parts.size();
  double sum = 0.0;
  for (int i = 0; i < c->t_n * P; ++i) {      /* every launch: armed steps x env partitions */
    hipEvent_t* ev = &c->tev[(size_t)i * (2 * MSK_K_KERNELS)];
    HIP_TRY(hipEventSynchronize(ev[last]));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, ev[first], ev[last]));
    sum += ms;
  }
  *total_ms = sum;
  *launches = c->t_n * P;
  return MSK_OK;
}

MSK_API int msk_query_create_pairs(msk_ctx* c, const int32_t* body_pairs, int npairs) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "query before finalize");
  if (c->queries.size() >= 16) return fail(c, MSK_ERR_CAPACITY, "too many queries");
  HIP_TRY(hipSetDevice(c->device));
  HostQuery q;
  q.npairs = npairs;
  ALLOC(q.d_pairs, 2 * (size_t)npairs);
  HIP_TRY(hipMemcpy(q.d_pairs, body_pairs, sizeof(int) * 2 * (size_t)npairs, hipMemcpyHostToDevice));
  ALLOC(q.d_out, (size_t)c->model.N * npairs * 3);
  c->queries.push_back(q);
  return (int)c->queries.size() - 1;
}

MSK_API int msk_query_create_bodies(msk_ctx* c, const int32_t* bodies, int nbodies) {
  std::vector<int32_t> pairs(2 * (size_t)(nbodies > 0 ? nbodies : 0));
  for (int i = 0; i < nbodies; ++i) { pairs[2 * i] = bodies[i]; pairs[2 * i + 1] = MSK_ANY_BODY; }
  return msk_query_create_pairs(c, pairs.data(), nbodies);
}

MSK_API void* msk_query_buffer(msk_ctx* c, int q, int64_t shape[2]) {
  if (q < 0 || q >= (int)c->queries.size()) return nullptr;
  shape[0] = (int64_t)c->model.N * c->queries[q].npairs; shape[1] = 3;
  return c->queries[q].d_out;
}

MSK_API int msk_query_run(msk_ctx* c, int q, void* stream) {
  if (q < 0 || q >= (int)c->queries.size()) return fail(c, MSK_ERR_INVALID, "bad query");
  const int N = c->model.N;
  hipLaunchKernelGGL(k_query, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->d_model, c->st,
                     c->queries[q].d_pairs, c->queries[q].npairs, c->queries[q].d_out);
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

MSK_API int msk_get_sizes(msk_ctx* c, int32_t out[8]) {
  const DModel& m = c->model;
  out[0] = m.nb; out[1] = m.na; out[2] = c->max_dof; out[3] = m.nv; out[4] = m.ns; out[5] = m.np; out[6] = m.N;
  out[7] = 0;
  if (c->finalized) {
    int flag = 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&flag, c->st.env_overflow, sizeof(int), hipMemcpyDeviceToHost));
    out[7] = flag;
    for (int k = 0; k < c->ncams; ++k) { /* a camera's record / list capacities ran over: its pictures may miss triangles */
      int cf = 0;
      HIP_TRY(hipMemcpy(&cf, c->cams[k].overflow, sizeof(int), hipMemcpyDeviceToHost));
      if (cf) out[7] |= 8;
    }
  }
  return MSK_OK;
}

/* ---- camera pipeline (include/msk_render.h) ------------------------------------------------------------ */
MSK_API int msk_render_add_mesh(msk_ctx* c, int body, const float local_pose[7], const float* verts, int nverts,
                                const int32_t* tris, int ntris, int seg_id) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "render shapes are added after finalize");
  if (c->render_finalized) return fail(c, MSK_ERR_INVALID, "render_add_mesh after render_finalize");
  if (body >= c->model.nb) return fail(c, MSK_ERR_INVALID, "bad body");
  if (!c->rmodel) {
    c->rmodel = new RModel();
    memset(c->rmodel, 0, sizeof(RModel));
    /* ManiSkill's default lighting (envs/sapien_env.py:849-853): ambient 0.3, directional (1, 1, -1) and (0, 0, -1), white */
    RModel& r0 = *c->rmodel;
    r0.ambient[0] = r0.ambient[1] = r0.ambient[2] = 0.3f;
    r0.nlights = 2;
    const float inv3 = 1.0f / sqrtf(3.0f);
    r0.ldir[0][0] = inv3; r0.ldir[0][1] = inv3; r0.ldir[0][2] = -inv3;
    r0.ldir[1][0] = 0.0f; r0.ldir[1][1] = 0.0f; r0.ldir[1][2] = -1.0f;
    for (int l = 0; l < 2; ++l) r0.lcol[l][0] = r0.lcol[l][1] = r0.lcol[l][2] = 1.0f;
  }
  RModel& r = *c->rmodel;
  if (r.ns >= MSK_MAX_RENDER_SHAPES || r.nv + nverts > MSK_MAX_RENDER_VERTS || r.nt + ntris > MSK_MAX_RENDER_TRIS)
    return fail(c, MSK_ERR_CAPACITY, "render geometry capacity exceeded");
  RShape& sh = r.shapes[r.ns];
  sh.body = body; sh.seg = seg_id; sh.local = pose_from7(local_pose);
  sh.color[0] = sh.color[1] = sh.color[2] = 0.8f; sh.color[3] = 1.0f;   /* until msk_render_set_base_color says otherwise */
  sh.xs = -1; sh.tex = -1; sh.v0 = r.nv;
  for (int i = 0; i < nverts; ++i) {
    r.verts[r.nv + i].x = verts[3 * i]; r.verts[r.nv + i].y = verts[3 * i + 1]; r.verts[r.nv + i].z = verts[3 * i + 2];
    r.vshape[r.nv + i] = (unsigned char)r.ns;
  }
  for (int i = 0; i < ntris; ++i) {
    for (int k = 0; k < 3; ++k)
      if (tris[3 * i + k] < 0 || tris[3 * i + k] >= nverts) return fail(c, MSK_ERR_INVALID, "triangle index out of range");
    RTri& t = r.tris[r.nt + i];
    t.v0 = r.nv + tris[3 * i]; t.v1 = r.nv + tris[3 * i + 1]; t.v2 = r.nv + tris[3 * i + 2]; t.shape = r.ns;
  }
  r.nv += nverts; r.nt += ntris;
  return r.ns++;
}

MSK_API int msk_render_set_base_color(msk_ctx* c, int render_shape, const float rgba[4]) {
  if (!c->rmodel || render_shape < 0 || render_shape >= c->rmodel->ns) return fail(c, MSK_ERR_INVALID, "bad render shape");
  if (c->render_finalized) return fail(c, MSK_ERR_INVALID, "render_set_base_color after render_finalize");
  for (int k = 0; k < 4; ++k) c->rmodel->shapes[render_shape].color[k] = rgba[k];
  return MSK_OK;
}

MSK_API int msk_render_set_texture(msk_ctx* c, int render_shape, const uint8_t* rgba, int width, int height, const float* uvs) {
  if (!c->rmodel || render_shape < 0 || render_shape >= c->rmodel->ns) return fail(c, MSK_ERR_INVALID, "bad render shape");
  if (c->render_finalized) return fail(c, MSK_ERR_INVALID, "render_set_texture after render_finalize");
  if (!rgba || !uvs || width <= 0 || height <= 0 || width > 4096 || height > 4096) return fail(c, MSK_ERR_INVALID, "render_set_texture: bad texture");
  RModel& r = *c->rmodel;
  /* the texture and its mip chain (each level the 2 x 2 box average of the one before, rounded; edges clamp), level after level */
  int total = 0;
  for (int w = width, h = height;; w = w > 1 ? w / 2 : 1, h = h > 1 ? h / 2 : 1) { total += w * h; if (w == 1 && h == 1) break; }
  if (r.ntex >= MSK_MAX_TEXTURES || r.ntexels + total > MSK_MAX_TEXELS) return fail(c, MSK_ERR_CAPACITY, "texture capacity exceeded");
  RTexture& t = r.tex[r.ntex];
  t.w = width; t.h = height; t.ofs = r.ntexels;
  c->texels.resize((size_t)r.ntexels + (size_t)total);
  unsigned* src = c->texels.data() + t.ofs;
  for (int i = 0; i < width * height; ++i)
    src[i] = (unsigned)rgba[4 * i] | ((unsigned)rgba[4 * i + 1] << 8) | ((unsigned)rgba[4 * i + 2] << 16) | ((unsigned)rgba[4 * i + 3] << 24);
  for (int w = width, h = height; !(w == 1 && h == 1);) {
    const int w2 = w > 1 ? w / 2 : 1, h2 = h > 1 ? h / 2 : 1;
    unsigned* dst = src + (size_t)w * h;
    for (int y = 0; y < h2; ++y)
      for (int x = 0; x < w2; ++x) {
        const int x0 = w > 1 ? 2 * x : 0, x1 = w > 1 ? 2 * x + 1 : 0, y0 = h > 1 ? 2 * y : 0, y1 = h > 1 ? 2 * y + 1 : 0;
        unsigned o = 0;
        for (int ch = 0; ch < 4; ++ch) {
          const unsigned sum = ((src[y0 * w + x0] >> (8 * ch)) & 0xFFu) + ((src[y0 * w + x1] >> (8 * ch)) & 0xFFu) +
                               ((src[y1 * w + x0] >> (8 * ch)) & 0xFFu) + ((src[y1 * w + x1] >> (8 * ch)) & 0xFFu);
          o |= ((sum + 2u) >> 2) << (8 * ch);
        }
        dst[y * w2 + x] = o;
      }
    src = dst; w = w2; h = h2;
  }
  r.ntexels += total;
  const int v0 = r.shapes[render_shape].v0, v1 = render_shape + 1 < r.ns ? r.shapes[render_shape + 1].v0 : r.nv;
  for (int i = v0; i < v1; ++i) { r.vuv[i][0] = uvs[2 * (i - v0)]; r.vuv[i][1] = uvs[2 * (i - v0) + 1]; }
  r.shapes[render_shape].tex = r.ntex;
  return r.ntex++;
}

MSK_API int msk_render_bind_env_box(msk_ctx* c, int render_shape, int shape) {
  if (!c->rmodel || render_shape < 0 || render_shape >= c->rmodel->ns) return fail(c, MSK_ERR_INVALID, "bad render shape");
  if (c->render_finalized) return fail(c, MSK_ERR_INVALID, "render_bind_env_box after render_finalize");
  if (shape < 0 || shape >= c->model.ns || c->model.xs_slot[shape] < 0) return fail(c, MSK_ERR_INVALID, "render_bind_env_box: shape was not declared");
  c->rmodel->shapes[render_shape].xs = c->model.xs_slot[shape];
  return MSK_OK;
}

MSK_API int msk_render_set_lights(msk_ctx* c, const float ambient[3], int ndir, const float* directions, const float* colors) {
  if (!c->rmodel) return fail(c, MSK_ERR_INVALID, "no render shapes");
  if (c->render_finalized) return fail(c, MSK_ERR_INVALID, "render_set_lights after render_finalize");
  if (ndir < 0 || ndir > MSK_MAX_LIGHTS) return fail(c, MSK_ERR_CAPACITY, "too many directional lights");
  RModel& r = *c->rmodel;
  for (int k = 0; k < 3; ++k) r.ambient[k] = ambient[k];
  r.nlights = ndir;
  for (int l = 0; l < ndir; ++l) {
    const float x = directions[3 * l], y = directions[3 * l + 1], z = directions[3 * l + 2];
    const float len = sqrtf(x * x + y * y + z * z);
    if (!(len > 0.0f)) return fail(c, MSK_ERR_INVALID, "zero light direction");
    r.ldir[l][0] = x / len; r.ldir[l][1] = y / len; r.ldir[l][2] = z / len;
    for (int k = 0; k < 3; ++k) r.lcol[l][k] = colors[3 * l + k];
  }
  return MSK_OK;
}

MSK_API int msk_render_set_local_lights(msk_ctx* c, int n, const float* lights) {
  if (!c->rmodel) return fail(c, MSK_ERR_INVALID, "no render shapes");
  if (c->render_finalized) return fail(c, MSK_ERR_INVALID, "render_set_local_lights after render_finalize");
  if (n < 0 || n > MSK_MAX_LOCAL_LIGHTS) return fail(c, MSK_ERR_CAPACITY, "too many point / spot lights");
  RModel& r = *c->rmodel;
  for (int l = 0; l < n; ++l) {
    const float* p = lights + l * MSK_LOCAL_LIGHT_FLOATS;
    const float len = sqrtf(p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
    const bool spot = p[9] > 0.0f;
    if (spot && !(len > 0.0f)) return fail(c, MSK_ERR_INVALID, "zero spot-light axis");
    if (spot && !(p[9] <= p[10] && p[10] < 3.1415927f * 2.0f)) return fail(c, MSK_ERR_INVALID, "spot light: 0 < inner_fov <= outer_fov < 2 pi");
    for (int k = 0; k < 3; ++k) { r.ppos[l][k] = p[k]; r.pdir[l][k] = spot ? p[3 + k] / len : 0.0f; r.pcol[l][k] = p[6 + k]; }
    r.pcone[l][0] = spot ? (float)cos(0.5 * (double)p[9]) : -2.0f;     /* cosines of the half angles; a point light passes every direction */
    r.pcone[l][1] = spot ? (float)cos(0.5 * (double)p[10]) : -3.0f;
  }
  r.nlocal = n;
  return MSK_OK;
}

MSK_API int msk_render_finalize(msk_ctx* c) {
  if (!c->rmodel) return fail(c, MSK_ERR_INVALID, "no render shapes");
  if (c->render_finalized) return fail(c, MSK_ERR_INVALID, "render_finalize twice");
  HIP_TRY(hipSetDevice(c->device));
  ALLOC(c->d_rmodel, 1);
  c->rmodel->texels = nullptr;
  if (!c->texels.empty()) {
    unsigned* d_tex = nullptr;
    ALLOC(d_tex, c->texels.size());
    HIP_TRY(hipMemcpy(d_tex, c->texels.data(), c->texels.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    c->rmodel->texels = d_tex;
  }
  HIP_TRY(hipMemcpy(c->d_rmodel, c->rmodel, sizeof(RModel), hipMemcpyHostToDevice));
  c->render_finalized = true;
  return MSK_OK;
}

MSK_API int msk_camera_create(msk_ctx* c, int width, int height, float fovy, float near_plane, float far_plane, int mount_body,
                              const float local_pose[7]) {
  if (!c->render_finalized) return fail(c, MSK_ERR_INVALID, "camera_create before render_finalize");
  if (c->ncams >= MSK_MAX_CAMERAS) return fail(c, MSK_ERR_CAPACITY, "too many cameras");
  if (width % 16 || height % 16 || width <= 0 || height <= 0 || (width / MSK_TW) * (height / MSK_TH) > MSK_MAX_TILES)
    return fail(c, MSK_ERR_INVALID, "camera size must be a multiple of 16 with at most 4096 tiles of 16 x 4 pixels (512 x 512)");
  if (mount_body >= c->model.nb) return fail(c, MSK_ERR_INVALID, "bad mount body");
  HIP_TRY(hipSetDevice(c->device));
  const size_t N = (size_t)c->model.N;
  RCamera& cam = c->cams[c->ncams];
  memset(&cam, 0, sizeof(cam));
  cam.W = width; cam.H = height; cam.mount = mount_body;
  cam.tiles_x = width / MSK_TW; cam.tiles_y = height / MSK_TH;
  cam.tile_cap = cam.tiles_x * cam.tiles_y;
  /* set_fovy(fovy, compute_x=True): square pixels, principal point at the image centre */
  cam.fy = (float)(0.5 * height / tan(0.5 * (double)fovy));
  cam.fx = cam.fy;
  cam.cx = 0.5f * width; cam.cy = 0.5f * height;
  cam.near_ = near_plane; cam.far_ = far_plane;
  cam.local = pose_from7(local_pose);
  /* LDS carve of k_render_env: every triangle can become two records (near clip); the first rcap live in LDS, the rest in a global spill
   * area.  rcap = 448 keeps a 128 x 128 camera's workgroup under 40 KB, four to a CU; the benchmarked scenes hold ~400 records. */
  const int nrec = 2 * c->rmodel->nt < 65535 ? 2 * c->rmodel->nt : 65535;      /* list entries are 16-bit record numbers */
  cam.ns = c->rmodel->ns;
  cam.dbg_cut = getenv("MSK_RENDER_CUT") ? atoi(getenv("MSK_RENDER_CUT")) : 0;
  cam.rcap = nrec < 448 ? (nrec > 0 ? nrec : 1) : 448;
  cam.spill_cap = nrec - cam.rcap > 0 ? nrec - cam.rcap : 0;
  cam.icap = std::max(std::max(2048, 2 * cam.tile_cap), nrec);
  if (render_lds_words(cam.ns, cam.rcap, cam.icap, cam.tile_cap) * sizeof(float) > 160 * 1024)
    return fail(c, MSK_ERR_CAPACITY, "camera: picture / model too large for the render workgroup's LDS");
  /* k_render_splat (small triangles off the tile lists): its tile-row lists hold a small record once per tile row it reaches (<= 5, ~2 on
   * the benchmarked scenes, where half of the triangles are culled): bcap = records, like icap an estimate guarded by the overflow flag
   * (msk_get_sizes()[7] & 8).  MSK_RENDER_MODE=0 keeps k_render_env (A/B runs, tools/gpu_render_probe.py). */
  cam.bcap = std::max(1024, nrec);
  cam.mode = getenv("MSK_RENDER_MODE") ? atoi(getenv("MSK_RENDER_MODE")) : 1;
  /* textured shapes: the planes of u / depth and v / depth of their screen triangles live in the workgroup's LDS (32 bytes each), at most 256 */
  cam.uvcap = 0;
  if (c->rmodel->ntex > 0) {
    int ttri = 0;
    for (int t = 0; t < c->rmodel->nt; ++t)
      if (c->rmodel->shapes[c->rmodel->tris[t].shape].tex >= 0) ++ttri;
    cam.uvcap = std::min(2 * ttri, 256);
  }
  if (width > 1023 || height > 1023 ||
      render_splat_lds_words(cam.ns, cam.rcap, cam.icap, cam.tile_cap, render_segments(cam.tiles_x, cam.tiles_y), cam.bcap, cam.uvcap) * sizeof(float) > 160 * 1024)
    cam.mode = 0;
  ALLOC(cam.setups, N * (size_t)(cam.spill_cap > 0 ? cam.spill_cap : 1) * MSK_SETUP_WORDS);
  ALLOC(cam.out, N * (size_t)width * height * 4);
  ALLOC(cam.depth, N * (size_t)width * height);
  ALLOC(cam.seg, N * (size_t)width * height);
  ALLOC(cam.overflow, 1);
  cam.want_tex = 1;
  return c->ncams++;
}

MSK_API int msk_camera_set_outputs(msk_ctx* c, int camera, int position_texture) {
  if (camera < 0 || camera >= c->ncams) return fail(c, MSK_ERR_INVALID, "bad camera");
  c->cams[camera].want_tex = (position_texture & 1) ? 1 : 0;   /* a kernel argument of the next msk_camera_take_picture (a captured step graph keeps the value it was captured with) */
  c->cam_no_color[camera] = (position_texture & MSK_CAM_OUT_NO_COLOR) != 0;
  return MSK_OK;
}

MSK_API void* msk_camera_buffer(msk_ctx* c, int camera, int64_t shape[4]) {
  if (camera < 0 || camera >= c->ncams) return nullptr;
  shape[0] = c->model.N; shape[1] = c->cams[camera].H; shape[2] = c->cams[camera].W; shape[3] = 4;
  return c->cams[camera].out;
}

MSK_API void* msk_camera_obs_buffer(msk_ctx* c, int camera, int which, int64_t shape[4]) {
  if (camera < 0 || camera >= c->ncams || which < MSK_CAM_DEPTH || which > MSK_CAM_COLOR) return nullptr;
  shape[0] = c->model.N; shape[1] = c->cams[camera].H; shape[2] = c->cams[camera].W; shape[3] = which == MSK_CAM_COLOR ? 4 : 1;
  if (which == MSK_CAM_COLOR) { /* allocated on first request: cameras that never hand out Color do not shade or store it */
    RCamera& cam = c->cams[camera];
    if (!cam.color) {
      if (hipSetDevice(c->device) != hipSuccess) return nullptr;
      if (dev_alloc(c, &cam.color, (size_t)c->model.N * cam.W * cam.H) < 0) return nullptr;
      if (cam.uvcap > 0 && cam.mode == 1 && dev_alloc(c, &cam.uvt, (size_t)c->model.N * cam.W * cam.H) < 0) return nullptr;
    }
    return cam.color;
  }
  return which == MSK_CAM_DEPTH ? (void*)c->cams[camera].depth : (void*)c->cams[camera].seg;
}

MSK_API int msk_camera_take_picture(msk_ctx* c, int camera, void* stream) {
  if (camera < 0 || camera >= c->ncams) return fail(c, MSK_ERR_INVALID, "bad camera");
  const int N = c->model.N;
  hipStream_t s = (hipStream_t)stream;
  if (c->kin_dirty) { /* link frames of the current (q, qd) */
    launch_kinematics(c->model, c->d_model, c->st, s);
    c->kin_dirty = false;
  }
  RCamera cam = c->cams[camera];
  if (c->cam_no_color[camera]) { cam.color = nullptr; cam.uvt = nullptr; }   /* msk_camera_set_outputs: neither shaded nor stored */
  if (cam.mode == 1) {
    const size_t lds = render_splat_lds_words(cam.ns, cam.rcap, cam.icap, cam.tile_cap, render_segments(cam.tiles_x, cam.tiles_y), cam.bcap, cam.uvcap) * sizeof(float);
    if (lds > 64 * 1024)   /* above the default dynamic LDS limit (large pictures, large models): the CU has 160 KB */
      HIP_TRY(hipFuncSetAttribute((const void*)k_render_splat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_render_splat, dim3(N), dim3(MSK_RENDER_THREADS), lds, s, c->d_model, c->st, c->d_rmodel, cam);
    if (cam.color && cam.uvt) { /* textured pixels: Color = texel * shade */
      const size_t npix = (size_t)N * cam.W * cam.H;
      hipLaunchKernelGGL(k_render_texture, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, c->d_rmodel, cam.color, cam.uvt, npix);
    }
  } else {
    const size_t lds = render_lds_words(cam.ns, cam.rcap, cam.icap, cam.tile_cap) * sizeof(float);
    if (lds > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute((const void*)k_render_env, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_render_env, dim3(N), dim3(MSK_RENDER_THREADS), lds, s, c->d_model, c->st, c->d_rmodel, cam);
  }
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

/* ---- fused task kernels (include/msk_task.h) ---------------------------------------------------------- */
MSK_API int msk_task_pickcube_init(msk_ctx* c, const msk_pickcube_desc* d) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "task init before finalize");
  const int nb = c->model.nb;
  const int ids[5] = {d->cube, d->goal, d->tcp, d->left_finger, d->right_finger};
  for (int i = 0; i < 5; ++i)
    if (ids[i] < 0 || ids[i] >= nb) return fail(c, MSK_ERR_INVALID, "pickcube: bad body id");
  if (d->arm_dofs + 2 != c->model.nd || 2 * (d->arm_dofs + 2) + 24 != 42) return fail(c, MSK_ERR_INVALID, "pickcube: expects a 7+2 dof arm");
  c->pickcube = *d;
  c->has_pickcube = true;
  /* candidate pairs between each finger and the object, ascending: what the observe kernel sums contact impulses over */
  for (int side = 0; side < 2; ++side) {
    PairSel* ps = side ? &c->pick_rsel : &c->pick_lsel;
    const int x = side ? d->right_finger : d->left_finger, y = d->cube;
    ps->n = 0;
    for (int p = 0; p < c->model.np; ++p) {
      const int ba = c->model.pinfo[p].ba, bb = c->model.pinfo[p].bb;
      float sgn;
      if (ba == x && bb == y) sgn = 1.0f;
      else if (ba == y && bb == x) sgn = -1.0f;
      else continue;
      if (ps->n >= 14) return fail(c, MSK_ERR_CAPACITY, "pickcube: too many shape pairs between a finger and the object");
      ps->idx[ps->n] = p; ps->sgn[ps->n] = sgn; ps->n++;
    }
  }
  return MSK_OK;
}

MSK_API int msk_task_pickcube_set_action(msk_ctx* c, const float* actions, void* stream) {
  if (!c->has_pickcube) return fail(c, MSK_ERR_INVALID, "pickcube task not initialised");
  const int N = c->model.N;
  hipLaunchKernelGGL(k_pickcube_set_action, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->d_model, c->st, c->pickcube, actions);
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

MSK_API int msk_task_pickcube_set_action_ee(msk_ctx* c, const float* actions, int action_dim, int root_body, float pos_bound,
                                            float rot_scale, float lambda, void* stream) {
  if (!c->has_pickcube) return fail(c, MSK_ERR_INVALID, "pickcube task not initialised");
  if (action_dim != 4 && action_dim != 7) return fail(c, MSK_ERR_INVALID, "ee control: action_dim is 4 (pos) or 7 (pose)");
  if (root_body < 0 || root_body >= c->model.nb || c->pickcube.arm_dofs != 7) return fail(c, MSK_ERR_INVALID, "ee control: bad root body / arm");
  const int N = c->model.N;
  if (c->kin_dirty) { /* the Jacobian is built from the link frames of the current qpos */
    launch_kinematics(c->model, c->d_model, c->st, (hipStream_t)stream);
    c->kin_dirty = false;
  }
  EeCtl ec;
  ec.root = root_body; ec.adim = action_dim; ec.pos_bound = pos_bound; ec.rot_scale = rot_scale; ec.damping = lambda;
  hipLaunchKernelGGL(k_pickcube_set_action_ee, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->d_model, c->st, c->pickcube, ec,
                     actions);
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

MSK_API int msk_task_pusht_init(msk_ctx* c, const msk_pusht_desc* d, const uint8_t* tee_render) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "task init before finalize");
  const int nb = c->model.nb;
  const int ids[3] = {d->tee, d->goal, d->tcp};
  for (int i = 0; i < 3; ++i)
    if (ids[i] < 0 || ids[i] >= nb) return fail(c, MSK_ERR_INVALID, "pusht: bad body id");
  if (d->arm_dofs != c->model.nd || d->arm_dofs > MSK_MAX_DOF) return fail(c, MSK_ERR_INVALID, "pusht: expects an arm without gripper joints");
  HIP_TRY(hipSetDevice(c->device));
  /* the mask's pixels in row-major order (the order of the boolean-mask gather of push_t.py:383), and the image of the goal
   * mask under final.permute(0, 2, 1).flip(1): a mark at (ix, iy) is compared with tee_render[63 - iy][ix] */
  std::vector<unsigned short> src;
  std::vector<unsigned> hit(128, 0u);
  for (int r = 0; r < 64; ++r)
    for (int col = 0; col < 64; ++col)
      if (tee_render[r * 64 + col]) src.push_back((unsigned short)(r << 8 | col));
  for (int ix = 0; ix < 64; ++ix)
    for (int iy = 0; iy < 64; ++iy)
      if (tee_render[(63 - iy) * 64 + ix]) hit[(ix * 64 + iy) >> 5] |= 1u << ((ix * 64 + iy) & 31);
  if (src.empty()) return fail(c, MSK_ERR_INVALID, "pusht: empty mask");
  unsigned short* d_src = nullptr;
  unsigned* d_hit = nullptr;
  ALLOC(d_src, src.size());
  ALLOC(d_hit, 128);
  HIP_TRY(hipMemcpy(d_src, src.data(), src.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_hit, hit.data(), 128 * sizeof(unsigned), hipMemcpyHostToDevice));
  c->pusht = *d;
  c->pusht_tb.src = d_src; c->pusht_tb.nsrc = (int)src.size(); c->pusht_tb.hit = d_hit;
  c->has_pusht = true;
  return MSK_OK;
}

MSK_API int msk_task_pusht_set_action(msk_ctx* c, const float* actions, void* stream) {
  if (!c->has_pusht) return fail(c, MSK_ERR_INVALID, "pusht task not initialised");
  const int N = c->model.N;
  hipLaunchKernelGGL(k_pusht_set_action, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->d_model, c->st, c->pusht, actions);
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

MSK_API int msk_task_pusht_observe(msk_ctx* c, float* obs, int obs_dim, float* reward, uint8_t* flags, int32_t* elapsed, int advance,
                                   void* stream) {
  if (!c->has_pusht) return fail(c, MSK_ERR_INVALID, "pusht task not initialised");
  if (obs_dim != 2 * c->pusht.arm_dofs + 7 && obs_dim != 2 * c->pusht.arm_dofs + 17) return fail(c, MSK_ERR_INVALID, "pusht: obs_dim is 21 or 31");
  if (c->kin_dirty) { /* the usual case behind msk_control_step: frames and observation in one launch */
    hipLaunchKernelGGL(k_pusht_observe<true>, dim3(c->model.N), dim3(64), (size_t)DynLds(c->model.nb, 0).total * sizeof(float), (hipStream_t)stream, c->d_model, c->st,
                       c->pusht, c->pusht_tb, obs, obs_dim, reward, flags, elapsed, advance);
    c->kin_dirty = false;
  } else
    hipLaunchKernelGGL(k_pusht_observe<false>, dim3(c->model.N), dim3(64), 0, (hipStream_t)stream, c->d_model, c->st, c->pusht, c->pusht_tb, obs,
                       obs_dim, reward, flags, elapsed, advance);
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

MSK_API int msk_task_peg_init(msk_ctx* c, const float* peg_half_sizes, const float* hole_offsets, const float* hole_radii) {
  if (!c->has_pickcube) return fail(c, MSK_ERR_INVALID, "peg: bind the pickcube task first (its cube = the peg, its goal = box_with_hole)");
  if (2 * (c->pickcube.arm_dofs + 2) + 25 != 43) return fail(c, MSK_ERR_INVALID, "peg: expects a 7+2 dof arm");
  const size_t N = (size_t)c->model.N;
  float *half = nullptr, *hole = nullptr, *radius = nullptr;
  ALLOC(half, N * 3); ALLOC(hole, N * 3); ALLOC(radius, N);
  HIP_TRY(hipMemcpy(half, peg_half_sizes, sizeof(float) * N * 3, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(hole, hole_offsets, sizeof(float) * N * 3, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(radius, hole_radii, sizeof(float) * N, hipMemcpyHostToDevice));
  c->peg_tb.half = half; c->peg_tb.hole = hole; c->peg_tb.radius = radius;
  c->has_peg = true;
  return MSK_OK;
}

MSK_API int msk_task_peg_observe(msk_ctx* c, float* obs, float* reward, uint8_t* flags, int32_t* elapsed, float* head_at_hole,
                                 int advance, void* stream) {
  if (!c->has_peg) return fail(c, MSK_ERR_INVALID, "peg task not initialised");
  const int N = c->model.N;
  const float cos_max = cosf(c->pickcube.max_angle_deg * 3.14159265358979323846f / 180.0f);
  if (c->kin_dirty) { /* the usual case behind msk_control_step: frames and observation in one launch */
    const int lpe = lanes_per_env(c->model), epb = 64 / lpe;
    const size_t lds = (size_t)DynLds(c->model.nb, 0).total * sizeof(float) * epb;
    if (lpe == 32)
      hipLaunchKernelGGL(k_peg_observe_kin<32>, dim3((N + 1) / 2), dim3(64), lds, (hipStream_t)stream, c->d_model, c->st, c->pickcube, c->pick_lsel, c->pick_rsel,
                         c->peg_tb, obs, reward, flags, elapsed, head_at_hole, advance, cos_max);
    else
      hipLaunchKernelGGL(k_peg_observe_kin<64>, dim3(N), dim3(64), lds, (hipStream_t)stream, c->d_model, c->st, c->pickcube, c->pick_lsel, c->pick_rsel,
                         c->peg_tb, obs, reward, flags, elapsed, head_at_hole, advance, cos_max);
    c->kin_dirty = false;
  } else
    hipLaunchKernelGGL(k_peg_observe, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, c->d_model, c->st, c->pickcube, c->pick_lsel, c->pick_rsel, c->peg_tb,
                       obs, reward, flags, elapsed, head_at_hole, advance, cos_max);
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

MSK_API int msk_compute_ik_delta(msk_ctx* c, const msk_ik_desc* d, const float* delta_pose, float* target_qpos, int commit_targets, void* stream) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "compute_ik_delta before finalize");
  const DModel& m = c->model;
  if (!d || !delta_pose) return fail(c, MSK_ERR_INVALID, "compute_ik_delta: null argument");
  if (d->njoints < 1 || d->njoints > MSK_IK_MAX_JOINTS) return fail(c, MSK_ERR_CAPACITY, "compute_ik_delta: 1 .. 8 controlled joints");
  if (d->ee_body < 0 || d->ee_body >= m.nb || d->root_body < 0 || d->root_body >= m.nb || m.bodies[d->ee_body].kind != MSK_BODY_LINK ||
      m.bodies[d->root_body].kind != MSK_BODY_LINK)
    return fail(c, MSK_ERR_INVALID, "compute_ik_delta: end link and root link must be articulation links");
  IkCtl ic;
  ic.ee_body = d->ee_body; ic.root_body = d->root_body; ic.njoints = d->njoints; ic.damping = d->damping; ic.alpha = d->alpha;
  for (int k = 0; k < MSK_IK_MAX_JOINTS; ++k) ic.dofs[k] = 0;
  for (int k = 0; k < d->njoints; ++k) {
    const int jl = d->joint_links[k];
    if (jl < 0 || jl >= m.nb || m.bodies[jl].kind != MSK_BODY_LINK || m.bodies[jl].dof < 0)
      return fail(c, MSK_ERR_INVALID, "compute_ik_delta: not the child link of a moving joint");
    const DBody& jb = m.bodies[jl];
    if (jb.jtype != MSK_JOINT_REVOLUTE && jb.jtype != MSK_JOINT_PRISMATIC) return fail(c, MSK_ERR_INVALID, "compute_ik_delta: joint type");
    bool on_chain = false;      /* the joint's child link is the end link or one of its ancestors, strictly below the root */
    for (int b = d->ee_body; b >= 0 && b != d->root_body; b = m.bodies[b].parent)
      if (b == jl) { on_chain = true; break; }
    bool below_root = false;
    for (int b = jb.parent; b >= 0; b = m.bodies[b].parent)
      if (b == d->root_body) { below_root = true; break; }
    if (!on_chain || !below_root) return fail(c, MSK_ERR_INVALID, "compute_ik_delta: a controlled joint does not lie between the root and the end link");
    for (int j = 0; j < k; ++j)
      if (d->joint_links[j] == jl) return fail(c, MSK_ERR_INVALID, "compute_ik_delta: joint listed twice");
    ic.dofs[k] = jb.dof;
  }
  if (c->kin_dirty) { /* the Jacobian is built from the link frames of the current qpos */
    launch_kinematics(c->model, c->d_model, c->st, (hipStream_t)stream);
    c->kin_dirty = false;
  }
  const int N = m.N;
  hipLaunchKernelGGL(k_ik_delta, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->d_model, c->st, ic, delta_pose, target_qpos, commit_targets);
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

MSK_API int msk_control_step(msk_ctx* c, int substeps, void* stream) {
  /* the link frames of the post-step state are left to the consumer (every one of them checks kin_dirty): the PickCube observation
   * computes them in its own launch (k_pickcube_observe_kin) */
  return msk_step_n(c, substeps, stream);
}

MSK_API int msk_task_pickcube_observe(msk_ctx* c, float* obs, float* reward, uint8_t* flags, int32_t* elapsed, int advance,
                                      void* stream) {
  if (!c->has_pickcube) return fail(c, MSK_ERR_INVALID, "pickcube task not initialised");
  const int N = c->model.N;
  const float cos_max = cosf(c->pickcube.max_angle_deg * 3.14159265358979323846f / 180.0f);
  if (c->kin_dirty) { /* the usual case behind msk_control_step: frames and observation in one launch */
    const int lpe = lanes_per_env(c->model), epb = 64 / lpe;
    const size_t lds = (size_t)DynLds(c->model.nb, 0).total * sizeof(float) * epb;
    if (lpe == 32)
      hipLaunchKernelGGL(k_pickcube_observe_kin<32>, dim3((N + 1) / 2), dim3(64), lds, (hipStream_t)stream, c->d_model, c->st, c->pickcube, c->pick_lsel,
                         c->pick_rsel, obs, reward, flags, elapsed, advance, cos_max);
    else
      hipLaunchKernelGGL(k_pickcube_observe_kin<64>, dim3(N), dim3(64), lds, (hipStream_t)stream, c->d_model, c->st, c->pickcube, c->pick_lsel,
                         c->pick_rsel, obs, reward, flags, elapsed, advance, cos_max);
    c->kin_dirty = false;
  } else
    hipLaunchKernelGGL(k_pickcube_observe, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, c->d_model, c->st, c->pickcube, c->pick_lsel,
                       c->pick_rsel, obs, reward, flags, elapsed, advance, cos_max);
  HIP_TRY(hipGetLastError());
  return MSK_OK;
}

#ifdef MSK_PROFILE_PHASES
MSK_API int msk_debug_reset(msk_ctx* c) {
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(c->st.dbg, 0, sizeof(long long) * (16 * (size_t)c->model.N + 64 + MSK_DBG_NP_BLOCKS)));
  return MSK_OK;
}
/* development aid (MSK_PROFILE_PHASES builds): per-env cycle stamps of the solver and dynamics phases, the narrowphase's counters, the 100 MHz stamps of the narrowphase's
 * workgroups: out[num_envs * 16 + 64 + MSK_DBG_NP_BLOCKS] */
MSK_API int msk_debug_phases(msk_ctx* c, long long* out) {
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, c->st.dbg, sizeof(long long) * (16 * (size_t)c->model.N + 64 + MSK_DBG_NP_BLOCKS), hipMemcpyDeviceToHost));
  return MSK_OK;
}
#endif

MSK_API int msk_declare_env_box(msk_ctx* c, int shape) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "declare_env_box after finalize");
  DModel& m = c->model;
  if (shape < 0 || shape >= m.ns || m.shapes[shape].type != MSK_SHAPE_BOX) return fail(c, MSK_ERR_INVALID, "declare_env_box: not a box shape");
  if (m.xs_slot[shape] < 0) m.xs_slot[shape] = (signed char)m.nxs++;
  return MSK_OK;
}

MSK_API int msk_declare_env_mass(msk_ctx* c, int body) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "declare_env_mass after finalize");
  DModel& m = c->model;
  if (body < 0 || body >= m.nb || m.bodies[body].kind != MSK_BODY_DYNAMIC) return fail(c, MSK_ERR_INVALID, "declare_env_mass: not a dynamic actor");
  const DBody& b = m.bodies[body];
  if (b.com.x != 0.0f || b.com.y != 0.0f || b.com.z != 0.0f || b.I6[3] != 0.0f || b.I6[4] != 0.0f || b.I6[5] != 0.0f)
    return fail(c, MSK_ERR_INVALID, "declare_env_mass: needs the centre of mass at the origin and a diagonal inertia");
  if (m.xb_slot[body] < 0) m.xb_slot[body] = (signed char)m.nxb++;
  return MSK_OK;
}

MSK_API int msk_set_env_boxes(msk_ctx* c, int shape, const float* half_sizes, const float* local_pos) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "set_env_boxes before finalize");
  const DModel& m = c->model;
  if (shape < 0 || shape >= m.ns || m.xs_slot[shape] < 0) return fail(c, MSK_ERR_INVALID, "set_env_boxes: shape was not declared");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t N = (size_t)m.N;
  for (size_t e = 0; e < N; ++e) {
    float* x = &c->h_xshape[(e * m.nxs + m.xs_slot[shape]) * 8];
    if (half_sizes) { x[0] = half_sizes[3 * e]; x[1] = half_sizes[3 * e + 1]; x[2] = half_sizes[3 * e + 2]; }
    if (local_pos) { x[4] = local_pos[3 * e]; x[5] = local_pos[3 * e + 1]; x[6] = local_pos[3 * e + 2]; }
  }
  HIP_TRY(hipMemcpy2D(c->st.env + m.lay.xshape, sizeof(float) * m.lay.stride, c->h_xshape.data(), sizeof(float) * m.nxs * 8,
                      sizeof(float) * m.nxs * 8, N, hipMemcpyHostToDevice));
  return MSK_OK;
}

MSK_API int msk_set_env_masses(msk_ctx* c, int body, const float* mass, const float* inertia) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "set_env_masses before finalize");
  const DModel& m = c->model;
  if (body < 0 || body >= m.nb || m.xb_slot[body] < 0) return fail(c, MSK_ERR_INVALID, "set_env_masses: body was not declared");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t N = (size_t)m.N;
  for (size_t e = 0; e < N; ++e) {
    float* x = &c->h_xbody[(e * m.nxb + m.xb_slot[body]) * 8];
    x[0] = mass[e];
    x[1] = 1.0f / inertia[3 * e]; x[2] = 1.0f / inertia[3 * e + 1]; x[3] = 1.0f / inertia[3 * e + 2];
  }
  HIP_TRY(hipMemcpy2D(c->st.env + m.lay.xbody, sizeof(float) * m.lay.stride, c->h_xbody.data(), sizeof(float) * m.nxb * 8,
                      sizeof(float) * m.nxb * 8, N, hipMemcpyHostToDevice));
  return MSK_OK;
}

MSK_API int msk_set_solver_classes(msk_ctx* c, const int32_t caps[3]) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "set_solver_classes before finalize");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  DModel& m = c->model;
  const int fit = (m.G == 16) ? CsLds<16, 16, 16>::fit() : ((m.G == 32) ? CsLds<32, 32, 32>::fit() : CsLds<64, 64, 64>::fit());
  m.cls_cap[0] = caps[0] < fit ? caps[0] : fit;
  m.cls_cap[1] = caps[1] < MSK_CLASS2_BLOCKS ? caps[1] : MSK_CLASS2_BLOCKS;
  m.cls_cap[2] = caps[2] < MSK_CLASS2_BLOCKS ? caps[2] : MSK_CLASS2_BLOCKS;
  HIP_TRY(hipMemcpy(c->d_model, &m, sizeof(DModel), hipMemcpyHostToDevice));
  for (auto& p : c->parts) {
    if (p.d_model == c->d_model) continue;
    DModel mp = m; mp.N = p.n;
    HIP_TRY(hipMemcpy(p.d_model, &mp, sizeof(DModel), hipMemcpyHostToDevice));
  }
  return MSK_OK;
}

MSK_API int msk_get_solver_class_counts(msk_ctx* c, int32_t out[5]) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "get_solver_class_counts before finalize");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  for (int k = 0; k < MSK_SOLVE_CLASSES; ++k) out[k] = 0;
  for (auto& p : c->parts) {      /* the lists are per env partition */
    int part[MSK_SOLVE_CLASSES];
    HIP_TRY(hipMemcpy(part, p.st.cls_count, sizeof(int) * MSK_SOLVE_CLASSES, hipMemcpyDeviceToHost));
    for (int k = 0; k < MSK_SOLVE_CLASSES; ++k) out[k] += part[k];
  }
  return MSK_OK;
}

MSK_API int msk_get_env_contact_counts(msk_ctx* c, int32_t* out) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "get_env_contact_counts before finalize");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, c->st.env_ncontacts, sizeof(int) * (size_t)c->model.N, hipMemcpyDeviceToHost));
  return MSK_OK;
}

MSK_API int msk_get_contacts(msk_ctx* c, int env, int32_t* ids, float* vals, int max_points) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "get_contacts before finalize");
  const DModel& m = c->model;
  if (env < 0 || env >= m.N) return fail(c, MSK_ERR_INVALID, "bad env");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  const int np = m.np;
  if (np == 0) return 0;
  std::vector<int> cnt(m.npp);
  std::vector<float> rec((size_t)m.npp * MSK_CT_REC);
  HIP_TRY(hipMemcpy(cnt.data(), c->st.ct_cnt + (size_t)env * m.npp, sizeof(int) * m.npp, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(rec.data(), c->st.ct_rec + (size_t)env * m.npp * MSK_CT_REC, sizeof(float) * rec.size(), hipMemcpyDeviceToHost));
  int total = 0;
  for (int p = 0; p < np; ++p)
    for (int k = 0; k < cnt[p]; ++k) {
      if (total < max_points) {
        const int sa = m.pairs[p].sa, sb = m.pairs[p].sb;
        ids[3 * total] = sa; ids[3 * total + 1] = sb;
        ids[3 * total + 2] = m.shapes[sa].body * 256 + (m.shapes[sb].body & 255);
        float* v = vals + 8 * total;
        const float* r = &rec[(size_t)p * MSK_CT_REC];
        v[0] = r[4 + k * 3 + 0]; v[1] = r[4 + k * 3 + 1]; v[2] = r[4 + k * 3 + 2];
        v[3] = r[0]; v[4] = r[1]; v[5] = r[2];
        v[6] = r[16 + k]; v[7] = r[20 + k * 3 + 0];
      }
      total++;
    }
  return total;
}
