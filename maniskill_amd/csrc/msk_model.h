/*
 * msk_model.h — device-resident description of the env template and the SoA state layout.
 *
 * HBM layout: everything that belongs to one env is contiguous (env-major): the
 * persistent state record (EnvLayout), the solver tables, the contact slots.  The kernels map a
 * wavefront (or a 16/32-lane group) to an env, so its loads are wide and coalesced.  The template
 * (bodies, shapes, hull vertices, candidate pairs) is shared by all envs, lives once in
 * HBM/L2 and is read through wave-uniform (scalar) loads.
 */
#ifndef MSK_MODEL_H
#define MSK_MODEL_H

#include "../../include/msk_physx.h"
#include "msk_math.h"

/* Persistent state of ONE env: a contiguous record of `stride` floats (offsets in floats).
 *   q qd qacc qf qt qdt : MSK_MAX_DOF each      off : scene offset xyz (+pad)
 *   bpose : nb x 8  [px py pz qw qx qy qz pad]   blin, bang, comw : nb x 4 [x y z pad]
 * A wavefront that works on one env (or a lane group on a few) reads its record with wide
 * contiguous loads; the record of PickCube (18 bodies) is 1840 bytes. */
struct EnvLayout { int q, qd, qacc, qf, qt, qdt, off, bpose, blin, bang, comw, xshape, xbody, stride; };
/*   xshape : nxs x 8 [half sizes, pad, local position, pad]   xbody : nxb x 8 [mass, inverse principal inertia, pad]
 *   (per-env instances of declared box shapes / dynamic actors: include/msk_physx.h msk_declare_env_box / _mass) */

struct DBody {
  int kind, art, parent, jtype, dof, vofs, nograv, movable;
  int root_dof;   /* root link of a floating articulation: first of its six coordinates (angular 3, linear 3: Pluecker about the env origin), else -1 */
  pose Xp, XcInv;
  float lim_lo, lim_hi, mass;
  v3 com;
  float I6[6], Iinv6[6];
  float armature, K, D, fmax, lin_damp, ang_damp;
  float jfriction;   /* PhysxArticulationJoint.friction: coefficient of the joint-friction row (msk_solve.h) */
  int drive_accel;   /* drive mode "acceleration": gains per unit of the joint's own inertia */
  unsigned lock;     /* dynamic actors: bit k = world axis k (linear x y z, angular x y z) is locked (msk_set_locked_axes) */
};

struct DShape {
  int body, type, nverts, vbase; /* vbase: first vertex in DModel::verts */
  pose local;
  float par[3];
  v3 aabb_c, aabb_h;
  float df;
};

struct DTendon { int dof_a, dof_b; float ca, cb, rest, K, D; };
struct DPair { int sa, sb; };
/* bodies of the two shapes; dynamic and static friction and restitution of the pair (averages: PhysX's default combine mode);
 * torsional patch of the pair (the larger of the two shapes'), used when the manifold has a single point */
struct DPairInfo { int ba, bb; float mu, mu_s; float rest; float patch_r, min_patch_r; unsigned ca, cb; /* DModel::body_coords of ba, bb (0 for the static world): one load for the solver's row assembly */
                   unsigned ca_hi, cb_hi; /* ... coordinates 32 .. 63 (read by the 64-coordinate solver only) */ };

#define MSK_SOLVE_CLASSES 5
#define MSK_FRICTION_ALIGN_SPEED 1.0e-2f   /* m/s: below it a contact's friction frame is the one msk_tangents() derives from the normal (oracle: ORC_FRICTION_ALIGN_SPEED) */
#define MSK_LIMIT_SLACK 5.0e-3f    /* a joint gets a limit row while it can reach the limit in this step: distance < slack + twice what its
                                   * unconstrained velocity covers towards it in dt (solver and classifier; oracle: ORC_LIMIT_SLACK) */

struct DModel {
  msk_config cfg;
  int nb, na, nd, nv, ns, np, nt, N;
  DBody bodies[MSK_MAX_BODIES];
  DShape shapes[MSK_MAX_SHAPES];
  DTendon tendons[MSK_MAX_TENDONS];
  DPair pairs[MSK_MAX_PAIRS];
  v3 verts[MSK_MAX_SHAPES * 48]; /* hull vertex pool (<= 3072 vertices per template) */
  int nverts_total;
  EnvLayout lay;
  /* tree tables for the wave-per-env dynamics (msk_dynamics.h) */
  int depth[MSK_MAX_BODIES];           /* links: distance to the root link; actors: 0            */
  int maxdepth;
  alignas(16) unsigned char desc[MSK_MAX_BODIES][MSK_MAX_BODIES];   /* links: the link's descendants in DESCENDING body index (msk_dynamics.h phase 3) */
  unsigned char ndesc[MSK_MAX_BODIES];
  alignas(16) unsigned char path[MSK_MAX_BODIES][MSK_MAX_BODIES];   /* links: [0] the root link, [d] the ancestor at depth d, [depth] the body itself (msk_dynamics.h forward_pass) */
  int child_off[MSK_MAX_BODIES + 1];   /* CSR of child links, each list in DESCENDING body index */
  int child_idx[MSK_MAX_BODIES];
  int dof_body[MSK_MAX_DOF];           /* link whose incoming joint is dof d                     */
  /* lane-group solver tables: one lane per generalized coordinate k (msk_solve.h) */
  int G;                               /* lanes per env = nv padded to 16 or 32                  */
  int npp;                             /* np padded to a multiple of G (contact slot stride)     */
  unsigned long long coord_moves[MSK_MAX_NV]; /* bit b: coordinate k moves body b                */
  int coord_body[MSK_MAX_NV];          /* free body whose first coordinate is k, else -1         */
  int coord_root[MSK_MAX_NV];          /* floating root link whose first coordinate is k, else -1 */
  unsigned char dof_body_is_root[MSK_MAX_DOF]; /* dof d is one of a floating root's six coordinates */
  unsigned long long body_coords[MSK_MAX_BODIES]; /* bit k: coordinate k moves body b (transpose of coord_moves) */
  float dof_lo[MSK_MAX_DOF], dof_hi[MSK_MAX_DOF];
  DPairInfo pinfo[MSK_MAX_PAIRS];
  int cls_cap[MSK_SOLVE_CLASSES - 1];  /* largest block count of solver classes 0 .. 3 (the wide class takes the rest: msk_config.contact_capacity) */
  int cap_contacts, cap_blocks;        /* contact points / solver blocks an env can have (msk_config.contact_capacity) */
  unsigned long long jfric_mask;               /* bit d: joint d has a friction coefficient (a joint-friction block in every step) */
  int njfric;                          /* popcount of it */
  int has_static;                      /* some pair has static != dynamic friction: the per-pair slide state is kept (DState::ct_slip) */
  int has_tors;                        /* some pair has a torsional patch radius */
  /* per-env instances: slot of a declared box shape / dynamic actor in the env record, -1 = the template's values */
  signed char xs_slot[MSK_MAX_SHAPES], xb_slot[MSK_MAX_BODIES];
  int nxs, nxb;
};

#define MSK_MAX_ROWS (3 * MSK_MAX_BLOCKS)
/* contact slot of one candidate pair of one env: 32 floats
 *   [0..2] normal  [3] impulse of the torsional row (one-point manifolds)  [4+3k..] point k  [16+k] separation k
 *   [20+3k+a] impulse k (normal, t1, t2) */
#define MSK_CT_REC 32

/* All device arrays of one context.  Sizes are in floats / ints per env times N. */
struct DState {
  /* persistent state, env-major: [N][lay.stride] (EnvLayout) */
  float *env;
  /* env-major tables for the lane-group solver (lane k of an env reads a contiguous row) */
  float *Scol;                                 /* [N][G][8]  motion subspace column of coordinate k (a, l, pad) */
  float *W;                                    /* [N][G][G]  block-diagonal inverse mass matrix, zero padded    */
  float *vfree;                                /* [N][G]     unconstrained velocity                             */
  /* contacts, env-major, one slot of <= 4 points per candidate pair (persistent: warm starting) */
  int *ct_cnt;                                 /* [N][npp] */
  float *ct_rec;                               /* [N][npp][MSK_CT_REC] */
  /* solver work lists by LDS capacity class (msk_solve.h): class 0 = packed small launch, 1..3 = one wave per env with
   * room for cls_cap[] blocks; built by the last narrowphase launch (classify_envs) */
  int *cls_list;                               /* [MSK_SOLVE_CLASSES][N] */
  int *np_done;                                /* [N/64]: per 64-env chunk, sign-offs of the narrowphase: the broadphase subtracts the chunk's hull items, every
                                                * plane / box-box block and every hull item adds one; whoever brings it to the number of plane / box-box
                                                * blocks classifies the chunk and resets it */
  int *cls_count;                              /* [MSK_SOLVE_CLASSES]; zeroed by k_dynamics (its broadphase tail) of the same substep */
  float *a_scratch;                            /* [solver workers][9 * 64 * 64]: A images of class-3 envs */
  float *wide_scratch;                         /* [wide workers][WideScratch::TOTAL]: Y and A images of the wide class (msk_config.contact_capacity = 1), else null */
  int wide_workers;
  /* narrowphase work lists per env and type (plane / box-box / GJK): surviving pair indices in pair order */
  int *np_count;                               /* [N][4] */
  int *np_items;                               /* [N][3][np] */
  int *hq_items;                               /* [N * np] the launch's hull (GJK) items of ALL envs, env * np + pair: one queue, dealt out to the hull
                                                * blocks of the narrowphase whatever env group they come from */
  int *hq_count;                               /* [1] filled by the broadphase (atomics), zeroed by the solver launch */
  float *ext_wrench;                           /* [N][nb][8] external force (0..2) / torque (4..6) of the next step; consumed and cleared by k_dynamics */
  long long *dbg;                              /* [N][8] phase time stamps (MSK_PROFILE_PHASES builds only) */
  int *env_ncontacts;                          /* [N] */
  int *ct_total;                               /* [N] constraint blocks of the env's contact slots: sum of ct_blocks() over its ct_cnt row, kept in step with
                                                * every write to it (device-scope atomics in the narrowphase): what the classification reads instead of the row */
  int *env_overflow;                           /* [1] */
  /* force-limited drives that k_dynamics hands to the solver as soft rows (oracle: orc_scratch.drv_*) */
  unsigned long long *drv_mask;                /* [N] bit d: the drive of joint d is a solver row in this substep */
  float *drv;                                  /* [N][G][4]: compliance 1 / g, velocity bias, impulse limit, pad */
  unsigned char *ct_slip;                      /* [N][npp] 1: the pair slides, its friction cone is the dynamic one (has_static only, else null) */
  unsigned long long *gjk_cache;               /* [N][npp] the simplex GJK ended on last step for this (env, pair): count + four vertex-number pairs, 0 = none (msk_collide.h) */
  float *jforce;                               /* [N][nb][6] link incoming joint wrenches of the last step (njfric > 0 only, else null): k_link_forces */
};

/* constraint blocks a pair with n contact points brings to its env's solve: one per point, and one for the torsional row of a
 * one-point manifold whose pair has a patch radius (what DState::ct_total adds up) */
MSK_DEV int ct_blocks(const DModel* m, int pair, int n) {
  return n + ((n == 1 && m->has_tors && (m->pinfo[pair].patch_r > 0.0f || m->pinfo[pair].min_patch_r > 0.0f)) ? 1 : 0);
}

/* accessors of the env record E (EnvLayout): scalar fields by offset, poses 8 floats, vectors 4 floats */
#define EREC(st, m, e) ((st).env + (size_t)(e) * (size_t)(m)->lay.stride)

MSK_DEV pose load_pose(const float* E, int ofs, int body) {
  const float* r = E + ofs + body * 8;
  pose p;
  p.p = v3_make(r[0], r[1], r[2]);
  p.q = quat_make(r[3], r[4], r[5], r[6]);
  return p;
}
MSK_DEV void store_pose(float* E, int ofs, int body, pose p) {
  float* r = E + ofs + body * 8;
  r[0] = p.p.x; r[1] = p.p.y; r[2] = p.p.z; r[3] = p.q.w; r[4] = p.q.x; r[5] = p.q.y; r[6] = p.q.z;
}
MSK_DEV v3 load_v3(const float* E, int ofs, int k) { const float* r = E + ofs + k * 4; return v3_make(r[0], r[1], r[2]); }
MSK_DEV void store_v3(float* E, int ofs, int k, v3 v) { float* r = E + ofs + k * 4; r[0] = v.x; r[1] = v.y; r[2] = v.z; }

#endif
