/*
 * msk_model.h — device-resident description of the env template and the SoA state layout.
 *
 * HBM layout: every per-env quantity is stored struct-of-arrays with the env
 * index fastest, arr[k * N + env].  A wavefront = 64 consecutive envs, so every load/store
 * of "field k of my env" is one fully coalesced 256-byte transaction.  The template
 * (bodies, shapes, hull vertices, candidate pairs) is shared by all envs, lives once in
 * HBM/L2 and is read through wave-uniform (scalar) loads.
 */
#ifndef MSK_MODEL_H
#define MSK_MODEL_H

#include "../../include/msk_physx.h"
#include "msk_math.h"

struct DBody {
  int kind, art, parent, jtype, dof, vofs, nograv, movable;
  pose Xp, XcInv;
  float lim_lo, lim_hi, mass;
  v3 com;
  float I6[6], Iinv6[6];
  float armature, K, D, fmax, lin_damp, ang_damp;
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
struct DPairInfo { int ba, bb; float mu; int pad; }; /* bodies of the two shapes, friction of the pair */

struct DModel {
  msk_config cfg;
  int nb, na, nd, nv, ns, np, nt, N;
  DBody bodies[MSK_MAX_BODIES];
  DShape shapes[MSK_MAX_SHAPES];
  DTendon tendons[MSK_MAX_TENDONS];
  DPair pairs[MSK_MAX_PAIRS];
  v3 verts[MSK_MAX_SHAPES * 16]; /* hull vertex pool (<= 1024 vertices per template) */
  /* lane-group solver tables: one lane per generalized coordinate k (msk_solve.h) */
  int G;                               /* lanes per env = nv padded to 16 or 32                  */
  int npp;                             /* np padded to a multiple of G (contact slot stride)     */
  unsigned long long coord_moves[MSK_MAX_NV]; /* bit b: coordinate k moves body b                */
  int coord_body[MSK_MAX_NV];          /* free body whose first coordinate is k, else -1         */
  float dof_lo[MSK_MAX_DOF], dof_hi[MSK_MAX_DOF];
  DPairInfo pinfo[MSK_MAX_PAIRS];
};

#define MSK_MAX_ROWS (2 * MSK_MAX_DOF + 3 * MSK_MAX_CONTACTS)
#define MSK_ROWS_LDS 64                        /* solver rows per env in the workgroup's LDS row pool
                                                  (shared: one env may use its neighbours' share); rows
                                                  that do not fit spill to HBM (st.ov_*)          */
/* contact slot of one candidate pair of one env: 32 floats
 *   [0..2] normal  [4+3k..] point k  [16+k] separation k  [20+3k+a] impulse k (normal, t1, t2) */
#define MSK_CT_REC 32

/* All device arrays of one context.  Sizes are in floats / ints per env times N. */
struct DState {
  /* persistent state */
  float *q, *qd, *qacc, *qf, *qt, *qdt;       /* [nd][N] */
  float *bpose;                                /* [nb*7][N]  px py pz qw qx qy qz */
  float *blin, *bang;                          /* [nb*3][N]  COM linear / angular velocity */
  /* per-step scratch */
  float *comw;                                 /* [nb*3][N] */
  /* env-major tables for the lane-group solver (lane k of an env reads a contiguous row) */
  float *Scol;                                 /* [N][G][8]  motion subspace column of coordinate k (a, l, pad) */
  float *W;                                    /* [N][G][G]  block-diagonal inverse mass matrix, zero padded    */
  float *vfree;                                /* [N][G]     unconstrained velocity                             */
  /* contacts, env-major, one slot of <= 4 points per candidate pair (persistent: warm starting) */
  int *ct_cnt;                                 /* [N][npp] */
  float *ct_rec;                               /* [N][npp][MSK_CT_REC] */
  /* solver rows that did not fit the LDS pool */
  float2 *ov_jy;                               /* [N][MSK_MAX_ROWS][G] */
  float4 *ov_rs;                               /* [N][MSK_MAX_ROWS] */
  float *ov_lam;                               /* [N][MSK_MAX_ROWS] */
  int *env_ncontacts;                          /* [N] */
  int *env_overflow;                           /* [1] */
  float *offsets;                              /* [3][N] */
};

#endif
