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

struct DModel {
  msk_config cfg;
  int nb, na, nd, nv, ns, np, nt, N;
  DBody bodies[MSK_MAX_BODIES];
  DShape shapes[MSK_MAX_SHAPES];
  DTendon tendons[MSK_MAX_TENDONS];
  DPair pairs[MSK_MAX_PAIRS];
  v3 verts[MSK_MAX_SHAPES * 16]; /* hull vertex pool (<= 1024 vertices per template) */
};

#define MSK_MAX_ROWS (2 * MSK_MAX_DOF + 3 * MSK_MAX_CONTACTS)

/* All device arrays of one context.  Sizes are in floats / ints per env times N. */
struct DState {
  /* persistent state */
  float *q, *qd, *qacc, *qf, *qt, *qdt;       /* [nd][N] */
  float *bpose;                                /* [nb*7][N]  px py pz qw qx qy qz */
  float *blin, *bang;                          /* [nb*3][N]  COM linear / angular velocity */
  /* per-step scratch */
  float *S;                                    /* [nd*6][N]  joint motion subspaces */
  float *comw;                                 /* [nb*3][N] */
  float *Minv;                                 /* [nd*nd][N] */
  float *Iwinv;                                /* [nb*6][N]  (dynamic actors only) */
  float *vfree;                                /* [nv][N] */
  /* contacts, one slot of <= 4 points per candidate pair (persistent: warm starting) */
  int *ct_cnt;                                 /* [np][N] */
  float *ct_pos;                               /* [np*12][N] */
  float *ct_n;                                 /* [np*3][N] */
  float *ct_sep;                               /* [np*4][N] */
  float *ct_lam;                               /* [np*12][N] */
  /* solver rows */
  float *rw_J, *rw_Y;                          /* [MSK_MAX_ROWS*nv][N] */
  float *rw_d;                                 /* [MSK_MAX_ROWS][N] */
  int *env_ncontacts;                          /* [N] */
  int *env_overflow;                           /* [1] */
  float *offsets;                              /* [3][N] */
};

#endif
