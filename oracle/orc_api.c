/*
 * orc_api.c — the oracle behind the same entry points as include/msk_physx.h, with the
 * prefix orc_ and host-memory buffers.  TEST INFRASTRUCTURE ONLY: loaded by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by maniskill_amd/.
 */
#include "orc_sim.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))

static pose pose_from7(const float* p) {
  pose r;
  r.p = v3_make(p[0], p[1], p[2]);
  r.q = quat_normalize(quat_make(p[3], p[4], p[5], p[6]));
  return r;
}

static int fail(orc_ctx* c, int code, const char* msg) {
  snprintf(c->err, sizeof(c->err), "%s", msg);
  return code;
}

ORC_EXPORT orc_ctx* orc_create(int device, const msk_config* cfg) {
  (void)device;
  orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
  c->cfg = *cfg;
  c->cap_contacts = cfg->contact_capacity ? MSK_MAX_CONTACTS_WIDE : MSK_MAX_CONTACTS;
  c->cap_blocks = cfg->contact_capacity ? MSK_MAX_BLOCKS_WIDE : MSK_MAX_BLOCKS;
  for (int i = 0; i < MSK_MAX_SHAPES; ++i) c->xs_slot[i] = -1;
  for (int i = 0; i < MSK_MAX_BODIES; ++i) c->xb_slot[i] = -1;
  if (cfg->sleep_threshold > 0.0f)
    snprintf(c->warn + strlen(c->warn), sizeof(c->warn) - strlen(c->warn), "sleep_threshold=%g accepted, not modelled: bodies never sleep\n", cfg->sleep_threshold);
  if (!cfg->enable_pcm)
    snprintf(c->warn + strlen(c->warn), sizeof(c->warn) - strlen(c->warn), "enable_pcm=0 accepted, no effect: contact manifolds are generated one-shot every step\n");
  return c;
}

ORC_EXPORT void orc_destroy(orc_ctx* c) {
  if (!c) return;
  free(c->envs);
  free(c->offsets);
  free(c->xshape);
  free(c->xbody);
  for (int i = 0; i < MSK_BUF_COUNT; ++i) free((c->buf_bound && i <= MSK_BUF_RIGID_BODY_TORQUE) ? c->buf_own[i] : c->buf[i]);
  free(c->wrench);
  free(c->gjk_cache);
  for (int i = 0; i < c->nqueries; ++i) { free(c->queries[i].pairs); free(c->queries[i].out); }
  free(c);
}

ORC_EXPORT const char* orc_last_error(orc_ctx* c) { return c->err; }
ORC_EXPORT const char* orc_warnings(orc_ctx* c) { return c ? c->warn : ""; }

ORC_EXPORT int orc_add_articulation(orc_ctx* c, const float root_pose[7]) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "add_articulation after finalize");
  if (c->na >= 8) return fail(c, MSK_ERR_CAPACITY, "too many articulations");
  c->art_root[c->na] = -1;
  c->art_dof0[c->na] = c->ndof;
  c->art_ndof[c->na] = 0;
  c->art_floating[c->na] = 0;
  /* root pose is attached to the root link when it is added */
  orc_body* tmp = &c->bodies[MSK_MAX_BODIES - 1];
  tmp->init_pose = pose_from7(root_pose);
  return c->na++;
}

ORC_EXPORT int orc_add_link(orc_ctx* c, int art, int parent_body, int joint_type, const float pose_in_parent[7],
                            const float pose_in_child[7], float limit_lo, float limit_hi, float mass,
                            const float com[3], const float inertia6[6], int disable_gravity, float armature,
                            float joint_friction) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "add_link after finalize");
  if (c->nb >= MSK_MAX_BODIES - 1) return fail(c, MSK_ERR_CAPACITY, "too many bodies");
  if (art != c->na - 1) return fail(c, MSK_ERR_INVALID, "links must be added to the most recent articulation");
  orc_body* b = &c->bodies[c->nb];
  pose root = c->bodies[MSK_MAX_BODIES - 1].init_pose;
  memset(b, 0, sizeof(*b));
  b->kind = MSK_BODY_LINK;
  b->art = art;
  b->parent = parent_body;
  b->jtype = (parent_body < 0) ? MSK_JOINT_FIXED : joint_type;
  b->Xp = pose_from7(pose_in_parent);
  b->XcInv = pose_inv(pose_from7(pose_in_child));
  b->lim_lo = limit_lo; b->lim_hi = limit_hi;
  b->mass = mass;
  b->com = v3_make(com[0], com[1], com[2]);
  memcpy(b->I6, inertia6, sizeof(b->I6));
  b->nograv = disable_gravity;
  b->armature = armature; b->jfriction = joint_friction;
  b->dof = -1; b->vofs = -1; b->root_dof = -1;
  if (parent_body < 0) {
    b->init_pose = root;
    c->art_root[art] = c->nb;
  } else {
    if (parent_body >= c->nb || c->bodies[parent_body].art != art) return fail(c, MSK_ERR_INVALID, "bad parent link");
    if (b->jtype != MSK_JOINT_FIXED) {
      if (c->ndof >= MSK_MAX_DOF - 1) return fail(c, MSK_ERR_CAPACITY, "too many dofs (63 per sub-scene)");
      b->dof = c->ndof++;
      c->art_ndof[art]++;
    }
    b->movable = (b->dof >= 0) || c->bodies[parent_body].movable;
  }
  return c->nb++;
}

ORC_EXPORT int orc_set_articulation_floating(orc_ctx* c, int art) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "set_articulation_floating after finalize");
  if (art < 0 || art >= c->na) return fail(c, MSK_ERR_INVALID, "set_articulation_floating: no such articulation");
  c->art_floating[art] = 1;
  return MSK_OK;
}

ORC_EXPORT int orc_set_locked_axes(orc_ctx* c, int body, uint32_t mask) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "set_locked_axes after finalize");
  if (body < 0 || body >= c->nb || c->bodies[body].kind != MSK_BODY_DYNAMIC) return fail(c, MSK_ERR_INVALID, "set_locked_axes: not a dynamic actor");
  c->bodies[body].lock = mask & 63u;
  return MSK_OK;
}

ORC_EXPORT int orc_set_drive(orc_ctx* c, int link_body, float K, float D, float force_limit, int mode_acc) {
  if (link_body < 0 || link_body >= c->nb || c->bodies[link_body].dof < 0) return fail(c, MSK_ERR_INVALID, "set_drive: not an active joint");
  orc_body* b = &c->bodies[link_body];
  b->K = K; b->D = D; b->fmax = force_limit; b->drive_accel = mode_acc;
  return MSK_OK;
}

ORC_EXPORT int orc_add_tendon(orc_ctx* c, int link_a, int link_b, float ca, float cb, float rest, float K, float D) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "add_tendon after finalize");
  if (c->nt >= MSK_MAX_TENDONS) return fail(c, MSK_ERR_CAPACITY, "too many tendons");
  if (link_a < 0 || link_b < 0 || link_a >= c->nb || link_b >= c->nb) return fail(c, MSK_ERR_INVALID, "bad tendon link");
  if (c->bodies[link_a].dof < 0 || c->bodies[link_b].dof < 0) return fail(c, MSK_ERR_INVALID, "tendon on a fixed joint");
  orc_tendon* t = &c->tendons[c->nt];
  t->dof_a = c->bodies[link_a].dof; t->dof_b = c->bodies[link_b].dof;
  t->ca = ca; t->cb = cb; t->rest = rest; t->K = K; t->D = D;
  return c->nt++;
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

ORC_EXPORT int orc_add_actor(orc_ctx* c, int kind, const float pose7[7], float mass, const float com[3],
                             const float inertia6[6], float lin_damp, float ang_damp, int disable_gravity) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "add_actor after finalize");
  if (c->nb >= MSK_MAX_BODIES - 1) return fail(c, MSK_ERR_CAPACITY, "too many bodies");
  if (kind != MSK_BODY_KINEMATIC && kind != MSK_BODY_DYNAMIC) return fail(c, MSK_ERR_INVALID, "bad actor kind");
  orc_body* b = &c->bodies[c->nb];
  memset(b, 0, sizeof(*b));
  b->kind = kind; b->art = -1; b->parent = -1; b->dof = -1; b->vofs = -1;
  b->init_pose = pose_from7(pose7);
  b->mass = mass;
  b->com = v3_make(com[0], com[1], com[2]);
  memcpy(b->I6, inertia6, sizeof(b->I6));
  if (kind == MSK_BODY_DYNAMIC) {
    if (!(mass > 0.0f)) return fail(c, MSK_ERR_INVALID, "dynamic actor needs positive mass");
    sym6_inverse(b->I6, b->Iinv6);
  }
  b->lin_damp = lin_damp; b->ang_damp = ang_damp; b->nograv = disable_gravity;
  b->movable = kind == MSK_BODY_DYNAMIC;
  return c->nb++;
}

ORC_EXPORT int orc_add_shape(orc_ctx* c, int body, int type, const float local_pose[7], const float params[3],
                             const float* verts, int nverts, float sf, float df, float rest, const uint32_t groups[4],
                             float patch_radius, float min_patch_radius) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "add_shape after finalize");
  if (c->ns >= MSK_MAX_SHAPES) return fail(c, MSK_ERR_CAPACITY, "too many shapes");
  if (body >= c->nb) return fail(c, MSK_ERR_INVALID, "bad body");
  orc_shape* s = &c->shapes[c->ns];
  memset(s, 0, sizeof(*s));
  s->body = body; s->type = type;
  s->local = pose_from7(local_pose);
  s->par[0] = params[0]; s->par[1] = params[1]; s->par[2] = params[2];
  s->sf = sf; s->df = df; s->rest = rest;
  memcpy(s->g, groups, sizeof(s->g));
  s->patch_r = patch_radius; s->min_patch_r = min_patch_radius;
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
    v3 lo = v3_make(3e38f, 3e38f, 3e38f), hi = v3_make(-3e38f, -3e38f, -3e38f);
    for (int i = 0; i < nverts; ++i) {
      s->verts[i] = v3_make(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
      lo = v3_make(fminf(lo.x, s->verts[i].x), fminf(lo.y, s->verts[i].y), fminf(lo.z, s->verts[i].z));
      hi = v3_make(fmaxf(hi.x, s->verts[i].x), fmaxf(hi.y, s->verts[i].y), fmaxf(hi.z, s->verts[i].z));
    }
    s->aabb_c = v3_scale(v3_add(lo, hi), 0.5f);
    s->aabb_h = v3_add(v3_scale(v3_sub(hi, lo), 0.5f), v3_make(s->par[0], s->par[0], s->par[0]));   /* core box + rounding radius */
  } else if (type == MSK_SHAPE_BOX) {
    s->aabb_c = v3_make(0, 0, 0);
    s->aabb_h = v3_make(params[0], params[1], params[2]);
  } else if (type == MSK_SHAPE_PLANE) {
    if (body >= 0) return fail(c, MSK_ERR_INVALID, "planes must be static");
  } else {
    return fail(c, MSK_ERR_INVALID, "shape type not supported");
  }
  return c->ns++;
}

ORC_EXPORT int orc_disable_collision(orc_ctx* c, int a, int b) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "disable_collision after finalize");
  if (c->ndisabled >= 256) return fail(c, MSK_ERR_CAPACITY, "too many disabled pairs");
  c->disabled[c->ndisabled][0] = a; c->disabled[c->ndisabled][1] = b;
  c->ndisabled++;
  return MSK_OK;
}

static int body_movable(const orc_ctx* c, int b) { return b >= 0 && c->bodies[b].movable; }

static int pair_enabled(const orc_ctx* c, const orc_shape* A, const orc_shape* B) {
  if (A->body == B->body) return 0;
  if (!body_movable(c, A->body) && !body_movable(c, B->body)) return 0;
  if (A->g[2] & B->g[2]) return 0;
  if (!((A->g[0] & B->g[1]) || (A->g[1] & B->g[0]))) return 0;
  if (A->body >= 0 && B->body >= 0) {
    const orc_body* ba = &c->bodies[A->body];
    const orc_body* bb = &c->bodies[B->body];
    if (ba->kind == MSK_BODY_LINK && bb->kind == MSK_BODY_LINK && ba->art == bb->art)
      if (ba->parent == B->body || bb->parent == A->body) return 0;
    for (int i = 0; i < c->ndisabled; ++i)
      if ((c->disabled[i][0] == A->body && c->disabled[i][1] == B->body) ||
          (c->disabled[i][0] == B->body && c->disabled[i][1] == A->body))
        return 0;
  }
  return 1;
}

static void env_reset(const orc_ctx* c, orc_env* e) {
  memset(e, 0, sizeof(*e));
  for (int i = 0; i < c->nb; ++i) {
    e->bpose[i] = c->bodies[i].init_pose;
    if (c->bodies[i].kind == MSK_BODY_LINK && c->bodies[i].parent >= 0) e->bpose[i].q = quat_make(1, 0, 0, 0);
  }
  orc_forward_kinematics(c, e);
}

ORC_EXPORT int orc_finalize(orc_ctx* c, int num_envs) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "finalize twice");
  if (!c->cfg.enable_tgs) return fail(c, MSK_ERR_INVALID, "only the TGS solver is implemented");
  /* floating roots: six coordinates each, behind the joint dofs (qpos / qvel keep the joints-only layout) */
  for (int a = 0; a < c->na; ++a) {
    if (!c->art_floating[a] || c->art_root[a] < 0) continue;
    if (c->ndof + 6 > MSK_MAX_DOF - 1) return fail(c, MSK_ERR_CAPACITY, "too many dofs (63 per sub-scene, 6 per floating root)");
    c->bodies[c->art_root[a]].root_dof = c->ndof;
    c->ndof += 6;
  }
  for (int i = 0; i < c->nb; ++i) { /* links below a floating root move even behind fixed joints */
    orc_body* b = &c->bodies[i];
    if (b->kind != MSK_BODY_LINK) continue;
    if (b->parent < 0) b->movable = b->root_dof >= 0;
    else b->movable = (b->dof >= 0) || c->bodies[b->parent].movable;
  }
  c->nv = c->ndof;
  for (int i = 0; i < c->nb; ++i)
    if (c->bodies[i].kind == MSK_BODY_DYNAMIC) { c->bodies[i].vofs = c->nv; c->nv += 6; }
  if (c->nv > MSK_MAX_NV) return fail(c, MSK_ERR_CAPACITY, "generalized velocity too large");
  c->max_dof = 0;
  for (int a = 0; a < c->na; ++a) {
    if (c->art_root[a] < 0) return fail(c, MSK_ERR_INVALID, "articulation without links");
    if (c->art_ndof[a] > c->max_dof) c->max_dof = c->art_ndof[a];
  }
  c->art_pitch = c->max_dof > 0 ? c->max_dof : 1;
  c->max_links = 0;
  {
    int count[8] = {0};
    for (int i = 0; i < c->nb; ++i) {
      c->link_slot[i] = -1;
      if (c->bodies[i].kind != MSK_BODY_LINK) continue;
      c->link_slot[i] = count[c->bodies[i].art]++;
      if (count[c->bodies[i].art] > c->max_links) c->max_links = count[c->bodies[i].art];
    }
  }
  c->npairs = 0;
  for (int i = 0; i < c->ns; ++i)
    for (int j = i + 1; j < c->ns; ++j)
      if (pair_enabled(c, &c->shapes[i], &c->shapes[j])) {
        if (c->npairs >= MSK_MAX_PAIRS) return fail(c, MSK_ERR_CAPACITY, "too many candidate pairs");
        c->pairs[c->npairs].sa = i; c->pairs[c->npairs].sb = j;
        c->npairs++;
      }
  c->num_envs = num_envs;
  c->envs = (orc_env*)calloc((size_t)num_envs, sizeof(orc_env));
  c->offsets = (float*)calloc((size_t)num_envs * 3, sizeof(float));
  if (c->npairs > 0) c->gjk_cache = (uint64_t*)calloc((size_t)num_envs * c->npairs, sizeof(uint64_t));
  /* per-env instance parameters start at the template's values */
  if (c->nxs > 0) c->xshape = (float*)calloc((size_t)num_envs * c->nxs * 8, sizeof(float));
  if (c->nxb > 0) c->xbody = (float*)calloc((size_t)num_envs * c->nxb * 8, sizeof(float));
  for (int e = 0; e < num_envs; ++e) {
    for (int si = 0; si < c->ns; ++si) {
      if (c->xs_slot[si] < 0) continue;
      float* x = c->xshape + ((size_t)e * c->nxs + c->xs_slot[si]) * 8;
      const orc_shape* sh = &c->shapes[si];
      x[0] = sh->par[0]; x[1] = sh->par[1]; x[2] = sh->par[2];
      x[4] = sh->local.p.x; x[5] = sh->local.p.y; x[6] = sh->local.p.z;
    }
    for (int bi = 0; bi < c->nb; ++bi) {
      if (c->xb_slot[bi] < 0) continue;
      float* x = c->xbody + ((size_t)e * c->nxb + c->xb_slot[bi]) * 8;
      const orc_body* b = &c->bodies[bi];
      x[0] = b->mass; x[1] = b->Iinv6[0]; x[2] = b->Iinv6[1]; x[3] = b->Iinv6[2];
    }
  }
  for (int e = 0; e < num_envs; ++e) env_reset(c, &c->envs[e]);
  size_t nrb = (size_t)num_envs * c->nb * 13;
  size_t nart = (size_t)num_envs * (c->na > 0 ? c->na : 1) * (c->max_dof > 0 ? c->max_dof : 1);
  c->buf[MSK_BUF_RIGID_BODY_DATA] = (float*)calloc(nrb, sizeof(float));
  for (int b = MSK_BUF_ART_QPOS; b <= MSK_BUF_ART_TARGET_QVEL; ++b) c->buf[b] = (float*)calloc(nart, sizeof(float));
  c->buf[MSK_BUF_RIGID_BODY_FORCE] = (float*)calloc((size_t)num_envs * c->nb * 4, sizeof(float));
  c->buf[MSK_BUF_ART_LINK_JOINT_FORCES] = (float*)calloc((size_t)num_envs * (c->na > 0 ? c->na : 1) * (c->max_links > 0 ? c->max_links : 1) * 6, sizeof(float));
  c->buf[MSK_BUF_RIGID_BODY_TORQUE] = (float*)calloc((size_t)num_envs * c->nb * 4, sizeof(float));
  c->wrench = (float*)calloc((size_t)num_envs * c->nb * 8, sizeof(float));
  c->wrench_pending = 0;
  c->finalized = 1;
  return MSK_OK;
}

ORC_EXPORT int orc_set_scene_offsets(orc_ctx* c, const float* offsets) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "set_scene_offsets before finalize");
  memcpy(c->offsets, offsets, sizeof(float) * 3 * (size_t)c->num_envs);
  return MSK_OK;
}

ORC_EXPORT void* orc_buffer(orc_ctx* c, int id, int64_t shape[2]) {
  if (!c->finalized || id < 0 || id >= MSK_BUF_COUNT) return NULL;
  if (id == MSK_BUF_RIGID_BODY_DATA) { shape[0] = (int64_t)c->num_envs * c->nb; shape[1] = 13; }
  else if (id == MSK_BUF_RIGID_BODY_FORCE || id == MSK_BUF_RIGID_BODY_TORQUE) { shape[0] = (int64_t)c->num_envs * c->nb; shape[1] = 4; }
  else if (id == MSK_BUF_ART_LINK_JOINT_FORCES) { shape[0] = (int64_t)c->num_envs * c->na * c->max_links; shape[1] = 6; }
  else { shape[0] = (int64_t)c->num_envs * c->na; shape[1] = c->art_pitch; }
  return c->buf[id];
}

static float* art_row(orc_ctx* c, int buf, int env, int art) {
  return c->buf[buf] + ((size_t)env * c->na + art) * c->art_pitch;
}

/* msk_bind_buffers / msk_batch of include/msk_physx.h (host memory here; the batch is a loop) */
ORC_EXPORT int orc_bind_buffers(orc_ctx* c, void* const ptrs[9], int64_t art_pitch) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "bind_buffers before finalize");
  if (art_pitch < (c->max_dof > 0 ? c->max_dof : 1)) return fail(c, MSK_ERR_INVALID, "bind_buffers: art_pitch below max_dof");
  for (int id = 0; id <= MSK_BUF_RIGID_BODY_TORQUE; ++id)
    if (!ptrs[id]) return fail(c, MSK_ERR_INVALID, "bind_buffers: null pointer");
  for (int id = 0; id <= MSK_BUF_RIGID_BODY_TORQUE; ++id) {
    if (!c->buf_bound) c->buf_own[id] = c->buf[id];   /* freed by orc_destroy */
    c->buf[id] = (float*)ptrs[id];
  }
  c->buf_bound = 1;
  c->art_pitch = (int)art_pitch;
  return MSK_OK;
}

static void apply_env(orc_ctx* c, int e, uint32_t mask);
static void fetch_env(orc_ctx* c, int e, uint32_t mask);

ORC_EXPORT int orc_apply(orc_ctx* c, uint32_t mask, void* stream) {
  (void)stream;
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "apply before finalize");
  if (mask & (MSK_APPLY_RIGID_FORCE | MSK_APPLY_RIGID_TORQUE)) { /* external wrench of the next step */
    const size_t rows = (size_t)c->num_envs * c->nb;
    for (size_t r = 0; r < rows; ++r)
      for (int k = 0; k < 3; ++k) {
        if (mask & MSK_APPLY_RIGID_FORCE) c->wrench[r * 8 + k] = c->buf[MSK_BUF_RIGID_BODY_FORCE][r * 4 + k];
        if (mask & MSK_APPLY_RIGID_TORQUE) c->wrench[r * 8 + 4 + k] = c->buf[MSK_BUF_RIGID_BODY_TORQUE][r * 4 + k];
      }
    c->wrench_pending = 1;
  }
  for (int e = 0; e < c->num_envs; ++e) apply_env(c, e, mask);
  return MSK_OK;
}

/* the env-record half of orc_apply for one env (orc_reset_masked applies the envs its mask names, nothing else) */
static void apply_env(orc_ctx* c, int e, uint32_t mask) {
  {
    orc_env* env = &c->envs[e];
    const float* off = c->offsets + 3 * e;
    int teleported = 0; /* a pose or joint position was overwritten: the env's contact cache is stale */
    for (int i = 0; i < c->nb; ++i) {
      const orc_body* b = &c->bodies[i];
      const float* r = c->buf[MSK_BUF_RIGID_BODY_DATA] + ((size_t)e * c->nb + i) * 13;
      int is_root = b->kind == MSK_BODY_LINK && b->parent < 0;
      if ((b->kind != MSK_BODY_LINK && (mask & MSK_APPLY_RIGID_DATA)) || (is_root && (mask & MSK_APPLY_ART_ROOT_POSE))) {
        /* rows the caller did not touch since the last fetch are left alone: (p + off) - off and
         * re-normalisation are not exact in fp32, and apply must not perturb untouched envs */
        const pose cur = env->bpose[i];
        int same = (r[0] == cur.p.x + off[0]) && (r[1] == cur.p.y + off[1]) && (r[2] == cur.p.z + off[2]) &&
                   (r[3] == cur.q.w) && (r[4] == cur.q.x) && (r[5] == cur.q.y) && (r[6] == cur.q.z);
        if (!same) {
          env->bpose[i].p = v3_make(r[0] - off[0], r[1] - off[1], r[2] - off[2]);
          env->bpose[i].q = quat_normalize(quat_make(r[3], r[4], r[5], r[6]));
          teleported = 1;
        }
        if (b->kind == MSK_BODY_DYNAMIC) {
          env->blin[i] = v3_make(r[7], r[8], r[9]);
          env->bang[i] = v3_make(r[10], r[11], r[12]);
        }
      }
      if (is_root && b->root_dof >= 0 && (mask & MSK_APPLY_ART_ROOT_VELOCITY)) { /* the root's six coordinates ARE (v_com, omega) */
        float* qd = env->qd + b->root_dof;
        for (int k = 0; k < 6; ++k) qd[k] = r[7 + k];
      }
    }
    for (int a = 0; a < c->na; ++a)
      for (int j = 0; j < c->art_ndof[a]; ++j) {
        int d = c->art_dof0[a] + j;
        if (mask & MSK_APPLY_ART_QPOS) {
          const float nq = art_row(c, MSK_BUF_ART_QPOS, e, a)[j];
          if (nq != env->q[d]) teleported = 1;
          env->q[d] = nq;
        }
        if (mask & MSK_APPLY_ART_QVEL) env->qd[d] = art_row(c, MSK_BUF_ART_QVEL, e, a)[j];
        if (mask & MSK_APPLY_ART_QF) env->qf[d] = art_row(c, MSK_BUF_ART_QF, e, a)[j];
        if (mask & MSK_APPLY_ART_TARGET_QPOS) env->qt[d] = art_row(c, MSK_BUF_ART_TARGET_QPOS, e, a)[j];
        if (mask & MSK_APPLY_ART_TARGET_QVEL) env->qdt[d] = art_row(c, MSK_BUF_ART_TARGET_QVEL, e, a)[j];
      }
    if (teleported) { /* no warm start across a teleport: replays from a state are reproducible */
      env->ncontacts = 0;
      if (c->gjk_cache) memset(c->gjk_cache + (size_t)e * c->npairs, 0, (size_t)c->npairs * sizeof(uint64_t));
    }
  }
}

ORC_EXPORT int orc_update_kinematics(orc_ctx* c, void* stream) {
  (void)stream;
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "update_kinematics before finalize");
  for (int e = 0; e < c->num_envs; ++e) orc_forward_kinematics(c, &c->envs[e]);
  return MSK_OK;
}

ORC_EXPORT int orc_fetch(orc_ctx* c, uint32_t mask, void* stream) {
  (void)stream;
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "fetch before finalize");
  for (int e = 0; e < c->num_envs; ++e) fetch_env(c, e, mask);
  return MSK_OK;
}

static void fetch_env(orc_ctx* c, int e, uint32_t mask) {
  {
    const orc_env* env = &c->envs[e];
    const float* off = c->offsets + 3 * e;
    if (mask & MSK_FETCH_RIGID_DATA)
      for (int i = 0; i < c->nb; ++i) {
        float* r = c->buf[MSK_BUF_RIGID_BODY_DATA] + ((size_t)e * c->nb + i) * 13;
        r[0] = env->bpose[i].p.x + off[0]; r[1] = env->bpose[i].p.y + off[1]; r[2] = env->bpose[i].p.z + off[2];
        r[3] = env->bpose[i].q.w; r[4] = env->bpose[i].q.x; r[5] = env->bpose[i].q.y; r[6] = env->bpose[i].q.z;
        r[7] = env->blin[i].x; r[8] = env->blin[i].y; r[9] = env->blin[i].z;
        r[10] = env->bang[i].x; r[11] = env->bang[i].y; r[12] = env->bang[i].z;
      }
    for (int a = 0; a < c->na; ++a)
      for (int j = 0; j < c->art_ndof[a]; ++j) {
        int d = c->art_dof0[a] + j;
        if (mask & MSK_FETCH_ART_QPOS) art_row(c, MSK_BUF_ART_QPOS, e, a)[j] = env->q[d];
        if (mask & MSK_FETCH_ART_QVEL) art_row(c, MSK_BUF_ART_QVEL, e, a)[j] = env->qd[d];
        if (mask & MSK_FETCH_ART_QACC) art_row(c, MSK_BUF_ART_QACC, e, a)[j] = env->qacc[d];
        if (mask & MSK_FETCH_ART_TARGETS) {
          art_row(c, MSK_BUF_ART_TARGET_QPOS, e, a)[j] = env->qt[d];
          art_row(c, MSK_BUF_ART_TARGET_QVEL, e, a)[j] = env->qdt[d];
        }
      }
    if (mask & MSK_FETCH_ART_LINK_FORCES) {
      float w[MSK_MAX_BODIES * 6];
      orc_link_joint_forces(c, &c->envs[e], w);
      for (int i = 0; i < c->nb; ++i)
        if (c->link_slot[i] >= 0)
          memcpy(c->buf[MSK_BUF_ART_LINK_JOINT_FORCES] + (((size_t)e * c->na + c->bodies[i].art) * c->max_links + c->link_slot[i]) * 6,
                 w + 6 * i, sizeof(float) * 6);
    }
  }
}

/* msk_reset_masked (include/msk_physx.h): the envs a mask names are fetched, overwritten from their prepared episode image and applied -- the host-side
 * partial reset of the reference (BaseEnv.reset with env_idx: masked writes through the torch views, then _gpu_apply_all: envs/sapien_env.py:857-978),
 * restricted to the named envs; the others are not touched (orc_apply leaves rows alone that did not change). */
ORC_EXPORT int orc_reset_masked(orc_ctx* c, const uint8_t* mask, const float* image, int slots, const int32_t* ent, int nent, int32_t* episode,
                                int32_t* elapsed, void* stream) {
  (void)stream;
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "reset_masked before finalize");
  if (!mask || !image || !ent || !episode || slots < 1 || nent < 0) return fail(c, MSK_ERR_INVALID, "reset_masked: null argument or empty ring");
  const uint32_t fetch_mask = MSK_FETCH_RIGID_DATA | MSK_FETCH_ART_QPOS | MSK_FETCH_ART_QVEL | MSK_FETCH_ART_QACC | MSK_FETCH_ART_TARGETS;
  const uint32_t apply_mask = MSK_APPLY_RIGID_DATA | MSK_APPLY_ART_ROOT_POSE | MSK_APPLY_ART_QPOS | MSK_APPLY_ART_QVEL | MSK_APPLY_ART_QF | MSK_APPLY_ART_TARGET_QPOS |
                              MSK_APPLY_ART_TARGET_QVEL;
  static const int ids[5] = {MSK_BUF_RIGID_BODY_DATA, MSK_BUF_ART_QPOS, MSK_BUF_ART_QVEL, MSK_BUF_ART_TARGET_QPOS, MSK_BUF_ART_TARGET_QVEL};
  for (int e = 0; e < c->num_envs; ++e) {
    if (!mask[e]) continue;
    fetch_env(c, e, fetch_mask);
    const int ep = episode[e];
    const float* img = image + ((size_t)e * slots + (size_t)(ep % slots)) * nent;
    for (int t = 0; t < nent; ++t) {
      const int which = ent[t] >> 24, word = ent[t] & 0xFFFFFF;
      if (which < 0 || which > 4) return fail(c, MSK_ERR_INVALID, "reset_masked: unknown buffer in an image entry");
      float* base = which == 0 ? c->buf[ids[0]] + (size_t)e * c->nb * 13 : art_row(c, ids[which], e, 0);
      base[word] = img[t];
    }
    apply_env(c, e, apply_mask);
    orc_forward_kinematics(c, &c->envs[e]);   /* (the HIP library leaves the frames to the next fetch / observe: same values) */
    episode[e] = ep + 1;
    if (elapsed) elapsed[e] = 0;
  }
  return MSK_OK;
}

/* msk_episode_book_step (include/msk_physx.h): ManiSkillVectorEnv.step's episode book-keeping (vector/wrappers/gymnasium.py:127-176) -- returns += rew,
 * success_once |= success, the report (clones of the book before it is cleared), terminations cleared when they are ignored, dones = terminated | truncated,
 * dones.any(), and the book of the envs about to be reset cleared (:104-125) -- env by env */
ORC_EXPORT int orc_episode_book_step(orc_ctx* c, int n, const msk_episode_book* b, void* stream) {
  (void)stream;
  if (n < 1 || !b || !b->terminated || !b->truncated || !b->out_terminated || !b->out_done || !b->any_done) return fail(c, MSK_ERR_INVALID, "episode_book_step: missing argument");
  if (b->record_metrics && (!b->reward || !b->elapsed || !b->returns || !b->out_return || !b->out_episode_len || !b->out_reward))
    return fail(c, MSK_ERR_INVALID, "episode_book_step: record_metrics needs reward, elapsed, returns and their outputs");
  int any = 0;
  for (int e = 0; e < n; ++e) {
    const int term = !b->ignore_terminations && b->terminated[(size_t)e * b->terminated_stride] != 0;
    const int done = term || b->truncated[(size_t)e * b->truncated_stride] != 0;
    if (b->record_metrics) {
      const float ret = b->returns[e] + b->reward[e];
      const int len = b->elapsed[e];
      b->out_return[e] = ret;
      b->out_episode_len[e] = len;
      b->out_reward[e] = ret / (float)len;
      b->returns[e] = (done && b->clear_done) ? 0.0f : ret;
      if (b->success) {
        const int now = b->success[(size_t)e * b->success_stride] != 0, once = b->success_once[e] != 0 || now;
        b->out_success_once[e] = (uint8_t)once;
        if (b->ignore_terminations) b->out_success_at_end[e] = (uint8_t)now;
        b->success_once[e] = (uint8_t)(once && !(done && b->clear_done));
      }
      if (b->fail) {
        const int now = b->fail[(size_t)e * b->fail_stride] != 0, once = b->fail_once[e] != 0 || now;
        b->out_fail_once[e] = (uint8_t)once;
        if (b->ignore_terminations) b->out_fail_at_end[e] = (uint8_t)now;
        b->fail_once[e] = (uint8_t)(once && !(done && b->clear_done));
      }
    }
    b->out_terminated[e] = (uint8_t)term;
    b->out_done[e] = (uint8_t)done;
    any |= done;
  }
  b->any_done[0] = any;
  return MSK_OK;
}

ORC_EXPORT int orc_step(orc_ctx* c, void* stream) {
  (void)stream;
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "step before finalize");
  int overflow = 0;
#pragma omp parallel for schedule(static) reduction(| : overflow) if (c->num_envs >= 16)
  for (int e = 0; e < c->num_envs; ++e) {
    orc_step_env(c, &c->envs[e]);
    overflow |= c->envs[e].overflow;
  }
  /* like the HIP library: contacts past the per-env capacity are dropped in (pair, point) order and the step goes on;
   * the sticky flag is read through msk_get_sizes()[7] */
  if (overflow) c->any_overflow = 1;
  if (c->wrench_pending) { /* forces last one step */
    memset(c->wrench, 0, sizeof(float) * 8 * (size_t)c->num_envs * c->nb);
    c->wrench_pending = 0;
  }
  return MSK_OK;
}

/* msk_step_n: `count` consecutive steps; msk_get_step_parts: the oracle steps its envs in one loop */
ORC_EXPORT int orc_step_n(orc_ctx* c, int count, void* stream) {
  for (int i = 0; i < count; ++i) { const int r = orc_step(c, stream); if (r < 0) return r; }
  return MSK_OK;
}
ORC_EXPORT int orc_get_step_parts(orc_ctx* c) { (void)c; return 1; }
ORC_EXPORT int orc_set_step_parts(orc_ctx* c, int parts) { (void)c; (void)parts; return 1; }

ORC_EXPORT int orc_batch(orc_ctx* const* ctxs, int n, int op, uint32_t mask, void* stream) {
  for (int i = 0; i < n; ++i) {
    int r;
    switch (op) {
      case MSK_BATCH_STEP: r = orc_step(ctxs[i], stream); break;
      case MSK_BATCH_APPLY: r = orc_apply(ctxs[i], mask, stream); break;
      case MSK_BATCH_FETCH: r = orc_fetch(ctxs[i], mask, stream); break;
      case MSK_BATCH_UPDATE_KINEMATICS: r = orc_update_kinematics(ctxs[i], stream); break;
      default: return fail(ctxs[i], MSK_ERR_INVALID, "batch: unknown op");
    }
    if (r < 0) return r;
  }
  return MSK_OK;
}

ORC_EXPORT int orc_query_create_pairs(orc_ctx* c, const int32_t* body_pairs, int npairs) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "query before finalize");
  if (c->nqueries >= 16) return fail(c, MSK_ERR_CAPACITY, "too many queries");
  int q = c->nqueries++;
  c->queries[q].npairs = npairs;
  c->queries[q].pairs = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)npairs);
  memcpy(c->queries[q].pairs, body_pairs, sizeof(int32_t) * 2 * (size_t)npairs);
  c->queries[q].out = (float*)calloc((size_t)c->num_envs * npairs * 3, sizeof(float));
  return q;
}

ORC_EXPORT int orc_query_create_bodies(orc_ctx* c, const int32_t* bodies, int nbodies) {
  int32_t pairs[2 * 64];
  if (nbodies < 0 || nbodies > 64) return fail(c, MSK_ERR_CAPACITY, "too many bodies in one query");
  for (int i = 0; i < nbodies; ++i) { pairs[2 * i] = bodies[i]; pairs[2 * i + 1] = MSK_ANY_BODY; }
  return orc_query_create_pairs(c, pairs, nbodies);
}

ORC_EXPORT void* orc_query_buffer(orc_ctx* c, int q, int64_t shape[2]) {
  if (q < 0 || q >= c->nqueries) return NULL;
  shape[0] = (int64_t)c->num_envs * c->queries[q].npairs; shape[1] = 3;
  return c->queries[q].out;
}

ORC_EXPORT int orc_query_run(orc_ctx* c, int q, void* stream) {
  (void)stream;
  if (q < 0 || q >= c->nqueries) return fail(c, MSK_ERR_INVALID, "bad query");
  int np = c->queries[q].npairs;
  for (int e = 0; e < c->num_envs; ++e) {
    const orc_env* env = &c->envs[e];
    for (int p = 0; p < np; ++p) {
      int x = c->queries[q].pairs[2 * p], y = c->queries[q].pairs[2 * p + 1];
      v3 sum = v3_make(0, 0, 0);
      for (int k = 0; k < env->ncontacts; ++k) {
        const orc_contact* ct = &env->contacts[k];
        float sgn = 0.0f;
        if (ct->ba == x && (ct->bb == y || y == MSK_ANY_BODY)) sgn = 1.0f;
        else if (ct->bb == x && (ct->ba == y || y == MSK_ANY_BODY)) sgn = -1.0f;
        else continue;
        v3 imp = v3_madd(v3_madd(v3_scale(ct->n, ct->lam[0]), ct->t1, ct->lam[1]), ct->t2, ct->lam[2]);
        sum = v3_madd(sum, imp, sgn);
      }
      float* o = c->queries[q].out + ((size_t)e * np + p) * 3;
      o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
    }
  }
  return MSK_OK;
}

ORC_EXPORT int orc_get_sizes(orc_ctx* c, int32_t out[8]) {
  out[0] = c->nb; out[1] = c->na; out[2] = c->max_dof; out[3] = c->nv; out[4] = c->ns; out[5] = c->npairs;
  out[6] = c->num_envs; out[7] = c->any_overflow;
  return MSK_OK;
}

ORC_EXPORT int orc_get_contacts(orc_ctx* c, int env, int32_t* ids, float* vals, int max_points) {
  if (env < 0 || env >= c->num_envs) return fail(c, MSK_ERR_INVALID, "bad env");
  const orc_env* e = &c->envs[env];
  int n = e->ncontacts < max_points ? e->ncontacts : max_points;
  for (int i = 0; i < n; ++i) {
    const orc_contact* ct = &e->contacts[i];
    ids[3 * i] = ct->sa; ids[3 * i + 1] = ct->sb; ids[3 * i + 2] = ct->ba * 256 + (ct->bb & 255);
    float* v = vals + 8 * i;
    v[0] = ct->pos.x; v[1] = ct->pos.y; v[2] = ct->pos.z;
    v[3] = ct->n.x; v[4] = ct->n.y; v[5] = ct->n.z;
    v[6] = ct->sep; v[7] = ct->lam[0];
  }
  return e->ncontacts;
}

/* msk_timing_*: the oracle has no kernels to time; kept so that both libraries export the same ABI */
ORC_EXPORT int orc_timing_enable(orc_ctx* c, int max_steps) { (void)c; (void)max_steps; return MSK_OK; }
ORC_EXPORT int orc_timing_read(orc_ctx* c, int slot, double* total_ms, int32_t* launches) {
  (void)c; (void)slot;
  *total_ms = 0.0; *launches = 0;
  return MSK_OK;
}

/* msk_declare_env_box / msk_declare_env_mass / msk_set_env_boxes / msk_set_env_masses: per-env instance parameters */
ORC_EXPORT int orc_declare_env_box(orc_ctx* c, int shape) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "declare_env_box after finalize");
  if (shape < 0 || shape >= c->ns || c->shapes[shape].type != MSK_SHAPE_BOX) return fail(c, MSK_ERR_INVALID, "declare_env_box: not a box shape");
  if (c->xs_slot[shape] < 0) c->xs_slot[shape] = c->nxs++;
  return MSK_OK;
}

ORC_EXPORT int orc_declare_env_mass(orc_ctx* c, int body) {
  if (c->finalized) return fail(c, MSK_ERR_INVALID, "declare_env_mass after finalize");
  if (body < 0 || body >= c->nb || c->bodies[body].kind != MSK_BODY_DYNAMIC) return fail(c, MSK_ERR_INVALID, "declare_env_mass: not a dynamic actor");
  const orc_body* b = &c->bodies[body];
  if (b->com.x != 0.0f || b->com.y != 0.0f || b->com.z != 0.0f || b->I6[3] != 0.0f || b->I6[4] != 0.0f || b->I6[5] != 0.0f)
    return fail(c, MSK_ERR_INVALID, "declare_env_mass: needs the centre of mass at the origin and a diagonal inertia");
  if (c->xb_slot[body] < 0) c->xb_slot[body] = c->nxb++;
  return MSK_OK;
}

ORC_EXPORT int orc_set_env_boxes(orc_ctx* c, int shape, const float* half_sizes, const float* local_pos) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "set_env_boxes before finalize");
  if (shape < 0 || shape >= c->ns || c->xs_slot[shape] < 0) return fail(c, MSK_ERR_INVALID, "set_env_boxes: shape was not declared");
  for (int e = 0; e < c->num_envs; ++e) {
    float* x = c->xshape + ((size_t)e * c->nxs + c->xs_slot[shape]) * 8;
    if (half_sizes) { x[0] = half_sizes[3 * e]; x[1] = half_sizes[3 * e + 1]; x[2] = half_sizes[3 * e + 2]; }
    if (local_pos) { x[4] = local_pos[3 * e]; x[5] = local_pos[3 * e + 1]; x[6] = local_pos[3 * e + 2]; }
  }
  return MSK_OK;
}

ORC_EXPORT int orc_set_env_masses(orc_ctx* c, int body, const float* mass, const float* inertia) {
  if (!c->finalized) return fail(c, MSK_ERR_INVALID, "set_env_masses before finalize");
  if (body < 0 || body >= c->nb || c->xb_slot[body] < 0) return fail(c, MSK_ERR_INVALID, "set_env_masses: body was not declared");
  for (int e = 0; e < c->num_envs; ++e) {
    float* x = c->xbody + ((size_t)e * c->nxb + c->xb_slot[body]) * 8;
    x[0] = mass[e];
    x[1] = 1.0f / inertia[3 * e]; x[2] = 1.0f / inertia[3 * e + 1]; x[3] = 1.0f / inertia[3 * e + 2];
  }
  return MSK_OK;
}

/* msk_set_solver_classes / msk_get_solver_class_counts: GPU scheduling knobs; the scalar restatement has one path */
ORC_EXPORT int orc_set_solver_classes(orc_ctx* c, const int32_t caps[3]) { (void)c; (void)caps; return MSK_OK; }
ORC_EXPORT int orc_get_solver_class_counts(orc_ctx* c, int32_t out[5]) {
  out[0] = c->num_envs; out[1] = out[2] = out[3] = out[4] = 0;
  return MSK_OK;
}

/* msk_get_env_contact_counts */
ORC_EXPORT int orc_get_env_contact_counts(orc_ctx* c, int32_t* out) {
  for (int e = 0; e < c->num_envs; ++e) out[e] = c->envs[e].ncontacts;
  return MSK_OK;
}
