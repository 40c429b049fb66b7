/*
 * msk_collide_lane.h — the one-thread-per-pair narrowphase functions (namespace perlane): plane-vs-vertices, box-box
 * SAT and the clipping manifold, the way the box-box and plane lists are processed.
 *
 * Those lists are long (box-box: ~1.3 pairs per env per substep): a wavefront holds up to 64 pairs, so the
 * per-pair instruction stream is fetched once per 64 pairs, and the per-thread arrays (polygons, candidates) cost a
 * few scratch accesses each.  The hull-vs-hull list is short and each pair is expensive (support scans over up
 * to 64 vertices, GJK iterations, EPA), so it gets the 16-lanes-per-pair code of msk_collide.h instead.
 * Same arithmetic, operation for operation, as oracle/orc_collide.c and as msk_collide.h.
 */
#ifndef MSK_COLLIDE_LANE_H
#define MSK_COLLIDE_LANE_H

#include "msk_collide.h"

namespace perlane {

struct DContactOut { v3 pos; v3 n; float sep; };

/* what the narrowphase functions need besides the two shapes: the hull vertex pool */
struct CCtx { const v3* verts; };

MSK_DEV v3 shape_vert(const CCtx& m, const CShape* sh, int i) {
  if (sh->type == MSK_SHAPE_BOX)
    return v3_make((i & 1) ? sh->par[0] : -sh->par[0], (i & 2) ? sh->par[1] : -sh->par[1],
                   (i & 4) ? sh->par[2] : -sh->par[2]);
  return m.verts[sh->vbase + i];
}

/* support point (world) of a box / hull in world direction d */
MSK_DEV v3 support(const CCtx& m, const CShape* sh, const pose* T, v3 d) {
  v3 dl = quat_rotate_inv(T->q, d);
  v3 pl;
  if (sh->type == MSK_SHAPE_BOX) {
    pl = v3_make(dl.x >= 0.0f ? sh->par[0] : -sh->par[0], dl.y >= 0.0f ? sh->par[1] : -sh->par[1],
                 dl.z >= 0.0f ? sh->par[2] : -sh->par[2]);
  } else {
    int best = 0;
    float bd = v3_dot(m.verts[sh->vbase], dl);
    for (int i = 1; i < sh->nverts; ++i) {
      float di = v3_dot(m.verts[sh->vbase + i], dl);
      if (di > bd) { bd = di; best = i; }
    }
    pl = m.verts[sh->vbase + best];
  }
  return pose_apply(*T, pl);
}

/* ---- manifold ------------------------------------------------------------------------ */
typedef struct { float u, v, h; } p3;   /* coordinates in the (t1, t2, n) contact frame */

/* support feature of `sh` along sign*n: up to 8 extreme points, CCW about n.
 * Two passes over the vertices (extreme height, then the eight directional maxima kept in registers side by side);
 * every comparison sees the same operands in the same vertex order as the oracle's direction-by-direction scan. */
MSK_DEV int select_feature(const CCtx& m, const CShape* sh, const pose* T, v3 n, v3 t1, v3 t2, float sign, p3* out) {
  const float DX[8] = {1.0f, 0.70710678f, 0.0f, -0.70710678f, -1.0f, -0.70710678f, 0.0f, 0.70710678f};
  const float DY[8] = {0.0f, 0.70710678f, 1.0f, 0.70710678f, 0.0f, -0.70710678f, -1.0f, -0.70710678f};
  v3 nl = quat_rotate_inv(T->q, n), t1l = quat_rotate_inv(T->q, t1), t2l = quat_rotate_inv(T->q, t2);
  float on = v3_dot(T->p, n), o1 = v3_dot(T->p, t1), o2 = v3_dot(T->p, t2);
  const int nv = shape_nverts(sh);
  float hbest = -3.0e38f;
  for (int i = 0; i < nv; ++i) {
    const float s = sign * v3_dot(shape_vert(m, sh, i), nl);
    if (s > hbest) hbest = s;
  }
  const float thr = hbest - ORC_FEAT_EPS;
  int sel[8];
  float bd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sel[k] = -1; bd[k] = -3.0e38f; }
  for (int i = 0; i < nv; ++i) {
    const v3 p = shape_vert(m, sh, i);
    if (sign * v3_dot(p, nl) < thr) continue;
    const float pu = v3_dot(p, t1l), pv = v3_dot(p, t2l);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float d = fmaf(pu, DX[k], pv * DY[k]);
      if (d > bd[k]) { bd[k] = d; sel[k] = i; }
    }
  }
  /* drop repeats of the previous kept vertex, and a last one equal to the first */
  int cnt = 0, first = -1, last = -1;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (cnt > 0 && sel[k] == last) continue;
    last = sel[k];
    if (cnt == 0) first = last;
    const v3 p = shape_vert(m, sh, last);
    out[cnt].u = v3_dot(p, t1l) + o1;
    out[cnt].v = v3_dot(p, t2l) + o2;
    out[cnt].h = v3_dot(p, nl) + on;
    cnt++;
  }
  if (cnt > 1 && last == first) cnt--;
  return cnt;
}

/* height (n coordinate) of a feature's surface above the in-plane point (u, v) */
MSK_DEV float feature_height(const p3* f, int n, float u, float v) {
  if (n == 1) return f[0].h;
  if (n == 2) {
    float du = f[1].u - f[0].u, dv = f[1].v - f[0].v;
    float l2 = fmaf(du, du, dv * dv);
    float t = (l2 > 1e-12f) ? fmaf(u - f[0].u, du, (v - f[0].v) * dv) / l2 : 0.0f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    return fmaf(t, f[1].h - f[0].h, f[0].h);
  }
  /* Newell normal and centroid */
  float mx = 0, my = 0, mz = 0, gu = 0, gv = 0, gh = 0;
  for (int i = 0; i < n; ++i) {
    const p3* a = &f[i];
    const p3* b = &f[(i + 1 == n) ? 0 : i + 1];
    mx += (a->v - b->v) * (a->h + b->h);
    my += (a->h - b->h) * (a->u + b->u);
    mz += (a->u - b->u) * (a->v + b->v);
    gu += a->u; gv += a->v; gh += a->h;
  }
  float inv = 1.0f / (float)n;
  gu *= inv; gv *= inv; gh *= inv;
  if (fabsf(mz) < 1e-12f) return gh;
  return gh - (mx * (u - gu) + my * (v - gv)) / mz;
}


/* clip the segment p0-p1 against the convex CCW polygon poly; returns number of points (0..2) */
MSK_DEV int clip_segment_poly(const p3* seg, const p3* poly, int np, float out[][2]) {
  float t0 = 0.0f, t1 = 1.0f;
  float dx = seg[1].u - seg[0].u, dy = seg[1].v - seg[0].v;
  for (int i = 0; i < np; ++i) {
    const p3* a = &poly[i];
    const p3* b = &poly[(i + 1 == np) ? 0 : i + 1];
    float ex = b->u - a->u, ey = b->v - a->v;
    float c0 = cross2(ex, ey, seg[0].u - a->u, seg[0].v - a->v);
    float cd = cross2(ex, ey, dx, dy);
    if (fabsf(cd) < 1e-12f) {
      if (c0 < -1e-7f) return 0;
      continue;
    }
    float t = -c0 / cd;
    if (cd > 0.0f) { if (t > t0) t0 = t; }
    else { if (t < t1) t1 = t; }
  }
  if (t0 > t1 + 1e-6f) return 0;
  out[0][0] = fmaf(t0, dx, seg[0].u); out[0][1] = fmaf(t0, dy, seg[0].v);
  if (t1 - t0 < 1e-6f) return 1;
  out[1][0] = fmaf(t1, dx, seg[0].u); out[1][1] = fmaf(t1, dy, seg[0].v);
  return 2;
}

/* Sutherland-Hodgman: subject polygon (CCW) clipped by convex CCW polygon */
MSK_DEV int clip_poly_poly(const p3* subj, int ns, const p3* clip, int nc, float out[][2]) {
  float bufa[24][2], bufb[24][2];
  int na = ns;
  for (int i = 0; i < ns; ++i) { bufa[i][0] = subj[i].u; bufa[i][1] = subj[i].v; }
  float(*in)[2] = bufa;
  float(*ot)[2] = bufb;
  for (int ci = 0; ci < nc && na > 0; ++ci) {
    const p3* a = &clip[ci];
    const p3* b = &clip[(ci + 1 == nc) ? 0 : ci + 1];
    float ex = b->u - a->u, ey = b->v - a->v;
    int no = 0;
    for (int i = 0; i < na; ++i) {
      const float* P = in[i];
      const float* Q = in[(i + 1 == na) ? 0 : i + 1];
      float cp = cross2(ex, ey, P[0] - a->u, P[1] - a->v);
      float cq = cross2(ex, ey, Q[0] - a->u, Q[1] - a->v);
      int pin = cp >= -1e-9f, qin = cq >= -1e-9f;
      if (pin && no < ORC_CLIP_MAXV) { ot[no][0] = P[0]; ot[no][1] = P[1]; no++; }
      if (pin != qin && no < ORC_CLIP_MAXV) {
        float t = cp / (cp - cq);
        ot[no][0] = fmaf(t, Q[0] - P[0], P[0]);
        ot[no][1] = fmaf(t, Q[1] - P[1], P[1]);
        no++;
      }
    }
    float(*tmp)[2] = in; in = ot; ot = tmp;
    na = no;
  }
  for (int i = 0; i < na; ++i) { out[i][0] = in[i][0]; out[i][1] = in[i][1]; }
  return na;
}

MSK_DEV int seg_seg(const p3* a, const p3* b, float out[][2]) {
  float d1x = a[1].u - a[0].u, d1y = a[1].v - a[0].v;
  float d2x = b[1].u - b[0].u, d2y = b[1].v - b[0].v;
  float rx = b[0].u - a[0].u, ry = b[0].v - a[0].v;
  float den = cross2(d1x, d1y, d2x, d2y);
  float l1 = fmaf(d1x, d1x, d1y * d1y), l2 = fmaf(d2x, d2x, d2y * d2y);
  if (den * den > 1e-6f * l1 * l2) {
    float s = cross2(rx, ry, d2x, d2y) / den;
    s = fminf(fmaxf(s, 0.0f), 1.0f);
    out[0][0] = fmaf(s, d1x, a[0].u); out[0][1] = fmaf(s, d1y, a[0].v);
    return 1;
  }
  /* parallel: overlap of b's endpoints projected on a */
  if (l1 < 1e-12f) { out[0][0] = a[0].u; out[0][1] = a[0].v; return 1; }
  float s0 = fmaf(rx, d1x, ry * d1y) / l1;
  float s1 = fmaf(b[1].u - a[0].u, d1x, (b[1].v - a[0].v) * d1y) / l1;
  float lo = fmaxf(fminf(s0, s1), 0.0f), hi = fminf(fmaxf(s0, s1), 1.0f);
  if (lo > hi) { float m = fminf(fmaxf(0.5f * (s0 + s1), 0.0f), 1.0f); lo = hi = m; }
  out[0][0] = fmaf(lo, d1x, a[0].u); out[0][1] = fmaf(lo, d1y, a[0].v);
  if (hi - lo < 1e-6f) return 1;
  out[1][0] = fmaf(hi, d1x, a[0].u); out[1][1] = fmaf(hi, d1y, a[0].v);
  return 2;
}

typedef struct { float u, v, hm, sep; } cand;

/* keep at most 4 candidates: deepest, farthest from it, and the extremes on both sides of that line */
MSK_DEV int reduce4(cand* cs, int n) {
  if (n <= 4) return n;
  int i0 = 0;
  for (int i = 1; i < n; ++i) if (cs[i].sep < cs[i0].sep) i0 = i;
  int i1 = -1; float best = -1.0f;
  for (int i = 0; i < n; ++i) {
    if (i == i0) continue;
    float du = cs[i].u - cs[i0].u, dv = cs[i].v - cs[i0].v;
    float d = fmaf(du, du, dv * dv);
    if (d > best) { best = d; i1 = i; }
  }
  float ex = cs[i1].u - cs[i0].u, ey = cs[i1].v - cs[i0].v;
  int i2 = -1, i3 = -1; float bp = 0.0f, bn = 0.0f;
  for (int i = 0; i < n; ++i) {
    if (i == i0 || i == i1) continue;
    float cr = cross2(ex, ey, cs[i].u - cs[i0].u, cs[i].v - cs[i0].v);
    if (cr > bp) { bp = cr; i2 = i; }
    if (cr < bn) { bn = cr; i3 = i; }
  }
  cand out[4];
  int m = 0;
  out[m++] = cs[i0]; out[m++] = cs[i1];
  if (i2 >= 0) out[m++] = cs[i2];
  if (i3 >= 0) out[m++] = cs[i3];
  for (int i = 0; i < m; ++i) cs[i] = out[i];
  return m;
}

MSK_DEV int build_manifold(const CCtx& m, const CShape* A, const pose* TA, const CShape* B, const pose* TB, v3 n,
                          float margin, v3 wa, v3 wb, float sep_hint, DContactOut* out) {
  v3 t1, t2;
  msk_tangents(n, &t1, &t2);
  p3 fa[8], fb[8];
  int ka = select_feature(m, A, TA, n, t1, t2, -1.0f, fa);
  int kb = select_feature(m, B, TB, n, t1, t2, 1.0f, fb);
  float pts[24][2];
  int np = 0;
  if (ka == 1) { pts[0][0] = fa[0].u; pts[0][1] = fa[0].v; np = 1; }
  else if (kb == 1) { pts[0][0] = fb[0].u; pts[0][1] = fb[0].v; np = 1; }
  else if (ka >= 3 && kb >= 3) np = clip_poly_poly(fa, ka, fb, kb, pts);
  else if (ka == 2 && kb >= 3) np = clip_segment_poly(fa, fb, kb, pts);
  else if (kb == 2 && ka >= 3) np = clip_segment_poly(fb, fa, ka, pts);
  else np = seg_seg(fa, fb, pts);
  cand cs[24];
  int nc = 0;
  for (int i = 0; i < np; ++i) {
    float ha = feature_height(fa, ka, pts[i][0], pts[i][1]);
    float hb = feature_height(fb, kb, pts[i][0], pts[i][1]);
    float sep = ha - hb;
    if (sep > margin) continue;
    cs[nc].u = pts[i][0]; cs[nc].v = pts[i][1]; cs[nc].hm = 0.5f * (ha + hb); cs[nc].sep = sep;
    nc++;
  }
  if (nc == 0) {
    if (sep_hint > margin) return 0;
    v3 mid = v3_scale(v3_add(wa, wb), 0.5f);
    out[0].pos = mid; out[0].n = n; out[0].sep = sep_hint;
    return 1;
  }
  nc = reduce4(cs, nc);
  for (int i = 0; i < nc; ++i) {
    out[i].pos = v3_madd(v3_madd(v3_scale(t1, cs[i].u), t2, cs[i].v), n, cs[i].hm);
    out[i].n = n;
    out[i].sep = cs[i].sep;
  }
  return nc;
}

/* ---- box-box SAT ---------------------------------------------------------------------- */
MSK_DEV int sat_box_box(const CShape* A, const pose* TA, const CShape* B, const pose* TB, float margin,
                       v3* n_out, float* sep_out) {
  m33 Ra = quat_to_m33(TA->q), Rb = quat_to_m33(TB->q);
  v3 au[3] = {m33_col(&Ra, 0), m33_col(&Ra, 1), m33_col(&Ra, 2)};
  v3 bu[3] = {m33_col(&Rb, 0), m33_col(&Rb, 1), m33_col(&Rb, 2)};
  const float* a = A->par;
  const float* b = B->par;
  v3 dc = v3_sub(TA->p, TB->p); /* from B to A */
  float R[3][3], AR[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { R[i][j] = v3_dot(au[i], bu[j]); AR[i][j] = fabsf(R[i][j]); }
  float best_f = -3.0e38f; v3 nf = v3_make(0, 0, 1);
  for (int i = 0; i < 3; ++i) {
    float t = v3_dot(dc, au[i]);
    float rb = fmaf(b[0], AR[i][0], fmaf(b[1], AR[i][1], b[2] * AR[i][2]));
    float s = fabsf(t) - (a[i] + rb);
    if (s > best_f) { best_f = s; nf = (t >= 0.0f) ? au[i] : v3_neg(au[i]); }
  }
  for (int j = 0; j < 3; ++j) {
    float t = v3_dot(dc, bu[j]);
    float ra = fmaf(a[0], AR[0][j], fmaf(a[1], AR[1][j], a[2] * AR[2][j]));
    float s = fabsf(t) - (b[j] + ra);
    if (s > best_f) { best_f = s; nf = (t >= 0.0f) ? bu[j] : v3_neg(bu[j]); }
  }
  if (best_f > margin) return 0;
  float best_e = -3.0e38f; v3 ne = nf;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      v3 L = v3_cross(au[i], bu[j]);
      float l2 = v3_len2(L);
      if (l2 < 1e-6f) continue;
      float inv = 1.0f / sqrtf(l2);
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      float ra = fmaf(a[i1], AR[i2][j], a[i2] * AR[i1][j]);
      float rb = fmaf(b[j1], AR[i][j2], b[j2] * AR[i][j1]);
      float t = v3_dot(dc, L);
      float s = (fabsf(t) - (ra + rb)) * inv;
      if (s > best_e) { best_e = s; ne = v3_scale(L, (t >= 0.0f) ? inv : -inv); }
    }
  if (best_e > margin) return 0;
  if (best_e > best_f + 5e-4f) { *n_out = ne; *sep_out = best_e; }
  else { *n_out = nf; *sep_out = best_f; }
  return 1;
}

/* ---- plane ----------------------------------------------------------------------------- */
MSK_DEV int plane_convex(const CCtx& m, const CShape* P, const pose* TP, const CShape* C, const pose* TC, float margin,
                        int plane_is_a, DContactOut* out) {
  v3 pn = quat_rotate(TP->q, v3_make(1, 0, 0));
  float pd = v3_dot(pn, TP->p);
  v3 t1, t2;
  msk_tangents(pn, &t1, &t2);
  cand cs[MSK_MAX_HULL_VERTS];
  int nc = 0;
  int nv = shape_nverts(C);
  for (int i = 0; i < nv; ++i) {
    v3 w = pose_apply(*TC, shape_vert(m, C, i));
    float sep = v3_dot(pn, w) - pd;
    if (sep > margin) continue;
    cs[nc].u = v3_dot(w, t1); cs[nc].v = v3_dot(w, t2); cs[nc].hm = pd + 0.5f * sep; cs[nc].sep = sep;
    nc++;
  }
  nc = reduce4(cs, nc);
  for (int i = 0; i < nc; ++i) {
    out[i].pos = v3_madd(v3_madd(v3_scale(t1, cs[i].u), t2, cs[i].v), pn, cs[i].hm);
    out[i].n = plane_is_a ? v3_neg(pn) : pn;
    out[i].sep = cs[i].sep;
  }
  return nc;
}

}  // namespace perlane

#endif
