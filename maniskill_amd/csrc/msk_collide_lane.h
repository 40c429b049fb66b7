/*
 * msk_collide_lane.h — the one-lane-per-pair clipping manifold (namespace perlane), used for the box-box list.
 *
 * That list is long (~1.3 to 1.6 pairs per env per substep): a wavefront holds up to 32 pairs, so the per-pair
 * instruction stream is issued once for all of them — the 16-lanes-per-pair code of msk_collide.h issues it once
 * per four pairs and is throughput-bound on such a list; it is kept for the short, expensive hull and plane lists.
 * What made one-lane-per-pair slow before was its per-thread arrays (features, polygons, candidates: dynamic
 * indices, hence scratch memory); here they live in LDS, interleaved over the 32 lanes (element k of lane l at
 * word k * 32 + l: conflict-free), 208 words per lane.
 * Same arithmetic, operation for operation, as oracle/orc_collide.c and as msk_collide.h.
 */
#ifndef MSK_COLLIDE_LANE_H
#define MSK_COLLIDE_LANE_H

#include "msk_collide.h"

namespace perlane {

#define PL_LANES 32          /* pairs per wavefront pass (x PL_WORDS words of LDS each) */
/* per-lane LDS arrays (words) */
#define PL_FA 0              /* p3[8]       */
#define PL_FB 24             /* p3[8]       */
#define PL_PTS 48            /* [16][2]     */
#define PL_BUFA 80           /* [16][2]     */
#define PL_BUFB 112          /* [16][2]     */
#define PL_CS 144            /* cand[16]    */
#define PL_WORDS 208

/* word k of this lane's array */
struct LArr {
  float* b;
  __device__ __forceinline__ float& operator()(int k) const { return b[k * PL_LANES]; }
  __device__ __forceinline__ LArr at(int k) const { LArr r; r.b = b + k * PL_LANES; return r; }
};
/* p3 {u, v, h}: words 3i, 3i+1, 3i+2; point {x, y}: 2i, 2i+1; cand {u, v, hm, sep}: 4i .. 4i+3 */

struct DContactOut { v3 pos; v3 n; float sep; };

MSK_DEV v3 box_vert(const CShape* sh, int i) {
  return v3_make((i & 1) ? sh->par[0] : -sh->par[0], (i & 2) ? sh->par[1] : -sh->par[1], (i & 4) ? sh->par[2] : -sh->par[2]);
}

/* support feature of the box along sign*n: up to 8 extreme points, CCW about n.
 * Two passes over the vertices (extreme height, then the eight directional maxima kept in registers side by side);
 * every comparison sees the same operands in the same vertex order as the oracle's direction-by-direction scan. */
MSK_DEV int select_feature(const CShape* sh, const pose* T, v3 n, v3 t1, v3 t2, float sign, float pen, const LArr out) {
  const float DX[8] = {1.0f, 0.70710678f, 0.0f, -0.70710678f, -1.0f, -0.70710678f, 0.0f, 0.70710678f};
  const float DY[8] = {0.0f, 0.70710678f, 1.0f, 0.70710678f, 0.0f, -0.70710678f, -1.0f, -0.70710678f};
  v3 nl = quat_rotate_inv(T->q, n), t1l = quat_rotate_inv(T->q, t1), t2l = quat_rotate_inv(T->q, t2);
  float on = v3_dot(T->p, n), o1 = v3_dot(T->p, t1), o2 = v3_dot(T->p, t2);
  float hbest = -3.0e38f, hworst = 3.0e38f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float s = sign * v3_dot(box_vert(sh, i), nl);
    if (s > hbest) hbest = s;
    if (s < hworst) hworst = s;
  }
  /* the band grows by the penetration depth, up to just short of the box's mid-plane (oracle: select_feature) */
  const float eps = fminf(ORC_FEAT_EPS + pen, fmaxf(ORC_FEAT_EPS, 0.45f * (hbest - hworst)));
  const float thr = hbest - eps;
  int sel[8];
  float bd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sel[k] = -1; bd[k] = -3.0e38f; }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const v3 p = box_vert(sh, i);
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
    const v3 p = box_vert(sh, last);
    out(cnt * 3 + 0) = v3_dot(p, t1l) + o1;
    out(cnt * 3 + 1) = v3_dot(p, t2l) + o2;
    out(cnt * 3 + 2) = v3_dot(p, nl) + on;
    cnt++;
  }
  if (cnt > 1 && last == first) cnt--;
  return cnt;
}

/* A polygonal feature's plane: Newell normal (mx, my, mz) and centroid (gu, gv, gh).  It belongs to the feature, not to the point whose height is asked for: computed
 * ONCE per feature (build_manifold) -- the height of every clipped point evaluated it again before (up to 8 points x 2 features x a loop over the feature's vertices in LDS:
 * a quarter of the manifold's instructions on a face-face pair).  Same operations in the same order as the oracle's feature_height, hence the same bits. */
struct FeatPlane { float mx, my, mz, gu, gv, gh; };
MSK_DEV FeatPlane feature_plane(const LArr f, int n) {
  FeatPlane pl = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (n < 3) return pl;
  float mx = 0, my = 0, mz = 0, gu = 0, gv = 0, gh = 0;
  for (int i = 0; i < n; ++i) {
    const int j = (i + 1 == n) ? 0 : i + 1;
    const float au = f(i * 3), av = f(i * 3 + 1), ah = f(i * 3 + 2);
    const float bu = f(j * 3), bv = f(j * 3 + 1), bh = f(j * 3 + 2);
    mx += (av - bv) * (ah + bh);
    my += (ah - bh) * (au + bu);
    mz += (au - bu) * (av + bv);
    gu += au; gv += av; gh += ah;
  }
  float inv = 1.0f / (float)n;
  gu *= inv; gv *= inv; gh *= inv;
  pl.mx = mx; pl.my = my; pl.mz = mz; pl.gu = gu; pl.gv = gv; pl.gh = gh;
  return pl;
}

/* height (n coordinate) of a feature's surface above the in-plane point (u, v) */
MSK_DEV float feature_height(const LArr f, int n, const FeatPlane& pl, float u, float v) {
  if (n == 1) return f(2);
  if (n == 2) {
    float du = f(3) - f(0), dv = f(4) - f(1);
    float l2 = fmaf(du, du, dv * dv);
    float t = (l2 > 1e-12f) ? fmaf(u - f(0), du, (v - f(1)) * dv) / l2 : 0.0f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    return fmaf(t, f(5) - f(2), f(2));
  }
  if (fabsf(pl.mz) < 1e-12f) return pl.gh;
  return pl.gh - (pl.mx * (u - pl.gu) + pl.my * (v - pl.gv)) / pl.mz;
}

/* clip the segment p0-p1 against the convex CCW polygon poly; returns number of points (0..2) */
MSK_DEV int clip_segment_poly(const LArr seg, const LArr poly, int np, const LArr out) {
  float t0 = 0.0f, t1 = 1.0f;
  const float s0u = seg(0), s0v = seg(1);
  float dx = seg(3) - s0u, dy = seg(4) - s0v;
  for (int i = 0; i < np; ++i) {
    const int j = (i + 1 == np) ? 0 : i + 1;
    const float au = poly(i * 3), av = poly(i * 3 + 1);
    float ex = poly(j * 3) - au, ey = poly(j * 3 + 1) - av;
    float c0 = cross2(ex, ey, s0u - au, s0v - av);
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
  out(0) = fmaf(t0, dx, s0u); out(1) = fmaf(t0, dy, s0v);
  if (t1 - t0 < 1e-6f) return 1;
  out(2) = fmaf(t1, dx, s0u); out(3) = fmaf(t1, dy, s0v);
  return 2;
}

/* Sutherland-Hodgman: subject polygon (CCW) clipped by convex CCW polygon; ping-pong between two LDS buffers */
MSK_DEV int clip_poly_poly(const LArr subj, int ns, const LArr clip, int nc, LArr in, LArr ot, const LArr out) {
  int na = ns;
  for (int i = 0; i < ns; ++i) { in(i * 2) = subj(i * 3); in(i * 2 + 1) = subj(i * 3 + 1); }
  for (int ci = 0; ci < nc && na > 0; ++ci) {
    const int cj = (ci + 1 == nc) ? 0 : ci + 1;
    const float au = clip(ci * 3), av = clip(ci * 3 + 1);
    float ex = clip(cj * 3) - au, ey = clip(cj * 3 + 1) - av;
    int no = 0;
    for (int i = 0; i < na; ++i) {
      const int j = (i + 1 == na) ? 0 : i + 1;
      const float P0 = in(i * 2), P1 = in(i * 2 + 1), Q0 = in(j * 2), Q1 = in(j * 2 + 1);
      float cp = cross2(ex, ey, P0 - au, P1 - av);
      float cq = cross2(ex, ey, Q0 - au, Q1 - av);
      int pin = cp >= -1e-9f, qin = cq >= -1e-9f;
      if (pin && no < ORC_CLIP_MAXV) { ot(no * 2) = P0; ot(no * 2 + 1) = P1; no++; }
      if (pin != qin && no < ORC_CLIP_MAXV) {
        float t = cp / (cp - cq);
        ot(no * 2) = fmaf(t, Q0 - P0, P0);
        ot(no * 2 + 1) = fmaf(t, Q1 - P1, P1);
        no++;
      }
    }
    const LArr tmp = in; in = ot; ot = tmp;
    na = no;
  }
  for (int i = 0; i < na; ++i) { out(i * 2) = in(i * 2); out(i * 2 + 1) = in(i * 2 + 1); }
  return na;
}

MSK_DEV int seg_seg(const LArr a, const LArr b, const LArr out) {
  const float a0u = a(0), a0v = a(1), b0u = b(0), b0v = b(1), b1u = b(3), b1v = b(4);
  float d1x = a(3) - a0u, d1y = a(4) - a0v;
  float d2x = b1u - b0u, d2y = b1v - b0v;
  float rx = b0u - a0u, ry = b0v - a0v;
  float den = cross2(d1x, d1y, d2x, d2y);
  float l1 = fmaf(d1x, d1x, d1y * d1y), l2 = fmaf(d2x, d2x, d2y * d2y);
  if (den * den > 1e-6f * l1 * l2) {
    float s = cross2(rx, ry, d2x, d2y) / den;
    s = fminf(fmaxf(s, 0.0f), 1.0f);
    out(0) = fmaf(s, d1x, a0u); out(1) = fmaf(s, d1y, a0v);
    return 1;
  }
  /* parallel: overlap of b's endpoints projected on a */
  if (l1 < 1e-12f) { out(0) = a0u; out(1) = a0v; return 1; }
  float s0 = fmaf(rx, d1x, ry * d1y) / l1;
  float s1 = fmaf(b1u - a0u, d1x, (b1v - a0v) * d1y) / l1;
  float lo = fmaxf(fminf(s0, s1), 0.0f), hi = fminf(fmaxf(s0, s1), 1.0f);
  if (lo > hi) { float mm = fminf(fmaxf(0.5f * (s0 + s1), 0.0f), 1.0f); lo = hi = mm; }
  out(0) = fmaf(lo, d1x, a0u); out(1) = fmaf(lo, d1y, a0v);
  if (hi - lo < 1e-6f) return 1;
  out(2) = fmaf(hi, d1x, a0u); out(3) = fmaf(hi, d1y, a0v);
  return 2;
}

typedef struct { float u, v, hm, sep; } cand;
MSK_DEV cand cand_get(const LArr cs, int i) { cand c; c.u = cs(i * 4); c.v = cs(i * 4 + 1); c.hm = cs(i * 4 + 2); c.sep = cs(i * 4 + 3); return c; }

/* keep at most 4 candidates: deepest, farthest from it, and the extremes on both sides of that line */
MSK_DEV int reduce4(const LArr cs, int n, cand res[4]) {
  if (n <= 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < n) res[i] = cand_get(cs, i);
    return n;
  }
  int i0 = 0;
  float s0 = cs(3);
  for (int i = 1; i < n; ++i) { const float s = cs(i * 4 + 3); if (s < s0) { s0 = s; i0 = i; } }
  const cand c0 = cand_get(cs, i0);
  int i1 = -1; float best = -1.0f;
  for (int i = 0; i < n; ++i) {
    if (i == i0) continue;
    float du = cs(i * 4) - c0.u, dv = cs(i * 4 + 1) - c0.v;
    float d = fmaf(du, du, dv * dv);
    if (d > best) { best = d; i1 = i; }
  }
  const cand c1 = cand_get(cs, i1);
  float ex = c1.u - c0.u, ey = c1.v - c0.v;
  int i2 = -1, i3 = -1; float bp = 0.0f, bn = 0.0f;
  for (int i = 0; i < n; ++i) {
    if (i == i0 || i == i1) continue;
    float cr = cross2(ex, ey, cs(i * 4) - c0.u, cs(i * 4 + 1) - c0.v);
    if (cr > bp) { bp = cr; i2 = i; }
    if (cr < bn) { bn = cr; i3 = i; }
  }
  int k = 2;
  res[0] = c0; res[1] = c1;
  if (i2 >= 0) { res[2] = cand_get(cs, i2); k = 3; }
  if (i3 >= 0) { const cand c = cand_get(cs, i3); if (k == 2) res[2] = c; else res[3] = c; k++; }
  return k;
}

/* lb = this lane's word 0 of the interleaved LDS arrays */
MSK_DEV int build_manifold(float* lb, const CShape* A, const pose* TA, const CShape* B, const pose* TB, v3 n,
                          float margin, v3 wa, v3 wb, float sep_hint, DContactOut* out) {
  v3 t1, t2;
  msk_tangents(n, &t1, &t2);
  LArr base; base.b = lb;
  const LArr fa = base.at(PL_FA), fb = base.at(PL_FB), pts = base.at(PL_PTS), cs = base.at(PL_CS);
  const float pen = fmaxf(0.0f, -sep_hint);
  int ka = select_feature(A, TA, n, t1, t2, -1.0f, pen, fa);
  int kb = select_feature(B, TB, n, t1, t2, 1.0f, pen, fb);
  int np = 0;
  if (ka == 1) { pts(0) = fa(0); pts(1) = fa(1); np = 1; }
  else if (kb == 1) { pts(0) = fb(0); pts(1) = fb(1); np = 1; }
  else if (ka >= 3 && kb >= 3) np = clip_poly_poly(fa, ka, fb, kb, base.at(PL_BUFA), base.at(PL_BUFB), pts);
  else if (ka == 2 && kb >= 3) np = clip_segment_poly(fa, fb, kb, pts);
  else if (kb == 2 && ka >= 3) np = clip_segment_poly(fb, fa, ka, pts);
  else np = seg_seg(fa, fb, pts);
  int nc = 0;
  const FeatPlane pla = feature_plane(fa, ka), plb = feature_plane(fb, kb);
  for (int i = 0; i < np; ++i) {
    const float pu = pts(i * 2), pv = pts(i * 2 + 1);
    float ha = feature_height(fa, ka, pla, pu, pv);
    float hb = feature_height(fb, kb, plb, pu, pv);
    float sep = ha - hb;
    if (sep > margin) continue;
    cs(nc * 4) = pu; cs(nc * 4 + 1) = pv; cs(nc * 4 + 2) = 0.5f * (ha + hb); cs(nc * 4 + 3) = sep;
    nc++;
  }
  if (nc == 0) {
    if (sep_hint > margin) return 0;
    v3 mid = v3_scale(v3_add(wa, wb), 0.5f);
    out[0].pos = mid; out[0].n = n; out[0].sep = sep_hint;
    return 1;
  }
  cand res[4];
  nc = reduce4(cs, nc, res);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < nc) {
      out[i].pos = v3_madd(v3_madd(v3_scale(t1, res[i].u), t2, res[i].v), n, res[i].hm);
      out[i].n = n;
      out[i].sep = res[i].sep;
    }
  }
  return nc;
}

}  // namespace perlane

#endif
