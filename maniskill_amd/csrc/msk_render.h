/*
 * msk_render.h — batched depth + segmentation rasteriser (include/msk_render.h), gfx950.
 *
 *   k_render_setup   one 256-thread workgroup per env.  Threads first build the camera-from-shape
 *                    transforms (body poses come straight from the simulator's env record), then
 *                    transform the render vertices into the camera frame (LDS), then set up the
 *                    triangles: near-plane clip, projection, back-face cull, edge + 1/depth plane
 *                    equations, pixel bounding box -> 64-byte TriSetup records, binned into
 *                    8x8-pixel tiles (LDS counters, prefix, per-tile record arrays).
 *   k_render_tiles   one wavefront per (4 consecutive 8x8 tiles, env), lane = pixel: a tile's records go through
 *                    LDS (read back as broadcasts) while the next tile's are already in flight; per pixel 3 edge
 *                    evaluations + 1/depth compare; the winner's camera-space position (mm, int16) and
 *                    segmentation id leave as one 8-byte store per pixel.
 *
 * The output (N x H x W x 8 bytes = 512 MiB at 4096 envs, 128x128) is the algorithmic traffic of
 * this path: it is HBM-write-bound by construction.  Arithmetic follows oracle/orc_render.c
 * statement by statement (the images are compared bit for bit).
 */
#ifndef MSK_RENDER_KERNELS_H
#define MSK_RENDER_KERNELS_H

#include "../../include/msk_render.h"
#include "msk_model.h"

#define MSK_TILE 8                 /* one wavefront rasterises an 8 x 8 tile, lane = pixel */
#define MSK_MAX_TILES 4096         /* 8 x 8 pixel tiles per picture: up to 512 x 512 (the human-render cameras, sapien_env.py _default_human_render_camera_configs);
                                    * the per-camera arrays and the setup kernel's LDS are sized by the camera's own tile count (RCamera::tile_cap) */
#define MSK_BIG_TILES 16           /* a triangle over more tiles than this is binned by the whole workgroup */
#define MSK_MAX_BIG 16
#define MSK_SEG_BIG 0x40000000      /* flag in TriSetup::seg: the record is in the env's list of large triangles */
#define MSK_TILES_PER_WAVE 4        /* consecutive tiles a wavefront walks, prefetching the next one's records */
#define MSK_SETUP_WORDS 16
#define MSK_RSHAPE_WORDS 12          /* LDS image of a render shape: camera-from-shape pose (7), pad, per-env scale (3), pad */

struct RShape { int body, seg; pose local; float color[4]; int xs; /* per-env box instance it follows (slot in the env record), -1: none */ };
#define MSK_MAX_LIGHTS 4
struct RTri { int v0, v1, v2, shape; };
struct RModel {
  int nv, nt, ns;
  RShape shapes[MSK_MAX_RENDER_SHAPES];
  v3 verts[MSK_MAX_RENDER_VERTS];
  unsigned char vshape[MSK_MAX_RENDER_VERTS];
  RTri tris[MSK_MAX_RENDER_TRIS];
  /* Color: flat Lambert shading by the scene's ambient + directional lights (directions in the env frame, normalised) */
  float ambient[3];
  int nlights;
  float ldir[MSK_MAX_LIGHTS][3], lcol[MSK_MAX_LIGHTS][3];
};
struct RCamera {
  int W, H, mount, tiles_x, tiles_y;
  int tile_cap;                    /* tiles_x * tiles_y: stride of the per-tile arrays below and of the setup kernel's LDS counters */
  float fx, fy, cx, cy, near_, far_;
  pose local;
  int setup_cap, list_cap;         /* per env */
  float* setups;                   /* [N][setup_cap][16]  */
  int* nsetup;                     /* [N]                 */
  int* tile_off;                   /* [N][ntiles + 1]     */
  float* tile_recs;                /* [N][list_cap][16]: per tile, the records of the small triangles that touch it */
  float* big_recs;                 /* [N][MSK_MAX_BIG][16]: triangles over many tiles (table, ground): tested by every tile */
  int* nbig;                       /* [N] */
  unsigned short* tile_bigmask;    /* [N][tile_cap]: bit b = large triangle b can cover a pixel centre of the tile */
  short* out;                      /* [N][H][W][4]        */
  unsigned* color;                 /* [N][H][W]: Color r8g8b8a8unorm (r in the low byte), 0 = background; null until asked for */
  short* depth;                    /* [N][H][W]: -z of out (Camera.get_obs's depth), written by the same store */
  short* seg;                      /* [N][H][W]: w of out                                                      */
  int* overflow;                   /* [1]                 */
};

/* One screen triangle: A,B,C of the three edge functions (inside = all >= 0), the 1/depth plane,
 * segmentation id, primitive id (tie break), pixel bounding box. */
struct TriSetup {   /* (word order = what the tile kernel loads: the first two edges' coefficients side by side for packed FMAs) */
  float A0, A1, B0, B1, C0, C1, A2, B2, C2, Aw, Bw, Cw;
  int seg, prim, bb;               /* bb = x0 | x1 << 8 | y0 << 16 | y1 << 24 (images are at most 256 x 256) */
  unsigned color;                  /* shaded r8g8b8a8 of the (flat) triangle */
};
#define BB_X0(bb) ((bb) & 0xFF)
#define BB_X1(bb) (((bb) >> 8) & 0xFF)
#define BB_Y0(bb) (((bb) >> 16) & 0xFF)
#define BB_Y1(bb) (((unsigned)(bb)) >> 24)

/* flat shading of one triangle (camera-frame corners p0 p1 p2, counter-clockwise seen from outside): per channel
 * base * min(1, ambient + sum_l light_l * max(0, n . -dir_l)), rounded to 8 bits; alpha = 255 */
MSK_DEV unsigned shade_triangle(v3 p0, v3 p1, v3 p2, const float* base, const float* ambient, int nl, const float* ldir_cam, const float* lcol) {
  v3 n = v3_cross(v3_sub(p1, p0), v3_sub(p2, p0));
  const float l = v3_len(n);
  n = (l > 0.0f) ? v3_scale(n, 1.0f / l) : v3_make(0, 0, 0);
  float lit[3] = {ambient[0], ambient[1], ambient[2]};
  for (int k = 0; k < nl; ++k) {
    const float d = fmaxf(0.0f, -(n.x * ldir_cam[k * 3] + n.y * ldir_cam[k * 3 + 1] + n.z * ldir_cam[k * 3 + 2]));
    lit[0] = fmaf(lcol[k * 3], d, lit[0]); lit[1] = fmaf(lcol[k * 3 + 1], d, lit[1]); lit[2] = fmaf(lcol[k * 3 + 2], d, lit[2]);
  }
  unsigned out = 0xFF000000u;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float v = fminf(fmaxf(base[ch] * fminf(lit[ch], 1.0f), 0.0f), 1.0f);
    out |= ((unsigned)rintf(v * 255.0f) & 0xFFu) << (8 * ch);
  }
  return out;
}

/* Can the triangle cover a pixel centre of tile (tx, ty)?  An edge function fma(A, x, fma(B, y, C)) is monotone in x and
 * in y (rounding is monotone), so over the tile's pixel centres it peaks at one of the four corner centres: if that
 * peak is negative for some edge, no centre of the tile passes the inside test.  Exact, not just conservative in R. */
MSK_DEV bool tile_touches(float A0, float B0, float C0, float A1, float B1, float C1, float A2, float B2, float C2, int tx, int ty) {
  const float x0 = (float)(tx * MSK_TILE) + 0.5f, x1 = (float)(tx * MSK_TILE + MSK_TILE - 1) + 0.5f;
  const float y0 = (float)(ty * MSK_TILE) + 0.5f, y1 = (float)(ty * MSK_TILE + MSK_TILE - 1) + 0.5f;
  const float m0 = fmaxf(fmaxf(fmaf(A0, x0, fmaf(B0, y0, C0)), fmaf(A0, x1, fmaf(B0, y0, C0))), fmaxf(fmaf(A0, x0, fmaf(B0, y1, C0)), fmaf(A0, x1, fmaf(B0, y1, C0))));
  const float m1 = fmaxf(fmaxf(fmaf(A1, x0, fmaf(B1, y0, C1)), fmaf(A1, x1, fmaf(B1, y0, C1))), fmaxf(fmaf(A1, x0, fmaf(B1, y1, C1)), fmaf(A1, x1, fmaf(B1, y1, C1))));
  const float m2 = fmaxf(fmaxf(fmaf(A2, x0, fmaf(B2, y0, C2)), fmaf(A2, x1, fmaf(B2, y0, C2))), fmaxf(fmaf(A2, x0, fmaf(B2, y1, C2)), fmaf(A2, x1, fmaf(B2, y1, C2))));
  return m0 >= 0.0f && m1 >= 0.0f && m2 >= 0.0f;
}

/* projects a camera-frame point (x forward, y left, z up): pixel coordinates and 1/depth */
MSK_DEV void project_point(const RCamera& cam, v3 p, float* u, float* v, float* w) {
  const float iw = 1.0f / p.x;
  *u = fmaf(cam.fx, -p.y * iw, cam.cx);
  *v = fmaf(-cam.fy, p.z * iw, cam.cy);
  *w = iw;
}

/* sets up the screen triangle (p0, p1, p2), all in front of the near plane; returns 0 if it is culled */
MSK_DEV int setup_triangle(const RCamera& cam, v3 p0, v3 p1, v3 p2, int seg, int prim, TriSetup* t) {
  float u0, v0, w0, u1, v1, w1, u2, v2, w2;
  project_point(cam, p0, &u0, &v0, &w0);
  project_point(cam, p1, &u1, &v1, &w1);
  project_point(cam, p2, &u2, &v2, &w2);
  /* image rows grow downwards: a triangle that is counter-clockwise seen from outside has negative area here */
  const float area = fmaf(u1 - u0, v2 - v0, -((v1 - v0) * (u2 - u0)));
  if (!(area < -1e-12f)) return 0;
  /* swap 1 <-> 2: positive orientation, interior = all edge functions >= 0 */
  float t_; t_ = u1; u1 = u2; u2 = t_; t_ = v1; v1 = v2; v2 = t_; t_ = w1; w1 = w2; w2 = t_;
  const float a2 = -area;
  const float umin = fminf(u0, fminf(u1, u2)), umax = fmaxf(u0, fmaxf(u1, u2));
  const float vmin = fminf(v0, fminf(v1, v2)), vmax = fmaxf(v0, fmaxf(v1, v2));
  /* pixel centres at +0.5 */
  int x0 = (int)ceilf(umin - 0.5f), x1 = (int)floorf(umax - 0.5f);
  int y0 = (int)ceilf(vmin - 0.5f), y1 = (int)floorf(vmax - 0.5f);
  if (!(umin < 1e9f && umax > -1e9f && vmin < 1e9f && vmax > -1e9f)) return 0;
  x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, cam.W - 1); y1 = min(y1, cam.H - 1);
  if (x0 > x1 || y0 > y1) return 0;
  /* edge a->b: E(x, y) = (b.u - a.u)(y - a.v) - (b.v - a.v)(x - a.u) = A x + B y + C */
  t->A0 = -(v1 - v0); t->B0 = u1 - u0; t->C0 = -fmaf(t->A0, u0, t->B0 * v0);
  t->A1 = -(v2 - v1); t->B1 = u2 - u1; t->C1 = -fmaf(t->A1, u1, t->B1 * v1);
  t->A2 = -(v0 - v2); t->B2 = u0 - u2; t->C2 = -fmaf(t->A2, u2, t->B2 * v2);
  /* 1/depth is affine on the screen: w = (E12 w0 + E20 w1 + E01 w2) / area */
  const float ia = 1.0f / a2;
  t->Aw = fmaf(t->A1, w0, fmaf(t->A2, w1, t->A0 * w2)) * ia;
  t->Bw = fmaf(t->B1, w0, fmaf(t->B2, w1, t->B0 * w2)) * ia;
  t->Cw = fmaf(t->C1, w0, fmaf(t->C2, w1, t->C0 * w2)) * ia;
  t->seg = seg; t->prim = prim;
  t->bb = x0 | (x1 << 8) | (y0 << 16) | (y1 << 24);
  t->color = 0u;
  return 1;
}

MSK_DEV v3 lerp_near(v3 a, v3 b, float near_) { /* point of segment a-b on the plane x = near */
  const float s = (near_ - a.x) / (b.x - a.x);
  return v3_make(near_, fmaf(s, b.y - a.y, a.y), fmaf(s, b.z - a.z, a.z));
}

__global__ void __launch_bounds__(256) k_render_setup(const DModel* __restrict__ m, DState st, const RModel* __restrict__ rm, RCamera cam) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int e = blockIdx.x, tid = threadIdx.x;
  const int ntiles = cam.tiles_x * cam.tiles_y;
  /* LDS: shape transforms and scales [ns][12] | tile counters [ntiles] | tile fill [ntiles] | nsetup | camera-frame vertices [nv][3] */
  float* Lshape = lds;
  int* Lcnt = (int*)(lds + MSK_MAX_RENDER_SHAPES * MSK_RSHAPE_WORDS);
  int* Lfill = Lcnt + cam.tile_cap + 4;
  int* Lns = Lfill + cam.tile_cap;
  int* Lnbig = Lns + 1;
  int* Lbig = Lns + 4;
  float* Llight = (float*)(Lbig + MSK_MAX_BIG);          /* light directions in the camera frame [MSK_MAX_LIGHTS][3] */
  float* Lv = Llight + MSK_MAX_LIGHTS * 3;
  const float* E = EREC(st, m, e);
  for (int i = tid; i < ntiles; i += 256) { Lcnt[i] = 0; Lfill[i] = 0; }
  if (tid == 0) { *Lns = 0; *Lnbig = 0; }
  /* camera-from-shape transforms */
  pose Tc = cam.local;
  if (cam.mount >= 0) Tc = pose_mul(load_pose(E, m->lay.bpose, cam.mount), cam.local);
  const pose Tci = pose_inv(Tc);
  if (tid < rm->nlights) {
    const v3 dcam = quat_rotate(Tci.q, v3_make(rm->ldir[tid][0], rm->ldir[tid][1], rm->ldir[tid][2]));
    Llight[tid * 3] = dcam.x; Llight[tid * 3 + 1] = dcam.y; Llight[tid * 3 + 2] = dcam.z;
  }
  for (int s = tid; s < rm->ns; s += 256) {
    const RShape* sh = &rm->shapes[s];
    pose L = sh->local;
    v3 scale = v3_make(1.0f, 1.0f, 1.0f);
    if (sh->xs >= 0) { /* follows a per-env box instance: unit-box vertices times the env's half sizes, the env's local position */
      const float* x = E + m->lay.xshape + sh->xs * 8;
      scale = v3_make(x[0], x[1], x[2]);
      L.p = v3_make(x[4], x[5], x[6]);
    }
    pose T = L;
    if (sh->body >= 0) T = pose_mul(load_pose(E, m->lay.bpose, sh->body), L);
    T = pose_mul(Tci, T);
    float* o = Lshape + s * MSK_RSHAPE_WORDS;
    o[0] = T.p.x; o[1] = T.p.y; o[2] = T.p.z; o[3] = T.q.w; o[4] = T.q.x; o[5] = T.q.y; o[6] = T.q.z;
    o[8] = scale.x; o[9] = scale.y; o[10] = scale.z;
  }
  __syncthreads();
  for (int vi = tid; vi < rm->nv; vi += 256) {
    const float* o = Lshape + rm->vshape[vi] * MSK_RSHAPE_WORDS;
    pose T;
    T.p = v3_make(o[0], o[1], o[2]);
    T.q = quat_make(o[3], o[4], o[5], o[6]);
    const v3 vl = rm->verts[vi];
    const v3 p = pose_apply(T, v3_make(vl.x * o[8], vl.y * o[9], vl.z * o[10]));
    Lv[vi * 3 + 0] = p.x; Lv[vi * 3 + 1] = p.y; Lv[vi * 3 + 2] = p.z;
  }
  __syncthreads();
  TriSetup* setups = (TriSetup*)(cam.setups + (size_t)e * cam.setup_cap * MSK_SETUP_WORDS);
  for (int ti = tid; ti < rm->nt; ti += 256) {
    const RTri tr = rm->tris[ti];
    const v3 p[3] = {v3_make(Lv[tr.v0 * 3], Lv[tr.v0 * 3 + 1], Lv[tr.v0 * 3 + 2]), v3_make(Lv[tr.v1 * 3], Lv[tr.v1 * 3 + 1], Lv[tr.v1 * 3 + 2]),
                     v3_make(Lv[tr.v2 * 3], Lv[tr.v2 * 3 + 1], Lv[tr.v2 * 3 + 2])};
    const int seg = rm->shapes[tr.shape].seg;
    const unsigned col = cam.color ? shade_triangle(p[0], p[1], p[2], rm->shapes[tr.shape].color, rm->ambient, rm->nlights, Llight, &rm->lcol[0][0]) : 0u;
    /* clip against the near plane x >= near: a triangle becomes 0, 1 or 2 triangles */
    const bool in0 = p[0].x >= cam.near_, in1 = p[1].x >= cam.near_, in2 = p[2].x >= cam.near_;
    const int nin = (int)in0 + (int)in1 + (int)in2;
    v3 q[4];
    int nq = 0;
    if (nin == 3) { q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; nq = 3; }
    else if (nin > 0) {
      for (int k = 0; k < 3; ++k) { /* Sutherland-Hodgman against one plane, keeps the winding */
        const v3 a = p[k], b = p[(k + 1) % 3];
        const bool ia = a.x >= cam.near_, ib = b.x >= cam.near_;
        if (ia) q[nq++] = a;
        if (ia != ib) q[nq++] = ia ? lerp_near(a, b, cam.near_) : lerp_near(b, a, cam.near_);
      }
    }
    for (int sub = 0; sub + 2 < nq; ++sub) {
      TriSetup t;
      if (!setup_triangle(cam, q[0], q[sub + 1], q[sub + 2], seg, ti * 2 + sub, &t)) continue;
      t.color = col;
      const int slot = atomicAdd(Lns, 1);
      if (slot >= cam.setup_cap) { atomicOr(cam.overflow, 1); continue; }
      const int tx0 = BB_X0(t.bb) / MSK_TILE, tx1 = BB_X1(t.bb) / MSK_TILE, ty0 = BB_Y0(t.bb) / MSK_TILE, ty1 = BB_Y1(t.bb) / MSK_TILE;
      if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > MSK_BIG_TILES) { /* table / ground sized: goes to the env's list of large triangles */
        const int b = atomicAdd(Lnbig, 1);
        if (b < MSK_MAX_BIG) {
          Lbig[b] = slot;
          t.seg |= MSK_SEG_BIG;
          setups[slot] = t;
          continue;
        } /* list full: binned like a small one */
      }
      setups[slot] = t;
      for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx)
          if (tile_touches(t.A0, t.B0, t.C0, t.A1, t.B1, t.C1, t.A2, t.B2, t.C2, tx, ty)) atomicAdd(&Lcnt[ty * cam.tiles_x + tx], 1);
    }
  }
  __threadfence_block();   /* the records are read back by other threads of the workgroup */
  __syncthreads();
  const int nbig = min(*Lnbig, MSK_MAX_BIG);
  __syncthreads();
  const int ns = min(*Lns, cam.setup_cap);
  int* toff = cam.tile_off + (size_t)e * (cam.tile_cap + 1);
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < ntiles; ++i) {
      const int c = Lcnt[i];
      if (acc + c > cam.list_cap) atomicOr(cam.overflow, 1);
      Lcnt[i] = acc;              /* Lcnt now holds the start of the tile's list */
      toff[i] = acc;
      acc = min(acc + c, cam.list_cap);
    }
    toff[ntiles] = acc;
    Lcnt[ntiles] = acc;
    cam.nsetup[e] = ns;
  }
  __syncthreads();
  /* every tile gets its own contiguous copy of the records that touch it: the tile kernel then needs one
   * dependent load (offsets -> records) instead of two (offsets -> indices -> records) */
  float4* recs = (float4*)(cam.tile_recs + (size_t)e * cam.list_cap * MSK_SETUP_WORDS);
  for (int s = tid; s < ns; s += 256) {
    const float4* src = (const float4*)&setups[s];
    const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
    const int bb = __float_as_int(r3.z);
    const int tx0 = BB_X0(bb) / MSK_TILE, tx1 = BB_X1(bb) / MSK_TILE, ty0 = BB_Y0(bb) / MSK_TILE, ty1 = BB_Y1(bb) / MSK_TILE;
    if (__float_as_int(r3.x) & MSK_SEG_BIG) continue;   /* lives in the list of large triangles */
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx) {
        if (!tile_touches(r0.x, r0.z, r1.x, r0.y, r0.w, r1.y, r1.z, r1.w, r2.x, tx, ty)) continue;   /* same test as the count pass (record words: A0 A1 B0 B1 | C0 C1 A2 B2 | C2 ...) */
        const int tile = ty * cam.tiles_x + tx;
        const int pos = Lcnt[tile] + atomicAdd(&Lfill[tile], 1);
        if (pos < Lcnt[tile + 1]) {
          float4* dst = recs + (size_t)pos * 4;
          dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
        }
      }
  }
  /* large triangles: one copy per env, every tile tests them against its own pixel box */
  float4* bigr = (float4*)(cam.big_recs + (size_t)e * MSK_MAX_BIG * MSK_SETUP_WORDS);
  for (int i = tid; i < nbig * 4; i += 256) bigr[i] = ((const float4*)&setups[Lbig[i / 4]])[i % 4];
  if (tid == 0) cam.nbig[e] = nbig;
  /* per tile, the large triangles that can cover one of its pixel centres (bit b = entry b of the list) */
  unsigned short* bmask = cam.tile_bigmask + (size_t)e * cam.tile_cap;
  for (int tile = tid; tile < ntiles; tile += 256) {
    const int tx = tile % cam.tiles_x, ty = tile / cam.tiles_x;
    unsigned mk = 0u;
    for (int b = 0; b < nbig; ++b) {
      const TriSetup* t = &setups[Lbig[b]];
      const int qx0 = tx * MSK_TILE, qy0 = ty * MSK_TILE;
      if (BB_X0(t->bb) > qx0 + MSK_TILE - 1 || BB_X1(t->bb) < qx0 || BB_Y0(t->bb) > qy0 + MSK_TILE - 1 || (int)BB_Y1(t->bb) < qy0) continue;
      if (tile_touches(t->A0, t->B0, t->C0, t->A1, t->B1, t->C1, t->A2, t->B2, t->C2, tx, ty)) mk |= 1u << b;
    }
    bmask[tile] = (unsigned short)mk;
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
/* Inside test + depth test of one record against my pixel: the three edge functions fma(A, x, fma(B, y, C)) as two packed FMAs (edges
 * 0 and 1) and two scalar ones, one min3 and one compare (7 VALU instructions per record instead of 11).  Measured: no faster, because
 * the loop is not bound by VALU issue but by the LDS return path -- every lane needs the same record, and a broadcast ds_read_b128
 * still hands 64 x 16 B to the VGPRs: 36 B of coefficients per record = 18 clocks of the CU's one LDS against 7 clocks of VALU per
 * record on each of its four SIMDs.  Reading the records through the scalar cache instead (constant address space, s_load into SGPRs,
 * four records in flight) was tried and is slower still (1.5 ms per picture instead of 0.85): DESIGN.md section 8. */
#define MSK_RASTER_RECORD(t4)                                                                                                   \
  do {                                                                                                                          \
    const float4 ta = (t4)[0], tb = (t4)[1], tc = (t4)[2];                                                                      \
    const f32x2 e01 = __builtin_elementwise_fma((f32x2){ta.x, ta.y}, X2, __builtin_elementwise_fma((f32x2){ta.z, ta.w}, Y2, (f32x2){tb.x, tb.y})); \
    const float e2 = fmaf(tb.z, x, fmaf(tb.w, y, tc.x));                                                                       \
    if (fminf(fminf(e01.x, e01.y), e2) >= 0.0f) {                                                                              \
      const float w = fmaf(tc.y, x, fmaf(tc.z, y, tc.w));                                                                      \
      const float4 td = (t4)[3];                                                                                               \
      const int prim = __float_as_int(td.y);                                                                                   \
      if (w >= wmin && (w > best_w || (w == best_w && prim < best_prim))) {                                                    \
        best_w = w; best_prim = prim; best_seg = __float_as_int(td.x); best_col = (unsigned)__float_as_int(td.w);              \
      }                                                                                                                         \
    }                                                                                                                           \
  } while (0)

/* One wavefront per (group of MSK_TILES_PER_WAVE consecutive 8 x 8 tiles, env), lane = pixel.  The launch is bound by
 * the dependent loads of a tile (offsets -> records) and by the LDS return path of the record broadcasts (see MSK_RASTER_RECORD), not
 * by arithmetic or by the 8 bytes per pixel it writes; a wave keeps the NEXT tile's records in flight (registers) while it rasterises
 * the current one out of LDS. */
__global__ void __launch_bounds__(64) k_render_tiles(RCamera cam) {
  __shared__ __attribute__((aligned(16))) float Ls[64 * MSK_SETUP_WORDS];
  __shared__ __attribute__((aligned(16))) float Lb[MSK_MAX_BIG * MSK_SETUP_WORDS];
  const int e = blockIdx.y, lane = threadIdx.x;
  const int ntiles = cam.tiles_x * cam.tiles_y;
  const int nbig = cam.nbig[e];
  if (lane < nbig) {
    const float4* src = (const float4*)(cam.big_recs + ((size_t)e * MSK_MAX_BIG + lane) * MSK_SETUP_WORDS);
    float4* dst = (float4*)Lb + lane * 4;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
  }
  const int t0 = blockIdx.x * MSK_TILES_PER_WAVE;
  const unsigned short* bigmask = cam.tile_bigmask + (size_t)e * cam.tile_cap;
  const int* toff = cam.tile_off + (size_t)e * (cam.tile_cap + 1);
  const float4* recs = (const float4*)(cam.tile_recs + (size_t)e * cam.list_cap * MSK_SETUP_WORDS);
  const float wmin = 1.0f / cam.far_;
  /* list bounds of my tiles: lane i holds toff[t0 + i] */
  const int myoff = (lane <= MSK_TILES_PER_WAVE && t0 + lane <= ntiles) ? toff[t0 + lane] : 0;
  int lo = __builtin_amdgcn_readlane(myoff, 0);
  float4 pf0, pf1, pf2, pf3;   /* prefetched record `lane` of the upcoming tile's first chunk */
  {
    const int hi = __builtin_amdgcn_readlane(myoff, 1);
    const int idx = min(lo + lane, max(hi - 1, lo));
    const float4* src = recs + (size_t)idx * 4;
    pf0 = src[0]; pf1 = src[1]; pf2 = src[2]; pf3 = src[3];
  }
#pragma unroll
  for (int i = 0; i < MSK_TILES_PER_WAVE; ++i) {
    const int tile = t0 + i;
    if (tile >= ntiles) break;
    const int l0 = lo, l1 = __builtin_amdgcn_readlane(myoff, i + 1);
    lo = l1;
    const int tx = tile % cam.tiles_x, ty = tile / cam.tiles_x;
    const int px = tx * MSK_TILE + (lane % MSK_TILE), py = ty * MSK_TILE + (lane / MSK_TILE);
    const float x = (float)px + 0.5f, y = (float)py + 0.5f;
    const f32x2 X2 = {x, x}, Y2 = {y, y};
    float best_w = 0.0f;
    int best_seg = 0, best_prim = 0x7FFFFFFF;
    unsigned best_col = 0u;
    /* the env's large triangles that reach this tile (LDS, staged once per wave) */
    for (unsigned mk = __builtin_amdgcn_readfirstlane((int)bigmask[tile]); mk != 0u; mk &= mk - 1u) {
      const int k = __builtin_ctz(mk);
      const float4* t4 = (const float4*)(Lb + k * MSK_SETUP_WORDS);
      MSK_RASTER_RECORD(t4);
    }
    for (int c0 = l0; c0 < l1; c0 += 64) {
      const int n = min(64, l1 - c0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (c0 == l0) { /* first chunk: already in registers */
        float4* dst = (float4*)Ls + lane * 4;
        dst[0] = pf0; dst[1] = pf1; dst[2] = pf2; dst[3] = pf3;
      } else if (lane < n) {
        const float4* src = recs + (size_t)(c0 + lane) * 4;
        float4* dst = (float4*)Ls + lane * 4;
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (c0 == l0 && i + 1 < MSK_TILES_PER_WAVE && tile + 1 < ntiles) { /* next tile's first chunk into flight */
        const int nhi = __builtin_amdgcn_readlane(myoff, (i + 2 <= MSK_TILES_PER_WAVE) ? i + 2 : MSK_TILES_PER_WAVE);
        const int idx = min(l1 + lane, max(nhi - 1, l1));
        const float4* src = recs + (size_t)idx * 4;
        pf0 = src[0]; pf1 = src[1]; pf2 = src[2]; pf3 = src[3];
      }
      for (int k = 0; k < n; ++k) {
        /* record: A0 A1 B0 B1 | C0 C1 A2 B2 | C2 Aw Bw Cw | seg prim bb color (same address in every lane: LDS broadcast) */
        const float4* t4 = (const float4*)(Ls + k * MSK_SETUP_WORDS);
        MSK_RASTER_RECORD(t4);
      }
    }
    if (l0 == l1 && i + 1 < MSK_TILES_PER_WAVE && tile + 1 < ntiles) { /* empty tile: still start the next prefetch */
      const int nhi = __builtin_amdgcn_readlane(myoff, (i + 2 <= MSK_TILES_PER_WAVE) ? i + 2 : MSK_TILES_PER_WAVE);
      const int idx = min(l1 + lane, max(nhi - 1, l1));
      const float4* src = recs + (size_t)idx * 4;
      pf0 = src[0]; pf1 = src[1]; pf2 = src[2]; pf3 = src[3];
    }
    /* camera-space OpenGL position in millimetres (x right, y up, z backwards), int16 saturated */
    short4 o = make_short4(0, 0, 0, 0);
    if (best_w > 0.0f) {
      const float d = 1.0f / best_w;
      const float gx = (x - cam.cx) / cam.fx * d, gy = -(y - cam.cy) / cam.fy * d, gz = -d;
      o.x = (short)fminf(fmaxf(rintf(gx * 1000.0f), -32768.0f), 32767.0f);
      o.y = (short)fminf(fmaxf(rintf(gy * 1000.0f), -32768.0f), 32767.0f);
      o.z = (short)fminf(fmaxf(rintf(gz * 1000.0f), -32768.0f), 32767.0f);
      o.w = (short)(best_seg & 0xFFFF);
    }
    const size_t pix = ((size_t)e * cam.H + py) * cam.W + px;
    ((short4*)cam.out)[pix] = o;
    if (cam.color) cam.color[pix] = best_col;
    cam.depth[pix] = (short)(-(int)o.z);   /* int16 negation wraps like the host-side `-position[..., 2]` */
    cam.seg[pix] = o.w;
  }
}

#endif
