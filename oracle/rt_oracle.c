/*
 * rt_oracle.c — CPU restatement of the reference tracer. TEST INFRASTRUCTURE ONLY
 * (see rt_oracle.h).  Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 *
 * Float discipline: every expression below is written with the same operand
 * order and association as the reference source it cites, in float32, with no
 * fused multiply-add (x86-64 SSE, -ffp-contract=off), so results are bit-equal
 * to the reference's own object code.
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- vec3 (maths.h:89-173) */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vld(const float *p) { return V(p[0], p[1], p[2]); }
static inline v3 vmuls(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }            /* maths.h:102 */
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }           /* maths.h:110 */
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }           /* maths.h:117 */
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }           /* maths.h:132 */
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         /* maths.h:140 */
static inline v3 vnormalize(v3 a)                                                          /* maths.h:158-162 */
{
    float len = sqrtf(vdot(a, a));
    return V(a.x / len, a.y / len, a.z / len);
}
static inline v3 vcross(v3 a, v3 b)                                                        /* maths.h:165-172 */
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* ---------------------------------------------------------------- constructors */
void rto_default_material(rto_material *m)                                                 /* main.cpp:34 */
{
    m->reflect = 0.0f; m->diffuse = 1.0f; m->spec = 1.0f;
    m->color[0] = m->color[1] = m->color[2] = 1.0f;
}

void rto_make_sphere(rto_prim *p, const float pos[3], float radius)                        /* main.cpp:53 */
{
    memset(p, 0, sizeof *p);
    p->kind = RTO_SPHERE;
    rto_default_material(&p->mat);
    p->a[0] = pos[0]; p->a[1] = pos[1]; p->a[2] = pos[2];
    p->b[0] = radius; p->b[1] = radius * radius;
}

void rto_make_plane(rto_prim *p, const float normal[3], float offset)                      /* main.cpp:81 */
{
    memset(p, 0, sizeof *p);
    p->kind = RTO_PLANE;
    rto_default_material(&p->mat);
    p->a[0] = normal[0]; p->a[1] = normal[1]; p->a[2] = normal[2];
    p->b[0] = offset;
}

void rto_make_triangle(rto_prim *p, const float v1[3], const float v2[3], const float v3_[3]) /* main.cpp:104-107 */
{
    memset(p, 0, sizeof *p);
    p->kind = RTO_TRIANGLE;
    rto_default_material(&p->mat);
    v3 a = vld(v1), e1 = vsub(vld(v2), a), e2 = vsub(vld(v3_), a);
    v3 n = vnormalize(vcross(e1, e2));
    p->a[0] = a.x;  p->a[1] = a.y;  p->a[2] = a.z;
    p->b[0] = e1.x; p->b[1] = e1.y; p->b[2] = e1.z;
    p->c[0] = e2.x; p->c[1] = e2.y; p->c[2] = e2.z;
    p->n[0] = n.x;  p->n[1] = n.y;  p->n[2] = n.z;
}

/* The flatten of main.cpp:228-231 for a whole mesh: one Triangle per index triple.
 * mats/tri_mat select a material per triangle (tri_mat NULL: mats[0] for all;
 * mats NULL: reference default material). */
void rto_make_mesh(rto_prim *out, const float *verts, const int *indices, int n_tris,
                   const rto_material *mats, const int *tri_mat)
{
    for (int k = 0; k < n_tris; k++) {
        rto_make_triangle(&out[k], verts + 3 * (size_t)indices[3 * k], verts + 3 * (size_t)indices[3 * k + 1],
                          verts + 3 * (size_t)indices[3 * k + 2]);
        if (mats) out[k].mat = mats[tri_mat ? tri_mat[k] : 0];
    }
}

void rto_default_camera(rto_camera *c)                                                     /* main.cpp:267-269 */
{
    c->origin[0] = 0.0f; c->origin[1] = 0.0f; c->origin[2] = -5.0f;
    c->pitch = 0.01f; c->off_x = -4.0f; c->off_y = -4.0f; c->plane_z = 0.0f;
}

/* ---------------------------------------------------------------- intersectors */
static inline int sphere_raycast(const rto_prim *s, v3 o, v3 d, float *t)                  /* main.cpp:56-68 */
{
    v3 l = vsub(vld(s->a), o);
    float distToCenter = vdot(l, d);
    if (distToCenter < 0.0f)
        return 0;
    float distToIntersectSq = vdot(l, l) - distToCenter * distToCenter;
    if (distToIntersectSq > s->b[1])
        return 0;
    *t = distToCenter - sqrtf(s->b[1] - distToIntersectSq);
    return 1;                                   /* no t>0 test: negative t is a hit (main.cpp:66-67) */
}

static inline int plane_raycast(const rto_prim *p, v3 o, v3 d, float *t)                   /* main.cpp:84-92 */
{
    v3 n = vld(p->a);
    float ldotn = vdot(n, d);
    if (ldotn == 0.0f)
        return 0;
    *t = (p->b[0] - vdot(n, o)) / ldotn;
    return *t > 0.0f;
}

static inline int tri_raycast(const rto_prim *tr, v3 o, v3 d, float *t)                    /* main.cpp:109-129 */
{
    v3 e1 = vld(tr->b), e2 = vld(tr->c);
    v3 P = vcross(d, e2);
    float det = vdot(e1, P);
    if (det > -0.0001f && det < 0.0001f)
        return 0;
    float invdet = 1.0f / det;
    v3 T = vsub(o, vld(tr->a));
    float u = vdot(T, P) * invdet;
    if (u < 0.0f || u > 1.0f)
        return 0;
    v3 Q = vcross(T, e1);
    float v = vdot(d, Q) * invdet;
    if (v < 0.0f || u + v > 1.0f)
        return 0;
    *t = vdot(e2, Q) * invdet;
    return *t > 0.0001f;
}

static inline int prim_raycast(const rto_prim *p, v3 o, v3 d, float *t)
{
    switch (p->kind) {
    case RTO_SPHERE: return sphere_raycast(p, o, d, t);
    case RTO_PLANE:  return plane_raycast(p, o, d, t);
    default:         return tri_raycast(p, o, d, t);
    }
}

static inline v3 prim_normal(const rto_prim *p, v3 pos)       /* main.cpp:70-73, 94-97, 131-134 */
{
    switch (p->kind) {
    case RTO_SPHERE: return vnormalize(vsub(pos, vld(p->a)));
    case RTO_PLANE:  return vld(p->a);
    default:         return vld(p->n);
    }
}

int rto_tri_raycast(const rto_prim *p, const float o[3], const float d[3], float *t) { return tri_raycast(p, vld(o), vld(d), t); }
int rto_sphere_raycast(const rto_prim *p, const float o[3], const float d[3], float *t) { return sphere_raycast(p, vld(o), vld(d), t); }
int rto_plane_raycast(const rto_prim *p, const float o[3], const float d[3], float *t) { return plane_raycast(p, vld(o), vld(d), t); }

/* ---------------------------------------------------------------- scene + CPU BVH */
typedef struct {
    float lo[3], hi[3];
    int left, right;        /* internal: child node indices; leaf: left = -1 */
    int first, count;       /* leaf: range in tri_order */
} bvh_node;

struct rto_scene {
    rto_prim *prims;
    int n_prims;
    int *lights;  int n_lights;     /* prim ids with is_light, scene order (main.cpp:170-173) */
    int *linear;  int n_linear;     /* non-triangle prim ids, scene order */
    int *tris;    int n_tris;       /* triangle prim ids, scene order */
    /* BVH */
    bvh_node *nodes; int n_nodes;
    int *tri_order;                 /* prim ids, leaf order */
    /* opt-in shading normals (not in the reference): 9 floats per triangle, by triangle ordinal */
    float *vn;
    int *tri_ordinal;               /* prim id -> index among the triangles */
};

rto_scene *rto_scene_create(const rto_prim *prims, int n)
{
    rto_scene *s = (rto_scene *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->prims = (rto_prim *)malloc(sizeof(rto_prim) * (size_t)(n > 0 ? n : 1));
    s->lights = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    s->linear = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    s->tris = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    if (n > 0) memcpy(s->prims, prims, sizeof(rto_prim) * (size_t)n);
    s->n_prims = n;
    for (int i = 0; i < n; i++) {
        if (prims[i].is_light) s->lights[s->n_lights++] = i;
        if (prims[i].kind == RTO_TRIANGLE) s->tris[s->n_tris++] = i;
        else s->linear[s->n_linear++] = i;
    }
    return s;
}

void rto_scene_destroy(rto_scene *s)
{
    if (!s) return;
    free(s->prims); free(s->lights); free(s->linear); free(s->tris);
    free(s->nodes); free(s->tri_order);
    free(s->vn); free(s->tri_ordinal);
    free(s);
}

/* Shading normals for the triangles, in scene order: n_tris x (at v1, at v2, at v3) x (x,y,z).  The
 * reference parses `vn` and drops it (objloader.cpp:56-63, main.cpp:222-226), so this has no reference
 * behaviour to match; the rule restated here is the one the HIP path documents.  NULL switches back. */
int rto_scene_set_corner_normals(rto_scene *s, const float *vn, int n_tris)
{
    free(s->vn); s->vn = NULL;
    free(s->tri_ordinal); s->tri_ordinal = NULL;
    if (!vn) return 0;
    if (n_tris != s->n_tris || n_tris == 0) return -1;
    s->vn = (float *)malloc(sizeof(float) * 9 * (size_t)n_tris);
    s->tri_ordinal = (int *)malloc(sizeof(int) * (size_t)s->n_prims);
    if (!s->vn || !s->tri_ordinal) return -1;
    memcpy(s->vn, vn, sizeof(float) * 9 * (size_t)n_tris);
    for (int i = 0; i < s->n_prims; i++) s->tri_ordinal[i] = -1;
    for (int k = 0; k < s->n_tris; k++) s->tri_ordinal[s->tris[k]] = k;
    return 0;
}

typedef struct { float c[3]; int id; } bvh_item;
static int g_sort_axis;   /* build is single-threaded */
static int cmp_item(const void *a, const void *b)
{
    float x = ((const bvh_item *)a)->c[g_sort_axis], y = ((const bvh_item *)b)->c[g_sort_axis];
    if (x < y) return -1;
    if (x > y) return 1;
    return ((const bvh_item *)a)->id - ((const bvh_item *)b)->id;
}

static void tri_bounds(const rto_prim *t, float lo[3], float hi[3])
{
    /* The triangle the intersector sees is v1, v1+e1, v1+e2. */
    for (int k = 0; k < 3; k++) {
        float a = t->a[k], b = t->a[k] + t->b[k], c = t->a[k] + t->c[k];
        float mn = a < b ? a : b, mx = a > b ? a : b;
        if (c < mn) mn = c;
        if (c > mx) mx = c;
        float m = fmaxf(fabsf(mn), fabsf(mx));
        float pad = 1e-4f * (1.0f + m) + 1e-3f * (mx - mn);   /* generous: only prunes */
        lo[k] = mn - pad;
        hi[k] = mx + pad;
    }
}

static int build_rec(rto_scene *s, bvh_item *items, int first, int count)
{
    int me = s->n_nodes++;
    bvh_node *nd = &s->nodes[me];
    float clo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, chi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int k = 0; k < 3; k++) { nd->lo[k] = FLT_MAX; nd->hi[k] = -FLT_MAX; }
    for (int i = first; i < first + count; i++) {
        float lo[3], hi[3];
        tri_bounds(&s->prims[items[i].id], lo, hi);
        for (int k = 0; k < 3; k++) {
            if (lo[k] < nd->lo[k]) nd->lo[k] = lo[k];
            if (hi[k] > nd->hi[k]) nd->hi[k] = hi[k];
            if (items[i].c[k] < clo[k]) clo[k] = items[i].c[k];
            if (items[i].c[k] > chi[k]) chi[k] = items[i].c[k];
        }
    }
    if (count <= 4) {
        nd->left = -1; nd->right = -1; nd->first = first; nd->count = count;
        for (int i = first; i < first + count; i++) s->tri_order[i] = items[i].id;
        return me;
    }
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    g_sort_axis = axis;
    qsort(items + first, (size_t)count, sizeof(bvh_item), cmp_item);
    int half = count / 2;
    int l = build_rec(s, items, first, half);
    int r = build_rec(s, items, first + half, count - half);
    nd = &s->nodes[me];
    nd->left = l; nd->right = r; nd->first = first; nd->count = count;
    return me;
}

int rto_scene_build_bvh(rto_scene *s)
{
    free(s->nodes); free(s->tri_order);
    s->nodes = NULL; s->tri_order = NULL; s->n_nodes = 0;
    if (s->n_tris == 0) return 0;
    s->nodes = (bvh_node *)malloc(sizeof(bvh_node) * (size_t)(2 * s->n_tris));
    s->tri_order = (int *)malloc(sizeof(int) * (size_t)s->n_tris);
    bvh_item *items = (bvh_item *)malloc(sizeof(bvh_item) * (size_t)s->n_tris);
    if (!s->nodes || !s->tri_order || !items) { free(items); return -1; }
    for (int i = 0; i < s->n_tris; i++) {
        const rto_prim *t = &s->prims[s->tris[i]];
        for (int k = 0; k < 3; k++)
            items[i].c[k] = t->a[k] + (t->b[k] + t->c[k]) * (1.0f / 3.0f);
        items[i].id = s->tris[i];
    }
    build_rec(s, items, 0, s->n_tris);
    free(items);
    return 0;
}

/* Conservative ray/box: returns 1 and the entry distance when the ray's line
 * segment t in [0, +inf) may touch the box. */
static inline int ray_box(const bvh_node *n, v3 o, v3 d, float *tnear)
{
    float t0 = 0.0f, t1 = FLT_MAX;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    for (int k = 0; k < 3; k++) {
        if (dd[k] == 0.0f) {
            if (oo[k] < n->lo[k] || oo[k] > n->hi[k]) return 0;
        } else {
            float inv = 1.0f / dd[k];
            float a = (n->lo[k] - oo[k]) * inv, b = (n->hi[k] - oo[k]) * inv;
            if (a > b) { float w = a; a = b; b = w; }
            /* widen by a few ulp so rounding never rejects a touching ray */
            a -= fabsf(a) * 4e-7f; b += fabsf(b) * 4e-7f;
            if (a > t0) t0 = a;
            if (b < t1) t1 = b;
            if (t0 > t1) return 0;
        }
    }
    *tnear = t0;
    return 1;
}

/* ---------------------------------------------------------------- queries */
typedef struct { int id; float t; } hit_rec;

/* Nearest hit: main.cpp:147-156.  `nearest > t` in scene order == the lowest id among minimal t. */
static hit_rec nearest_hit(const rto_scene *s, v3 o, v3 d, int use_bvh, rto_counts *cn)
{
    hit_rec best = { -1, FLT_MAX };
    float t;
    if (!use_bvh || !s->nodes) {
        for (int i = 0; i < s->n_prims; i++) {
            if (prim_raycast(&s->prims[i], o, d, &t) && best.t > t) { best.t = t; best.id = i; }
        }
        if (cn) cn->prim_tests += (uint64_t)s->n_prims;
        return best;
    }
    for (int k = 0; k < s->n_linear; k++) {
        int i = s->linear[k];
        if (prim_raycast(&s->prims[i], o, d, &t) && best.t > t) { best.t = t; best.id = i; }
    }
    if (cn) cn->prim_tests += (uint64_t)s->n_linear;
    int stack[128], sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const bvh_node *n = &s->nodes[stack[--sp]];
        float tn;
        if (cn) cn->box_tests++;
        if (!ray_box(n, o, d, &tn) || tn > best.t) continue;
        if (n->left < 0) {
            for (int i = n->first; i < n->first + n->count; i++) {
                int id = s->tri_order[i];
                if (cn) cn->tri_tests++;
                if (tri_raycast(&s->prims[id], o, d, &t) &&
                    (best.t > t || (best.t == t && id < best.id))) { best.t = t; best.id = id; }
            }
        } else {
            stack[sp++] = n->left;
            stack[sp++] = n->right;
        }
    }
    return best;
}

/* Shadow query: main.cpp:181-188 — any non-light primitive whose Raycast returns
 * true, at ANY distance, blocks the light. */
static int occluded(const rto_scene *s, v3 o, v3 d, int use_bvh, rto_counts *cn)
{
    float t;
    if (!use_bvh || !s->nodes) {
        for (int i = 0; i < s->n_prims; i++) {
            if (cn) cn->prim_tests++;
            if (!s->prims[i].is_light && prim_raycast(&s->prims[i], o, d, &t)) return 1;
        }
        return 0;
    }
    for (int k = 0; k < s->n_linear; k++) {
        int i = s->linear[k];
        if (cn) cn->prim_tests++;
        if (!s->prims[i].is_light && prim_raycast(&s->prims[i], o, d, &t)) return 1;
    }
    int stack[128], sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const bvh_node *n = &s->nodes[stack[--sp]];
        float tn;
        if (cn) cn->box_tests++;
        if (!ray_box(n, o, d, &tn)) continue;
        if (n->left < 0) {
            for (int i = n->first; i < n->first + n->count; i++) {
                int id = s->tri_order[i];
                if (cn) cn->tri_tests++;
                if (!s->prims[id].is_light && tri_raycast(&s->prims[id], o, d, &t)) return 1;
            }
        } else {
            stack[sp++] = n->left;
            stack[sp++] = n->right;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- raytrace (main.cpp:144-217) */
static v3 raytrace(const rto_scene *s, v3 ro, v3 rd, int depth, int max_depth, int use_bvh,
                   hit_rec *first_hit, rto_counts *cn)
{
    hit_rec h = nearest_hit(s, ro, rd, use_bvh, cn);
    if (first_hit) *first_hit = h;
    if (h.id < 0)
        return V(0.0f, 0.0f, 0.0f);                                            /* main.cpp:159-160 */

    const rto_prim *np = &s->prims[h.id];
    v3 col = V(0.0f, 0.0f, 0.0f);
    v3 pos = vadd(ro, vmuls(rd, h.t));                                          /* main.cpp:163 */
    v3 N = prim_normal(np, pos);                                                /* main.cpp:164 */
    if (s->vn && np->kind == RTO_TRIANGLE) {
        /* N = normalize(n1*(1-u-v) + n2*u + n3*v), u and v as in Triangle::Raycast (main.cpp:110-124);
         * nine zeros keep the geometric normal */
        const float *q = s->vn + 9 * (size_t)s->tri_ordinal[h.id];
        int any = 0;
        for (int k = 0; k < 9; k++) any |= q[k] != 0.0f;
        if (any) {
            v3 e1 = vld(np->b), e2 = vld(np->c);
            v3 P = vcross(rd, e2);
            float invdet = 1.0f / vdot(e1, P);
            v3 T = vsub(ro, vld(np->a));
            float u = vdot(T, P) * invdet;
            v3 Q = vcross(T, e1);
            float v = vdot(rd, Q) * invdet;
            float w = 1.0f - u - v;
            N = vnormalize(vadd(vadd(vmuls(vld(q), w), vmuls(vld(q + 3), u)), vmuls(vld(q + 6), v)));
        }
    }
    v3 Mc = vld(np->mat.color);

    if (np->is_light) {
        col = Mc;                                                               /* main.cpp:166-167 */
    } else {
        for (int li = 0; li < s->n_lights; li++) {                              /* main.cpp:170-173 */
            const rto_prim *p = &s->prims[s->lights[li]];
            v3 Lc = vld(p->mat.color);
            float shade = 1.0f;
            v3 L = vnormalize(vsub(vld(p->a), pos));                            /* main.cpp:178 */
            v3 so = vadd(pos, vmuls(L, 0.01f));                                 /* main.cpp:179 */
            if (cn) cn->shadow++;
            if (occluded(s, so, L, use_bvh, cn))                                /* main.cpp:181-188 */
                shade = 0.0f;
            if (np->mat.diffuse > 0.0f) {                                       /* main.cpp:191-195 */
                float diffuse = vdot(N, L) * np->mat.diffuse;
                col = vadd(col, vmuls(vmuls(vmul(Lc, Mc), diffuse), shade));
            }
            if (np->mat.spec > 0.0f) {                                          /* main.cpp:198-204 */
                v3 R = vsub(L, vmuls(vmuls(N, vdot(L, N)), 2.0f));
                float dot = vdot(rd, R);
                if (dot > 0.0f)
                    col = vadd(col, vmuls(vmuls(vmuls(vmul(Lc, Mc), powf(dot, 20.0f)),
                                                np->mat.spec), shade));
            }
        }
        if (np->mat.reflect > 0.0f && depth < max_depth) {                      /* main.cpp:208-213 */
            v3 R = vsub(rd, vmuls(vmuls(N, 2.0f), vdot(rd, N)));
            if (cn) cn->reflect++;
            v3 reflectCol = raytrace(s, pos, R, depth + 1, max_depth, use_bvh, NULL, cn);
            col = vadd(col, vmuls(vmul(reflectCol, Mc), np->mat.reflect));
        }
    }
    return col;
}

void rto_raytrace(const rto_scene *s, const float origin[3], const float dir[3], int depth,
                  int max_depth, int use_bvh, float rgb_out[3], int32_t *hit_id, float *hit_t,
                  rto_counts *counts)
{
    hit_rec h;
    v3 c = raytrace(s, vld(origin), vld(dir), depth, max_depth, use_bvh, &h, counts);
    rgb_out[0] = c.x; rgb_out[1] = c.y; rgb_out[2] = c.z;
    if (hit_id) *hit_id = h.id;
    if (hit_t) *hit_t = h.t;
}

static inline float clamp01(float f) { return f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f); }  /* main.cpp:139-142 */

uint8_t rto_quantise(float c)                                                              /* main.cpp:273 */
{
    /* (char)(clamp01(c)*255.0f): the Debug (-O0) build converts float->int by
     * truncation and keeps the low byte; the value is in [0,255] so it is wrap-free.
     * A NaN passes clamp01 unchanged; cvttss2si then yields INT_MIN whose low byte is 0. */
    float f = clamp01(c) * 255.0f;
    if (!(f == f)) return 0;
    return (uint8_t)(int)f;
}

int rto_render(const rto_scene *s, const rto_camera *cam, int W, int H, int y0, int y1,
               int max_depth, int use_bvh, int n_threads, int x_step, int y_step,
               uint8_t *rgb8, float *rgb32, int32_t *hit_id, float *hit_t, rto_counts *counts)
{
    if (!s || !cam || W <= 0 || H <= 0 || y0 < 0 || y1 > H || y0 > y1 || max_depth < 0)
        return -1;
    if (x_step < 1) x_step = 1;
    if (y_step < 1) y_step = 1;
    rto_counts total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
#endif
    {
        rto_counts cn;
        memset(&cn, 0, sizeof cn);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int y = y0; y < y1; y++) {                                         /* main.cpp:262 */
            if (y % y_step) continue;
            for (int x = 0; x < W; x += x_step) {                               /* main.cpp:264 */
                v3 o = vld(cam->origin);                                        /* main.cpp:267 */
                float offsetX = x * cam->pitch + cam->off_x;                    /* main.cpp:268 */
                float offsetY = y * cam->pitch + cam->off_y;                    /* main.cpp:269 */
                v3 dir = vnormalize(vsub(V(offsetX, offsetY, cam->plane_z), o));/* main.cpp:270 */
                hit_rec h;
                cn.primary++;
                v3 col = raytrace(s, o, dir, 0, max_depth, use_bvh, &h, &cn);   /* main.cpp:272 */
                size_t idx = (size_t)(y - y0) * (size_t)W + (size_t)x;
                if (rgb8) {                                                     /* main.cpp:273 */
                    rgb8[3 * idx + 0] = rto_quantise(col.x);
                    rgb8[3 * idx + 1] = rto_quantise(col.y);
                    rgb8[3 * idx + 2] = rto_quantise(col.z);
                }
                if (rgb32) { rgb32[3 * idx] = col.x; rgb32[3 * idx + 1] = col.y; rgb32[3 * idx + 2] = col.z; }
                if (hit_id) hit_id[idx] = h.id;
                if (hit_t) hit_t[idx] = h.t;
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.primary += cn.primary; total.shadow += cn.shadow; total.reflect += cn.reflect;
            total.prim_tests += cn.prim_tests; total.box_tests += cn.box_tests; total.tri_tests += cn.tri_tests;
        }
    }
    if (counts) *counts = total;
    return 0;
}

/* ---------------------------------------------------------------- OBJ ingest (objloader.cpp:18-105) */
/* std::getline(stream, tok, delim) on the remaining text [*cur, end): fails (returns 0)
 * when no characters remain; otherwise extracts up to and consuming the delimiter. */
static int next_token(const char **cur, const char *end, char delim, char *tok, size_t cap)
{
    tok[0] = 0;
    if (*cur >= end) return 0;
    const char *p = *cur;
    size_t n = 0;
    while (p < end && *p != delim) { if (n + 1 < cap) tok[n++] = *p; p++; }
    tok[n] = 0;
    *cur = (p < end) ? p + 1 : end;
    return 1;
}

int rto_load_obj(const char *path, float **verts_out, int *n_verts, int **idx_out, int *n_idx)
{
    *verts_out = NULL; *idx_out = NULL; *n_verts = 0; *n_idx = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return 0;                                     /* objloader.cpp:22,104: silent */
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 1);
    if (!buf) { fclose(f); return -1; }
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { free(buf); fclose(f); return -1; }
    fclose(f);
    size_t vcap = 1024, icap = 4096, nv = 0, ni = 0;
    float *verts = (float *)malloc(sizeof(float) * 3 * vcap);
    int *inds = (int *)malloc(sizeof(int) * icap);
    int rc = 0;
    const char *p = buf, *fend = buf + sz;
    char tok[256], a[256], b[256], c[256];
    while (p < fend) {                                    /* objloader.cpp:24 */
        const char *le = p;
        while (le < fend && *le != '\n') le++;
        const char *cur = p;
        p = (le < fend) ? le + 1 : fend;
        next_token(&cur, le, ' ', tok, sizeof tok);       /* objloader.cpp:27 */
        if (!strcmp(tok, "v")) {                          /* objloader.cpp:29-40 */
            next_token(&cur, le, ' ', a, sizeof a);
            next_token(&cur, le, ' ', b, sizeof b);
            next_token(&cur, le, ' ', c, sizeof c);
            if (nv == vcap) { vcap *= 2; verts = (float *)realloc(verts, sizeof(float) * 3 * vcap); }
            verts[3 * nv + 0] = (float)atof(a);
            verts[3 * nv + 1] = (float)atof(b);
            verts[3 * nv + 2] = (float)atof(c);
            nv++;
        } else if (!strcmp(tok, "f")) {                   /* objloader.cpp:64-100 */
            int q[4];
            for (int i = 0; i < 4; i++) {
                char bits[256], bit[256];
                next_token(&cur, le, ' ', bits, sizeof bits);
                const char *bc = bits, *be = bits + strlen(bits);
                if (next_token(&bc, be, '/', bit, sizeof bit)) {
                    q[i] = atoi(bit) - 1;                 /* objloader.cpp:77 */
                } else {
                    if (i != 3) rc = -2;                  /* objloader.cpp:83 assert(i == 3) */
                    q[i] = -1;
                }
            }
            if (ni + 6 > icap) { icap *= 2; inds = (int *)realloc(inds, sizeof(int) * icap); }
            inds[ni++] = q[0]; inds[ni++] = q[1]; inds[ni++] = q[2];
            if (q[3] != -1) { inds[ni++] = q[0]; inds[ni++] = q[2]; inds[ni++] = q[3]; }
        }
        /* "vt"/"vn" are parsed by the reference but never reach the tracer (main.cpp:222-231) */
    }
    free(buf);
    *verts_out = verts; *idx_out = inds; *n_verts = (int)nv; *n_idx = (int)ni;
    return rc;
}

void rto_free(void *p) { free(p); }
