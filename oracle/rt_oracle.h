/*
 * rt_oracle.h — CPU restatement of the reference tracer. TEST INFRASTRUCTURE ONLY.
 *
 * This is the checker the HIP path is compared against.  It is never the thing
 * shipped or measured: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.
 *
 * Every function restates, operation for operation and in IEEE float32, the
 * reference's CPU trace path:
 *   Raytracer/main.cpp:20-135   Ray / Material / Sphere / Plane / Triangle
 *   Raytracer/main.cpp:139-217  clamp01, raytrace
 *   Raytracer/main.cpp:262-273  per-pixel camera loop + 8-bit store
 *   Raytracer/maths.h:89-173    vec3 operators (dot is left-assoc, normalize divides)
 *
 * Parity pinning: the restatement is checked byte-for-byte against the
 * unmodified reference built from /root/reference (oracle/ref_harness.cpp,
 * oracle/build_ref.sh) and against the fixtures in tests/golden/ that were
 * generated from that build (tests/golden/make_golden.py).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Raytracer/main.cpp:29-36 (defaults: reflect 0, diffuse 1, spec 1, white) */
typedef struct {
    float reflect, diffuse, spec;
    float color[3];
} rto_material;

enum { RTO_SPHERE = 0, RTO_PLANE = 1, RTO_TRIANGLE = 2 };

/* One entry of the reference's `std::vector<Primitive*> scene` (main.cpp:137).
 * The index in the array is the primitive id; array order is scene order. */
typedef struct {
    int kind;
    int is_light;          /* main.cpp:42; only spheres may be lights (main.cpp:178 casts to Sphere*) */
    rto_material mat;
    float a[3];            /* sphere: pos      | plane: normal | triangle: v1 */
    float b[3];            /* sphere: r, r*r,- | plane: offset | triangle: e1 = v2-v1 */
    float c[3];            /*                  |               | triangle: e2 = v3-v1 */
    float n[3];            /*                  |               | triangle: N = normalize(e1 x e2) */
} rto_prim;

/* Pinhole camera of main.cpp:267-270, with the three literals made parameters:
 *   offsetX = x*pitch + off_x ; offsetY = y*pitch + off_y   (reference: 0.01f, -4.0f, -4.0f)
 *   dir     = normalize((offsetX, offsetY, plane_z) - origin)   (reference: plane_z 0, origin (0,0,-5)) */
typedef struct {
    float origin[3];
    float pitch, off_x, off_y, plane_z;
} rto_camera;

typedef struct {
    uint64_t primary, shadow, reflect;   /* rays as the reference would cast them */
    uint64_t prim_tests;                 /* Raycast() calls actually made (brute force) */
    uint64_t box_tests, tri_tests;       /* when the CPU BVH is used */
} rto_counts;

void rto_default_material(rto_material *m);
void rto_make_sphere(rto_prim *p, const float pos[3], float radius);          /* main.cpp:53 */
void rto_make_plane(rto_prim *p, const float normal[3], float offset);        /* main.cpp:81 */
void rto_make_triangle(rto_prim *p, const float v1[3], const float v2[3],
                       const float v3[3]);                                    /* main.cpp:104-107 */
void rto_make_mesh(rto_prim *out, const float *verts, const int *indices, int n_tris,
                   const rto_material *mats, const int *tri_mat);             /* main.cpp:228-231 */
void rto_default_camera(rto_camera *c);

/* Opaque scene: primitives in scene order + optional CPU BVH over the triangles. */
typedef struct rto_scene rto_scene;
rto_scene *rto_scene_create(const rto_prim *prims, int n_prims);
void rto_scene_destroy(rto_scene *s);
/* Build a median-split BVH over the scene's triangles (spheres/planes stay in a
 * linear list).  The BVH only prunes; every surviving triangle goes through the
 * same Raycast arithmetic and the same first-in-scene-order tie-break, so the
 * result equals brute force unless a padded box wrongly culls a hit. */
int rto_scene_build_bvh(rto_scene *s);

/* Opt-in shading normals (no reference behaviour: the reference parses `vn` and drops it,
 * objloader.cpp:56-63, main.cpp:222-226).  vn = n_tris x 3 corners x (x,y,z), triangles in scene
 * order; NULL switches back to Triangle::GetNormal (main.cpp:131-134).  Returns 0, or -1 on a
 * count mismatch. */
int rto_scene_set_corner_normals(rto_scene *s, const float *vn, int n_tris);

/* Trace one ray exactly as raytrace() (main.cpp:144-217) with `max_depth` in place
 * of the constant at main.cpp:18.  Outputs colour and the nearest primitive / t of
 * THIS ray (hit_id -1 on a miss). */
void rto_raytrace(const rto_scene *s, const float origin[3], const float dir[3], int depth,
                  int max_depth, int use_bvh, float rgb_out[3], int32_t *hit_id, float *hit_t,
                  rto_counts *counts);

/* Render rows [y0,y1) of a W x H frame (main.cpp:262-277).  Any output may be NULL.
 * Buffers are compact: row y lands at index (y-y0)*W + x, as image[y*W+x] in the
 * reference.  x_step/y_step > 1 render only the lattice x%x_step==0, y%y_step==0
 * (other pixels are left untouched).  n_threads <= 1 is single-threaded like the
 * reference; > 1 uses OpenMP over rows. */
int rto_render(const rto_scene *s, const rto_camera *cam, int W, int H, int y0, int y1,
               int max_depth, int use_bvh, int n_threads, int x_step, int y_step,
               uint8_t *rgb8, float *rgb32, int32_t *hit_id, float *hit_t, rto_counts *counts);

/* Triangle::Raycast restated (main.cpp:109-129); exposed for unit tests. */
int rto_tri_raycast(const rto_prim *tri, const float o[3], const float d[3], float *t);
int rto_sphere_raycast(const rto_prim *sph, const float o[3], const float d[3], float *t);
int rto_plane_raycast(const rto_prim *pl, const float o[3], const float d[3], float *t);

/* Quantise as the reference's Debug build does (main.cpp:139-142, 273): clamp to
 * [0,1], multiply by 255.0f, truncate toward zero, keep the low byte. */
uint8_t rto_quantise(float c);

/* Wavefront OBJ ingest restated from objloader.cpp:18-105 (single-space tokens,
 * position index only, quad fan split, silent on a missing file).  Returns
 * malloc'ed arrays the caller frees with rto_free. */
int rto_load_obj(const char *path, float **verts_xyz, int *n_verts, int **indices, int *n_indices);
void rto_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
