/*
 * rt_shim.h — C ABI of the MI355X trace path (librt_amd.so).
 *
 * The reference (zanders3/raytracer) has no plugin or FFI interface; its trace
 * path is reached through plain C++ seams with global state.  Each entry point
 * below replaces one of those seams, cited as Raytracer/<file>:<line>:
 *
 *   scene construction   main.cpp:240-255 (`scene.push_back(new Sphere/Plane…)`)
 *   mesh flatten         main.cpp:219-232 (`LoadModel(const char*)` → `new Triangle`)
 *   camera literals      main.cpp:267-270
 *   per-pixel loop       main.cpp:262-277 (calls raytrace(), main.cpp:144-217, and
 *                        quantises into `color image[]`, texturerenderer.h:14-16)
 *
 * Conventions: plain pointers and sizes only.  Every call returns 0 on success
 * or a negative RT_E_* code; rt_last_error() gives the text for the calling
 * thread.  The host owns host buffers, the library owns device buffers.  One
 * host thread per context.  There is no CPU fallback: without a usable HIP
 * device rt_create() fails.
 */
#ifndef RT_SHIM_H
#define RT_SHIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: rt_stats grew (ms_traverse, bvh_sah_cost, bvh_refs, box_tests_occupied, staging_ms) and rt_get_stats_sized exists.
 * A host checks rt_abi_version() == RT_ABI_VERSION once at start-up; rt_get_stats writes sizeof(rt_stats) bytes of THIS
 * header's layout, so a host built against an older header must call rt_get_stats_sized with its own sizeof instead. */
#define RT_ABI_VERSION 2

enum {
    RT_OK = 0,
    RT_E_INVALID = -1,     /* bad argument */
    RT_E_DEVICE = -2,      /* HIP runtime error */
    RT_E_NOMEM = -3,       /* device or host allocation failed */
    RT_E_STATE = -4,       /* call order (e.g. render before a scene is set) */
    RT_E_LIMIT = -5        /* size beyond a stated maximum */
};

typedef struct rt_ctx rt_ctx;

/* struct Material, main.cpp:29-36.  Reference defaults: 0, 1, 1, (1,1,1). */
typedef struct {
    float reflect, diffuse, spec;
    float color[3];
} rt_material;

enum { RT_SPHERE = 0, RT_PLANE = 1 };

/* One analytic entry of the reference's `scene` vector (main.cpp:137):
 * struct Sphere main.cpp:48-74 (p = centre, s = radius) or
 * struct Plane  main.cpp:76-98 (p = normal, s = offset).
 * is_light is Primitive::isLight (main.cpp:42); only spheres may be lights because
 * main.cpp:178 reads the light position through a Sphere*. */
typedef struct {
    int32_t kind;
    int32_t is_light;
    rt_material mat;
    float p[3];
    float s;
} rt_analytic;

/* The camera of main.cpp:267-270 with its literals as fields:
 *   offsetX = x*pitch + off_x, offsetY = y*pitch + off_y,
 *   direction = normalize((offsetX, offsetY, plane_z) - origin).
 * Reference values: origin (0,0,-5), pitch 0.01f, off_x = off_y = -4.0f, plane_z 0. */
typedef struct {
    float origin[3];
    float pitch, off_x, off_y, plane_z;
} rt_camera;

/* Which image rows a call renders, in bands (for tile partitioning across GPUs).
 * Local row r maps to image row
 *   y = (first_band + (r / band_rows) * band_stride) * band_rows + r % band_rows
 * and n_rows local rows are rendered into a compact buffer (row r at r*W).
 * A contiguous slab [y0,y1) is {first_band=y0, band_rows=1, band_stride=1, n_rows=y1-y0}. */
typedef struct {
    int32_t first_band, band_rows, band_stride, n_rows;
} rt_rowset;

/* Filled by rt_get_stats().  Ray counts follow the reference's own casting rules:
 * one primary ray per pixel (main.cpp:270-272), one shadow ray per light per
 * non-emissive hit (main.cpp:170-188), one reflection ray per hit with reflect > 0
 * and depth < maxDepth (main.cpp:208-211). */
typedef struct {
    uint64_t rays_primary, rays_shadow, rays_reflect;
    uint64_t box_tests, tri_tests, analytic_tests;   /* only with RT_OPT_COUNT_TESTS = 1 */
    float ms_frame;        /* device time of the last frame, first kernel to last kernel */
    float ms_trace;        /* everything before the last kernel: traversal and ray generation (staged pipeline), or the
                              per-level bounce kernels (fused) */
    float ms_resolve;      /* the last kernel: shading of every level + fold + quantise (staged), or fold + quantise (fused) */
    int32_t n_launches;    /* kernels launched for the frame */
    int32_t n_analytic, n_triangles;
    int32_t bvh_nodes, bvh_depth, bvh_leaf_max;
    int32_t accel;         /* 0 = linear scan over triangles, 1 = LBVH */
    float ms_build;        /* device time of the last LBVH build */
    float ms_traverse;     /* LBVH scenes: sum of the traversal kernels alone (part of ms_trace) */
    float bvh_sah_cost;    /* surface-area-heuristic cost of the hierarchy as built: (sum of inner-node areas + sum of leaf areas x
                              triangles in the leaf) / root area; lower is better, refits do not update it */
    int32_t bvh_refs;      /* triangle references under the hierarchy (> n_triangles when large triangles were cut, RT_OPT_SPLIT) */
    /* -- fields below exist from RT_ABI_VERSION 2 on; new fields are only ever appended -- */
    uint64_t box_tests_occupied;   /* with RT_OPT_COUNT_TESTS: the box tests that met a child slot holding a child (box_tests charges
                                      all four slots of every node step; slots of nodes with two or three children hold nothing) */
    float staging_ms;      /* rt_render_frame / rt_render_frame_planes: host time of the last call's device-to-host hand-over, from
                              the moment the device finished the frame until the caller's buffers were filled (copies into the
                              pinned staging buffer + host copies on into the caller's memory); 0 for the device-buffer entry points */
    int32_t reserved0;
} rt_stats;

enum {
    RT_OPT_ACCEL = 1,        /* 0 auto (LBVH above RT_BRUTE_MAX triangles), 1 force linear scan, 2 force LBVH */
    RT_OPT_COUNT_TESTS = 2,  /* 1: count box / triangle / analytic tests (slower kernels) */
    RT_OPT_LEAF_MAX = 3,     /* triangles per LBVH leaf, 1..8 (takes effect at the next rt_set_*triangles) */
    RT_OPT_TIMING = 4,       /* 1 (default): record per-kernel device events for rt_get_stats */
    RT_OPT_PIPELINE = 5,     /* LBVH scenes: 0 auto (= 2), 1 one fused kernel per bounce level, 2 separate traversal / ray-generation / shading stages */
    RT_OPT_BUILDER = 6,      /* how the hierarchy is built (always on the GPU): 0 auto (= 3), 1 Karras LBVH over Morton codes, 2 PLOC
                                agglomeration over the Morton order, 3 top-down binned surface-area-heuristic sweep */
    RT_OPT_SPLIT = 8,        /* hierarchy build: the most cells per axis a large triangle is cut into (each cell it reaches becomes one
                                reference with a clipped box; hits are unchanged): 0 auto (= 32), 1 never cut, up to 32; takes effect at
                                the next build */
    RT_OPT_GRAPH = 7,        /* 1 (default): with RT_OPT_TIMING and RT_OPT_COUNT_TESTS off, a frame asked for twice in a row is captured as a hipGraph and replayed while nothing changes */
    RT_OPT_CLAIM_ORDER = 9   /* LBVH scenes, staged pipeline.  A traversal launch ends with its longest rays; when those start last the GPU
                                idles behind them.  1 (default): when a layout (frame size, rows, depth, light count, primitives set) is
                                asked for the second time in a row, that frame is preceded by one counting pass of the same frame that
                                records the longest ray of every 64-ray work unit (and by two short timing runs that pick the camera-ray
                                kernel); every later frame of that layout hands the units out longest first.  The order decides when a
                                ray is traced, never what it hits: frames are bit-identical with it on or off, and a moved camera or
                                refitted mesh keeps using it (its worth fades as the view moves: it records which units hold the longest rays).  That one frame costs about four frames and the
                                call waits for it (one host synchronisation); a caller that renders a single frame never pays it.
                                0: off.  2: drop the order; the next repeating layout makes a new one (then back to 1). */
};

#define RT_MAX_ANALYTIC 1024
#define RT_MAX_DEPTH 64
#define RT_BRUTE_MAX 48

int rt_abi_version(void);
const char *rt_last_error(void);

/* Context on HIP device `device_ordinal`. */
rt_ctx *rt_create(int device_ordinal);
void rt_destroy(rt_ctx *ctx);

int rt_set_option(rt_ctx *ctx, int option, int value);

/* Replaces the `scene.push_back(new Sphere…/new Plane…)` block, main.cpp:240-255.
 * Primitive ids 0..n-1 in array order (scene order decides ties, main.cpp:151). */
int rt_set_analytic(rt_ctx *ctx, const rt_analytic *prims, int n);

/* Replaces LoadModel(const char*), main.cpp:219-232: every index triple becomes a
 * Triangle (constructor main.cpp:104-107 runs on the device), appended after the
 * analytic primitives (ids n_analytic + k).  `mat` NULL = reference default material.
 * Builds the LBVH on the device.  verts = n_verts x (x,y,z).  The call replaces the mesh: when it fails (an index
 * outside the vertices -> RT_E_INVALID, where the reference reads out of bounds) the context holds no mesh at all. */
int rt_set_mesh(rt_ctx *ctx, const float *verts, int n_verts, const int32_t *indices, int n_indices,
                const rt_material *mat);

/* Same with the triangles already flattened: tri_verts = n_tris x (v1,v2,v3) x (x,y,z). */
int rt_set_triangles(rt_ctx *ctx, const float *tri_verts, int n_tris, const rt_material *mat);

/* Per-triangle materials (the reference gives every mesh triangle the default
 * material, main.cpp:230 + 34; this is the data-driven generalisation):
 * tri_material[k] indexes mats[]. */
int rt_set_triangle_materials(rt_ctx *ctx, const rt_material *mats, int n_mats,
                              const int32_t *tri_material, int n_tris);

/* Opt-in shading normals from the `vn` statements the reference parses and never uses
 * (objloader.cpp:56-63; main.cpp:222-226 drops them; Triangle::GetNormal, main.cpp:131-134, returns the
 * geometric normal).  corner_normals = n_tris x (at v1, at v2, at v3) x (x,y,z).  With them set, the
 * normal of a triangle hit is normalize(n1*(1-u-v) + n2*u + n3*v) with u, v as Triangle::Raycast computes
 * them (main.cpp:119, 124); a triangle whose nine values are all zero keeps its geometric normal.  This
 * departs from the reference's image, so it is off until called; NULL turns it off again.  Replaced
 * meshes (rt_set_mesh / rt_set_triangles) drop the normals; rt_update_* keep them. */
int rt_set_triangle_normals(rt_ctx *ctx, const float *corner_normals, int n_tris);

/* Texture coordinates from the `vt` statements the reference parses and never uses (objloader.cpp:48-55;
 * the reference has no textures): corner_uv = n_tris x (at v1, at v2, at v3) x (u,v).  Nothing in the image
 * reads them; their one consumer is the uv plane of rt_render_frame_planes.  NULL drops them; a replaced mesh
 * drops them, rt_update_* keep them. */
int rt_set_triangle_texcoords(rt_ctx *ctx, const float *corner_uv, int n_tris);

/* Moving geometry for the per-tick re-render (draw(dt), main.cpp:284-287, glwt.mm:68-82; the reference
 * itself redraws one static frame): new positions for the triangles already loaded, same count and
 * order.  Materials and the hierarchy's topology stay; the Triangle constructor (main.cpp:104-107) runs
 * again and the hierarchy's boxes are refitted on the device (rt_stats.ms_build then reports the refit).
 * rt_update_triangles pairs with rt_set_triangles (n_tris x 9 floats), rt_update_mesh_vertices with
 * rt_set_mesh (n_verts x 3 floats under the indices kept from that call).  RT_E_STATE on a count mismatch. */
int rt_update_triangles(rt_ctx *ctx, const float *tri_verts, int n_tris);
int rt_update_mesh_vertices(rt_ctx *ctx, const float *verts, int n_verts);

/* Replaces the literals of main.cpp:267-270.  NULL restores the reference camera. */
int rt_set_camera(rt_ctx *ctx, const rt_camera *cam);

/* Replaces the pixel loop main.cpp:262-277 for rows [y0,y1) of a W x H frame with
 * `max_depth` in place of the constant at main.cpp:18.  Synchronous; outputs are
 * HOST buffers, any may be NULL:
 *   rgb8    3*W*(y1-y0) bytes, pixel (x,y) at 3*((y-y0)*W+x), as `color image[]`
 *   hit_id  W*(y1-y0) int32: index in scene order of the primary ray's nearest primitive, -1 = miss
 *   rgb32   3*W*(y1-y0) floats: the colour before clamp/quantise */
int rt_render_frame(rt_ctx *ctx, int W, int H, int y0, int y1, int max_depth,
                    uint8_t *rgb8, int32_t *hit_id, float *rgb32);

/* Host memory for the frame that the device can write directly (pinned; no counterpart in the reference, which
 * allocates its image with `new color[w*h]`, Raytracer/main.cpp:259, once, and keeps it).  An output pointer of
 * rt_render_frame / rt_render_frame_planes that lies wholly inside such a buffer is filled by one device-to-host copy
 * at bus rate; any other host pointer works too and is filled through the context's own pinned staging buffer plus
 * one more host copy.  (Registering memory the caller allocated itself is not offered: the runtime's host registration
 * is not available on every system this library runs on.)  NULL on failure (rt_last_error).  The buffer lives until
 * rt_free_host_buffer or rt_destroy. */
void *rt_alloc_host_buffer(rt_ctx *ctx, uint64_t bytes);
int rt_free_host_buffer(rt_ctx *ctx, void *buffer);

/* The same frame with two more planes for tooling (replaces nothing in the reference, which keeps only
 * the 8-bit colour, Raytracer/main.cpp:273): hit_t = the camera ray's nearest t (FLT_MAX on a miss,
 * main.cpp:147-156), normal = GetNormal at that hit (main.cpp:164; 0,0,0 on a miss), 3 floats per
 * pixel.  Any pointer may be NULL.  Synchronous; host buffers; rows in buffer order like rt_render_frame. */
typedef struct rt_frame_planes {
    uint8_t *rgb8;
    int32_t *hit_id;
    float *rgb32;
    float *hit_t;
    float *normal;
    float *uv;          /* 2 floats per pixel: the texture coordinate at the camera ray's hit, interpolated with Triangle::Raycast's
                           u, v (main.cpp:119, 124); (0,0) on a miss, on a sphere or plane, or without rt_set_triangle_texcoords */
} rt_frame_planes;
int rt_render_frame_planes(rt_ctx *ctx, int W, int H, int y0, int y1, int max_depth, const rt_frame_planes *out);

/* Same work with DEVICE output buffers and an explicit HIP stream (hipStream_t as
 * void*, NULL = the context's own stream).  Asynchronous: returns after enqueueing.
 * hit_t (optional) receives the primary ray's nearest t (FLT_MAX on a miss). */
int rt_render_rows_device(rt_ctx *ctx, int W, int H, const rt_rowset *rows, int max_depth,
                          void *d_rgb8, void *d_hit_id, void *d_rgb32, void *d_hit_t, void *stream);

/* Frames in flight.  No counterpart in the reference (its setup() renders one frame, main.cpp:262-277).  The same call
 * as rt_render_rows_device on frame slot `slot` (0 <= slot < RT_MAX_FRAME_SLOTS; rt_render_rows_device is slot 0).
 * Every slot has its own workspace (ray queues, hit and occlusion planes, counters), captured graph and completion
 * event, so frames of DIFFERENT slots may be in flight at the same time on different streams over one scene and one
 * hierarchy; frames of one slot are ordered one behind the other.  A rank's share of a partitioned frame is too small to
 * fill the GPU (fewer rays than resident lanes, six dependent launches): two to four slots nearly double its frame
 * rate.  rt_get_stats reports the slot rendered last; the scene-changing calls wait for every slot. */
#define RT_MAX_FRAME_SLOTS 8
int rt_render_rows_device_slot(rt_ctx* ctx, int slot, int W, int H, const rt_rowset* rows, int max_depth,
                               void* d_rgb8, void* d_hit_id, void* d_rgb32, void* d_hit_t, void* stream);

/* The same frame as rt_render_rows_device for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference, which allocates its one image with `new` (Raytracer/main.cpp:259) and nothing per ray.
 * Calls no hipMalloc / hipFree and creates no events or graphs: the ray queues, per-level records and
 * counters of the frame live in a static device array of RT_NOALLOC_SCRATCH_BYTES that every GPU the
 * library runs on holds until the process ends.  The frame runs as one kernel per bounce level (the
 * pipeline RT_OPT_PIPELINE = 1 selects), whatever RT_OPT_PIPELINE says; outputs are bit-identical to
 * rt_render_rows_device's.  Limits: RT_E_LIMIT when
 *   256 + 16*(max_depth+1)*W*n_rows + 2*(32*Q + 4*(Q/64+1)) > RT_NOALLOC_SCRATCH_BYTES (each term rounded up
 *   to 256 bytes), Q = W*n_rows padded to whole 8x8 pixel tiles, rounded up to a multiple of 256
 * (800x600 at depth 3 and 512x512 at depth 4 fit; render a larger frame in row sets, one call each);
 * RT_E_STATE when the mesh needs a hierarchy that is not built for the current options (it is built by
 * rt_set_mesh / rt_set_triangles; options changed since take effect in rt_render_rows_device).
 * All contexts on one GPU share that GPU's array: calls that overlap on one GPU race, so keep them on one
 * stream or order them yourself.  RT_OPT_COUNT_TESTS and RT_OPT_TIMING do not apply (rt_get_stats reports
 * ray counts, no times). */
#define RT_NOALLOC_SCRATCH_BYTES (64u << 20)
int rt_render_rows_device_noalloc(rt_ctx *ctx, int W, int H, const rt_rowset *rows, int max_depth,
                                  void *d_rgb8, void *d_hit_id, void *d_rgb32, void *d_hit_t, void *stream);

/* Scatter a gathered set of compact band buffers back into a W x H image:
 * src holds n_parts consecutive parts, part k rendered with
 * {first_band=k, band_rows, band_stride=n_parts}, each padded to rows_per_part rows.
 * bytes_per_pixel: 3 (rgb8), 4 (hit_id) or 12 (rgb32).  Device pointers, async on stream. */
int rt_deinterleave_bands(rt_ctx *ctx, const void *d_src, void *d_dst, int W, int H, int band_rows,
                          int n_parts, int rows_per_part, int bytes_per_pixel, void *stream);

/* Waits for the context's outstanding work and reports counters/timings of the last frame. */
int rt_get_stats(rt_ctx *ctx, rt_stats *out);

/* The same for a caller that may have been compiled against another version of this header (no counterpart in the
 * reference): writes at most `size` bytes of rt_stats, i.e. the fields the caller's struct has; fields are only ever
 * appended, so a prefix is always a valid older rt_stats.  RT_E_INVALID when size is 0 or not a multiple of 4. */
int rt_get_stats_sized(rt_ctx *ctx, rt_stats *out, uint64_t size);

/* Wait for everything the context has enqueued. */
int rt_synchronize(rt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
