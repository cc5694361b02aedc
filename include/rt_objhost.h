/*
 * rt_objhost.h — C ABI over the host's OBJ loader (libobjhost.so), for callers that are not C++
 * (bench.py loads a driver-supplied sponza.obj through it).
 *
 * It exposes LoadModel of Raytracer/objloader.h:15 (parser Raytracer/objloader.cpp:18-105) as the
 * host restates it in raytracer_amd/host/objloader.cpp: `relaxed` = 0 keeps the reference's exact
 * tokenizer (single spaces, quads fan-split, everything else ignored); `relaxed` = 1 is the tolerant
 * parser with `mtllib` / `usemtl` materials.  Plain pointers and sizes; host memory only, no GPU.
 */
#ifndef RT_OBJHOST_H
#define RT_OBJHOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_obj rt_obj;

/* Parse `path`.  Never NULL for a readable or missing file (a missing file is an empty mesh, as
 * objloader.cpp:22,104); NULL only when out of memory. */
rt_obj *rt_obj_load(const char *path, int relaxed);
void rt_obj_free(rt_obj *o);

/* counts[0] = vertices, [1] = indices (3 per triangle), [2] = materials (>= 1; [0] is Material(),
 * Raytracer/main.cpp:34), [3] = malformed or dropped faces. */
void rt_obj_counts(const rt_obj *o, int32_t counts[4]);

/* Copy out: verts 3 floats each; indices 0-based; tri_material one per triangle; materials 6 floats
 * each (reflect, diffuse, spec, r, g, b — struct Material, main.cpp:29-36).  Any pointer may be NULL. */
void rt_obj_copy(const rt_obj *o, float *verts, int32_t *indices, int32_t *tri_material, float *materials);

#ifdef __cplusplus
}
#endif
#endif
