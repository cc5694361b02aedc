// objhost_c.cpp — the C entry points of include/rt_objhost.h over this host's LoadModel / LoadModelEx
// (the reference's geometry source, Raytracer/objloader.h:15).
#include "../../include/rt_objhost.h"

#include <stddef.h>

#include <new>

#include "objloader.h"

struct rt_obj {
    ObjModel m;
};

extern "C" {

rt_obj *rt_obj_load(const char *path, int relaxed)
{
    rt_obj *o = new (std::nothrow) rt_obj();
    if (!o) return nullptr;
    try {                               // (the parser's vectors may run out of memory: NULL, never an exception through a C ABI)
    if (relaxed) {
        (void)LoadModelEx(path, o->m);
    } else {
        o->m.dropped_faces = LoadModel(path, o->m.vertices, o->m.texcoords, o->m.normals, o->m.indices);
        o->m.tri_material.assign(o->m.indices.size() / 3, 0);
    }
    if (o->m.materials.empty()) {      // Material(), Raytracer/main.cpp:34
        ObjMaterial d = {};
        d.reflect = 0.0f; d.diffuse = 1.0f; d.spec = 1.0f;
        d.color[0] = d.color[1] = d.color[2] = 1.0f;
        o->m.materials.push_back(d);
    }
    } catch (...) {
        delete o;
        return nullptr;
    }
    return o;
}

void rt_obj_free(rt_obj *o) { delete o; }

void rt_obj_counts(const rt_obj *o, int32_t counts[4])
{
    counts[0] = (int32_t)o->m.vertices.size();
    counts[1] = (int32_t)o->m.indices.size();
    counts[2] = (int32_t)o->m.materials.size();
    counts[3] = (int32_t)o->m.dropped_faces;
}

void rt_obj_copy(const rt_obj *o, float *verts, int32_t *indices, int32_t *tri_material, float *materials)
{
    const ObjModel &m = o->m;
    if (verts)
        for (size_t k = 0; k < m.vertices.size(); k++) {
            verts[3 * k] = m.vertices[k].x; verts[3 * k + 1] = m.vertices[k].y; verts[3 * k + 2] = m.vertices[k].z;
        }
    if (indices)
        for (size_t k = 0; k < m.indices.size(); k++) indices[k] = m.indices[k];
    if (tri_material)
        for (size_t k = 0; k < m.indices.size() / 3; k++) tri_material[k] = k < m.tri_material.size() ? m.tri_material[k] : 0;
    if (materials)
        for (size_t k = 0; k < m.materials.size(); k++) {
            const ObjMaterial &a = m.materials[k];
            float *d = materials + 6 * k;
            d[0] = a.reflect; d[1] = a.diffuse; d[2] = a.spec; d[3] = a.color[0]; d[4] = a.color[1]; d[5] = a.color[2];
        }
}

}  // extern "C"
