// objloader.h — Wavefront OBJ ingest, same entry point as the reference
// (Raytracer/objloader.h:15).
#pragma once

#include <vector>

#include "maths.h"

// Appends to the caller's vectors.  Behaviour follows Raytracer/objloader.cpp:18-105:
// tokens are split on single spaces, only the position index of each face corner is
// kept (text before the first '/'), indices become 0-based, a fourth corner makes a quad
// that is fan-split into (0,1,2) + (0,2,3), every other statement is ignored and a
// missing file is silently an empty mesh.  Returns the number of malformed face lines
// (the reference asserts on those, objloader.cpp:83); 0 for a clean file.
int LoadModel(const char *objFile, std::vector<vec3> &vertices, std::vector<vec2> &texcoords,
              std::vector<vec3> &normals, std::vector<int> &indices);

// ---- Beyond the reference: a tolerant parser with materials (opt-in; the entry point above keeps
// the reference's exact behaviour, quirks included).
//
// LoadModelEx accepts what Raytracer/objloader.cpp:18-105 trips over: runs of spaces and tabs,
// CRLF line ends, comments, negative (relative) indices, polygons with any number of corners
// (fan-split around the first corner, the way the reference splits quads, objloader.cpp:90-99),
// and it follows `mtllib` / `usemtl` (which the reference ignores, cube.obj:28) so that every
// triangle gets a material.  Faces that name a vertex outside the file are dropped and counted.
struct ObjMaterial {
    char name[64];
    float reflect, diffuse, spec;   // struct Material of Raytracer/main.cpp:29-36, same defaults
    float color[3];
};

struct ObjModel {
    std::vector<vec3> vertices;
    std::vector<vec2> texcoords;
    std::vector<vec3> normals;
    std::vector<int> indices;              // 3 per triangle, 0-based positions
    std::vector<int> normal_indices;       // 3 per triangle, 0-based into normals, -1 where the corner names none
    std::vector<int> texcoord_indices;     // 3 per triangle, 0-based into texcoords, -1 where the corner names none
    std::vector<int> tri_material;         // 1 per triangle, index into materials
    std::vector<ObjMaterial> materials;    // [0] is the reference's default Material()
    int dropped_faces = 0;                 // faces with out-of-range or unparsable corners
    int missing_materials = 0;             // usemtl names that no loaded .mtl defines (fall back to [0])
};

// .mtl mapping (there is no reference behaviour to match; this is the rule):
//   Kd r g b -> color;  Ks r g b -> spec = largest component;  a material whose Kd is absent keeps
//   white, whose Ks is absent keeps spec 1;  diffuse stays 1;  `illum` 3..7 (ray-traced reflection
//   models) -> reflect = largest Ks component, or 1 if Ks is absent;  the non-standard statement
//   `reflect k` sets the coefficient directly and wins.
bool LoadModelEx(const char *objFile, ObjModel &out);

// The `vn` normals the reference parses and drops (objloader.cpp:56-63, main.cpp:222-226) in the layout
// of rt_set_triangle_normals: 9 floats per triangle; a triangle with a corner that names no normal
// gets nine zeros (= keep the geometric normal).  Returns how many triangles carry normals.
int CornerNormals(const ObjModel &m, std::vector<float> &out);
// The same for the `vt` coordinates (objloader.cpp:48-55), 6 floats per triangle, in the layout of
// rt_set_triangle_texcoords; a triangle with a corner that names none gets zeros.
int CornerTexcoords(const ObjModel &m, std::vector<float> &out);
