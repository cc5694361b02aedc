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
