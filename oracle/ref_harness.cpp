// ref_harness.cpp — drives the UNMODIFIED reference tracer for oracle validation.
// TEST INFRASTRUCTURE ONLY: never linked into, or called by, the product path.
//
// The reference's trace path lives in one translation unit with no header
// (Raytracer/main.cpp: types 20-135, raytrace 144-217, LoadModel 219-232, setup
// 234-282).  This file textually includes that translation unit from where it
// lies (-I/root/reference/Raytracer; nothing is copied) so that tests can call
// the reference's own raytrace(), Raycast() methods and LoadModel() on arbitrary
// scenes, and supplies the four application-boundary functions that the macOS
// window / OpenGL layer (glwt.mm, texturerenderer.cpp — both out of scope and
// not buildable on Linux) would otherwise define: a headless frame sink and a
// window host that just runs setup().  No GL, Cocoa or OpenCL header is faked.
//
// Bounce depth: main.cpp:18 fixes maxDepth = 3 and main.cpp:208 tests
// `depth < maxDepth`, where depth only ever appears in that test and as
// depth+1.  Starting a primary ray at depth (3 - D) therefore yields exactly D
// reflection segments without touching the source.
//
// Built by oracle/build_ref.sh into oracle/_ref/ (git-ignored).
#include <time.h>      // main.cpp:238 uses clock() without including <time.h>
#include <stdint.h>
#include <string.h>

#define main reference_main
#include "main.cpp"
#undef main

static color *g_sunk_pixels = nullptr;
static int g_sunk_w = 0, g_sunk_h = 0;

void texturerenderer_setup() {}
void texturerenderer_draw() {}
void texturerenderer_displaytexture(color *pixels, int w, int h)
{
    g_sunk_pixels = pixels; g_sunk_w = w; g_sunk_h = h;
}
extern "C" int initglwt(const char *, int, int, bool)
{
    setup();
    return 0;
}

static void set_material(Primitive *p, const float m[6])
{
    p->material.reflect = m[0]; p->material.diffuse = m[1]; p->material.spec = m[2];
    p->material.color = vec3(m[3], m[4], m[5]);
}

extern "C" {

// Runs the reference's own main() -> setup(): stock scene, 800x600, depth 3.
// Copies the frame exactly as handed to texturerenderer_displaytexture.
int ref_stock_frame(uint8_t *rgb8, int cap_bytes)
{
    scene.clear();
    g_sunk_pixels = nullptr;
    reference_main(0, nullptr);
    scene.clear();
    if (!g_sunk_pixels || cap_bytes < g_sunk_w * g_sunk_h * 3) return -1;
    memcpy(rgb8, g_sunk_pixels, (size_t)g_sunk_w * g_sunk_h * 3);
    delete[] g_sunk_pixels;
    g_sunk_pixels = nullptr;
    return g_sunk_w * 65536 + g_sunk_h;
}

void ref_scene_clear() { scene.clear(); }
int ref_scene_size() { return (int)scene.size(); }

// material = {reflect, diffuse, spec, r, g, b}
int ref_add_sphere(const float pos[3], float radius, const float material[6], int is_light)
{
    Sphere *s = new Sphere(vec3(pos[0], pos[1], pos[2]), radius);
    set_material(s, material);
    s->isLight = is_light != 0;
    scene.push_back(s);
    return (int)scene.size() - 1;
}

int ref_add_plane(const float n[3], float offset, const float material[6])
{
    Plane *p = new Plane(vec3(n[0], n[1], n[2]), offset);
    set_material(p, material);
    scene.push_back(p);
    return (int)scene.size() - 1;
}

int ref_add_triangle(const float v1[3], const float v2[3], const float v3_[3], const float material[6])
{
    Triangle *t = new Triangle(vec3(v1[0], v1[1], v1[2]), vec3(v2[0], v2[1], v2[2]),
                               vec3(v3_[0], v3_[1], v3_[2]));
    set_material(t, material);
    scene.push_back(t);
    return (int)scene.size() - 1;
}

// The reference's own flatten (main.cpp:219-232) on the reference's own parser;
// `material` (may be NULL = keep defaults) is then applied to the new triangles.
int ref_load_model(const char *path, const float *material)
{
    size_t before = scene.size();
    LoadModel(path);
    if (material)
        for (size_t i = before; i < scene.size(); i++) set_material(scene[i], material);
    return (int)(scene.size() - before);
}

// Raw parser output (objloader.cpp:18-105).
int ref_parse_obj(const char *path, float *verts, int verts_cap, int *inds, int inds_cap,
                  int *n_verts, int *n_inds)
{
    std::vector<vec3> v; std::vector<vec2> uv; std::vector<vec3> n; std::vector<int> idx;
    LoadModel(path, v, uv, n, idx);
    *n_verts = (int)v.size(); *n_inds = (int)idx.size();
    if ((int)v.size() * 3 > verts_cap || (int)idx.size() > inds_cap) return -1;
    for (size_t i = 0; i < v.size(); i++) { verts[3*i] = v[i].x; verts[3*i+1] = v[i].y; verts[3*i+2] = v[i].z; }
    for (size_t i = 0; i < idx.size(); i++) inds[i] = idx[i];
    return 0;
}

// Triangle data as the reference's constructor computed it (main.cpp:104-107).
int ref_get_triangle(int id, float out12[12])
{
    Triangle *t = dynamic_cast<Triangle *>(scene[id]);
    if (!t) return -1;
    const vec3 *src[4] = { &t->v1, &t->e1, &t->e2, &t->N };
    for (int k = 0; k < 4; k++) { out12[3*k] = src[k]->x; out12[3*k+1] = src[k]->y; out12[3*k+2] = src[k]->z; }
    return 0;
}

// One call of the reference's raytrace() with D reflection segments allowed.
void ref_raytrace(const float o[3], const float d[3], int bounces, float rgb[3])
{
    Ray r(vec3(o[0], o[1], o[2]), vec3(d[0], d[1], d[2]));
    vec3 c = raytrace(r, maxDepth - bounces);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

// Per-pixel loop of main.cpp:262-273 with the three camera literals as parameters.
// Outputs may be NULL; rgb8 uses truncation (the Debug-build meaning of main.cpp:273).
int ref_render(const float origin[3], float pitch, float off_x, float off_y, float plane_z,
               int W, int H, int y0, int y1, int bounces, int x_step, int y_step,
               uint8_t *rgb8, float *rgb32, int32_t *hit_id, float *hit_t)
{
    for (int y = y0; y < y1; y++) {
        if (y % y_step) continue;
        for (int x = 0; x < W; x += x_step) {
            vec3 o(origin[0], origin[1], origin[2]);
            float offsetX = x * pitch + off_x;
            float offsetY = y * pitch + off_y;
            Ray r(o, (vec3(offsetX, offsetY, plane_z) - o).normalize());
            vec3 col = raytrace(r, maxDepth - bounces);
            size_t idx = (size_t)(y - y0) * W + x;
            if (rgb8) {
                float c[3] = { clamp01(col.x) * 255.0f, clamp01(col.y) * 255.0f, clamp01(col.z) * 255.0f };
                for (int k = 0; k < 3; k++) rgb8[3*idx+k] = (c[k] == c[k]) ? (uint8_t)(int)c[k] : 0;
            }
            if (rgb32) { rgb32[3*idx] = col.x; rgb32[3*idx+1] = col.y; rgb32[3*idx+2] = col.z; }
            if (hit_id || hit_t) {
                // the nearest-hit scan of main.cpp:147-156, through the reference's own Raycast()
                float nearest = FLT_MAX, t; int id = -1;
                for (size_t i = 0; i < scene.size(); i++)
                    if (scene[i]->Raycast(r, t) && nearest > t) { nearest = t; id = (int)i; }
                if (hit_id) hit_id[idx] = id;
                if (hit_t) hit_t[idx] = nearest;
            }
        }
    }
    return 0;
}

}  // extern "C"
