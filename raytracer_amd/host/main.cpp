// main.cpp — the application: same shape as the reference's (Raytracer/main.cpp:234-292:
// main -> initglwt -> setup() renders the whole frame -> draw() presents it), with the
// per-pixel loop (main.cpp:262-277) handed to the GPU through the C ABI and the window
// replaced by files.  Defaults reproduce the reference's compile-time constants
// (main.cpp:18: 800 x 600, maxDepth 3) and its stock scene (main.cpp:240-255).
//
//   raytracer [--out frame.ppm] [--raw-order] [--pfm frame.pfm] [--ids ids.bin]
//             [--width W] [--height H] [--depth D] [--obj mesh.obj] [--mesh-reflect K]
//             [--no-mirror] [--no-floor] [--device N] [--dump-obj mesh.obj]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "glwt.h"
#include "objloader.h"
#include "rt_shim.h"
#include "texturerenderer.h"

namespace {

struct Config {
    int imageWidth = 800, imageHeight = 600, maxDepth = 3;   // main.cpp:18
    std::string out = "frame.ppm", pfm, ids, obj;
    bool raw_order = false, mirror = true, floor = true;
    float mesh_reflect = 0.0f;                               // main.cpp:34: meshes do not reflect
    int device = 0;
} cfg;

color *image = nullptr;                                      // main.cpp:16
std::vector<float> image_rgb32;
std::vector<int32_t> image_ids;
rt_ctx *ctx = nullptr;
int exit_code = 0;

rt_material material(float reflect, float diffuse, float spec, float r, float g, float b)
{
    rt_material m = { reflect, diffuse, spec, { r, g, b } };
    return m;
}

rt_analytic sphere(float x, float y, float z, float radius, rt_material m, bool light)
{
    rt_analytic a;
    a.kind = RT_SPHERE; a.is_light = light; a.mat = m;
    a.p[0] = x; a.p[1] = y; a.p[2] = z; a.s = radius;
    return a;
}

rt_analytic plane(float nx, float ny, float nz, float offset, rt_material m)
{
    rt_analytic a;
    a.kind = RT_PLANE; a.is_light = 0; a.mat = m;
    a.p[0] = nx; a.p[1] = ny; a.p[2] = nz; a.s = offset;
    return a;
}

bool check(int rc, const char *what)
{
    if (rc == RT_OK) return true;
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, rt_last_error());
    exit_code = 1;
    return false;
}

// Counterpart of LoadModel(const char*), main.cpp:219-232: parse, then hand the index
// triples to the GPU, which runs the Triangle constructor for each.
bool LoadModel(const char *model)
{
    std::vector<vec3> verts;
    std::vector<vec2> uvs;
    std::vector<vec3> normals;
    std::vector<int> inds;
    int malformed = LoadModel(model, verts, uvs, normals, inds);
    if (malformed) {
        fprintf(stderr, "%s: %d malformed face line(s)\n", model, malformed);
        exit_code = 1;
        return false;
    }
    rt_material m = material(cfg.mesh_reflect, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f);
    static_assert(sizeof(vec3) == 3 * sizeof(float), "vec3 must be packed");
    return check(rt_set_mesh(ctx, verts.empty() ? nullptr : &verts[0].x, (int)verts.size(), inds.data(), (int)inds.size(), &m),
                 "rt_set_mesh");
}

}  // namespace

void setup()
{
    texturerenderer_setup();

    ctx = rt_create(cfg.device);
    if (!ctx) {
        fprintf(stderr, "rt_create failed: %s\n", rt_last_error());
        exit_code = 1;
        return;
    }

    // the stock scene, main.cpp:240-255
    std::vector<rt_analytic> scene;
    if (cfg.mirror) scene.push_back(sphere(0.0f, 0.0f, 0.0f, 2.5f, material(1.0f, 0.0f, 1.0f, 1.0f, 1.0f, 1.0f), false));
    scene.push_back(sphere(2.0f, 5.0f, 1.0f, 0.1f, material(0.0f, 1.0f, 1.0f, 0.7f, 0.7f, 0.9f), true));
    scene.push_back(sphere(-2.0f, 5.0f, -3.0f, 0.1f, material(0.0f, 1.0f, 1.0f, 0.9f, 0.9f, 0.4f), true));
    if (cfg.floor) scene.push_back(plane(0.0f, 1.0f, 0.0f, -4.0f, material(0.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f)));
    if (!check(rt_set_analytic(ctx, scene.data(), (int)scene.size()), "rt_set_analytic")) return;

    if (!cfg.obj.empty() && !LoadModel(cfg.obj.c_str())) return;               // main.cpp:257

    // camera of main.cpp:267-270; for another width the 8-unit horizontal span is kept
    rt_camera cam = { { 0.0f, 0.0f, -5.0f }, 0.01f, -4.0f, -4.0f, 0.0f };
    if (cfg.imageWidth != 800) cam.pitch = 8.0f / (float)cfg.imageWidth;
    if (!check(rt_set_camera(ctx, &cam), "rt_set_camera")) return;

    const size_t n = (size_t)cfg.imageWidth * cfg.imageHeight;
    image = new color[n];                                                       // main.cpp:259
    if (!cfg.pfm.empty()) image_rgb32.resize(3 * n);
    if (!cfg.ids.empty()) image_ids.resize(n);

    printf("Rendering...\n");                                                   // main.cpp:261
    if (!check(rt_render_frame(ctx, cfg.imageWidth, cfg.imageHeight, 0, cfg.imageHeight, cfg.maxDepth, (uint8_t *)image,
                               image_ids.empty() ? nullptr : image_ids.data(),
                               image_rgb32.empty() ? nullptr : image_rgb32.data()),
               "rt_render_frame"))
        return;
    rt_stats st;
    if (check(rt_get_stats(ctx, &st), "rt_get_stats")) {
        double rays = (double)(st.rays_primary + st.rays_shadow + st.rays_reflect);
        printf("Render took %f seconds (device), %.0f rays, %.1f Mrays/s\n", st.ms_frame * 1e-3, rays,   // main.cpp:279
               st.ms_frame > 0 ? rays / (st.ms_frame * 1e3) : 0.0);
    }

    texturerenderer_set_output(cfg.out.empty() ? nullptr : cfg.out.c_str(), cfg.raw_order);
    if (!cfg.pfm.empty()) texturerenderer_set_float_output(cfg.pfm.c_str(), image_rgb32.data());
    texturerenderer_displaytexture(image, cfg.imageWidth, cfg.imageHeight);    // main.cpp:281
    if (!cfg.ids.empty()) {
        if (FILE *f = fopen(cfg.ids.c_str(), "wb")) {
            fwrite(image_ids.data(), sizeof(int32_t), n, f);
            fclose(f);
        }
    }
}

void draw(float)
{
    texturerenderer_draw();                                                     // main.cpp:284-287
}

static int dump_obj(const char *path)
{
    std::vector<vec3> v, n;
    std::vector<vec2> t;
    std::vector<int> idx;
    int malformed = LoadModel(path, v, t, n, idx);
    printf("verts %zu texcoords %zu normals %zu indices %zu malformed %d\n", v.size(), t.size(), n.size(), idx.size(), malformed);
    for (const vec3 &p : v) printf("v %.9g %.9g %.9g\n", p.x, p.y, p.z);
    for (size_t i = 0; i < idx.size(); i++) printf("i %d\n", idx[i]);
    return 0;
}

int main(int argc, char *argv[])
{
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a.c_str()); exit(2); }
            return argv[++i];
        };
        if (a == "--out") cfg.out = val();
        else if (a == "--pfm") cfg.pfm = val();
        else if (a == "--ids") cfg.ids = val();
        else if (a == "--obj") cfg.obj = val();
        else if (a == "--width") cfg.imageWidth = atoi(val());
        else if (a == "--height") cfg.imageHeight = atoi(val());
        else if (a == "--depth") cfg.maxDepth = atoi(val());
        else if (a == "--mesh-reflect") cfg.mesh_reflect = (float)atof(val());
        else if (a == "--device") cfg.device = atoi(val());
        else if (a == "--raw-order") cfg.raw_order = true;
        else if (a == "--no-mirror") cfg.mirror = false;
        else if (a == "--no-floor") cfg.floor = false;
        else if (a == "--dump-obj") return dump_obj(val());
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    int rc = initglwt("Raytracer", cfg.imageWidth, cfg.imageHeight, false);    // main.cpp:291
    if (ctx) rt_destroy(ctx);
    delete[] image;
    return rc ? rc : exit_code;
}
