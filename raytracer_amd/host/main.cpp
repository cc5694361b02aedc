// main.cpp — the application: same shape as the reference's (Raytracer/main.cpp:234-292:
// main -> initglwt -> setup() renders the whole frame -> draw() presents it), with the
// per-pixel loop (main.cpp:262-277) handed to the GPU through the C ABI and the window
// replaced by files.  Defaults reproduce the reference's compile-time constants
// (main.cpp:18: 800 x 600, maxDepth 3) and its stock scene (main.cpp:240-255).
//
//   raytracer [--out frame.ppm] [--raw-order] [--pfm frame.pfm] [--ids ids.bin]
//             [--width W] [--height H] [--depth D] [--obj mesh.obj] [--mesh-reflect K]
//             [--no-mirror] [--no-floor] [--device N] [--dump-obj mesh.obj]
//             [--relaxed] [--exr frame.exr] [--frames N] [--wobble A] [--smooth] [--dump-obj-relaxed mesh.obj]
//             [--gpus N] [--band-rows R]
//
// --gpus N renders every frame on GPUs device .. device+N-1 from this one process: one context per GPU with the
// same scene, the rows dealt out in R-row bands round-robin, the 8-bit slices all-gathered with RCCL (multigpu.cpp);
// 8-bit colour only.  --gpus 1 takes the same path with a communicator of one.
// --relaxed reads the OBJ with the tolerant parser and takes materials from its .mtl files;
// --frames N re-renders from draw(dt) with the camera swinging sideways, N - 1 more times;
// --smooth (with --relaxed) shades with the file's `vn` normals, interpolated (off: the reference's flat normals);
// --wobble A also sways the mesh vertices by up to A units per tick (the hierarchy is refitted).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "glwt.h"
#include "multigpu.h"
#include "objloader.h"
#include "rt_shim.h"
#include "texturerenderer.h"

namespace {

struct Config {
    int imageWidth = 800, imageHeight = 600, maxDepth = 3;   // main.cpp:18
    std::string out = "frame.ppm", pfm, ids, obj, exr;
    bool raw_order = false, mirror = true, floor = true, relaxed = false, smooth = false;
    int frames = 1;
    int builder = 0, leaf_max = 0;                           // hierarchy options (0: the library's defaults)
    float mesh_reflect = 0.0f;                               // main.cpp:34: meshes do not reflect
    float wobble = 0.0f;
    int device = 0;
    int gpus = 0, band_rows = 16;                            // --gpus N: N > 0 renders through multigpu.cpp
} cfg;

color *image = nullptr;                                      // main.cpp:16
std::vector<float> image_rgb32, image_t, image_normal, image_uv;
std::vector<int32_t> image_ids;
rt_ctx *ctx = nullptr;                                       // the context (device cfg.device); with --gpus N the first of N
std::vector<rt_ctx *> contexts;                              // every context: one per GPU
MultiGpuFrame *multi = nullptr;
bool image_pinned = false;                                   // `image` came from rt_alloc_host_buffer
int exit_code = 0;
int frames_rendered = 0;
double total_rays = 0.0, total_device_ms = 0.0, total_refit_ms = 0.0, clock_s = 0.0;
std::vector<float> rest_verts, moved_verts;                  // mesh vertices as loaded / as animated (--wobble)
rt_camera base_cam;
bool render_once(bool report);
bool have_uv = false;

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

// Counterpart of LoadModel(const char*), main.cpp:219-232: parse (once per process), then hand the index
// triples to the GPU of context c, which runs the Triangle constructor for each.
bool LoadModel(rt_ctx *c, const char *model)
{
    static_assert(sizeof(vec3) == 3 * sizeof(float), "vec3 must be packed");
    rt_ctx *const ctx = c;       // every call below goes to this context
    if (cfg.relaxed) {
        static ObjModel m;
        static bool parsed = false;
        if (!parsed && !LoadModelEx(model, m)) {
            fprintf(stderr, "%s: cannot read\n", model);
            exit_code = 1;
            return false;
        }
        if (!parsed && (m.dropped_faces || m.missing_materials))
            fprintf(stderr, "%s: %d face(s) dropped, %d unknown material name(s)\n", model, m.dropped_faces, m.missing_materials);
        const bool first = !parsed;
        parsed = true;
        std::vector<rt_material> mats;
        for (const ObjMaterial &om : m.materials)
            mats.push_back(material(om.reflect, om.diffuse, om.spec, om.color[0], om.color[1], om.color[2]));
        if (cfg.mesh_reflect > 0.0f) mats[0].reflect = cfg.mesh_reflect;      // faces without usemtl
        if (!m.vertices.empty()) rest_verts.assign(&m.vertices[0].x, &m.vertices[0].x + 3 * m.vertices.size());
        if (!check(rt_set_mesh(ctx, m.vertices.empty() ? nullptr : &m.vertices[0].x, (int)m.vertices.size(), m.indices.data(),
                               (int)m.indices.size(), &mats[0]),
                   "rt_set_mesh"))
            return false;
        if (m.tri_material.empty()) return true;
        if (!check(rt_set_triangle_materials(ctx, mats.data(), (int)mats.size(), m.tri_material.data(), (int)m.tri_material.size()),
                   "rt_set_triangle_materials"))
            return false;
        if (!cfg.exr.empty()) {                                  // the EXR's u, v channels
            std::vector<float> vt;
            if (CornerTexcoords(m, vt) > 0) {
                if (!check(rt_set_triangle_texcoords(ctx, vt.data(), (int)m.tri_material.size()), "rt_set_triangle_texcoords")) return false;
                have_uv = true;
            }
        }
        if (cfg.smooth) {
            std::vector<float> vn;
            const int with = CornerNormals(m, vn);
            if (first) printf("%d of %zu triangles carry vn normals\n", with, m.tri_material.size());
            if (with > 0 && !check(rt_set_triangle_normals(ctx, vn.data(), (int)m.tri_material.size()), "rt_set_triangle_normals"))
                return false;
        }
        return true;
    }
    static std::vector<vec3> verts;
    static std::vector<vec2> uvs;
    static std::vector<vec3> normals;
    static std::vector<int> inds;
    static bool parsed = false;
    const int malformed = parsed ? 0 : LoadModel(model, verts, uvs, normals, inds);
    parsed = true;
    if (malformed) {
        fprintf(stderr, "%s: %d malformed face line(s)\n", model, malformed);
        exit_code = 1;
        return false;
    }
    rt_material m = material(cfg.mesh_reflect, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f);
    if (!verts.empty()) rest_verts.assign(&verts[0].x, &verts[0].x + 3 * verts.size());
    return check(rt_set_mesh(ctx, verts.empty() ? nullptr : &verts[0].x, (int)verts.size(), inds.data(), (int)inds.size(), &m),
                 "rt_set_mesh");
}

// One pass of the pixel loop (main.cpp:262-277) on the GPU into `image`.
bool render_once(bool report)
{
    if (multi) {
        uint64_t rays3[3];
        float ms = 0.f;
        if (!multigpu_render(multi, cfg.maxDepth, (uint8_t *)image, rays3, &ms)) {
            exit_code = 1;
            return false;
        }
        const double rays = (double)(rays3[0] + rays3[1] + rays3[2]);
        frames_rendered++;
        total_rays += rays;
        total_device_ms += ms;
        if (report)
            printf("Render took %f seconds (longest device of %d), %.0f rays, %.1f Mrays/s\n", ms * 1e-3, (int)contexts.size(), rays,
                   ms > 0 ? rays / (ms * 1e3) : 0.0);
        return true;
    }
    if (!image_t.empty()) {                        // EXR wanted: colour plus the t and normal planes
        rt_frame_planes planes = { (uint8_t *)image, image_ids.data(), image_rgb32.data(), image_t.data(), image_normal.data(),
                                   image_uv.empty() ? nullptr : image_uv.data() };
        if (!check(rt_render_frame_planes(ctx, cfg.imageWidth, cfg.imageHeight, 0, cfg.imageHeight, cfg.maxDepth, &planes),
                   "rt_render_frame_planes"))
            return false;
    } else if (!check(rt_render_frame(ctx, cfg.imageWidth, cfg.imageHeight, 0, cfg.imageHeight, cfg.maxDepth, (uint8_t *)image,
                                      image_ids.empty() ? nullptr : image_ids.data(),
                                      image_rgb32.empty() ? nullptr : image_rgb32.data()),
                      "rt_render_frame"))
        return false;
    rt_stats st;
    if (!check(rt_get_stats(ctx, &st), "rt_get_stats")) return false;
    const double rays = (double)(st.rays_primary + st.rays_shadow + st.rays_reflect);
    frames_rendered++;
    total_rays += rays;
    total_device_ms += st.ms_frame;
    if (report)
        printf("Render took %f seconds (device), %.0f rays, %.1f Mrays/s\n", st.ms_frame * 1e-3, rays,   // main.cpp:279
               st.ms_frame > 0 ? rays / (st.ms_frame * 1e3) : 0.0);
    return true;
}

}  // namespace

int glwt_headless_frames = 1;

void setup()
{
    texturerenderer_setup();

    if (rt_abi_version() != RT_ABI_VERSION) {
        fprintf(stderr, "librt_amd.so speaks ABI version %d, this program %d\n", rt_abi_version(), RT_ABI_VERSION);
        exit_code = 1;
        return;
    }
    // the stock scene, main.cpp:240-255
    std::vector<rt_analytic> scene;
    if (cfg.mirror) scene.push_back(sphere(0.0f, 0.0f, 0.0f, 2.5f, material(1.0f, 0.0f, 1.0f, 1.0f, 1.0f, 1.0f), false));
    scene.push_back(sphere(2.0f, 5.0f, 1.0f, 0.1f, material(0.0f, 1.0f, 1.0f, 0.7f, 0.7f, 0.9f), true));
    scene.push_back(sphere(-2.0f, 5.0f, -3.0f, 0.1f, material(0.0f, 1.0f, 1.0f, 0.9f, 0.9f, 0.4f), true));
    if (cfg.floor) scene.push_back(plane(0.0f, 1.0f, 0.0f, -4.0f, material(0.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f)));
    // camera of main.cpp:267-270; for another width the 8-unit horizontal span is kept
    rt_camera cam = { { 0.0f, 0.0f, -5.0f }, 0.01f, -4.0f, -4.0f, 0.0f };
    if (cfg.imageWidth != 800) cam.pitch = 8.0f / (float)cfg.imageWidth;
    base_cam = cam;

    // one context per GPU, each with the whole scene (the hierarchy is built on every GPU: deterministic, so identical)
    std::vector<int> devices;
    for (int k = 0; k < (cfg.gpus > 0 ? cfg.gpus : 1); k++) {
        rt_ctx *c = rt_create(cfg.device + k);
        if (!c) {
            fprintf(stderr, "rt_create(%d) failed: %s\n", cfg.device + k, rt_last_error());
            exit_code = 1;
            return;
        }
        contexts.push_back(c);
        devices.push_back(cfg.device + k);
        if (k == 0) ctx = c;
        if (cfg.builder && !check(rt_set_option(c, RT_OPT_BUILDER, cfg.builder), "rt_set_option(builder)")) return;
        if (cfg.leaf_max && !check(rt_set_option(c, RT_OPT_LEAF_MAX, cfg.leaf_max), "rt_set_option(leaf max)")) return;
        if (!check(rt_set_analytic(c, scene.data(), (int)scene.size()), "rt_set_analytic")) return;
        if (!cfg.obj.empty() && !LoadModel(c, cfg.obj.c_str())) return;        // main.cpp:257
        if (!check(rt_set_camera(c, &cam), "rt_set_camera")) return;
    }

    const size_t n = (size_t)cfg.imageWidth * cfg.imageHeight;
    // main.cpp:259 allocates the image with new; here it comes pinned from the library, so the device writes it directly
    image = (color *)rt_alloc_host_buffer(ctx, sizeof(color) * n);
    image_pinned = image != nullptr;
    if (!image) image = new color[n];
    if (cfg.gpus > 0) {
        if (!cfg.pfm.empty() || !cfg.ids.empty() || !cfg.exr.empty()) {
            fprintf(stderr, "--gpus renders the 8-bit frame only (no --pfm, --ids, --exr)\n");
            exit_code = 2;
            return;
        }
        multi = multigpu_create(contexts, devices, cfg.imageWidth, cfg.imageHeight, cfg.band_rows);
        if (!multi) {
            exit_code = 1;
            return;
        }
    }
    if (!cfg.pfm.empty() || !cfg.exr.empty()) image_rgb32.resize(3 * n);
    if (!cfg.ids.empty() || !cfg.exr.empty()) image_ids.resize(n);
    if (!cfg.exr.empty()) { image_t.resize(n); image_normal.resize(3 * n); if (have_uv) image_uv.resize(2 * n); }

    printf("Rendering...\n");                                                   // main.cpp:261
    if (!render_once(true)) return;

    texturerenderer_set_output(cfg.out.empty() ? nullptr : cfg.out.c_str(), cfg.raw_order);
    if (!cfg.pfm.empty()) texturerenderer_set_float_output(cfg.pfm.c_str(), image_rgb32.data());
    if (!cfg.exr.empty()) texturerenderer_set_exr_output(cfg.exr.c_str(), image_rgb32.data(), image_ids.data(), image_t.data(), image_normal.data(),
                                                         image_uv.empty() ? nullptr : image_uv.data());
    texturerenderer_displaytexture(image, cfg.imageWidth, cfg.imageHeight);    // main.cpp:281
    if (!cfg.ids.empty()) {
        if (FILE *f = fopen(cfg.ids.c_str(), "wb")) {
            fwrite(image_ids.data(), sizeof(int32_t), n, f);
            fclose(f);
        }
    }
}

// The reference's draw(dt) only re-presents the one frame of setup() (main.cpp:284-287).  With
// --frames N every further tick re-renders: the scene and its hierarchy stay on the device, only
// the camera moves (a sideways swing of the eye and of the image plane), which turns the one-shot
// render into a sustained-throughput run.
void draw(float dt)
{
    if (cfg.frames > 1 && ctx && image && exit_code == 0 && frames_rendered < cfg.frames) {
        clock_s += dt;
        rt_camera cam = base_cam;
        const float swing = 1.5f * sinf((float)clock_s * 2.0f);
        cam.origin[0] += swing;
        cam.off_x += swing;
        bool ok = true;
        for (rt_ctx *c : contexts) ok = ok && check(rt_set_camera(c, &cam), "rt_set_camera");
        if (ok && cfg.wobble != 0.0f && !rest_verts.empty()) {
            // a travelling wave through the mesh: same triangles, new positions, tree refitted on the device
            moved_verts.resize(rest_verts.size());
            for (size_t k = 0; k < rest_verts.size(); k += 3) {
                const float x = rest_verts[k], y = rest_verts[k + 1], z = rest_verts[k + 2];
                const float ph = (float)clock_s * 3.0f;
                moved_verts[k] = x + cfg.wobble * sinf(ph + y * 0.7f);
                moved_verts[k + 1] = y + cfg.wobble * 0.5f * sinf(ph * 1.3f + z * 0.5f);
                moved_verts[k + 2] = z + cfg.wobble * cosf(ph + x * 0.6f);
            }
            for (rt_ctx *c : contexts)
                ok = ok && check(rt_update_mesh_vertices(c, moved_verts.data(), (int)(moved_verts.size() / 3)), "rt_update_mesh_vertices");
            rt_stats st;
            if (ok && rt_get_stats(ctx, &st) == RT_OK) total_refit_ms += st.ms_build;
        }
        if (ok) render_once(false);
        if (frames_rendered == cfg.frames && total_device_ms > 0.0) {
            printf("%d frames, %.3f ms/frame (device), %.1f Mrays/s sustained\n", frames_rendered, total_device_ms / frames_rendered,
                   total_rays / (total_device_ms * 1e3));
            if (cfg.wobble != 0.0f && !rest_verts.empty())
                printf("refit %.3f ms/frame (device)\n", total_refit_ms / (frames_rendered - 1));
        }
        texturerenderer_displaytexture(image, cfg.imageWidth, cfg.imageHeight);
        if (frames_rendered < cfg.frames) return;          // files are written once, with the last frame
    }
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

static int dump_obj_relaxed(const char *path)
{
    ObjModel m;
    if (!LoadModelEx(path, m)) { printf("unreadable\n"); return 1; }
    printf("verts %zu texcoords %zu normals %zu indices %zu materials %zu dropped %d missing %d\n", m.vertices.size(),
           m.texcoords.size(), m.normals.size(), m.indices.size(), m.materials.size(), m.dropped_faces, m.missing_materials);
    for (const vec3 &p : m.vertices) printf("v %.9g %.9g %.9g\n", p.x, p.y, p.z);
    for (size_t i = 0; i < m.indices.size(); i++) printf("i %d\n", m.indices[i]);
    for (size_t i = 0; i < m.tri_material.size(); i++) printf("t %d\n", m.tri_material[i]);
    for (size_t i = 0; i < m.normal_indices.size(); i++) printf("n %d %d\n", m.normal_indices[i], m.texcoord_indices[i]);
    for (const ObjMaterial &om : m.materials)
        printf("m %s|%.9g %.9g %.9g %.9g %.9g %.9g\n", om.name, om.reflect, om.diffuse, om.spec, om.color[0], om.color[1], om.color[2]);
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
        else if (a == "--dump-obj-relaxed") return dump_obj_relaxed(val());
        else if (a == "--relaxed") cfg.relaxed = true;
        else if (a == "--smooth") cfg.smooth = true;
        else if (a == "--exr") cfg.exr = val();
        else if (a == "--frames") cfg.frames = atoi(val());
        else if (a == "--wobble") cfg.wobble = (float)atof(val());
        else if (a == "--builder") cfg.builder = atoi(val());
        else if (a == "--leaf-max") cfg.leaf_max = atoi(val());
        else if (a == "--gpus") cfg.gpus = atoi(val());
        else if (a == "--band-rows") cfg.band_rows = atoi(val());
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    glwt_headless_frames = cfg.frames > 1 ? cfg.frames : 1;
    if (cfg.gpus < 0 || cfg.gpus > 64 || cfg.band_rows < 1) { fprintf(stderr, "--gpus takes 1..64, --band-rows a positive count\n"); return 2; }
    int rc = initglwt("Raytracer", cfg.imageWidth, cfg.imageHeight, false);    // main.cpp:291
    multigpu_destroy(multi);
    if (image_pinned) (void)rt_free_host_buffer(ctx, image);
    else delete[] image;
    for (rt_ctx *c : contexts) rt_destroy(c);
    return rc ? rc : exit_code;
}
