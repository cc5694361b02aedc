// texturerenderer.cpp — see texturerenderer.h.
#include "texturerenderer.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
std::string g_ppm, g_pfm, g_exr;
const float *g_exr_rgb = nullptr, *g_exr_t = nullptr, *g_exr_n = nullptr, *g_exr_uv = nullptr;
const int *g_exr_ids = nullptr;
bool g_raw = false;
const color *g_pixels = nullptr;
const float *g_rgb32 = nullptr;
int g_w = 0, g_h = 0;
bool g_written = false;
}  // namespace

void texturerenderer_setup() {}

void texturerenderer_set_output(const char *ppm_path, bool raw_order)
{
    g_ppm = ppm_path ? ppm_path : "";
    g_raw = raw_order;
}

void texturerenderer_set_float_output(const char *pfm_path, const float *rgb32)
{
    g_pfm = pfm_path ? pfm_path : "";
    g_rgb32 = rgb32;
}

void texturerenderer_set_exr_output(const char *exr_path, const float *rgb32, const int *ids, const float *hit_t, const float *normal,
                                    const float *uv)
{
    g_exr_uv = uv;
    g_exr = exr_path ? exr_path : "";
    g_exr_rgb = rgb32;
    g_exr_ids = ids;
    g_exr_t = hit_t;
    g_exr_n = normal;
}

namespace {

// ---- minimal OpenEXR 2 writer: single part, scanline, no compression, one scan line per chunk
void put(std::vector<unsigned char> &b, const void *p, size_t n) { b.insert(b.end(), (const unsigned char *)p, (const unsigned char *)p + n); }
void put_str(std::vector<unsigned char> &b, const char *s) { put(b, s, strlen(s) + 1); }
void put_i32(std::vector<unsigned char> &b, int32_t v) { put(b, &v, 4); }
void attr(std::vector<unsigned char> &b, const char *name, const char *type, const void *val, int32_t n)
{
    put_str(b, name); put_str(b, type); put_i32(b, n); put(b, val, (size_t)n);
}

bool write_exr(const char *path, int w, int h, const float *rgb32, const int *ids, const float *hit_t, const float *normal,
               const float *uv)
{
    // channels in alphabetical order (upper case sorts first): B G [N.x N.y N.z] R [Z] [id] [u v]
    struct Chan { const char *name; int type; const float *src; int stride, comp; const int *isrc; };
    std::vector<Chan> chans;
    chans.push_back({ "B", 2, rgb32, 3, 2, nullptr });
    chans.push_back({ "G", 2, rgb32, 3, 1, nullptr });
    if (normal) {
        chans.push_back({ "N.x", 2, normal, 3, 0, nullptr });
        chans.push_back({ "N.y", 2, normal, 3, 1, nullptr });
        chans.push_back({ "N.z", 2, normal, 3, 2, nullptr });
    }
    chans.push_back({ "R", 2, rgb32, 3, 0, nullptr });
    if (hit_t) chans.push_back({ "Z", 2, hit_t, 1, 0, nullptr });
    if (ids) chans.push_back({ "id", 0, nullptr, 1, 0, ids });
    if (uv) {
        chans.push_back({ "u", 2, uv, 2, 0, nullptr });
        chans.push_back({ "v", 2, uv, 2, 1, nullptr });
    }

    std::vector<unsigned char> hd;
    const unsigned char magic[8] = { 0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0 };
    put(hd, magic, 8);
    std::vector<unsigned char> ch;
    for (const Chan &c : chans) {
        put_str(ch, c.name);
        put_i32(ch, c.type);                        // 2 FLOAT, 0 UINT
        const unsigned char lin[4] = { 0, 0, 0, 0 };
        put(ch, lin, 4);
        put_i32(ch, 1); put_i32(ch, 1);             // sampling
    }
    ch.push_back(0);
    attr(hd, "channels", "chlist", ch.data(), (int32_t)ch.size());
    const unsigned char none = 0;
    attr(hd, "compression", "compression", &none, 1);
    const int32_t win[4] = { 0, 0, w - 1, h - 1 };
    attr(hd, "dataWindow", "box2i", win, 16);
    attr(hd, "displayWindow", "box2i", win, 16);
    attr(hd, "lineOrder", "lineOrder", &none, 1);   // increasing y
    const float one = 1.0f, zero2[2] = { 0.0f, 0.0f };
    attr(hd, "pixelAspectRatio", "float", &one, 4);
    attr(hd, "screenWindowCenter", "v2f", zero2, 8);
    attr(hd, "screenWindowWidth", "float", &one, 4);
    hd.push_back(0);

    FILE *f = fopen(path, "wb");
    if (!f) return false;
    const size_t line_bytes = (size_t)w * 4 * chans.size();
    fwrite(hd.data(), 1, hd.size(), f);
    uint64_t off = hd.size() + 8ull * (uint64_t)h;
    for (int y = 0; y < h; y++, off += 8 + line_bytes) fwrite(&off, 8, 1, f);
    std::vector<float> row((size_t)w);
    std::vector<uint32_t> irow((size_t)w);
    for (int y = 0; y < h; y++) {
        const size_t src = (size_t)(h - 1 - y) * (size_t)w;     // buffer row 0 is the bottom of the picture
        const int32_t hdr[2] = { y, (int32_t)line_bytes };
        fwrite(hdr, 4, 2, f);
        for (const Chan &c : chans) {
            if (c.type == 2) {
                for (int x = 0; x < w; x++) row[(size_t)x] = c.src[(size_t)c.stride * (src + (size_t)x) + (size_t)c.comp];
                fwrite(row.data(), 4, (size_t)w, f);
            } else {
                for (int x = 0; x < w; x++) irow[(size_t)x] = (uint32_t)(c.isrc[src + (size_t)x] + 1);
                fwrite(irow.data(), 4, (size_t)w, f);
            }
        }
    }
    fclose(f);
    return true;
}

}  // namespace

void texturerenderer_displaytexture(color *pixels, int w, int h)
{
    g_pixels = pixels; g_w = w; g_h = h;
    g_written = false;
}

void texturerenderer_draw()
{
    if (g_written || !g_pixels) return;
    g_written = true;
    if (!g_ppm.empty()) {
        if (FILE *f = fopen(g_ppm.c_str(), "wb")) {
            fprintf(f, "P6\n%d %d\n255\n", g_w, g_h);
            for (int r = 0; r < g_h; r++) {
                int y = g_raw ? r : g_h - 1 - r;
                fwrite(g_pixels + (size_t)y * g_w, 3, (size_t)g_w, f);
            }
            fclose(f);
        } else {
            fprintf(stderr, "cannot write %s\n", g_ppm.c_str());
        }
    }
    if (!g_pfm.empty() && g_rgb32) {
        // PFM stores rows bottom-to-top, which is exactly the buffer order; negative scale = little endian
        if (FILE *f = fopen(g_pfm.c_str(), "wb")) {
            fprintf(f, "PF\n%d %d\n-1.0\n", g_w, g_h);
            fwrite(g_rgb32, sizeof(float) * 3, (size_t)g_w * g_h, f);
            fclose(f);
        } else {
            fprintf(stderr, "cannot write %s\n", g_pfm.c_str());
        }
    }
    if (!g_exr.empty() && g_exr_rgb) {
        if (!write_exr(g_exr.c_str(), g_w, g_h, g_exr_rgb, g_exr_ids, g_exr_t, g_exr_n, g_exr_uv)) fprintf(stderr, "cannot write %s\n", g_exr.c_str());
    }
}
