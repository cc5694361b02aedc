// texturerenderer.cpp — see texturerenderer.h.
#include "texturerenderer.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
std::string g_ppm, g_pfm, g_exr;
const float *g_exr_rgb = nullptr;
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

void texturerenderer_set_exr_output(const char *exr_path, const float *rgb32, const int *ids)
{
    g_exr = exr_path ? exr_path : "";
    g_exr_rgb = rgb32;
    g_exr_ids = ids;
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

bool write_exr(const char *path, int w, int h, const float *rgb32, const int *ids)
{
    std::vector<unsigned char> hd;
    const unsigned char magic[8] = { 0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0 };
    put(hd, magic, 8);
    // channels, alphabetical: B G R then id (upper case sorts first)
    std::vector<unsigned char> ch;
    const char *names[4] = { "B", "G", "R", "id" };
    const int n_ch = ids ? 4 : 3;
    for (int c = 0; c < n_ch; c++) {
        put_str(ch, names[c]);
        put_i32(ch, c < 3 ? 2 : 0);                 // FLOAT / UINT
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
    const size_t line_bytes = (size_t)w * 4 * (size_t)n_ch;
    fwrite(hd.data(), 1, hd.size(), f);
    uint64_t off = hd.size() + 8ull * (uint64_t)h;
    for (int y = 0; y < h; y++, off += 8 + line_bytes) fwrite(&off, 8, 1, f);
    std::vector<float> row((size_t)w);
    std::vector<uint32_t> irow((size_t)w);
    for (int y = 0; y < h; y++) {
        const int src = h - 1 - y;                  // buffer row 0 is the bottom of the picture
        const int32_t hdr[2] = { y, (int32_t)line_bytes };
        fwrite(hdr, 4, 2, f);
        for (int c = 0; c < 3; c++) {               // B, G, R
            const int comp = 2 - c;
            for (int x = 0; x < w; x++) row[(size_t)x] = rgb32[3 * ((size_t)src * w + x) + comp];
            fwrite(row.data(), 4, (size_t)w, f);
        }
        if (ids) {
            for (int x = 0; x < w; x++) irow[(size_t)x] = (uint32_t)(ids[(size_t)src * w + x] + 1);
            fwrite(irow.data(), 4, (size_t)w, f);
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
        if (!write_exr(g_exr.c_str(), g_w, g_h, g_exr_rgb, g_exr_ids)) fprintf(stderr, "cannot write %s\n", g_exr.c_str());
    }
}
