// texturerenderer.cpp — see texturerenderer.h.
#include "texturerenderer.h"

#include <cstdio>
#include <string>

namespace {
std::string g_ppm, g_pfm;
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
}
