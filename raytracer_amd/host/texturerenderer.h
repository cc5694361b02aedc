// texturerenderer.h — frame sink with the reference's three entry points
// (Raytracer/texturerenderer.h:14-20).  Instead of uploading the frame to a GL texture
// (texturerenderer.cpp:90-93) it keeps it and, on draw, writes it to the configured files.
#pragma once

struct color {
    char r, g, b;   // 3-byte packed pixel, row-major, row 0 = bottom of the displayed image
};

void texturerenderer_setup();
void texturerenderer_displaytexture(color *pixels, int w, int h);
void texturerenderer_draw();

// Headless outputs.  NULL disables.  PPM is written upright (last buffer row first) unless
// raw_order is set, in which case rows go out in buffer order.
void texturerenderer_set_output(const char *ppm_path, bool raw_order);
// Optional float frame (3 floats per pixel, same layout) written as a PFM file.
void texturerenderer_set_float_output(const char *pfm_path, const float *rgb32);
// Optional OpenEXR file (scanline, uncompressed): float channels B, G, R from rgb32 and, for every plane
// that is not NULL, N.x N.y N.z (primary-hit normal), Z (primary-hit t) and a UINT channel "id" holding
// hit id + 1 (0 = no hit), and u v (the hit's texture coordinate).  Top row first, as EXR expects.
void texturerenderer_set_exr_output(const char *exr_path, const float *rgb32, const int *ids, const float *hit_t = nullptr,
                                    const float *normal = nullptr, const float *uv = nullptr);
