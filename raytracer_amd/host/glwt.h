// glwt.h — the application callback seam of the reference's window toolkit
// (Raytracer/glwt.h:18-20), headless: there is no window, no GL.
#pragma once

#ifdef __cplusplus
extern "C" {
#endif

// Runs setup() once, then draw(1/60) once per frame wanted (glwt_headless_frames, default 1), and
// returns 0 (the reference returns 0 after the Cocoa run loop ends, -1 when GL cannot be
// initialised: glwt.mm:140-157).
int initglwt(const char *title, int width, int height, bool fullscreen);
void setup();
void draw(float dt);

#ifdef __cplusplus
}
#endif
