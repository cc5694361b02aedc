// glwt_headless.cpp — headless stand-in for the Cocoa/OpenGL window layer
// (Raytracer/glwt.mm:86-158): same call order, no window.
#include "glwt.h"

int glwt_headless_frames __attribute__((weak)) = 1;   // ticks to run; the application may set it before initglwt

extern "C" int initglwt(const char *, int, int, bool)
{
    setup();      // glwt.mm:153
    // glwt.mm:68-82 ticks draw(dt) from a display timer; here a fixed 1/60 s step, once per frame
    // wanted.  The first tick presents the frame setup() rendered; the frame sink flushes its files
    // on the last one.
    for (int i = 0; i < glwt_headless_frames; i++) draw(1.0f / 60.0f);
    return 0;
}
