// glwt_headless.cpp — headless stand-in for the Cocoa/OpenGL window layer
// (Raytracer/glwt.mm:86-158): same call order, no window.
#include "glwt.h"

extern "C" int initglwt(const char *, int, int, bool)
{
    setup();      // glwt.mm:153
    draw(0.0f);   // one tick of glwt.mm:68-82; the frame sink flushes its files here
    return 0;
}
