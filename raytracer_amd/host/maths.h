// maths.h — the two vector types the host side needs (counterpart of the reference's
// Raytracer/maths.h:31-173; the trace arithmetic itself runs on the GPU).
#pragma once

struct vec2 {
    float x = 0.0f, y = 0.0f;
    vec2() {}
    vec2(float x_, float y_) : x(x_), y(y_) {}
};

struct vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    vec3() {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
