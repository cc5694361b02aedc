// multigpu.h — one frame on N GPUs of one node from ONE host process.  The reference is single-device
// (Raytracer/main.cpp:289-292 -> setup() renders the whole frame); its pixel loop has no state across pixels
// (main.cpp:262-277), so the rows are dealt to the GPUs in bands round-robin, every GPU renders its bands into a
// compact slice through rt_render_rows_device, the slices are all-gathered with RCCL and GPU 0 puts the bands
// back into image order (rt_deinterleave_bands).
#pragma once

#include <stdint.h>

#include <vector>

#include "rt_shim.h"

struct MultiGpuFrame;

// One context per device must exist already (scene, hierarchy, camera set on each).  Creates the communicator, the
// streams and the slice / gather / frame buffers.  NULL on failure (message on stderr).
MultiGpuFrame *multigpu_create(const std::vector<rt_ctx *> &contexts, const std::vector<int> &devices, int W, int H, int band_rows);
// Renders one frame into `rgb8` (3*W*H bytes, host; fastest when it comes from rt_alloc_host_buffer of contexts[0]).
// rays[3] receives the primary / shadow / reflection rays of the whole frame, device_ms the longest device's frame time.
bool multigpu_render(MultiGpuFrame *f, int max_depth, uint8_t *rgb8, uint64_t rays[3], float *device_ms);
void multigpu_destroy(MultiGpuFrame *f);
