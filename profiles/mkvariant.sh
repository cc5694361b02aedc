#!/bin/bash
# profiles/mkvariant.sh NAME -DFLAG=... : build raytracer_amd/variants/NAME.so with extra compile flags (A/B runs)
R="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
mkdir -p "$R/raytracer_amd/variants"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  -I"$R/include" "$@" "$R"/raytracer_amd/csrc/{rt_shim,trace,wavefront,shade,lbvh}.hip -o "$R/raytracer_amd/variants/$name.so" 2>&1 | grep -i "error" 
echo "built $name"
