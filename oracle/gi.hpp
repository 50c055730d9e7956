// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
// GI overlays drawn inside the Lighting pass (IGlobalIlluminator::render_to_lit_scene,
// RenderCore/render/gi/global_illuminator.hpp:38-40).  Each returns false on `discard`.
#pragma once
#include "../include/sah_hip.h"
#include "image.hpp"
#include "math.hpp"

namespace orc {

Image img2d(const sah_plane& p);
Image img3d(const sah_volume& v);
M4 mat(const float* m);
F3 viewspace_position_glsl(const sah_view_data& view, int x, int y, float depth);
F3 worldspace_location_slang(const sah_view_data& view, int x, int y, float depth);

struct GBufferTexels {
    float depth;
    Texel color, normal, data, emission;
};

bool gi_lpv_frag(const sah_lighting_desc& d, int x, int y, float depth, const Texel& color, const Texel& normal, const Texel& data,
                 F out[4]);
bool gi_cache_frag(const sah_lighting_desc& d, int x, int y, float depth, const Texel& color, const Texel& normal,
                   const Texel& data, F out[4]);
bool gi_rtgi_frag(const sah_lighting_desc& d, int x, int y, float depth, const Texel& color, const Texel& normal, const Texel& data,
                  F out[4]);

}  // namespace orc
