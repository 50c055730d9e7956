// Texel <-> direction of the octahedral probe maps (RenderCore/shaders/common/octahedral.slangi:25-50), shared by the probe
// maintenance passes (probes.hip) and the probe ray generator (rt.hip).
#pragma once
#include "numerics.hpp"

namespace sah {

struct F2d {
    Fn x, y;
};
// octahedral.slangi:25-39
SAH_DEV F2d normalized_octahedral_coordinates(uint32_t tx, uint32_t ty, uint32_t nx, uint32_t ny) {
    Fn cx = Fn((float)(tx % nx)), cy = Fn((float)(ty % ny));
    cx = cx + Fn(0.5f);
    cy = cy + Fn(0.5f);
    cx = cx / Fn((float)nx);
    cy = cy / Fn((float)ny);
    cx = cx * Fn(2.f);
    cy = cy * Fn(2.f);
    return {cx - Fn(1.f), cy - Fn(1.f)};
}
// octahedral.slangi:44-50
SAH_DEV F3 octahedral_direction(F2d c) {
    F3 d = {c.x, c.y, Fn(1.f) - nabs(c.x) - nabs(c.y)};
    const Fn sx = Fn(d.x.v >= 0.f ? 1.f : -1.f), sy = Fn(d.y.v >= 0.f ? 1.f : -1.f);
    const Fn nx = (Fn(1.f) - nabs(d.y)) * sx, ny = (Fn(1.f) - nabs(d.x)) * sy;
    const bool fold = d.z.v < 0.f;
    d.x = fold ? nx : d.x;
    d.y = fold ? ny : d.y;
    return normalize(d);
}

}  // namespace sah
