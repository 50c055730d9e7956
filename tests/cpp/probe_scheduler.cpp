// CPU-only check of sah::ProbeScheduler (include/sah_host.hpp) — RenderCore/render/gi/irradiance_cache.cpp:351-360,496-583.
// Prints one line per scenario; tests/test_probe_scheduler.py reads them.  No GPU call is made (the façade header only needs the HIP
// runtime declarations to compile).
#include <cstdio>
#include <random>
#include <set>

#include "sah_host.hpp"

using sah::ProbeScheduler;

int main() {
    // 1. fresh cache, frame 1: every probe is invalid; each is requested with probability priority / total = 0.25 until the budget
    //    (1024) is hit, walking cascade 0 in foreach order (x outermost, z innermost)
    {
        ProbeScheduler s;
        s.find_probes_to_update(1);
        const auto& list = s.get_probes_to_update();
        // replay the documented algorithm by hand
        auto rng = std::default_random_engine{1u};
        auto dist = std::uniform_real_distribution<float>{0.f, 1.f};
        size_t i = 0;
        bool same = true;
        for (uint32_t x = 0; x < 32 && i < list.size(); x++)
            for (uint32_t y = 0; y < 8 && i < list.size(); y++)
                for (uint32_t z = 0; z < 32 && i < list.size(); z++)
                    if (dist(rng) < 0.25f) {
                        same = same && list[i][0] == x && list[i][1] == y && list[i][2] == z;
                        i++;
                    }
        std::set<ProbeScheduler::ProbeIndex> unique(list.begin(), list.end());
        printf("fresh count=%zu replay=%d unique=%zu first=%u,%u,%u\n", list.size(), (int)(same && i == list.size()), unique.size(), list[0][0], list[0][1], list[0][2]);
        // the requested probes are now valid and stamped
        uint32_t valid = 0;
        for (const auto& p : s.cascade(0).probes) valid += p.is_valid && p.last_update_frame == 1;
        printf("fresh valid_in_cascade0=%u\n", valid);
    }
    // 2. same frame number, same list (the engine is seeded with the frame count)
    {
        ProbeScheduler a, b;
        a.find_probes_to_update(7);
        b.find_probes_to_update(7);
        ProbeScheduler c;
        c.find_probes_to_update(8);
        printf("seeded same=%d different=%d\n", (int)(a.get_probes_to_update() == b.get_probes_to_update()), (int)(a.get_probes_to_update() != c.get_probes_to_update()));
    }
    // 3. everything valid and fresh: the first pass requests nothing; the second scores log(seconds since update): negative for less
    //    than a second (60 frames), so nothing is requested at frame 30 and something is at frame 6000 (log(100) * 0.25 > 1: all of them)
    {
        ProbeScheduler s(4096);
        for (uint32_t c = 0; c < 4; c++)
            for (auto& p : s.cascade(c).probes) p.is_valid = true;
        s.find_probes_to_update(30);
        const size_t young = s.get_probes_to_update().size();
        s.find_probes_to_update(6000);
        const auto& list = s.get_probes_to_update();
        // quirk: the index is cascade-local (no y + 8 * cascade), so every y stays below 8 even when later cascades are reached
        uint32_t max_y = 0;
        for (const auto& p : list) max_y = p[1] > max_y ? p[1] : max_y;
        printf("aged young=%zu old=%zu max_y=%u\n", young, list.size(), max_y);
    }
    // 4. the budget is a hard cap and a full list short-circuits the next call
    {
        ProbeScheduler s(10);
        s.find_probes_to_update(3);
        const size_t first = s.get_probes_to_update().size();
        s.find_probes_to_update(4);
        const size_t second = s.get_probes_to_update().size();
        s.clear_probes_to_update();
        s.find_probes_to_update(5);
        printf("budget first=%zu second=%zu after_clear=%zu refused=%d\n", first, second, s.get_probes_to_update().size(), (int)!s.request_probe_update({0, 0, 0}));
    }
    return 0;
}
