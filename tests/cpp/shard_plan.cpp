// CPU-only: prints sah::shard_chain_plan (include/sah_host.hpp) for the (height, world) pairs on the command line, one line per rank;
// tests/test_shard_chain.py compares every field with androidrenderer_amd/shard.py: chain_plan.  No GPU call is made.
#include <cstdio>
#include <cstdlib>

#include "sah_host.hpp"

int main(int argc, char** argv) {
    for (int i = 1; i + 1 < argc; i += 2) {
        const uint32_t height = (uint32_t)atoi(argv[i]), world = (uint32_t)atoi(argv[i + 1]);
        for (uint32_t r = 0; r < world; r++) {
            const sah::ShardPlan p = sah::shard_chain_plan(height, world, r);
            const sah_chain_plan& c = p.chain;
            printf("%u %u %u  %u %u  %u %u  %u %u  %u %u  %u %u  %u %u  %u %u %u %u %u %u\n", height, world, r, c.out_rows[0], c.out_rows[1], c.mip1_rows[0], c.mip1_rows[1],
                   c.mip0_rows[0], c.mip0_rows[1], c.aa_rows[0], c.aa_rows[1], p.lit_rows[0], p.lit_rows[1], p.lit_wrap_rows[0], p.lit_wrap_rows[1], c.rows_per_rank,
                   c.mip1_rows_per_rank, c.out_allocated_rows, c.mip1_allocated_rows, p.mip0_height, p.mip1_height);
        }
    }
    return 0;
}
