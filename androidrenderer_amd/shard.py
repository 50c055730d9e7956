"""Row sharding of one frame over N ranks: which rows every rank shades, copies, reduces and composites so that
the two exchanges — half-resolution bloom mip 0, final RGBA8 image — reassemble exactly what one GPU computes.

Every pass of the frame is row-local except the bloom pyramid.  Dependencies, in rows:
  final image row y         samples the scene upside down (scene_upsample.frag, fullscreen.vert: v = 1 - (y + 0.5) / H): `antialiased`
                            rows H - 1 - y +- 1, every bloom mip (global);
  antialiased row j         "Copy scene" (copy_with_sampler.frag.slang): lit rows j - 1 .. j + 1 — and its sampler REPEATS
                            (scene_renderer.cpp:74-79 sets only the filters), so row 0 also taps row H - 1 and row H - 1 taps row 0
                            (with a weight of about 1e-7 or exactly 0, but the texel has to be the right one);
  bloom mip 0 row j         bloom_downsample.comp: antialiased rows 2j - 2 .. 2j + 3;
  bloom mips 1..            mip 0 (replicated after the exchange: 1/4 of mip 0's work in total).
Rank r owns mip 0 rows [r q, (r + 1) q) — gathered in rank order — and the final rows of slot N - 1 - r — gathered in reversed rank
order (sah_allgather_rows_reversed), because of the vertical flip those are the rows whose scene rows it has anyway.
"""
from dataclasses import dataclass


def _clip(a, b, n):
    a, b = max(0, min(a, n)), max(0, min(b, n))
    return (a, max(a, b))


@dataclass(frozen=True)
class ChainPlan:
    rank: int
    world: int
    height: int
    rows_per_rank: int    # final-image slot height: ceil(H / N)
    mip0_height: int
    mip0_rows_per_rank: int
    out_rows: tuple       # final rows this rank composites (slot world - 1 - rank), possibly empty
    mip0_rows: tuple      # bloom mip 0 rows this rank produces (slot rank), possibly empty
    aa_rows: tuple        # antialiased rows it needs (its composite + its mip 0 rows)
    lit_rows: tuple       # lit rows it shades ...
    lit_wrap_rows: tuple  # ... plus the row on the opposite edge that the REPEAT sampler of "Copy scene" taps ((0, 0): none)

    @property
    def out_slot(self):
        return self.world - 1 - self.rank


def chain_plan(height, world, rank):
    per = -(-height // world)
    h2 = max(1, height // 2)  # images.bloom_mip_sizes: mip 0 = output / 2
    q = -(-h2 // world)
    slot = world - 1 - rank
    out = _clip(slot * per, (slot + 1) * per, height)
    m0 = _clip(rank * q, (rank + 1) * q, h2)
    need = []
    if out[1] > out[0]:  # composite: scene rows H - 1 - y for y in out, one row either side for the bilinear taps
        need.append((height - out[1] - 1, height - out[0] + 1))
    if m0[1] > m0[0]:
        need.append((2 * m0[0] - 2, 2 * (m0[1] - 1) + 3 + 1))
    if need:
        aa = _clip(min(a for a, _ in need), max(b for _, b in need), height)
        lit = _clip(aa[0] - 1, aa[1] + 1, height)
        wrap = (0, 0)
        if aa[0] == 0 and lit[1] < height:
            wrap = (height - 1, height)
        elif aa[1] == height and lit[0] > 0:
            wrap = (0, 1)
    else:
        aa = lit = wrap = (0, 0)
    return ChainPlan(rank, world, height, per, h2, q, out, m0, aa, lit, wrap)


def lighting_rows(height, world, rank):
    """Lighting-only workloads: contiguous blocks of ceil(H / N) rows in rank order (equal gather slots, the last ones short or empty)."""
    per = -(-height // world)
    return _clip(rank * per, (rank + 1) * per, height)
