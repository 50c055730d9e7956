"""Row sharding of one frame over N ranks: which rows every rank shades, copies, reduces and composites so that
the two exchanges — the quarter-resolution bloom mip 1, the final RGBA8 image — reassemble exactly what one GPU computes.

Every pass of the frame is row-local except the bloom pyramid.  Dependencies, in rows:
  final image row y         samples the scene upside down (scene_upsample.frag, fullscreen.vert: v = 1 - (y + 0.5) / H): `antialiased`
                            rows H - 1 - y +- 1, bloom mip 0 rows within 3 of (1 - (y + 0.5) / H) * H0 - 0.5 (tent taps one texel either
                            side, each a bilinear pair), every smaller mip (global);
  antialiased row j         "Copy scene" (copy_with_sampler.frag.slang): lit rows j - 1 .. j + 1 — and its sampler REPEATS
                            (scene_renderer.cpp:74-79 sets only the filters), so row 0 also taps row H - 1 and row H - 1 taps row 0
                            (with a weight of about 1e-7 or exactly 0, but the texel has to be the right one);
  bloom mip 0 row j         bloom_downsample.comp: antialiased rows 2j - 2 .. 2j + 3 (when the mip is exactly half as high; in general the
                            rows within two of (j + 0.5) * H / H0 - 0.5: _downsample_sources);
  bloom mip 1 row j         mip 0 rows, likewise;
  bloom mips 2..            mip 1 (replicated after the exchange: 1/16 of mip 0's work in total).
Rank r owns mip 1 rows [r q, (r + 1) q) — gathered in rank order — and the final rows of slot N - 1 - r — gathered in reversed rank
order (sah_allgather_rows_reversed): because of the vertical flip those are the rows whose scene rows, and whose mip 0 rows, it has anyway.
Mip 0 is NOT exchanged (round 4; rounds 2-3 gathered it, 16.6 MB at 4K against mip 1's 4.1 MB): every rank computes the mip 0 rows its
own mip 1 rows and its own rows of the composite read, from a few more rows of lighting and copy than its band (about 24 of 270 at N = 8).
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
    mip1_height: int
    mip1_rows_per_rank: int
    out_rows: tuple       # final rows this rank composites (slot world - 1 - rank), possibly empty
    mip1_rows: tuple      # bloom mip 1 rows this rank produces and contributes to the exchange (slot rank), possibly empty
    mip0_rows: tuple      # bloom mip 0 rows it computes for itself (sources of its mip 1 rows and of its composite; not exchanged)
    aa_rows: tuple        # antialiased rows it needs (its composite + its mip 0 rows)
    lit_rows: tuple       # lit rows it shades ...
    lit_wrap_rows: tuple  # ... plus the row on the opposite edge that the REPEAT sampler of "Copy scene" taps ((0, 0): none)

    @property
    def out_slot(self):
        return self.world - 1 - self.rank


def _downsample_sources(j0, j1, src_height, dst_height):
    """Source rows the bloom downsample reads for destination rows [j0, j1) (bloom_downsample.comp: 20 bilinear taps within two source
    texels of the centre c = (j + 0.5) * hs / hd - 0.5 — which is 2j + 0.5 only when hs = 2 hd: a 75-row mip under a 37-row one drifts by
    a row over its height).  Exact integer floors; when the mip is not exactly half as high, one row of slack either side for the shader's
    fp32 coordinate (with hs = 2 hd every tap coordinate is an integer + 0.5: no rounding can move its floor)."""
    slack = 0 if src_height == 2 * dst_height else 1
    lo = ((2 * j0 + 1) * src_height - dst_height - 4 * dst_height) // (2 * dst_height) - slack
    hi = ((2 * (j1 - 1) + 1) * src_height - dst_height + 4 * dst_height) // (2 * dst_height) + 1 + slack
    return (lo, hi + 1)


def _hull(ranges):
    ranges = [r for r in ranges if r[1] > r[0]]
    return (min(a for a, _ in ranges), max(b for _, b in ranges)) if ranges else (0, 0)


def chain_plan(height, world, rank):
    per = -(-height // world)
    h0 = max(1, height // 2)  # images.bloom_mip_sizes: mip 0 = output / 2, every further mip half of that
    h1 = max(1, h0 // 2)
    q = -(-h1 // world)
    slot = world - 1 - rank
    out = _clip(slot * per, (slot + 1) * per, height)
    m1 = _clip(rank * q, (rank + 1) * q, h1)
    need0, need_aa = [], []
    if out[1] > out[0]:
        # composite: scene rows H - 1 - y for y in out, one row either side for the bilinear taps; mip 0 rows around
        # p(y) = (1 - (y + 0.5) / H) * H0 - 0.5 = ((2 (H - y) - 1) * H0 - H) / (2 H) (exactly), three either side of floor(p)
        need_aa.append((height - out[1] - 1, height - out[0] + 1))
        # (taps at p - 1 .. p + 1, each with its bilinear partner: rows floor(p) - 1 .. floor(p) + 2; a row of slack either side unless
        # mip 0 is exactly half as high as the image — p is then an integer + 0.25 or + 0.75 and no fp32 rounding moves a floor)
        slack = 0 if height == 2 * h0 else 1
        p_lo = ((2 * (height - (out[1] - 1)) - 1) * h0 - height) // (2 * height)
        p_hi = ((2 * (height - out[0]) - 1) * h0 - height) // (2 * height)
        need0.append((p_lo - 1 - slack, p_hi + 2 + slack + 1))
    if m1[1] > m1[0]:
        need0.append(_downsample_sources(m1[0], m1[1], h0, h1))
    m0 = _clip(*_hull(need0), h0)
    if m0[1] > m0[0]:
        need_aa.append(_downsample_sources(m0[0], m0[1], height, h0))
    if need_aa:
        aa = _clip(*_hull(need_aa), height)
        lit = _clip(aa[0] - 1, aa[1] + 1, height)
        wrap = (0, 0)
        if aa[0] == 0 and lit[1] < height:
            wrap = (height - 1, height)
        elif aa[1] == height and lit[0] > 0:
            wrap = (0, 1)
    else:
        aa = lit = wrap = (0, 0)
    return ChainPlan(rank, world, height, per, h0, h1, q, out, m1, m0, aa, lit, wrap)


def lighting_rows(height, world, rank):
    """Lighting-only workloads: contiguous blocks of ceil(H / N) rows in rank order (equal gather slots, the last ones short or empty)."""
    per = -(-height // world)
    return _clip(rank * per, (rank + 1) * per, height)
