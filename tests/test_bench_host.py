"""Host-side pieces of bench.py that need no GPU: the workload table, the roofline object's bound naming, the CPU count the baseline uses."""
import json
import os

import bench


def test_usable_cpus_is_what_the_process_may_run_on():
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))


def test_every_workload_has_a_static_counter_record():
    """profiles/roofline_static.json: one record per bench.py workload, each naming its kernel and its source file"""
    table = json.load(open(os.path.join(bench.ROOT, "profiles", "roofline_static.json")))
    for name, spec in bench.WORKLOADS.items():
        assert name in table, f"no static record for {name}"
        rec = table[name]
        assert rec["pixels"] == spec["res"][0] * spec["res"][1]
        assert rec["valu_wave_insts_per_launch"] > 0 and os.path.exists(os.path.join(bench.ROOT, rec["source"]))
        tiled = bool(spec.get("lights")) or spec.get("gi") == "cache"
        assert ("k_lighting_tiled" if tiled else "k_lighting_fast") in rec["kernel"]


def test_roofline_names_a_bound_only_with_a_record():
    with_rec = bench.roofline("4k_deferred_gi", 1, 1750.0, 0.17, None, "x", 298598400, 3840 * 2160, "k")
    assert with_rec["bound"] in ("valu", "hbm", "fp32") and with_rec["traffic"] and with_rec["valu_issue"]["frac"] <= 1.0
    without = bench.roofline("no_such_workload", 1, 1750.0, 0.17, None, "x", 298598400, 3840 * 2160, "k")
    assert without["bound"] is None and "bound_note" in without and without["traffic"] is None
