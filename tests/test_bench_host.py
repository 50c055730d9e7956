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
    assert with_rec["bound"] in ("valu", "hbm", "fp32", "latency") and with_rec["traffic"] and with_rec["valu_issue"]["frac"] <= 1.0
    assert with_rec["traffic_over_algorithmic"] > 1.0 and 0 < with_rec["traffic_frac_of_peak"] < 1
    # a launch that takes three times as long with the same instructions and bytes is bound by neither roofline
    slow = bench.roofline("4k_deferred_gi", 1, 1750.0 / 3, 0.17 * 3, None, "x", 298598400, 3840 * 2160, "k")
    assert slow["bound"] == "latency" and slow["bound_by_fraction"] in ("valu", "hbm") and "latency" in slow["bound_note"]
    without = bench.roofline("no_such_workload", 1, 1750.0, 0.17, None, "x", 298598400, 3840 * 2160, "k")
    assert without["bound"] is None and "bound_note" in without and without["traffic"] is None


def test_metric_names_what_was_measured():
    """the headline keeps BASELINE.json's metric verbatim; any other workload (the N > 1 default is the configs[3] frame) names itself"""
    assert bench.metric_name("4k_deferred_gi", False, 3840, 2160) == json.load(open(os.path.join(bench.ROOT, "BASELINE.json")))["metric"].split(";")[0]
    chain = bench.metric_name("4k_probe_gi_chain", True, 3840, 2160)
    assert "final-image" in chain and "4k_probe_gi_chain" in chain and "deferred+GI pass" not in chain
    assert "8k_1024_lights_gi" in bench.metric_name("8k_1024_lights_gi", False, 7680, 4320)


def test_gpus_n_without_a_launcher_starts_its_ranks_as_a_child_before_anything_touches_the_gpu(tmp_path):
    """`python bench.py --gpus 2 ...` with WORLD_SIZE unset: the parent must start `python -m torch.distributed.run ... bench.py <same arguments>` as a
    child, relay the child's one JSON line and its exit code — and must not have imported torch (let alone called HIP) before it does so.  The child is
    a stand-in here: a `python` whose `-m torch.distributed.run` prints what it was given."""
    import subprocess
    import sys
    import textwrap
    fake = tmp_path / "fakepy"
    fake.write_text(textwrap.dedent(f"""\
        #!{sys.executable}
        import json, os, sys
        assert sys.argv[1:3] == ["-m", "torch.distributed.run"], sys.argv
        print("RCCL banner that is not the line")
        print(json.dumps({{"metric": "m", "argv": sys.argv[3:], "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}}))
        sys.exit(int(os.environ.get("FAKE_RC", "0")))
        """))
    fake.chmod(0o755)
    probe = textwrap.dedent(f"""\
        import os, sys
        sys.path.insert(0, {bench.ROOT!r})
        import bench
        sys.executable = {str(fake)!r}
        os.environ.pop("WORLD_SIZE", None)
        try:
            bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse-on-one-gpu"])
        except SystemExit as e:
            print("TORCH_IMPORTED" if "torch" in sys.modules else "TORCH_NOT_IMPORTED", file=sys.stderr)
            raise
        """)
    for rc in (0, 3):
        out = subprocess.run([sys.executable, "-c", probe], env=dict(os.environ, FAKE_RC=str(rc)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        assert out.returncode == rc, out.stderr[-2000:]
        lines = out.stdout.splitlines()
        assert len(lines) == 1, out.stdout  # the one line, nothing else on stdout
        d = json.loads(lines[0])
        a = d["argv"]
        assert a[a.index("--nproc-per-node") + 1] == "2" and a[a.index("--master-addr") + 1] == "127.0.0.1" and "--nnodes=1" in a
        assert a[-7:] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse-on-one-gpu"] and a[-8].endswith("bench.py")
        assert d["ipc"] == "0"
        assert "TORCH_NOT_IMPORTED" in out.stderr and "RCCL banner" in out.stderr


def test_a_launcher_with_another_world_size_is_refused():
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(bench.ROOT, "bench.py"), "--gpus", "4"], env=dict(os.environ, WORLD_SIZE="2", RANK="0"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE is 2" in out.stderr
