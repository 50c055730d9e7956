"""CPU: sah::ProbeScheduler (include/sah_host.hpp), the probe scheduling of the irradiance cache (CPU side) —
RenderCore/render/gi/irradiance_cache.cpp:351-360,496-583.  tests/cpp/probe_scheduler.cpp runs the scenarios; no GPU is touched."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run():
    src = os.path.join(ROOT, "tests", "cpp", "probe_scheduler.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "probe_scheduler")
    hdr = os.path.join(ROOT, "include", "sah_host.hpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", src,
                               "-o", exe])
    out = subprocess.check_output([exe], text=True)
    rows = {}
    for line in out.splitlines():
        name, *fields = line.split()
        rows.setdefault(name, {}).update(dict(f.split("=") for f in fields))
    return rows


def test_probe_scheduler_follows_the_reference_algorithm():
    r = _run()
    # a fresh cache fills its budget from cascade 0 in foreach order with the standard library's default engine seeded by the frame
    assert r["fresh"]["count"] == "1024" and r["fresh"]["replay"] == "1" and r["fresh"]["unique"] == "1024"
    assert r["fresh"]["valid_in_cascade0"] == "1024"
    assert r["seeded"] == {"same": "1", "different": "1"}
    # ageing: log(seconds) < 0 within the first second; after 100 s every probe scores above 1; indices stay cascade-local (quirk)
    assert r["aged"] == {"young": "0", "old": "4096", "max_y": "7"}
    assert r["budget"] == {"first": "10", "second": "10", "after_clear": "10", "refused": "1"}
