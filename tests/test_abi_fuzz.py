"""Every entry point of include/sah_hip.h answers made-up arguments with a status code: tests/abi_fuzz_child.py throws
random extents, pitches, formats, row ranges, device addresses and null sub-pointers at a context that has no device behind it
(sah_debug_create_detached), in a child process so that a fault fails the test instead of ending the run.  CPU only: where a HIP device
exists the hook refuses (the made-up addresses must never reach a GPU) and the test is skipped."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_no_argument_combination_crashes_an_entry_point(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_fuzz_child.py"), str(seed), "1500"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-25:])
    if "SKIP:" in r.stdout:
        pytest.skip(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0, f"the fuzz child ended with code {r.returncode} (negative: a signal — a fault inside the library):\n{tail}"
    assert f"OK: 1500 iterations" in r.stdout, tail
