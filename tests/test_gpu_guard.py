"""The out-of-bounds hunt as part of the driver's `-m gpu` run.

Round 5 found two defects that four rounds of green tests had hidden (a conv staging read past tensors shorter than a tile; a weight-gradient
kernel summing across row ends at odd widths) by placing every device tensor at the END of its own mapping, so that an access past it
faults instead of landing in a neighbouring allocator block.  A fault kills the process -- so each hunt runs in a SUBPROCESS and the test
asserts on its exit code and last line."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(args, timeout):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    return p.returncode, p.stdout.decode(errors="replace")


def test_abi_entry_points_stay_inside_end_aligned_tensors():
    """tools/abi_guard_sweep.py: every comparison of tests/test_gpu_abi_ref.py / test_gpu_abi_sweeps.py (one argument list to the HIP entry
    point and its plain-C twin; widths 1..13 x strides x taps, edge shapes) with each device buffer at the end of its own 20 MB segment"""
    rc, out = _run([os.path.join(ROOT, "tools", "abi_guard_sweep.py")], 1500)
    assert rc == 0 and "no out-of-bounds access reached an unmapped page" in out, out[-3000:]


def test_engine_steps_under_the_guard_allocator():
    """tools/engine_guard_run.py tools/guard_engine_steps.py: training steps at one and three clips of odd lengths, g = 1 / 4, bf16 / f32,
    three losses, with torch's allocator replaced by one hipMalloc block per tensor, end-aligned (tools/guard_alloc)"""
    rc, out = _run([os.path.join(ROOT, "tools", "engine_guard_run.py"), os.path.join(ROOT, "tools", "guard_engine_steps.py")], 1500)
    assert rc == 0 and "guard steps ok" in out, out[-3000:]
