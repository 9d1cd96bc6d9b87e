"""Host-side checks of round 6 (no GPU): the build gate of ADVICE round 5, the timing experiments kept out of the product objects."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_build_gate_refuses_a_forced_build_on_a_fresh_tree():
    """DD_NO_BUILD / --no-build: build() never starts a compile — not on a stale tree (require_fresh raises) and not with force=True on a
    fresh one (that passed the gate and compiled under the profiler)."""
    from dropoutdecoding_amd import build
    build.build()                                       # make the tree fresh (compiles here at most once per session)
    code = ("import os; os.environ['DD_NO_BUILD'] = '1'\n"
            "from dropoutdecoding_amd import build\n"
            "assert build.build().endswith('libdropdec.so')\n"
            "try:\n    build.build(force=True)\nexcept RuntimeError as e:\n    print('REFUSED', e)\nelse:\n    print('COMPILED')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    assert "REFUSED" in r.stdout and "may not compile" in r.stdout


def test_timing_experiments_are_compiled_into_the_tools_objects_only():
    """dd_gemv_slices.h's timing-only branches (skipped stage-in, dropped partial sums: garbage results) sit behind DD_TEXP, which is a
    constant 0 unless -DDD_TIMING_EXPERIMENTS — a flag only the tools variant of dd_gemv.hip gets."""
    from dropoutdecoding_amd import build
    assert "-DDD_TIMING_EXPERIMENTS" in build.TOOLS_VARIANTS["dd_gemv.hip"]
    assert "-DDD_TIMING_EXPERIMENTS" not in build.FLAGS
    src = open(os.path.join(build.CSRC, "dd_gemv_slices.h")).read()
    assert "a.temporal & 2" not in src and "a.temporal & 4" not in src and "a.temporal & 8" not in src
    assert "#define DD_TEXP(a_, bit_) 0" in src
