"""Build libdropdec.so (the C-ABI HIP library) and libdropdec_tools.so (the same objects + the measurement hooks of
csrc/dd_tools.hip, for bench.py's roofline leg and tools/) in-tree with hipcc for gfx950.

    python -m dropoutdecoding_amd.build [--force]

The built .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libdropdec.so")
TOOLS_LIB = os.path.join(HERE, "libdropdec_tools.so")
TOOLS_SOURCES = ["dd_tools.hip"]      # only in libdropdec_tools.so
SOURCES = ["dd_dropout.hip", "dd_lm_kernels.hip", "dd_gemv.hip", "dd_attn_decode.hip", "dd_prefill.hip", "dd_engine.hip", "dd_tp.hip", "dd_vision.hip"]
HEADERS = ["dd_common.h", "dd_lm_kernels.h", "dd_lm_device.h", "dd_gemv_slices.h", "dd_engine_internal.h",
           os.path.join(ROOT, "include", "dropdec_tools.h"), os.path.join(ROOT, "include", "dropdec.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]


def _file_hash(paths) -> str:
    import hashlib
    h = hashlib.sha256()
    for d in paths:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _header_paths():
    return [x if os.path.isabs(x) else os.path.join(CSRC, x) for x in HEADERS]


def _src_hash() -> str:
    return _file_hash([os.path.join(CSRC, s) for s in SOURCES + TOOLS_SOURCES] + _header_paths()) + "|" + " ".join(FLAGS)


def _stale() -> bool:
    """Content hash, not mtimes: the gpurun snapshot does not preserve modification times."""
    stamp = LIB + ".srchash"
    if not (os.path.exists(LIB) and os.path.exists(TOOLS_LIB) and os.path.exists(stamp)):
        return True
    return open(stamp).read().strip() != _src_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the sources whose content (or a header's) changed — all of them in parallel — and link."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, jobs = [], []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    tools_objs = []
    for s in SOURCES + TOOLS_SOURCES:
        o = os.path.join(bdir, s.replace(".hip", ".o"))
        (tools_objs if s in TOOLS_SOURCES else objs).append(o)
        want = _file_hash([os.path.join(CSRC, s)] + _header_paths()) + "|" + " ".join(FLAGS)
        stamp = o + ".srchash"
        if not force and os.path.exists(o) and os.path.exists(stamp) and open(stamp).read().strip() == want:
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        jobs.append((subprocess.Popen(cmd), cmd, stamp, want))
    for pr, cmd, stamp, want in jobs:
        if pr.wait() != 0:
            for other, *_ in jobs:
                if other.poll() is None:
                    other.kill()
            raise subprocess.CalledProcessError(pr.returncode, cmd)
        with open(stamp, "w") as f:
            f.write(want)
    for lib, members in ((LIB, objs), (TOOLS_LIB, objs + tools_objs)):
        tmp = lib + f".tmp{os.getpid()}"
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + members
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, lib)                   # atomic: concurrent ranks never see a half-written library
    with open(LIB + ".srchash", "w") as f:
        f.write(_src_hash())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
