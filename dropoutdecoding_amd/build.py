"""Build libdropdec.so (the C-ABI HIP library) and libdropdec_tools.so (the same objects + the measurement hooks of
csrc/dd_tools.hip, for bench.py's roofline leg and tools/) in-tree with hipcc for gfx950.

    python -m dropoutdecoding_amd.build [--force]

The built .so is git-ignored but travels to the GPU box with the gpurun snapshot.

Gate: every object is compiled with -Rpass-analysis=kernel-resource-usage and the build FAILS when a kernel of the product
library reports private scratch (ScratchSize > 0).  Round 3's one run-to-run token difference sat in the one kernel of 381
that had any (DESIGN.md "Determinism"); a by-value struct indexed at run time or an unroll budget that is exceeded puts
arrays into scratch silently, and scratch is both slow and allocated per queue by the runtime at dispatch time.
"""
from __future__ import annotations

import fcntl
import os
import re
import signal
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libdropdec.so")
TOOLS_LIB = os.path.join(HERE, "libdropdec_tools.so")
TOOLS_SOURCES = ["dd_tools.hip"]      # only in libdropdec_tools.so
SOURCES = ["dd_dropout.hip", "dd_lm_kernels.hip", "dd_gemv.hip", "dd_attn_decode.hip", "dd_prefill.hip", "dd_engine.hip", "dd_tp.hip", "dd_vision.hip"]
# sources that libdropdec_tools.so takes in a second compilation with extra macros (A/B variants that must not be in the product)
TOOLS_VARIANTS = {"dd_dropout.hip": ["-DDD_KEEP_SCRATCH_SAMPLER"],
                  # the slice GEMVs' timing experiments (dd_gemv_slices.h DD_TEXP: skipped stage-in, dropped partial sums, round-5 store
                  # placement) are compiled into the tools library only
                  "dd_gemv.hip": ["-DDD_TIMING_EXPERIMENTS"]}
HEADERS = ["dd_common.h", "dd_lm_kernels.h", "dd_lm_device.h", "dd_gemv_slices.h", "dd_engine_internal.h", "dd_sampler_block.h",
           os.path.join(ROOT, "include", "dropdec_tools.h"), os.path.join(ROOT, "include", "dropdec.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage"]
# Product objects are compiled WITHOUT packed FP32 VALU instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32).  Found in round 4 (DESIGN.md
# "Determinism"): on gfx950 a wave's v_pk_fma_f32 over operands just returned from LDS gives wrong sums while a workgroup of an MFMA kernel (the
# slice-resident GEMVs of another branch of the step) is resident on the same CU — tools/sampler_repro.py `pv_step_probe` shows it in isolation,
# tests/test_gpu_sampler_repro.py guards it.  The scalar forms are unaffected; nothing on the path is bound by VALU issue.  dd_tools.hip keeps the
# packed forms: the probe needs them.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# kernels allowed to use scratch in the PRODUCT library: none
SCRATCH_ALLOWED: tuple = ()


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
    return _file_hash([os.path.join(CSRC, s) for s in SOURCES + TOOLS_SOURCES] + _header_paths() + [__file__]) + "|" + " ".join(FLAGS)


def _stale() -> bool:
    """Content hash, not mtimes: the gpurun snapshot does not preserve modification times."""
    stamp = LIB + ".srchash"
    if not (os.path.exists(LIB) and os.path.exists(TOOLS_LIB) and os.path.exists(stamp)):
        return True
    return open(stamp).read().strip() != _src_hash()


def require_fresh() -> None:
    """Raise instead of compiling: for processes that must not spawn hipcc (a program running under rocprofv3 has the GPU initialised by the
    profiler's preload before it starts, and on this pool an exec from such a process takes the machine down — ADVICE round 4)."""
    if _stale():
        raise RuntimeError("libdropdec.so is stale and this process may not compile (DD_NO_BUILD / --no-build): run "
                           "`python3 -m dropoutdecoding_amd.build` first")


_REMARK = re.compile(r"remark: (?:\s*)(Function Name|ScratchSize \[bytes/lane\]|VGPRs|AGPRs|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)")


def parse_resource_remarks(text: str):
    """[{name, scratch, vgprs, agprs, lds, occupancy}] from hipcc's -Rpass-analysis=kernel-resource-usage output."""
    out, cur = [], None
    for line in text.splitlines():
        m = _REMARK.search(line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v, "scratch": 0, "vgprs": 0, "agprs": 0, "lds": 0, "occupancy": 0}
            out.append(cur)
        elif cur is not None:
            key = {"ScratchSize [bytes/lane]": "scratch", "VGPRs": "vgprs", "AGPRs": "agprs", "LDS Size [bytes/block]": "lds",
                   "Occupancy [waves/SIMD]": "occupancy"}[k]
            cur[key] = int(v)
    return out


def _demangle(name: str) -> str:
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True, timeout=10).stdout.strip() or name
    except Exception:
        return name


def kernel_resources():
    """Resource usage of every kernel of the last build: {object file name: [records]} (from build/<obj>.remarks)."""
    bdir = os.path.join(HERE, "build")
    res = {}
    for f in sorted(os.listdir(bdir)) if os.path.isdir(bdir) else []:
        if f.endswith(".remarks"):
            res[f[:-len(".remarks")]] = parse_resource_remarks(open(os.path.join(bdir, f)).read())
    return res


def _check_scratch(objs) -> None:
    bad = []
    for o in objs:
        rp = o + ".remarks"
        if not os.path.exists(rp):
            continue
        for k in parse_resource_remarks(open(rp).read()):
            if k["scratch"] > 0 and not any(a in k["name"] for a in SCRATCH_ALLOWED):
                bad.append((os.path.basename(o), _demangle(k["name"]), k["scratch"]))
    if bad:
        msg = "\n".join(f"  {o}: {n}: ScratchSize {s} bytes/lane" for o, n, s in bad)
        raise RuntimeError("kernels of the product library use private scratch (see dropoutdecoding_amd/build.py):\n" + msg)


def packed_fp32_ops(obj_path: str) -> int:
    """Number of packed FP32 VALU instructions in the gfx950 code of a (fat) object file."""
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="ddobj")
    try:
        shutil.copy(obj_path, os.path.join(d, "x.o"))
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", "x.o"], cwd=d, capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            raise RuntimeError(f"llvm-objdump --offloading {obj_path}: rc {r.returncode}: {r.stderr[-300:]}")
        n, seen = 0, 0
        for f in os.listdir(d):
            if "amdgcn" in f:
                seen += 1
                r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f], cwd=d, capture_output=True, text=True,
                                   timeout=600)
                if r.returncode != 0 or "s_endpgm" not in r.stdout:
                    raise RuntimeError(f"llvm-objdump -d {f} (from {obj_path}): rc {r.returncode}, no gfx950 code in the output")
                n += len(re.findall(r"\bv_pk_(?:fma|add|mul)_f32\b", r.stdout))
        if not seen:       # a tool or naming change must not turn the gate into a pass; an object without device code (dd_tp.hip: host only) may
            h = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-h", "x.o"], cwd=d, capture_output=True, text=True, timeout=120)
            if h.returncode == 0 and ".hip_fatbin" not in h.stdout:
                return 0
            raise RuntimeError(f"no gfx950 code object extracted from {obj_path} (llvm-objdump --offloading produced: {sorted(os.listdir(d))})")
        return n
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _check_packed_fp32(objs) -> None:
    bad = [(os.path.basename(o), n) for o in objs for n in [packed_fp32_ops(o)] if n]
    if bad:
        raise RuntimeError("product objects contain packed FP32 VALU instructions (see NO_PACKED_FP32 in dropoutdecoding_amd/build.py): " + str(bad))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the sources whose content (or a header's) changed and link.  One builder at a time (file lock), a bounded number of
    hipcc jobs, objects and stamps written to temporaries and renamed."""
    if not force and not _stale():
        return LIB
    if os.environ.get("DD_NO_BUILD", "0") not in ("", "0"):
        # from here on a compile WOULD start (stale tree, or force=True on a fresh one): a process that may not spawn hipcc stops here
        require_fresh()
        raise RuntimeError("build(force=True) under DD_NO_BUILD / --no-build: this process may not compile (run "
                           "`python3 -m dropoutdecoding_amd.build` from a process that has not touched the GPU)")
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    with open(os.path.join(bdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)             # spawned ranks / test workers after a source change: the second one waits
        if not force and not _stale():
            return LIB
        return _build_locked(force, verbose, bdir)


def _build_locked(force: bool, verbose: bool, bdir: str) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, tools_objs, todo = [], [], []
    units = [(s, [] if s in TOOLS_SOURCES else NO_PACKED_FP32, s.replace(".hip", ".o"), False) for s in SOURCES + TOOLS_SOURCES]
    units += [(s, NO_PACKED_FP32 + extra, s.replace(".hip", "_tools.o"), True) for s, extra in TOOLS_VARIANTS.items()]
    for s, extra, oname, is_variant in units:
        o = os.path.join(bdir, oname)
        if is_variant:
            pass                                          # variant objects: picked per library below
        elif s in TOOLS_SOURCES:
            tools_objs.append(o)
        else:
            objs.append(o)
        want = _file_hash([os.path.join(CSRC, s)] + _header_paths()) + "|" + " ".join(FLAGS + extra)
        stamp = o + ".srchash"
        if not force and os.path.exists(o) and os.path.exists(stamp) and os.path.exists(o + ".remarks") and open(stamp).read().strip() == want:
            continue
        todo.append((s, extra, o, stamp, want))
    max_jobs = max(1, min(len(todo), int(os.environ.get("DD_BUILD_JOBS", "0")) or max(1, (os.cpu_count() or 2) // 2)))
    running, failed = [], None

    def reap(block: bool):
        nonlocal failed
        for item in list(running):
            pr, cmd, o, stamp, want, tmp_o, err_path, err_f = item
            rc = pr.wait() if block else pr.poll()
            if rc is None:
                continue
            running.remove(item)
            err_f.close()
            text = open(err_path, errors="replace").read()
            lines = text.splitlines()
            other = []                                # warnings / errors with the lines that follow them; the resource remarks stay in the file
            for i, ln in enumerate(lines):
                if re.search(r"(warning|error|fatal error): ", ln) and "[-Rpass-analysis" not in ln:
                    other.extend(lines[i:i + 4])
            if rc != 0:
                failed = failed or subprocess.CalledProcessError(rc, cmd, stderr="\n".join(other[-40:]))
                sys.stderr.write("\n".join(other[-40:]) + "\n")
                for leftover in (tmp_o, err_path):        # a failed unit leaves nothing behind
                    try:
                        os.unlink(leftover)
                    except OSError:
                        pass
                continue
            if other and verbose:
                sys.stderr.write("\n".join(other) + "\n")
            os.replace(tmp_o, o)
            os.replace(err_path, o + ".remarks")
            with open(stamp + f".tmp{os.getpid()}", "w") as f:
                f.write(want)
            os.replace(stamp + f".tmp{os.getpid()}", stamp)

    for s, extra, o, stamp, want in todo:
        while len(running) >= max_jobs and failed is None:
            reap(block=False)
            if len(running) >= max_jobs:
                running[0][0].wait()
        if failed is not None:
            break
        tmp_o = o + f".tmp{os.getpid()}"
        err_path = o + f".err{os.getpid()}"
        cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, *extra, "-c", os.path.join(CSRC, s), "-o", tmp_o]
        if verbose:
            print(" ".join(cmd))
        err_f = open(err_path, "w")
        # its own session = its own process group: hipcc is a driver whose clang / lld children outlive a kill() of the driver alone
        running.append((subprocess.Popen(cmd, stderr=err_f, start_new_session=True), cmd, o, stamp, want, tmp_o, err_path, err_f))
    if failed is not None:                             # one unit failed: the others are not worth waiting for
        for item in running:
            try:
                os.killpg(item[0].pid, signal.SIGKILL)   # the whole group, so that no child keeps writing the temporaries reap() unlinks
            except (ProcessLookupError, PermissionError):
                item[0].kill()
    while running:
        reap(block=True)
    if failed is not None:
        raise failed
    _check_scratch(objs)
    _check_packed_fp32(objs)
    variant_of = {os.path.join(bdir, s.replace(".hip", ".o")): os.path.join(bdir, s.replace(".hip", "_tools.o")) for s in TOOLS_VARIANTS}
    tools_members = [variant_of.get(o, o) for o in objs] + tools_objs
    for lib, members in ((LIB, objs), (TOOLS_LIB, tools_members)):
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
    if "--resources" in sys.argv:
        for obj, ks in kernel_resources().items():
            for k in ks:
                if k["scratch"]:
                    print(f"{obj}: {_demangle(k['name'])}: scratch {k['scratch']} B/lane")
