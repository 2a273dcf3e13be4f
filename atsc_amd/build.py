"""Builds libatsc_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# ATSC_BUILD_VARIANT=<name> ATSC_BUILD_DEFS="-DX ..." builds a side-by-side dev library libatsc_hip_<name>.so
# (objects under build_<name>/), loaded by atsc_amd.capi when ATSC_LIB_VARIANT=<name> (tools/stamp_probe.py)
VARIANT = os.environ.get("ATSC_BUILD_VARIANT", "")
LIB = os.path.join(HERE, "libatsc_hip%s.so" % ("_" + VARIANT if VARIANT else ""))
SOURCES = ["atsc_kernels.hip", "atsc_large.hip", "atsc_decode.hip", "atsc_host.cpp", "atsc_stream.cpp",
           "atsc_vsri.cpp"]
CLI = os.path.join(HERE, "bin", "atsc")
CLI_SRC = "atsc_cli.cpp"
CLI2 = os.path.join(HERE, "bin", "csv-compressor")
CLI2_SRC = "csv_compressor_cli.cpp"
DEPS = SOURCES + ["atsc_device.h", "atsc_internal.h", "atsc_large_cols.h", "atsc_large_fast.h",
                  os.path.join("..", "..", "include", "atsc_hip.h")]
# -ffp-contract=off: the f64 spline / rounding arithmetic must evaluate exactly as written
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libatsc_hip.so")


def stale():
    if VARIANT:
        return True
    if not os.path.exists(LIB) or not os.path.exists(CLI) or not os.path.exists(CLI2):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(CLI), os.path.getmtime(CLI2))
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS + [CLI_SRC, CLI2_SRC])


OBJDIR = os.path.join(HERE, "build" + ("_" + VARIANT if VARIANT else ""))
HEADERS = ["atsc_device.h", "atsc_internal.h", "atsc_large_cols.h", "atsc_large_fast.h",
           os.path.join("..", "..", "include", "atsc_hip.h")]
CFLAGS = [f for f in FLAGS if f != "-shared"] + os.environ.get("ATSC_BUILD_DEFS", "").split()


# per-source additions to CFLAGS.  atsc_kernels.hip: no machine-level loop-invariant code motion -- in kernels this
# long it hoists dozens of constant materialisations out of the frame's loops and keeps them in registers across
# everything else (k_compress<1,5,false,256>: 32 -> 16 spilled SGPRs; 1-2 % on every frame length; the resident-workgroup
# experiment does not fit its registers without it).  ATSC_BUILD_MLICM=1 builds with the compiler's default.
FILE_FLAGS = {}
if not os.environ.get("ATSC_BUILD_MLICM"):
    FILE_FLAGS["atsc_kernels.hip"] = ["-mllvm", "-disable-machine-licm"]
for _f in os.environ.get("ATSC_BUILD_NOMLICM_ALSO", "").split(","):  # (A/B aid: the same for other sources)
    if _f:
        FILE_FLAGS[_f] = ["-mllvm", "-disable-machine-licm"]


def _flags_key(src=None):
    return " ".join(CFLAGS + FILE_FLAGS.get(src, []))


def _obj_stale(src, obj):
    """An object is stale when a source or header is newer than it, or when it was compiled with other flags
    (ATSC_BUILD_DEFS of a dev variant, a changed FLAGS): the flag string is kept next to the object."""
    if not os.path.exists(obj):
        return True
    try:
        if open(obj + ".flags").read() != _flags_key(src):
            return True
    except OSError:
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in [src] + HEADERS)


def _compile_objects(force, verbose):
    """One object per source (kept under atsc_amd/build/, git-ignored), compiled side by side: the three
    kernel files take about a minute each, so an edit to one of them costs one of them."""
    os.makedirs(OBJDIR, exist_ok=True)
    jobs, objs = [], []
    for s in SOURCES:
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _obj_stale(s, obj):
            cmd = [_hipcc()] + CFLAGS + FILE_FLAGS.get(s, []) + ["-c", "-o", obj, os.path.join(CSRC, s)]
            if verbose:
                print(" ".join(cmd))
            jobs.append((s, obj, subprocess.Popen(cmd, cwd=CSRC)))
    bad = []
    for s, obj, p in jobs:
        if p.wait() != 0:
            bad.append(s)
        else:
            with open(obj + ".flags", "w") as f:
                f.write(_flags_key(s))
    if bad:
        raise RuntimeError("hipcc failed on " + ", ".join(bad))
    return objs


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    objs = _compile_objects(force, verbose)
    cmd = [_hipcc()] + [f for f in FLAGS if f != "-ffp-contract=off"] + ["-o", LIB] + objs  # the compile flags, -shared included
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    if VARIANT:
        return LIB
    # the `atsc` command line front end (plain C++ over the C ABI)
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cli = [_hipcc(), "-O2", "-std=c++17", "-o", CLI, os.path.join(CSRC, CLI_SRC), "-L" + HERE, "-latsc_hip",
           "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cli))
    subprocess.check_call(cli, cwd=CSRC)
    # the `csv-compressor` front end (csv-compressor/src/main.rs)
    cli2 = [_hipcc(), "-O2", "-std=c++17", "-o", CLI2, os.path.join(CSRC, CLI2_SRC), "-L" + HERE, "-latsc_hip",
            "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cli2))
    subprocess.check_call(cli2, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
    print(LIB)
