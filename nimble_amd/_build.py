"""Build the in-tree native library (hipcc, gfx950 only). Used by __graft_entry__.build() and by developers."""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libnimble_amd.so")
SOURCES = ["nb_kernels.hip", "nb_lane.hip", "nb_anchor.hip", "nb_index.cpp", "nb_ingest.cpp", "nb_inflate.cpp", "nb_report.cpp", "nb_host.cpp", "nb_bam.cpp"]
HEADERS = [os.path.join(CSRC, "nb_internal.h"), os.path.join(CSRC, "nb_inflate.h"), os.path.join(CSRC, "nb_wave.h"), os.path.join(CSRC, "nb_device.h"), os.path.join(REPO_ROOT, "include", "nimble_amd.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile nimble_amd/csrc/* into nimble_amd/libnimble_amd.so for gfx950. hipcc cross-compiles without a GPU."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl
    # several ranks of one job may get here together (torchrun, one process per GPU): one of them builds, the others wait
    # for the lock and then find the library up to date. The output is written beside the target and renamed into place.
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return LIB_PATH
        srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
        tmp = LIB_PATH + f".{os.getpid()}.tmp"
        cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", "-ffp-contract=off",
               "-I" + os.path.join(REPO_ROOT, "include"), "-I" + CSRC, "-o", tmp] + srcs + ["-lz"]
        cmd[1:1] = os.environ.get("NB_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DNB_SE_WAVES_PER_EU=8 for tuning experiments
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        try:
            subprocess.run(cmd, check=True)
            os.replace(tmp, LIB_PATH)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
