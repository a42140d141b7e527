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
        objdir = os.path.join(PKG_DIR, "build", f"obj.{os.getpid()}")
        os.makedirs(objdir, exist_ok=True)
        flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-pthread", "-ffp-contract=off",
                 "-I" + os.path.join(REPO_ROOT, "include"), "-I" + CSRC]
        flags[0:0] = os.environ.get("NB_EXTRA_HIPCC_FLAGS", "").split()   # e.g. -DNB_ANCHOR_PROF for tuning experiments
        objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in srcs]

        def compile_one(pair):
            src, obj = pair
            cmd = [_hipcc()] + flags + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)

        try:
            # the translation units are independent (one hipcc each, side by side: the three kernel files take minutes), then one link
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1, 8)) as pool:
                list(pool.map(compile_one, zip(srcs, objs)))
            cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-pthread", "-o", tmp] + objs + ["-lz"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
            os.replace(tmp, LIB_PATH)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
            for o in objs:
                if os.path.exists(o):
                    os.remove(o)
            try:
                os.rmdir(objdir)
            except OSError:
                pass
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
