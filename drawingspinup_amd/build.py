"""Build libdsu_hip.so (gfx950) in-tree with hipcc.

    python -m drawingspinup_amd.build [--force]
    python -m drawingspinup_amd.build --variant NAME -DFOO=1 [-DBAR ...]

The second form compiles every source with extra preprocessor definitions into
drawingspinup_amd/variants/libdsu_hip_NAME.so (objects under csrc/_obj/NAME/); a process started
with DSU_HIP_LIB=<that path> loads it instead of the default library.  Several variants can thus
be measured in ONE visit to a GPU box (build here, run `DSU_HIP_LIB=... python tools/...` there).

The library has no torch / pybind dependency: it is a plain C-ABI shared object
(include/dsu_hip.h) loaded through ctypes by drawingspinup_amd._lib.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libdsu_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics",
         "-ffp-contract=off", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(src, extra=()):
    h = hashlib.sha1()
    h.update(" ".join([*FLAGS, *extra]).encode())
    for p in [src] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
            [os.path.join(HERE, "..", "include", "dsu_hip.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _compile(src, force, extra=(), objdir=None):
    obj = os.path.join(objdir or OBJ, os.path.basename(src)[:-4] + ".o")
    stamp_file = obj + ".stamp"
    stamp = _stamp(src, extra)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file) \
            and open(stamp_file).read() == stamp:
        return obj, False
    cmd = [_hipcc(), *FLAGS, *extra, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[dsu build] linked {LIB}")
    elif verbose:
        print(f"[dsu build] up to date: {LIB}")
    return LIB


def build_variant(name, defines, verbose=True):
    """Every source compiled with the extra -D... flags -> variants/libdsu_hip_<name>.so."""
    if not name.replace("_", "").isalnum():
        raise ValueError("variant name: letters, digits, underscore")
    objdir = os.path.join(OBJ, name)
    os.makedirs(objdir, exist_ok=True)
    outdir = os.path.join(HERE, "variants")
    os.makedirs(outdir, exist_ok=True)
    lib = os.path.join(outdir, f"libdsu_hip_{name}.so")
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = [o for o, _ in ex.map(lambda s: _compile(s, False, tuple(defines), objdir), srcs)]
    r = subprocess.run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[dsu build] variant {name} ({' '.join(defines)}): {lib}")
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], [a for a in sys.argv[1:] if a.startswith("-D")])
    else:
        build(force="--force" in sys.argv)
