"""Build libanatomask_hip.so (gfx950) in-tree with hipcc.  `python -m anatomask_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libanatomask_hip.so")
AB_DIR = os.path.join(os.path.dirname(HERE), "build_ab")      # the tools-only -DAM_ABLATE variant and its objects: never next to the product library
SOURCES = ["conv_igemm.hip", "conv_k3.hip", "conv_rw.hip", "conv_gather.hip", "conv_wgrad.hip", "conv_wgk3.hip", "stream_ops.hip", "step_ops.hip", "head_ops.hip", "aug_ops.hip", "layer_ops.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-result"]


def source_digest(ablate: bool = False) -> str:
    """sha256 over everything the library is made of: the .hip sources, the three headers and the compiler flags.  `build` writes it to
    <library>.stamp; the loader (hip.HipLib) recomputes it from the sources lying next to the library and refuses a binary that was
    not built from them -- a prebuilt .so that travelled to a GPU box proves itself against the source that travelled with it."""
    import hashlib
    h = hashlib.sha256()
    flags = FLAGS + (["-DAM_ABLATE"] if ablate else [])
    h.update(" ".join(flags).encode())
    for f in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_plan.h"),
                                                          os.path.join(os.path.dirname(HERE), "include", "anatomask_hip.h")]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def stamp_path(lib_path: str) -> str:
    return lib_path + ".stamp"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, ablate: bool = False) -> str:
    """ablate=True: the tools-only variant build_ab/libanatomask_hip_ablate.so (-DAM_ABLATE: timing-ablation switches read from the
    environment, results wrong on purpose); the product library never contains them."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    sfx = "_ablate" if ablate else ""
    out = os.path.join(AB_DIR, "libanatomask_hip_ablate.so") if ablate else OUT
    objdir = AB_DIR if ablate else CSRC
    os.makedirs(objdir, exist_ok=True)
    flags = FLAGS + (["-DAM_ABLATE"] if ablate else [])
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_plan.h"), os.path.join(os.path.dirname(HERE), "include", "anatomask_hip.h")]
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", sfx + ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc, *flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    digest = source_digest(ablate)
    stamped = os.path.exists(stamp_path(out)) and open(stamp_path(out)).read().strip() == digest
    if force or procs or _stale(out, objs) or not stamped:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp_path(out), "w") as fh:
            fh.write(digest + "\n")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, ablate="--ablate" in sys.argv))
