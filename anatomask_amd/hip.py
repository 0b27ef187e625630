"""ctypes binding of libanatomask_hip.so (the C ABI declared in include/anatomask_hip.h).

The HIP library is the product path: if it is missing this module raises at import of the
symbol table -- there is no CPU or torch fallback anywhere in the package."""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libanatomask_hip.so")      # the in-tree build; no environment variable redirects the product loader
HEADER = os.path.join(os.path.dirname(_HERE), "include", "anatomask_hip.h")

DT_F32, DT_BF16, DT_F32S = 0, 1, 2
CONV_FWD, CONV_DGRAD, CONVT_FWD, CONVT_DGRAD = 0, 1, 2, 3
ACT_NONE, ACT_LRELU, ACT_RELU6 = 0, 1, 2

_CT = {"int": C.c_int, "long": C.c_long, "float": C.c_float, "double": C.c_double}


def declared_functions(header: str = HEADER):
    """Parse `int am_xxx(args);` prototypes from the header -> {name: [ctypes argtypes]}."""
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\bint\s+(am_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    types.append(C.c_void_p)
                else:
                    types.append(_CT[a.replace("const ", "").split(" ")[0]])
        out[name] = types
    return out


class HipLib:
    ANSWERS = ("am_head_stencil_supported", "am_conv3d_wgrad_uses_k3", "am_conv3d_prenorm_supported", "am_version")     # entry points whose return value is an answer, not an error code: call them on ._lib

    def __init__(self, path: str = LIB_PATH):
        import torch  # noqa: F401  torch's bundled HIP runtime must be the one in the process (same soname as /opt/rocm's):
        # loading our library first would bring in a second runtime that knows nothing of torch's streams/allocations
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not built: run `python -m anatomask_amd.build` (no fallback path exists)")
        if os.path.abspath(path) == LIB_PATH:
            # the product library must be the build of the sources lying next to it (build.py writes their digest beside the .so)
            from . import build as _b
            sp = _b.stamp_path(path)
            have = open(sp).read().strip() if os.path.exists(sp) else None
            if have != _b.source_digest():
                raise RuntimeError(f"{path} was not built from the csrc/ + include/ sources next to it (stamp {str(have)[:12]}.. != sources "
                                   f"{_b.source_digest()[:12]}..): run `python -m anatomask_amd.build`")
        self._lib = C.CDLL(path)
        self.functions = declared_functions()
        for name, argtypes in self.functions.items():
            fn = getattr(self._lib, name)       # AttributeError if the .so lacks a declared symbol
            fn.argtypes = argtypes
            fn.restype = C.c_int
            if name not in self.ANSWERS:
                setattr(self, name[3:], self._wrap(name, fn))

    @staticmethod
    def _wrap(name, fn):
        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise RuntimeError(f"{name} failed with code {rc}")
        call.__name__ = name
        return call


_LIB = None


def lib() -> HipLib:
    global _LIB
    if _LIB is None:
        _LIB = HipLib()
    return _LIB


def use_library(path: str) -> HipLib:
    """tools/ only (timing-ablation builds, A/B of alternative builds): bind an explicitly named library instead of the in-tree
    one.  Must be called before the first kernel call of the process."""
    global _LIB
    if _LIB is not None:
        raise RuntimeError("use_library() after the library was loaded")
    _LIB = HipLib(path)
    return _LIB
