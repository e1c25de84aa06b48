"""ctypes binding of libpresight_hip.so.  Prototypes are read from include/presight_hip.h so that
the header is the single source of truth for the C ABI.  There is NO fallback: if the library is
missing or a call fails, an exception is raised."""
import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRESIGHT_HIP_LIB", os.path.join(_HERE, "libpresight_hip.so"))  # env override: ablation builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "presight_hip.h")


class PresightHipError(RuntimeError):
    pass


def _ctype(decl: str):
    d = re.sub(r"/\*.*?\*/", "", decl).strip()
    if "*" in d:
        return ctypes.c_void_p
    base = d.replace("const", "").split()
    ty = base[0]
    return {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "int32_t": ctypes.c_int32,
            "uint8_t": ctypes.c_uint8, "uint32_t": ctypes.c_uint32, "double": ctypes.c_double}[ty]


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[object, List[object]]]:
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(ps_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        restype = ctypes.c_char_p if "char" in ret else (ctypes.c_int64 if ret == "int64_t" else ctypes.c_int)
        protos[name] = (restype, argtypes)
    return protos


_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it brings its own copy of the HIP runtime, and the library must bind to THAT instance (the one that owns
    # torch's device context and streams); loaded the other way round the process ends up with two runtimes and every
    # launch from here fails with "no ROCm-capable device is detected"
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise PresightHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback for the hot path.")
    h = ctypes.CDLL(LIB_PATH)
    for name, (ret, argtypes) in parse_header().items():
        fn = getattr(h, name)  # AttributeError if the header declares a symbol the library lacks
        fn.restype = ret
        fn.argtypes = argtypes
    _lib = h
    return h


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().ps_last_error()
        raise PresightHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
