"""Build libpresight_hip.so (HIP, gfx950 only) in-tree with hipcc.  No torch headers are involved:
the library exposes a plain C ABI (include/presight_hip.h) and is loaded through ctypes."""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpresight_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
         "-Wno-unused-result", "-I" + CSRC]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp"))


def _headers_mtime():
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".hpp"))


def _compile(src):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    spath = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(spath), _headers_mtime()):
        return obj, False
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    return obj, True


def build(verbose: bool = True, jobs: int = 4) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    rebuilt = any(ch for _, ch in results)
    if rebuilt or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[presight_amd] built {LIB} from {len(objs)} objects", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build())
