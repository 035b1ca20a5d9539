"""Builds libshgvqa.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container as well as on the GPU box.
Objects are cached under shg_vqa_amd/csrc/_obj and rebuilt when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libshgvqa.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-result"]
# per-file extras.  attention.hip: the softmax works on the MFMA results with VALU instructions every key tile
# (scale, max, exp, rescale of the output accumulators); with the accumulators in AGPRs (the compiler's default
# at this register count) each touch is a v_accvgpr_read/write: ~80 extra VALU instructions per tile in a
# VALU-bound loop.  VGPR-form MFMAs remove them and raise the occupancy of the backward kernels.
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "shg_vqa.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), _headers_mtime()):
        return obj
    cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + ["-c", path, "-o", obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, res.stderr[-4000:]))
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n%s" % res.stderr[-4000:])
    _build_fastcall()
    if verbose:
        print("built", LIB)
    return LIB


def _build_fastcall():
    """Host-side CPython extension (csrc_host/_fastcall.c): low-overhead trampoline into the C ABI; optional - without
    it _lib.call goes through ctypes."""
    import sysconfig
    src = os.path.join(HERE, "csrc_host", "_fastcall.c")
    out = os.path.join(HERE, "_fastcall" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    inc = sysconfig.get_paths()["include"]
    if not os.path.exists(os.path.join(inc, "Python.h")):
        return None
    res = subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-I" + inc, "-o", out, src], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("gcc failed for _fastcall.c:\n%s" % res.stderr[-2000:])
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
