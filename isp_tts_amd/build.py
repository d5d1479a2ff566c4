"""Builds libispk.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
`isp_tts_amd/libispk.so` travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libispk.so")
# the tools' build (tools/stamp_*.py, tools/sweep_gemm.py): the same sources with -DISPK_EXPERIMENTS, which compiles in the
# ISPK_* environment knobs, the in-kernel stamps and the ablation hooks.  Never loaded by the product (runtime.LIB_PATH).
LIB_EXP = os.path.join(HERE, "libispk_exp.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-file extras.  ffn2.hip: keep the GELU's fp32 arithmetic scalar - the SLP vectoriser would pack adjacent values into
# v_pk_*_f32, which issue slowly beside MFMAs (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
EXTRA_FLAGS = {"ffn2.hip": ["-fno-slp-vectorize"], "attention.hip": ["-fno-slp-vectorize"]}


def _sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "ispk.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, verbose: bool, experiments: bool = False) -> str:
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + (".exp.o" if experiments else ".o"))
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), _deps_mtime()):
        return obj
    cmd = [HIPCC, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), *(["-DISPK_EXPERIMENTS"] if experiments else []),
           "-c", src, "-o", obj]
    if verbose:
        cmd.insert(-4, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        sys.stderr.write(r.stderr)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}")
    return obj


def build_lib(force: bool = False, verbose: bool = False, experiments: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    lib = LIB_EXP if experiments else LIB
    if force:
        for f in os.listdir(OBJ):
            if f.endswith(".exp.o") == experiments:
                os.remove(os.path.join(OBJ, f))
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose, experiments), srcs))
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(o) for o in objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose="-v" in sys.argv, experiments="--experiments" in sys.argv))
