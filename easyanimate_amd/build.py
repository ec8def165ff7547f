"""Builds libea_mi355x.so (HIP kernels + C ABI) for gfx950 with hipcc.  No GPU is needed to build.

    python -m easyanimate_amd.build [--force]
    EA_BUILD_VARIANTS=1 EA_LIB_OUT=easyanimate_amd/lib/variants/libea_variants.so python -m easyanimate_amd.build
        # a second library beside the product one, with the cross-check / experiment kernels compiled in; load it with
        # EA_LIB_PATH=<that file> (tests: the EA_BUILD_VARIANTS-only cases run; tools/ab_*.py: in-process A/Bs)
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libea_mi355x.so")
_OUT = os.environ.get("EA_LIB_OUT")     # build another library file (own object directory and stamp); the product library stays
if _OUT:
    LIB = os.path.abspath(_OUT)
    LIBDIR = os.path.dirname(LIB)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# per-file extras: the one-wave-per-SIMD attention kernel keeps its MFMA accumulators in arch VGPRs (they are read by
# the softmax VALU code every block); without this hipcc parks them in AGPRs and copies ~340 registers per tile.
FLAGS += os.environ.get("EA_HIPCC_EXTRA", "").split()   # diagnostic builds, e.g. EA_HIPCC_EXTRA=-DEA_GEMM_TIMESTAMPS
if os.environ.get("EA_BUILD_VARIANTS", "0") not in ("", "0"):
    # also compile the superseded kernel generations (attention v1, 32x32x16 row-slab conv, four-wave GEMM) as cross-checks
    FLAGS += ["-DEA_BUILD_VARIANTS=1"]
# -fno-slp-vectorize: the attention main loop is scheduled by hand in source (fenced groups); the SLP vectoriser would
# re-pair its adds / packs across the fences' dataflow and spill.
EXTRA_FLAGS = {"ea_attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/ea_mi355x.h"]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def _compile(src: str, objdir: str) -> str:
    obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
    cmd = [HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.sha256") if not _OUT else LIB + ".sha256"
    dg = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dg:
        return LIB
    objdir = os.path.join(HERE, "build") if not _OUT else os.path.join(HERE, "build", "out_" + os.path.basename(LIB))
    os.makedirs(objdir, exist_ok=True)
    srcs = _sources()
    if verbose:
        print(f"[easyanimate_amd.build] hipcc {' '.join(FLAGS)}: {', '.join(srcs)}", file=sys.stderr)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, objdir), srcs))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    open(stamp, "w").write(dg + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
