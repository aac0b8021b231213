"""Builds libegot2x.so (HIP, gfx950) in-tree with hipcc. No torch types cross this boundary."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(CSRC, "_obj")
LIB_PATH = os.path.join(PKG_DIR, "libegot2x.so")
SOURCES = ["gemm.hip", "norm.hip", "attention.hip", "fused.hip", "fused_bwd.hip", "ffn_cut.hip", "tiled_attn.hip", "feature_sink.hip", "train.hip", "decoder.hip", "wide_gemm.hip", "wide_attn.hip", "wide_rows.hip", "wide_host.hip", "wide_decoder.hip", "comm.hip", "encoder.hip"]
HEADERS = ["common.h", "kernels.h", "fused.h", "fused_dev.h", "wide.h", "wide_host.h", os.path.join("..", "..", "include", "egot2x.h")]
EXTRA = os.environ.get("EGX_CXXFLAGS", "").split()
FLAGS = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libegot2x.so cannot be built")
    return exe


def _digest(paths) -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False, variant: str = "", extra_flags=()) -> str:
    """Compile every HIP translation unit for gfx950 and link libegot2x.so. Returns the library path.
    `variant` (development aid): build egot2_amd/_variants/lib_<variant>.so with `extra_flags` added, next to the product
    library (objects in csrc/_obj/<variant>/); load it with EGX_LIB=<path>."""
    global FLAGS
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    OBJ_DIR = os.path.join(CSRC, "_obj", variant) if variant else os.path.join(CSRC, "_obj")
    LIB_PATH = os.path.join(PKG_DIR, "_variants", f"lib_{variant}.so") if variant else os.path.join(PKG_DIR, "libegot2x.so")
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    base_flags = FLAGS
    if extra_flags:
        FLAGS = list(extra_flags) + FLAGS
    try:
        return _build(srcs, OBJ_DIR, LIB_PATH, force, verbose)
    finally:
        FLAGS = base_flags


def _build(srcs, OBJ_DIR, LIB_PATH, force, verbose) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_paths = [os.path.join(CSRC, h) for h in HEADERS]
    hipcc = _hipcc()
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        stamp = obj + ".sha"
        dg = _digest([src] + hdr_paths)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg:
            continue
        jobs.append((src, obj, stamp, dg))

    def compile_one(job):
        src, obj, stamp, dg = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(dg)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or force or not os.path.exists(LIB_PATH):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    # python -m egot2_amd.build [--force] [--variant NAME -DFLAG ...]
    argv = sys.argv[1:]
    variant = argv[argv.index("--variant") + 1] if "--variant" in argv else ""
    extra = [a for a in argv if a.startswith("-D") or a.startswith("-m") or a.startswith("-f")]
    print(build(force="--force" in argv, verbose=True, variant=variant, extra_flags=extra))
