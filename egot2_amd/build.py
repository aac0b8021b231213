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


# ---- host-side sanitizer build -------------------------------------------------------------------------
# The C++ orchestration (workspace layouts, configuration validation, error paths, the RCCL binding) runs on the host and can be
# exercised without a GPU. build_sanitized() recompiles exactly those translation units with AddressSanitizer + UBSan on the HOST side
# (-fno-gpu-sanitize: GPU ASAN / XNACK are not available on this pool and are never asked for) and links them with the ordinary
# objects of the rest; tests/test_cpu_host.py runs the host-only entry points against it in a subprocess with the ASAN runtime preloaded.
SANITIZED_SOURCES = ["encoder.hip", "wide_host.hip", "wide_decoder.hip", "comm.hip"]
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"]


def asan_runtime() -> str:
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not hits:
        raise RuntimeError("libclang_rt.asan-x86_64.so not found under /opt/rocm/lib/llvm")
    return hits[-1]


def build_sanitized(verbose: bool = False) -> str:
    build(verbose=verbose)                      # the ordinary objects of every other translation unit
    hipcc = _hipcc()
    obj_dir = os.path.join(CSRC, "_obj", "asan")
    lib_path = os.path.join(PKG_DIR, "_variants", "lib_asan.so")
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    hdr_paths = [os.path.join(CSRC, h) for h in HEADERS]
    flags = SAN_FLAGS + [f for f in FLAGS if f != "-O3"] + ["-O1"]
    objs, rebuilt = [], False
    for s in SOURCES:
        plain = os.path.join(CSRC, "_obj", s.replace(".hip", ".o"))
        if s not in SANITIZED_SOURCES:
            objs.append(plain)
            continue
        src, obj = os.path.join(CSRC, s), os.path.join(obj_dir, s.replace(".hip", ".o"))
        stamp, dg = obj + ".sha", _digest([src] + hdr_paths) + "asan"
        objs.append(obj)
        if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg:
            continue
        r = subprocess.run([hipcc] + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc (sanitized) failed for {src}:\n{r.stdout}\n{r.stderr}")
        open(stamp, "w").write(dg)
        rebuilt = True
    newest = max(os.path.getmtime(o) for o in objs)
    if rebuilt or not os.path.exists(lib_path) or os.path.getmtime(lib_path) < newest:
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", lib_path] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link (sanitized) failed:\n{r.stdout}\n{r.stderr}")
    return lib_path


if __name__ == "__main__":
    # python -m egot2_amd.build [--force] [--variant NAME -DFLAG ...]
    argv = sys.argv[1:]
    if "--sanitize" in argv:
        print(build_sanitized(verbose=True))
        sys.exit(0)
    variant = argv[argv.index("--variant") + 1] if "--variant" in argv else ""
    extra = [a for a in argv if a.startswith("-D") or a.startswith("-m") or a.startswith("-f")]
    print(build(force="--force" in argv, verbose=True, variant=variant, extra_flags=extra))
