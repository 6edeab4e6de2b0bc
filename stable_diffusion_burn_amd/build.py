"""Build libsdmi.so (HIP kernels + C++ engine + C ABI) for gfx950, in-tree.

    python -m stable_diffusion_burn_amd.build [--force]

hipcc cross-compiles gfx950 code objects without a GPU.  The shared library
lands in stable_diffusion_burn_amd/lib/libsdmi.so (git-ignored, but shipped
to the GPU box by gpurun).  No torch, no cmake: four kernel translation units
and two host ones, compiled in parallel and linked with hipcc.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libsdmi.so"
OBJDIR = PKG / "build"

SOURCES = ["k_gemm.hip", "k_gemm2.hip", "k_attn.hip", "k_norm.hip", "k_elem.hip", "engine.cpp", "sdmi_capi.cpp"]
HEADERS = ["kernels.hpp", "engine.hpp", "../../include/sdmi.h"]
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest() -> str:
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        h.update(name.encode())
        h.update((CSRC / name).read_bytes())
    h.update(" ".join(CXXFLAGS).encode())
    return h.hexdigest()


def _compile(src: str) -> Path:
    obj = OBJDIR / (src.replace("/", "_") + ".o")
    cmd = [hipcc(), *CXXFLAGS, "-c", str(CSRC / src), "-o", str(obj)]
    if src.endswith(".cpp"):
        cmd[1:1] = ["-x", "hip"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = True) -> Path:
    LIBDIR.mkdir(exist_ok=True)
    OBJDIR.mkdir(exist_ok=True)
    stamp = LIBDIR / "libsdmi.sha256"
    digest = _digest()
    if not force and LIB.exists() and stamp.exists() and stamp.read_text().strip() == digest:
        if verbose:
            print(f"[sdmi build] up to date: {LIB}")
        return LIB
    if verbose:
        print(f"[sdmi build] compiling {len(SOURCES)} translation units for {ARCH} ...")
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(digest + "\n")
    if verbose:
        print(f"[sdmi build] wrote {LIB} ({LIB.stat().st_size / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
