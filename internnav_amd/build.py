"""In-tree build of libinternnav_amd.so for gfx950 with hipcc (no torch, no cmake).

`python -m internnav_amd.build` (or `__graft_entry__.build()`) compiles every translation unit under
`internnav_amd/csrc/` into objects and links one shared library next to this file. Objects are rebuilt only
when a source or header is newer, so repeated calls are cheap. hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = CSRC / "build"
LIB = ROOT / "libinternnav_amd.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast"]


def _sources():
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _headers():
    return list(CSRC.glob("*.h")) + list((ROOT.parent / "include").glob("*.h"))


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def _compile(src: Path, verbose: bool) -> Path:
    obj = OBJ / (src.name + ".o")
    if _stale(obj, [src] + _headers()):
        cmd = ["hipcc", *FLAGS, "-x", "hip", "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
    return obj


def build(verbose: bool = True) -> Path:
    OBJ.mkdir(parents=True, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    if _stale(LIB, objs):
        cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *map(str, objs), "-o", str(LIB)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build())
