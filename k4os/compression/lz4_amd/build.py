"""Builds libk4lz4.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

The shared object lives next to this file so that it travels with the repository snapshot and is
the library the Python host layer (and a .NET host through P/Invoke) loads."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libk4lz4.so")
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libk4lz4.so)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) +
                  [os.path.join(PKG_DIR, "..", "..", "..", "include", "k4lz4.h")])


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in sources() if os.path.exists(s))


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           os.path.join(CSRC, "k4lz4_capi.hip"), "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
