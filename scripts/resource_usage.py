#!/usr/bin/env python
"""Registers, spills, scratch, occupancy and LDS of every kernel of the library as the compiler reports them
(-Rpass-analysis=kernel-resource-usage): python scripts/resource_usage.py > profiles/rNN_resource_usage.txt"""
import os, re, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "k4os", "compression", "lz4_amd", "csrc", "k4lz4_capi.hip")
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        "-c", SRC, "-o", os.path.join(d, "o.o")], capture_output=True, text=True)
t = r.stderr
for m in re.finditer(r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)", t, re.S):
    n = re.sub(r"^_ZN2k4\d+", "", m.group(1))
    n = re.sub(r"E(NS_|P|j|i|x).*$", "", n)
    print(f"{n:36s} VGPR {m.group(3):>3} AGPR {m.group(4):>3} SGPR {m.group(2):>3} sspill {m.group(7):>3} vspill {m.group(8):>2} scratch {m.group(5):>3} occ {m.group(6)} lds {m.group(9)}")
