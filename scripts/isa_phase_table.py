#!/usr/bin/env python
"""Per-phase instruction table of the fast encoder's round loop from the ISA listing.

Builds the kernels with -DK4_PHASE_MARKS (k4lz4_common.hpp: K4_PHASE leaves a comment in the listing and pins the phases' order in
that build), then, per kernel, counts the instructions laid out between two markers: total, VALU, SALU, LDS, VMEM, branches,
and spill traffic (v_readlane / v_writelane on the spill register, scratch_*).  These are STATIC counts of the code as laid
out -- a phase's loops (the hop chain, the group loop, flush's rare paths) run a data-dependent number of times -- so the table is
read next to the dynamic figures: SQ_INSTS_* per launch / rounds per launch (scripts/pmc_summary.py, profiles/*_phase_probe.txt).
Usage: python scripts/isa_phase_table.py [kernel-substring ...]   (default: the two bench encoder kernels)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "k4os", "compression", "lz4_amd", "csrc", "k4lz4_capi.hip")
want = sys.argv[1:] or ["k4_encode_fast_kernelE", "k4_encode_fast_gtab_kernelE"]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "dev.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-DK4_PHASE_MARKS",
                           "-S", SRC, "-o", out], stderr=subprocess.DEVNULL)
    txt = open(out).read()
is_inst = re.compile(r"\s+(s_|v_|ds_|global_|buffer_|flat_|scratch_)")
for f in re.split(r"\n\s*\.globl\s+", txt)[1:]:
    name = f.split("\n", 1)[0].strip()
    if not any(w in name for w in want):
        continue
    lines = f.split("\n")
    end = next(i for i, l in enumerate(lines) if l.startswith(".Lfunc_end"))
    spillv = set(re.findall(r"v_writelane_b32\s+(v\d+)", "\n".join(lines[:end])))
    # the kernel holds one copy of the round loop per table type (byU16 / byU32+hash5 / byU32+hash4: compress_fast_block picks one
    # per block); the copies are told apart by their "load" markers, in the order the compiler laid them out
    phase, order, acc, inst = "(before the round loop)", [], {}, 0
    for l in lines[:end]:
        m = re.search(r"; k4phase (\S+)", l)
        if m:
            if m.group(1) in ("load", "front"):
                inst += 1
            phase = f"{m.group(1)} #{inst}" if inst else m.group(1)
            continue
        if not is_inst.match(l):
            continue
        op = l.split()[0]
        a = acc.setdefault(phase, dict(total=0, valu=0, salu=0, lds=0, vmem=0, branch=0, spill=0))
        if phase not in order:
            order.append(phase)
        a["total"] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            a["branch"] += 1
        if op.startswith("v_"):
            a["valu"] += 1
        elif op.startswith("s_"):
            a["salu"] += 1
        elif op.startswith("ds_"):
            a["lds"] += 1
        else:
            a["vmem"] += 1
        if op.startswith("scratch_") or (op in ("v_readlane_b32", "v_writelane_b32") and any(re.search(r"\b%s\b" % v, l) for v in spillv)):
            a["spill"] += 1
    print(f"== {name.split('E')[0].replace('_ZN2k4', '')[2:]}  (static instruction counts per phase of the round loop, as laid out)")
    print(f"{'phase':26s} {'total':>6s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'branch':>6s} {'spill ld/st':>11s}")
    for ph in order:
        a = acc[ph]
        print(f"{ph:26s} {a['total']:6d} {a['valu']:6d} {a['salu']:6d} {a['lds']:5d} {a['vmem']:5d} {a['branch']:6d} {a['spill']:11d}")
    for k in range(1, inst + 1):
        loop = [p for p in order if p.endswith(f" #{k}") and not p.startswith("round-end")]
        print(f"{'round loop, copy #%d' % k:26s} {sum(acc[p]['total'] for p in loop):6d} {sum(acc[p]['valu'] for p in loop):6d} {sum(acc[p]['salu'] for p in loop):6d}"
              f" {sum(acc[p]['lds'] for p in loop):5d} {sum(acc[p]['vmem'] for p in loop):5d} {sum(acc[p]['branch'] for p in loop):6d} {sum(acc[p]['spill'] for p in loop):11d}")
