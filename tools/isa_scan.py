#!/usr/bin/env python3
"""Scan the gfx950 ISA of the library's kernels for MFMA loops the compiler has damaged.

r4 found the sparse grouped kernels at 0.59 of the MFMA peak because their tap / chunk cursors BRANCHED inside the K loop: the loop was
cut into basic blocks of 8 MFMAs and the register allocator copied both accumulator tiles in and out of the MFMA registers at every
block boundary (59 v_mov + a drain of the matrix pipe per 8 MFMAs).  This tool compiles every csrc/*.hip to assembly (hipcc cross-compiles,
no GPU needed) and prints, per kernel, the basic blocks that hold MFMAs: (mfma, other instructions, v_mov / v_accvgpr copies, VALU, LDS,
VMEM, SALU, s_nop) -- blocks with as many copies as MFMAs, or K loops split into many small MFMA blocks, are what to look for.

    python tools/isa_scan.py [file.hip ...]        # default: every file of partner_amd/csrc
"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "partner_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
    except OSError:
        return n


for f in files:
    if os.path.basename(f) == "pn_common.hip":
        continue
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        r = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-S", "-o", tmp.name,
                            f, "--cuda-device-only"], capture_output=True, text=True)
        if r.returncode:
            print(f"{os.path.basename(f)}: compile failed"); continue
        s = open(tmp.name).read()
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end\d+:", s, re.S):
        name, body = m.group(1), m.group(2)
        if body.count("v_mfma") < 8:
            continue
        rows = []
        for b in re.split(r"\n(?=\.LBB\d+_\d+:)", body):
            ins = [l.strip() for l in b.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
            nm = sum("v_mfma" in i for i in ins)
            if nm:
                cnt = lambda p: sum(i.startswith(p) for i in ins)
                rows.append((nm, len(ins) - nm, cnt("v_mov") + cnt("v_accvgpr"), sum(i.startswith("v_") and "mfma" not in i for i in ins), cnt("ds_"),
                             cnt("buffer") + cnt("global"), cnt("s_") - cnt("s_nop"), cnt("s_nop")))
        flag = "  <-- accumulator copies?" if any(r[2] >= r[0] and r[0] >= 4 for r in rows) else ""
        print(f"{os.path.basename(f)}: {demangle(name)[:80]}: {len(rows)} MFMA block(s) (mfma, other, copies, valu, lds, vmem, salu, nop) {sorted(rows, reverse=True)[:4]}{flag}")
