#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of the library, from the AMDGPU metadata hipcc writes with -S.

usage: python tools/kernel_resources.py [file.hip ...] [--json out.json]
Cross-compiles for gfx950 (no GPU needed), prints one line per kernel: VGPRs, AGPRs, spilled VGPRs / SGPRs, scratch
bytes per lane, static LDS, workgroup size.  tests/test_tools_cpu.py asserts the decode kernels stay free of scratch.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "point_slam_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "--cuda-device-only", "-S"]
KEYS = [".vgpr_count", ".agpr_count", ".vgpr_spill_count", ".sgpr_count", ".sgpr_spill_count",
        ".private_segment_fixed_size", ".group_segment_fixed_size", ".max_flat_workgroup_size"]


def parse_asm(text):
    """[{name, vgpr_count, ...}] from the amdhsa.kernels metadata block of a device assembly listing."""
    out = []
    m = re.search(r"amdhsa\.kernels:\n(.*?)\namdhsa\.target", text, re.S)
    if not m:
        return out
    for blk in re.split(r"\n  - ", "\n" + m.group(1))[1:]:
        d = {}
        nm = re.search(r"^\s*\.name:\s+(\S+)$", blk, re.M)
        if not nm:
            continue
        d["name"] = nm.group(1)
        for k in KEYS:
            mm = re.search(r"^\s*" + re.escape(k) + r":\s+(\d+)$", blk, re.M)
            d[k[1:]] = int(mm.group(1)) if mm else None
        out.append(d)
    return out


def resources(path, hipcc="/opt/rocm/bin/hipcc"):
    with tempfile.TemporaryDirectory() as td:
        o = os.path.join(td, "k.s")
        subprocess.run([hipcc, *FLAGS, "-I", CSRC, "-o", o, path], check=True, stderr=subprocess.DEVNULL)
        return parse_asm(open(o).read())


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.split("\n")
    except Exception:
        return names


def main(argv):
    js = None
    if "--json" in argv:
        i = argv.index("--json"); js = argv[i + 1]; argv = argv[:i] + argv[i + 2:]
    files = argv or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    allk = {}
    for f in files:
        ks = resources(f)
        for k, dn in zip(ks, demangle([k["name"] for k in ks])):
            short = re.sub(r"\(.*", "", dn)
            print(f"{os.path.basename(f):22s} {short[:60]:60s} vgpr {k['vgpr_count']:3d} agpr {k['agpr_count']:3d} "
                  f"spill v{k['vgpr_spill_count']} s{k['sgpr_spill_count']} scratch {k['private_segment_fixed_size']:3d} B "
                  f"lds {k['group_segment_fixed_size']:6d} wg {k['max_flat_workgroup_size']}")
            k["file"] = os.path.basename(f); k["demangled"] = short
            allk[k["name"]] = k
    if js:
        json.dump(allk, open(js, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
