#!/usr/bin/env python3
"""Registers, spills and LDS of every kernel in the built libmpcore.so, read from the code object's metadata (no
compile, no GPU): objcopy the .hip_fatbin section, unbundle the gfx950 code object, llvm-readelf --notes.
The register screen kernels sit at the 128-VGPR edge of four wavefronts per SIMD; a change elsewhere in the file
can tip them into spilling (it did once) -- tests/test_abi_and_host.py asserts they do not.
Usage: python scripts/kernel_resources.py [substring]"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "matching-pursuit_amd", "lib", "libmpcore.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib=LIB):
    """-> {mangled kernel name: dict(vgpr, spill, sgpr, lds, scratch)}"""
    with tempfile.TemporaryDirectory() as tmp:
        fat, dev = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(tmp, "x")])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", dev], text=True)
    out = {}
    keys = {".vgpr_count": "vgpr", ".vgpr_spill_count": "spill", ".sgpr_count": "sgpr",
            ".group_segment_fixed_size": "lds", ".private_segment_fixed_size": "scratch"}
    # one YAML list item per kernel under amdhsa.kernels (items start at two spaces of indentation; the kernel's
    # arguments are nested lists further in)
    for rec in re.split(r"(?m)^  - (?=\.)", notes)[1:]:
        cur = {}
        for line in rec.splitlines():
            m = re.match(r"\s{0,4}(\.[a-z_]+):\s+(\S+)", line)
            if m and m.group(1) in keys:
                cur[keys[m.group(1)]] = int(m.group(2))
            elif m and m.group(1) == ".name":
                cur["name"] = m.group(2)
        if "name" in cur:
            out[cur.pop("name")] = cur
    return out


if __name__ == "__main__":
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, r in sorted(kernel_resources().items()):
        if want in name:
            short = subprocess.run(["c++filt", "-p", name], capture_output=True, text=True).stdout.strip() or name
            print(f"{short[:70]:70s} vgpr {r.get('vgpr', -1):4d} spill {r.get('spill', -1):4d} lds {r.get('lds', -1):6d} "
                  f"scratch {r.get('scratch', -1):5d}")
