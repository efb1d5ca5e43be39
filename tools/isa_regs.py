#!/usr/bin/env python3
"""Register / LDS / spill figures of the trace kernels from hipcc's assembly metadata.
Usage: python tools/isa_regs.py [-DFOO=1 ...] [--filter ELi8ELi0ELi0ELi0E]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    filt = "trace_kernel"
    if "--filter" in sys.argv:
        filt = sys.argv[sys.argv.index("--filter") + 1]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "t.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
                        "-S", "--cuda-device-only", "-o", out] + defs +
                       [os.path.join(ROOT, "nn_bvh_amd/csrc/bvh_trace.hip")], check=True,
                       stderr=subprocess.DEVNULL)
        text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    for block in meta.split("  - .agpr_count:")[1:]:
        get = lambda k: re.search(rf"\.{k}:\s*(\S+)", block).group(1)  # noqa: E731
        name = get("name")
        if filt in name:
            print(f"{name[14:60]:48s} vgpr {get('vgpr_count'):>3s} sgpr {get('sgpr_count'):>3s} "
                  f"spill {get('vgpr_spill_count'):>2s} scratch {get('private_segment_fixed_size'):>3s} "
                  f"lds {get('group_segment_fixed_size'):>5s}")


if __name__ == "__main__":
    main()
