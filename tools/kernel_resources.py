#!/usr/bin/env python3
"""kernel_resources.py FILE.hip [name filter] — VGPR / SGPR / spill / scratch figures of every kernel of a HIP source
(compiles the device side to assembly with the library's flags and reads the kernel descriptors' metadata)."""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    out = d + "/k.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                    "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
for blk in s.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or flt not in name.group(1):
        continue
    g = lambda k: (re.search(k + r":\s+(\d+)", blk) or [None, "?"])[1]
    dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    print(f"{dem[:70]:70s} vgpr {g(r'.vgpr_count'):>3} sgpr {g(r'.sgpr_count'):>3} vspill {g(r'.vgpr_spill_count'):>3} sspill {g(r'.sgpr_spill_count'):>3} "
          f"lds {g(r'.group_segment_fixed_size'):>6} scratch {g(r'.private_segment_fixed_size'):>5}")
