#!/usr/bin/env python3
"""scripts/kernel_registers.py <remarks file> [filter] — tabulates what `hipcc -Rpass-analysis=kernel-resource-usage` reports
for every `search_kernel` instantiation of one translation unit: VGPRs, spills, waves per SIMD. Template arguments are
<metric, scalar, lanes per row, variant, scratch mode, top cells per lane, frontier>."""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
needle = sys.argv[2] if len(sys.argv) > 2 else "search_kernel"
for block in re.split(r"remark: (?:\S+: )?Function Name: ", text)[1:]:  # with or without a file:line prefix
    name = block.split(" [")[0].strip()
    pretty = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if needle not in pretty:
        continue

    def field(key):
        found = re.search(key + r": (\d+)", block)
        return int(found.group(1)) if found else -1

    arguments = re.search(r"<(.*?)>", pretty)
    waves, scratch = field(r"Occupancy \[waves/SIMD\]"), field(r"ScratchSize \[bytes/lane\]")
    print(f"{pretty.split('<')[0].split()[-1]}<{arguments.group(1) if arguments else ''}>  VGPRs {field('VGPRs')}  "
          f"AGPRs {field('AGPRs')}  spilled {field('VGPRs Spill')}  SGPR spills {field('SGPRs Spill')}  waves/SIMD {waves}  "
          f"scratch {scratch}")
