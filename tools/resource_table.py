#!/usr/bin/env python3
"""Register / spill table of every kernel of the production build from hipcc's -Rpass-analysis=kernel-resource-usage remarks:
  for f in igemm attention norm misc train; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden \
      -Rpass-analysis=kernel-resource-usage -c agenda_amd/csrc/$f.hip -o /tmp/rp_$f.o 2> /tmp/rp_$f.log; done
  python tools/resource_table.py /tmp/rp_*.log > profiles/rNN_kernel_resource_usage.csv"""
import re
import subprocess
import sys

print("kernel,SGPRs,VGPRs,AGPRs,scratch_bytes_per_lane,waves_per_SIMD,SGPR_spills,VGPR_spills")
pat = re.compile(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                 r"Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", re.S)
for path in sys.argv[1:]:
    for m in pat.finditer(open(path).read()):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        print(f"\"{name}\"," + ",".join(m.groups()[1:]))
