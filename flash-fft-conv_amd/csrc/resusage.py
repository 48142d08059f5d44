"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin) as one line per kernel."""
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPR Spill|SGPR Spill)[^:]*:\s*(\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur:
            print(cur)
        cur = {"fn": v}
    else:
        cur[k.split()[0]] = v
if cur:
    print(cur)
