"""Per-kernel resource usage from `hipcc -Rpass-analysis=kernel-resource-usage` output: python tools/kres.py <remarks file> <name filter>..."""
import re
import sys

txt = open(sys.argv[1]).read()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0]
    if not any(f in name for f in sys.argv[2:]):
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    print(name[:100], "VGPR", g("VGPRs"), "AGPR", g("AGPRs"), "SGPR", g("SGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
          "occ", g(r"Occupancy \[waves/SIMD\]"), "LDS", g(r"LDS Size \[bytes/block\]"))
