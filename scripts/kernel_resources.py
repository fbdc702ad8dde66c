"""Per-kernel register / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics',
                      '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/tmp/_kr.o'],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\d+)', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip() or k
        name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0]
        print(f"{name[:70]:70s} vgpr {v.get('VGPRs', -1):4d} agpr {v.get('AGPRs', -1):4d} spill {v.get('VGPRs Spill', -1):4d} "
              f"scratch {v.get('ScratchSize [bytes/lane]', -1):5d} occ {v.get('Occupancy [waves/SIMD]', -1)} lds {v.get('LDS Size [bytes/block]', -1)}")
