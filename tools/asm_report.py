"""Register / scratch / occupancy / LDS of every kernel, from `make -C vcfgl_amd/csrc asm` remarks.
usage: (cd vcfgl_amd/csrc && make asm 2>&1) | python tools/asm_report.py [substring ...]"""
import re, subprocess, sys
rows, cur = [], None
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = [m.group(1)]; rows.append(cur); continue
    m = re.search(r'(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', l)
    if m and cur is not None:
        cur.append(m.group(1).split()[0] + '=' + m.group(2))
names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, name in zip(rows, names):
    name = name.replace('void ', '').split('(')[0]
    if len(sys.argv) == 1 or any(a in name for a in sys.argv[1:]):
        print(f"{name:50s}", ' '.join(r[1:]))
