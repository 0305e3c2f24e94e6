#!/bin/bash
# Compile one HIP source for gfx950 with the product flags and print a compact per-kernel resource table
# (VGPR / AGPR / spills / occupancy / LDS), keeping the .s next to the object:   tools/kres.sh SRC OUTDIR [-DFOO ...]
src=$1; out=$2; shift 2
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -munsafe-fp-atomics -fno-slp-vectorize \
  -Wno-unused-function -I /root/repo/include -I /root/repo/videogpa_amd/csrc "$@" -Rpass-analysis=kernel-resource-usage -save-temps=obj \
  -c $src -o $out/k.o 2> $out/res.txt || { tail -30 $out/res.txt; exit 1; }
python3 - $out/res.txt <<'PY'
import re,sys
cur=None
rows={}
for l in open(sys.argv[1]):
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r"remark: [^ ]+\s+(VGPRs|AGPRs|SGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|ScratchSize \[bytes/lane\]): (\d+)",l)
    if m and cur: rows[cur][m.group(1)]=int(m.group(2))
import subprocess
for k,v in rows.items():
    name=subprocess.run(["c++filt",k],capture_output=True,text=True).stdout.strip().split("(")[0]
    print(f"{name[:70]:70s} v{v.get('VGPRs',0):4d} a{v.get('AGPRs',0):4d} s{v.get('SGPRs',0):4d} vspill{v.get('VGPRs Spill',0):4d} scratch{v.get('ScratchSize [bytes/lane]',0):5d} occ{v.get('Occupancy [waves/SIMD]',0)} lds{v.get('LDS Size [bytes/block]',0)}")
PY
