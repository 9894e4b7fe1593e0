#!/bin/bash
# usage: regs.sh <lib.so> <pattern>  -> vgpr/sgpr/spill/lds of matching kernels
so=$1; pat=$2
tmp=$(mktemp -d)
cd $tmp
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$so >/dev/null 2>&1
# the fat binary sits in section .hip_fatbin
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=fat.bin $so $tmp/copy.so 2>/dev/null
python3 - "$pat" <<'PY'
import sys,re,subprocess,os
pat=sys.argv[1]
d=open('fat.bin','rb').read()
# split concatenated bundles
magic=b'__CLANG_OFFLOAD_BUNDLE__'
pos=[m.start() for m in re.finditer(magic,d)]
n=0
for i,p in enumerate(pos):
    e=pos[i+1] if i+1<len(pos) else len(d)
    open(f'b{i}.bin','wb').write(d[p:e])
    r=subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler','--unbundle','--type=o','--input',f'b{i}.bin','--targets=hipv4-amdgcn-amd-amdhsa--gfx950','--output',f'c{i}.co'],capture_output=True)
    if not os.path.exists(f'c{i}.co'): continue
    out=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf','--notes',f'c{i}.co'],capture_output=True,text=True).stdout
    # parse yaml-ish kernel entries
    for blk in out.split('  - .agpr_count')[1:]:
        name=re.search(r'\.name:\s+(\S+)',blk)
        if not name: continue
        nm=subprocess.run(['c++filt',name.group(1)],capture_output=True,text=True).stdout.strip()
        if not re.search(pat,nm): continue
        g=lambda k: (re.search(r'\.'+k+r':\s+(\d+)',blk) or [0,'?'])[1]
        print(f"{nm[:60]:60s} vgpr {g('vgpr_count')} sgpr {g('sgpr_count')} vspill {g('vgpr_spill_count')} sspill {g('sgpr_spill_count')} lds {g('group_segment_fixed_size')} scratch {g('private_segment_fixed_size')}")
PY
rm -rf $tmp
