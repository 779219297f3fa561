#!/bin/bash
# kernels of the built objects (pointcloudlib_amd/csrc/_obj) with register spills or scratch, from the code-object metadata
cd "$(dirname "$0")/../pointcloudlib_amd/csrc" || exit 1
B=/opt/rocm/lib/llvm/bin
for o in _obj/*.o; do
  $B/llvm-objcopy -O binary --only-section=.hip_fatbin $o /tmp/_fb.bin 2>/dev/null || continue
  $B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/_fb.bin --output=/tmp/_co.elf --unbundle 2>/dev/null || continue
  $B/llvm-readelf --notes /tmp/_co.elf 2>/dev/null | python3 -c "
import sys,re
txt=sys.stdin.read()
n=0
for it in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    g=lambda k: int(re.search(r'\.'+k+r':\s+(\d+)', it).group(1))
    name=re.search(r'\.name:\s+(\S+)', it).group(1)
    n+=1
    if g('vgpr_spill_count') or g('private_segment_fixed_size'):
        print('  %-90s vgprs %3d  vgpr spills %3d  sgpr spills %3d  scratch %4d B' % (name[:90], g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size')))
print('$o: %d kernels' % n)
"
done
