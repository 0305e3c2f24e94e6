#!/bin/bash
# Build timing variants of the w1 kernels into var/lib_w1_NAME.so (select with VGPA_LIB=...):
#   tools/w1_variants.sh NAME "ABLATE-LIST" "KNOBS" [NAME2 ...]     e.g.  tools/w1_variants.sh novalu novalu "" lead4 "" lead=4
# The generated .inc files in csrc/ are restored to the product form at the end.
set -e
cd "$(dirname "$0")/.."
python -m videogpa_amd.build >/dev/null
mkdir -p var
while [ $# -ge 3 ]; do
  name=$1; abl=$2; knobs=$3; shift 3
  W1_ABLATE="$abl" W1_KNOBS="$knobs" python tools/gen_w1_asm.py >/dev/null
  mkdir -p /tmp/vobj_w1_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function \
    -I include -I videogpa_amd/csrc -c videogpa_amd/csrc/attention_w1.hip -o /tmp/vobj_w1_$name/attention_w1.o
  objs=$(ls videogpa_amd/csrc/_obj/*.o | grep -v attention_w1.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/vobj_w1_$name/attention_w1.o -o var/lib_w1_$name.so
  echo "built var/lib_w1_$name.so  (ablate='$abl' knobs='$knobs')"
done
python tools/gen_w1_asm.py >/dev/null
