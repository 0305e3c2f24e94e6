#!/bin/bash
# Build a variant libvgpa_hip.so (-DVGPA_VARIANTS: + the measured-slower kernels and probes of tools/variants/) with extra -D flags on attention.hip (for A/B runs inside ONE gpurun session):
#   tools/build_variant.sh NAME [-DFOO ...]   ->  var/lib_NAME.so   (select with: VGPA_LIB=$PWD/var/lib_NAME.so python tools/attn_bench.py;
#   the product library videogpa_amd/csrc/libvgpa_hip.so is never overwritten.  var/ is git-ignored; it travels to the GPU box
#   only when .gpurunignore's `var/` line is commented out for an A/B session)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m videogpa_amd.build >/dev/null
mkdir -p var /tmp/vobj_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function \
  -I include -I videogpa_amd/csrc -I tools/variants -DVGPA_VARIANTS "$@" -c videogpa_amd/csrc/attention.hip -o /tmp/vobj_$name/attention.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function \
  -I include -I videogpa_amd/csrc -I tools/variants -DVGPA_VARIANTS "$@" -c tools/variants/gemm_w1.hip -o /tmp/vobj_$name/gemm_w1.o
objs=$(ls videogpa_amd/csrc/_obj/*.o | grep -v "/attention.o\|/gemm_w1.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/vobj_$name/attention.o /tmp/vobj_$name/gemm_w1.o -o var/lib_$name.so
echo "built var/lib_$name.so"
