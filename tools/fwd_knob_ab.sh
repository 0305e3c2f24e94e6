#!/bin/bash
# one-session A/B of a W1_KNOBS setting of the head_dim-64 forward loop (scratch copy on the GPU box; the product tree is not touched):
#   tools/fwd_knob_ab.sh pksum=1      -> time and joules per launch of attn_fwd for the product loop and for the knob
set -e
R=$GRAFT_REPO_ROOT
knob=$1
rm -rf /tmp/tree_knob && mkdir -p /tmp/tree_knob && cp -r $R/videogpa_amd $R/tools $R/include $R/profiles /tmp/tree_knob/
cd /tmp/tree_knob
W1_KNOBS=$knob python tools/gen_w1_asm.py > /dev/null
for f in w1_dq_loop w1_dkv_loop w1_fwd128_loop w1_fwd128f8_loop w1_dkv128_loop w1_dq128_loop w1_dq128x2_loop; do cp $R/videogpa_amd/csrc/$f.inc videogpa_amd/csrc/$f.inc; done
python -m videogpa_amd.build --force > /dev/null 2>&1
for rep in 1 2; do
  echo "== product loop (run $rep)"; cd $R && python tools/attn_bench.py --iters 3 --which fwd --energy 1.5 2>&1 | grep -E "^attn_fwd "
  echo "== $knob (run $rep)"; cd /tmp/tree_knob && python tools/attn_bench.py --iters 3 --which fwd --energy 1.5 2>&1 | grep -E "^attn_fwd "
done
cd /tmp/tree_knob && python tools/w1_check_fwd.py 2>&1 | tail -2
