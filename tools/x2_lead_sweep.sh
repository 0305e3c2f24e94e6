#!/bin/bash
# timing-only sweep of the Dq128x2Loop's fragment lead on the GPU box (scratch copies; the product tree is not touched)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lead in 4 6 8 10; do
  rm -rf /tmp/tree_$lead && mkdir -p /tmp/tree_$lead && cp -r $R/videogpa_amd $R/tools $R/include $R/profiles /tmp/tree_$lead/ 2>/dev/null
  cd /tmp/tree_$lead
  W1_KNOBS=lead=$lead python tools/gen_w1_asm.py > /dev/null
  git -C $R diff --quiet 2>/dev/null || true
  # only the x2 loop may differ from the product tree: restore the others
  for f in w1_dq_loop w1_dkv_loop w1_fwd_loop w1_fwd128_loop w1_fwd128f8_loop w1_dkv128_loop w1_dq128_loop w1_gemm_loop; do cp $R/videogpa_amd/csrc/$f.inc videogpa_amd/csrc/$f.inc; done
  python -m videogpa_amd.build --force > /dev/null 2>&1
  echo "== lead $lead"
  PYTHONPATH=/tmp/tree_$lead rocprofv3 --kernel-trace --stats -d /tmp/prof_lead_$lead --output-format csv -- python tools/attn128_time.py --B 2 --product-only --iters 4 2>&1 | grep -E "^bwd"
  f=$(find /tmp/prof_lead_$lead -name "*kernel_stats.csv" | head -1); python profiles/summarize.py $f 6 | grep -E "dq_w1x2"
  cd /tmp
done
