#!/bin/bash
# Round evidence, run ON THE GPU BOX (through gpurun):  tools/profile_round.sh TAG
#  1. rocprofv3 --kernel-trace --stats over the default bench command (2 timed steps) -> gpurun_out/prof_TAG/ + kernel_stats csv
#  2. PMC traffic of the attention kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one, MI355X_MICROARCH.md
#     "rocprofv3 PMC slots"), no trace domains next to --pmc -> gpurun_out/pmc_TAG_{fetch,write}/
#  3. gpurun_out/pmc_traffic_TAG.json: per kernel HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB  (gfx950: FETCH_SIZE counts
#     128-B requests as 64 B -> doubled, as the guide prescribes; WRITE_SIZE taken as reported)
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline \
    > $R/gpurun_out/prof_${tag}_bench.json 2> $R/gpurun_out/prof_$tag.log
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_bench_kernel_stats.csv && python $R/profiles/summarize.py $f 40 > $R/gpurun_out/${tag}_bench_kernel_stats_summary.txt
find $R/gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete    # large; the stats are what is kept
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${tag}_fetch --output-format csv -- python $R/tools/attn_bench.py --iters 2 > $R/gpurun_out/pmc_${tag}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${tag}_write --output-format csv -- python $R/tools/attn_bench.py --iters 2 > $R/gpurun_out/pmc_${tag}_write.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${tag}_fetch_ew --output-format csv -- python $R/tools/ew_bench.py > $R/gpurun_out/pmc_${tag}_fetch_ew.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${tag}_write_ew --output-format csv -- python $R/tools/ew_bench.py > $R/gpurun_out/pmc_${tag}_write_ew.log 2>&1
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_${tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
alias = {"attn_fwd_pipe_kernel": "attn_fwd_kernel"}
out = {}
for k, cs in acc.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs and not k.startswith("at::") and "Cijk" not in k:
        # the largest dispatches of a name are the full-size launches (split tails / merges have their own names or are smaller)
        fs, ws = sorted(cs["FETCH_SIZE"])[-max(1, len(cs["FETCH_SIZE"]) // 2):], sorted(cs["WRITE_SIZE"])[-max(1, len(cs["WRITE_SIZE"]) // 2):]
        fetch, write = sum(fs) / len(fs), sum(ws) / len(ws)
        out[alias.get(k, k)] = {"fetch_size_kb_reported": fetch, "write_size_kb_reported": write,
                                "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                                "note": "2 x FETCH_SIZE (gfx950 wide-load correction) + WRITE_SIZE, KB -> bytes; mean over the larger half of the dispatches"}
json.dump(out, open("$R/gpurun_out/pmc_traffic_${tag}.json", "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items()}, indent=1))
PY
