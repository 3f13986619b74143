#!/bin/bash
# round 4, call 3: configs[3] shard (APTAI wav2vec2-large, 24 layers, 8 x 10 s): the new -m gpu tests, bench line, kernel statistics
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c; mkdir -p "$O"; cd "$R"
timeout -k 10 900 python -m pytest tests/test_gpu_graphed.py -m gpu -q -s -k bucketed_pr > "$O/pytest.log" 2>&1 || { tail -60 "$O/pytest.log"; exit 1; }
grep -E "bands|passed|failed" "$O/pytest.log" | tail -8
timeout -k 10 400 python bench.py --model large --steps 10 --warmup 3 > "$O/bench_large.json" 2> "$O/bench_large.err" || { tail -30 "$O/bench_large.err"; exit 1; }
cut -c1-400 "$O/bench_large.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/large_stats" -- python3 "$R/bench.py" --model large --steps 10 --warmup 3 --no-cpu-baseline > "$O/large_stats.log" 2>&1 || { tail -20 "$O/large_stats.log"; exit 1; }
find "$O/large_stats" -name "*kernel_stats.csv" | head -2
