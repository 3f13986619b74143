set -eo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_r02b
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/force_stats" -- python3 "$R/bench.py" --workload force --steps 5 --warmup 3 --no-cpu-baseline > "$O/force_stats.log" 2>&1
cd "$R"
python3 bench.py --workload force > "$O/bench_force.json" 2> "$O/bench_force.log"
tail -1 "$O/bench_force.json" | cut -c1-200
python3 bench.py > "$O/bench.json" 2> "$O/bench.log"
tail -1 "$O/bench.json" | cut -c1-200
