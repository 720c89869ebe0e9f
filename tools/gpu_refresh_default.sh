cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
R="$PWD"
timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-200
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_default" -o dflt -- python "$R/bench.py" --no-cpu-baseline > "$R/$O/rocprof_default.log" 2>&1); echo "rc=$?"
