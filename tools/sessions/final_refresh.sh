# After the last source edit of a round: the traffic file (stamped with the sources' hash) and the bench line it feeds.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
PROF_SQ=1 timeout 900 bash tools/profile.sh r03 > gpurun_out/r03/profile.log 2>&1; tail -5 gpurun_out/r03/profile.log
mkdir -p profiles_tmp && cp gpurun_out/prof/r03/traffic.json profiles/r03_traffic.json
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; tail -c 2500 gpurun_out/r03/bench_default.json
rmdir profiles_tmp
