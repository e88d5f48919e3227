cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05b; rm -rf $O; mkdir -p $O
python3 bench.py 2>/dev/null | tail -1 > $O/bench_cfg2_result.json
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_cfg2_driver_flags_result.json
for c in 3 4 5; do timeout 600 python3 bench.py --config $c 2>/dev/null | tail -1 > $O/bench_cfg${c}_result.json; done
