cd $GRAFT_REPO_ROOT; ulimit -c 0
timeout 600 python -m pytest tests/test_round4_paths_gpu.py -q -x 2>&1 | grep -v "^$" | tail -40
timeout 300 python bench.py --no-cpu-baseline --no-hbm-kernels --steps 3 --warmup 1 2>&1 | tail -15
