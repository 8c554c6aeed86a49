cd $GRAFT_REPO_ROOT; ulimit -c 0
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "thin_role or prefetch_workgroups or lora_rows" 2>&1 | grep -v "^$" | tail -40
