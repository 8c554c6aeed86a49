for cfg in "prio1:MRB_BENCH_PRIO=1" "prio1:MRB_BENCH_PRIO=1" "prio0:MRB_BENCH_PRIO=0" "prio0:MRB_BENCH_PRIO=0" "nolook:MRB_X=1"; do
  tag=${cfg%%:*}; e=${cfg#*:}; extra=""; [ "$tag" = "nolook" ] && extra="--no-lookahead"
  out=$(env $e MRB_BENCH_SHARE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29917 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline $extra 2>&1)
  echo "== $tag: $(echo "$out" | grep -o "error word[^]]*" | sort | uniq -c | tr '\n' ' ') $(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1)" >> gpurun_out/t12.log
done
