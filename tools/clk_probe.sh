# sclk / power while a GEMM loop runs:  bash tools/clk_probe.sh <cfg> [lib]
cfg=$1
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do sleep 0.7; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Power\|Average Graphics" | tr '\n' ' '; echo; done ) > gpurun_out/clk_$cfg.txt 2>&1 < /dev/null &
SMI=$!
timeout 120 python tools/gemm_one.py 8192 8192 8192 $cfg 6000 < /dev/null
wait $SMI
tail -8 gpurun_out/clk_$cfg.txt | cut -c1-220
