# sclk while a command runs:  bash tools/clk_probe2.sh <tag> <cmd...>
tag=$1; shift
( for i in 1 2 3 4 5 6 7 8 9 10; do sleep 0.8; rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | tr '\n' ' '; echo; done ) > gpurun_out/clk_$tag.txt 2>&1 < /dev/null &
SMI=$!
timeout 120 "$@" < /dev/null > /dev/null 2>&1
wait $SMI
echo $tag: $(grep -o "([0-9]*Mhz)" gpurun_out/clk_$tag.txt | tr '\n' ' ')
