# power / sclk while a command runs:  bash tools/pwr_probe.sh <tag> <cmd...>
tag=$1; shift
( for i in 1 2 3 4 5; do sleep 1.0; rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Package Power\|sclk" | grep -o "([0-9]*Mhz)\|: [0-9.]*$" | tr '\n' ' '; echo; done ) > gpurun_out/pwr_$tag.txt 2>&1 < /dev/null &
SMI=$!
timeout 120 "$@" < /dev/null 2>&1 | tail -1
wait $SMI
echo "   $tag:" $(sed -n 3p gpurun_out/pwr_$tag.txt)
