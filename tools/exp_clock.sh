# poll clocks / power while the bench is running
(timeout 120 python bench.py --no-cpu --workload ${1:-d8_127} --steps 30000 --warmup 5 > gpurun_out/clock_bench.txt 2>&1) &
BP=$!
for i in $(seq 1 60); do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|power \(W\)" | sed 's/=//g' | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
tail -1 gpurun_out/clock_bench.txt | cut -c1-200
