# c320 tail (k_fir_generic, /5, 161 taps, 2^22 inputs): block shapes, same box.   gpurun -- bash tools/tail_shapes.sh
for sh in "" 256,1 256,3 128,3 64,3 64,1; do
  for rep in 1 2; do
  PDDC_GEN_SHAPE=$sh python bench.py --workload c320 --no-cpu --steps 100 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('shape [$sh]', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], round(d['ms_per_step']-d['roofline']['kernel_ms'],4), d['verified']['ok'])"
  done
done
