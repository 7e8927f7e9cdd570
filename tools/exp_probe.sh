for wl in d8_127 c320 d8_255; do echo "== $wl"; python bench.py --no-cpu --workload $wl --steps 30 --warmup 5 2>&1 | grep -E "probe|metric" | cut -c1-200; done
./tools/ubench/fma_issue 2>&1 | tail -4
