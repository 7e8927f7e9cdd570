#!/usr/bin/env python3
"""Test scaffolding (tests/test_bench_contract.py): bench.py with its rank body replaced by tests/bench_dry_rank.run, so that
the launcher / rendezvous / relay path -- `python bench.py --gpus N` as a plain command, or under torch.distributed.run --
can be exercised where no GPU exists.  bench.py itself holds no hook for this: this script imports it, swaps run_rank and
calls its main(); the launcher starts its children as the script it was started as, i.e. as this file."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import bench            # noqa: E402
import bench_dry_rank   # noqa: E402

bench.run_rank = lambda a: bench_dry_rank.run(a, bench)
if __name__ == "__main__":
    bench.main()
