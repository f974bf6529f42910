#!/usr/bin/env python
"""cProfile of the host side of the C3 streaming run (where does the wall time outside the kernels go?)."""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench_configs.py", "--config", "c3", "--steps", "120", "--nt", "6", "--nslots", "3"]
import tools.bench_configs as bc  # noqa: E402

pr = cProfile.Profile()
pr.enable()
bc.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue()[:6000])
