"""bench.py, in parts: common (step, timing, flags), core (one rank's measurement), ranks + dataparallel (N > 1), baselines, report."""
