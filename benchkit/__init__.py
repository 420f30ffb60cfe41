"""Support modules of bench.py (repo root): the workload table and hardware peaks, the rank launcher, the quotes from committed
profiles, and the record that goes to stdout.  bench.py holds the measurement itself and - alone - the CPU-baseline leg, the only
code outside tests/ that touches oracle/."""
