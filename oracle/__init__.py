"""CPU checkers. TEST INFRASTRUCTURE: import only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
