"""CPU oracle for the vokselis raycast hot path.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/vokselis_oracle.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package.
"""
