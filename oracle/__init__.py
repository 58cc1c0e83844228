"""CPU oracle for the text-index path (TEST INFRASTRUCTURE ONLY — see cpu_ref.cpp header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from .oracle import OracleIndex, brute_count, build_oracle_lib  # noqa: F401
