"""TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference CRIS path (see cris_oracle.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
