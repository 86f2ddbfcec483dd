"""ORACLE package -- test infrastructure only.  See oracle/README.md.

Nothing under skghoi_amd/ (the product) imports this package; only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg do, and only as the checker.
"""
