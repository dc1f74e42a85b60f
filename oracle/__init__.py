"""CPU oracle for the make_step hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under do_mpc_amd/ (the product) may import this package; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
"""
