"""CPU oracle for the MaxSim hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under colpali_amd/ does (tests/test_no_oracle_in_product.py
enforces it).
"""
