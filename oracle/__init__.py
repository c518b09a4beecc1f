"""oracle — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
friture_amd/ (tests/test_no_oracle_in_product.py enforces that).
"""
