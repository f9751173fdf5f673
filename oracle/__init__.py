"""CPU oracle for the getVariations hot path -- TEST INFRASTRUCTURE ONLY (see kcf_oracle.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
