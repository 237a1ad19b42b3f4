"""TEST INFRASTRUCTURE ONLY: CPU oracle of the particle-filter path (see dpomp_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
