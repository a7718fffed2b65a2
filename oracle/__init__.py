"""CPU oracle of the particle update: TEST INFRASTRUCTURE ONLY (see
particles_oracle.c).  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the particles_amd package."""
