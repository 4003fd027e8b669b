"""Test infrastructure: CPU restatement of the reference hot path (``orion_oracle.cpp``)
and, in the development container, the real reference classes behind a C harness
(``_ref/liborion_ref.so``).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product path never does."""
