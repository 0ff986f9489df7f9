"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's train-step hot path (PyTorch-CPU, because the reference is
Python/PyTorch) plus the harness that pins it against the real reference imported in the build
container.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (gw_depth_amd) never does and fails loudly without its HIP library.
"""
