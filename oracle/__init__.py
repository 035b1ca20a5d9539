"""CPU oracle for the SHG-VQA hot path - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything
from this package; the product (shg_vqa_amd/) never does.  See oracle/shg_ref.py.
"""
