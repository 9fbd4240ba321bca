"""CPU oracle: restatement of the reference's algorithms for the hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import anything under oracle/.  The product path (self-supervised-wafermaps_amd/) never imports it
and fails loudly when the HIP library is missing.
"""
