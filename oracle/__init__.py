"""CPU oracle for the DynamicFusion hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (dynamicfusion_body_amd/) never does: it fails loudly when the HIP
library is missing instead of falling back to anything in here.
"""
