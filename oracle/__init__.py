"""oracle/ -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker.  Nothing under ``slak_amd/`` imports it (tests/test_boundary.py
greps for that).  See oracle/dwconv_oracle.c, oracle/mask_oracle.py and oracle/optim_ema_oracle.py for the reference file:line
each function follows, and tests/test_oracle.py for how the oracle itself is pinned.
"""
from .dwconv import (dwconv2d_fwd, dwconv2d_bwd_data, dwconv2d_bwd_filter, bf16_round, build as build_c_oracle)
from .mask_oracle import (magnitude_prune, gradient_growth, random_growth, apply_mask, truncate_weights, cosine_prune_rate)

__all__ = ["dwconv2d_fwd", "dwconv2d_bwd_data", "dwconv2d_bwd_filter", "bf16_round", "build_c_oracle",
           "magnitude_prune", "gradient_growth", "random_growth", "apply_mask", "truncate_weights", "cosine_prune_rate"]
