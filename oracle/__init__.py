"""oracle -- CPU restatement of the NeRSemble per-sample hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import this package; nothing under nersemble_amd/ does
(tests/test_boundary.py checks that).  See oracle/nsx_oracle.h for the parity
status (third-party kernels "parity unpinned"; reference-owned glue pinned by
tests/golden/).
"""
from .capi import lib, build, GridGeom, grid_geometry  # noqa: F401
from . import hashgrid, march, mlp  # noqa: F401
