"""CPU oracle (test infrastructure -- NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; see oracle/ndt_oracle.h for provenance ("parity unpinned").
"""
from .binding import *  # noqa: F401,F403
