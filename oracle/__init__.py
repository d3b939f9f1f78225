"""CPU oracle package (test infrastructure only — see oracle/afvo.h).  PARITY UNPINNED."""
from .binding import *  # noqa: F401,F403
