#!/usr/bin/env python3
"""The spill scan lives in the package (elector_amd/check_spills.py: the build runs it); this is its command line."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elector_amd.check_spills import check, main  # noqa: E402,F401

if __name__ == "__main__":
    sys.exit(main())
