"""elector_amd -- MI355X-native triplet-MSA engine for ELECTOR's hot path.

Only what the path needs lives here:

  csrc/         hand-written HIP kernels (gfx950) + the C-ABI host layer
  lib/          the built shared library (git-ignored; `python -m elector_amd.build`)
  _capi.py      ctypes binding of include/elector_poa.h
  poa.py        PoaEngine: batches of (reference, corrected, uncorrected) windows -> MSA rows
  alignment.py  drop-in mirror of the reference's elector/alignment.py (getPOA, fpoa)

There is no CPU fallback in this package: if the HIP library is missing or no
gfx950 device is present, the compute entry points raise.
"""
import os as _os

# The engine contexts keep several launch chains and batches in flight, each on a HIP stream of its own.  The
# runtime maps streams onto 4 hardware queues unless told otherwise; with 16 the chains really run side by side
# (bench.py: +12 % with four contexts).  Only effective when set before the process' first HIP call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

__version__ = "0.1.0"
