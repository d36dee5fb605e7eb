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
__version__ = "0.1.0"
