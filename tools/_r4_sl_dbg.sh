#!/bin/bash
# first call of a process through the split / whole launches, against the two-kernel path of the same process
O=gpurun_out/${1:-r4sldbg}; mkdir -p $O
ELECTOR_DEBUG_BINS=1 timeout -k 10 300 python tools/_r4_sl_dbg.py > $O/split.log 2>&1
ELECTOR_POA_SPLIT=0 ELECTOR_DEBUG_BINS=1 timeout -k 10 300 python tools/_r4_sl_dbg.py > $O/whole.log 2>&1
grep "^windows\|^bad\|k_poa lists" $O/split.log $O/whole.log | cut -c1-300
