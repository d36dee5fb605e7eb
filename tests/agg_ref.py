"""The host half of computeMetrics (elector/computeStats.py:519-675) as plain Python loops: the cross-check of the
native aggregation (elector_amd/csrc/report_host.cpp, elector_report_aggregate) in the tests.  Test infrastructure;
the product calls the library."""
import statistics

import numpy as np

from elector_amd.computeStats import (ES_TP, ES_FP, ES_FN, ES_COR, ES_UNC, ES_UCOR, ES_UUNC, ES_GC_REF, ES_GC_COR, ES_INS_U,
                                      ES_DEL_U, ES_SUB_U, ES_INS_C, ES_DEL_C, ES_SUB_C, ES_LEN_REF, ES_LEN_COR, ES_LEN_UNC,
                                      ES_EXT_LEFT, ES_EXT_RIGHT, ES_MISSING, ES_MISSING_LAST, ES_PROCESSED, THRESH)


def aggregate_python(pieces, counters, ratios, outPerReadMetrics):
    """The host half of computeMetrics (computeStats.py:519-675): per-read ratios in
    read order from the integer counters."""
    nbReadsToDivide = 0
    countReadSplit = countReadExtended = countReadTrimmed = 0
    extendedBasesCount, missingSize = [], []
    indelsubsCorr, indelsubsUncorr = [0, 0, 0], [0, 0, 0]
    allLenCorrected, allLenUncorrected = [], []
    precision, recall, corBasesRate, uncorCorBasesRate = [], [], [], []
    totalCorBases = totalUncorBases = 0
    GCRateRef, GCRateCorr = [], []
    n_reads = len(pieces.read_first) - 1
    # plain Python integers from here on (the same values, several times cheaper to index than numpy scalars)
    read_first = np.asarray(pieces.read_first).tolist()
    counters = np.asarray(counters).tolist()
    per_read = []
    for r in range(n_reads):
        p0, p1 = read_first[r], read_first[r + 1]
        nfrag = p1 - p0
        split = nfrag > 1
        if split:
            countReadSplit += 1
        isExtended = isTrimmed = False
        TPs = FPs = FNs = cors = uncs = ucors = uuncs = 0
        any_piece = False
        gcr = gcc = 0
        emitted = False
        missingInRead = 0
        for k, p in enumerate(range(p0, p1)):
            c = counters[p]
            if not c[ES_PROCESSED]:
                continue
            any_piece = True
            if k == 0 or not split:
                allLenUncorrected.append(int(c[ES_LEN_UNC]))
            for side in (ES_EXT_LEFT, ES_EXT_RIGHT):
                if c[side] >= 0:
                    isExtended = True
                    extendedBasesCount.append(int(c[side]))
            missingInRead = int(c[ES_MISSING])
            if missingInRead > THRESH:
                isTrimmed = True
            indelsubsCorr[0] += int(c[ES_INS_C]); indelsubsCorr[1] += int(c[ES_DEL_C]); indelsubsCorr[2] += int(c[ES_SUB_C])
            indelsubsUncorr[0] += int(c[ES_INS_U]); indelsubsUncorr[1] += int(c[ES_DEL_U]); indelsubsUncorr[2] += int(c[ES_SUB_U])
            TPs += int(c[ES_TP]); FPs += int(c[ES_FP]); FNs += int(c[ES_FN])
            cors += int(c[ES_COR]); uncs += int(c[ES_UNC]); ucors += int(c[ES_UCOR]); uuncs += int(c[ES_UUNC])
            allLenCorrected.append(int(c[ES_LEN_COR]))
            gcr = round(int(c[ES_GC_REF]) * 1.0 / int(c[ES_LEN_REF]), 3)
            gcc = round(int(c[ES_GC_COR]) * 1.0 / int(c[ES_LEN_COR]), 3)
            if split and p == p1 - 1:
                missingInRead = int(c[ES_MISSING_LAST])
                emitted = True
            elif not split:
                emitted = True
        if not emitted:
            continue
        # outputMetrics (computeStats.py:444-468); a processed piece always left entries in the lists
        if any_piece:
            rec = TPs / (TPs + FNs) if (TPs + FNs) != 0 else 0
            prec = TPs / (TPs + FPs) if (TPs + FPs) != 0 else 0
            if missingInRead != 0:
                missingSize.append(missingInRead)
            corBRate = cors / (cors + uncs) if (cors + uncs) != 0 else 0
            uncorCorBRate = ucors / (ucors + uuncs) if (ucors + uuncs) != 0 else 0
            per_read.append(str(rec) + " recall\n" + str(prec) + " precision\n" + str(corBRate) + " correct_rate\n")
            recall.append(rec); precision.append(prec)
            corBasesRate.append(corBRate); uncorCorBasesRate.append(uncorCorBRate)
            totalCorBases += cors
            totalUncorBases += uncs
        GCRateRef.append(gcr)
        GCRateCorr.append(gcc)
        if isExtended:
            countReadExtended += 1
        if isTrimmed and not split:
            countReadTrimmed += 1
        nbReadsToDivide += 1

    outPerReadMetrics.write("".join(per_read))
    GCRateRef = round(sum(GCRateRef) / len(GCRateRef), 3)
    GCRateCorr = round(sum(GCRateCorr) / len(GCRateCorr), 3)
    recall = sum(recall) * 1.0 / nbReadsToDivide if nbReadsToDivide != 0 else 0
    precision = sum(precision) * 1.0 / nbReadsToDivide if nbReadsToDivide != 0 else 0
    corBasesRate = sum(corBasesRate) * 1.0 / nbReadsToDivide if nbReadsToDivide != 0 else 0
    uncorCorBasesRate = sum(uncorCorBasesRate) * 1.0 / nbReadsToDivide if nbReadsToDivide != 0 else 0
    throughput = sum(allLenCorrected)
    uncorThroughput = sum(allLenUncorrected)
    errorRate = 1 - (totalCorBases / (totalCorBases + totalUncorBases))
    uncorErrorRate = 1 - (totalUncorBases / (totalCorBases + totalUncorBases))
    meanRatioHomopolymers = statistics.mean(ratios) if len(ratios) > 1 else 1
    return (nbReadsToDivide, throughput, uncorThroughput, precision, recall, corBasesRate, errorRate,
            uncorCorBasesRate, uncorErrorRate, missingSize, GCRateRef, GCRateCorr, indelsubsUncorr, indelsubsCorr,
            meanRatioHomopolymers, allLenCorrected, countReadSplit, countReadTrimmed, countReadExtended,
            extendedBasesCount)
