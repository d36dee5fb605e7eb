"""Read-set preparation in front of getPOA, without Biopython: the sort /
count / duplicate step of the reference's `processReadsForAlignment` (elector/readAndSortFiles.py:150-191,
483-522) with the reference's names, arguments and output files.  The header adaptors of the individual
correctors and the simulator converters (readAndSortFiles.py:196-444) are out of scope: the three FASTA files
are taken as the reference's `-perfect / -uncorrected / -corrected` mode takes them.

Parity: unpinned (Biopython is not in this image, the reference has no fixture for these functions); the FASTA
rules are Bio.SeqIO's `SimpleFastaParser`, see oracle/sort_oracle.py, which tests/test_sort_cpu.py holds this
module against."""
import re

_RECORD = re.compile(r"^>", re.M)


def _records(path):
    """[(description, sequence)]: a record starts at every line that begins with '>'; anything in front of the
    first one is not a record; description = the rest of that line, right-stripped; sequence = the lines up to
    the next record, right-stripped and joined, blanks and carriage returns taken out."""
    with open(path, "r") as f:                 # universal newlines, the reference's mode "rU"
        text = f.read()
    cuts = [m.start() for m in _RECORD.finditer(text)]
    cuts.append(len(text))
    out = []
    for a, b in zip(cuts, cuts[1:]):
        head, _, body = text[a + 1:b].partition("\n")
        seq = "".join(line.rstrip() for line in body.split("\n")) if body else ""
        out.append((head.rstrip(), seq.replace(" ", "").replace("\r", "")))
    return out


def readAndSortFasta(infileName, outfileName):
    """elector/readAndSortFiles.py:150-167: the records sorted by header (stable), two lines each;
    -> {header: how many records carry it}"""
    recs = _records(infileName)
    recs.sort(key=lambda r: r[0])
    occurrenceEachRead = {}
    parts = []
    for desc, seq in recs:
        parts.append(">" + desc + "\n" + seq + "\n")
        occurrenceEachRead[desc] = occurrenceEachRead.get(desc, 0) + 1     # equal headers are neighbours after the sort
    with open(outfileName, "w") as out:
        out.write("".join(parts))
    return occurrenceEachRead


def duplicateRefReads(reference, uncorrected, occurrenceEachRead, size, newUncoName, newRefName):
    """elector/readAndSortFiles.py:171-191.  `size` is unused in effect: the reference's test
    `occurrenceEachRead != [1]*size` compares a dict with a list and always holds, so the two files are always
    written -- every record whose header has corrected reads k times, named header_0 .. header_(k-1), the
    others dropped.  The files are walked line by line in lock step, the header taken from the reference file."""
    with open(reference) as f:
        refLines = f.readlines()
    with open(uncorrected) as f:
        uncoLines = f.readlines()
    ref_out, unco_out = [], []
    header = None
    for unco, ref in zip(uncoLines, refLines):
        if ">" in ref:
            header = ref.rstrip()[1:]
            continue
        if header is None:
            raise ValueError(reference + ": sequence line in front of the first header")
        for times in range(occurrenceEachRead.get(header, 0)):
            name = ">" + header + "_" + str(times) + "\n"
            ref_out.append(name + ref.rstrip() + "\n")
            unco_out.append(name + unco.rstrip() + "\n")
    with open(newRefName, "w") as f:
        f.write("".join(ref_out))
    with open(newUncoName, "w") as f:
        f.write("".join(unco_out))
    return newRefName, newUncoName


def sortAndDuplicate(corrector, reference, uncorrected, corrected, size, outputDirPath):
    """Steps 2 and 3 of `processReadsForAlignment` (elector/readAndSortFiles.py:483-522) for reads whose headers
    already agree across the three files (no simulator, headers as `formatHeader` leaves them): the sorted and the
    duplicated files under the reference's names in outputDirPath.
    -> (corrected, reference, uncorrected) file names to hand to alignment.getPOA"""
    tag = "_" + corrector if corrector is not None else ""
    by = "_by_" + corrector if corrector is not None else ""
    sortedCorrectedFileName = outputDirPath + "/corrected_sorted" + by + ".fa"
    sortedUncoFileName = outputDirPath + "/uncorrected_sorted" + tag + ".fa"
    newUncoFileName = outputDirPath + "/uncorrected_sorted_duplicated" + tag + ".fa"
    sortedRefFileName = outputDirPath + "/reference_sorted" + tag + ".fa"
    newRefFileName = outputDirPath + "/reference_sorted_duplicated" + tag + ".fa"
    readAndSortFasta(uncorrected, sortedUncoFileName)
    readAndSortFasta(reference, sortedRefFileName)
    occurrenceEachRead = readAndSortFasta(corrected, sortedCorrectedFileName)
    duplicateRefReads(sortedRefFileName, sortedUncoFileName, occurrenceEachRead, size, newUncoFileName, newRefFileName)
    return sortedCorrectedFileName, newRefFileName, newUncoFileName
