"""TEST INFRASTRUCTURE -- CPU restatement of the read-set preparation step that feeds the hot path
(SURVEY.md section 8(f) row 4): sort the three FASTA files by header, count the corrected reads per header,
duplicate reference / uncorrected records accordingly.  Only tests/ may import this file.

PARITY UNPINNED: the reference functions (elector/readAndSortFiles.py:150-191) parse FASTA with Biopython's
SeqIO, which is absent from this image, and the reference holds no test or output fixture for them.  The FASTA
rules below restate Bio.SeqIO.FastaIO.SimpleFastaParser (Biopython 1.7x, the version range the reference's
setup asks for): text before the first line starting with '>' is skipped; title = that line without '>' and
without trailing whitespace; the sequence is the following lines, each right-stripped, joined, with spaces and
carriage returns removed.  The expected strings in tests/test_sort_cpu.py are derived by hand from these rules
and from the reference's call sites (readAndSortFiles.py:509-522)."""


def fasta_records(path):
    """-> [(description, sequence)] in file order (readAndSortFiles.py:151-152, SeqIO.parse(handle, "fasta"))"""
    out = []
    title, lines = None, []
    with open(path, "r") as handle:            # newline=None: what mode "rU" meant
        for line in handle:
            if line[0] == ">":
                if title is not None:
                    out.append((title, "".join(lines).replace(" ", "").replace("\r", "")))
                title, lines = line[1:].rstrip(), []
            elif title is not None:
                lines.append(line.rstrip())
    if title is not None:
        out.append((title, "".join(lines).replace(" ", "").replace("\r", "")))
    return out


def read_and_sort_fasta(infile, outfile):
    """readAndSortFiles.py:150-167 -> {description: number of records with it}"""
    recs = sorted(fasta_records(infile), key=lambda r: r[0])
    occ = {}
    prev = ""
    with open(outfile, "w") as out:
        for desc, seq in recs:
            out.write(">" + desc + "\n")
            out.write(seq + "\n")
            if desc == prev:
                occ[desc] += 1
            else:
                occ[desc] = 1
                prev = desc
    return occ


def duplicate_ref_reads(reference, uncorrected, occ, size, new_unco, new_ref):
    """readAndSortFiles.py:171-191.  The reference compares the dict with a list ([1] * size), which is never
    equal: the files are always rewritten, every kept record with the suffix _0 .. _(k-1), records whose header
    has no corrected read dropped."""
    with open(reference) as f:
        ref_lines = f.readlines()
    with open(uncorrected) as f:
        unc_lines = f.readlines()
    header = None
    with open(new_unco, "w") as nu, open(new_ref, "w") as nr:
        for unco, ref in zip(unc_lines, ref_lines):
            if ">" not in ref:
                if header in occ:
                    for t in range(occ[header]):
                        nr.write(">" + header + "_" + str(t) + "\n")
                        nr.write(ref.rstrip() + "\n")
                        nu.write(">" + header + "_" + str(t) + "\n")
                        nu.write(unco.rstrip() + "\n")
            else:
                header = ref.rstrip()[1:]
    return new_ref, new_unco
