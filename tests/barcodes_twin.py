"""TEST TWIN (not product code; the product rule is C++: csrc/nb_host.cpp nb_whitelist_correct). 10x cell-barcode whitelist correction
restated in Python from /root/reference/nimble/fastq_barcode_processor.py:17-128.

Rule (correct_cell_barcode, :73-128): exact whitelist match -> keep; else the whitelist entries at Hamming distance 1;
none -> the pair is dropped; one -> it; several -> the one whose differing base has the LOWEST phred quality in the raw
read. Results are cached per raw barcode, so the FIRST read seen with a given raw barcode decides for all later ones
(:90-92) — kept. The reference breaks quality ties by Python set iteration order (not reproducible across runs);
here ties go to the lexicographically smallest candidate. UMIs are never corrected (:176)."""
import gzip

_ACGT = "ACGT"


def load_whitelist(path):
    """load_cb_whitelist (:39-70) without the 64x variant index: candidates are found by substituting into the raw CB."""
    opener = gzip.open if path.endswith(".gz") else open
    wl = set()
    with opener(path, "rt") as f:
        for line in f:
            line = line.strip()
            if line:
                wl.add(line)
    return wl


class BarcodeCorrector:
    def __init__(self, whitelist):
        self.whitelist = whitelist
        self.cache = {}
        self.stats = {"cb_perfect_match": 0, "cb_corrected": 0, "cb_no_correction": 0}

    def candidates(self, raw):
        out = []
        for i, ch in enumerate(raw):
            for b in _ACGT:
                if b != ch:
                    c = raw[:i] + b + raw[i + 1:]
                    if c in self.whitelist:
                        out.append((i, c))
        return out

    def correct(self, raw, quals):
        """raw CB string, quals = phred ints (or the raw quality string) of the CB bases. Returns corrected CB or None."""
        if raw in self.cache:
            res = self.cache[raw]
        elif raw in self.whitelist:
            res = self.cache[raw] = raw
        else:
            cands = self.candidates(raw)
            if not cands:
                res = None
            elif len(cands) == 1:
                res = cands[0][1]
            else:
                q = [ord(x) - 33 for x in quals] if isinstance(quals, str) else quals
                res = min(cands, key=lambda ic: (q[ic[0]], ic[1]))[1]
            self.cache[raw] = res
        if res is None:
            self.stats["cb_no_correction"] += 1
        elif res == raw:
            self.stats["cb_perfect_match"] += 1
        else:
            self.stats["cb_corrected"] += 1
        return res
