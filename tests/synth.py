"""Seeded synthetic FASTA + SAM generator for the parity tests (SURVEY.md section 8(d) recipe).

Contigs are uniform iid ACGT.  Reads: start positions uniform then sorted per contig, FLAG in
{0,16} (or a caller-supplied set), per-base substitution rate `sub_rate`, a fraction `indel_frac`
of reads carry one insertion or deletion of length 1..3 at least 10 bases from either end,
CIGAR uses M/I/D (optionally a trailing S), `MD:Z` is followed by `NM:i` (as BWA writes it),
QUAL constant 'I', QNAME r<index>, MAPQ 60, RNEXT * PNEXT 0 TLEN 0.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_contig(rng, n):
    return _ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]


def fasta_text(contigs, width=60):
    """contigs: list of (name, uint8 array).  Returns FASTA bytes with `width` bases per line."""
    out = []
    for name, seq in contigs:
        out.append(b">" + name.encode() + b"\n")
        n = len(seq)
        full = (n // width) * width
        if full:
            body = seq[:full].reshape(-1, width)
            lines = np.concatenate([body, np.full((body.shape[0], 1), 10, dtype=np.uint8)], axis=1)
            out.append(lines.tobytes())
        if n > full:
            out.append(seq[full:].tobytes() + b"\n")
    return b"".join(out)


def _md_and_nm(ref, start, ops, seq):
    """Build MD string and NM from the aligned pairs.  ops: list of (op, len)."""
    md = []
    run = 0
    nm = 0
    rpos = start
    qpos = 0
    prev_del = False
    for op, ln in ops:
        if op == "M":
            for _ in range(ln):
                if seq[qpos] == ref[rpos]:
                    run += 1
                else:
                    md.append(str(run))
                    md.append(chr(ref[rpos]))
                    run = 0
                    nm += 1
                rpos += 1
                qpos += 1
            prev_del = False
        elif op == "I" or op == "S":
            qpos += ln
            if op == "I":
                nm += ln
        elif op == "D":
            md.append(str(run))
            md.append("^" + bytes(ref[rpos:rpos + ln]).decode())
            run = 0
            rpos += ln
            nm += ln
            prev_del = True
    md.append(str(run))
    return "".join(md), nm


def make_reads(rng, contig, n_reads, L, sub_rate=0.003, indel_frac=0.02, flags=(0, 16),
               trailing_s_frac=0.0, dup_pos_frac=0.0, max_start=None):
    """Returns list of dict(pos, flag, cigar, seq, md, nm) sorted by pos (1-based POS)."""
    n = len(contig)
    hi = (n - L - 8) if max_start is None else max_start
    starts = np.sort(rng.integers(0, hi, size=n_reads))
    if dup_pos_frac > 0:
        dup = rng.random(n_reads) < dup_pos_frac
        for i in range(1, n_reads):
            if dup[i]:
                starts[i] = starts[i - 1]
    flag_choices = np.asarray(flags)
    fl = flag_choices[rng.integers(0, len(flag_choices), size=n_reads)]
    nsub = rng.binomial(L, sub_rate, size=n_reads) if sub_rate > 0 else np.zeros(n_reads, dtype=np.int64)
    has_indel = rng.random(n_reads) < indel_frac
    has_s = rng.random(n_reads) < trailing_s_frac
    recs = []
    for i in range(n_reads):
        s = int(starts[i])
        if nsub[i] == 0 and not has_indel[i] and not has_s[i]:
            seq = contig[s:s + L]
            recs.append(dict(pos=s + 1, flag=int(fl[i]), cigar="%dM" % L, seq=seq.tobytes(), md=str(L), nm=0))
            continue
        ops = [("M", L)]
        body_len = L
        if has_s[i]:
            k = int(rng.integers(1, 6))
            body_len = L - k
            ops = [("M", body_len), ("S", k)]
        if has_indel[i] and body_len > 30:
            k = int(rng.integers(1, 4))
            o = int(rng.integers(10, body_len - 10 - k))
            tail = ops[1:] if len(ops) > 1 else []
            if rng.random() < 0.5:
                ops = [("M", o), ("I", k), ("M", body_len - o - k)] + tail
            else:
                ops = [("M", o), ("D", k), ("M", body_len - o)] + tail
        # assemble the read from the reference
        parts = []
        rpos = s
        mpos = []  # read indices that are M bases
        qpos = 0
        for op, ln in ops:
            if op == "M":
                parts.append(contig[rpos:rpos + ln].copy())
                mpos.extend(range(qpos, qpos + ln))
                rpos += ln
                qpos += ln
            elif op == "I" or op == "S":
                parts.append(_ACGT[rng.integers(0, 4, size=ln)])
                qpos += ln
            elif op == "D":
                rpos += ln
        seq = np.concatenate(parts)
        assert len(seq) == L
        if nsub[i] > 0:
            where = rng.choice(len(mpos), size=min(int(nsub[i]), len(mpos)), replace=False)
            for w in where:
                q = mpos[int(w)]
                old = seq[q]
                alt = _ACGT[(int(np.where(_ACGT == old)[0][0]) + int(rng.integers(1, 4))) % 4]
                seq[q] = alt
        md, nm = _md_and_nm(contig, s, ops, seq)
        cigar = "".join("%d%s" % (ln, op) for op, ln in ops)
        recs.append(dict(pos=s + 1, flag=int(fl[i]), cigar=cigar, seq=seq.tobytes(), md=md, nm=nm))
    return recs


def sam_text(records_by_contig, header=True, md_last=False, qual_char=b"I", start_index=0):
    """records_by_contig: list of (name, contig_len, [records]).  Returns SAM bytes."""
    out = []
    if header:
        out.append(b"@HD\tVN:1.6\tSO:coordinate\n")
        for name, clen, _ in records_by_contig:
            out.append(("@SQ\tSN:%s\tLN:%d\n" % (name, clen)).encode())
    idx = start_index
    for name, _, recs in records_by_contig:
        nb = name.encode()
        for r in recs:
            seq = r["seq"]
            qual = qual_char * len(seq)
            if md_last:
                tags = ("NM:i:%d\tMD:Z:%s" % (r["nm"], r["md"])).encode()
            else:
                tags = ("MD:Z:%s\tNM:i:%d" % (r["md"], r["nm"])).encode()
            out.append(b"r%d\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t%s\t%s\t%s\n" % (
                idx, r["flag"], nb, r["pos"], r["cigar"].encode(), seq, qual, tags))
            idx += 1
    return b"".join(out)


def dataset(seed, contig_lens, reads_per_contig, L, names=None, **kw):
    """Convenience: returns (fasta_bytes, sam_bytes, records_by_contig, contigs)."""
    rng = np.random.default_rng(seed)
    contigs = []
    rbc = []
    for ci, (clen, nr) in enumerate(zip(contig_lens, reads_per_contig)):
        name = names[ci] if names else "chr%d" % (ci + 1)
        c = make_contig(rng, clen)
        contigs.append((name, c))
        rbc.append((name, clen, make_reads(rng, c, nr, L, **kw)))
    return fasta_text(contigs), sam_text(rbc), rbc, contigs


def shared_variant_dataset(seed, clen, n_reads, L, site_every, err, indel_sites=0):
    """Reads whose SNPs are SHARED: every read covering a variant site carries the site's alternative base
    (plus independent errors at rate `err`).  This is what real alignments look like, and it is the case
    where var contexts repeat ("the known variant d bases ahead").  Returns (fasta_bytes, sam_bytes)."""
    rng = np.random.default_rng(seed)
    contig = make_contig(rng, clen)
    alt = contig.copy()
    sites = rng.choice(clen, size=max(clen // site_every, 1), replace=False)
    for s in sites:
        alt[s] = _ACGT[(int(np.where(_ACGT == contig[s])[0][0]) + 1 + int(rng.integers(0, 3))) % 4]
    starts = np.sort(rng.integers(0, clen - L - 8, size=n_reads))
    out = []
    for i, s in enumerate(starts):
        s = int(s)
        seq = alt[s:s + L].copy()
        ne = int(rng.binomial(L, err)) if err > 0 else 0
        for q in (rng.choice(L, size=ne, replace=False) if ne else []):
            seq[q] = _ACGT[(int(np.where(_ACGT == seq[q])[0][0]) + 1 + int(rng.integers(0, 3))) % 4]
        md, nm = _md_and_nm(contig, s, [("M", L)], seq)
        out.append(b"r%d\t%d\tc\t%d\t60\t%dM\t*\t0\t0\t%s\t%s\tMD:Z:%s\tNM:i:%d\n" % (
            i, 16 * int(rng.integers(0, 2)), s + 1, L, seq.tobytes(), b"I" * L, md.encode(), nm))
    return fasta_text([("c", contig)]), b"".join(out)
