#!/usr/bin/env python3
"""Generates tests/golden/*.json.

WHAT THESE ARE: small (SAM, FASTA, expected stream, expected decoded reads) vectors.
  * survey_kat.json holds the one output of the REAL reference available to this repo: the 12-byte
    stream prefix recorded in SURVEY.md section 8(a) for read length 100 (-DDEBUG build).
  * every other file is produced by oracle/cbc_oracle.c (the CPU restatement) and is a REGRESSION
    vector, not a reference output: the reference cannot be built in this image (its headers need
    libssh's dev package) and ships no tests or fixtures, so parity with it is "unpinned" beyond the
    prefix above.  They freeze today's behaviour so that the oracle, the kernel emulation and the
    HIP path cannot drift apart silently, and they travel to the GPU box.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import blockref  # noqa: E402
import synth  # noqa: E402
from cbc_amd import host  # noqa: E402
from oracle import oracle  # noqa: E402

CASES = {
    "perfect_L100": dict(seed=101, lens=[60000], reads=[300], L=100, kw=dict(sub_rate=0.0, indel_frac=0.0)),
    "snps_both_strands_L150": dict(seed=102, lens=[80000], reads=[300], L=150, kw=dict(sub_rate=0.01, indel_frac=0.0)),
    "indels_L100": dict(seed=103, lens=[60000], reads=[250], L=100, kw=dict(sub_rate=0.005, indel_frac=0.6)),
    "trailing_softclip_L100": dict(seed=104, lens=[60000], reads=[250], L=100,
                                   kw=dict(sub_rate=0.005, indel_frac=0.3, trailing_s_frac=0.4)),
    "two_contigs_dup_pos_L100": dict(seed=105, lens=[50000, 30000], reads=[200, 150], L=100,
                                     kw=dict(sub_rate=0.01, indel_frac=0.2, dup_pos_frac=0.2)),
    "sam_flags_L150": dict(seed=106, lens=[80000], reads=[300], L=150, kw=dict(flags=(0, 16, 83, 99, 147, 163))),
    "pos_escape_heavy_L100": dict(seed=107, lens=[400000], reads=[300], L=100, kw=dict()),
}


def main():
    with open(os.path.join(HERE, "survey_kat.json"), "w") as f:
        json.dump({"source": "SURVEY.md section 8(a): output of the real reference (-DDEBUG) on a 4-read, 100 bp file",
                   "read_length": 100, "stream_prefix_hex": "00 00 00 64 55 ff ff d4 85 79 db 94",
                   "file_bytes_for_4_reads": 121}, f, indent=1)
    for name, c in CASES.items():
        fa, sam, rbc, _ = synth.dataset(c["seed"], c["lens"], c["reads"], c["L"], **c["kw"])
        stream = oracle.encode(sam, fa)
        text, nr = oracle.decode(stream, fa)
        seqs = b"".join(r["seq"] + b"\n" for cc in rbc for r in cc[2])
        assert text == seqs, name
        # per-block payloads for block_reads=128: the oracle run on each block alone (rebased POS + window)
        pb = host.pack_sam(sam, fa, block_reads=128)
        lines = blockref.mapped_sam_lines(sam)
        blocks = []
        for b in range(pb.n_blocks):
            bsam, bfa = blockref.block_alone_inputs(pb, lines, b)
            blocks.append(oracle.encode(bsam, bfa).hex())
        pb.close()
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump({"generator": "oracle/cbc_oracle.c via tests/golden/make_golden.py (regression vector, not a "
                                    "reference output)",
                       "sam": sam.decode(), "fasta": fa.decode(), "stream_hex": stream.hex(),
                       "n_reads": nr, "block_reads": 128, "block_payload_hex": blocks}, f)
        print(name, len(sam), len(stream))


if __name__ == "__main__":
    main()
