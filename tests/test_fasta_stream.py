"""CPU tests of the streaming FASTA reader (ribbit_amd/csrc/fasta_stream.cpp, ribbit_fasta_*): the records it hands out
must be exactly those the reference's reader loop delimits (ribbit.cpp:269-280: getline; a '>' line ends the previous
record if it has bases and names the next by the text up to the first space; other lines are appended; the last record
is always processed), for files whose lines straddle the reader's 16-MB blocks, have no final newline,
carry '\\r', empty lines, empty records and headers without bases."""
import os

import numpy as np
import pytest

import ribbit_amd


def reference_loop(data: bytes):
    """ribbit.cpp:269-280 restated: [(name, sequence, is_last)]"""
    out = []
    name, seq = "", []                  # seq: the lines appended so far (`sequence += line`)
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()                     # getline does not yield an empty line after a final '\n'
    for line in lines:
        if line[:1] == b">":
            if any(seq):
                out.append((name, b"".join(seq), False))
            name = line[1:].split(b" ", 1)[0].decode()
            seq = []
        else:
            seq.append(line)
    out.append((name, b"".join(seq), True))
    return out


CASES = {
    "empty_file": b"",
    "one_record": b">chr1 some text\nACGT\nAC\n",
    "no_final_newline": b">a\nACGT\n>b\nTTTT",
    "header_without_bases_is_skipped": b">a\n>b\nAC\n>c\n>d x y\nGG\n",
    "ends_with_header": b">a\nAC\n>b",
    "ends_with_header_and_newline": b">a\nAC\n>b\n",
    "bases_before_any_header": b"ACGT\n>a\nGG\n",
    "carriage_returns_stay": b">a\r\nAC\r\nGT\r\n",
    "empty_lines": b">a\n\nAC\n\n\nGT\n\n",
    "gt_inside_a_line_is_a_base": b">a\nAC>GT\nA\n",
    "only_newlines": b"\n\n\n",
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_records_equal_the_reference_loops(tmp_path, name):
    path = tmp_path / "x.fa"
    path.write_bytes(CASES[name])
    assert ribbit_amd.read_fasta(str(path)) == reference_loop(CASES[name])


def test_lines_that_straddle_blocks_and_records_larger_than_a_block(tmp_path):
    rs = np.random.RandomState(3)
    parts = []
    for k, n in enumerate([70_000, 18_500_000, 10, 5_000_000, 17_200_000, 33]):
        seq = bytes(np.frombuffer(b"ACGTN", dtype=np.uint8)[rs.randint(0, 5, n)])
        width = int(rs.choice([60, 80, 1000, 17_000_000]))        # a single line longer than the 16-MB block too
        body = b"\n".join(seq[i:i + width] for i in range(0, n, width))
        parts.append(b">rec%d description %d\n" % (k, k) + body + b"\n")
    data = b"".join(parts)
    path = tmp_path / "big.fa"
    path.write_bytes(data)
    got = ribbit_amd.read_fasta(str(path))
    want = reference_loop(data)
    assert [(n, len(s), l) for n, s, l in got] == [(n, len(s), l) for n, s, l in want]
    assert all(a[1] == b[1] for a, b in zip(got, want))


def test_missing_file_is_an_error():
    with pytest.raises(ribbit_amd.RibbitHipError):
        ribbit_amd.read_fasta("/nonexistent/file.fa")
