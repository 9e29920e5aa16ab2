"""Damaged FASTA / FASTQ text and what the HOST parser reads in it (no GPU, no torch): used by the differential fuzz of the device-side
parser (tools/fuzz_raw_text.py) and by the host-layer test that the parallel parser hands on what ONE thread would
(tests/test_host_cli.py::test_damaged_files_are_read_as_one_thread_reads_them)."""
import numpy as np


def host_parser(text):
    """host/bank.cpp RecordParser: -> the read stream it hands on (one sequence per record, '\\n' behind each)"""
    out, st, seq, qleft = [], "HEADER", None, 0
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()                                  # (finish() only handles a non-empty remainder)

    def strip(ln):
        return ln.translate(None, b"\r \t")

    for ln in lines:
        if st == "HEADER":
            if ln[:1] == b">":
                st, seq = "SEQ_FA", b""
            elif ln[:1] == b"@":
                st, seq = "SEQ_FQ", b""
        elif st == "SEQ_FA":
            if ln[:1] == b">":
                out.append(seq); seq = b""
            else:
                seq += strip(ln)
        elif st == "SEQ_FQ":
            if ln[:1] == b"+":
                st, qleft = "QUAL", len(seq)
                if qleft == 0:
                    out.append(seq); seq = None; st = "HEADER"
            else:
                seq += strip(ln)
        else:
            q = len(ln) - ln.count(b"\r")
            if q >= qleft:
                out.append(seq); seq = None; st = "HEADER"
            else:
                qleft -= q
    if seq is not None:
        out.append(seq)
    return b"".join(s + b"\n" for s in out)


def base_text(rng, fmt):
    n = int(rng.integers(3, 60))
    eol = b"\r\n" if rng.random() < 0.2 else b"\n"
    alpha = np.frombuffer(b"ACGTACGTACGTacgtN", dtype=np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(0, 120))
        seq = bytes(rng.choice(alpha, L))
        if fmt == "fq":
            q = bytes(rng.integers(33, 74, L, dtype=np.uint8))
            out += [b"@r%d" % i + eol, seq + eol, b"+" + eol, q + eol]
        else:
            out += [b">s%d" % i + eol]
            w = int(rng.choice([30, 60, 1000]))
            out += [seq[a: a + w] + eol for a in range(0, L, w)]
    return out


def damage(rng, lines, fmt):
    lines = list(lines)
    for _ in range(int(rng.integers(0, 4))):
        if not lines:
            break
        i = int(rng.integers(0, len(lines)))
        kind = int(rng.integers(0, 8))
        if kind == 0:
            del lines[i]
        elif kind == 1:
            lines.insert(i, lines[i])
        elif kind == 2 and len(lines[i]) > 3:
            c = int(rng.integers(1, len(lines[i]) - 1)); lines[i: i + 1] = [lines[i][:c] + b"\n", lines[i][c:]]
        elif kind == 3 and i + 1 < len(lines):
            lines[i: i + 2] = [lines[i].rstrip(b"\r\n") + lines[i + 1]]
        elif kind == 4 and len(lines[i]) > 2:
            c = int(rng.integers(0, len(lines[i]) - 1)); lines[i] = lines[i][:c] + bytes([int(rng.choice(list(b"ACGT@>+ \t\r;N")))]) + lines[i][c:]
        elif kind == 5 and len(lines[i]) > 2:
            c = int(rng.integers(0, len(lines[i]) - 1)); lines[i] = lines[i][:c] + lines[i][c + 1:]
        elif kind == 6:
            lines.insert(i, b"\n")
        elif kind == 7:                                # a record of the other format at a record border
            j = next((x for x in range(i, len(lines)) if lines[x][:1] in (b"@", b">")), None)
            if j is not None:
                lines[j:j] = [b">x\n", b"ACGTTGCAACGTTGCAACGTTGCAACGTTGCAAC\n"] if fmt == "fq" else [b"@x\n", b"ACGTTGCAACGTTGCAACGTTGCAACGTTGCAAC\n", b"+\n", b"IIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n"]
    text = b"".join(lines)
    if rng.random() < 0.3:
        text = text.rstrip(b"\r\n")
    return text
