"""Small host utilities mirrored from stag/helpers.py (zellerlab/stag 0.8.2)."""
import errno
import os
import subprocess
import sys


def is_tool(name):
    """helpers.py:68-75: a tool counts as present unless the attempt to start it says "no such file"."""
    try:
        subprocess.run([name], stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except FileNotFoundError:
        return False
    except OSError as e:
        return e.errno != errno.ENOENT
    return True


def read_fasta(fasta_stream, head_start=0, is_binary=True):
    """helpers.py:107-121 as a slow-path restatement (the fast path is stag_amd.ingest.fasta_batches):
    every line right-stripped; a line starting with '>' closes the record before it; a record is
    reported as (header[head_start:], its lines joined) only if it had at least one line; lines before
    the first header form a record whose id is None."""
    header, lines = None, []
    for raw in fasta_stream:
        text = (raw.decode("utf-8") if is_binary else raw).rstrip()
        if text[:1] != ">":
            lines.append(text)
            continue
        if lines:
            yield header, "".join(lines)
        header, lines = text[head_start:], []
    if lines:
        yield header, "".join(lines)


def check_file_exists(file_name, isfasta=False):
    """helpers.py:37-58 in spirit: the file must exist (and start with '>' when isfasta)."""
    try:
        with open(file_name, "r") as fh:
            if isfasta and not fh.readline().startswith(">"):
                sys.stderr.write("[E::main] Error: Not a fasta file: " + file_name + "\n")
                sys.stderr.write("          Fasta file is expected to start with '>'\n")
                sys.exit(1)
    except Exception as e:
        sys.stderr.write("[E::main] Error: Cannot open file: " + file_name + "\n" + str(e) + "\n")
        sys.exit(1)
