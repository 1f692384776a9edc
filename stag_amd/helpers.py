"""Small host utilities mirrored from stag/helpers.py (zellerlab/stag 0.8.2)."""
import errno
import os
import subprocess
import sys


def is_tool(name):
    """helpers.py:68-75: True unless starting `name` fails with ENOENT."""
    try:
        with open(os.devnull) as devnull:
            subprocess.Popen([name], stdout=devnull, stderr=devnull).communicate()
    except OSError as e:
        if e.errno == errno.ENOENT:
            return False
    return True


def read_fasta(fasta_stream, head_start=0, is_binary=True):
    """helpers.py:107-121: streaming (id, sequence); lines are rstrip()ped and joined; a record
    with an empty sequence is not yielded; head_start=1 drops the '>'."""
    sid, seq = None, []
    for line in fasta_stream:
        if is_binary:
            line = line.decode("utf-8")
        line = line.rstrip()
        if line.startswith(">"):
            if seq:
                yield sid, "".join(seq)
            sid = line[head_start:]
            seq.clear()
        else:
            seq.append(line)
    if seq:
        yield sid, "".join(seq)


def check_file_exists(file_name, isfasta=False):
    """helpers.py:37-58 in spirit: the file must exist (and start with '>' when isfasta)."""
    try:
        with open(file_name, "r") as fh:
            if isfasta and not fh.readline().startswith(">"):
                sys.stderr.write("[E::main] Error: Not a fasta file: " + file_name + "\n")
                sys.exit(1)
    except Exception as e:
        sys.stderr.write("[E::main] Error: Cannot open file: " + file_name + "\n" + str(e) + "\n")
        sys.exit(1)
