"""stag_amd: MI355X-native implementation of STAG's post-alignment classification path.

Drop-in for ``stag classify`` / ``stag classify_genome`` (zellerlab/stag 0.8.2) on one path
only: aligned text -> one-hot encoding -> per-node logistic scores -> greedy tree walk ->
rank cut -> TSV.  All arithmetic of that path runs in hand-written HIP kernels for gfx950
(``stag_amd/csrc``) behind the C ABI of ``include/stag_hip.h``; this package is the host
side: table packing, the ctypes binding and the mirror of the reference's Python interface.
"""
__version__ = "0.8.2"   # tool version of the reference this build mirrors (stag/__init__.py:1)
