"""CPU-side check of the C++ front end's VCF float formatter (htslib kputd semantics) against every
float token the reference's golden VCFs contain: format(float32(token)) must give the token back."""
import os
import struct
import subprocess

import numpy as np
import pytest

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vcfgl_amd", "bin", "vcfgl_hip")


def _tokens():
    toks = set()
    refdir = os.path.join(gu.REFVCF, "reference")
    for d in sorted(os.listdir(refdir)):
        path = os.path.join(refdir, d, d + ".vcf")
        if not os.path.exists(path):
            continue
        for line in open(path):
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            for kv in f[7].split(";"):
                k, _, v = kv.partition("=")
                if k in ("QS", "I16"):
                    toks.update(v.split(","))
            keys = f[8].split(":")
            for smp in f[9:]:
                for k, v in zip(keys, smp.split(":")):
                    if k in ("GL", "GP"):
                        toks.update(v.split(","))
    toks.discard(".")
    return sorted(toks)


@pytest.mark.skipif(not os.path.exists(BIN), reason="vcfgl_hip not built")
def test_float_formatter_round_trips_every_golden_token():
    toks = _tokens()
    assert len(toks) > 500
    hexes = [format(struct.unpack("<I", struct.pack("<f", np.float32(float(t))))[0], "08x") for t in toks]
    out = []
    for i in range(0, len(hexes), 400):
        r = subprocess.run([BIN, "--format-floats"] + hexes[i:i + 400], capture_output=True, text=True, check=True)
        out += r.stdout.split("\n")[:-1]
    bad = [(t, o) for t, o in zip(toks, out) if t != o]
    assert not bad, bad[:20]
