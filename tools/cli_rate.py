#!/usr/bin/env python3
"""End-to-end rate of the host program (VCF text in -> GPU simulation -> file out) per output mode and
--threads value, on a synthetic input of SITES x SAMPLES phased binary genotypes.  Everything is
inside the measured wall time: process start, input parsing, PCIe copies, record encoding, compression.
usage (GPU box): python tools/cli_rate.py [sites] [samples]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
BIN = os.path.join(ROOT, "vcfgl_amd", "bin", "vcfgl_hip")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
d = tempfile.mkdtemp(prefix="clirate")
vcf = os.path.join(d, "in.vcf")
gt = synth.binary_sites(0, S, N)                      # packed: low nibble allele 0, high nibble allele 1 (0 = REF, 1 = ALT in ACGT space)
tok = np.array(["0|0", "1|0", "0|1", "1|1"])
with open(vcf, "w") as f:
    f.write("##fileformat=VCFv4.2\n##contig=<ID=chr1,length=%d>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n" % (S + 1))
    f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("ind%d" % i for i in range(N)) + "\n")
    for i in range(S):
        g = gt[i]
        idx = (g & 0xF).astype(np.int64) + 2 * (g >> 4).astype(np.int64)
        f.write("chr1\t%d\t.\t0\t1\t.\tPASS\t.\tGT\t" % (i + 1) + "\t".join(tok[idx]) + "\n")
print(f"input: {S} sites x {N} samples, {os.path.getsize(vcf) / 1e6:.1f} MB of VCF text")
flags = "--seed 42 --depth 20 -e 0.01 --error-qs 2 --beta-variance 1e-5 -GL 2".split() + os.environ.get("CLI_EXTRA", "").split()   # e.g. CLI_EXTRA="--devices 0,0 --tile-sites 8192"
runs = (("v", 0), ("v", 1), ("u", 0), ("u", 1), ("u", 16), ("b", 0), ("b", 1), ("b", 64))
if os.environ.get("CLI_MODES"):                              # e.g. CLI_MODES="u:0,u:16"
    runs = tuple((m.split(":")[0], int(m.split(":")[1])) for m in os.environ["CLI_MODES"].split(","))
for mode, threads in runs:
    out = os.path.join(d, f"o_{mode}{threads}")
    t0 = time.perf_counter()
    th = ["--threads", str(threads)] if threads else []          # 0: not given (the program's default: up to 8 internal threads)
    r = subprocess.run([BIN, "-i", vcf, "-o", out, "-O", mode, "--verbose", "1"] + th + flags, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-800:]
    fn = out + {"v": ".vcf", "z": ".vcf.gz", "u": ".bcf", "b": ".bcf"}[mode]
    print(f"-O {mode} --threads {threads:3d}: {dt:7.2f} s  {S * N / dt:10.3e} evals/s  output {os.path.getsize(fn) / 1e6:8.1f} MB")
    print("    " + [l for l in r.stderr.splitlines() if l.startswith("[timing]")][-1])
    os.remove(fn)
