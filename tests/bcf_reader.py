"""A small BCF 2.2 / BGZF reader written from the published specifications (hts-specs VCFv4.3 section 6,
SAMv1 section 4.1), used only to check what the host program writes: it decodes a file back into VCF
text lines (floats through a caller-supplied formatter of the float32 bit pattern)."""
import struct
import zlib

INT_MISSING = {1: -128, 2: -32768, 3: -2147483648}
INT_VEND = {1: -127, 2: -32767, 3: -2147483647}
F_MISSING, F_VEND = 0x7F800001, 0x7F800002


def bgzf_blocks(raw):
    """Split a BGZF file into its members, checking every header field, CRC and size; returns the payloads."""
    out, off = [], 0
    while off < len(raw):
        assert raw[off:off + 4] == b"\x1f\x8b\x08\x04", "gzip member with FEXTRA expected"
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        assert xlen == 6 and raw[off + 12:off + 16] == b"BC\x02\x00", "BC subfield expected"
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        cdata = raw[off + 18:off + bsize - 8]
        crc, isize = struct.unpack_from("<II", raw, off + bsize - 8)
        data = zlib.decompress(cdata, -15)
        assert len(data) == isize and zlib.crc32(data) == crc and isize <= 65536
        out.append(data)
        off += bsize
    assert off == len(raw)
    assert out and out[-1] == b"", "BGZF end-of-file marker block missing"
    assert raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return out


class Reader:
    def __init__(self, path):
        raw = open(path, "rb").read()
        self.compressed = raw[:2] == b"\x1f\x8b"
        if self.compressed:
            raw = b"".join(bgzf_blocks(raw))
        assert raw[:5] == b"BCF\x02\x02", "BCF2.2 magic"
        l_text = struct.unpack_from("<I", raw, 5)[0]
        text = raw[9:9 + l_text]
        assert text[-1:] == b"\x00"
        self.header = text[:-1].decode().rstrip("\n").split("\n")
        assert self.header[-1].startswith("#CHROM")
        self.samples = self.header[-1].split("\t")[9:]
        self.dict, self.contigs, self.info_type, self.fmt_type = {}, {}, {}, {}
        nd, nc = 0, 0
        if not any(h.startswith("##FILTER=<ID=PASS") for h in self.header):
            self.dict[0] = "PASS"; nd = 1
        for h in self.header:
            kind = h[2:h.find("=")] if h.startswith("##") else ""
            if kind not in ("FILTER", "INFO", "FORMAT", "contig") or "<ID=" not in h:
                continue
            body = h[h.find("<") + 1:h.rfind(">")]
            attrs = dict(kv.split("=", 1) for kv in _split_attrs(body))
            ident = attrs["ID"]
            if kind == "contig":
                idx = int(attrs["IDX"]) if "IDX" in attrs else nc
                self.contigs[idx] = ident; nc = max(nc, idx + 1)
                continue
            known = [k for k, v in self.dict.items() if v == ident]
            idx = int(attrs["IDX"]) if "IDX" in attrs else (known[0] if known else nd)
            assert not known or known[0] == idx, "one dictionary index per ID"
            self.dict[idx] = ident; nd = max(nd, idx + 1)
            if kind == "INFO":
                self.info_type[ident] = attrs["Type"]
            if kind == "FORMAT":
                self.fmt_type[ident] = attrs["Type"]
        self.raw, self.off = raw, 9 + l_text

    # ---- typed values
    def _typed(self):
        b = self.raw[self.off]; self.off += 1
        n, t = b >> 4, b & 15
        if n == 15:
            (n,) = self._typed_ints()
        return t, n

    def _typed_ints(self):
        t, n = self._typed()
        return self._ints(t, n)

    def _ints(self, t, n):
        fmt = {1: "b", 2: "h", 3: "i"}[t]
        v = struct.unpack_from("<%d%s" % (n, fmt), self.raw, self.off)
        self.off += n * struct.calcsize(fmt)
        return [None if x == INT_MISSING[t] else ("END" if x == INT_VEND[t] else x) for x in v]

    def _value(self):
        t, n = self._typed()
        if t == 0:
            return 0, []
        if t in (1, 2, 3):
            return t, self._ints(t, n)
        if t == 5:
            v = list(struct.unpack_from("<%dI" % n, self.raw, self.off)); self.off += 4 * n
            return t, v
        assert t == 7, "unknown BCF type %d" % t
        v = self.raw[self.off:self.off + n].decode(); self.off += n
        return t, v

    def records(self):
        """Yield dict records with decoded fields (floats as uint32 bit patterns)."""
        while self.off < len(self.raw):
            l_shared, l_indiv = struct.unpack_from("<II", self.raw, self.off); self.off += 8
            end_shared = self.off + l_shared
            chrom, pos, rlen, qual, nai, nfs = struct.unpack_from("<iiiIII", self.raw, self.off); self.off += 24
            n_allele, n_info, n_fmt, n_sample = nai >> 16, nai & 0xFFFF, nfs >> 24, nfs & 0xFFFFFF
            assert n_sample == len(self.samples)
            _, ident = self._value()
            alleles = [self._value()[1] for _ in range(n_allele)]
            _, filt = self._value()
            info = []
            for _ in range(n_info):
                (k,) = self._typed_ints()
                info.append((self.dict[k],) + self._value())
            assert self.off == end_shared, "l_shared does not match the shared block"
            end_indiv = self.off + l_indiv
            fmt = []
            for _ in range(n_fmt):
                (k,) = self._typed_ints()
                t, n = self._typed()
                per = []
                for _s in range(n_sample):
                    if t == 5:
                        per.append(list(struct.unpack_from("<%dI" % n, self.raw, self.off))); self.off += 4 * n
                    else:
                        per.append(self._ints(t, n))
                fmt.append((self.dict[k], t, per))
            assert self.off == end_indiv, "l_indiv does not match the individual block"
            yield dict(chrom=self.contigs[chrom], pos0=pos, rlen=rlen, qual=qual, id=(ident if ident else "."), alleles=alleles,     # zero-length ID string = missing (htslib)
                       filter=[self.dict[f] for f in filt], info=info, fmt=fmt)

    def vcf_lines(self, fmt_float):
        """Records as VCF text; fmt_float(bits) renders one float32 bit pattern."""
        def fl(bits):
            return "." if bits == F_MISSING else fmt_float(bits)

        for r in self.records():
            info = []
            for k, t, v in r["info"]:
                if t == 0:
                    info.append(k)
                elif t == 7:
                    info.append(k + "=" + v)
                elif t == 5:
                    info.append(k + "=" + ",".join(fl(x) for x in v if x != F_VEND))
                else:
                    info.append(k + "=" + ",".join("." if x is None else str(x) for x in v if x != "END"))
            cols = [r["chrom"], str(r["pos0"] + 1), r["id"], r["alleles"][0], ",".join(r["alleles"][1:]) or ".",
                    "." if r["qual"] == F_MISSING else fmt_float(r["qual"]), ";".join(r["filter"]) or ".", ";".join(info) or "."]
            if self.samples:
                cols.append(":".join(k for k, _, _ in r["fmt"]) or ".")
                for s in range(len(self.samples)):
                    parts = []
                    for k, t, per in r["fmt"]:
                        v = per[s]
                        if k == "GT":
                            txt = ""
                            for i, x in enumerate(v):
                                if x == "END":
                                    break
                                if i:
                                    txt += "|" if (x & 1) else "/"
                                txt += "." if (x >> 1) == 0 else str((x >> 1) - 1)
                            parts.append(txt)
                        elif t == 5:
                            parts.append(",".join(fl(x) for x in v if x != F_VEND))
                        else:
                            parts.append(",".join("." if x is None else str(x) for x in v if x != "END"))
                    cols.append(":".join(parts) or ".")
            yield "\t".join(cols)


def _split_attrs(body):
    """Split 'ID=x,Number=1,Description="a, b"' on commas outside quotes."""
    out, cur, q = [], "", False
    for ch in body:
        if ch == '"':
            q = not q
        if ch == "," and not q:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur:
        out.append(cur)
    return [x for x in out if "=" in x]
