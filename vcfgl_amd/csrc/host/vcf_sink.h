// vcf_sink.h -- output side of the host program: VCF text (-O v), bgzip'd VCF (-O z), BCF 2.2
// uncompressed (-O u) and BGZF-compressed (-O b), the four --output-mode values of the reference
// (io.cpp:1204-1262).  htslib is not available on this platform, so the containers are written
// here from the published specifications (VCFv4.3 / BCFv2.2, SAM spec section 4.1 for BGZF):
//   * typed values are encoded the way htslib does (smallest integer type that holds every
//     value of a vector, bcf_enc_vint; float32 bit patterns untouched),
//   * the string dictionary is the order of first appearance of the FILTER/INFO/FORMAT IDs with
//     PASS = 0, written explicitly as IDX= in the header lines.
// Records come in two forms: a complete VCF text line (truth file, --depth inf, gVCF blocks), or the
// eight fixed columns as text plus typed descriptors of the FORMAT arrays (the simulated records,
// whose float32 values must reach a BCF file bit for bit).  In binary modes float INFO values travel
// inside the text as "~" + 8 hex digits of the bit pattern.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace vsink {

[[noreturn]] void fail(const char* fmt, ...);          // provided by the host program (prints, exits 1)

// ---------------------------------------------------------------------------------------
// fn(i) for i in [0, n) on up to `threads` threads (contiguous index ranges)
template <class F>
static void parallel_for(int n, int threads, F fn) {
    if (threads <= 1 || n <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
    const int T = std::min(threads, n);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([=]() { for (int i = (int)((int64_t)n * t / T), e = (int)((int64_t)n * (t + 1) / T); i < e; i++) fn(i); });
    for (auto& x : th) x.join();
}

// ---------------------------------------------------------------------------------------
// BGZF: a series of gzip members of at most 64 KiB, each carrying its compressed size in a 'BC'
// extra field, ended by an empty member.  Members are independent, so --threads N compresses N
// at a time (the role of htslib's thread pool, vcfgl.cpp:1790-1803); the file does not depend on N.
class Bgzf {
  public:
    void open(FILE* f, int threads_) { fp = f; threads = std::max(1, threads_); buf.reserve(BLOCK * (size_t)batch()); }
    void write(const void* p, size_t n) {
        const uint8_t* b = (const uint8_t*)p;
        const size_t cap = BLOCK * (size_t)batch();
        while (n) {
            const size_t k = std::min(n, cap - buf.size());
            buf.insert(buf.end(), b, b + k); b += k; n -= k;
            if (buf.size() == cap) flush();
        }
    }
    void flush() {
        if (buf.empty()) return;
        const int nb = (int)((buf.size() + BLOCK - 1) / BLOCK);
        std::vector<std::vector<uint8_t>> out(nb);
        parallel_for(nb, threads, [&](int i) {
            const size_t off = (size_t)i * BLOCK, len = std::min(BLOCK, buf.size() - off);
            compress_block(buf.data() + off, len, out[i]);
        });
        for (auto& o : out) if (fwrite(o.data(), 1, o.size(), fp) != o.size()) fail("write error");
        buf.clear();
    }
    void close() {
        flush();
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (fwrite(eof, 1, 28, fp) != 28) fail("write error");
    }
  private:
    static constexpr size_t BLOCK = 0xff00;
    FILE* fp = nullptr;
    int threads = 1;
    std::vector<uint8_t> buf;
    int batch() const { return threads == 1 ? 1 : threads * 4; }
    static void compress_block(const uint8_t* in, size_t len, std::vector<uint8_t>& o) {
        o.resize(65536);
        uint8_t* out = o.data();
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) fail("deflateInit2 failed");
        zs.next_in = const_cast<uint8_t*>(in); zs.avail_in = (uInt)len;
        zs.next_out = out + 18; zs.avail_out = 65536 - 18 - 8;
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) fail("BGZF block did not fit");       // 0xff00 input bytes always fit
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(out, hdr, 16);
        const uint16_t bsize = (uint16_t)(clen + 18 + 8 - 1);
        out[16] = (uint8_t)(bsize & 0xff); out[17] = (uint8_t)(bsize >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in, (uInt)len), isz = (uint32_t)len;
        uint8_t* t = out + 18 + clen;
        for (int i = 0; i < 4; i++) { t[i] = (uint8_t)(crc >> (8 * i)); t[4 + i] = (uint8_t)(isz >> (8 * i)); }
        o.resize(clen + 26);
    }
};

// ---------------------------------------------------------------------------------------
enum { BT_INT8 = 1, BT_INT16 = 2, BT_INT32 = 3, BT_FLOAT = 5, BT_CHAR = 7 };
enum { HT_FLAG = 0, HT_INT = 1, HT_FLOAT = 2, HT_STR = 3 };
static const int32_t I32_MISSING = INT32_MIN, I32_VEND = INT32_MIN + 1;
static const uint32_t F32_MISSING = 0x7F800001u, F32_VEND = 0x7F800002u;

// FORMAT array of one record: value of sample s, element k at base[s * ss + k * sk]
struct FmtDesc { const char* key; bool is_float; int n; const void* base; size_t ss, sk; };

class Sink {
  public:
    bool binary() const { return mode == 'u' || mode == 'b'; }

    // float / int tokens of the text handed to write_line() / write_rec()
    void put_float(std::string& s, float f) const {
        if (!binary()) { text_float(s, f); return; }
        uint32_t b; memcpy(&b, &f, 4); char t[16]; snprintf(t, sizeof t, "~%08x", b); s += t;
    }
    static void put_int(std::string& s, int32_t v) {
        if (v == I32_MISSING) { s += '.'; return; }
        char t[16]; snprintf(t, sizeof t, "%d", v); s += t;
    }
    void (*text_float)(std::string&, float) = nullptr;     // the kputd()-style formatter of the host program

    // header = the '##' lines in output order; records may only use contigs / keys these lines define
    // (define_missing() adds the definitions htslib would add with a warning)
    void open(const std::string& path, char mode_, std::vector<std::string> header, const std::vector<std::string>& samples, int threads = 1) {
        mode = mode_; N = (int)samples.size();
        fp = fopen(path.c_str(), "wb");
        if (!fp) fail("Could not open file: %s", path.c_str());
        if (mode == 'z' || mode == 'b') bg.open(fp, threads);
        if (binary()) build_dictionaries(header);
        std::string text;
        for (const std::string& h : header) { text += h; text += '\n'; }
        text += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
        if (N > 0) { text += "\tFORMAT"; for (const std::string& s : samples) { text += '\t'; text += s; } }
        text += '\n';
        if (binary()) {
            const uint32_t l_text = (uint32_t)text.size() + 1;
            std::string h = "BCF\2\2";
            put_u32(h, l_text);
            h += text; h += '\0';
            emit(h.data(), h.size());
        } else emit(text.data(), text.size());
    }

    // contigs, FILTER values and INFO keys used by the input records but not defined by its header
    static void define_missing(std::vector<std::string>& header, const std::vector<std::string>& contigs,
                               const std::vector<std::string>& filters, const std::vector<std::string>& info_keys) {
        auto has = [&](const std::string& pre, const std::string& id) {
            for (const std::string& h : header) if (h.compare(0, pre.size(), pre) == 0 && id_of(h) == id) return true;
            return false;
        };
        for (const std::string& c : contigs) if (!has("##contig=<", c)) header.push_back("##contig=<ID=" + c + ">");
        for (const std::string& f : filters) if (f != "PASS" && f != "." && !has("##FILTER=<", f)) header.push_back("##FILTER=<ID=" + f + ",Description=\"Dummy\">");
        for (const std::string& k : info_keys) if (!has("##INFO=<", k)) header.push_back("##INFO=<ID=" + k + ",Number=1,Type=String,Description=\"Dummy\">");
    }

    // Records are encoded into byte strings by encode_line() / encode_rec() (const: several threads may
    // encode different records at once) and appended to the file in order by put().
    void put(const std::string& bytes) { emit(bytes.data(), bytes.size()); }
    void write_line(const std::string& line) { std::string b; encode_line(line, b); put(b); }
    void write_rec(const std::string& shared8, const std::vector<FmtDesc>& fmt) { std::string b; encode_rec(shared8, fmt, b); put(b); }

    // a complete VCF record as text (without the newline)
    void encode_line(const std::string& line, std::string& out) const {
        if (!binary()) { out += line; out += '\n'; return; }
        std::vector<std::string> col; split(line, '\t', col);
        if (col.size() < 8) fail("internal: short VCF line");
        std::string shared, indiv;
        uint32_t n_fmt = 0;
        if (col.size() > 9 && col[8] != ".") n_fmt = encode_format_text(col, indiv);
        encode_shared(col, n_fmt, shared);
        finish(shared, indiv, out);
    }

    // eight fixed columns as text + typed FORMAT arrays
    void encode_rec(const std::string& shared8, const std::vector<FmtDesc>& fmt, std::string& out) const {
        if (!binary()) {
            std::string& line = out;
            line += shared8;
            line += '\t';
            if (fmt.empty()) line += '.';
            for (size_t i = 0; i < fmt.size(); i++) { if (i) line += ':'; line += fmt[i].key; }
            for (int s = 0; s < N; s++) {
                line += '\t';
                if (fmt.empty()) line += '.';
                for (size_t i = 0; i < fmt.size(); i++) {
                    if (i) line += ':';
                    const FmtDesc& d = fmt[i];
                    for (int k = 0; k < d.n; k++) {
                        if (k) line += ',';
                        if (d.is_float) text_float(line, ((const float*)d.base)[s * d.ss + k * d.sk]);
                        else put_int(line, ((const int32_t*)d.base)[s * d.ss + k * d.sk]);
                    }
                }
            }
            line += '\n';
            return;
        }
        std::vector<std::string> col; split(shared8, '\t', col);
        if (col.size() != 8) fail("internal: shared part needs 8 columns");
        std::string shared, indiv;
        for (const FmtDesc& d : fmt) {
            enc_int1(indiv, dict_id(d.key));
            if (d.is_float) {
                enc_size(indiv, d.n, BT_FLOAT);
                const size_t at = indiv.size();
                indiv.resize(at + (size_t)4 * d.n * N);
                uint8_t* o = (uint8_t*)&indiv[at];
                if (d.sk == 1 && d.ss == (size_t)d.n) memcpy(o, d.base, (size_t)4 * d.n * N);      // sample-major input: the record's array as BCF stores it
                else for (int s = 0; s < N; s++) for (int k = 0; k < d.n; k++, o += 4) memcpy(o, &((const float*)d.base)[s * d.ss + k * d.sk], 4);
            } else {
                const int32_t* b = (const int32_t*)d.base;
                int32_t mx = INT32_MIN, mn = INT32_MAX;
                const bool linear = (d.sk == 1 && d.ss == (size_t)d.n) || d.n == 1;       // values already in the record's order
                const size_t total = (size_t)d.n * N;
                if (linear) {
                    for (size_t j = 0; j < total; j++) {
                        const int32_t v = b[j];
                        if (v == I32_MISSING || v == I32_VEND) continue;
                        if (v > mx) mx = v; if (v < mn) mn = v;
                    }
                } else
                for (int s = 0; s < N; s++) for (int k = 0; k < d.n; k++) {
                    const int32_t v = b[s * d.ss + k * d.sk];
                    if (v == I32_MISSING || v == I32_VEND) continue;
                    if (v > mx) mx = v; if (v < mn) mn = v;
                }
                const int bt = int_type(mn, mx);
                enc_size(indiv, d.n, bt);
                if (linear) put_typed_ints(indiv, b, total, bt);
                else for (int s = 0; s < N; s++) for (int k = 0; k < d.n; k++) put_typed_int(indiv, b[s * d.ss + k * d.sk], bt);
            }
        }
        encode_shared(col, (uint32_t)fmt.size(), shared);
        finish(shared, indiv, out);
    }

    void close() {
        if (!fp) return;
        if (mode == 'z' || mode == 'b') bg.close();
        if (fclose(fp) != 0) fail("write error");
        fp = nullptr;
    }

  private:
    char mode = 'v';
    int N = 0;
    FILE* fp = nullptr;
    Bgzf bg;
    std::map<std::string, int> dict, contig;                 // string dictionary, contig dictionary
    std::map<std::string, int> info_type, fmt_type;          // key -> HT_*

    void emit(const void* p, size_t n) {
        if (mode == 'z' || mode == 'b') bg.write(p, n);
        else if (fwrite(p, 1, n, fp) != n) fail("write error");
    }
    static void finish(const std::string& shared, const std::string& indiv, std::string& out) {
        put_u32(out, (uint32_t)shared.size()); put_u32(out, (uint32_t)indiv.size());
        out += shared; out += indiv;
    }
    static void split(const std::string& s, char c, std::vector<std::string>& out) {
        out.clear(); size_t b = 0;
        while (true) { const size_t e = s.find(c, b); if (e == std::string::npos) { out.push_back(s.substr(b)); break; } out.push_back(s.substr(b, e - b)); b = e + 1; }
    }
    static void put_u32(std::string& s, uint32_t v) { for (int i = 0; i < 4; i++) s += (char)(v >> (8 * i)); }
    static std::string id_of(const std::string& h) {
        const size_t a = h.find("<ID=");
        if (a == std::string::npos) return "";
        return h.substr(a + 4, h.find_first_of(",>", a + 4) - a - 4);
    }
    static std::string attr_of(const std::string& h, const char* key) {
        const std::string k = std::string(",") + key + "=";
        const size_t a = h.find(k);
        if (a == std::string::npos) return "";
        return h.substr(a + k.size(), h.find_first_of(",>", a + k.size()) - a - k.size());
    }

    // dictionaries in order of first appearance, PASS first; the lines get an explicit IDX=
    void build_dictionaries(std::vector<std::string>& header) {
        bool have_pass = false;
        for (const std::string& h : header) if (h.compare(0, 10, "##FILTER=<") == 0 && id_of(h) == "PASS") have_pass = true;
        if (!have_pass) header.insert(header.begin() + (header.empty() ? 0 : 1), "##FILTER=<ID=PASS,Description=\"All filters passed\">");
        dict["PASS"] = 0;
        for (std::string& h : header) {
            const bool fil = h.compare(0, 10, "##FILTER=<") == 0, inf = h.compare(0, 8, "##INFO=<") == 0, fmt = h.compare(0, 10, "##FORMAT=<") == 0;
            const bool ctg = h.compare(0, 10, "##contig=<") == 0;
            if (!(fil || inf || fmt || ctg)) continue;
            const std::string id = id_of(h);
            if (id.empty()) fail("header line without ID: %s", h.c_str());
            int idx;
            if (ctg) { if (!contig.count(id)) { idx = (int)contig.size(); contig[id] = idx; } else idx = contig[id]; }
            else { if (!dict.count(id)) { idx = (int)dict.size(); dict[id] = idx; } else idx = dict[id]; }
            if (inf || fmt) {
                const std::string t = attr_of(h, "Type");
                const int ht = (t == "Integer") ? HT_INT : (t == "Float") ? HT_FLOAT : (t == "Flag") ? HT_FLAG : HT_STR;
                (inf ? info_type : fmt_type)[id] = ht;
            }
            // an input header (BCF in particular) may carry its own IDX= values: the index written is always the one the
            // records are encoded with, so a stale attribute is replaced (FORMAT/GT is dropped from the output, which shifts
            // every key that was defined after it)
            for (size_t a; (a = h.find(",IDX=")) != std::string::npos;) {
                size_t e = a + 5; while (e < h.size() && h[e] >= '0' && h[e] <= '9') e++;
                h.erase(a, e - a);
            }
            if (h.back() == '>') { char t[32]; snprintf(t, sizeof t, ",IDX=%d>", idx); h.pop_back(); h += t; }
        }
    }
    int dict_id(const std::string& k) const {
        auto it = dict.find(k);
        if (it == dict.end()) fail("tag %s is not defined in the output header", k.c_str());
        return it->second;
    }

    // ---- typed values (BCF2 section 6.3; size / type choices as htslib's bcf_enc_*)
    static int int_type(int32_t mn, int32_t mx) {
        if (mx <= 127 && mn >= -120) return BT_INT8;
        if (mx <= 32767 && mn >= -32760) return BT_INT16;
        return BT_INT32;
    }
    static void put_typed_int(std::string& s, int32_t v, int bt) {
        if (bt == BT_INT8) s += (char)(v == I32_MISSING ? 0x80 : v == I32_VEND ? 0x81 : v);
        else if (bt == BT_INT16) { const uint16_t x = (uint16_t)(v == I32_MISSING ? 0x8000 : v == I32_VEND ? 0x8001 : v); s += (char)(x & 0xff); s += (char)(x >> 8); }
        else put_u32(s, (uint32_t)v);
    }
    // `n` values of one typed vector, written in one resize (the per-value appends of put_typed_int dominate the encoding of a wide record)
    static void put_typed_ints(std::string& s, const int32_t* v, size_t n, int bt) {
        const size_t at = s.size();
        if (bt == BT_INT8) {
            s.resize(at + n);
            uint8_t* o = (uint8_t*)&s[at];
            for (size_t j = 0; j < n; j++) o[j] = (uint8_t)(v[j] == I32_MISSING ? 0x80 : v[j] == I32_VEND ? 0x81 : v[j]);
        } else if (bt == BT_INT16) {
            s.resize(at + 2 * n);
            uint8_t* o = (uint8_t*)&s[at];
            for (size_t j = 0; j < n; j++) { const uint16_t x = (uint16_t)(v[j] == I32_MISSING ? 0x8000 : v[j] == I32_VEND ? 0x8001 : v[j]); o[2 * j] = (uint8_t)(x & 0xff); o[2 * j + 1] = (uint8_t)(x >> 8); }
        } else {
            s.resize(at + 4 * n);
            memcpy(&s[at], v, 4 * n);                                    // little endian host, like the file
        }
    }
    static void enc_int1(std::string& s, int32_t v) {
        const int bt = (v == I32_MISSING || v == I32_VEND) ? BT_INT8 : int_type(v, v);
        s += (char)(1 << 4 | bt); put_typed_int(s, v, bt);
    }
    static void enc_size(std::string& s, int n, int bt) {
        if (n < 15) { s += (char)(n << 4 | bt); return; }
        s += (char)(15 << 4 | bt);
        enc_int1(s, n);
    }
    static void enc_vint(std::string& s, const std::vector<int32_t>& v) {
        if (v.empty()) { s += (char)0; return; }                     // htslib: bcf_enc_size(0, BCF_BT_NULL)
        if (v.size() == 1) { enc_int1(s, v[0]); return; }
        int32_t mx = INT32_MIN, mn = INT32_MAX;
        for (int32_t x : v) { if (x == I32_MISSING || x == I32_VEND) continue; if (x > mx) mx = x; if (x < mn) mn = x; }
        const int bt = int_type(mn, mx);
        enc_size(s, (int)v.size(), bt);
        for (int32_t x : v) put_typed_int(s, x, bt);
    }
    static void enc_str(std::string& s, const std::string& v) { enc_size(s, (int)v.size(), BT_CHAR); s += v; }
    static uint32_t float_bits(const std::string& t) {
        if (t == ".") return F32_MISSING;
        if (t[0] == '~') return (uint32_t)strtoul(t.c_str() + 1, nullptr, 16);
        const float f = strtof(t.c_str(), nullptr);
        uint32_t b; memcpy(&b, &f, 4); return b;
    }
    static int32_t int_value(const std::string& t) { return t == "." ? I32_MISSING : (int32_t)strtol(t.c_str(), nullptr, 10); }

    // CHROM .. INFO (BCF2 section 6.3.1)
    void encode_shared(const std::vector<std::string>& col, uint32_t n_fmt, std::string& out) const {
        auto c = contig.find(col[0]);
        if (c == contig.end()) fail("contig %s is not defined in the output header", col[0].c_str());
        std::vector<std::string> alleles, alts, infos, filt;
        alleles.push_back(col[3]);
        if (col[4] != ".") { split(col[4], ',', alts); for (auto& x : alts) alleles.push_back(x); }
        if (col[7] != "." && !col[7].empty()) split(col[7], ';', infos);
        const long pos0 = strtol(col[1].c_str(), nullptr, 10) - 1;
        long rlen = (long)col[3].size();
        std::string info_bytes; uint32_t n_info = 0;
        for (const std::string& kv : infos) {
            if (kv.empty()) continue;
            const size_t eq = kv.find('=');
            const std::string k = kv.substr(0, eq), v = (eq == std::string::npos) ? "" : kv.substr(eq + 1);
            auto ti = info_type.find(k);
            if (ti == info_type.end()) fail("INFO key %s is not defined in the output header", k.c_str());
            enc_int1(info_bytes, dict_id(k)); n_info++;
            std::vector<std::string> tok; if (!v.empty()) split(v, ',', tok);
            if (ti->second == HT_FLAG || tok.empty()) info_bytes += (char)0;
            else if (ti->second == HT_INT) {
                std::vector<int32_t> iv; for (auto& t : tok) iv.push_back(int_value(t));
                enc_vint(info_bytes, iv);
                if (k == "END" && !iv.empty() && iv[0] != I32_MISSING) rlen = iv[0] - pos0;
            } else if (ti->second == HT_FLOAT) {
                enc_size(info_bytes, (int)tok.size(), BT_FLOAT);
                for (auto& t : tok) put_u32(info_bytes, float_bits(t));
            } else enc_str(info_bytes, v);
        }
        put_u32(out, (uint32_t)c->second); put_u32(out, (uint32_t)pos0); put_u32(out, (uint32_t)rlen);
        put_u32(out, col[5] == "." ? F32_MISSING : float_bits(col[5]));
        put_u32(out, (uint32_t)alleles.size() << 16 | n_info);
        put_u32(out, n_fmt << 24 | (uint32_t)N);
        if (col[2] == ".") enc_size(out, 0, BT_CHAR); else enc_str(out, col[2]);     // htslib writes a missing ID as a zero-length string
        for (const std::string& a : alleles) enc_str(out, a);
        std::vector<int32_t> fv;
        if (col[6] != "." && !col[6].empty()) { split(col[6], ';', filt); for (auto& f : filt) fv.push_back(dict_id(f)); }
        enc_vint(out, fv);
        out += info_bytes;
    }

    // FORMAT column + sample columns of a text record (BCF2 section 6.3.3); returns the number of fields
    uint32_t encode_format_text(const std::vector<std::string>& col, std::string& out) const {
        std::vector<std::string> keys; split(col[8], ':', keys);
        if ((int)col.size() - 9 != N) fail("internal: record with %d sample columns, %d samples", (int)col.size() - 9, N);
        std::vector<std::vector<std::string>> smp(N);
        for (int s = 0; s < N; s++) split(col[9 + s], ':', smp[s]);
        std::vector<std::string> tok;
        for (size_t f = 0; f < keys.size(); f++) {
            const std::string& k = keys[f];
            enc_int1(out, dict_id(k));
            std::vector<std::vector<int32_t>> vals(N);      // int32 values or float bit patterns
            const bool is_gt = (k == "GT");
            auto ft = fmt_type.find(k);
            if (ft == fmt_type.end()) fail("FORMAT key %s is not defined in the output header", k.c_str());
            const bool is_float = !is_gt && ft->second == HT_FLOAT;
            if (!is_gt && ft->second != HT_INT && ft->second != HT_FLOAT) fail("FORMAT key %s: only Integer, Float and GT are written", k.c_str());
            size_t n = 0;
            for (int s = 0; s < N; s++) {
                const std::string v = f < smp[s].size() ? smp[s][f] : ".";
                if (is_gt) {                                 // (allele + 1) << 1 | phased
                    size_t b = 0;
                    while (b <= v.size()) {
                        const size_t e = v.find_first_of("|/", b);
                        const std::string a = v.substr(b, e == std::string::npos ? std::string::npos : e - b);
                        const int phased = (b > 0 && v[b - 1] == '|') ? 1 : 0;
                        vals[s].push_back(((a == "." || a.empty()) ? 0 : ((int32_t)strtol(a.c_str(), nullptr, 10) + 1) << 1) | phased);
                        if (e == std::string::npos) break;
                        b = e + 1;
                    }
                } else {
                    split(v, ',', tok);
                    for (auto& t : tok) vals[s].push_back(is_float ? (int32_t)float_bits(t) : int_value(t));
                }
                if (vals[s].size() > n) n = vals[s].size();
            }
            if (is_float) {
                enc_size(out, (int)n, BT_FLOAT);
                for (int s = 0; s < N; s++) for (size_t i = 0; i < n; i++) put_u32(out, i < vals[s].size() ? (uint32_t)vals[s][i] : F32_VEND);
            } else {
                int32_t mx = INT32_MIN, mn = INT32_MAX;
                for (int s = 0; s < N; s++) for (int32_t x : vals[s]) { if (x == I32_MISSING) continue; if (x > mx) mx = x; if (x < mn) mn = x; }
                const int bt = int_type(mn, mx);
                enc_size(out, (int)n, bt);
                for (int s = 0; s < N; s++) for (size_t i = 0; i < n; i++) put_typed_int(out, i < vals[s].size() ? vals[s][i] : I32_VEND, bt);
            }
        }
        return (uint32_t)keys.size();
    }
};

}  // namespace vsink
