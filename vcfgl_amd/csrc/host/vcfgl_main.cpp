// vcfgl_hip -- command-line front end with vcfgl's flag surface (io.cpp:538-752) over the C ABI
// of libvcfgl_hip.so.  It restates, on the host, what surrounds the hot path in the reference:
//   * the flag parser / defaults / range checks          io.cpp:428-526, 538-752, 757-1000
//   * the record loop incl. -explode and input filters   vcfgl.cpp:75-163, 1456-1639
//   * tag formatting of add_tags()                       bcf_utils.cpp:426-507
//   * VCF text in; VCF text / bgzip'd VCF / BCF out (vcf_sink.h; neither box has htslib): float
//     printing follows htslib's kputd() (6 significant digits, %g outside [1e-4, 999999]), so
//     `diff -I '^##'` against the reference's golden VCFs is empty in --rng-mode 1 (serial) runs.
// Records are batched into tiles and simulated on the GPU by vgl_simulate_tile(); there is no
//   * the gVCF block builder prepare_gvcf_block()         bcf_utils.cpp:662-942
//   * --depth inf (simulate_record_true_values, vcfgl.cpp:1089-1262): no sampling at all, the true
//     genotype gets GL 0 / GP 1 / PL 0 and every other genotype -inf / 0 / 255; written directly
//   * -printTruth 1: the decoded input records (incl. exploded ones) as <prefix>.truth.vcf
// CPU simulation path here.  Input: VCF text, gzip / bgzip'd VCF, BCF (raw or BGZF).
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <errno.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/vcfgl_hip.h"
#include "vcf_sink.h"

// Records are parsed, simulated and encoded on several threads, and any of them may hit a fatal input error: the first
// one reports and leaves through _exit() (no static destructors run under the feet of the threads still working, which is
// what exit() from two threads at once did), the others park.
#include <atomic>
#include <unistd.h>
static std::atomic<bool> g_dying{false};
[[noreturn]] static void die_v(const char* fmt, va_list ap) {
    if (g_dying.exchange(true)) for (;;) pause();
    fflush(stdout);
    fprintf(stderr, "\n\n*******\n[ERROR] "); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n*******\n");
    fflush(NULL);
    _exit(1);                                  // shared.h:292-299 exit(1)
}
[[noreturn]] static void die(const char* fmt, ...) { va_list ap; va_start(ap, fmt); die_v(fmt, ap); }
[[noreturn]] void vsink::fail(const char* fmt, ...) { va_list ap; va_start(ap, fmt); die_v(fmt, ap); }

// ---------------------------------------------------------------------------------------
struct Args {
    int seed = -1, source = 0, error_qs = 0, gl_model = 2, precise_gl = 0, i16_mapq = 20, adjust_qs = 0;
    int explode = 0, rm_invar = 0, rm_empty = 0, do_unobserved = 1, do_gvcf = 0, print_pileup = 0, print_truth = 0;
    int print_bpe = 0, print_qs_err = 0, print_gl_err = 0, print_qscores = 0;     // per-read / per-site TSV lines on stdout
    int add_gl = 1, add_gp = 0, add_pl = 0, add_i16 = 0, add_qs = 0, add_fmt_dp = 1, add_info_dp = 0;
    int add_fmt_ad = 0, add_info_ad = 0, add_fmt_adf = 0, add_info_adf = 0, add_fmt_adr = 0, add_info_adr = 0;
    int rng_mode = VGL_RNG_TILE, beta_sampler = -1, tile_sites = 4096, device = 0, verbose = 0, threads = 1, enc_threads = 0;
    bool threads_given = false;
    double depth = -1.0, error_rate = -1.0, beta_variance = -1.0, gl1_theta = 0.83, adjust_by = 0.499;
    bool have_depth = false, depth_inf = false;
    std::string in_fn, out_prefix = "output", output_mode = "b", depths_fn, qs_bins_fn, command;
    std::vector<double> depths;
    std::vector<int32_t> qs_bins;
    std::string gvcf_dps_str;
    std::vector<int> gvcf_dps;
    std::vector<int> devices;          // --devices 0,1,...: one context + host thread per GPU, tiles dealt round robin
};

static const char USAGE[] =
    "\nvcfgl_hip: genotype-likelihood simulation on an MI355X (vcfgl's flags; every flag takes one value)\n\n"
    "Usage: vcfgl_hip -i <in.vcf|vcf.gz|bcf> -e <error rate> -d <depth>|inf | -df <depths file> [options]\n\n"
    "  input / output   -i --input FILE    -o --output PREFIX [output]    -O --output-mode b|u|z|v [b]    --source 0|1 [0: binary alleles, 1: ACGT]\n"
    "                   -@ --threads INT [1]    -V --verbose INT [0]    -s --seed INT [time]\n"
    "  depth            -d --depth FLOAT|inf    -df --depths-file FILE (one mean depth per sample)\n"
    "  errors           -e --error-rate FLOAT    -eq --error-qs 0|1|2 [0]    -bv --beta-variance FLOAT    --qs-bins FILE (lo,hi,value per line)\n"
    "                   --adjust-qs 0..31 [0: bit 1 GL, 2 QS tag, 4 pileup, 8 -printQScores, 16 -printGlError]    --adjust-by FLOAT [0.499]\n"
    "  likelihoods      -GL --gl-model 1|2 [2]    --gl1-theta FLOAT [0.83]    --precise-gl 0|1 [0]    --i16-mapq INT [20]\n"
    "  sites            -explode 0|1 [0]    --rm-invar-sites 0..7 [0]    --rm-empty-sites 0|1 [0]    -doUnobserved 0..5 [1]\n"
    "                   -doGVCF 0|1 [0]    --gvcf-dps INT,INT,... (with -doGVCF 1)\n"
    "  tags             -addGL [1] -addGP [0] -addPL [0] -addI16 [0] -addQS [0] -addFormatDP [1] -addInfoDP [0]\n"
    "                   -addFormatAD -addInfoAD -addFormatADF -addInfoADF -addFormatADR -addInfoADR [0]\n"
    "  extra files      -printPileup 0|1 (<prefix>.pileup.gz)    -printTruth 0|1 (<prefix>.truth.*)\n"
    "  lines on stdout  -printBasePickError -printQsError -printGlError -printQScores 0|1\n"
    "  this program     --rng-mode 0|1 [0: counter-addressed windows of the rand48 sequence (fast, shards over GPUs);\n"
    "                                   1: the reference program's own draw order (reproduces its output)]\n"
    "                   --beta-sampler 0|1 [0: the rand48 sampler, 1: std::mt19937 (default with --rng-mode 1)]\n"
    "                   --tile-sites INT [4096]    --device INT [0]    --devices INT,INT,... (several GPUs of the node: sites shard by\n"
    "                   absolute index, the output does not depend on the device count; --rng-mode 0 only)    --encode-threads INT\n"
    "                   -v --version    -vv    -h --help\n\n";

static Args parse_args(int argc, char** argv) {
    Args a;
    a.command = "Command: vcfgl_hip";
    for (int i = 1; i < argc; i++) { a.command += " "; a.command += argv[i]; }
    auto I = [&](const char* v) { return atoi(v); };
    auto D = [&](const char* v) { return atof(v); };
    // io.cpp:538-752 compares the long simulation flags with strcasecmp (--error-qs, -addGL, -printPileup, -GL, -bv, -eq ...) and the
    // short / common ones (-s, -i, -o, -O, -d, -e, -V, -@ and their long forms) with strcmp: the former are matched in any case here too
    static const char* const nocase[] = {"--adjust-by", "--adjust-qs", "--beta-variance", "--error-qs", "--gl-model", "--gl1-theta", "--gvcf-dps",
        "--i16-mapq", "--precise-gl", "--qs-bins", "--rm-empty-sites", "--rm-invar-sites", "-GL", "-addFormatAD", "-addFormatADF", "-addFormatADR",
        "-addFormatDP", "-addFormatGL", "-addFormatGP", "-addFormatI16", "-addFormatPL", "-addFormatQS", "-addGL", "-addGP", "-addI16", "-addInfoAD",
        "-addInfoADF", "-addInfoADR", "-addInfoDP", "-addPL", "-addQS", "-bv", "-doGVCF", "-doUnobserved", "-eq", "-explode", "-printBasePickError",
        "-printGlError", "-printPileup", "-printQScores", "-printQsError", "-printTruth"};
    for (int i = 1; i < argc; i += 2) {
        std::string f = argv[i];
        for (const char* c : nocase) if (strcasecmp(c, f.c_str()) == 0) { f = c; break; }
        if (f == "-h" || f == "--help") { fputs(USAGE, stderr); exit(0); }
        if (f == "--version" || f == "-v") { fprintf(stderr, "vcfgl_hip [libvcfgl_hip ABI %d] [gfx950] [flag surface of vcfgl v1.3.0]\n\n", vgl_abi_version()); exit(0); }
        if (f == "-vv") { fprintf(stderr, "libvcfgl_hip ABI %d\n", vgl_abi_version()); exit(0); }
        if (i + 1 >= argc) die("Argument %s requires a value", argv[i]);
        const char* v = argv[i + 1];
        if (f == "--seed" || f == "-s") a.seed = I(v);
        else if (f == "--input" || f == "-i") a.in_fn = v;
        else if (f == "--source") a.source = I(v);
        else if (f == "--output" || f == "-o") a.out_prefix = v;
        else if (f == "--output-mode" || f == "-O") a.output_mode = v;
        else if (f == "--depth" || f == "-d") {
            if (!strcmp(v, "inf")) { a.depth_inf = true; a.depth = 0.0; }
            else a.depth = D(v);
            a.have_depth = true;
        } else if (f == "--depths-file" || f == "-df") a.depths_fn = v;
        else if (f == "--error-rate" || f == "-e") a.error_rate = D(v);
        else if (f == "--error-qs" || f == "-eq") a.error_qs = I(v);
        else if (f == "--beta-variance" || f == "-bv") a.beta_variance = D(v);
        else if (f == "--gl-model" || f == "-GL") a.gl_model = I(v);
        else if (f == "--gl1-theta") a.gl1_theta = D(v);
        else if (f == "--qs-bins") a.qs_bins_fn = v;
        else if (f == "--precise-gl") a.precise_gl = I(v);
        else if (f == "--i16-mapq") a.i16_mapq = I(v);
        else if (f == "--gvcf-dps") a.gvcf_dps_str = v;
        else if (f == "--adjust-qs") a.adjust_qs = I(v);
        else if (f == "--adjust-by") a.adjust_by = D(v);
        else if (f == "-explode") a.explode = I(v);
        else if (f == "--rm-invar-sites") a.rm_invar = I(v);
        else if (f == "--rm-empty-sites") a.rm_empty = I(v);
        else if (f == "-doUnobserved") a.do_unobserved = I(v);
        else if (f == "-doGVCF") a.do_gvcf = I(v);
        else if (f == "-printPileup") a.print_pileup = I(v);
        else if (f == "-printTruth") a.print_truth = I(v);
        else if (f == "-printBasePickError") a.print_bpe = I(v);
        else if (f == "-printQsError") a.print_qs_err = I(v);
        else if (f == "-printGlError") a.print_gl_err = I(v);
        else if (f == "-printQScores") a.print_qscores = I(v);
        else if (f == "-addGL" || f == "-addFormatGL") a.add_gl = I(v);
        else if (f == "-addGP" || f == "-addFormatGP") a.add_gp = I(v);
        else if (f == "-addPL" || f == "-addFormatPL") a.add_pl = I(v);
        else if (f == "-addI16" || f == "-addFormatI16") a.add_i16 = I(v);
        else if (f == "-addQS" || f == "-addFormatQS") a.add_qs = I(v);
        else if (f == "-addFormatDP") a.add_fmt_dp = I(v);
        else if (f == "-addInfoDP") a.add_info_dp = I(v);
        else if (f == "-addFormatAD") a.add_fmt_ad = I(v);
        else if (f == "-addInfoAD") a.add_info_ad = I(v);
        else if (f == "-addFormatADF") a.add_fmt_adf = I(v);
        else if (f == "-addInfoADF") a.add_info_adf = I(v);
        else if (f == "-addFormatADR") a.add_fmt_adr = I(v);
        else if (f == "-addInfoADR") a.add_info_adr = I(v);
        else if (f == "--verbose" || f == "-V") a.verbose = I(v);
        else if (f == "--threads" || f == "-@") { a.threads = I(v); a.threads_given = true; }
        else if (f == "--encode-threads") a.enc_threads = I(v);      // extension: record-encoding threads (any output mode)
        // extensions of this implementation
        else if (f == "--rng-mode") a.rng_mode = I(v);
        else if (f == "--beta-sampler") a.beta_sampler = I(v);
        else if (f == "--tile-sites") a.tile_sites = I(v);
        else if (f == "--device") a.device = I(v);
        else if (f == "--devices") { a.devices.clear(); for (const char* q = v; *q;) { char* e; const long d = strtol(q, &e, 10); if (e == q || d < 0) die("Could not parse --devices %s", v); a.devices.push_back((int)d); q = (*e == ',') ? e + 1 : e; if (*e && *e != ',') die("Could not parse --devices %s", v); } }
        else die("Unknown argument: %s", argv[i]);
    }
    // ---- validation (io.cpp:757-1000, the rules that concern the hot path)
    auto range = [&](double v, double lo, double hi, const char* s) { if (v < lo || v > hi) die("[Bad argument value: '%s %g'] Allowed range is [%g,%g]", s, v, lo, hi); };
    if (a.in_fn.empty()) die("Input file is not specified. Please use -i/--input option to specify the input file.");
    if (!a.have_depth && a.depths_fn.empty()) die("Average per-site read depth value is required. Please set it using --depth or --depths-file and re-run.");
    if (a.depths_fn.empty()) range(a.depth, 0.0, 500.0, "--depth");
    if (a.depth_inf) {                                                          // io.cpp:781-850, 1011-1018
        if (a.rm_invar & 4) die("[--rm-invar-sites %d] Cannot skip invariable sites when --depth inf is set.", a.rm_invar);
        if (a.do_gvcf) die("[-doGVCF 1] Cannot output gVCF when --depth inf is set.");
        if (a.add_qs) die("(-addQS 1) QS tag cannot be added when --depth inf is set.");
        if (a.add_i16) die("(-addI16 1) I16 tag cannot be added when --depth inf is set.");
    }
    if (a.error_rate < 0) die("Error rate is not specified. Please use --error-rate option to specify the error rate. Allowed range: [0.0, 1.0]");
    if (a.error_rate >= 1.0) die("[Bad argument value: '--error-rate %f'] Allowed range is [0.0,1.0]", a.error_rate);
    range(a.source, 0, 1, "--source"); range(a.error_qs, 0, 2, "--error-qs"); range(a.gl_model, 1, 2, "--gl-model");
    range(a.gl1_theta, 0, 1, "--gl1-theta"); range(a.precise_gl, 0, 1, "--precise-gl"); range(a.i16_mapq, 0, 60, "--i16-mapq");
    range(a.adjust_qs, 0, 31, "--adjust-qs"); range(a.do_unobserved, 0, 5, "-doUnobserved"); range(a.rm_invar, 0, 7, "--rm-invar-sites");
    if (a.adjust_qs && a.adjust_by == 0.0) die("--adjust-qs %d requires a non-zero value for --adjust-by. Please set --adjust-by and rerun.", a.adjust_qs);
    if ((a.adjust_qs & 1) && a.precise_gl) die("--adjust-qs 1 requires --precise-gl 0. Please set --precise-gl 0 and rerun.");
    if ((a.adjust_qs & 2) && !a.add_qs) die("--adjust-qs 2 requires -addQS 1. Please set -addQS 1 and rerun.");
    if ((a.adjust_qs & 4) && !a.print_pileup) die("--adjust-qs 4 requires --printPileup 1. Please set --printPileup 1 and rerun.");   // io.cpp:891-898
    if ((a.adjust_qs & 8) && !a.print_qscores) die("--adjust-qs 8 requires --printQScores 1. Please set --printQScores and rerun.");
    if ((a.adjust_qs & 16) && !a.print_gl_err) die("--adjust-qs 16 requires --printGlError 1. Please set --printGlError 1 and rerun.");
    range(a.print_pileup, 0, 1, "-printPileup"); range(a.print_truth, 0, 1, "-printTruth"); range(a.print_bpe, 0, 1, "-printBasePickError");
    range(a.print_qs_err, 0, 1, "-printQsError"); range(a.print_gl_err, 0, 1, "-printGlError"); range(a.print_qscores, 0, 1, "-printQScores");
    if (a.print_gl_err && a.gl_model == 1)                                                                                              // io.cpp:993
        die("-> [-printGlError 1] Printing the error probability used in genotype likelihood calculations (-printGlError 1) is not supported with genotype likelihood model 1 (--gl-model 1).");
    if (a.gl_model == 1 && a.precise_gl) die("Precise genotype likelihood error (--precise-gl 1) is not supported with genotype likelihood model 1 (--gl-model 1).");
    if (a.error_qs == 0 && a.beta_variance >= 0) die("--beta-variance %e requires --error-qs 1 or 2.", a.beta_variance);
    if (a.error_qs != 0 && !(a.error_rate > 0)) die("--error-qs 1 or 2 requires --error-rate > 0 (found %f).", a.error_rate);
    if (a.error_qs != 0 && !(a.beta_variance > 0)) die("--error-qs 1 or 2 requires --beta-variance > 0 (found %e).", a.beta_variance);
    if (a.do_gvcf == 1) {                                                       // io.cpp:958-985
        if (!a.add_fmt_dp) die("[-doGVCF 1] -addFormatDP 1 is required for gVCF output. Please set -addFormatDP 1 and rerun.");
        if (a.rm_invar != 0) die("-> [-doGVCF 1] --rm-invar-sites 0 is required. Please set --rm-invar-sites 0 and rerun.");
        if (a.gvcf_dps_str.empty()) die("-> [-doGVCF 1] --gvcf-dps is required. Please set --gvcf-dps and rerun.");
        if (!(a.do_unobserved == 1 || a.do_unobserved == 2 || a.do_unobserved == 4 || a.do_unobserved == 5))
            die("-> [-doGVCF 1] Adding unobserved alleles is required for gVCF output. Please set -doUnobserved to 1 or 2 and rerun.");
        if (!a.add_pl) die("-> [-doGVCF 1] -addPL 1 is required for gVCF output. Please set -addPL 1 and rerun.");
        std::vector<std::string> parts; std::string cur;                          // gvcfData_init, bcf_utils.cpp:946-985
        for (char ch : a.gvcf_dps_str) { if (ch == ',') { parts.push_back(cur); cur.clear(); } else cur += ch; }
        parts.push_back(cur);
        for (auto& x : parts) { if (x.empty()) die("Could not parse --gvcf-dps %s", a.gvcf_dps_str.c_str()); const int d = atoi(x.c_str()); if (d < 1) die("Invalid DP range: %d", d); a.gvcf_dps.push_back(d); }
    } else if (!a.gvcf_dps_str.empty()) die("-> [--gvcf-dps] --gvcf-dps requires -doGVCF 1. Please set -doGVCF 1 and rerun.");
    if (a.output_mode != "v" && a.output_mode != "z" && a.output_mode != "u" && a.output_mode != "b")
        die("[Bad argument value: '--output-mode %s'] Allowed values are b, u, z, v", a.output_mode.c_str());
    if ((a.output_mode == "v" || a.output_mode == "z") && a.threads > 1)                      // io.cpp:1206-1210
        die("Multithreading is not supported for VCF output. Please set --threads 1 and rerun.");
    if (a.seed == -1) { a.seed = (int)time(NULL); fprintf(stderr, "\n-> No seed was given. Setting the random seed to the randomly chosen value: %d\n", a.seed); }
    if (a.beta_sampler < 0) a.beta_sampler = (a.rng_mode == VGL_RNG_SERIAL) ? VGL_BETA_STD : VGL_BETA_RAND48;
    if (!a.depths_fn.empty()) {
        FILE* fp = fopen(a.depths_fn.c_str(), "r"); if (!fp) die("Could not open file: %s", a.depths_fn.c_str());
        double d; while (fscanf(fp, "%lf", &d) == 1) a.depths.push_back(d);
        fclose(fp);
    }
    if (!a.qs_bins_fn.empty()) {
        FILE* fp = fopen(a.qs_bins_fn.c_str(), "r"); if (!fp) die("Could not open file: %s", a.qs_bins_fn.c_str());
        int x, y, z; while (fscanf(fp, "%d,%d,%d", &x, &y, &z) == 3) { a.qs_bins.push_back(x); a.qs_bins.push_back(y); a.qs_bins.push_back(z); }
        fclose(fp);
    }
    return a;
}

// ---------------------------------------------------------------------------------------
// htslib kputd(): floats of VCF text.  0 -> "0"; outside [1e-4, 999999] -> "%g"; otherwise
// trunc(d*1e10) plus half a unit of the 6th significant digit, cut to 6 significant digits,
// trailing zeros removed.
static void put_float(std::string& s, float f) {
    uint32_t bits; memcpy(&bits, &f, 4);
    if (bits == VGL_FLOAT_MISSING_BITS) { s += '.'; return; }
    double d = f;
    if (isnan(d)) { s += "nan"; return; }
    if (d == 0) { s += signbit(d) ? "-0" : "0"; return; }
    if (d < 0) { s += '-'; d = -d; }
    char buf[64];
    if (!(d >= 0.0001 && d <= 999999)) { snprintf(buf, sizeof buf, "%g", d); s += buf; return; }
    uint64_t i = (uint64_t)(d * 10000000000LL);
    if (d < .0001) i += 0; else if (d < 0.001) i += 5; else if (d < 0.01) i += 50; else if (d < 0.1) i += 500;
    else if (d < 1) i += 5000; else if (d < 10) i += 50000; else if (d < 100) i += 500000; else if (d < 1000) i += 5000000;
    else if (d < 10000) i += 50000000; else if (d < 100000) i += 500000000; else i += 5000000000LL;
    char dig[32]; int n = snprintf(dig, sizeof dig, "%llu", (unsigned long long)i);   // d*1e10 as an integer
    std::string out;
    if (n <= 10) {                       // d < 1: "0." + leading zeros + 6 significant digits
        out = "0.";
        out.append(10 - n, '0');
        out.append(dig, n < 6 ? n : 6);
    } else {                             // integer part has n-10 digits; 6 significant digits in all
        const int ip = n - 10;
        out.append(dig, ip);
        if (ip < 6) { out += '.'; out.append(dig + ip, 6 - ip); }
    }
    if (out.find('.') != std::string::npos) {
        while (out.back() == '0') out.pop_back();
        if (out.back() == '.') out.pop_back();
    }
    s += out;
}

static void put_int(std::string& s, int32_t v) {
    if (v == VGL_INT32_MISSING) { s += '.'; return; }
    char buf[16]; snprintf(buf, sizeof buf, "%d", v); s += buf;
}

// ---------------------------------------------------------------------------------------
struct Rec {
    std::string chrom, id, qual, filt, info;
    long pos0;
    std::vector<std::string> alleles;
    std::vector<int8_t> gt;            // 2 per sample, allele index or -1
    std::vector<std::string> gt_str;   // GT tokens as written in the input (for -printTruth)
    char ref_char;
};

struct Vcf {
    std::vector<std::string> header;   // '##' lines
    std::vector<std::string> samples;
    std::map<std::string, long> contig_len;
    std::vector<Rec> recs;
};

static void split(const std::string& s, char c, std::vector<std::string>& out) {
    out.clear(); size_t b = 0;
    while (true) { size_t e = s.find(c, b); if (e == std::string::npos) { out.push_back(s.substr(b)); break; } out.push_back(s.substr(b, e - b)); b = e + 1; }
}

// ---------------------------------------------------------------------------------------
// BCF 2.x input (raw or BGZF; zlib reads the gzip members): header text, then records decoded back
// into the same Rec the text reader fills -- QUAL / FILTER / INFO as VCF text, GT as allele indices.
struct BcfIn {
    std::vector<uint8_t> buf; size_t off = 0;
    std::map<int, std::string> dict, contig; std::map<std::string, int> info_is_flag;
    uint32_t u32() { if (off + 4 > buf.size()) die("truncated BCF record"); uint32_t v; memcpy(&v, &buf[off], 4); off += 4; return v; }
    void typed(int& type, int& n) {
        if (off >= buf.size()) die("truncated BCF record");
        const uint8_t b = buf[off++]; type = b & 15; n = b >> 4;
        if (n == 15) { int t2, n2; typed(t2, n2); std::vector<int32_t> v; ints(t2, n2, v); if (v.empty() || v[0] < 0) die("bad BCF vector length"); n = v[0]; }
    }
    void ints(int type, int n, std::vector<int32_t>& out) {          // missing -> INT32_MIN, end-of-vector -> INT32_MIN + 1
        const int w = type == 1 ? 1 : type == 2 ? 2 : type == 3 ? 4 : 0;
        if (!w) die("BCF: integer vector expected (type %d)", type);
        if (off + (size_t)w * n > buf.size()) die("truncated BCF record");
        out.clear();
        for (int i = 0; i < n; i++, off += w) {
            int32_t v;
            if (w == 1) { const int8_t x = (int8_t)buf[off]; v = x == -128 ? INT32_MIN : x == -127 ? INT32_MIN + 1 : x; }
            else if (w == 2) { int16_t x; memcpy(&x, &buf[off], 2); v = x == -32768 ? INT32_MIN : x == -32767 ? INT32_MIN + 1 : x; }
            else memcpy(&v, &buf[off], 4);
            out.push_back(v);
        }
    }
    std::string str(int n) { if (off + n > buf.size()) die("truncated BCF record"); std::string r((const char*)&buf[off], n); off += n; return r; }
    static std::string attr(const std::string& h, const char* key) {
        const std::string k = std::string(key) + "=";
        size_t a = h.find("<" + k); if (a == std::string::npos) a = h.find("," + k); if (a == std::string::npos) return "";
        a += k.size() + 1;
        return h.substr(a, h.find_first_of(",>", a) - a);
    }
};

static Vcf read_bcf(std::vector<uint8_t>&& raw, bool keep_gt_text) {
    BcfIn in; in.buf = std::move(raw);
    if (in.buf.size() < 9 || memcmp(in.buf.data(), "BCF\2", 4) != 0) die("not a BCF2 file");
    in.off = 5;
    const uint32_t l_text = in.u32();
    if (in.off + l_text > in.buf.size()) die("truncated BCF header");
    std::string text((const char*)&in.buf[in.off], l_text); in.off += l_text;
    while (!text.empty() && (text.back() == '\0' || text.back() == '\n')) text.pop_back();
    Vcf v; std::vector<std::string> lines, f; split(text, '\n', lines);
    int nd = 0, nc = 0; bool have_pass = false;
    for (const std::string& h : lines) if (h.compare(0, 10, "##FILTER=<") == 0 && BcfIn::attr(h, "ID") == "PASS") have_pass = true;
    if (!have_pass) { in.dict[0] = "PASS"; nd = 1; }
    for (const std::string& h : lines) {
        if (h.compare(0, 2, "##") != 0) { if (!h.empty() && h[0] == '#') { split(h, '\t', f); for (size_t i = 9; i < f.size(); i++) v.samples.push_back(f[i]); } continue; }
        v.header.push_back(h);
        const bool fil = h.compare(0, 10, "##FILTER=<") == 0, inf = h.compare(0, 8, "##INFO=<") == 0, fmt = h.compare(0, 10, "##FORMAT=<") == 0;
        const bool ctg = h.compare(0, 10, "##contig=<") == 0;
        if (!(fil || inf || fmt || ctg)) continue;
        const std::string id = BcfIn::attr(h, "ID"), idx_s = BcfIn::attr(h, "IDX");
        if (ctg) {
            const int idx = idx_s.empty() ? nc : atoi(idx_s.c_str());
            in.contig[idx] = id; nc = std::max(nc, idx + 1);
            const std::string len = BcfIn::attr(h, "length"); v.contig_len[id] = len.empty() ? -1 : atol(len.c_str());
            continue;
        }
        int idx = -1;
        for (auto& kv : in.dict) if (kv.second == id) idx = kv.first;
        if (idx < 0) idx = idx_s.empty() ? nd : atoi(idx_s.c_str());
        in.dict[idx] = id; nd = std::max(nd, idx + 1);
        if (inf) in.info_is_flag[id] = BcfIn::attr(h, "Type") == "Flag";
    }
    const size_t N = v.samples.size();
    std::vector<int32_t> iv; int type, n;
    while (in.off < in.buf.size()) {
        const uint32_t l_shared = in.u32(), l_indiv = in.u32();
        const size_t rec_end = in.off + (size_t)l_shared + l_indiv, shared_end = in.off + l_shared;
        if (rec_end > in.buf.size()) die("truncated BCF record");
        Rec r;
        const int32_t chrom = (int32_t)in.u32(); r.pos0 = (int32_t)in.u32(); (void)in.u32();
        const uint32_t qual = in.u32(), nai = in.u32(), nfs = in.u32();
        const int n_allele = nai >> 16, n_info = nai & 0xFFFF, n_fmt = nfs >> 24; const size_t n_sample = nfs & 0xFFFFFF;
        if (!in.contig.count(chrom)) die("BCF record with an undefined contig index %d", chrom);
        if (n_sample != N) die("Record at position %ld has %zu samples, the header names %zu samples.", r.pos0 + 1, n_sample, N);
        r.chrom = in.contig[chrom];
        if (qual == VGL_FLOAT_MISSING_BITS) r.qual = "."; else { float q; memcpy(&q, &qual, 4); put_float(r.qual, q); }
        in.typed(type, n); r.id = (type == 7) ? in.str(n) : "."; if (r.id.empty()) r.id = ".";
        for (int i = 0; i < n_allele; i++) { in.typed(type, n); if (type != 7) die("BCF: allele string expected"); r.alleles.push_back(in.str(n)); }
        if (r.alleles.empty() || r.alleles[0].empty()) die("Empty REF at position %ld.", r.pos0 + 1);
        r.ref_char = r.alleles[0][0];
        in.typed(type, n);
        if (type == 0 || n == 0) r.filt = "."; else { in.ints(type, n, iv); for (size_t i = 0; i < iv.size(); i++) { if (i) r.filt += ';'; if (!in.dict.count(iv[i])) die("BCF: undefined FILTER index"); r.filt += in.dict[iv[i]]; } }
        for (int i = 0; i < n_info; i++) {
            in.typed(type, n); in.ints(type, n, iv);
            if (iv.empty() || !in.dict.count(iv[0])) die("BCF: undefined INFO key");
            const std::string key = in.dict[iv[0]];
            if (!r.info.empty()) r.info += ';';
            r.info += key;
            in.typed(type, n);
            if (type == 0 || n == 0) continue;                              // flag
            r.info += '=';
            if (type == 7) r.info += in.str(n);
            else if (type == 5) { for (int k = 0; k < n; k++) { const uint32_t b = in.u32(); if (b == 0x7F800002u) continue; if (k) r.info += ','; float x; memcpy(&x, &b, 4); put_float(r.info, x); } }
            else { in.ints(type, n, iv); bool first = true; for (int32_t x : iv) { if (x == INT32_MIN + 1) continue; if (!first) r.info += ','; first = false; put_int(r.info, x); } }
        }
        if (r.info.empty()) r.info = ".";
        if (in.off != shared_end) die("BCF record: shared block length mismatch at position %ld", r.pos0 + 1);
        bool have_gt = false;
        r.gt.assign(2 * N, -1);
        for (int k = 0; k < n_fmt; k++) {
            in.typed(type, n); in.ints(type, n, iv);
            if (iv.empty() || !in.dict.count(iv[0])) die("BCF: undefined FORMAT key");
            const bool is_gt = in.dict[iv[0]] == "GT";
            in.typed(type, n);
            const size_t w = type == 1 ? 1 : type == 2 ? 2 : (type == 3 || type == 5) ? 4 : type == 7 ? 1 : 0;
            if (!is_gt) { if (in.off + w * n * N > in.buf.size()) die("truncated BCF record"); in.off += w * n * N; continue; }
            have_gt = true;
            for (size_t s = 0; s < N; s++) {
                in.ints(type, n, iv);
                int8_t a[2] = {-1, -1}; std::string txt;
                for (int j = 0; j < n && iv[j] != INT32_MIN + 1; j++) {
                    const int al = (iv[j] >> 1) - 1;
                    if (j < 2) a[j] = (int8_t)al;
                    if (keep_gt_text) { if (j) txt += (iv[j] & 1) ? '|' : '/'; if (al < 0) txt += '.'; else { char t[16]; snprintf(t, sizeof t, "%d", al); txt += t; } }
                }
                if (n == 1 || (n >= 2 && iv[1] == INT32_MIN + 1)) a[1] = a[0];     // haploid call: both alleles, as the text reader does
                r.gt[2 * s] = a[0]; r.gt[2 * s + 1] = a[1];
                if (keep_gt_text) r.gt_str.push_back(txt.empty() ? "." : txt);
            }
        }
        if (!have_gt) die("Could not find GT tag at position %ld.", r.pos0 + 1);
        in.off = rec_end;
        v.recs.push_back(std::move(r));
    }
    return v;
}

// keep_gt_text: the GT tokens as written are needed only by -printTruth
// one record line [lb, le) -> Rec (thread safe: records are parsed in parallel)
static void parse_record(const char* lb, const char* le, const size_t n_hdr, const bool keep_gt_text, Rec& r) {
    const char* col_at[10]; int nc = 0; col_at[0] = lb;
    for (const char* q = lb; q < le && nc < 9; q++) if (*q == '\t') col_at[++nc] = q + 1;
    if (nc < 9) die("VCF record with fewer than 10 columns (a FORMAT/GT column is required)");
    auto col = [&](int k) { return std::string(col_at[k], (size_t)(col_at[k + 1] - 1 - col_at[k])); };
    std::vector<std::string> g, fmt;
    r.chrom = col(0); r.pos0 = atol(col(1).c_str()) - 1; r.id = col(2); r.qual = col(5); r.filt = col(6); r.info = col(7);
    const std::string ref = col(3), alt = col(4);
    r.alleles.push_back(ref);
    if (alt != ".") { split(alt, ',', g); for (auto& x : g) r.alleles.push_back(x); }
    if (ref.empty()) die("Empty REF at position %ld.", r.pos0 + 1);
    r.ref_char = ref[0];
    split(col(8), ':', fmt);
    int gti = -1; for (size_t i = 0; i < fmt.size(); i++) if (fmt[i] == "GT") gti = (int)i;
    if (gti < 0) die("Could not find GT tag at position %ld.", r.pos0 + 1);
    r.gt.assign(2 * n_hdr, -1);
    const char* p = col_at[9];
    const char* const end = le;
    size_t s = 0;
    while (true) {                                       // p at the start of a sample column
        const char* ce = (const char*)memchr(p, '\t', (size_t)(end - p)); if (!ce) ce = end;
        if (s >= n_hdr) { s++; if (ce == end) break; p = ce + 1; continue; }
        const char* t = p;                               // the gti-th ':'-separated subfield; trailing ones may be dropped
        for (int k = 0; k < gti && t; k++) { t = (const char*)memchr(t, ':', (size_t)(ce - t)); if (t) t++; }
        const char* te = t ? (const char*)memchr(t, ':', (size_t)(ce - t)) : nullptr; if (t && !te) te = ce;
        int8_t a0 = -1, a1 = -1;
        if (t) {
            const char* sep = t; while (sep < te && *sep != '|' && *sep != '/') sep++;
            auto allele = [](const char* b, const char* e) -> int8_t { return (b == e || *b == '.') ? (int8_t)-1 : (int8_t)atoi(std::string(b, e).c_str()); };
            a0 = allele(t, sep);
            a1 = (sep == te) ? a0 : allele(sep + 1, te);
            if (keep_gt_text) r.gt_str.emplace_back(t, te);
        } else if (keep_gt_text) r.gt_str.emplace_back(".");
        r.gt[2 * s] = a0; r.gt[2 * s + 1] = a1;
        s++;
        if (ce == end) break;
        p = ce + 1;
    }
    if (s != n_hdr) die("Record at position %ld has %zu sample columns, the header names %zu samples.", r.pos0 + 1, s, n_hdr);
}

// keep_gt_text: the GT tokens as written are needed only by -printTruth.  The (decompressed) file is read whole,
// the header lines are taken in order and the record lines are parsed on `threads` threads.
static Vcf read_vcf(const std::string& fn, bool keep_gt_text, int threads) {
    gzFile fp = gzopen(fn.c_str(), "r");               // plain text, gzip / BGZF, or BCF inside either
    if (!fp) die("Could not open file: %s", fn.c_str());
    gzbuffer(fp, 1 << 20);
    std::vector<uint8_t> raw, chunk(1 << 22);
    int k;
    while ((k = gzread(fp, chunk.data(), (unsigned)chunk.size())) > 0) raw.insert(raw.end(), chunk.begin(), chunk.begin() + k);
    gzclose(fp);
    if (raw.size() >= 3 && !memcmp(raw.data(), "BCF", 3)) return read_bcf(std::move(raw), keep_gt_text);
    Vcf v;
    std::vector<std::string> f;
    std::vector<std::pair<const char*, const char*>> rec_lines;
    const char* p = (const char*)raw.data(); const char* const end = p + raw.size();
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        if (le > p) {
            if (le - p >= 2 && p[0] == '#' && p[1] == '#') {
                const std::string line(p, le);
                v.header.push_back(line);
                if (line.compare(0, 10, "##contig=<") == 0) {
                    size_t a = line.find("ID="), l = line.find("length=");
                    if (a != std::string::npos) {
                        std::string id = line.substr(a + 3, line.find_first_of(",>", a) - a - 3);
                        v.contig_len[id] = (l != std::string::npos) ? atol(line.c_str() + l + 7) : -1;
                    }
                }
            } else if (p[0] == '#') { split(std::string(p, le), '\t', f); for (size_t i = 9; i < f.size(); i++) v.samples.push_back(f[i]); }
            else rec_lines.emplace_back(p, le);
        }
        p = nl ? nl + 1 : end;
    }
    v.recs.resize(rec_lines.size());
    const size_t n_hdr = v.samples.size();
    vsink::parallel_for((int)rec_lines.size(), threads, [&](int i) { parse_record(rec_lines[i].first, rec_lines[i].second, n_hdr, keep_gt_text, v.recs[i]); });
    return v;
}

// allele_char_to_int, vcfgl.cpp:20-50
static int allele_to_int(const std::string& a) {
    if (a.size() > 1) return (a == "<*>" || a == "<NON_REF>") ? 4 : -1;
    switch (a[0]) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

// ---------------------------------------------------------------------------------------
// qScores on the host, for the TSV lines of -printQsError / -printGlError / -printQScores and the adjusted
// pileup (--adjust-qs 4): the library hands over the deviates, this is vcfgl.cpp:494-523 / :1664-1693 on them.
static void host_errprob_to_qs(const Args& a, double ep, int& q, int& aq) {
    q = -1; aq = -1;
    if (0.0 == ep) { q = 63; if (a.error_qs != 2) aq = 63; }                   // CAP_BASEQ; the preCalc block also sets adjqs (:1668-1673)
    else if (1.0 == ep) { q = 0; if (a.error_qs != 2) aq = 0; }
    else if (0.0 < ep && ep < 1.0) {
        const double tmp = -10.0 * log10(ep);
        q = (int)tmp;
        if (a.adjust_qs) aq = (int)(tmp + a.adjust_by);
    } else die("Bad error probability value: %f", ep);
    auto bins = [&](int v) {                                                    // apply_qs_bins, vcfgl.cpp:57-64
        for (size_t i = 0; i + 2 < a.qs_bins.size(); i += 3) if (v >= a.qs_bins[i] && v <= a.qs_bins[i + 1]) return (int)a.qs_bins[i + 2];
        die("Could not find a range for the simulated qs value: %d", v);
        return 0;
    };
    if (!a.qs_bins.empty()) { q = bins(q); if (a.adjust_qs) aq = bins(aq); }
    else { q = q > 63 ? 63 : q; if (a.adjust_qs) aq = aq > 63 ? 63 : aq; }
}
// QS_TO_ERRPROB (shared.h:493): the table shared.cpp:31 holds 10^(-q/10) at 7 significant digits (checked for
// every entry by tests/test_cli_format_cpu.py against the reference's own table)
static double host_qs_to_errprob(int q) {
    if (q == 0) return 1.0;
    if (q >= 63) return 0.0000005011872;
    char b[40]; snprintf(b, sizeof b, "%.7g", pow(10.0, -(double)q / 10.0));
    return strtod(b, nullptr);
}

// One site in simulation order.  Its contig is rec->chrom: a record that -explode 1 synthesises keeps the contig of the
// template record it was copied from (reference quirk, bcf_copy at vcfgl.cpp:1490, visible in test/reference/test18).
struct SiteMeta { const Rec* rec; long pos0; char ref_char; };

// check_rec_alleles (vcfgl.cpp:75-163) + the n_allele==1 filter (vcfgl.cpp:335-338); false = skipped.
// gt_row[N] receives the packed true genotypes; truth_line (when given) the record as -printTruth writes it.
static bool make_site(const Args& a, const Rec& rec, long pos0, bool blank, int N, uint8_t* gt_row, SiteMeta& out, std::string* truth_line) {
    // (the n_allele == 1 filter below belongs to simulate_record_values and is not applied with --depth inf)
    const int n_alleles = (int)rec.alleles.size();
    if (n_alleles > 5) die("Multiallelic sites with more than 4 alleles are not supported.");
    int ra[5] = {-1, -1, -1, -1, -1};
    for (int i = 0; i < n_alleles; i++) {
        if (a.source == 1) { ra[i] = allele_to_int(rec.alleles[i]); if (ra[i] == -1) die("Allele '%s' at position %ld is not a valid base.", rec.alleles[i].c_str(), pos0 + 1); }
        else {
            const int x = rec.alleles[i][0] - '0';
            if (x != 0 && x != 1) die("[--source %d] Found allele '%s' at position %ld. Only 0 and 1 are allowed when using binary GT source.", a.source, rec.alleles[i].c_str(), pos0 + 1);
            ra[i] = x;
        }
    }
    if (a.source == 0 && n_alleles > 2) die("Multiallelic sites are not supported when using binary GT source.");
    long allelesum = 0;
    for (int s = 0; s < N; s++) {
        int g0 = blank ? 0 : rec.gt[2 * s], g1 = blank ? 0 : rec.gt[2 * s + 1];
        int b0 = 0xF, b1 = 0xF;
        if (g0 >= 0) { if (g0 >= n_alleles) die("GT allele index out of range at position %ld", pos0 + 1); allelesum += g0; b0 = ra[g0] & 0xF; }
        if (g1 >= 0) { if (g1 >= n_alleles) die("GT allele index out of range at position %ld", pos0 + 1); allelesum += g1; b1 = ra[g1] & 0xF; }
        gt_row[s] = (uint8_t)(b0 | (b1 << 4));
    }
    if ((a.rm_invar & 1) && allelesum == 0) return false;
    if (a.rm_invar & 2) for (int k = 1; k < n_alleles; k++) if ((long)k * N * 2 == allelesum) return false;
    if (truth_line) {                                  // bcf_write(out_truth_fp, ...) at vcfgl.cpp:1518,1548,1607
        std::string& l = *truth_line;
        l = rec.chrom; char hb[48]; snprintf(hb, sizeof hb, "\t%ld\t", pos0 + 1); l += hb; l += rec.id; l += '\t';
        if (a.source == 0) l += "A\tC";                // binary source: alleles become A,C (vcfgl.cpp:127)
        else { l += rec.alleles[0]; l += '\t'; if (n_alleles == 1) l += '.'; else for (int k = 1; k < n_alleles; k++) { if (k > 1) l += ','; l += rec.alleles[k]; } }
        l += '\t'; l += rec.qual; l += '\t'; l += rec.filt; l += '\t'; l += rec.info; l += "\tGT";
        for (int s = 0; s < N; s++) { l += '\t'; l += blank ? std::string("0|0") : rec.gt_str[s]; }
    }
    if (!a.depth_inf && (a.rm_invar & 3) && n_alleles == 1) return false;
    out.rec = &rec; out.pos0 = pos0; out.ref_char = (a.source == 0) ? 'A' : rec.ref_char;
    return true;
}

// main_simulate_record_values (vcfgl.cpp:1456-1639): the sites in simulation order, produced one at a time -- the sites that
// -explode 1 adds are never materialised (BASELINE config C5: 50M exploded sites x 500 samples), the truth file is written
// as the sites go by.
struct SiteStream {
    const Args& a; const Vcf& v; const int N;
    vsink::Sink* truth = nullptr;
    size_t ri = 0; const Rec* tpl = nullptr; std::string last; long n_in = 0, tail_size = -1; bool tail = false, done = false;
    std::string tl;
    SiteStream(const Args& a_, const Vcf& v_, int N_) : a(a_), v(v_), N(N_) {}
    bool emit(const Rec& r, long pos0, bool blank, uint8_t* gt_row, SiteMeta& m) {
        tl.clear();
        const bool ok = make_site(a, r, pos0, blank, N, gt_row, m, truth ? &tl : nullptr);
        if (truth && !tl.empty()) truth->write_line(tl);
        return ok;
    }
    bool next(uint8_t* gt_row, SiteMeta& m) {
        while (!done) {
            if (!tail) {
                if (ri == v.recs.size()) {
                    if (a.explode == 1 && !v.recs.empty()) {           // to the end of the last contig (vcfgl.cpp:1565-1625)
                        auto it = v.contig_len.find(v.recs.back().chrom);
                        tail_size = (it == v.contig_len.end()) ? -1 : it->second;
                        tail = true;
                        continue;
                    }
                    done = true; break;
                }
                const Rec& r = v.recs[ri];
                if (r.chrom != last) { n_in = 0; last = r.chrom; }
                if (a.explode == 1 && n_in != r.pos0) {
                    // the reference's `while (n_in != pos)` (vcfgl.cpp:1481) never ends on such input
                    if (r.pos0 < n_in) die("[-explode 1] Record %s:%ld is not after the previous record of its contig (duplicate or unsorted positions cannot be exploded).", r.chrom.c_str(), r.pos0 + 1);
                    if (!tpl) tpl = &r;                                // bcf_copy(explode_rec, in_rec): keeps its contig (reference quirk)
                    const long p0 = n_in++;
                    if (emit(*tpl, p0, true, gt_row, m)) return true;
                    continue;
                }
                ri++; n_in++;
                if (emit(r, r.pos0, false, gt_row, m)) return true;
            } else {
                if (!(tail_size >= 0 && n_in < tail_size)) { done = true; break; }
                if (!tpl) tpl = &v.recs.back();
                const long p0 = n_in++;
                if (emit(*tpl, p0, true, gt_row, m)) return true;
            }
        }
        return false;
    }
};

// page-locked host array (vgl_host_alloc): grows, never shrinks
template <class T> struct PBuf {
    T* p = nullptr; size_t n = 0;
    static int& device() { static thread_local int d = -1; return d; }   // >= 0: place the memory for DMA from that device (vgl_host_alloc_on)
    void resize(size_t m) {
        if (m <= n) return;
        if (p) vgl_host_free(p);
        p = (T*)(device() >= 0 ? vgl_host_alloc_on(device(), m * sizeof(T)) : vgl_host_alloc(m * sizeof(T)));
        if (!p) die("%s", vgl_last_error());
        n = m;
    }
    T* data() { return p; }
    const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    PBuf() = default; PBuf(const PBuf&) = delete; PBuf& operator=(const PBuf&) = delete;
    ~PBuf() { if (p) vgl_host_free(p); }
};

// ---------------------------------------------------------------------------------------
// gVCF blocks: prepare_gvcf_block(), bcf_utils.cpp:662-942.  Invariant records (one observed
// allele) whose minimum per-sample depth falls in the same --gvcf-dps range are merged into one
// record with END / MIN_DP, per-sample minimum DP and the smallest (REF,ALT),(ALT,ALT) PLs.
struct SiteView {
    const std::string* chrom; long pos0; int n_obs, n_alleles, N, G;
    const int32_t* dp;        // [N]
    const int32_t* pl;        // [G][N] planes of this site
    const float* qs;          // [n_alleles] or null
    std::string alleles;      // "A,<NON_REF>"
};
struct GvcfBlocker {
    enum { NO_WRITE = 0, FLUSH_BLOCK = 1, WRITE_SIMREC = 2 };
    std::vector<int> block_dps;
    int current_dpr = 0;
    std::vector<int32_t> dp, pl;
    std::vector<float> qsum;
    std::string chrom, alleles;
    long start_pos = -1, end_pos = -1;
    int32_t min_dp = 0;

    int prepare(const SiteView* sv) {
        if (!sv) return current_dpr == 0 ? NO_WRITE : FLUSH_BLOCK;
        if (current_dpr == 0) { if (sv->n_obs != 1) return WRITE_SIMREC; }
        else {
            if (sv->n_obs != 1) return FLUSH_BLOCK;                       // broken by a variant site
            if (*sv->chrom != chrom) return FLUSH_BLOCK;                  // other contig
            if (sv->pos0 > end_pos + 1) return FLUSH_BLOCK;               // gap
        }
        int32_t mdp = sv->dp[0];
        for (int s = 1; s < sv->N; ++s) if (mdp > sv->dp[s]) mdp = sv->dp[s];
        int r = 0;
        for (r = 0; r < (int)block_dps.size(); ++r) if (mdp < block_dps[r]) break;
        const int dp_range = r;
        if (!dp_range) return current_dpr == 0 ? WRITE_SIMREC : FLUSH_BLOCK;
        if (current_dpr != 0 && current_dpr != dp_range) return FLUSH_BLOCK;
        if (current_dpr == 0) {                                           // founder of a new block
            dp.assign(sv->dp, sv->dp + sv->N);
            const int nG = sv->n_alleles * (sv->n_alleles + 1) / 2;
            pl.assign((size_t)sv->N * nG, 0);
            pl.assign(sv->pl, sv->pl + (size_t)sv->N * nG);                 // sample-major, like the block's own array
            qsum.clear(); if (sv->qs) qsum.assign(sv->qs, sv->qs + sv->n_alleles);
            chrom = *sv->chrom; start_pos = sv->pos0; alleles = sv->alleles; min_dp = mdp; current_dpr = dp_range;
        } else {
            if (min_dp > mdp) min_dp = mdp;
            for (int s = 0; s < sv->N; ++s) if (dp[s] > sv->dp[s]) dp[s] = sv->dp[s];
            if (sv->n_alleles != 2 || pl.size() != (size_t)sv->N * 3) die("Unexpected number of PL values: %d", sv->N * sv->n_alleles * (sv->n_alleles + 1) / 2);
            for (int s = 0; s < sv->N; ++s) {
                const int32_t p1 = sv->pl[(size_t)3 * s + 1], p2 = sv->pl[(size_t)3 * s + 2];
                if (pl[3 * s + 1] > p1) { pl[3 * s + 1] = p1; pl[3 * s + 2] = p2; }
                else if (pl[3 * s + 1] == p1 && pl[3 * s + 2] > p2) pl[3 * s + 2] = p2;
            }
        }
        end_pos = sv->pos0;
        return NO_WRITE;
    }

    void emit(vsink::Sink& out, int N) {
        const long end1 = end_pos + 1;                                    // 0-based -> 1-based
        std::string line = chrom;
        char hb[64]; snprintf(hb, sizeof hb, "\t%ld\t.\t", start_pos + 1); line += hb;
        const size_t c = alleles.find(',');
        line += alleles.substr(0, c); line += '\t'; line += (c == std::string::npos) ? "." : alleles.substr(c + 1);
        line += "\t.\t.\t";
        if (end1 - start_pos >= 2) { snprintf(hb, sizeof hb, "END=%ld;", end1); line += hb; }
        snprintf(hb, sizeof hb, "MIN_DP=%d", min_dp); line += hb;
        if (!qsum.empty()) { line += ";QS="; for (size_t k = 0; k < qsum.size(); k++) { if (k) line += ','; out.put_float(line, qsum[k]); } }
        line += "\tPL:DP";
        const size_t nG = pl.size() / (size_t)N;
        for (int s = 0; s < N; ++s) {
            line += '\t';
            for (size_t g = 0; g < nG; ++g) { if (g) line += ','; put_int(line, pl[(size_t)s * nG + g]); }
            line += ':'; put_int(line, dp[s]);
        }
        out.write_line(line);
        current_dpr = 0; chrom.clear();
    }
};

// <prefix>.arg: the run log the reference writes beside its outputs (io.cpp:1031,1109; vcfgl.cpp:1657-1871)
struct RunLog {
    FILE* fp = nullptr; std::string prefix; time_t t0 = 0; clock_t c0 = 0;
    void open(const Args& a) {
        prefix = a.out_prefix; t0 = time(NULL); c0 = clock();
        fp = fopen((prefix + ".arg").c_str(), "w");
        if (!fp) die("Could not open file: %s.arg", prefix.c_str());
        char when[64]; struct tm tmv; localtime_r(&t0, &tmv); strftime(when, sizeof when, "%a %b %d %H:%M:%S %Y", &tmv);
        fprintf(fp, "vcfgl_hip (libvcfgl_hip ABI %d, gfx950)\n\n%s\n\n\n[Program start] %s\n", vgl_abi_version(), a.command.c_str(), when);
    }
    void finish(const std::string& summary, const std::vector<std::string>& files) {
        if (!fp) return;
        fputs(summary.c_str(), fp);
        fprintf(fp, "\n\tElapsed time (CPU): %f seconds\n\tElapsed time (Real): %f seconds\n", (double)(clock() - c0) / CLOCKS_PER_SEC, difftime(time(NULL), t0));
        fprintf(fp, "\n-> Log file: %s.arg\n", prefix.c_str());
        for (const std::string& f : files) fprintf(fp, "%s\n", f.c_str());
        fclose(fp); fp = nullptr;
    }
};

// ---------------------------------------------------------------------------------------
int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "--format-floats")) {
        // self-test hook of the VCF float formatter: hex float32 bit patterns in, formatted text out
        for (int i = 2; i < argc; i++) { uint32_t b = (uint32_t)strtoul(argv[i], NULL, 16); float f; memcpy(&f, &b, 4); std::string s; put_float(s, f); printf("%s\n", s.c_str()); }
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "--qs-to-errprob")) {        // test hook: QS_TO_ERRPROB of every argument, 17 digits
        for (int i = 2; i < argc; i++) printf("%.17g\n", host_qs_to_errprob(atoi(argv[i])));
        return 0;
    }
    if (argc >= 5 && !strcmp(argv[1], "--encode-selftest")) {
        // self-test hook of the BCF writer: --encode-selftest <mode> <out path> <int> [<int> ...] writes one record whose
        // FORMAT/X holds the given integers for sample s1 (and their reverse for s2) and INFO/Y the same list
        vsink::Sink out; out.text_float = put_float;
        std::vector<std::string> hdr = {"##fileformat=VCFv4.2", "##contig=<ID=c1,length=10>",
                                        "##INFO=<ID=Y,Number=.,Type=Integer,Description=\"y\">", "##FORMAT=<ID=X,Number=.,Type=Integer,Description=\"x\">"};
        out.open(argv[3], argv[2][0], hdr, {"s1", "s2"});
        std::vector<int32_t> v; for (int i = 4; i < argc; i++) v.push_back(!strcmp(argv[i], ".") ? VGL_INT32_MISSING : (int32_t)strtol(argv[i], NULL, 10));
        const int n = (int)v.size();
        std::vector<int32_t> plane(2 * (size_t)n);
        for (int k = 0; k < n; k++) { plane[(size_t)k * 2] = v[k]; plane[(size_t)k * 2 + 1] = v[n - 1 - k]; }
        std::string sh = "c1\t5\trs1\tA\tC,<*>\t.\tPASS\tY=";
        for (int k = 0; k < n; k++) { if (k) sh += ','; put_int(sh, v[k]); }
        out.write_rec(sh, {{"X", false, n, plane.data(), 1, 2}});
        out.close();
        return 0;
    }
    Args a = parse_args(argc, argv);
    RunLog runlog; runlog.open(a);
    // --verbose 1: wall-clock seconds per stage on stderr at the end
    double t_stage[8] = {0, 0, 0, 0, 0, 0, 0, 0};             // read, sites, context, waiting for the device, encode, write, tile buffers (page-locked), teardown
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    double t_mark = now();
    auto lap = [&](int k) { const double t = now(); t_stage[k] += t - t_mark; t_mark = t; };
    // host threads for parsing and record encoding: --encode-threads, else --threads when given, else up to 8 of the
    // machine's threads -- the bytes written do not depend on it
    int enc_threads = a.enc_threads > 0 ? a.enc_threads : a.threads;
    if (a.enc_threads <= 0 && !a.threads_given) { const unsigned hc = std::thread::hardware_concurrency(); enc_threads = (int)std::max(1u, std::min(8u, hc)); }
    // the HIP runtime initialises (about 0.07 s) while the input is read and parsed
    std::thread hip_warm;
    // (on the first device the run selected: a primary context on GPU 0 would otherwise be created for a run that never uses it;
    //  a failure here is not swallowed for good -- vgl_ctx_create on the same device reports it below)
    const int warm_dev = a.devices.empty() ? a.device : a.devices[0];
    if (!a.depth_inf) hip_warm = std::thread([warm_dev] { vgl_host_free(vgl_host_alloc_on(warm_dev, 4096)); });
    Vcf vcf = read_vcf(a.in_fn, a.print_truth != 0, enc_threads);
    if (hip_warm.joinable()) hip_warm.join();
    lap(0);
    const int N = (int)vcf.samples.size();
    if (N <= 0) die("no samples in %s", a.in_fn.c_str());
    if (!a.depths.empty() && (int)a.depths.size() != N) die("--depths-file must hold one depth per sample (%zu given, %d samples)", a.depths.size(), N);
    const char mode = a.output_mode[0];
    const std::string ext = mode == 'v' ? ".vcf" : mode == 'z' ? ".vcf.gz" : ".bcf";
    // binary output needs every contig / FILTER / INFO key of the records defined in the header (exploded sites carry the
    // contig of an input record)
    auto complete_header = [&](std::vector<std::string>& hdr) {
        if (mode != 'u' && mode != 'b') return;
        std::vector<std::string> contigs, filters, keys, tmp;
        auto add = [](std::vector<std::string>& v, const std::string& x) { if (!x.empty() && std::find(v.begin(), v.end(), x) == v.end()) v.push_back(x); };
        for (const Rec& r : vcf.recs) {
            add(contigs, r.chrom);
            split(r.filt, ';', tmp); for (auto& f : tmp) add(filters, f);
            if (r.info != ".") { split(r.info, ';', tmp); for (auto& kv : tmp) add(keys, kv.substr(0, kv.find('='))); }
        }
        vsink::Sink::define_missing(hdr, contigs, filters, keys);
    };
    SiteStream stream(a, vcf, N);
    vsink::Sink truth_sink; truth_sink.text_float = put_float;
    if (a.print_truth) {                                 // written as the sites go by (vcfgl.cpp:1518,1548,1607)
        std::vector<std::string> hdr = vcf.header;
        hdr.push_back("##source=vcfgl_hip"); hdr.push_back("##source=" + a.command);
        complete_header(hdr);
        truth_sink.open(a.out_prefix + ".truth" + ext, mode, hdr, vcf.samples);
        stream.truth = &truth_sink;
    }
    size_t n_sites_total = 0;

    if (a.depth_inf) {                                   // simulate_record_true_values, vcfgl.cpp:1089-1262
        vsink::Sink out; out.text_float = put_float;
        std::vector<std::string> hdr;
        for (const std::string& h : vcf.header) if (h.find("##FORMAT=<ID=GT,") == std::string::npos) hdr.push_back(h);
        hdr.push_back("##source=vcfgl_hip"); hdr.push_back("##source=" + a.command);
        if (a.add_gl) hdr.push_back("##FORMAT=<ID=GL,Number=G,Type=Float,Description=\"log10 genotype likelihoods, best = 0\">");
        if (a.add_gp) hdr.push_back("##FORMAT=<ID=GP,Number=G,Type=Float,Description=\"Genotype probabilities\">");
        if (a.add_pl) hdr.push_back("##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"Phred-scaled genotype likelihoods\">");
        complete_header(hdr);
        out.open(a.out_prefix + ext, mode, hdr, vcf.samples);
        const bool explode_acgt = a.do_unobserved >= 3;
        const bool add_unobs = (a.do_unobserved == 1 || a.do_unobserved == 2 || a.do_unobserved == 4 || a.do_unobserved == 5);
        const char* nonref = (a.do_unobserved == 1 || a.do_unobserved == 4) ? "<*>" : "<NON_REF>";
        std::string line;
        std::vector<uint8_t> gtrow(N);
        SiteMeta S;
        while (stream.next(gtrow.data(), S)) {
            n_sites_total++;
            int ac[4] = {0, 0, 0, 0};
            for (int s = 0; s < N; s++) {
                const int b0 = gtrow[s] & 0xF, b1 = (gtrow[s] >> 4) & 0xF;
                if (b0 > 3 || b1 > 3) die("--depth inf needs complete A/C/G/T genotypes (position %ld)", S.pos0 + 1);
                ac[b0]++; ac[b1]++;
            }
            int order[4] = {0, 1, 2, 3}, n_obs = 0;
            for (int i = 0; i < 4; ++i) {
                if (ac[i] > 0) n_obs++;
                for (int j = i; j > 0 && ac[order[j]] > ac[order[j - 1]]; j--) std::swap(order[j], order[j - 1]);
            }
            std::vector<std::string> al;
            int idx_of[5] = {-1, -1, -1, -1, -1};
            const int n_acgt = explode_acgt ? 4 : n_obs;
            for (int i = 0; i < n_acgt; i++) { idx_of[order[i]] = (int)al.size(); al.push_back(std::string(1, "ACGT"[order[i]])); }
            if (add_unobs) al.push_back(nonref);
            const int nA = (int)al.size(), nG = nA * (nA + 1) / 2;
            line = S.rec->chrom; char hb[64]; snprintf(hb, sizeof hb, "\t%ld\t", S.pos0 + 1); line += hb;
            line += S.rec->id; line += '\t'; line += al[0]; line += '\t';
            if (nA == 1) line += '.'; else for (int k = 1; k < nA; k++) { if (k > 1) line += ','; line += al[k]; }
            line += '\t'; line += S.rec->qual; line += '\t'; line += S.rec->filt; line += '\t'; line += S.rec->info; line += '\t';
            std::string fmt;
            if (a.add_gl) fmt += "GL"; if (a.add_gp) { if (!fmt.empty()) fmt += ':'; fmt += "GP"; } if (a.add_pl) { if (!fmt.empty()) fmt += ':'; fmt += "PL"; }
            line += fmt.empty() ? "." : fmt;
            for (int s = 0; s < N; s++) {
                const int i0 = idx_of[gtrow[s] & 0xF], i1 = idx_of[(gtrow[s] >> 4) & 0xF];
                const int tg = i0 > i1 ? i0 * (i0 + 1) / 2 + i1 : i1 * (i1 + 1) / 2 + i0;
                line += '\t';
                bool first = true;
                auto sep = [&]() { if (!first) line += ':'; first = false; };
                if (a.add_gl) { sep(); for (int g = 0; g < nG; g++) { if (g) line += ','; line += (g == tg) ? "0" : "-inf"; } }
                if (a.add_gp) { sep(); for (int g = 0; g < nG; g++) { if (g) line += ','; line += (g == tg) ? "1" : "0"; } }
                if (a.add_pl) { sep(); for (int g = 0; g < nG; g++) { if (g) line += ','; line += (g == tg) ? "0" : "255"; } }
                if (first) line += '.';
            }
            out.write_line(line);
        }
        out.close();
        if (a.print_truth) truth_sink.close();
        char sb[512]; snprintf(sb, sizeof sb, "\n\n-> Simulation finished successfully.\n\nSummary:\n\tNumber of samples: %d\n\tTotal number of sites simulated: %zu\n", N, n_sites_total);
        fputs(sb, stderr);
        std::vector<std::string> files = {"-> Simulation output file: " + a.out_prefix + ext};
        if (a.print_truth) files.push_back("-> True genotypes output file: " + a.out_prefix + ".truth" + ext);
        runlog.finish(sb, files);
        return 0;
    }

    vgl_params p; memset(&p, 0, sizeof p);
    p.abi_version = VGL_ABI_VERSION; p.out_layout = VGL_LAYOUT_SAMPLE_MAJOR; p.seed = a.seed; p.n_samples = N; p.rng_mode = a.rng_mode; p.beta_sampler = a.beta_sampler;
    p.depth = a.depth; p.depths = a.depths.empty() ? nullptr : a.depths.data();
    p.error_rate = a.error_rate; p.error_qs = a.error_qs; p.beta_variance = a.beta_variance; p.gl_model = a.gl_model;
    p.gl1_theta = a.gl1_theta; p.precise_gl = a.precise_gl; p.adjust_qs = a.adjust_qs; p.adjust_by = a.adjust_by;
    p.n_qs_bins = (int)a.qs_bins.size() / 3; p.qs_bins = a.qs_bins.empty() ? nullptr : a.qs_bins.data(); p.i16_mapq = a.i16_mapq;
    p.do_unobserved = a.do_unobserved; p.rm_invar_sites = a.rm_invar; p.rm_empty_sites = a.rm_empty; p.do_gvcf = a.do_gvcf;
    p.add_gl = a.add_gl; p.add_gp = a.add_gp; p.add_pl = a.add_pl; p.add_i16 = a.add_i16; p.add_qs = a.add_qs;
    p.add_fmt_dp = a.add_fmt_dp; p.add_info_dp = a.add_info_dp; p.add_fmt_ad = a.add_fmt_ad; p.add_info_ad = a.add_info_ad;
    p.add_fmt_adf = a.add_fmt_adf; p.add_info_adf = a.add_info_adf; p.add_fmt_adr = a.add_fmt_adr; p.add_info_adr = a.add_info_adr;
    int TS = a.tile_sites > 0 ? a.tile_sites : 4096;
    if (a.print_pileup || a.print_qs_err || a.print_gl_err || a.print_qscores)      // per-read dumps: bounded host / device staging
        TS = std::max(1, std::min(TS, (int)((64u << 20) / ((size_t)1024 * (size_t)std::max(N, 1)) + 1)));
    // ---- devices: one context and one host thread per GPU; tiles are dealt to them round robin and come back to the writer
    //      (this thread) in site order.  Every value depends only on the absolute site index (VGL_RNG_TILE), so the file does
    //      not depend on the number of devices.  VGL_RNG_SERIAL consumes its streams in call order: one device.
    std::vector<int> devices = a.devices.empty() ? std::vector<int>{a.device} : a.devices;
    if (devices.size() > 1 && a.rng_mode == VGL_RNG_SERIAL) die("--devices: --rng-mode 1 (the reference's serial draw order) does not shard; use one device");
    const int D = (int)devices.size();
    std::vector<vgl_ctx*> ctxs(D, nullptr);
    t_mark = now();
    for (int d = 0; d < D; d++) if (vgl_ctx_create(&p, devices[d], TS, &ctxs[d]) != VGL_OK) die("%s", vgl_last_error());
    lap(2);
    const int A = vgl_max_alleles(&p), G = vgl_max_genotypes(&p);

    // ---- output header (set_hdr, bcf_utils.cpp:511-615): input header minus FORMAT/GT, plus our tags
    vsink::Sink out; out.text_float = put_float;
    {
        std::vector<std::string> hdr;
        char hb[128];
        for (const std::string& h : vcf.header) if (h.find("##FORMAT=<ID=GT,") == std::string::npos) hdr.push_back(h);
        snprintf(hb, sizeof hb, "##source=vcfgl_hip (libvcfgl_hip ABI %d, gfx950)", vgl_abi_version()); hdr.push_back(hb);
        hdr.push_back("##source=" + a.command);
        if (a.do_unobserved == 1 || a.do_unobserved == 4) hdr.push_back("##ALT=<ID=*,Description=\"Any other alternative allele (unobserved)\">");
        if (a.do_unobserved == 2 || a.do_unobserved == 5) hdr.push_back("##ALT=<ID=NON_REF,Description=\"Any other alternative allele (unobserved)\">");
        if (a.do_gvcf) { hdr.push_back("##INFO=<ID=END,Number=1,Type=Integer,Description=\"Last position of the non-variant block\">");
                         hdr.push_back("##INFO=<ID=MIN_DP,Number=1,Type=Integer,Description=\"Smallest per-sample depth within the block\">"); }
        if (a.add_fmt_dp) hdr.push_back("##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Simulated read depth of the sample\">");
        if (a.add_info_dp) hdr.push_back("##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Read depth summed over samples\">");
        if (a.add_gl) hdr.push_back("##FORMAT=<ID=GL,Number=G,Type=Float,Description=\"log10 genotype likelihoods, best = 0\">");
        if (a.add_pl) hdr.push_back("##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"Phred-scaled genotype likelihoods\">");
        if (a.add_gp) hdr.push_back("##FORMAT=<ID=GP,Number=G,Type=Float,Description=\"Genotype probabilities\">");
        if (a.add_qs) hdr.push_back("##INFO=<ID=QS,Number=R,Type=Float,Description=\"Normalised per-allele base quality sum\">");
        if (a.add_i16) hdr.push_back("##INFO=<ID=I16,Number=16,Type=Float,Description=\"bcftools call auxiliary tag\">");
        if (a.add_fmt_ad) hdr.push_back("##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"Allelic depths\">");
        if (a.add_fmt_adf) hdr.push_back("##FORMAT=<ID=ADF,Number=R,Type=Integer,Description=\"Allelic depths, forward strand\">");
        if (a.add_fmt_adr) hdr.push_back("##FORMAT=<ID=ADR,Number=R,Type=Integer,Description=\"Allelic depths, reverse strand\">");
        if (a.add_info_ad) hdr.push_back("##INFO=<ID=AD,Number=R,Type=Integer,Description=\"Total allelic depths\">");
        if (a.add_info_adf) hdr.push_back("##INFO=<ID=ADF,Number=R,Type=Integer,Description=\"Total allelic depths, forward strand\">");
        if (a.add_info_adr) hdr.push_back("##INFO=<ID=ADR,Number=R,Type=Integer,Description=\"Total allelic depths, reverse strand\">");
        complete_header(hdr);
        // BGZF compression threads: --threads as in the reference; when it is not given, up to 8 (same bytes either way)
        out.open(a.out_prefix + ext, mode, hdr, vcf.samples, a.threads_given ? a.threads : (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
    }
    FILE* pile_fp = nullptr; vsink::Bgzf pile;            // the reference writes the pileup through htslib's BGZF (vcfgl.cpp:1776-1783)
    if (a.print_pileup) {
        pile_fp = fopen((a.out_prefix + ".pileup.gz").c_str(), "wb"); if (!pile_fp) die("Could not open pileup output");
        pile.open(pile_fp, 1);
    }
    // ---- TSV lines on stdout (vcfgl.cpp:430-435, 533-554, 1745-1755)
    int pre_q = -1, pre_adjq = -1;                                              // preCalc->qScore / adj_qScore
    if (a.print_bpe && a.error_qs != 1) printf("base_pick_error_prob\tNA\tNA\tNA\tNA\t%f\n", a.error_rate);   // io.cpp:1089-1100
    if (a.error_qs != 2) {
        host_errprob_to_qs(a, a.error_rate, pre_q, pre_adjq);
        if (a.print_gl_err) printf("gl_error_prob\tNA\tNA\tNA\tNA\t%f\n", a.precise_gl ? a.error_rate : host_qs_to_errprob((a.adjust_qs & 1) ? pre_adjq : pre_q));
        if (a.print_qs_err) printf("qs_error_prob\tNA\tNA\tNA\tNA\t%f\n", a.error_rate);
        if (a.print_qscores) printf("qs\tNA\tNA\tNA\tNA\t%d\n", a.adjust_qs ? pre_adjq : pre_q);
    }
    const bool dump_reads = a.error_qs == 2 && (a.print_qs_err || a.print_gl_err || a.print_qscores);
    const bool want_errp = dump_reads || (a.error_qs == 2 && pile_fp && (a.adjust_qs & 4));
    const bool dump_pick = a.error_qs == 1 && a.print_bpe;
    // per-read dump rows: the library's own staging capacity (vgl_host.cpp: depth + 8 sqrt(depth) + 16)
    double dmax = a.depth; for (double d : a.depths) dmax = std::max(dmax, d); if (!(dmax >= 0)) dmax = 0;
    const int pile_cap = (((int)ceil(dmax + 8.0 * sqrt(dmax) + 16.0)) + 3) & ~3;
    const char* nonref = (a.do_unobserved == 1 || a.do_unobserved == 4) ? "<*>" : "<NON_REF>";

    // ---- tile buffers (host side of vgl_tile_out): only what this run prints is requested from the device
    const bool want_dp = a.add_fmt_dp || a.do_gvcf || pile_fp || dump_reads;
    struct TileBufs {
        int ns = 0; int64_t t0 = 0; int dev = 0;
        std::vector<SiteMeta> meta; std::vector<uint8_t> gt;
        // outputs live in page-locked memory (vgl_host_alloc): the device writes them by DMA while the next tile is computed
        PBuf<uint8_t> reads;
        PBuf<int32_t> st, na, nobs, idp, iad, iadf, iadr, dp, pl, ad, adf, adr;
        PBuf<int8_t> a2b; PBuf<float> qs, i16, gl, gp; PBuf<double> errp, pick;
        vgl_tile_out o;
        std::mutex m; std::condition_variable cv; bool done = false;
    };
    const size_t E = (size_t)TS * N;
    const int R = 2 * D;                                        // tiles in flight: two per device
    std::vector<std::unique_ptr<TileBufs>> ring(R);
    // An entry's buffers are page-locked when the entry is first used (about 0.04 s per 250 MB): the second entry of a device is
    // prepared while the device already works on the first tile
    auto alloc_entry = [&](size_t ri) {
        auto& up = ring[ri];
        if (up) return;
        up.reset(new TileBufs());
        TileBufs& B = *up;
        // ring entry ri only ever serves device ri % D (tiles are dealt round robin, two entries per device): its page-locked
        // buffers are placed next to that device
        const int dev_of_entry = devices[ri % (size_t)D];
        PBuf<uint8_t>::device() = dev_of_entry; PBuf<int32_t>::device() = dev_of_entry; PBuf<int8_t>::device() = dev_of_entry;
        PBuf<float>::device() = dev_of_entry; PBuf<double>::device() = dev_of_entry;
        B.meta.resize(TS); B.gt.resize(E);
        B.st.resize(TS); B.na.resize(TS); B.nobs.resize(TS); B.a2b.resize((size_t)TS * 5);
        memset(&B.o, 0, sizeof B.o);
        B.o.site_status = B.st.data(); B.o.n_alleles = B.na.data(); B.o.n_alleles_obs = B.nobs.data(); B.o.alleles2acgt = B.a2b.data();
        B.idp.resize(TS); B.o.info_dp = B.idp.data();           // also tells which sites reach the read loop (TSV dumps)
        if (a.add_info_ad) { B.iad.resize((size_t)TS * A); B.o.info_ad = B.iad.data(); }
        if (a.add_info_adf) { B.iadf.resize((size_t)TS * A); B.o.info_adf = B.iadf.data(); }
        if (a.add_info_adr) { B.iadr.resize((size_t)TS * A); B.o.info_adr = B.iadr.data(); }
        if (a.add_qs) { B.qs.resize((size_t)TS * A); B.o.qs = B.qs.data(); }
        if (a.add_i16) { B.i16.resize((size_t)TS * 16); B.o.i16 = B.i16.data(); }
        if (want_dp) { B.dp.resize(E); B.o.fmt_dp = B.dp.data(); }
        if (a.add_gl) { B.gl.resize(E * G); B.o.gl = B.gl.data(); }
        if (a.add_pl) { B.pl.resize(E * G); B.o.pl = B.pl.data(); }
        if (a.add_gp) { B.gp.resize(E * G); B.o.gp = B.gp.data(); }
        if (a.add_fmt_ad) { B.ad.resize(E * A); B.o.fmt_ad = B.ad.data(); }
        if (a.add_fmt_adf) { B.adf.resize(E * A); B.o.fmt_adf = B.adf.data(); }
        if (a.add_fmt_adr) { B.adr.resize(E * A); B.o.fmt_adr = B.adr.data(); }
    };
    alloc_entry(0);
    // one worker per device: simulates the tiles handed to it, in order
    struct Worker { std::thread th; std::mutex m; std::condition_variable cv; std::vector<TileBufs*> q; size_t head = 0; bool stop = false;
                    long tiles = 0, sites = 0; double t_first = -1.0, t_last = 0.0; };          // --verbose 1: what this device did (written by its own thread, read after the join)
    // bytes a finished tile brings back over the link, per site (the FORMAT arrays dominate: sample-major slabs, copied whole)
    const double bytes_per_site = (double)N * ((want_dp ? 4.0 : 0.0) + 4.0 * G * ((a.add_gl ? 1 : 0) + (a.add_pl ? 1 : 0) + (a.add_gp ? 1 : 0)) +
                                               4.0 * A * ((a.add_fmt_ad ? 1 : 0) + (a.add_fmt_adf ? 1 : 0) + (a.add_fmt_adr ? 1 : 0))) + 64.0;
    std::vector<std::unique_ptr<Worker>> workers(D);
    for (int d = 0; d < D; d++) {
        workers[d].reset(new Worker());
        Worker* W = workers[d].get();
        vgl_ctx* ctx = ctxs[d];
        W->th = std::thread([W, ctx, &now]() {
            // a tile is submitted (vgl_simulate_tile_async) before the previous one is waited for: its kernels run while the
            // previous tile's tags are still on their way to the host
            TileBufs* prev = nullptr; int32_t prev_ticket = 0;
            for (;;) {
                TileBufs* B = nullptr;
                {
                    std::unique_lock<std::mutex> lk(W->m);
                    if (!prev) W->cv.wait(lk, [&] { return W->stop || W->head < W->q.size(); });
                    if (W->head < W->q.size()) B = W->q[W->head++];
                    else if (!prev) return;
                }
                int32_t ticket = 0;
                if (B && W->t_first < 0.0) W->t_first = now();
                if (B && vgl_simulate_tile_async(ctx, B->t0, B->ns, B->gt.data(), &B->o, &ticket) != VGL_OK) die("%s", vgl_last_error());
                if (prev) {
                    if (vgl_tile_wait(ctx, prev_ticket) != VGL_OK) die("%s", vgl_last_error());
                    W->tiles += 1; W->sites += prev->ns; W->t_last = now();
                    { std::lock_guard<std::mutex> lk(prev->m); prev->done = true; }
                    prev->cv.notify_all();
                }
                prev = B; prev_ticket = ticket;
            }
        });
    }

    long n_out = 0, n_skipped = 0;
    std::string line, tsv;
    GvcfBlocker gv;
    std::vector<std::string> enc;
    gv.block_dps = a.gvcf_dps;
    // one simulated record of a tile: the eight fixed columns as text, the allele strings and the
    // typed FORMAT arrays (reads the tile buffers only: records of a tile are built on several threads)
    auto build_record = [&](const TileBufs& B, int i, std::string& line, std::vector<std::string>& al, std::vector<vsink::FmtDesc>& fmt) {
        const SiteMeta& S = B.meta[i];
        const int nA = B.na[i], nG = nA * (nA + 1) / 2;
        char hb[64];
        line += S.rec->chrom; snprintf(hb, sizeof hb, "\t%ld\t", S.pos0 + 1); line += hb;
        line += S.rec->id; line += '\t';
        // alleles (vcfgl.cpp:739-762; no-reads site :250-280)
        al.clear();
        for (int k = 0; k < nA; k++) { const int b = B.a2b[(size_t)i * 5 + k]; al.push_back(b == 4 ? nonref : (b >= 0 ? std::string(1, "ACGT"[b]) : ".")); }
        if (al.empty()) al.push_back(".");
        line += al[0]; line += '\t';
        if (al.size() == 1) line += '.';
        else for (size_t k = 1; k < al.size(); k++) { if (k > 1) line += ','; line += al[k]; }
        line += '\t'; line += S.rec->qual; line += '\t'; line += S.rec->filt; line += '\t';
        // INFO in add_tags() order: DP, QS, I16, AD, ADF, ADR (after the input record's own INFO)
        std::string info = (S.rec->info == ".") ? "" : S.rec->info;
        auto add_key = [&](const char* k) { if (!info.empty()) info += ';'; info += k; info += '='; };
        if (a.add_info_dp) { add_key("DP"); put_int(info, B.idp[i]); }
        if (a.add_qs) { add_key("QS"); for (int k = 0; k < nA; k++) { if (k) info += ','; out.put_float(info, B.qs[(size_t)i * A + k]); } }
        if (a.add_i16) { add_key("I16"); for (int k = 0; k < 16; k++) { if (k) info += ','; out.put_float(info, B.i16[(size_t)i * 16 + k]); } }
        if (a.add_info_ad) { add_key("AD"); for (int k = 0; k < nA; k++) { if (k) info += ','; put_int(info, B.iad[(size_t)i * A + k]); } }
        if (a.add_info_adf) { add_key("ADF"); for (int k = 0; k < nA; k++) { if (k) info += ','; put_int(info, B.iadf[(size_t)i * A + k]); } }
        if (a.add_info_adr) { add_key("ADR"); for (int k = 0; k < nA; k++) { if (k) info += ','; put_int(info, B.iadr[(size_t)i * A + k]); } }
        line += info.empty() ? "." : info;
        // FORMAT keys: DP, GL, PL, GP, AD, ADF, ADR.  The library writes the multi-valued tags sample-major (VGL_LAYOUT_SAMPLE_MAJOR):
        // the slab of site i holds the record's array as the reference keeps it for bcf_update_format_*(), element k of sample s
        // at slab[s * n + k] with the site's own n -- the encoders below read (and for BCF copy) it front to back
        fmt.clear();
        const size_t sN = (size_t)N, sG = (size_t)nG, sA = (size_t)nA;
        if (a.add_fmt_dp) fmt.push_back({"DP", false, 1, &B.dp[(size_t)i * N], 1, sN});
        if (a.add_gl) fmt.push_back({"GL", true, nG, &B.gl[(size_t)i * G * N], sG, 1});
        if (a.add_pl) fmt.push_back({"PL", false, nG, &B.pl[(size_t)i * G * N], sG, 1});
        if (a.add_gp) fmt.push_back({"GP", true, nG, &B.gp[(size_t)i * G * N], sG, 1});
        if (a.add_fmt_ad) fmt.push_back({"AD", false, nA, &B.ad[(size_t)i * A * N], sA, 1});
        if (a.add_fmt_adf) fmt.push_back({"ADF", false, nA, &B.adf[(size_t)i * A * N], sA, 1});
        if (a.add_fmt_adr) fmt.push_back({"ADR", false, nA, &B.adr[(size_t)i * A * N], sA, 1});
    };
    // everything the writer does with one finished tile, in site order (TSV lines, pileup, gVCF blocks, records)
    auto write_tile = [&](TileBufs& B) {
        const int ns = B.ns;
        for (int i = 0; i < ns; i++) {
            const SiteMeta& S = B.meta[i];
            if ((dump_pick || dump_reads) && B.st[i] != VGL_SITE_SKIP_EMPTY && B.idp[i] > 0) {      // sites that reach the read loop (vcfgl.cpp:396-404)
                tsv.clear();
                char hb[96];
                if (dump_pick) for (int s = 0; s < N; s++) {                                   // vcfgl.cpp:430-435
                    tsv += "base_pick_error_prob\t"; tsv += vcf.samples[s]; tsv += '\t'; tsv += S.rec->chrom;
                    snprintf(hb, sizeof hb, "\t%ld\tNA\t%f\n", S.pos0 + 1, B.pick[i]); tsv += hb;
                }
                if (dump_reads) for (int s = 0; s < N; s++) {                                  // vcfgl.cpp:533-554
                    const int n = B.dp[(size_t)i * N + s];
                    for (int r = 0; r < n; r++) {
                        const double ep = B.errp[((size_t)r * ns + i) * N + s];
                        int q, aq; host_errprob_to_qs(a, ep, q, aq);
                        auto head = [&](const char* type) { tsv += type; tsv += '\t'; tsv += vcf.samples[s]; tsv += '\t'; tsv += S.rec->chrom; snprintf(hb, sizeof hb, "\t%ld\t%d\t", S.pos0 + 1, r); tsv += hb; };
                        if (a.print_qs_err) { head("qs_error_prob"); snprintf(hb, sizeof hb, "%f\n", ep); tsv += hb; }
                        if (a.print_qscores) { head("qs"); snprintf(hb, sizeof hb, "%d\n", (a.adjust_qs & 8) ? aq : q); tsv += hb; }
                        if (a.print_gl_err) { head("gl_error_prob"); snprintf(hb, sizeof hb, "%f\n", a.precise_gl ? ep : host_qs_to_errprob((a.adjust_qs & 16) ? aq : q)); tsv += hb; }
                    }
                }
                fwrite(tsv.data(), 1, tsv.size(), stdout);
            }
            if (pile_fp && B.st[i] != VGL_SITE_SKIP_EMPTY) {              // vcfgl.cpp:414-416, 616-634 (printed before skip decisions)
                line.clear();
                char hb[64]; snprintf(hb, sizeof hb, "\t%ld\t%c", S.pos0 + 1, S.ref_char);
                line += S.rec->chrom; line += hb;
                for (int s = 0; s < N; s++) {
                    const int n = B.dp[(size_t)i * N + s];
                    if (n == 0) { line += "\t0\t*\t*"; continue; }
                    snprintf(hb, sizeof hb, "\t%d\t", n); line += hb;
                    for (int r = 0; r < n; r++) line += "ACGT"[B.reads[((size_t)r * ns + i) * N + s] & 3];
                    line += '\t';
                    if (!(a.adjust_qs & 4)) for (int r = 0; r < n; r++) line += (char)((B.reads[((size_t)r * ns + i) * N + s] >> 2) + 33);
                    else if (a.error_qs != 2) line.append((size_t)n, (char)(pre_adjq + 33));                  // PROGRAM_WILL_ADJUST_QS_FOR_PILEUP
                    else for (int r = 0; r < n; r++) { int q, aq; host_errprob_to_qs(a, B.errp[((size_t)r * ns + i) * N + s], q, aq); line += (char)(aq + 33); }
                }
                line += '\n';
                pile.write(line.data(), line.size());
            }
            if (B.st[i] < 0) { n_skipped++; continue; }
            if (!a.do_gvcf) continue;                                    // plain records: encoded in parallel below
            // gVCF: write_record_values (vcfgl.cpp:167-206) carries the open block from record to record (and from tile to
            // tile, whichever device simulated it)
            const int nA = B.na[i];
            std::vector<std::string> al; std::vector<vsink::FmtDesc> fmt;
            line.clear();
            build_record(B, i, line, al, fmt);
            SiteView sv;
            sv.chrom = &S.rec->chrom; sv.pos0 = S.pos0; sv.n_obs = B.nobs[i]; sv.n_alleles = nA; sv.N = N; sv.G = G;
            sv.dp = &B.dp[(size_t)i * N]; sv.pl = &B.pl[(size_t)i * G * N]; sv.qs = a.add_qs ? &B.qs[(size_t)i * A] : nullptr;
            sv.alleles = al[0]; for (size_t k = 1; k < al.size(); k++) { sv.alleles += ','; sv.alleles += al[k]; }
            int ret = gv.prepare(&sv);
            if (ret == GvcfBlocker::FLUSH_BLOCK) { gv.emit(out, N); n_out++; ret = gv.prepare(&sv); }
            if (ret == GvcfBlocker::WRITE_SIMREC) { out.write_rec(line, fmt); n_out++; }
        }
        if (!a.do_gvcf) {
            enc.resize(ns);
            vsink::parallel_for(ns, enc_threads, [&](int i) {
                enc[i].clear();
                if (B.st[i] < 0) return;
                std::string sh; std::vector<std::string> al; std::vector<vsink::FmtDesc> fmt;
                build_record(B, i, sh, al, fmt);
                out.encode_rec(sh, fmt, enc[i]);
            });
            lap(4);
            for (int i = 0; i < ns; i++) if (B.st[i] >= 0) { out.put(enc[i]); n_out++; }
            lap(5);
        }
    };

    // ---- the tile ring: produce (decode sites, hand the tile to its device) up to R tiles ahead, write in order
    size_t produced = 0, consumed = 0;
    bool eof = false;
    lap(6);
    for (;;) {
        while (!eof && produced - consumed < (size_t)R) {
            alloc_entry(produced % R);
            TileBufs& B = *ring[produced % R];
            B.ns = 0; B.t0 = (int64_t)n_sites_total; B.done = false; B.dev = (int)(produced % D);
            while (B.ns < TS && stream.next(&B.gt[(size_t)B.ns * N], B.meta[B.ns])) B.ns++;
            if (B.ns < TS) eof = true;
            if (B.ns == 0) break;
            n_sites_total += (size_t)B.ns;
            PBuf<uint8_t>::device() = devices[B.dev]; PBuf<double>::device() = devices[B.dev];      // dump buffers of this entry: next to its device, like the rest
            if (pile_fp) { B.reads.resize((size_t)pile_cap * TS * N); memset(B.reads.data(), 0xFF, (size_t)pile_cap * B.ns * N); B.o.reads = B.reads.data(); B.o.read_capacity = pile_cap; }   // capacity of the per-read dump: the library stages at most read_cap reads; ask generously
            if (want_errp) { B.errp.resize((size_t)pile_cap * TS * N); B.o.read_errp = B.errp.data(); B.o.read_capacity = pile_cap; }
            if (dump_pick) { B.pick.resize(TS); B.o.site_pick_err = B.pick.data(); }
            Worker* W = workers[B.dev].get();
            { std::lock_guard<std::mutex> lk(W->m); W->q.push_back(&B); }
            W->cv.notify_one();
            produced++;
        }
        lap(1);
        if (consumed == produced) break;
        TileBufs& B = *ring[consumed % R];
        { std::unique_lock<std::mutex> lk(B.m); B.cv.wait(lk, [&] { return B.done; }); }
        lap(3);
        write_tile(B);
        lap(5);
        consumed++;
    }
    for (auto& W : workers) { { std::lock_guard<std::mutex> lk(W->m); W->stop = true; } W->cv.notify_all(); W->th.join(); }
    if (a.do_gvcf && gv.prepare(nullptr) == GvcfBlocker::FLUSH_BLOCK) { gv.emit(out, N); n_out++; }
    t_mark = now();
    out.close();
    if (a.print_truth) truth_sink.close();
    lap(5);
    if (pile_fp) { pile.close(); fclose(pile_fp); }
    // contexts and page-locked buffers are not torn down one by one (0.06 s): the process ends below with _exit(), after the run
    // log, and the driver releases everything at once
    lap(7);
    if (a.verbose) {
        // per device: tiles, sites, bytes of tags copied back and the rate over the device's own busy interval (first submit to last
        // completed wait) -- a multi-GPU run shows an idle or slow device (or link) here at once
        for (int d = 0; d < D; d++) {
            const Worker& W = *workers[d];
            vgl_ctx_info_t ci; memset(&ci, 0, sizeof ci); ci.size = (int32_t)sizeof ci;
            (void)vgl_ctx_info(ctxs[d], &ci);
            const double dt = W.t_last - W.t_first, gb = bytes_per_site * (double)W.sites / 1e9;
            fprintf(stderr, "[device %d] %ld tiles, %ld sites, %.3f GB of tags copied back in %.3f s = %.1f GB/s, %.3g evaluations/s; context: %.2f GB workspace, k_sample build %d, fused %d (split %d)\n",
                    devices[d], W.tiles, W.sites, gb, dt > 0 ? dt : 0.0, dt > 0 ? gb / dt : 0.0, dt > 0 ? (double)W.sites * N / dt : 0.0,
                    (double)ci.workspace_bytes / 1e9, ci.sample_lean, ci.fused, ci.fused_split);
        }
    }
    if (a.verbose) fprintf(stderr, "\n[timing] read input %.3f s, decode sites %.3f s, device context(s) %.3f s, waiting for the device(s) (simulation incl. PCIe, overlapped with the writer) %.3f s, encode %.3f s, write/compress %.3f s, tile buffers %.3f s, teardown %.3f s\n",
                           t_stage[0], t_stage[1], t_stage[2], t_stage[3], t_stage[4], t_stage[5], t_stage[6], t_stage[7]);
    char sb[512];
    snprintf(sb, sizeof sb, "\n\n-> Simulation finished successfully.\n\nSummary:\n\tNumber of samples: %d\n\tTotal number of sites simulated: %zu\n"
                            "\tNumber of sites included in simulation output file: %ld\n\tNumber of sites skipped: %ld\n", N, n_sites_total, n_out, n_skipped);
    fputs(sb, stderr);
    std::vector<std::string> files = {"-> Simulation output file: " + a.out_prefix + ext};
    if (a.print_pileup) files.push_back("-> Pileup output file: " + a.out_prefix + ".pileup.gz");
    if (a.print_truth) files.push_back("-> True genotypes output file: " + a.out_prefix + ".truth" + ext);
    if (a.print_bpe) files.push_back("-> Base pick error output: stdout");
    if (a.print_qs_err) files.push_back("-> QS error output: stdout");
    if (a.print_gl_err) files.push_back("-> GL error output: stdout");
    if (a.print_qscores) files.push_back("-> Qscores output: stdout");
    fflush(stdout);
    runlog.finish(sb, files);
    // a flush that fails here (ENOSPC, EPIPE on stdout's TSV listings) is a failed run; VCFGL_HIP_NORMAL_EXIT=1 leaves through
    // exit() instead of _exit(), so that atexit handlers (profilers, sanitizers) run -- at the price of the piecewise teardown
    const int flush_rc = fflush(NULL);
    if (flush_rc != 0) { fprintf(stderr, "\n[ERROR] could not flush the output streams: %s\n", strerror(errno)); _exit(1); }
    if (getenv("VCFGL_HIP_NORMAL_EXIT")) exit(0);
    _exit(0);
}
