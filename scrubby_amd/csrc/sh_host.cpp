// sh_host.cpp — host side of the replaced path, in C++ (the reference's host language, Rust, is not on this
// image).  Mirrors, function by function, what surrounds the kernel path in the reference:
//
//   get_id                         /root/reference/src/utils.rs:91-103     first whitespace token of the header
//   parse_fastx_file_with_check    /root/reference/src/utils.rs:359-383    None for an empty file; gzip by magic bytes
//   Cleaner::run_minimap2_rs       /root/reference/src/cleaner.rs:443-575  index, ingest R1(/R2), classify, id set
//   FastqCleaner::clean_reads      /root/reference/src/cleaner.rs:731-760  keep/drop by id, output compression by extension
//   ReadDifference::get_difference /root/reference/src/utils.rs:250-285    reads_in / reads_out / difference over both files
//   ScrubbyReport                  /root/reference/src/report.rs:10-88     JSON schema (key order = struct order), TSV of ids
//
// Containers: plain, gzip, bzip2 and xz in and out (sh_codec.h: niffler's set); legacy .lzma streams are refused by name;
// a corrupt FASTQ is an error instead of a logged partial id set.
#include "sh_common.h"
#include "sh_host.h"
#include <zlib.h>
#include <sys/stat.h>
#include <chrono>
#include <thread>
#include <ctime>
#include <unordered_set>
#include <algorithm>
#include "sh_codec.h"

namespace {

struct FastxRecord { std::string header, seq, qual; bool fastq = false; };

// FASTA / FASTQ in any of the containers sh_codec.h reads (plain, gzip, bzip2, xz: sniffed by magic bytes like needletail does); multi-line
// FASTA, 4-line FASTQ
class FastxReader {
    shc::In f_;
    std::string line_, pending_;
    bool have_pending_ = false, eof_ = false;
    std::vector<char> buf_;
    size_t bpos_ = 0, blen_ = 0;
    bool getline(std::string &out)
    {
        out.clear();
        for (;;) {
            if (bpos_ == blen_) {
                if (eof_) return !out.empty();
                const long got = f_.read(buf_.data(), buf_.size());
                if (got < 0) { error = f_.error; eof_ = true; return !out.empty(); }      // a truncated stream is an error, not a short input
                if (got == 0) { eof_ = true; return !out.empty(); }
                bpos_ = 0; blen_ = (size_t)got;
            }
            const char *b = buf_.data() + bpos_;
            const char *nl = (const char *)memchr(b, '\n', blen_ - bpos_);
            if (nl) {
                out.append(b, (size_t)(nl - b));
                bpos_ += (size_t)(nl - b) + 1;
                if (!out.empty() && out.back() == '\r') out.pop_back();
                return true;
            }
            out.append(b, blen_ - bpos_);
            bpos_ = blen_;
        }
    }
public:
    std::string error;
    explicit FastxReader(const char *path) : buf_(1 << 18) { if (!f_.open(path)) error = f_.error; }
    bool ok() const { return f_.is_open(); }
    // 1 = record, 0 = end, -1 = malformed
    int next(FastxRecord &r)
    {
        std::string l;
        if (have_pending_) { l = pending_; have_pending_ = false; }
        else { do { if (!getline(l)) return error.empty() ? 0 : -1; } while (l.empty()); }
        if (!error.empty()) return -1;
        if (l[0] == '@') {
            r.fastq = true; r.header = l.substr(1);
            std::string plus;
            const bool got_all = getline(r.seq) && getline(plus) && !plus.empty() && plus[0] == '+' && getline(r.qual);
            if (!error.empty()) return -1;      // the stream itself failed (truncated / corrupt container): that, not what the partial record looks like
            if (!got_all) { error = "truncated FASTQ record: " + r.header; return -1; }
            if (r.qual.size() != r.seq.size()) { error = "sequence/quality length mismatch: " + r.header; return -1; }
            return 1;
        }
        if (l[0] == '>') {
            r.fastq = false; r.header = l.substr(1); r.seq.clear(); r.qual.clear();
            while (getline(l)) {
                if (!l.empty() && l[0] == '>') { pending_ = l; have_pending_ = true; break; }
                r.seq += l;
            }
            return error.empty() ? 1 : -1;
        }
        error = "not a FASTA/FASTQ record: " + l.substr(0, 40);
        return -1;
    }
};

bool file_is_empty(const char *path, bool &exists)
{   // is_file_empty + compression sniffing: an empty compressed stream counts as empty
    shc::In f;
    exists = f.open(path);
    if (!exists) return true;
    char c;
    return f.read(&c, 1) == 0;      // (a stream that errors before its first byte is not empty: the reader reports it)
}

// get_id: first whitespace-delimited token of the header; a header without one is an error
bool get_id(const std::string &header, std::string &id)
{
    size_t b = 0, n = header.size();
    while (b < n && isspace((unsigned char)header[b])) ++b;
    size_t e = b;
    while (e < n && !isspace((unsigned char)header[e])) ++e;
    if (e == b) return false;
    id.assign(header, b, e - b);
    return true;
}

// output compression chosen by extension (CompressionExt::from_path, utils.rs:28-36), level 6
class FastxWriter {
    shc::Out o_;
public:
    std::string error;
    explicit FastxWriter(const std::string &path, int level = 6) { if (!o_.open(path, level)) error = o_.error; }
    ~FastxWriter() { o_.close(); }
    bool ok() const { return error.empty(); }
    void put(const std::string &s) { if (!o_.write(s.data(), s.size()) && error.empty()) error = o_.error; }
    bool finish() { if (!o_.close() && error.empty()) error = o_.error; return error.empty(); }
    void write(const FastxRecord &r)
    {   // needletail record.write(writer, None): '\n' line endings, bare '+', full original header
        std::string o;
        o.reserve(r.header.size() + r.seq.size() * 2 + 8);
        o += r.fastq ? '@' : '>'; o += r.header; o += '\n'; o += r.seq; o += '\n';
        if (r.fastq) { o += "+\n"; o += r.qual; o += '\n'; }
        put(o);
    }
};

std::string json_escape(const std::string &s)
{
    std::string o;
    for (unsigned char c : s) {
        if (c == '"') o += "\\\""; else if (c == '\\') o += "\\\\"; else if (c == '\n') o += "\\n"; else if (c == '\t') o += "\\t";
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; } else o += (char)c;
    }
    return o;
}

const char *preset_variant(const std::string &display)
{   // Preset serialises as the Rust variant name
    static const char *tab[][2] = {{"sr", "Sr"}, {"map-ont", "MapOnt"}, {"lr:hq", "LrHq"}, {"lr", "Lr"}, {"asm", "Asm"}, {"asm5", "Asm5"},
                                   {"asm10", "Asm10"}, {"asm20", "Asm20"}, {"ava-ont", "AvaOnt"}, {"ava-pb", "AvaPb"}, {"map-hifi", "MapHifi"},
                                   {"map-pb", "MapPb"}, {"splice", "Splice"}, {"splice:hq", "SpliceHq"}};
    for (auto &t : tab) if (display == t[0]) return t[1];
    return nullptr;
}

sh_status filter_fastx(const char *in, const char *out, const std::unordered_set<std::string> &ids, bool extract, uint64_t *n_in, uint64_t *n_out)
{
    bool exists;
    if (file_is_empty(in, exists)) {
        SH_CHECK(exists, SH_ERR_IO, "cannot open %s", in);
        fprintf(stderr, "[scrubby-hip] warning: Input file is empty: %s\n", in);     // output file is NOT created (App. C Q6)
        return SH_OK;
    }
    FastxReader rd(in);
    SH_CHECK(rd.ok(), SH_ERR_IO, "cannot open %s", in);
    FastxWriter wr(out);
    SH_CHECK(wr.ok(), SH_ERR_IO, "%s", wr.error.c_str());
    FastxRecord r; std::string id; int st;
    while ((st = rd.next(r)) == 1) {
        SH_CHECK(get_id(r.header, id), SH_ERR_IO, "record without an id in %s", in);
        if (n_in) ++*n_in;
        const bool hit = ids.count(id) != 0;
        if (hit == extract) { wr.write(r); if (n_out) ++*n_out; }        // deplete: keep misses; extract: keep hits
    }
    SH_CHECK(st == 0, SH_ERR_IO, "%s: %s", in, rd.error.c_str());
    SH_CHECK(wr.finish(), SH_ERR_IO, "%s: %s", out, wr.error.c_str());
    return SH_OK;
}

}  // namespace

bool shi_unsupported_compression(const char *path)
{   // bzip2 and xz are read (sh_codec.h); what is left to refuse by name is the legacy LZMA_alone stream, which niffler does not sniff either
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    unsigned char m[6] = {0, 0, 0, 0, 0, 0};
    const size_t n = fread(m, 1, 6, f);
    fclose(f);
    if (!(n >= 3 && m[0] == 0x5D && m[1] == 0x00 && m[2] == 0x00)) return false;
    sh_set_error("lzma-alone compressed input is not supported (plain, gzip, bzip2 or xz): %s", path);
    return true;
}

int shi_default_threads()
{
    long n = (long)std::thread::hardware_concurrency();
    if (n <= 0) n = 4;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {          // cgroup v2: "<quota> <period>" or "max <period>"
        char q[64]; long p = 0;
        if (fscanf(f, "%63s %ld", q, &p) == 2 && strcmp(q, "max") != 0 && p > 0) n = std::min(n, std::max(1L, atol(q) / p));
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        long q = -1, p = 100000;
        if (fscanf(g, "%ld", &q) != 1) q = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &p) != 1) p = 100000; fclose(h); }
        if (q > 0 && p > 0) n = std::min(n, std::max(1L, q / p));
    }
    return (int)std::min(n, 64L);
}


const char *shi_preset_variant(const std::string &display) { return preset_variant(display); }

// ---- pieces exposed for the CPU tests (no GPU needed) ------------------------------------------------------------
extern "C" sh_status sh_host_get_id(const char *header, char *out, size_t cap)
{
    SH_CHECK(header && out && cap, SH_ERR_BAD_ARG, "sh_host_get_id: null argument");
    std::string id;
    SH_CHECK(get_id(header, id), SH_ERR_IO, "NeedletailFastqHeader: header has no id");
    SH_CHECK(id.size() + 1 <= cap, SH_ERR_BAD_ARG, "id buffer too small");
    memcpy(out, id.c_str(), id.size() + 1);
    return SH_OK;
}

extern "C" sh_status sh_host_filter_fastx(const char *in, const char *out, const char *const *ids, uint64_t n_ids, int32_t extract,
                                          uint64_t *n_in, uint64_t *n_out)
{
    SH_CHECK(in && out && (ids || n_ids == 0), SH_ERR_BAD_ARG, "sh_host_filter_fastx: null argument");
    if (shi_unsupported_compression(in)) return SH_ERR_IO;
    std::unordered_set<std::string> set;
    for (uint64_t i = 0; i < n_ids; ++i) set.insert(ids[i]);
    uint64_t a = 0, b = 0;
    sh_status st = filter_fastx(in, out, set, extract != 0, &a, &b);
    if (n_in) *n_in = a;
    if (n_out) *n_out = b;
    return st;
}

static sh_status read_difference(const char *const *inputs, const char *const *outputs, uint32_t n, uint64_t *reads_in, uint64_t *reads_out,
                                 uint64_t *difference, std::unordered_set<std::string> *diff_ids)
{
    uint64_t in_total = 0, out_total = 0, diff_total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        std::unordered_set<std::string> out_ids;
        bool exists;
        FastxRecord r; std::string id; int st;
        if (!file_is_empty(outputs[i], exists)) {
            FastxReader rd(outputs[i]);
            while ((st = rd.next(r)) == 1) { SH_CHECK(get_id(r.header, id), SH_ERR_IO, "record without an id in %s", outputs[i]); out_ids.insert(id); ++out_total; }
            SH_CHECK(st == 0, SH_ERR_IO, "%s: %s", outputs[i], rd.error.c_str());
        }
        if (!file_is_empty(inputs[i], exists)) {
            FastxReader rd(inputs[i]);
            while ((st = rd.next(r)) == 1) {
                SH_CHECK(get_id(r.header, id), SH_ERR_IO, "record without an id in %s", inputs[i]);
                if (!out_ids.count(id)) { if (diff_ids) diff_ids->insert(id); ++diff_total; }
                ++in_total;
            }
            SH_CHECK(st == 0, SH_ERR_IO, "%s: %s", inputs[i], rd.error.c_str());
        } else {
            SH_CHECK(exists, SH_ERR_IO, "cannot open %s", inputs[i]);
            fprintf(stderr, "[scrubby-hip] warning: Input file is empty: %s\n", inputs[i]);
        }
    }
    *reads_in = in_total; *reads_out = out_total; *difference = diff_total;
    return SH_OK;
}

extern "C" sh_status sh_host_read_difference(const char *const *inputs, const char *const *outputs, uint32_t n, uint64_t *reads_in,
                                             uint64_t *reads_out, uint64_t *difference)
{
    SH_CHECK(inputs && outputs && reads_in && reads_out && difference && n >= 1 && n <= 2, SH_ERR_BAD_ARG, "sh_host_read_difference: bad argument");
    return read_difference(inputs, outputs, n, reads_in, reads_out, difference, nullptr);
}


static std::string json_f64(double v)
{   // serde_json prints f64 with the shortest round-trip form and always a fractional part
    char b[64];
    for (int prec = 1; prec <= 17; ++prec) { snprintf(b, sizeof b, "%.*g", prec, v); if (strtod(b, nullptr) == v) break; }
    std::string o = b;
    if (o.find('.') == std::string::npos && o.find('e') == std::string::npos && o.find("inf") == std::string::npos && o.find("nan") == std::string::npos) o += ".0";
    return o;
}

sh_status shi_write_report_json(const char *const *input, const char *const *output, uint32_t n_files, const char *command,
                                   const ReportSettings &st, const sh_reads_result *r, const char *path)
{
    FILE *f = fopen(path, "wb");
    SH_CHECK(f, SH_ERR_IO, "cannot open %s", path);
    char date[64];
    time_t now = time(nullptr);
    struct tm tmv;
    gmtime_r(&now, &tmv);
    strftime(date, sizeof date, "%Y-%m-%dT%H:%M:%SZ", &tmv);          // to_rfc3339_opts(SecondsFormat::Secs, true)
    auto paths = [&](const char *const *p, uint32_t n) {
        std::string o = "[";
        for (uint32_t i = 0; i < n; ++i) { o += i ? ",\n    \"" : "\n    \""; o += json_escape(p[i]); o += "\""; }
        o += n ? "\n  ]" : "]";
        return o;
    };
    auto strs = [&](const std::vector<std::string> &v) {
        std::string o = "[";
        for (size_t i = 0; i < v.size(); ++i) { o += i ? ",\n      \"" : "\n      \""; o += json_escape(v[i]); o += "\""; }
        o += v.empty() ? "]" : "\n    ]";
        return o;
    };
    auto opt = [&](const char *v) { return v ? "\"" + json_escape(v) + "\"" : std::string("null"); };
    std::string o = "{\n";
    o += "  \"version\": \"1.0.2\",\n";                                 // crate_version!() of the reference this drops into
    o += std::string("  \"date\": \"") + date + "\",\n";
    o += "  \"command\": \"" + json_escape(command ? command : "") + "\",\n";
    o += "  \"input\": " + paths(input, n_files) + ",\n";
    o += "  \"output\": " + paths(output, n_files) + ",\n";
    o += "  \"reads_in\": " + std::to_string(r->reads_in) + ",\n";
    o += "  \"reads_out\": " + std::to_string(r->reads_out) + ",\n";
    o += "  \"reads_removed\": " + std::to_string(r->reads_removed) + ",\n";
    o += "  \"reads_extracted\": " + std::to_string(r->reads_extracted) + ",\n";
    o += "  \"settings\": {\n";
    o += "    \"aligner\": " + opt(st.aligner) + ",\n    \"classifier\": " + opt(st.classifier) + ",\n";
    o += "    \"index\": " + opt(st.index) + ",\n";
    o += "    \"alignment\": " + opt(st.alignment) + ",\n    \"reads\": " + opt(st.reads) + ",\n    \"report\": " + opt(st.report) + ",\n";
    o += "    \"taxa\": " + strs(st.taxa) + ",\n    \"taxa_direct\": " + strs(st.taxa_direct) + ",\n";
    o += "    \"classifier_args\": " + opt(st.classifier_args) + ",\n    \"aligner_args\": null,\n";
    o += "    \"preset\": " + opt(st.preset_variant) + ",\n";
    o += "    \"min_len\": " + std::to_string(st.min_len) + ",\n    \"min_cov\": " + json_f64(st.min_cov) + ",\n    \"min_mapq\": " + std::to_string(st.min_mapq) + ",\n";
    o += std::string("    \"extract\": ") + (st.extract ? "true" : "false") + "\n  }\n}";
    fwrite(o.data(), 1, o.size(), f);
    fclose(f);
    return SH_OK;
}

// after the filter step: counts from re-reading the files, TSV of ids, JSON (ScrubbyReport::create, report.rs:24-57)
static sh_status finish_report(const char *const *input, const char *const *output, uint32_t n_files, bool extract, const char *json,
                               const char *read_ids, const char *command, const ReportSettings &st, sh_reads_result *res)
{
    if (!json && !read_ids) return SH_OK;
    std::unordered_set<std::string> diff_ids;
    uint64_t rin, rout, diff;
    sh_status s = read_difference(input, output, n_files, &rin, &rout, &diff, &diff_ids);
    if (s != SH_OK) return s;
    res->reads_in = rin; res->reads_out = rout;
    res->reads_removed = extract ? 0 : diff; res->reads_extracted = extract ? diff : 0;
    if (read_ids) {
        FastxWriter w(read_ids, 9);
        SH_CHECK(w.ok(), SH_ERR_IO, "%s", w.error.c_str());
        w.put("id\n");
        for (auto &id : diff_ids) w.put(id + "\n");
        SH_CHECK(w.finish(), SH_ERR_IO, "%s: %s", read_ids, w.error.c_str());
    }
    if (json) return shi_write_report_json(input, output, n_files, command, st, res, json);
    return SH_OK;
}

// ---- taxid path: the in-tree decision rule of /root/reference/src/classifier.rs ------------------------------------------
enum TaxLevel { TL_None, TL_Unclassified, TL_NoRank, TL_Root, TL_Domain, TL_Kingdom, TL_Phylum, TL_Class, TL_Order, TL_Family, TL_Genus, TL_Species, TL_Unspecified };

static TaxLevel tax_level_of(const std::string &s)
{   // get_tax_level, classifier.rs:345-373: first letter of a Kraken rank code, or Metabuli's lower-case rank words
    auto sw = [&](const char *p) { return s.compare(0, strlen(p), p) == 0; };
    if (sw("U")) return TL_Unclassified;
    if (sw("no rank")) return TL_NoRank;
    if (sw("R")) return TL_Root;
    if (sw("D") || sw("superkingdom")) return TL_Domain;
    if (sw("K") || sw("kingdom")) return TL_Kingdom;
    if (sw("P") || sw("phylum")) return TL_Phylum;
    if (sw("C") || sw("class")) return TL_Class;
    if (sw("O") || sw("order")) return TL_Order;
    if (sw("F") || sw("family")) return TL_Family;
    if (sw("G") || sw("genus")) return TL_Genus;
    if (sw("S") || sw("species")) return TL_Species;
    return TL_Unspecified;
}

static std::string trim(const std::string &s)
{
    size_t b = 0, e = s.size();
    while (b < e && isspace((unsigned char)s[b])) ++b;
    while (e > b && isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}

static std::vector<std::string> split_tab(const std::string &l)
{
    std::vector<std::string> f;
    size_t b = 0;
    for (;;) { size_t t = l.find('\t', b); f.push_back(l.substr(b, t == std::string::npos ? t : t - b)); if (t == std::string::npos) break; b = t + 1; }
    return f;
}

static bool parse_u64_strict(const std::string &s, uint64_t &v)
{   // Rust str::parse::<u64>: optional '+', digits only, no surrounding white space
    size_t i = s.size() && s[0] == '+' ? 1 : 0;
    if (i >= s.size()) return false;
    v = 0;
    for (; i < s.size(); ++i) { if (s[i] < '0' || s[i] > '9') return false; v = v * 10 + (uint64_t)(s[i] - '0'); }
    return true;
}

static bool read_lines(const char *path, std::vector<std::string> &lines)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    std::string cur; int c;
    while ((c = fgetc(f)) != EOF) { if (c == '\n') { if (!cur.empty() && cur.back() == '\r') cur.pop_back(); lines.push_back(cur); cur.clear(); } else cur += (char)c; }
    if (!cur.empty()) lines.push_back(cur);
    fclose(f);
    return true;
}

// get_taxids_from_report, classifier.rs:124-252
sh_status shi_taxids_from_report(const char *report, const std::vector<std::string> &taxa_in, const std::vector<std::string> &direct_in,
                                    std::unordered_set<std::string> &taxids)
{
    std::vector<std::string> lines;
    SH_CHECK(read_lines(report, lines), SH_ERR_IO, "cannot open %s", report);
    std::vector<std::string> taxa, direct;
    for (auto &t : taxa_in) taxa.push_back(trim(t));
    for (auto &t : direct_in) direct.push_back(trim(t));
    auto has = [](const std::vector<std::string> &v, const std::string &x) { for (auto &e : v) if (e == x) return true; return false; };
    TaxLevel extract_level = TL_None;
    std::string extract_parent;
    for (auto &line : lines) {
        auto f = split_tab(line);
        SH_CHECK(f.size() >= 6, SH_ERR_IO, "malformed report line (%zu fields): %s", f.size(), line.c_str());
        uint64_t reads, reads_direct;
        SH_CHECK(parse_u64_strict(f[1], reads), SH_ERR_IO, "KrakenReportReadFieldConversion: %s", line.c_str());
        SH_CHECK(parse_u64_strict(f[2], reads_direct), SH_ERR_IO, "KrakenReportDirectReadFieldConversion: %s", line.c_str());
        const std::string rank = trim(f[3]), tax_id = trim(f[4]), tax_name = trim(f[5]);
        const TaxLevel lv = tax_level_of(rank);
        if (has(direct, tax_name) || has(direct, tax_id)) taxids.insert(tax_id);            // regardless of rank or read count (Q8)
        if (lv < TL_Domain) continue;                                                       // above Domain: ignored, even inside a window (Q9)
        if (has(taxa, tax_name) || has(taxa, tax_id)) {
            extract_level = lv; extract_parent = tax_name;                                  // opens a sub-tree window (Q10)
            if (reads_direct > 0) taxids.insert(tax_id);
        } else {
            if (extract_level == TL_None) continue;
            if (lv <= extract_level && rank.size() == 1) extract_level = TL_None;           // a single-letter rank at or above closes it
            else if (reads_direct > 0) {
                taxids.insert(tax_id);
                SH_CHECK(!extract_parent.empty(), SH_ERR_IO, "KrakenReportTaxonParent");
            }
        }
    }
    return SH_OK;
}

// get_taxid_reads_kraken / _metabuli, classifier.rs:270-320: ids of the reads assigned to one of the taxids
static sh_status taxid_reads(const std::unordered_set<std::string> &taxids, const char *reads, bool metabuli, std::unordered_set<std::string> &ids)
{
    std::vector<std::string> lines;
    if (!read_lines(reads, lines)) return SH_OK;            // a missing file is an empty set, not an error (Q11)
    const size_t need = metabuli ? 7 : 5;
    for (auto &line : lines) {
        auto f = split_tab(line);
        SH_CHECK(f.size() >= need, SH_ERR_IO, "malformed read classification line (%zu fields): %s", f.size(), line.c_str());
        if (taxids.count(trim(f[2]))) ids.insert(trim(f[1]));
    }
    return SH_OK;
}

extern "C" sh_status sh_classifier_taxids(const char *report, const char *const *taxa, uint32_t n_taxa, const char *const *taxa_direct,
                                          uint32_t n_direct, char *out, size_t cap, uint64_t *n_out)
{
    SH_CHECK(report && out && cap && n_out, SH_ERR_BAD_ARG, "sh_classifier_taxids: null argument");
    std::vector<std::string> t, d;
    for (uint32_t i = 0; i < n_taxa; ++i) t.push_back(taxa[i]);
    for (uint32_t i = 0; i < n_direct; ++i) d.push_back(taxa_direct[i]);
    std::unordered_set<std::string> set;
    sh_status st = shi_taxids_from_report(report, t, d, set);
    if (st != SH_OK) return st;
    std::vector<std::string> v(set.begin(), set.end());
    std::sort(v.begin(), v.end());
    std::string joined;
    for (auto &x : v) { joined += x; joined += '\n'; }
    SH_CHECK(joined.size() + 1 <= cap, SH_ERR_BAD_ARG, "taxid buffer too small (%zu needed)", joined.size() + 1);
    memcpy(out, joined.c_str(), joined.size() + 1);
    *n_out = v.size();
    return SH_OK;
}

// `scrubby classifier`: Cleaner::run_classifier_output (cleaner.rs:177-194) -> parse_classifier_output (:375-382) -> clean_reads
extern "C" sh_status sh_classifier_run(const sh_classifier_config *c, sh_reads_result *res)
{
    SH_CHECK(c && res, SH_ERR_BAD_ARG, "sh_classifier_run: null argument");
    SH_CHECK(c->n_files >= 1 && c->n_files <= 2, SH_ERR_BAD_ARG, "one or two input files are supported (got %u)", c->n_files);
    SH_CHECK(c->report, SH_ERR_BAD_ARG, "MissingClassifierClassificationReport");
    SH_CHECK(c->reads, SH_ERR_BAD_ARG, "MissingClassifierReadClassfications");
    SH_CHECK(c->classifier && (!strcmp(c->classifier, "kraken2") || !strcmp(c->classifier, "metabuli")), SH_ERR_BAD_ARG, "MissingClassifier");
    SH_CHECK(c->n_taxa + c->n_taxa_direct > 0, SH_ERR_BAD_ARG, "MissingTaxa: --taxa or --taxa-direct is required");    // scrubby.rs:843-845
    memset(res, 0, sizeof(*res));
    ReportSettings st;
    for (uint32_t i = 0; i < c->n_taxa; ++i) st.taxa.push_back(c->taxa[i]);
    for (uint32_t i = 0; i < c->n_taxa_direct; ++i) st.taxa_direct.push_back(c->taxa_direct[i]);
    std::unordered_set<std::string> taxids, ids;
    sh_status s = shi_taxids_from_report(c->report, st.taxa, st.taxa_direct, taxids);
    if (s != SH_OK) return s;
    s = taxid_reads(taxids, c->reads, !strcmp(c->classifier, "metabuli"), ids);
    if (s != SH_OK) return s;
    res->n_depleted_ids = ids.size();
    for (uint32_t i = 0; i < c->n_files; ++i) {
        s = filter_fastx(c->input[i], c->output[i], ids, c->extract != 0, nullptr, nullptr);
        if (s != SH_OK) return s;
    }
    st.classifier = c->classifier; st.reads = c->reads; st.report = c->report; st.extract = c->extract != 0;
    return finish_report(c->input, c->output, c->n_files, c->extract != 0, c->json, c->read_ids, c->command, st, res);
}

// ---- Cleaner::run_minimap2_rs + clean_reads + ScrubbyReport::create, collect-then-map form -------------------------
// The reference's own shape (three passes, everything materialised first).  sh_reads_run (sh_stream.cpp) is the streaming
// form and hands over to this one only for the corner case of an empty input file (App. C Q6).
sh_status shi_reads_run_legacy(const sh_reads_config *c, sh_reads_result *res)
{
    SH_CHECK(c && res, SH_ERR_BAD_ARG, "sh_reads_run: null argument");
    SH_CHECK(c->n_files >= 1 && c->n_files <= 2, SH_ERR_BAD_ARG, "one or two input files are supported (got %u)", c->n_files);
    for (uint32_t i = 0; i < c->n_files; ++i) SH_CHECK(c->input[i] && c->output[i], SH_ERR_BAD_ARG, "input/output %u missing", i);
    SH_CHECK(c->index, SH_ERR_BAD_ARG, "MissingAlignmentIndex");
    memset(res, 0, sizeof(*res));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };

    // default preset: Sr for two files, MapOnt for one (/root/reference/src/scrubby.rs:935-951)
    const std::string preset = c->preset && c->preset[0] ? c->preset : (c->n_files == 2 ? "sr" : "map-ont");
    sh_opts opts;
    sh_status st = sh_preset(preset.c_str(), &opts);
    if (st != SH_OK) return st;

    auto t0 = now();
    sh_index *idx = nullptr;
    {
        const std::string ip = c->index;
        const bool cache = ip.size() > 6 && ip.compare(ip.size() - 6, 6, ".shidx") == 0;
        st = cache ? sh_index_load(c->index, c->device, &idx) : sh_index_build_fasta(c->index, &opts, c->device, &idx);
        if (st != SH_OK) return st == SH_ERR_IO ? st : SH_ERR_INDEX;
        // a prebuilt index (.mmi, .shidx) brings its own k and w; they prevail over the preset's, as with minimap2
        { sh_index_info ii; if (sh_index_info_get(idx, &ii) == SH_OK) { opts.k = ii.k; opts.w = ii.w; } }
    }
    auto t1 = now();

    // ingest: every record of R1 then R2 becomes one (id, seq) item (cleaner.rs:484-549)
    std::vector<std::string> ids;
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets(1, 0);
    for (uint32_t i = 0; i < c->n_files; ++i) {
        bool exists;
        if (file_is_empty(c->input[i], exists)) {
            if (!exists) { sh_index_free(idx); sh_set_error("cannot open %s", c->input[i]); return SH_ERR_IO; }
            fprintf(stderr, "[scrubby-hip] warning: Input file is empty: %s\n", c->input[i]);
            continue;
        }
        FastxReader rd(c->input[i]);
        FastxRecord r; std::string id; int s;
        while ((s = rd.next(r)) == 1) {
            if (!get_id(r.header, id)) { sh_index_free(idx); sh_set_error("record without an id in %s", c->input[i]); return SH_ERR_IO; }
            ids.push_back(id);
            bases.insert(bases.end(), r.seq.begin(), r.seq.end());
            offsets.push_back(bases.size());
        }
        if (s != 0) { sh_index_free(idx); sh_set_error("%s: %s", c->input[i], rd.error.c_str()); return SH_ERR_IO; }
    }
    auto t2 = now();

    // classify: aligner.map(..).len() > 0 per record (cleaner.rs:550-558); any per-read error aborts (:566)
    std::vector<uint8_t> flags(ids.size() ? ids.size() : 1, 0);
    bases.resize(bases.size() + 32, 'N');
    sh_stats cs;
    memset(&cs, 0, sizeof cs);
    st = sh_classify_batch(idx, &opts, bases.data(), offsets.data(), ids.size(), flags.data(), nullptr, &cs);
    sh_index_free(idx);
    if (st != SH_OK) return st;
    res->n_ext_unresolved = cs.n_ext_unresolved; res->n_rmq_open = cs.n_rmq_open;
    auto t3 = now();

    // id set (cleaner.rs:564-570), then the filter/writer over each input file (clean_reads, :236-254)
    std::unordered_set<std::string> depleted;
    for (size_t i = 0; i < ids.size(); ++i) if (flags[i] == 1) depleted.insert(ids[i]);
    res->n_depleted_ids = depleted.size();
    for (uint32_t i = 0; i < c->n_files; ++i) {
        st = filter_fastx(c->input[i], c->output[i], depleted, c->extract != 0, nullptr, nullptr);
        if (st != SH_OK) return st;
    }
    auto t4 = now();

    // report (only if -j or -r was given, scrubby.rs:276-278): counts come from re-reading the files
    {
        ReportSettings rs;
        rs.aligner = "minimap2-rs"; rs.index = c->index; rs.preset_variant = preset_variant(preset); rs.extract = c->extract != 0;
        st = finish_report(c->input, c->output, c->n_files, c->extract != 0, c->json, c->read_ids, c->command, rs, res);
        if (st != SH_OK) return st;
    }
    res->ms_index = ms(t0, t1); res->ms_ingest = ms(t1, t2); res->ms_classify = ms(t2, t3); res->ms_write = ms(t3, t4);
    return SH_OK;
}

void shi_mkdir_p(const std::string &dir)
{
    for (size_t i = 1; i <= dir.size(); ++i)
        if (i == dir.size() || dir[i] == '/') mkdir(dir.substr(0, i).c_str(), 0777);
}

// ---- Cleaner::run_kraken (cleaner.rs:288-330) in process: GPU classification instead of the external kraken2 ---------
// kraken.reads / kraken.report are written into the workdir exactly where the reference expects them (:296-297), then
// the same parse_classifier_output + clean_reads steps run on those files.  Divergence: column 5 of
// kraken.reads carries the k-mer and hit-group totals, not Kraken2's positional hit list (the reference only stores it,
// classifier.rs:401-419).
// collect-then-classify form; sh_kraken_run (sh_stream.cpp) hands over for empty inputs and as the A/B baseline
sh_status shi_kraken_run_legacy(const sh_kraken_config *c, sh_reads_result *res)
{
    SH_CHECK(c && res, SH_ERR_BAD_ARG, "sh_kraken_run: null argument");
    SH_CHECK(c->n_files >= 1 && c->n_files <= 2, SH_ERR_BAD_ARG, "one or two input files are supported (got %u)", c->n_files);
    for (uint32_t i = 0; i < c->n_files; ++i) SH_CHECK(c->input[i] && c->output[i], SH_ERR_BAD_ARG, "input/output %u missing", i);
    SH_CHECK(c->db, SH_ERR_BAD_ARG, "MissingClassifierIndex");
    SH_CHECK(c->n_taxa + c->n_taxa_direct > 0, SH_ERR_BAD_ARG, "MissingTaxa: --taxa or --taxa-direct is required");
    memset(res, 0, sizeof(*res));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const bool paired = c->n_files == 2;

    auto t0 = now();
    sh_k2_db *db = nullptr;
    sh_status st = sh_k2_open(c->db, c->device, &db);
    if (st != SH_OK) return st;
    sh_k2_opts opts;                 // k, l, masks and the down-sampling threshold come from the database
    sh_k2_db_opts(db, &opts);
    if (c->confidence >= 0.0) opts.confidence = c->confidence;           // -C "--confidence x"
    if (c->min_hit_groups > 0) opts.min_hit_groups = c->min_hit_groups;   // -C "--minimum-hit-groups n"
    auto t1 = now();

    // ingest: mates interleaved (records 2i, 2i+1); the id Kraken 2 prints for a pair is mate 1's first token with a trailing /1 removed
    std::vector<std::string> ids;
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets(1, 0);
    std::vector<uint32_t> len1, len2;
    auto fail = [&](sh_status s) { sh_k2_free(db); return s; };
    {
        bool ex0 = true, ex1 = true;
        const bool empty0 = file_is_empty(c->input[0], ex0), empty1 = paired ? file_is_empty(c->input[1], ex1) : false;
        if (!ex0 || !ex1) { sh_set_error("cannot open %s", !ex0 ? c->input[0] : c->input[1]); return fail(SH_ERR_IO); }
        if (!empty0 && !(paired && empty1)) {
            FastxReader r1(c->input[0]);
            FastxReader *r2 = paired ? new FastxReader(c->input[1]) : nullptr;
            FastxRecord a, b; std::string id; int s1;
            while ((s1 = r1.next(a)) == 1) {
                if (!get_id(a.header, id)) { delete r2; sh_set_error("record without an id in %s", c->input[0]); return fail(SH_ERR_IO); }
                if (paired && id.size() > 2 && id.compare(id.size() - 2, 2, "/1") == 0) id.resize(id.size() - 2);
                ids.push_back(id);
                bases.insert(bases.end(), a.seq.begin(), a.seq.end()); offsets.push_back(bases.size()); len1.push_back((uint32_t)a.seq.size());
                if (paired) {
                    const int s2 = r2->next(b);
                    if (s2 != 1) { std::string e = s2 == 0 ? "fewer records than mate 1" : r2->error; delete r2; sh_set_error("%s: %s", c->input[1], e.c_str()); return fail(SH_ERR_IO); }
                    bases.insert(bases.end(), b.seq.begin(), b.seq.end()); offsets.push_back(bases.size()); len2.push_back((uint32_t)b.seq.size());
                }
            }
            if (s1 != 0) { std::string e = r1.error; delete r2; sh_set_error("%s: %s", c->input[0], e.c_str()); return fail(SH_ERR_IO); }
            if (paired && r2->next(b) != 0) { delete r2; sh_set_error("%s: more records than mate 1", c->input[1]); return fail(SH_ERR_IO); }
            delete r2;
        }
    }
    auto t2 = now();

    std::vector<sh_k2_result> results(std::max<size_t>(ids.size(), 1));
    bases.resize(bases.size() + 64, 'N');
    st = sh_k2_classify_batch(db, &opts, bases.data(), offsets.data(), offsets.size() - 1, paired ? 1 : 0, results.data(), nullptr);
    if (st != SH_OK) return fail(st);
    auto t3 = now();

    // kraken.reads and kraken.report in the workdir (cleaner.rs:291-297)
    std::string dir = c->workdir && c->workdir[0] ? c->workdir : (getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp");
    if (c->workdir && c->workdir[0]) shi_mkdir_p(dir);                         // create_dir_all (cleaner.rs:293)
    const std::string reads_path = dir + "/kraken.reads", report_path = dir + "/kraken.report";
    {
        FILE *f = fopen(reads_path.c_str(), "w");
        if (!f) { sh_set_error("cannot write %s", reads_path.c_str()); return fail(SH_ERR_IO); }
        for (size_t i = 0; i < ids.size(); ++i) {
            const sh_k2_result &r = results[i];
            if (paired) fprintf(f, "%c\t%s\t%u\t%u|%u\tkmers=%u groups=%u\n", r.call ? 'C' : 'U', ids[i].c_str(), r.taxid, len1[i], len2[i], r.total_kmers, r.hit_groups);
            else fprintf(f, "%c\t%s\t%u\t%u\tkmers=%u groups=%u\n", r.call ? 'C' : 'U', ids[i].c_str(), r.taxid, len1[i], r.total_kmers, r.hit_groups);
        }
        if (fclose(f) != 0) { sh_set_error("short write to %s", reads_path.c_str()); return fail(SH_ERR_IO); }
    }
    st = sh_k2_write_report(db, results.data(), ids.size(), report_path.c_str());
    sh_k2_free(db);
    if (st != SH_OK) return st;

    // parse_classifier_output (cleaner.rs:375-382) + clean_reads on what was just written
    ReportSettings rs;
    for (uint32_t i = 0; i < c->n_taxa; ++i) rs.taxa.push_back(c->taxa[i]);
    for (uint32_t i = 0; i < c->n_taxa_direct; ++i) rs.taxa_direct.push_back(c->taxa_direct[i]);
    std::unordered_set<std::string> taxids, dep;
    st = shi_taxids_from_report(report_path.c_str(), rs.taxa, rs.taxa_direct, taxids);
    if (st != SH_OK) return st;
    st = taxid_reads(taxids, reads_path.c_str(), false, dep);
    if (st != SH_OK) return st;
    res->n_depleted_ids = dep.size();
    for (uint32_t i = 0; i < c->n_files; ++i) {
        st = filter_fastx(c->input[i], c->output[i], dep, c->extract != 0, nullptr, nullptr);
        if (st != SH_OK) return st;
    }
    auto t4 = now();
    rs.classifier = "kraken2"; rs.index = c->db; rs.extract = c->extract != 0;
    st = finish_report(c->input, c->output, c->n_files, c->extract != 0, c->json, c->read_ids, c->command, rs, res);
    if (st != SH_OK) return st;
    res->ms_index = ms(t0, t1); res->ms_ingest = ms(t1, t2); res->ms_classify = ms(t2, t3); res->ms_write = ms(t3, t4);
    return SH_OK;
}

// ---- `scrubby alignment`: Cleaner::run_aligner_output (cleaner.rs:206-219) with ReadAlignment (src/alignment.rs:33-114,242-276) ----
static bool gz_lines(const char *path, std::vector<std::string> &lines)
{   // (any container sh_codec.h reads)
    shc::In f;
    if (!f.open(path)) return false;
    std::vector<char> buf(1 << 16);
    std::string cur;
    for (;;) {
        const long got = f.read(buf.data(), buf.size());
        if (got < 0) return false;      // a stream that stops in the middle
        if (got == 0) break;
        for (long i = 0; i < got; ++i) {
            if (buf[i] == '\n') { if (!cur.empty() && cur.back() == '\r') cur.pop_back(); lines.push_back(cur); cur.clear(); }
            else cur += buf[i];
        }
    }
    if (!cur.empty()) lines.push_back(cur);
    return true;
}

static sh_status alignment_ids(const char *path, const char *format, uint64_t min_len, double min_cov, uint32_t min_mapq, std::unordered_set<std::string> &ids)
{
    std::string fmt = format ? format : "";
    if (fmt.empty()) {       // Path::extension(): only the LAST extension counts, so "x.paf.gz" is not recognised (alignment.rs:47-55)
        std::string p = path;
        size_t dot = p.rfind('.'), slash = p.rfind('/');
        std::string ext = (dot != std::string::npos && (slash == std::string::npos || dot > slash)) ? p.substr(dot + 1) : "";
        if (ext == "paf" || ext == "gaf") fmt = "paf"; else if (ext == "txt") fmt = "txt";
        else { sh_set_error("AlignmentInputFormatNotRecognized: %s", path); return SH_ERR_BAD_ARG; }
    }
    if (fmt == "gaf") fmt = "paf";
    SH_CHECK(fmt == "paf" || fmt == "txt", SH_ERR_BAD_ARG, "AlignmentInputFormatInvalid: %s (SAM/BAM/CRAM need the reference's htslib feature)", fmt.c_str());
    bool exists;
    if (file_is_empty(path, exists)) { SH_CHECK(exists, SH_ERR_IO, "cannot open %s", path); return SH_OK; }
    std::vector<std::string> lines;
    SH_CHECK(gz_lines(path, lines), SH_ERR_IO, "cannot read %s (missing, or a truncated gzip stream)", path);
    if (fmt == "txt") { for (auto &l : lines) ids.insert(l); return SH_OK; }       // one id per line, verbatim
    for (auto &l : lines) {
        auto f = split_tab(l);
        SH_CHECK(f.size() >= 12, SH_ERR_IO, "malformed PAF line (%zu fields): %s", f.size(), l.c_str());
        uint64_t qlen, qs, qe, v, mapq;
        for (int k : {1, 2, 3, 6, 7, 8, 9, 10, 11}) SH_CHECK(parse_u64_strict(f[k], v), SH_ERR_IO, "PAF integer field %d: %s", k + 1, l.c_str());
        parse_u64_strict(f[1], qlen); parse_u64_strict(f[2], qs); parse_u64_strict(f[3], qe); parse_u64_strict(f[11], mapq);
        SH_CHECK(mapq <= 255, SH_ERR_IO, "PAF mapq out of range: %s", l.c_str());
        SH_CHECK(qe >= qs, SH_ERR_IO, "PAF query end before start: %s", l.c_str());
        const uint64_t qalen = qe - qs;
        const double qcov = qlen == 0 ? 0.0 : (double)qalen / (double)qlen;
        if ((qalen >= min_len || qcov >= min_cov) && mapq >= min_mapq) ids.insert(f[0]);
    }
    return SH_OK;
}

extern "C" sh_status sh_alignment_run(const sh_alignment_config *c, sh_reads_result *res)
{
    SH_CHECK(c && res, SH_ERR_BAD_ARG, "sh_alignment_run: null argument");
    SH_CHECK(c->n_files >= 1 && c->n_files <= 2, SH_ERR_BAD_ARG, "one or two input files are supported (got %u)", c->n_files);
    SH_CHECK(c->alignment, SH_ERR_BAD_ARG, "MissingAlignment");
    memset(res, 0, sizeof(*res));
    std::unordered_set<std::string> ids;
    sh_status s = alignment_ids(c->alignment, c->format, c->min_len, c->min_cov, c->min_mapq, ids);
    if (s != SH_OK) return s;
    res->n_depleted_ids = ids.size();
    for (uint32_t i = 0; i < c->n_files; ++i) {
        s = filter_fastx(c->input[i], c->output[i], ids, c->extract != 0, nullptr, nullptr);
        if (s != SH_OK) return s;
    }
    ReportSettings st;
    st.alignment = c->alignment; st.min_len = c->min_len; st.min_cov = c->min_cov; st.min_mapq = c->min_mapq; st.extract = c->extract != 0;
    return finish_report(c->input, c->output, c->n_files, c->extract != 0, c->json, c->read_ids, c->command, st, res);
}
