// sh_stream.cpp — streaming host side of the replaced path (§8f N1).
//
// The reference materialises every (id, sequence) before the first read is mapped (/root/reference/src/cleaner.rs:484-549,
// the author's own note at :445-447), filters by re-reading the inputs (:236-254, :731-760) and counts by re-reading inputs
// AND outputs (src/utils.rs:250-285).  Same results here, different shape:
//
//   pass 1  one reader thread per input file cuts the decompressed byte stream into chunks at record boundaries and parses
//           them in place (a record = seven offsets into its chunk, no per-record allocation).  A device thread classifies
//           chunk after chunk (sh_classify_device, one context) while the readers carry on, and folds the ids of the flagged
//           records into a flat hash set (cleaner.rs:564-570).
//   pass 2  per file, pool workers decide every record by its id (FastqCleaner::clean_reads), format the kept ones - a run
//           of kept records already in the writer's form is one memcpy - and, for .gz outputs, deflate each chunk as its
//           own gzip member (level 6, niffler's default in get_fastx_writer, utils.rs:56-74); output sizes are published in chunk
//           order, which fixes each chunk's file offset, and the workers pwrite side by side.
//           Chunks retained in host memory (budget: half of MemAvailable, SCRUBBY_HIP_RETAIN_MB) are not read again;
//           past the budget the files are streamed a second time.
//   report  reads_in / reads_out / difference come from pass 2's counters: keep/drop is a function of the id alone, so
//           "input records whose id is not among that file's output ids" (utils.rs:265-279) = records not written.
//
// An empty input file (App. C Q6: warning, output not created, counts from whatever the output path holds) is the one
// case handed to the collect-then-map form in sh_host.cpp.
#include "sh_host.h"
#include <zlib.h>
#include "sh_codec.h"
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_set>

namespace {

constexpr size_t NPOS = ~(size_t)0;

struct Rec {
    uint32_t beg, hdr, hdr_len, seq, seq_len, qual, end;   // offsets into Chunk::data; [beg, end) = the record as it stands in the file
    uint8_t fastq, canon;                                  // canon: [beg, end) is byte for byte what the writer would emit
};

// chunk buffers: 2 MiB-aligned and advised as huge pages - 6.5 GB of 4 KiB pages cost ~1 s to fault in (pass 1's readers) and
// ~1 s to give back (munmap serialises on the address-space lock, whichever thread frees)
inline char *chunk_alloc(size_t bytes)
{
    const size_t HP = 2u << 20;
    if (bytes < 2 * HP) return (char *)malloc(bytes);
    const size_t sz = (bytes + HP - 1) / HP * HP;
    void *p = nullptr;
    if (posix_memalign(&p, HP, sz) != 0) return nullptr;
    madvise(p, sz, MADV_HUGEPAGE);
    return (char *)p;
}

struct DevBuf;
struct Chunk {
    char *data = nullptr;
    size_t len = 0;
    std::vector<Rec> recs;
    std::vector<uint8_t> bases;       // the read batch of include/scrubby_hip.h; dropped once classified
    std::vector<uint64_t> offsets;
    uint32_t file = 0, range = 0;     // which input file, which byte range of it (0: the only one), and which chunk of that
    size_t seq_no = 0;
    bool parsed = false;
    DevBuf *dev = nullptr;     // pass 1: where a parse worker has put the batch in HBM
    uint64_t n_bases = 0;
    uint32_t max_len = 0;
    std::vector<uint8_t> flags;       // pass 1: the device's verdicts, on their way to the fold threads
    Chunk() = default;
    Chunk(const Chunk &) = delete;
    ~Chunk() { free(data); }
    size_t footprint() const { return len + recs.capacity() * sizeof(Rec); }
};

size_t eol(const char *d, size_t from, size_t n)
{
    if (from >= n) return NPOS;
    const void *q = memchr(d + from, '\n', n - from);
    return q ? (size_t)((const char *)q - d) : NPOS;
}

// parses records from the start of c; *consumed = start of the first incomplete record.  false = malformed (error set).
// batch: also lay out the read batch (pass 1); pass 2 only needs the records.
bool parse_chunk(Chunk &c, bool at_eof, bool batch, size_t *consumed, std::string &error)
{
    auto fail = [&](const std::string &m) { error = m; return false; };
    char *d = c.data;
    const size_t n = c.len;
    size_t p = 0;
    uint64_t n_bases = 0;
    for (;;) {
        for (;;) {      // blank lines before a record
            if (p < n && d[p] == '\n') { ++p; continue; }
            if (p + 1 < n && d[p] == '\r' && d[p + 1] == '\n') { p += 2; continue; }
            if (p + 1 == n && d[p] == '\r' && at_eof) { ++p; continue; }
            break;
        }
        *consumed = p;
        if (p >= n || (p + 1 == n && d[p] == '\r')) break;
        Rec r{};
        r.beg = (uint32_t)p;
        bool cr = false;
        auto strip = [&](size_t b, size_t e) { if (e > b && d[e - 1] == '\r') { cr = true; return e - 1; } return e; };
        if (d[p] == '@') {
            const size_t e1 = eol(d, p + 1, n);
            const size_t e2 = e1 == NPOS ? NPOS : eol(d, e1 + 1, n);
            const size_t e3 = e2 == NPOS ? NPOS : eol(d, e2 + 1, n);
            if (e3 == NPOS) {
                if (!at_eof) break;
                return fail("truncated FASTQ record: " + std::string(d + p + 1, std::min<size_t>(n - p - 1, 80)));
            }
            size_t e4 = eol(d, e3 + 1, n);
            bool term = true;
            if (e4 == NPOS) {
                if (!at_eof) break;
                if (e3 + 1 >= n) return fail("truncated FASTQ record: " + std::string(d + p + 1, strip(p + 1, e1) - p - 1));
                e4 = n; term = false;
            }
            r.fastq = 1;
            r.hdr = (uint32_t)(p + 1); r.hdr_len = (uint32_t)(strip(p + 1, e1) - (p + 1));
            r.seq = (uint32_t)(e1 + 1); r.seq_len = (uint32_t)(strip(e1 + 1, e2) - (e1 + 1));
            const size_t pl = strip(e2 + 1, e3) - (e2 + 1);
            if (pl == 0 || d[e2 + 1] != '+') return fail("truncated FASTQ record: " + std::string(d + r.hdr, r.hdr_len));
            r.qual = (uint32_t)(e3 + 1);
            const size_t ql = strip(e3 + 1, e4) - (e3 + 1);
            if (ql != r.seq_len) return fail("sequence/quality length mismatch: " + std::string(d + r.hdr, r.hdr_len));
            r.end = (uint32_t)(term ? e4 + 1 : e4);
            r.canon = !cr && pl == 1 && term;
        } else if (d[p] == '>') {
            size_t e1 = eol(d, p + 1, n);
            if (e1 == NPOS) { if (!at_eof) break; e1 = n; }
            // the record runs to the next line that starts with '>'
            size_t next = NPOS;
            for (size_t s = e1 + 1; s < n;) {
                if (d[s] == '>') { next = s; break; }
                const size_t e = eol(d, s, n);
                if (e == NPOS) break;
                s = e + 1;
            }
            if (next == NPOS && !at_eof) break;
            const size_t rec_end = next == NPOS ? n : next;
            r.hdr = (uint32_t)(p + 1); r.hdr_len = (uint32_t)(strip(p + 1, e1) - (p + 1));
            const size_t q = std::min(e1 + 1, rec_end);
            size_t w = q, lines = 0;
            bool term_last = e1 < n;
            for (size_t s = q; s < rec_end;) {     // join the sequence lines in place
                size_t e = eol(d, s, rec_end);
                bool t = true;
                if (e == NPOS) { e = rec_end; t = false; }
                const size_t l = strip(s, e) - s;
                if (l) { if (w != s) memmove(d + w, d + s, l); w += l; }
                ++lines; term_last = t;
                s = t ? e + 1 : e;
            }
            r.seq = (uint32_t)q; r.seq_len = (uint32_t)(w - q);
            r.qual = 0;
            r.end = (uint32_t)rec_end;
            r.canon = !cr && lines == 1 && term_last;
        } else {
            return fail("not a FASTA/FASTQ record: " + std::string(d + p, std::min<size_t>(n - p, 40)));
        }
        n_bases += r.seq_len;
        c.recs.push_back(r);
        p = r.end;
    }
    c.parsed = true;
    if (!batch) return true;
    // the read batch: concatenated sequences + offsets
    c.offsets.resize(c.recs.size() + 1);
    c.bases.resize(n_bases);
    uint64_t o = 0;
    for (size_t i = 0; i < c.recs.size(); ++i) {
        c.offsets[i] = o;
        memcpy(c.bases.data() + o, d + c.recs[i].seq, c.recs[i].seq_len);
        o += c.recs[i].seq_len;
    }
    c.offsets[c.recs.size()] = o;
    return true;
}


// A record boundary near the end of d[0, n), found without parsing from the start: the last line that starts a FASTA record
// ('>'), or the last line that starts with '@' whose second line after starts with '+'.  A quality line may start with
// '@' too, but then the second line after it is a sequence; on input odd enough to fool this (a sequence line starting
// with '+' ...) the chunk BEFORE the cut ends in an incomplete record, its parse fails, and the caller falls back to the
// sequential reader - so a cut that survives parsing is a true boundary (induction from the start of the file).
size_t find_split(const char *d, size_t n, bool fasta)
{
    size_t pos = n;
    while (pos > 0) {
        const void *q = memrchr(d, '\n', pos - 1);
        if (!q) return NPOS;
        const size_t nl = (size_t)((const char *)q - d), L = nl + 1;
        pos = nl;
        if (L >= n || nl == 0) continue;
        if (fasta) { if (d[L] == '>') return L; continue; }
        if (d[L] != '@') continue;
        const size_t e1 = eol(d, L, n);
        const size_t e2 = e1 == NPOS ? NPOS : eol(d, e1 + 1, n);
        if (e2 == NPOS || e2 + 1 >= n) continue;
        if (d[e2 + 1] == '+') return L;
    }
    return NPOS;
}

// The same guess looking forward: the first record start after d[0] (d[0] itself may be anywhere in a line).
size_t find_split_forward(const char *d, size_t n, bool fasta)
{
    size_t nl = eol(d, 0, n);
    while (nl != NPOS && nl + 1 < n) {
        const size_t L = nl + 1;
        if (fasta) { if (d[L] == '>') return L; }
        else if (d[L] == '@') {
            const size_t e1 = eol(d, L, n);
            const size_t e2 = e1 == NPOS ? NPOS : eol(d, e1 + 1, n);
            if (e2 == NPOS || e2 + 1 >= n) return NPOS;
            if (d[e2 + 1] == '+') return L;
        }
        nl = eol(d, L, n);
    }
    return NPOS;
}

// Plain (not gzip) regular files big enough to bother are read by several readers, each over a byte range that starts at a
// guessed record boundary.  Verified like every other cut: each chunk must parse to its last byte, so a range that starts inside
// a record makes the previous range's last chunk fail and pass 1 is rerun with the sequential reader.
struct FilePlan { int fd = -1; bool fasta = false; std::vector<uint64_t> bounds; };     // bounds.size() - 1 ranges; empty: one stream reader
FilePlan plan_ranges(const char *path, int max_ranges, uint64_t min_range_bytes)
{
    FilePlan p;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return p;
    struct stat sb;
    unsigned char head[256];
    ssize_t hl = 0;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || (hl = pread(fd, head, sizeof head, 0)) < 2 || (head[0] == 0x1f && head[1] == 0x8b)) { close(fd); return p; }
    ssize_t q = 0;
    while (q < hl && (head[q] == '\n' || head[q] == '\r')) ++q;
    if (q >= hl || (head[q] != '>' && head[q] != '@')) { close(fd); return p; }
    const uint64_t size = (uint64_t)sb.st_size;
    const int R = (int)std::min<uint64_t>((uint64_t)std::max(1, max_ranges), size / std::max<uint64_t>(min_range_bytes, 1));
    if (R < 2) { close(fd); return p; }
    p.fasta = head[q] == '>';
    p.bounds.push_back(0);
    std::vector<char> win(4u << 20);
    for (int k = 1; k < R; ++k) {
        const uint64_t nominal = size / (uint64_t)R * (uint64_t)k;
        const ssize_t got = pread(fd, win.data(), win.size(), (off_t)nominal);
        if (got <= 0) continue;
        const size_t L = find_split_forward(win.data(), (size_t)got, p.fasta);
        if (L != NPOS && nominal + L > p.bounds.back()) p.bounds.push_back(nominal + L);     // no boundary in the window: the range merges with its neighbour
    }
    p.bounds.push_back(size);
    if (p.bounds.size() < 3) { p.bounds.clear(); close(fd); return p; }
    p.fd = fd;
    return p;
}

// Cuts a FASTA / FASTQ byte stream (plain, gzip, bzip2 or xz: sh_codec.h sniffs the container) into chunks that end on record
// boundaries.  Accepts what the legacy line reader accepts: 4-line FASTQ, multi-line FASTA, CRLF, blank lines between
// records, a last line without '\n'.  Sequential mode parses as it cuts; split-only mode cuts at find_split() and leaves
// the parsing (and its verification) to whoever takes the chunk.
class ChunkReader {
    shc::In in_;
    bool have_in_ = false;
    int fd_ = -1;                     // byte-range source (plain files): pread from pos_ up to end_
    uint64_t pos_ = 0, end_ = 0;
    size_t target_;
    bool batch_, split_only_;
    int fasta_ = -1;
    std::vector<char> carry_;
    bool eof_ = false;

public:
    std::string error;
    ChunkReader(const char *path, size_t target, bool batch, bool split_only = false) : target_(std::max<size_t>(target, 64)), batch_(batch), split_only_(split_only)
    {
        have_in_ = in_.open(path);
        if (!have_in_) error = in_.error;
    }
    // one byte range [begin, end) of a plain file; `fasta` as the whole file's first record says (the range may start anywhere in it)
    ChunkReader(int fd, uint64_t begin, uint64_t end, bool fasta, size_t target, bool batch) : fd_(fd), pos_(begin), end_(end), target_(std::max<size_t>(target, 64)),
                                                                                            batch_(batch), split_only_(true), fasta_(fasta ? 1 : 0) {}
    ChunkReader(const ChunkReader &) = delete;
    bool ok() const { return have_in_ || fd_ >= 0; }

    // 1 = a chunk with at least one record, 0 = end of input, -1 = malformed input (error set)
    int next(Chunk &c)
    {
        if (eof_ && carry_.empty()) return 0;
        size_t cap = carry_.size() + target_;
        c.data = chunk_alloc(cap + 1);
        if (!c.data) { error = "out of host memory"; return -1; }
        c.len = carry_.size();
        if (c.len) memcpy(c.data, carry_.data(), c.len);
        carry_.clear();
        for (;;) {
            while (!eof_ && c.len < cap && fd_ >= 0) {
                const size_t want = (size_t)std::min<uint64_t>(cap - c.len, end_ - pos_);
                const ssize_t got = want ? pread(fd_, c.data + c.len, want, (off_t)pos_) : 0;
                if (got < 0 || (want && got == 0)) { error = "read error"; return -1; }
                pos_ += (uint64_t)got; c.len += (size_t)got;
                if (pos_ >= end_) eof_ = true;
            }
            while (!eof_ && c.len < cap) {
                // a short read is the end of the stream; a truncated or corrupt stream is an error (-1), never a silently partial input
                const size_t want = cap - c.len;
                const long got = in_.read(c.data + c.len, want);
                if (got < 0) { error = in_.error; return -1; }
                if ((size_t)got < want) eof_ = true;
                c.len += (size_t)got;
            }
            size_t consumed = 0;
            if (split_only_) {
                if (fasta_ < 0) { size_t p = 0; while (p < c.len && (c.data[p] == '\n' || c.data[p] == '\r')) ++p; if (p < c.len) fasta_ = c.data[p] == '>'; }
                consumed = eof_ ? c.len : find_split(c.data, c.len, fasta_ == 1);
                if (consumed != NPOS) {
                    carry_.assign(c.data + consumed, c.data + c.len);
                    c.len = consumed;
                    return c.len ? 1 : 0;
                }
            } else {
                c.recs.clear();
                c.recs.reserve(c.len / 256 + 16);
                if (!parse_chunk(c, eof_, batch_, &consumed, error)) return -1;
                if (!c.recs.empty() || eof_) {
                    carry_.assign(c.data + consumed, c.data + c.len);
                    c.len = consumed;
                    return c.recs.empty() ? 0 : 1;
                }
            }
            // not one complete record in cap bytes: a record longer than the chunk - grow and read on
            if (cap >= (3ull << 30)) { error = "record longer than 3 GiB"; return -1; }
            cap = std::min<size_t>(cap * 2, 3ull << 30);
            char *nd = chunk_alloc(cap + 1);
            if (!nd) { error = "out of host memory"; return -1; }
            memcpy(nd, c.data, c.len);
            free(c.data);
            c.data = nd;
        }
    }
};

bool file_is_empty(const char *path, bool &exists)
{
    shc::In f;
    exists = f.open(path);
    if (!exists) return true;
    char ch;
    return f.read(&ch, 1) == 0;      // (a stream that errors before its first byte is not empty: the reader reports it)
}

// get_id (utils.rs:91-103): first whitespace-delimited token of the header
inline bool id_of(const char *h, uint32_t n, const char **id, uint32_t *len)
{
    auto sp = [](unsigned char ch) { return ch == ' ' || (ch >= '\t' && ch <= '\r'); };
    uint32_t b = 0;
    while (b < n && sp((unsigned char)h[b])) ++b;
    uint32_t e = b;
    while (e < n && !sp((unsigned char)h[e])) ++e;
    *id = h + b; *len = e - b;
    return e > b;
}

inline uint64_t hash_bytes(const char *s, size_t n)
{
    uint64_t h = 0x9E3779B97F4A7C15ULL ^ (n * 0xff51afd7ed558ccdULL);
    while (n >= 8) { uint64_t v; memcpy(&v, s, 8); h = (h ^ v) * 0x9FB21C651E98DF25ULL; h ^= h >> 29; s += 8; n -= 8; }
    if (n) { uint64_t v = 0; memcpy(&v, s, n); h = (h ^ v) * 0x9FB21C651E98DF25ULL; h ^= h >> 29; }
    h *= 0xff51afd7ed558ccdULL;
    return h ^ (h >> 32);
}

// the HashSet<String> of cleaner.rs:564-570: open addressing over (hash, offset:40 | length:24) into one byte arena;
// built by one thread, read by many
class IdSet {
    struct Ent { uint64_t h, ol; };
    static constexpr uint64_t EMPTY = ~0ull;
    std::vector<Ent> tab_;
    std::vector<char> arena_;
    size_t n_ = 0, mask_;
    void grow()
    {
        std::vector<Ent> old;
        old.swap(tab_);
        tab_.assign(old.size() * 2, Ent{0, EMPTY});
        mask_ = tab_.size() - 1;
        for (const Ent &e : old)
            if (e.ol != EMPTY) { size_t i = e.h & mask_; while (tab_[i].ol != EMPTY) i = (i + 1) & mask_; tab_[i] = e; }
    }
public:
    IdSet() : tab_(1 << 16, Ent{0, EMPTY}), mask_((1 << 16) - 1) {}
    size_t size() const { return n_; }
    static uint32_t clip(uint32_t len) { return len >= (1u << 24) ? (1u << 24) - 1 : len; }     // ids are compared on their first 16 MiB
    bool insert(const char *s, uint32_t len) { len = clip(len); return insert_h(s, len, hash_bytes(s, len)); }
    bool contains(const char *s, uint32_t len) const { len = clip(len); return contains_h(s, len, hash_bytes(s, len)); }
    bool insert_h(const char *s, uint32_t len, uint64_t h)
    {
        if ((n_ + 1) * 10 > tab_.size() * 7) grow();
        for (size_t i = h & mask_;; i = (i + 1) & mask_) {
            Ent &e = tab_[i];
            if (e.ol == EMPTY) {
                e.h = h; e.ol = ((uint64_t)arena_.size() << 24) | len;
                arena_.insert(arena_.end(), s, s + len);
                ++n_;
                return true;
            }
            if (e.h == h && (e.ol & 0xFFFFFF) == len && !memcmp(arena_.data() + (e.ol >> 24), s, len)) return false;
        }
    }
    bool contains_h(const char *s, uint32_t len, uint64_t h) const
    {
        for (size_t i = h & mask_;; i = (i + 1) & mask_) {
            const Ent &e = tab_[i];
            if (e.ol == EMPTY) return false;
            if (e.h == h && (e.ol & 0xFFFFFF) == len && !memcmp(arena_.data() + (e.ol >> 24), s, len)) return true;
        }
    }
    template <class F> void for_each(F f) const
    {
        for (const Ent &e : tab_) if (e.ol != EMPTY) f(arena_.data() + (e.ol >> 24), (uint32_t)(e.ol & 0xFFFFFF));
    }
};

// the same set, split 64 ways by the top hash bits so that several threads can fold ids into it at once; reads take no lock
class ShardedIdSet {
    static constexpr int N = 64;
    struct Shard { IdSet set; std::mutex mu; };
    std::unique_ptr<Shard[]> sh_{new Shard[N]};
public:
    bool insert(const char *s, uint32_t len)
    {
        len = IdSet::clip(len);
        const uint64_t h = hash_bytes(s, len);
        Shard &x = sh_[h >> 58];
        std::lock_guard<std::mutex> lk(x.mu);
        return x.set.insert_h(s, len, h);
    }
    bool contains(const char *s, uint32_t len) const
    {
        len = IdSet::clip(len);
        const uint64_t h = hash_bytes(s, len);
        return sh_[h >> 58].set.contains_h(s, len, h);
    }
    size_t size() const { size_t n = 0; for (int i = 0; i < N; ++i) n += sh_[i].set.size(); return n; }
    template <class F> void for_each(F f) const { for (int i = 0; i < N; ++i) sh_[i].set.for_each(f); }
    void clear() { sh_.reset(new Shard[N]); }
};

bool ends_with(const std::string &s, const char *suf)
{
    const size_t k = strlen(suf);
    return s.size() >= k && s.compare(s.size() - k, k, suf) == 0;
}

// one complete gzip member
bool gz_member(const char *src, size_t n, int level, std::string &dst)
{
    z_stream zs{};
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    dst.resize(deflateBound(&zs, (uLong)n) + 64);
    zs.next_out = (Bytef *)dst.data();
    size_t out_left = dst.size(), in_left = n;
    const Bytef *in = (const Bytef *)src;
    int rc = Z_OK;
    for (;;) {
        const uInt ai = (uInt)std::min<size_t>(in_left, 1u << 30), ao = (uInt)std::min<size_t>(out_left, 1u << 30);
        zs.next_in = (Bytef *)in; zs.avail_in = ai; zs.avail_out = ao;
        rc = deflate(&zs, in_left == ai ? Z_FINISH : Z_NO_FLUSH);
        const size_t used_in = ai - zs.avail_in, used_out = ao - zs.avail_out;
        in += used_in; in_left -= used_in; out_left -= used_out;
        if (rc != Z_OK && rc != Z_BUF_ERROR) break;
        if (used_in == 0 && used_out == 0) break;        // no progress: out of space
    }
    const bool ok = rc == Z_STREAM_END;
    dst.resize(dst.size() - out_left);
    deflateEnd(&zs);
    return ok;
}

// FastqCleaner::clean_reads over one chunk (cleaner.rs:731-760): keep a record iff (id in set) == extract
struct FilterOut {
    std::string bytes;
    uint64_t n_in = 0, n_out = 0;
    std::vector<std::string> dropped;       // ids of records not written; collected only on request
    std::string error;
};

void filter_chunk(const Chunk &c, const ShardedIdSet &ids, bool extract, bool gz, bool want_dropped, FilterOut &o)
{
    std::string plain;
    plain.reserve(c.len + 64);
    const char *d = c.data;
    size_t run_b = NPOS, run_e = 0;
    auto flush = [&]() { if (run_b != NPOS) { plain.append(d + run_b, run_e - run_b); run_b = NPOS; } };
    for (const Rec &r : c.recs) {
        const char *id; uint32_t il;
        if (!id_of(d + r.hdr, r.hdr_len, &id, &il)) { o.error = "record without an id"; return; }
        ++o.n_in;
        if (ids.contains(id, il) != extract) { if (want_dropped) o.dropped.emplace_back(id, il); continue; }
        ++o.n_out;
        if (r.canon) {
            if (run_b != NPOS && run_e == r.beg) run_e = r.end;
            else { flush(); run_b = r.beg; run_e = r.end; }
        } else {     // needletail record.write(writer, None): '\n' endings, bare '+', full header, sequence on one line
            flush();
            plain += r.fastq ? '@' : '>';
            plain.append(d + r.hdr, r.hdr_len); plain += '\n';
            plain.append(d + r.seq, r.seq_len); plain += '\n';
            if (r.fastq) { plain += "+\n"; plain.append(d + r.qual, r.seq_len); plain += '\n'; }
        }
    }
    flush();
    if (!gz) { o.bytes.swap(plain); return; }
    if (!plain.empty() && !gz_member(plain.data(), plain.size(), 6, o.bytes)) o.error = "deflate failed";
}

// pass 2 for one file: sources chunks (retained, or streamed again), filters on `n_workers` threads, writes in order
struct FileFilter {
    const char *in_path, *out_path;
    std::vector<std::shared_ptr<Chunk>> *retained;          // nullptr: stream the file again; else consumed: a chunk is released by the worker that filtered it
    size_t chunk_bytes;
    const ShardedIdSet *ids;
    bool extract, want_dropped;
    int n_workers;
    uint64_t n_in = 0, n_out = 0;
    std::vector<std::string> dropped;
    std::string error;

    sh_status run()
    {
        const std::string op = out_path;
        const bool gz = ends_with(op, ".gz");
        // bzip2 / xz outputs (CompressionExt::from_path, utils.rs:28-36): ONE stream, fed in chunk order by whichever worker holds the turn
        // (the gzip form is multi-member and deflated by the workers side by side; these two stay single-stream for every reader's sake)
        const shc::Kind okind = shc::kind_by_extension(op);
        const bool seq = okind == shc::Kind::Bzip2 || okind == shc::Kind::Xz;
        shc::Out enc;
        std::unique_ptr<ChunkReader> rd;
        if (!retained) {
            rd.reset(new ChunkReader(in_path, chunk_bytes, false));
            SH_CHECK(rd->ok(), SH_ERR_IO, "cannot open %s", in_path);
        }
        int fd = -1;
        if (seq) { SH_CHECK(enc.open(op, 6), SH_ERR_IO, "%s", enc.error.c_str()); }
        else { fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0666); SH_CHECK(fd >= 0, SH_ERR_IO, "cannot open %s", out_path); }

        // Chunks are taken in file order; a worker filters (and deflates) its chunk, then publishes the size of its output
        // in chunk order - which fixes its offset in the file - and writes with pwrite beside the other workers.  Only the
        // size publication is ordered, so the copies into the page cache run in parallel.
        std::mutex src_mu, out_mu;
        std::condition_variable cv;
        size_t next = 0, published = 0;
        uint64_t cur_off = 0;
        bool src_done = false, failed = false, io_ok = true;
        std::atomic<uint64_t> us_filter{0}, us_wait{0}, us_write{0};       // SCRUBBY_HIP_DBG_HOST=1
        auto tick = [] { return std::chrono::steady_clock::now(); };
        auto usd = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };

        auto worker = [&]() {
            for (;;) {
                std::shared_ptr<Chunk> ch;
                size_t i = 0;
                {
                    std::lock_guard<std::mutex> lk(src_mu);
                    if (src_done) break;
                    if (retained) {
                        if (next >= retained->size()) { src_done = true; break; }
                        ch = (*retained)[next];
                    } else {
                        ch = std::make_shared<Chunk>();
                        const int r = rd->next(*ch);
                        if (r <= 0) {
                            src_done = true;
                            if (r < 0) { std::lock_guard<std::mutex> l2(out_mu); failed = true; error = std::string(in_path) + ": " + rd->error; cv.notify_all(); }
                            break;
                        }
                    }
                    i = next++;
                }
                FilterOut o;
                const auto t0 = tick();
                filter_chunk(*ch, *ids, extract, gz, want_dropped, o);
                if (retained) (*retained)[i].reset();      // each index is visited once; the unmap of ~64 MB happens here, on this worker
                ch.reset();
                const auto t1 = tick();
                us_filter += usd(t0, t1);
                uint64_t my_off = 0;
                {
                    std::unique_lock<std::mutex> lk(out_mu);
                    cv.wait(lk, [&] { return failed || published == i; });
                    us_wait += usd(t1, tick());
                    if (!o.error.empty() && !failed) { failed = true; error = o.error + " in " + in_path; }
                    if (!failed) {
                        my_off = cur_off; cur_off += o.bytes.size();
                        n_in += o.n_in; n_out += o.n_out;
                        for (auto &s : o.dropped) dropped.push_back(std::move(s));
                        if (seq && !enc.write(o.bytes.data(), o.bytes.size())) { io_ok = false; failed = true; }
                    }
                    published = i + 1;          // also on failure: nobody may wait for this chunk for ever
                    cv.notify_all();
                    if (failed) break;
                }
                const auto t2 = tick();
                size_t done = seq ? o.bytes.size() : 0;
                while (done < o.bytes.size()) {
                    const ssize_t w = pwrite(fd, o.bytes.data() + done, o.bytes.size() - done, (off_t)(my_off + done));
                    if (w <= 0) { std::lock_guard<std::mutex> lk(out_mu); io_ok = false; failed = true; cv.notify_all(); break; }
                    done += (size_t)w;
                }
                us_write += usd(t2, tick());
            }
        };
        const auto t_run = tick();
        std::vector<std::thread> pool;
        for (int t = 0; t < n_workers; ++t) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
        if (const char *e = getenv("SCRUBBY_HIP_DBG_HOST")) if (*e == '1')
            fprintf(stderr, "[scrubby-hip] pass 2 %s: %d workers, wall %.0f ms; filter%s %.0f ms, waiting for the offset %.0f ms, pwrite %.0f ms (summed over workers)\n", out_path,
                    n_workers, usd(t_run, tick()) / 1e3, gz ? " + deflate" : "", us_filter / 1e3, us_wait / 1e3, us_write / 1e3);
        if (gz && cur_off == 0 && error.empty() && io_ok) {     // no record kept: still a valid (empty) gzip stream, as gzclose would leave
            std::string m;
            gz_member("", 0, 6, m);
            io_ok = pwrite(fd, m.data(), m.size(), 0) == (ssize_t)m.size();
        }
        if (seq) io_ok = enc.close() && io_ok;
        else io_ok = (close(fd) == 0) && io_ok;
        SH_CHECK(error.empty(), SH_ERR_IO, "%s", error.c_str());
        SH_CHECK(io_ok, SH_ERR_IO, "short write to %s", out_path);
        return SH_OK;
    }
};

size_t env_mb(const char *name, size_t dflt_bytes)
{
    const char *e = getenv(name);
    return e && *e ? (size_t)strtoull(e, nullptr, 10) << 20 : dflt_bytes;
}

size_t retain_budget()
{
    size_t avail_kb = 0;
    if (FILE *f = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f))
            if (sscanf(line, "MemAvailable: %zu kB", &avail_kb) == 1) break;
        fclose(f);
    }
    return env_mb("SCRUBBY_HIP_RETAIN_MB", avail_kb * 1024 / 2);
}

bool write_id_table(const char *path, const std::string &body)
{   // ReadDifference::write_read_ids (utils.rs:207-214): header `id`, one id per line; container by extension (level 9, legacy writer)
    const std::string p = path;
    if (ends_with(p, ".gz")) {
        std::string out;
        if (!gz_member(body.data(), body.size(), 9, out)) return false;
        FILE *f = fopen(path, "wb");
        if (!f) return false;
        const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
        return (fclose(f) == 0) && ok;
    }
    shc::Out o;      // plain, bzip2, xz
    if (!o.open(p, 9)) return false;
    const bool ok = o.write(body.data(), body.size());
    return o.close() && ok;
}

// a batch's place in HBM; filled by a parse worker (its own stream), consumed by the device thread
struct DevBuf {
    uint8_t *d_bases = nullptr, *d_flags = nullptr;
    uint64_t *d_off = nullptr;
    size_t cap_bases = 0, cap_reads = 0;
    void release()
    {
        if (d_bases) hipFree(d_bases);
        if (d_flags) hipFree(d_flags);
        if (d_off) hipFree(d_off);
        d_bases = d_flags = nullptr; d_off = nullptr; cap_bases = cap_reads = 0;
    }
    // H2D from pageable memory is staged by the calling thread: done by the parse workers, it scales with them
    sh_status upload(Chunk &c, hipStream_t s)
    {
        const uint64_t n = c.recs.size(), nb = c.offsets[n];
        if (nb + 64 > cap_bases) { if (d_bases) hipFree(d_bases); d_bases = nullptr; cap_bases = nb + nb / 8 + 64; SH_HIP(hipMalloc(&d_bases, cap_bases)); }
        if (n + 1 > cap_reads) {
            if (d_off) hipFree(d_off);
            if (d_flags) hipFree(d_flags);
            d_off = nullptr; d_flags = nullptr;
            cap_reads = n + n / 8 + 1;
            SH_HIP(hipMalloc(&d_off, cap_reads * 8));
            SH_HIP(hipMalloc(&d_flags, cap_reads));
        }
        if (nb) SH_HIP(hipMemcpyAsync(d_bases, c.bases.data(), nb, hipMemcpyHostToDevice, s));
        SH_HIP(hipMemcpyAsync(d_off, c.offsets.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
        SH_HIP(hipStreamSynchronize(s));
        c.n_bases = nb;
        c.max_len = 0;
        for (const Rec &r : c.recs) c.max_len = std::max(c.max_len, r.seq_len);
        std::vector<uint8_t>().swap(c.bases);
        std::vector<uint64_t>().swap(c.offsets);
        return SH_OK;
    }
};

class DevBufPool {
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<DevBuf *> free_;
    std::vector<std::unique_ptr<DevBuf>> all_;
    bool abort_ = false;
public:
    explicit DevBufPool(size_t n) { for (size_t i = 0; i < n; ++i) { all_.emplace_back(new DevBuf); free_.push_back(all_.back().get()); } }
    ~DevBufPool() { for (auto &b : all_) b->release(); }
    DevBuf *take()
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return abort_ || !free_.empty(); });
        if (abort_) return nullptr;
        DevBuf *b = free_.back();
        free_.pop_back();
        return b;
    }
    void give(DevBuf *b) { std::lock_guard<std::mutex> lk(mu_); free_.push_back(b); cv_.notify_one(); }
    void stop() { std::lock_guard<std::mutex> lk(mu_); abort_ = true; cv_.notify_all(); }
};

// dst[i] = src[i] + add: a chunk's read offsets moved to its place in a concatenated batch
__global__ void k_rebase_offsets(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, uint64_t n, uint64_t add)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + add;
}

// The context the last run left behind (one per process): taken by the next run if it was built with the same options and is large
// enough, re-pointed at that run's index (shi_ctx_rebind).  Freed by sh_release_cached_ctx(), never kept with SCRUBBY_HIP_CTX_CACHE=0.
struct CtxCache {
    std::mutex mu;
    sh_ctx *ctx = nullptr;
    sh_opts opts{};
    std::string env;          // the environment switches a context captures when it is created
    static std::string env_sig()
    {
        std::string e;
        for (const char *v : {"SCRUBBY_HIP_ARENA_MB", "SCRUBBY_HIP_NO_FLAG_STOP", "SCRUBBY_HIP_NO_PAIR", "SCRUBBY_HIP_PAIR_MIN", "SCRUBBY_HIP_NO_S1", "SCRUBBY_HIP_NO_LEMMA", "SCRUBBY_HIP_EXT_MB", "SCRUBBY_HIP_DBG", "SCRUBBY_HIP_LEXT_P_KB", "SCRUBBY_HIP_RMQ_EXACT_MAX", "SCRUBBY_HIP_RMQ_ONE_LANE"}) {
            const char *x = getenv(v); e += x ? x : "-"; e += '|';
        }
        return e;
    }
    sh_ctx *take(const sh_opts &o, const sh_index *idx)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!ctx) return nullptr;
        sh_ctx *c = ctx;
        ctx = nullptr;
        if (memcmp(&opts, &o, sizeof(o)) != 0 || env != env_sig() || shi_ctx_rebind(c, idx) != SH_OK) { sh_ctx_destroy(c); return nullptr; }
        return c;
    }
    void give(sh_ctx *c, const sh_opts &o)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (ctx) sh_ctx_destroy(ctx);
        ctx = nullptr;
        const char *off = getenv("SCRUBBY_HIP_CTX_CACHE");
        if (off && atoi(off) == 0) { sh_ctx_destroy(c); return; }
        ctx = c; opts = o; env = env_sig();
    }
    void release()
    {
        std::lock_guard<std::mutex> lk(mu);
        if (ctx) sh_ctx_destroy(ctx);
        ctx = nullptr;
    }
};
CtxCache g_ctx_cache;
extern "C" sh_status sh_release_cached_ctx(void) { g_ctx_cache.release(); return SH_OK; }

// the device thread's side of pass 1: one context (minimap2's thread buffer), one stream; kernels and the flags' way back
struct DeviceSide {
    const sh_index *idx;
    sh_opts opts;
    sh_ctx *ctx = nullptr;
    uint64_t ctx_reads = 0, ctx_bases = 0;
    uint32_t ctx_len = 0;
    hipStream_t s = nullptr;
    uint64_t n_ext_unresolved = 0, n_rmq_open = 0;      // summed over the calls (sh_stats)

    ~DeviceSide()
    {
        if (ctx) g_ctx_cache.give(ctx, opts);
        if (s) hipStreamDestroy(s);
        if (cat_bases) hipFree(cat_bases);
        if (cat_off) hipFree(cat_off);
        if (cat_flags) hipFree(cat_flags);
    }
    // A context for 4 Mi short reads holds ~35 GB; creating one costs ~0.6 s (the driver wipes HBM it hands out), so it must not grow in
    // small steps: the first one is sized from the input files (hint_reads, set by the caller), a later one at least 4x its predecessor.
    uint64_t hint_reads = 0;
    bool tried_cache = false;
    sh_status ensure_ctx(uint64_t n, uint64_t nb, uint32_t max_len)
    {
        if (!s) SH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        if (!ctx && !tried_cache) {
            tried_cache = true;
            ctx = g_ctx_cache.take(opts, idx);
            if (ctx) { ctx_reads = shi_ctx_max_reads(ctx); ctx_bases = shi_ctx_max_bases(ctx); ctx_len = shi_ctx_max_len(ctx); }
        }
        if (!ctx || n > ctx_reads || nb > ctx_bases || max_len > ctx_len) {
            if (ctx) { sh_ctx_destroy(ctx); ctx = nullptr; }
            const uint64_t avg = n ? (nb + n - 1) / n : 1;
            uint64_t want = std::max<uint64_t>(n + n / 4 + 1024, std::min<uint64_t>(hint_reads, (5ull << 20)));
            want = std::max<uint64_t>(want, std::min<uint64_t>(ctx_reads * 4, 5ull << 20));
            const uint64_t base_cap = avg > 1024 ? (1ull << 30) : (5ull << 30);      // long reads: the front end's per-base buffers bound the batch
            if (want * avg > base_cap) want = std::max<uint64_t>(n + n / 4 + 1024, base_cap / std::max<uint64_t>(avg, 1));
            ctx_reads = std::max<uint64_t>(ctx_reads, want);
            ctx_bases = std::max<uint64_t>(ctx_bases, std::max<uint64_t>(nb + nb / 4 + 4096, want * avg + want * avg / 8 + 4096));
            ctx_len = std::max<uint32_t>(ctx_len, max_len <= 1024 ? std::max<uint32_t>(max_len, 256) : (uint32_t)std::min<uint64_t>((uint64_t)max_len * 5 / 4, UINT32_MAX));
            const auto t0 = std::chrono::steady_clock::now();
            sh_status st = sh_ctx_create(idx, &opts, ctx_reads, ctx_bases, ctx_len, &ctx);
            ms_ctx += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            ++n_ctx;
            if (st != SH_OK) return st;
        }
        return SH_OK;
    }
    double ms_ctx = 0; int n_ctx = 0;
    sh_status classify(Chunk &c)
    {
        const uint64_t n = c.recs.size(), nb = c.n_bases;
        sh_status st = ensure_ctx(n, nb, c.max_len);
        if (st != SH_OK) return st;
        c.flags.resize(n);
        sh_stats cs;
        st = sh_classify_device(ctx, c.dev->d_bases, c.dev->d_off, n, nb, c.dev->d_flags, nullptr, s, &cs);
        if (st != SH_OK) return st;
        n_ext_unresolved += cs.n_ext_unresolved; n_rmq_open += cs.n_rmq_open;
        SH_HIP(hipMemcpyAsync(c.flags.data(), c.dev->d_flags, n, hipMemcpyDeviceToHost, s));
        SH_HIP(hipStreamSynchronize(s));
        return SH_OK;
    }
    // Several chunks in ONE call: a call has a floor of tens of milliseconds whatever its size (the passes of the repeat path and of the
    // extension stage synchronise with the host, and one satellite read's cluster is thousands of dependent DP steps), so the chunks that
    // piled up while the previous call ran are concatenated in HBM (device-to-device copies, offsets rebased by a kernel) and classified
    // together.  The slower the device side is relative to the parsers, the larger its batches get.
    uint8_t *cat_bases = nullptr, *cat_flags = nullptr; uint64_t *cat_off = nullptr;
    uint64_t cat_cap_bases = 0, cat_cap_reads = 0;
    std::vector<uint8_t> h_flags;
    sh_status classify_many(std::vector<std::shared_ptr<Chunk>> &v)
    {
        if (v.size() == 1) return classify(*v[0]);
        uint64_t n = 0, nb = 0; uint32_t max_len = 0;
        for (auto &c : v) { n += c->recs.size(); nb += c->n_bases; max_len = std::max(max_len, c->max_len); }
        sh_status st = ensure_ctx(n, nb, max_len);
        if (st != SH_OK) return st;
        if (nb + 64 > cat_cap_bases) { if (cat_bases) hipFree(cat_bases); cat_bases = nullptr; cat_cap_bases = nb + nb / 4 + 64; SH_HIP(hipMalloc(&cat_bases, cat_cap_bases)); }
        if (n + 1 > cat_cap_reads) {
            if (cat_off) hipFree(cat_off);
            if (cat_flags) hipFree(cat_flags);
            cat_off = nullptr; cat_flags = nullptr;
            cat_cap_reads = n + n / 4 + 1;
            SH_HIP(hipMalloc(&cat_off, cat_cap_reads * 8));
            SH_HIP(hipMalloc(&cat_flags, cat_cap_reads));
        }
        uint64_t r0 = 0, b0 = 0;
        for (auto &c : v) {
            const uint64_t ni = c->recs.size();
            if (c->n_bases) SH_HIP(hipMemcpyAsync(cat_bases + b0, c->dev->d_bases, c->n_bases, hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(k_rebase_offsets, dim3((uint32_t)((ni + 1 + 255) / 256)), dim3(256), 0, s, c->dev->d_off, cat_off + r0, ni + 1, b0);
            r0 += ni; b0 += c->n_bases;
        }
        sh_stats cs;
        st = sh_classify_device(ctx, cat_bases, cat_off, n, nb, cat_flags, nullptr, s, &cs);
        if (st != SH_OK) return st;
        n_ext_unresolved += cs.n_ext_unresolved; n_rmq_open += cs.n_rmq_open;
        h_flags.resize(n);
        SH_HIP(hipMemcpyAsync(h_flags.data(), cat_flags, n, hipMemcpyDeviceToHost, s));
        SH_HIP(hipStreamSynchronize(s));
        r0 = 0;
        for (auto &c : v) { const uint64_t ni = c->recs.size(); c->flags.assign(h_flags.begin() + r0, h_flags.begin() + r0 + ni); r0 += ni; }
        return SH_OK;
    }
};

struct ChunkQueue {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Chunk>> q;
    size_t cap;
    int producers;
    bool abort = false;
    bool push(std::shared_ptr<Chunk> c)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return abort || q.size() < cap; });
        if (abort) return false;
        q.push_back(std::move(c));
        cv.notify_all();
        return true;
    }
    void producer_done() { std::lock_guard<std::mutex> lk(mu); --producers; cv.notify_all(); }
    std::shared_ptr<Chunk> pop()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !q.empty() || producers == 0; });
        if (q.empty()) return nullptr;
        auto c = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return c;
    }
    std::shared_ptr<Chunk> try_pop()      // what is waiting right now, never blocks
    {
        std::lock_guard<std::mutex> lk(mu);
        if (q.empty()) return nullptr;
        auto c = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return c;
    }
    void stop() { std::lock_guard<std::mutex> lk(mu); abort = true; cv.notify_all(); }
};

constexpr sh_status SH_RETRY_SEQUENTIAL = -1000;     // internal: never crosses the ABI

// pass 1 of sh_reads_run.  parallel: one reader per file only reads and cuts (find_split), `threads` workers parse the
// chunks (a chunk that does not parse cleanly to its last byte -> SH_RETRY_SEQUENTIAL); else the readers parse as they cut.
struct Pass1 {
    const sh_reads_config *c;
    const sh_index *idx;
    sh_opts opts;
    size_t chunk_bytes, budget;
    int threads;
    std::vector<std::shared_ptr<Chunk>> *kept;      // [2]
    ShardedIdSet *depleted;
    double classify_ms = 0;
    uint64_t n_ext_unresolved = 0, n_rmq_open = 0;
    std::atomic<bool> retain{true};
    std::atomic<size_t> kept_bytes{0};
    std::atomic<uint64_t> us_read{0}, us_rpush{0}, us_parse{0}, us_ppush{0}, us_dev_wait{0}, us_ids{0}, us_h2d{0};     // SCRUBBY_HIP_DBG_HOST=1
    std::mutex mu;                                  // errors + the kept vectors
    sh_status err_st = SH_OK;
    std::string err_msg;
    ChunkQueue rawq, devq, foldq;
    DevBufPool pool{24};         // chunks uploaded and waiting for the device thread: it classifies all of them in one call
    uint64_t n_calls = 0;
    double ms_ctx = 0; int n_ctx = 0;

    static std::chrono::steady_clock::time_point tick() { return std::chrono::steady_clock::now(); }
    static uint64_t us(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); }

    void set_err(sh_status s, const std::string &m)
    {
        { std::lock_guard<std::mutex> lk(mu); if (err_st == SH_OK) { err_st = s; err_msg = m; } }
        rawq.stop(); devq.stop(); foldq.stop(); pool.stop();
    }
    bool has_err() { std::lock_guard<std::mutex> lk(mu); return err_st != SH_OK; }

    FilePlan plan[2];
    std::vector<std::vector<std::shared_ptr<Chunk>>> kept_r[2];      // [file][range]: chunks in range order; concatenated into kept[] at the end

    void reader(uint32_t i, uint32_t range, bool parallel)
    {
        std::unique_ptr<ChunkReader> rd;
        if (!plan[i].bounds.empty()) rd.reset(new ChunkReader(plan[i].fd, plan[i].bounds[range], plan[i].bounds[range + 1], plan[i].fasta, chunk_bytes, true));
        else rd.reset(new ChunkReader(c->input[i], chunk_bytes, true, parallel));
        if (!rd->ok()) set_err(SH_ERR_IO, std::string("cannot open ") + c->input[i]);
        else
            for (size_t k = 0;; ++k) {
                auto ch = std::make_shared<Chunk>();
                const auto t0 = tick();
                const int r = rd->next(*ch);
                const auto t1 = tick();
                us_read += us(t0, t1);
                if (r < 0) { set_err(SH_ERR_IO, std::string(c->input[i]) + ": " + rd->error); break; }
                if (r == 0) break;
                ch->file = i; ch->range = range; ch->seq_no = k;
                const bool pushed = rawq.push(ch);
                us_rpush += us(t1, tick());
                if (!pushed) break;
            }
        rawq.producer_done();
    }

    void parser(bool parallel)
    {
        hipStream_t hs = nullptr;
        if (hipSetDevice(idx->device) != hipSuccess || hipStreamCreateWithFlags(&hs, hipStreamNonBlocking) != hipSuccess) set_err(SH_ERR_HIP, "parse worker: no HIP stream");
        while (auto ch = rawq.pop()) {
            if (has_err()) continue;
            const auto t0 = tick();
            if (!ch->parsed) {
                size_t consumed = 0;
                std::string e;
                ch->recs.reserve(ch->len / 256 + 16);
                if (!parse_chunk(*ch, true, true, &consumed, e) || consumed != ch->len) {
                    set_err(parallel ? SH_RETRY_SEQUENTIAL : SH_ERR_IO, std::string(c->input[ch->file]) + ": " + e);
                    continue;
                }
            }
            bool ids_ok = true;
            for (const Rec &rc : ch->recs) {
                const char *id; uint32_t il;
                if (!id_of(ch->data + rc.hdr, rc.hdr_len, &id, &il)) { ids_ok = false; break; }
            }
            if (!ids_ok) { set_err(SH_ERR_IO, std::string("record without an id in ") + c->input[ch->file]); continue; }
            if (retain.load()) {
                if (kept_bytes.fetch_add(ch->footprint()) + ch->footprint() > budget) retain.store(false);
                else {
                    std::lock_guard<std::mutex> lk(mu);
                    auto &v = kept_r[ch->file][ch->range];
                    if (v.size() <= ch->seq_no) v.resize(ch->seq_no + 1);
                    v[ch->seq_no] = ch;
                }
            }
            const auto t1 = tick();
            us_parse += us(t0, t1);
            if (!ch->recs.empty()) {
                DevBuf *b = pool.take();
                if (!b) continue;
                const auto t2 = tick();
                ch->dev = b;
                if (b->upload(*ch, hs) != SH_OK) { set_err(SH_ERR_HIP, sh_last_error()); continue; }
                us_h2d += us(t2, tick());
                devq.push(ch);
            }
            us_ppush += us(t1, tick());
        }
        if (hs) hipStreamDestroy(hs);
        devq.producer_done();
    }

    // cleaner.rs:564-570: ids of the mapped records into the set; an empty read's Err aborts the run (:552,566)
    void folder()
    {
        while (auto ch = foldq.pop()) {
            if (has_err()) continue;
            const auto t0 = tick();
            const size_t n = ch->recs.size();
            for (size_t r = 0; r < n; ++r) {
                const uint8_t f = ch->flags[r];
                if (f == 1) {
                    const char *id; uint32_t il;
                    id_of(ch->data + ch->recs[r].hdr, ch->recs[r].hdr_len, &id, &il);
                    depleted->insert(id, il);
                } else if (f == 2) {
                    set_err(SH_ERR_EMPTY_READ, "Sequence is empty (read " + std::string(ch->data + ch->recs[r].hdr, ch->recs[r].hdr_len) + ")");
                    break;
                }
            }
            std::vector<uint8_t>().swap(ch->flags);
            us_ids += us(t0, tick());
        }
    }

    sh_status run(bool parallel)
    {
        retain.store(budget > 0);
        const int n_parse = std::max(1, threads);
        // plain files of >= 256 MB: up to 4 range readers each, none under 128 MB (SCRUBBY_HIP_READERS; 1 = one stream reader per file, as for gzip)
        const int max_readers = getenv("SCRUBBY_HIP_READERS") ? atoi(getenv("SCRUBBY_HIP_READERS")) : 4;
        const uint64_t min_range = env_mb("SCRUBBY_HIP_RANGE_MB", 128ull << 20);
        int n_readers = 0;
        for (uint32_t i = 0; i < c->n_files; ++i) {
            if (parallel && max_readers > 1) plan[i] = plan_ranges(c->input[i], max_readers, min_range);
            const size_t nr = plan[i].bounds.empty() ? 1 : plan[i].bounds.size() - 1;
            kept_r[i].assign(nr, {});
            n_readers += (int)nr;
        }
        rawq.cap = (size_t)n_parse + 2; rawq.producers = n_readers;
        const int n_fold = std::max(1, std::min(8, threads / 2));          // id-set inserts are cache-missy (~0.4 us each): with range readers they bound pass 1
        devq.cap = 32; devq.producers = n_parse;
        const bool coalesce = !(getenv("SCRUBBY_HIP_NO_COALESCE") && *getenv("SCRUBBY_HIP_NO_COALESCE") == '1');          // the device-buffer pool is what bounds the chunks in flight
        foldq.cap = 8; foldq.producers = 1;
        std::vector<std::thread> thr;
        for (uint32_t i = 0; i < c->n_files; ++i)
            for (uint32_t rg = 0; rg < kept_r[i].size(); ++rg) thr.emplace_back([this, i, rg, parallel] { reader(i, rg, parallel); });
        for (int t = 0; t < n_parse; ++t) thr.emplace_back([this, parallel] { parser(parallel); });
        for (int t = 0; t < n_fold; ++t) thr.emplace_back([this] { folder(); });
        {
            DeviceSide dev{idx, opts};
            if (coalesce) {        // records to expect, from the file sizes (~300 B of FASTQ per 150-base record; gzip ~4x): sizes the first context
                uint64_t bytes = 0;
                for (uint32_t i = 0; i < c->n_files; ++i) {
                    struct stat sb;
                    if (stat(c->input[i], &sb) == 0) bytes += (uint64_t)sb.st_size * (ends_with(c->input[i], ".gz") ? 4 : 1);
                }
                dev.hint_reads = bytes / 300;
            }
            if (hipSetDevice(idx->device) != hipSuccess) set_err(SH_ERR_HIP, "hipSetDevice failed");
            for (;;) {
                const auto w0 = tick();
                auto ch = devq.pop();
                us_dev_wait += us(w0, tick());
                if (!ch) break;
                std::vector<std::shared_ptr<Chunk>> batch{ch};
                uint64_t bn = ch->recs.size(), bb = ch->n_bases;
                while (coalesce && bn < (4ull << 20) && bb < (1ull << 30)) {      // whatever else has arrived, up to 4 Mi records / 1 GiB of bases a call
                    auto more = devq.try_pop();
                    if (!more) break;
                    bn += more->recs.size(); bb += more->n_bases;
                    batch.push_back(std::move(more));
                }
                if (!has_err()) {
                    const auto a = tick();
                    const sh_status st = dev.classify_many(batch);
                    classify_ms += us(a, tick()) / 1e3;
                    ++n_calls;
                    if (st != SH_OK) set_err(st, sh_last_error());
                }
                for (auto &b : batch) {
                    pool.give(b->dev);
                    b->dev = nullptr;
                    if (!has_err()) foldq.push(b);
                }
            }
            foldq.producer_done();
            ms_ctx = dev.ms_ctx; n_ctx = dev.n_ctx;
            n_ext_unresolved = dev.n_ext_unresolved; n_rmq_open = dev.n_rmq_open;
        }
        for (auto &t : thr) t.join();
        for (uint32_t i = 0; i < c->n_files; ++i) {
            if (plan[i].fd >= 0) close(plan[i].fd);
            for (auto &v : kept_r[i]) for (auto &ch : v) if (ch) kept[i].push_back(std::move(ch));       // file order = range order, then chunk order
            kept_r[i].clear();
        }
        if (const char *e = getenv("SCRUBBY_HIP_DBG_HOST")) if (*e == '1')
            fprintf(stderr, "[scrubby-hip] pass 1 (%s): readers read %.0f ms, blocked on push %.0f ms | %d parse workers: parse %.0f ms, H2D %.0f ms, blocked %.0f ms | device thread: "
                    "waiting %.0f ms, classify %.0f ms in %llu calls (of which %.0f ms creating %d contexts) | %d fold threads: id set %.0f ms\n", parallel ? "cut + parallel parse" : "sequential parse", us_read / 1e3, us_rpush / 1e3, n_parse,
                    us_parse / 1e3, us_h2d / 1e3, (us_ppush - us_h2d) / 1e3, us_dev_wait / 1e3, classify_ms, (unsigned long long)n_calls, ms_ctx, n_ctx, n_fold, us_ids / 1e3);
        if (err_st != SH_OK) { if (err_st != SH_RETRY_SEQUENTIAL) sh_set_error("%s", err_msg.c_str()); return err_st; }
        return SH_OK;
    }
};

}  // namespace

// ---- test hook (no GPU): the chunked filter of pass 2 on its own, ids given by the caller --------------------------
extern "C" sh_status sh_host_filter_fastx_stream(const char *in, const char *out, const char *const *ids, uint64_t n_ids, int32_t extract,
                                                 uint64_t chunk_bytes, int32_t threads, int32_t retain, uint64_t *n_in, uint64_t *n_out)
{
    SH_CHECK(in && out && (ids || n_ids == 0), SH_ERR_BAD_ARG, "sh_host_filter_fastx_stream: null argument");
    if (shi_unsupported_compression(in)) return SH_ERR_IO;
    ShardedIdSet set;
    for (uint64_t i = 0; i < n_ids; ++i) set.insert(ids[i], (uint32_t)strlen(ids[i]));
    std::vector<std::shared_ptr<Chunk>> kept;
    FilePlan plan;
    if (retain == 3) plan = plan_ranges(in, 4, std::max<uint64_t>(chunk_bytes, 64));      // 3: several byte ranges, each cut and parsed like 2
    if (retain == 3 && !plan.bounds.empty()) {
        for (size_t k = 0; k + 1 < plan.bounds.size(); ++k) {
            ChunkReader rd(plan.fd, plan.bounds[k], plan.bounds[k + 1], plan.fasta, chunk_bytes, false);
            for (;;) {
                auto c = std::make_shared<Chunk>();
                const int r = rd.next(*c);
                if (r < 0) { close(plan.fd); sh_set_error("%s: %s", in, rd.error.c_str()); return SH_ERR_IO; }
                if (r == 0) break;
                size_t consumed = 0;
                std::string e;
                if (!(parse_chunk(*c, true, false, &consumed, e) && consumed == c->len)) { close(plan.fd); sh_set_error("boundary guess failed (%s): the sequential reader decides", e.c_str()); return SH_ERR_IO; }
                if (!c->recs.empty()) kept.push_back(std::move(c));
            }
        }
        close(plan.fd);
    } else if (retain) {       // 1: the sequential reader; 2: cut at guessed boundaries, then parse each chunk on its own (pass 1's parallel form)
        ChunkReader rd(in, chunk_bytes, false, retain >= 2);
        SH_CHECK(rd.ok(), SH_ERR_IO, "cannot open %s", in);
        for (;;) {
            auto c = std::make_shared<Chunk>();
            const int r = rd.next(*c);
            SH_CHECK(r >= 0, SH_ERR_IO, "%s: %s", in, rd.error.c_str());
            if (r == 0) break;
            if (retain >= 2) {
                size_t consumed = 0;
                std::string e;
                SH_CHECK(parse_chunk(*c, true, false, &consumed, e) && consumed == c->len, SH_ERR_IO, "boundary guess failed (%s): the sequential reader decides", e.c_str());
                if (c->recs.empty()) continue;
            }
            kept.push_back(std::move(c));
        }
    }
    FileFilter ff{in, out, retain ? &kept : nullptr, (size_t)chunk_bytes, &set, extract != 0, false, std::max(1, threads)};
    sh_status st = ff.run();
    if (n_in) *n_in = ff.n_in;
    if (n_out) *n_out = ff.n_out;
    return st;
}

// ---- Cleaner::run_minimap2_rs + clean_reads + ScrubbyReport::create (cleaner.rs:443-575, :236-254; report.rs:24-57) ------
extern "C" sh_status sh_reads_run(const sh_reads_config *c, sh_reads_result *res)
{
    SH_CHECK(c && res, SH_ERR_BAD_ARG, "sh_reads_run: null argument");
    SH_CHECK(c->n_files >= 1 && c->n_files <= 2, SH_ERR_BAD_ARG, "one or two input files are supported (got %u)", c->n_files);
    for (uint32_t i = 0; i < c->n_files; ++i) SH_CHECK(c->input[i] && c->output[i], SH_ERR_BAD_ARG, "input/output %u missing", i);
    SH_CHECK(c->index, SH_ERR_BAD_ARG, "MissingAlignmentIndex");
    if (const char *e = getenv("SCRUBBY_HIP_LEGACY_HOST")) if (*e == '1') return shi_reads_run_legacy(c, res);     // A/B switch for bench.py
    for (uint32_t i = 0; i < c->n_files; ++i) {
        bool exists;
        if (shi_unsupported_compression(c->input[i])) return SH_ERR_IO;
        if (file_is_empty(c->input[i], exists)) return shi_reads_run_legacy(c, res);
    }
    memset(res, 0, sizeof(*res));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const int threads = c->threads > 0 ? c->threads : shi_default_threads();
    const size_t chunk_bytes = env_mb("SCRUBBY_HIP_CHUNK_MB", 64ull << 20);
    const size_t budget = retain_budget();

    // default preset: Sr for two files, MapOnt for one (/root/reference/src/scrubby.rs:935-951)
    const std::string preset = c->preset && c->preset[0] ? c->preset : (c->n_files == 2 ? "sr" : "map-ont");
    sh_opts opts;
    sh_status st = sh_preset(preset.c_str(), &opts);
    if (st != SH_OK) return st;

    const auto t0 = now();
    sh_index *idx = nullptr;
    {
        const std::string ip = c->index;
        st = ends_with(ip, ".shidx") ? sh_index_load(c->index, c->device, &idx) : sh_index_build_fasta(c->index, &opts, c->device, &idx);
        if (st != SH_OK) return st == SH_ERR_IO ? st : SH_ERR_INDEX;
        // a prebuilt index (.mmi, .shidx) brings its own k and w; they prevail over the preset's, as with minimap2
        { sh_index_info ii; if (sh_index_info_get(idx, &ii) == SH_OK) { opts.k = ii.k; opts.w = ii.w; } }
    }
    const auto t1 = now();

    // ---- pass 1: read -> parse -> classify, fold flagged ids ----
    std::vector<std::shared_ptr<Chunk>> kept[2];
    ShardedIdSet depleted;
    double classify_ms = 0;
    bool retained = false;
    {
        Pass1 p1{c, idx, opts, chunk_bytes, budget, threads, kept, &depleted};
        st = p1.run(true);
        if (st == SH_RETRY_SEQUENTIAL) {     // a chunk cut at a guessed boundary did not parse: let the sequential reader decide
            kept[0].clear(); kept[1].clear();
            depleted.clear();
            Pass1 p1s{c, idx, opts, chunk_bytes, budget, threads, kept, &depleted};
            st = p1s.run(false);
            classify_ms = p1s.classify_ms; retained = p1s.retain.load(); res->n_ext_unresolved = p1s.n_ext_unresolved; res->n_rmq_open = p1s.n_rmq_open;
        } else { classify_ms = p1.classify_ms; retained = p1.retain.load(); res->n_ext_unresolved = p1.n_ext_unresolved; res->n_rmq_open = p1.n_rmq_open; }
        if (res->n_ext_unresolved || res->n_rmq_open)
            fprintf(stderr, "[scrubby-hip] note: %llu read(s) kept their chain-level answer (mapped) and %llu long read(s) met a long-join tie beyond the exact path's size (DESIGN.md 1)\n",
                    (unsigned long long)res->n_ext_unresolved, (unsigned long long)res->n_rmq_open);
    }
    sh_index_free(idx);
    if (st != SH_OK) return st;
    if (!retained) { kept[0].clear(); kept[1].clear(); }
    res->n_depleted_ids = depleted.size();
    const auto t2 = now();

    // ---- pass 2: filter + write, the files side by side (clean_reads' par_iter over files, cleaner.rs:239) ----
    const bool want_dropped = c->read_ids && c->extract;      // deplete: the dropped ids ARE the depleted set
    FileFilter ff[2];
    sh_status fst[2] = {SH_OK, SH_OK};
    std::string ferr[2];
    std::vector<std::thread> filters;
    for (uint32_t i = 0; i < c->n_files; ++i) {
        ff[i] = FileFilter{c->input[i], c->output[i], retained ? &kept[i] : nullptr, chunk_bytes, &depleted, c->extract != 0, (bool)want_dropped,
                           std::max(1, threads / (int)c->n_files)};
        filters.emplace_back([&, i]() { fst[i] = ff[i].run(); if (fst[i] != SH_OK) ferr[i] = sh_last_error(); });
    }
    for (auto &t : filters) t.join();
    for (uint32_t i = 0; i < c->n_files; ++i)
        if (fst[i] != SH_OK) { sh_set_error("%s", ferr[i].c_str()); return fst[i]; }
    kept[0].clear(); kept[1].clear();
    const auto t3 = now();

    // ---- report (ScrubbyReport::create, report.rs:24-57; ReadDifference, utils.rs:250-285) ----
    uint64_t rin = 0, rout = 0;
    for (uint32_t i = 0; i < c->n_files; ++i) { rin += ff[i].n_in; rout += ff[i].n_out; }
    res->reads_in = rin; res->reads_out = rout;
    res->reads_removed = c->extract ? 0 : rin - rout;
    res->reads_extracted = c->extract ? rin - rout : 0;
    if (c->read_ids) {
        std::string body = "id\n";
        if (!c->extract) depleted.for_each([&](const char *s, uint32_t n) { body.append(s, n); body += '\n'; });
        else {
            IdSet uniq;
            for (uint32_t i = 0; i < c->n_files; ++i)
                for (auto &s : ff[i].dropped)
                    if (uniq.insert(s.data(), (uint32_t)s.size())) { body += s; body += '\n'; }
        }
        const std::string p = c->read_ids;
        SH_CHECK(write_id_table(c->read_ids, body), SH_ERR_IO, "cannot write %s", c->read_ids);
    }
    if (c->json) {
        ReportSettings rs;
        rs.aligner = "minimap2-rs"; rs.index = c->index; rs.preset_variant = shi_preset_variant(preset); rs.extract = c->extract != 0;
        st = shi_write_report_json(c->input, c->output, c->n_files, c->command, rs, res, c->json);
        if (st != SH_OK) return st;
    }
    res->ms_index = ms(t0, t1); res->ms_ingest = ms(t1, t2); res->ms_classify = classify_ms; res->ms_write = ms(t2, now());
    (void)t3;
    return SH_OK;
}

// ---- `scrubby reads -c kraken2`: Cleaner::run_kraken (cleaner.rs:288-330) with the chunked reader, the GPU classifier, and the
// parallel filter of pass 2.  The reference runs the external kraken2, which writes kraken.reads / kraken.report, then reads both
// files back (parse_classifier_output, :375-382) and filters (clean_reads).  Here: both inputs are parsed in place side by side
// (chunks retained), mates are interleaved into one batch (records 2i, 2i + 1), sh_k2_classify_batch classifies it, the two text
// files still land in the workdir (formatted by a pool, written at their offsets), the taxid set comes from the report exactly
// as the reference derives it (shi_taxids_from_report), and a read is selected iff the taxid printed on its line is in that set -
// decided from the results in memory instead of re-reading the 10^7 lines just written (get_taxid_reads_kraken compares the
// same decimal string).
namespace {

struct RecRef { const Chunk *ch; const Rec *r; };

// all chunks of a file, sequentially parsed, plus a flat record index
struct ParsedFile {
    std::vector<std::shared_ptr<Chunk>> chunks;
    std::vector<uint64_t> first;        // ordinal of each chunk's first record; first.back() = record count
    std::string error;
    sh_status read(const char *path, size_t chunk_bytes)
    {
        ChunkReader rd(path, chunk_bytes, false);
        if (!rd.ok()) { error = std::string("cannot open ") + path; return SH_ERR_IO; }
        first.assign(1, 0);
        for (;;) {
            auto c = std::make_shared<Chunk>();
            const int r = rd.next(*c);
            if (r < 0) { error = std::string(path) + ": " + rd.error; return SH_ERR_IO; }
            if (r == 0) break;
            first.push_back(first.back() + c->recs.size());
            chunks.push_back(std::move(c));
        }
        return SH_OK;
    }
    uint64_t n() const { return first.back(); }
    // cursor over records [lo, hi)
    template <class F> void for_range(uint64_t lo, uint64_t hi, F f) const
    {
        if (lo >= hi) return;
        size_t ci = (size_t)(std::upper_bound(first.begin(), first.end(), lo) - first.begin()) - 1;
        uint64_t o = lo;
        while (o < hi) {
            const Chunk &c = *chunks[ci];
            const uint64_t end = std::min<uint64_t>(hi, first[ci + 1]);
            for (uint64_t k = o; k < end; ++k) f(k, c, c.recs[(size_t)(k - first[ci])]);
            o = end; ++ci;
        }
    }
};

template <class F> void parallel_ranges(uint64_t n, int threads, F f)
{
    const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(1, threads), n / 65536 + 1));
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { f(t, n * (uint64_t)t / (uint64_t)T, n * (uint64_t)(t + 1) / (uint64_t)T); });
    for (auto &x : th) x.join();
}

}  // namespace

extern "C" sh_status sh_kraken_run(const sh_kraken_config *c, sh_reads_result *res)
{
    SH_CHECK(c && res, SH_ERR_BAD_ARG, "sh_kraken_run: null argument");
    SH_CHECK(c->n_files >= 1 && c->n_files <= 2, SH_ERR_BAD_ARG, "one or two input files are supported (got %u)", c->n_files);
    for (uint32_t i = 0; i < c->n_files; ++i) SH_CHECK(c->input[i] && c->output[i], SH_ERR_BAD_ARG, "input/output %u missing", i);
    SH_CHECK(c->db, SH_ERR_BAD_ARG, "MissingClassifierIndex");
    SH_CHECK(c->n_taxa + c->n_taxa_direct > 0, SH_ERR_BAD_ARG, "MissingTaxa: --taxa or --taxa-direct is required");
    if (const char *e = getenv("SCRUBBY_HIP_LEGACY_HOST")) if (*e == '1') return shi_kraken_run_legacy(c, res);
    for (uint32_t i = 0; i < c->n_files; ++i) {
        bool exists;
        if (shi_unsupported_compression(c->input[i])) return SH_ERR_IO;
        if (file_is_empty(c->input[i], exists)) return shi_kraken_run_legacy(c, res);       // App. C Q6 corner: the line-by-line form keeps its behaviour
    }
    memset(res, 0, sizeof(*res));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const bool paired = c->n_files == 2;
    const int threads = c->threads > 0 ? c->threads : shi_default_threads();
    const size_t chunk_bytes = env_mb("SCRUBBY_HIP_CHUNK_MB", 64ull << 20);

    const auto t0 = now();
    sh_k2_db *db = nullptr;
    sh_status st = sh_k2_open(c->db, c->device, &db);
    if (st != SH_OK) return st;
    struct DbGuard { sh_k2_db *d; ~DbGuard() { if (d) sh_k2_free(d); } } guard{db};
    sh_k2_opts opts;                 // k, l, masks and the down-sampling threshold come from the database
    sh_k2_db_opts(db, &opts);
    if (c->confidence >= 0.0) opts.confidence = c->confidence;           // -C "--confidence x"
    if (c->min_hit_groups > 0) opts.min_hit_groups = c->min_hit_groups;   // -C "--minimum-hit-groups n"
    const auto t1 = now();

    // ---- ingest: both files side by side, parsed in place ----
    ParsedFile pf[2];
    sh_status pst[2] = {SH_OK, SH_OK};
    {
        std::vector<std::thread> th;
        for (uint32_t i = 0; i < c->n_files; ++i) th.emplace_back([&, i] { pst[i] = pf[i].read(c->input[i], chunk_bytes); });
        for (auto &x : th) x.join();
    }
    for (uint32_t i = 0; i < c->n_files; ++i) if (pst[i] != SH_OK) { sh_set_error("%s", pf[i].error.c_str()); return pst[i]; }
    const uint64_t n_units = pf[0].n();
    if (paired && pf[1].n() != n_units) {
        sh_set_error("%s: %s", c->input[1], pf[1].n() < n_units ? "fewer records than mate 1" : "more records than mate 1");
        return SH_ERR_IO;
    }
    // mates interleaved: records 2i, 2i + 1 (kraken2 --paired); lengths first, then the bases in parallel
    const uint64_t n_rec = paired ? 2 * n_units : n_units;
    std::vector<uint64_t> offsets(n_rec + 1, 0);
    std::atomic<int> bad_id{0};
    for (uint32_t f = 0; f < c->n_files; ++f)
        parallel_ranges(n_units, threads, [&, f](int, uint64_t lo, uint64_t hi) {
            pf[f].for_range(lo, hi, [&](uint64_t k, const Chunk &ch, const Rec &r) {
                offsets[(paired ? 2 * k + f : k) + 1] = r.seq_len;
                const char *id; uint32_t il;
                if (f == 0 && !id_of(ch.data + r.hdr, r.hdr_len, &id, &il)) bad_id = 1;
            });
        });
    if (bad_id.load()) { sh_set_error("record without an id in %s", c->input[0]); return SH_ERR_IO; }
    for (uint64_t i = 0; i < n_rec; ++i) offsets[i + 1] += offsets[i];
    std::vector<uint8_t> bases(offsets[n_rec] + 64, (uint8_t)'N');
    for (uint32_t f = 0; f < c->n_files; ++f)
        parallel_ranges(n_units, threads, [&, f](int, uint64_t lo, uint64_t hi) {
            pf[f].for_range(lo, hi, [&](uint64_t k, const Chunk &ch, const Rec &r) { memcpy(bases.data() + offsets[paired ? 2 * k + f : k], ch.data + r.seq, r.seq_len); });
        });
    const auto t2 = now();

    std::vector<sh_k2_result> results(std::max<uint64_t>(n_units, 1));
    st = sh_k2_classify_batch(db, &opts, bases.data(), offsets.data(), n_rec, paired ? 1 : 0, results.data(), nullptr);
    if (st != SH_OK) return st;
    std::vector<uint8_t>().swap(bases);
    const auto t3 = now();

    // ---- kraken.reads and kraken.report in the workdir (cleaner.rs:291-297) ----
    std::string dir = c->workdir && c->workdir[0] ? c->workdir : (getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp");
    if (c->workdir && c->workdir[0]) shi_mkdir_p(dir);                       // create_dir_all (cleaner.rs:293)
    const std::string reads_path = dir + "/kraken.reads", report_path = dir + "/kraken.report";
    // the id Kraken 2 prints for a pair is mate 1's first token with a trailing /1 removed
    auto unit_id = [&](const Chunk &ch, const Rec &r, const char **id, uint32_t *il) {
        id_of(ch.data + r.hdr, r.hdr_len, id, il);
        if (paired && *il > 2 && (*id)[*il - 2] == '/' && (*id)[*il - 1] == '1') *il -= 2;
    };
    {
        const int fd = open(reads_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        SH_CHECK(fd >= 0, SH_ERR_IO, "cannot write %s", reads_path.c_str());
        const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)threads, n_units / 65536 + 1));
        std::vector<std::string> part((size_t)T);
        parallel_ranges(n_units, T, [&](int t, uint64_t lo, uint64_t hi) {
            std::string &o = part[(size_t)t];
            o.reserve((size_t)(hi - lo) * 64);
            char num[96];
            pf[0].for_range(lo, hi, [&](uint64_t k, const Chunk &ch, const Rec &r) {
                const sh_k2_result &x = results[k];
                const char *id; uint32_t il;
                unit_id(ch, r, &id, &il);
                o += x.call ? 'C' : 'U'; o += '\t'; o.append(id, il);
                int m;
                if (paired) m = snprintf(num, sizeof num, "\t%u\t%u|%u\tkmers=%u groups=%u\n", x.taxid, r.seq_len, (uint32_t)(offsets[2 * k + 2] - offsets[2 * k + 1]), x.total_kmers, x.hit_groups);
                else m = snprintf(num, sizeof num, "\t%u\t%u\tkmers=%u groups=%u\n", x.taxid, r.seq_len, x.total_kmers, x.hit_groups);
                o.append(num, (size_t)m);
            });
        });
        bool ok = true;
        uint64_t off = 0;
        std::vector<uint64_t> offs((size_t)T);
        for (int t = 0; t < T; ++t) { offs[(size_t)t] = off; off += part[(size_t)t].size(); }
        parallel_ranges((uint64_t)T, T, [&](int, uint64_t lo, uint64_t hi) {
            for (uint64_t t = lo; t < hi; ++t) {
                size_t done = 0;
                while (done < part[t].size()) { const ssize_t w = pwrite(fd, part[t].data() + done, part[t].size() - done, (off_t)(offs[t] + done)); if (w <= 0) break; done += (size_t)w; }
            }
        });
        struct stat sb;
        ok = fstat(fd, &sb) == 0 && (uint64_t)sb.st_size == off;
        ok = (close(fd) == 0) && ok;
        SH_CHECK(ok, SH_ERR_IO, "short write to %s", reads_path.c_str());
    }
    st = sh_k2_write_report(db, results.data(), n_units, report_path.c_str());
    if (st != SH_OK) return st;

    // ---- parse_classifier_output (cleaner.rs:375-382): taxids from the report, then the reads carrying one of them ----
    ReportSettings rs;
    for (uint32_t i = 0; i < c->n_taxa; ++i) rs.taxa.push_back(c->taxa[i]);
    for (uint32_t i = 0; i < c->n_taxa_direct; ++i) rs.taxa_direct.push_back(c->taxa_direct[i]);
    std::unordered_set<std::string> taxids;
    st = shi_taxids_from_report(report_path.c_str(), rs.taxa, rs.taxa_direct, taxids);
    if (st != SH_OK) return st;
    std::unordered_set<uint32_t> tax_num;          // a line's taxid column is printed with %u: only canonical decimal strings can equal it
    for (const std::string &t : taxids) {
        if (t.empty() || t.size() > 10 || (t.size() > 1 && t[0] == '0')) continue;
        bool dig = true; uint64_t v = 0;
        for (char ch : t) { if (ch < '0' || ch > '9') { dig = false; break; } v = v * 10 + (uint64_t)(ch - '0'); }
        if (dig && v <= 0xffffffffull) tax_num.insert((uint32_t)v);
    }
    ShardedIdSet dep;
    parallel_ranges(n_units, threads, [&](int, uint64_t lo, uint64_t hi) {
        pf[0].for_range(lo, hi, [&](uint64_t k, const Chunk &ch, const Rec &r) {
            if (!tax_num.count(results[k].taxid)) return;
            const char *id; uint32_t il;
            unit_id(ch, r, &id, &il);
            dep.insert(id, il);
        });
    });
    res->n_depleted_ids = dep.size();

    // ---- clean_reads over the retained chunks ----
    if (c->read_ids) {
        const std::string p = c->read_ids;
    }
    // the id table lists the input records MISSING from the outputs (ReadDifference, utils.rs:265-279), not the depletion set: a pair id
    // with its "/1" stripped may match no record at all
    const bool want_dropped = c->read_ids != nullptr;
    FileFilter ff[2];
    sh_status fst[2] = {SH_OK, SH_OK};
    std::string ferr[2];
    {
        std::vector<std::thread> filters;
        for (uint32_t i = 0; i < c->n_files; ++i) {
            ff[i] = FileFilter{c->input[i], c->output[i], &pf[i].chunks, chunk_bytes, &dep, c->extract != 0, (bool)want_dropped, std::max(1, threads / (int)c->n_files)};
            filters.emplace_back([&, i]() { fst[i] = ff[i].run(); if (fst[i] != SH_OK) ferr[i] = sh_last_error(); });
        }
        for (auto &t : filters) t.join();
    }
    for (uint32_t i = 0; i < c->n_files; ++i) if (fst[i] != SH_OK) { sh_set_error("%s", ferr[i].c_str()); return fst[i]; }
    const auto t4 = now();

    uint64_t rin = 0, rout = 0;
    for (uint32_t i = 0; i < c->n_files; ++i) { rin += ff[i].n_in; rout += ff[i].n_out; }
    res->reads_in = rin; res->reads_out = rout;
    res->reads_removed = c->extract ? 0 : rin - rout;
    res->reads_extracted = c->extract ? rin - rout : 0;
    if (c->read_ids) {
        std::string body = "id\n";
        IdSet uniq;
        for (uint32_t i = 0; i < c->n_files; ++i)
            for (auto &s : ff[i].dropped)
                if (uniq.insert(s.data(), (uint32_t)s.size())) { body += s; body += '\n'; }
        SH_CHECK(write_id_table(c->read_ids, body), SH_ERR_IO, "cannot write %s", c->read_ids);
    }
    if (c->json) {
        rs.classifier = "kraken2"; rs.classifier_args = c->classifier_args && c->classifier_args[0] ? c->classifier_args : nullptr; rs.index = c->db; rs.extract = c->extract != 0;
        st = shi_write_report_json(c->input, c->output, c->n_files, c->command, rs, res, c->json);
        if (st != SH_OK) return st;
    }
    res->ms_index = ms(t0, t1); res->ms_ingest = ms(t1, t2); res->ms_classify = ms(t2, t3); res->ms_write = ms(t3, t4);
    return SH_OK;
}
