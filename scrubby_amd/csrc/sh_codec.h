// sh_codec.h — byte sources and sinks by container, the set niffler gives the reference: plain, gzip, bzip2, xz.
//
// The reference opens its inputs through needletail / niffler, which sniff the container by its magic bytes
// (/root/reference/src/utils.rs:377-383 parse_fastx_file_with_check; gzip 1f 8b, bzip2 "BZh", xz fd 37 7a 58 5a 00), and picks the
// container of an OUTPUT by the file's extension (CompressionExt::from_path, utils.rs:28-36: .gz -> gzip, .bz / .bz2 -> bzip2,
// .lzma / .xz -> niffler's "Lzma", which is the xz container of the xz2 crate; get_fastx_writer :56-74).
// zlib is linked; bzip2 and xz come from the system's libbz2.so.1.0 / liblzma.so.5, loaded on first use (this image ships the
// libraries without their headers, so the few entry points and the two stream structs are declared here as their ABI has them).
// A host without one of the libraries gets an error that names it when such a file is met - nothing else depends on them.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <dlfcn.h>
#include <zlib.h>

namespace shc {

// ---- libbz2 ---------------------------------------------------------------------------------------------------------------------
struct bz_stream_t {
    char *next_in; unsigned int avail_in, total_in_lo32, total_in_hi32;
    char *next_out; unsigned int avail_out, total_out_lo32, total_out_hi32;
    void *state; void *(*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void *opaque;
};
enum { BZ_RUN_ = 0, BZ_FINISH_ = 2, BZ_OK_ = 0, BZ_RUN_OK_ = 1, BZ_FINISH_OK_ = 3, BZ_STREAM_END_ = 4 };
struct Bz2Api {
    int (*DecompressInit)(bz_stream_t *, int, int) = nullptr; int (*Decompress)(bz_stream_t *) = nullptr; int (*DecompressEnd)(bz_stream_t *) = nullptr;
    int (*CompressInit)(bz_stream_t *, int, int, int) = nullptr; int (*Compress)(bz_stream_t *, int) = nullptr; int (*CompressEnd)(bz_stream_t *) = nullptr;
    bool ok = false;
    static const Bz2Api &get()
    {
        static Bz2Api a = [] {
            Bz2Api x;
            void *h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) return x;
            x.DecompressInit = (int (*)(bz_stream_t *, int, int))dlsym(h, "BZ2_bzDecompressInit");
            x.Decompress = (int (*)(bz_stream_t *))dlsym(h, "BZ2_bzDecompress");
            x.DecompressEnd = (int (*)(bz_stream_t *))dlsym(h, "BZ2_bzDecompressEnd");
            x.CompressInit = (int (*)(bz_stream_t *, int, int, int))dlsym(h, "BZ2_bzCompressInit");
            x.Compress = (int (*)(bz_stream_t *, int))dlsym(h, "BZ2_bzCompress");
            x.CompressEnd = (int (*)(bz_stream_t *))dlsym(h, "BZ2_bzCompressEnd");
            x.ok = x.DecompressInit && x.Decompress && x.DecompressEnd && x.CompressInit && x.Compress && x.CompressEnd;
            return x;
        }();
        return a;
    }
};

// ---- liblzma --------------------------------------------------------------------------------------------------------------------
struct lzma_stream_t {
    const uint8_t *next_in; size_t avail_in; uint64_t total_in;
    uint8_t *next_out; size_t avail_out; uint64_t total_out;
    const void *allocator; void *internal;
    void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
    uint64_t reserved_int1, reserved_int2; size_t reserved_int3, reserved_int4;
    int reserved_enum1, reserved_enum2;
};
enum { LZMA_RUN_ = 0, LZMA_FINISH_ = 3, LZMA_OK_ = 0, LZMA_STREAM_END_ = 1, LZMA_CHECK_CRC64_ = 4, LZMA_CONCATENATED_ = 0x08 };
struct LzmaApi {
    int (*stream_decoder)(lzma_stream_t *, uint64_t, uint32_t) = nullptr; int (*easy_encoder)(lzma_stream_t *, uint32_t, int) = nullptr;
    int (*code)(lzma_stream_t *, int) = nullptr; void (*end)(lzma_stream_t *) = nullptr;
    bool ok = false;
    static const LzmaApi &get()
    {
        static LzmaApi a = [] {
            LzmaApi x;
            void *h = dlopen("liblzma.so.5", RTLD_NOW | RTLD_LOCAL);
            if (!h) return x;
            x.stream_decoder = (int (*)(lzma_stream_t *, uint64_t, uint32_t))dlsym(h, "lzma_stream_decoder");
            x.easy_encoder = (int (*)(lzma_stream_t *, uint32_t, int))dlsym(h, "lzma_easy_encoder");
            x.code = (int (*)(lzma_stream_t *, int))dlsym(h, "lzma_code");
            x.end = (void (*)(lzma_stream_t *))dlsym(h, "lzma_end");
            x.ok = x.stream_decoder && x.easy_encoder && x.code && x.end;
            return x;
        }();
        return a;
    }
};

enum class Kind { Plain, Gzip, Bzip2, Xz };

inline Kind kind_by_extension(const std::string &p)
{   // CompressionExt::from_path (utils.rs:28-36): the LAST extension decides
    auto ends = [&](const char *s) { const size_t k = strlen(s); return p.size() >= k && p.compare(p.size() - k, k, s) == 0; };
    if (ends(".gz")) return Kind::Gzip;
    if (ends(".bz") || ends(".bz2")) return Kind::Bzip2;
    if (ends(".lzma") || ends(".xz")) return Kind::Xz;
    return Kind::Plain;
}

// ---- a sequential, decompressing reader (what gzread is for plain and gzip files) -----------------------------------------------------
class In {
    Kind kind_ = Kind::Plain;
    gzFile gz_ = nullptr;
    FILE *fp_ = nullptr;
    bz_stream_t bz_{}; lzma_stream_t xz_{};
    bool codec_open_ = false, src_eof_ = false, end_ = false;
    std::vector<uint8_t> in_;
public:
    std::string error;
    In() = default;
    In(const In &) = delete;
    ~In() { close(); }
    bool open(const char *path)
    {
        close();
        error.clear();
        FILE *f = fopen(path, "rb");
        if (!f) { error = std::string("cannot open ") + path; return false; }
        unsigned char m[6] = {0, 0, 0, 0, 0, 0};
        const size_t n = fread(m, 1, 6, f);
        if (n >= 3 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h') kind_ = Kind::Bzip2;
        else if (n >= 6 && m[0] == 0xFD && m[1] == '7' && m[2] == 'z' && m[3] == 'X' && m[4] == 'Z' && m[5] == 0x00) kind_ = Kind::Xz;
        else kind_ = Kind::Plain;      // (gzip and plain alike: gzread passes plain bytes through)
        if (kind_ == Kind::Plain) {
            fclose(f);
            gz_ = gzopen(path, "rb");
            if (!gz_) { error = std::string("cannot open ") + path; return false; }
            gzbuffer(gz_, 1 << 20);
            return true;
        }
        rewind(f);
        fp_ = f;
        in_.resize(1 << 20);
        if (kind_ == Kind::Bzip2) {
            const Bz2Api &A = Bz2Api::get();
            if (!A.ok) { error = std::string("bzip2-compressed input, but libbz2.so.1.0 could not be loaded: ") + path; close(); return false; }
            memset(&bz_, 0, sizeof(bz_));
            if (A.DecompressInit(&bz_, 0, 0) != BZ_OK_) { error = "BZ2_bzDecompressInit failed"; close(); return false; }
        } else {
            const LzmaApi &A = LzmaApi::get();
            if (!A.ok) { error = std::string("xz-compressed input, but liblzma.so.5 could not be loaded: ") + path; close(); return false; }
            memset(&xz_, 0, sizeof(xz_));
            if (A.stream_decoder(&xz_, UINT64_MAX, LZMA_CONCATENATED_) != LZMA_OK_) { error = "lzma_stream_decoder failed"; close(); return false; }
        }
        codec_open_ = true;
        return true;
    }
    bool is_open() const { return gz_ != nullptr || fp_ != nullptr; }
    // up to n bytes; 0 = end of the stream; -1 = error (message in `error`: a truncated or corrupt stream is an error, never a short input)
    long read(void *buf, size_t n)
    {
        if (n == 0) return 0;
        if (gz_) {
            size_t done = 0;
            while (done < n) {
                const unsigned want = (unsigned)std::min<size_t>(n - done, 1u << 30);
                const int got = gzread(gz_, (char *)buf + done, want);
                if (got < 0) { int e; error = std::string("read error: ") + gzerror(gz_, &e); return -1; }
                done += (size_t)got;
                if ((unsigned)got < want) {
                    int e = Z_OK;
                    const char *msg = gzerror(gz_, &e);      // a truncated .gz ends with Z_BUF_ERROR, not with a clean end of stream
                    if (e != Z_OK && e != Z_STREAM_END) { error = std::string("read error: ") + (msg && *msg ? msg : "truncated gzip stream"); return -1; }
                    break;
                }
            }
            return (long)done;
        }
        if (!fp_ || end_) return 0;
        size_t done = 0;
        while (done < n && !end_) {
            const bool bz = kind_ == Kind::Bzip2;
            const size_t avail_in = bz ? bz_.avail_in : xz_.avail_in;
            if (avail_in == 0 && !src_eof_) {
                const size_t got = fread(in_.data(), 1, in_.size(), fp_);
                if (got < in_.size()) { if (ferror(fp_)) { error = "read error"; return -1; } src_eof_ = true; }
                if (bz) { bz_.next_in = (char *)in_.data(); bz_.avail_in = (unsigned)got; } else { xz_.next_in = in_.data(); xz_.avail_in = got; }
            }
            if (bz) {
                const Bz2Api &A = Bz2Api::get();
                bz_.next_out = (char *)buf + done; bz_.avail_out = (unsigned)std::min<size_t>(n - done, 1u << 30);
                const unsigned before = bz_.avail_out;
                const int rc = A.Decompress(&bz_);
                done += before - bz_.avail_out;
                if (rc == BZ_STREAM_END_) {
                    // concatenated streams (bzip2 -c a b, pbzip2): go on with the next one if input is left
                    if (bz_.avail_in == 0 && src_eof_) { end_ = true; break; }
                    if (bz_.avail_in == 0) { const size_t got = fread(in_.data(), 1, in_.size(), fp_); if (got < in_.size()) src_eof_ = true; bz_.next_in = (char *)in_.data(); bz_.avail_in = (unsigned)got; if (got == 0) { end_ = true; break; } }
                    std::vector<uint8_t> rest(bz_.next_in, bz_.next_in + bz_.avail_in);
                    A.DecompressEnd(&bz_);
                    memset(&bz_, 0, sizeof(bz_));
                    if (A.DecompressInit(&bz_, 0, 0) != BZ_OK_) { error = "BZ2_bzDecompressInit failed"; return -1; }
                    memcpy(in_.data(), rest.data(), rest.size());
                    bz_.next_in = (char *)in_.data(); bz_.avail_in = (unsigned)rest.size();
                } else if (rc != BZ_OK_) { error = "read error: corrupt bzip2 stream"; return -1; }
                else if (before == bz_.avail_out && bz_.avail_in == 0 && src_eof_) { error = "read error: truncated bzip2 stream"; return -1; }
            } else {
                const LzmaApi &A = LzmaApi::get();
                xz_.next_out = (uint8_t *)buf + done; xz_.avail_out = n - done;
                const size_t before = xz_.avail_out;
                const int rc = A.code(&xz_, src_eof_ && xz_.avail_in == 0 ? LZMA_FINISH_ : LZMA_RUN_);
                done += before - xz_.avail_out;
                if (rc == LZMA_STREAM_END_) { end_ = true; break; }
                if (rc != LZMA_OK_) { error = rc == 10 /* LZMA_BUF_ERROR */ ? "read error: truncated xz stream" : "read error: corrupt xz stream"; return -1; }
            }
        }
        return (long)done;
    }
    void close()
    {
        if (gz_) { gzclose(gz_); gz_ = nullptr; }
        if (codec_open_) { if (kind_ == Kind::Bzip2) Bz2Api::get().DecompressEnd(&bz_); else if (kind_ == Kind::Xz) LzmaApi::get().end(&xz_); codec_open_ = false; }
        if (fp_) { fclose(fp_); fp_ = nullptr; }
        src_eof_ = end_ = false;
    }
};

// ---- a sequential, compressing writer; the container by the path's extension, like the reference's get_fastx_writer ---------------------------
class Out {
    Kind kind_ = Kind::Plain;
    gzFile gz_ = nullptr;
    FILE *fp_ = nullptr;
    bz_stream_t bz_{}; lzma_stream_t xz_{};
    bool codec_open_ = false;
    std::vector<uint8_t> out_;
    bool pump(int action)      // run the encoder until it has taken all input (and, on finish, until the stream has ended)
    {
        for (;;) {
            int rc; size_t produced;
            if (kind_ == Kind::Bzip2) {
                bz_.next_out = (char *)out_.data(); bz_.avail_out = (unsigned)out_.size();
                rc = Bz2Api::get().Compress(&bz_, action ? BZ_FINISH_ : BZ_RUN_);
                produced = out_.size() - bz_.avail_out;
                if (rc < 0) { error = "bzip2 compression failed"; return false; }
                if (produced && fwrite(out_.data(), 1, produced, fp_) != produced) { error = "write error"; return false; }
                if (action ? rc == BZ_STREAM_END_ : bz_.avail_in == 0) return true;
            } else {
                xz_.next_out = out_.data(); xz_.avail_out = out_.size();
                rc = LzmaApi::get().code(&xz_, action ? LZMA_FINISH_ : LZMA_RUN_);
                produced = out_.size() - xz_.avail_out;
                if (rc != LZMA_OK_ && rc != LZMA_STREAM_END_) { error = "xz compression failed"; return false; }
                if (produced && fwrite(out_.data(), 1, produced, fp_) != produced) { error = "write error"; return false; }
                if (action ? rc == LZMA_STREAM_END_ : xz_.avail_in == 0) return true;
            }
        }
    }
public:
    std::string error;
    Out() = default;
    Out(const Out &) = delete;
    ~Out() { close(); }
    // level: niffler's compression level (the reference writes with Level::Six; the id table with Nine)
    bool open(const std::string &path, int level = 6)
    {
        kind_ = kind_by_extension(path);
        error.clear();
        if (kind_ == Kind::Gzip) {
            const std::string mode = "wb" + std::to_string(level);
            gz_ = gzopen(path.c_str(), mode.c_str());
            if (!gz_) error = "cannot open " + path;
            return gz_ != nullptr;
        }
        fp_ = fopen(path.c_str(), "wb");
        if (!fp_) { error = "cannot open " + path; return false; }
        if (kind_ == Kind::Plain) return true;
        out_.resize(1 << 20);
        if (kind_ == Kind::Bzip2) {
            const Bz2Api &A = Bz2Api::get();
            if (!A.ok) { error = "bzip2 output asked for, but libbz2.so.1.0 could not be loaded: " + path; return false; }
            memset(&bz_, 0, sizeof(bz_));
            if (A.CompressInit(&bz_, level < 1 ? 1 : (level > 9 ? 9 : level), 0, 0) != BZ_OK_) { error = "BZ2_bzCompressInit failed"; return false; }
        } else {
            const LzmaApi &A = LzmaApi::get();
            if (!A.ok) { error = "xz output asked for, but liblzma.so.5 could not be loaded: " + path; return false; }
            memset(&xz_, 0, sizeof(xz_));
            if (A.easy_encoder(&xz_, (uint32_t)(level < 0 ? 0 : (level > 9 ? 9 : level)), LZMA_CHECK_CRC64_) != LZMA_OK_) { error = "lzma_easy_encoder failed"; return false; }
        }
        codec_open_ = true;
        return true;
    }
    bool ok() const { return error.empty() && (gz_ || fp_); }
    bool write(const void *data, size_t n)
    {
        if (!error.empty()) return false;
        if (n == 0) return true;
        if (gz_) {
            for (size_t o = 0; o < n;) { const unsigned w = (unsigned)std::min<size_t>(n - o, 1u << 30); if (gzwrite(gz_, (const char *)data + o, w) != (int)w) { error = "write error"; return false; } o += w; }
            return true;
        }
        if (kind_ == Kind::Plain) { if (fwrite(data, 1, n, fp_) != n) { error = "write error"; return false; } return true; }
        for (size_t o = 0; o < n;) {
            const size_t w = std::min<size_t>(n - o, 1u << 30);
            if (kind_ == Kind::Bzip2) { bz_.next_in = (char *)data + o; bz_.avail_in = (unsigned)w; } else { xz_.next_in = (const uint8_t *)data + o; xz_.avail_in = w; }
            if (!pump(0)) return false;
            o += w;
        }
        return true;
    }
    bool close()
    {
        bool good = error.empty();
        if (gz_) { if (gzclose(gz_) != Z_OK) { error = "write error"; good = false; } gz_ = nullptr; }
        if (codec_open_) {
            if (good) { if (kind_ == Kind::Bzip2) { bz_.next_in = nullptr; bz_.avail_in = 0; } else { xz_.next_in = nullptr; xz_.avail_in = 0; } good = pump(1); }
            if (kind_ == Kind::Bzip2) Bz2Api::get().CompressEnd(&bz_); else LzmaApi::get().end(&xz_);
            codec_open_ = false;
        }
        if (fp_) { if (fclose(fp_) != 0) { error = "write error"; good = false; } fp_ = nullptr; }
        return good;
    }
};

}  // namespace shc
