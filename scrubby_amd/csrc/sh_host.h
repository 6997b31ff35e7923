// sh_host.h — declarations shared by the two host-side translation units (sh_host.cpp, sh_stream.cpp).
#pragma once
#include "sh_common.h"
#include <unordered_set>

struct ReportSettings {       // ScrubbySettings, /root/reference/src/report.rs:71-88 (field order = key order)
    const char *aligner = nullptr, *classifier = nullptr, *index = nullptr, *alignment = nullptr, *reads = nullptr, *report = nullptr, *preset_variant = nullptr;
    std::vector<std::string> taxa, taxa_direct;
    uint64_t min_len = 0; double min_cov = 0.0; uint32_t min_mapq = 0;
    bool extract = false;
    const char *classifier_args = nullptr;
};

// ScrubbyReport JSON (report.rs:10-88)
sh_status shi_write_report_json(const char *const *input, const char *const *output, uint32_t n_files, const char *command,
                                const ReportSettings &st, const sh_reads_result *r, const char *path);
// bzip2 / xz inputs (the reference reads them through niffler, utils.rs:377-383): recognised by their magic bytes and refused by name -
// this image ships libbz2 / liblzma without headers, so the backend links zlib only.  true = refused, error set.
bool shi_unsupported_compression(const char *path);
// -t <= 0: the CPUs this process may use (hardware threads, capped by the cgroup CPU quota and by 64)
int shi_default_threads();
// Preset's serde name ("Sr", "MapOnt", ...) from its Display form
const char *shi_preset_variant(const std::string &display);
// collect-then-map form of the whole path (sh_host.cpp)
sh_status shi_reads_run_legacy(const sh_reads_config *c, sh_reads_result *res);
// `scrubby reads -c kraken2`, collect-then-classify form (sh_host.cpp)
sh_status shi_kraken_run_legacy(const sh_kraken_config *c, sh_reads_result *res);
// get_taxids_from_report (/root/reference/src/classifier.rs:124-252)
sh_status shi_taxids_from_report(const char *report, const std::vector<std::string> &taxa_in, const std::vector<std::string> &direct_in,
                                 std::unordered_set<std::string> &out);
void shi_mkdir_p(const std::string &dir);
