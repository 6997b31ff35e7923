// scrubby-hip — command-line front end of the replaced path, with the flags of `scrubby reads`
// (/root/reference/src/terminal.rs:57-157).  Only what the mm2 aligner path uses is implemented; options that
// select other aligners / classifiers are accepted and rejected with the reason.
#include "../../include/scrubby_hip.h"
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static void usage()
{
    fprintf(stderr,
            "scrubby-hip reads -i <R1> [R2] -o <O1> [O2] -I <ref.fa[.gz]|index.shidx> [-p sr|map-ont|lr:hq|map-hifi]\n"
            "                  [-e] [-j report.json] [-r read_ids.tsv[.gz]] [-t threads] [-a minimap2-rs] [-w workdir]\n"
            "scrubby-hip reads -i <R1> [R2] -o <O1> [O2] -c kraken2 -I <kraken2 db dir> [-T taxa..] [-D taxa..] [-w workdir]\n"
            "                  [-C \"--confidence x --minimum-hit-groups n\"] [-e] [-j report.json] [-r read_ids.tsv[.gz]]\n"
            "scrubby-hip classifier -i <R1> [R2] -o <O1> [O2] -k <report> -j <reads> -c kraken2|metabuli [-T taxa..] [-D taxa..]\n"
            "                  [-e] [--json report.json] [-r read_ids.tsv]\n");
}

// `scrubby classifier` (/root/reference/src/terminal.rs:204-279): clean reads from precomputed Kraken2 / Metabuli outputs
static int main_classifier(int argc, char **argv, const std::string &command)
{
    std::vector<std::string> in, out, taxa, direct;
    std::string report, reads, classifier, json, ids;
    int extract = 0;
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        auto multi = [&](std::vector<std::string> &v) { while (i + 1 < argc && argv[i + 1][0] != '-') v.push_back(argv[++i]); };
        if (a == "-i" || a == "--input") multi(in);
        else if (a == "-o" || a == "--output") multi(out);
        else if (a == "-k" || a == "--report") report = val();
        else if (a == "-j" || a == "--reads") reads = val();          // the reference binds -j to --reads AND --json (App. C Q13): --json has no short here
        else if (a == "-c" || a == "--classifier") classifier = val();
        else if (a == "-T" || a == "--taxa") multi(taxa);
        else if (a == "-D" || a == "--taxa-direct") multi(direct);
        else if (a == "--json") json = val();
        else if (a == "-r" || a == "--read-ids") ids = val();
        else if (a == "-w" || a == "--workdir") val();
        else if (a == "-e" || a == "--extract") extract = 1;
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (in.empty() || in.size() > 2 || in.size() != out.size()) { fprintf(stderr, "error: one or two inputs and as many outputs are required\n"); return 2; }
    std::vector<const char *> tp, dp;
    for (auto &t : taxa) tp.push_back(t.c_str());
    for (auto &t : direct) dp.push_back(t.c_str());
    sh_classifier_config c{};
    for (size_t k = 0; k < in.size(); ++k) { c.input[k] = in[k].c_str(); c.output[k] = out[k].c_str(); }
    c.n_files = (uint32_t)in.size(); c.extract = extract;
    c.report = report.empty() ? nullptr : report.c_str(); c.reads = reads.empty() ? nullptr : reads.c_str();
    c.classifier = classifier.empty() ? nullptr : classifier.c_str();
    c.taxa = tp.data(); c.n_taxa = (uint32_t)tp.size(); c.taxa_direct = dp.data(); c.n_taxa_direct = (uint32_t)dp.size();
    c.json = json.empty() ? nullptr : json.c_str(); c.read_ids = ids.empty() ? nullptr : ids.c_str(); c.command = command.c_str();
    sh_reads_result r{};
    sh_status st = sh_classifier_run(&c, &r);
    if (st != SH_OK) { fprintf(stderr, "error (%d): %s\n", st, sh_last_error()); return 1; }
    fprintf(stderr, "[scrubby-hip] read ids selected by taxid: %llu\n", (unsigned long long)r.n_depleted_ids);
    return 0;
}

// `scrubby alignment` (/root/reference/src/terminal.rs:281-360)
static int main_alignment(int argc, char **argv, const std::string &command)
{
    std::vector<std::string> in, out;
    std::string aln, fmt, json, ids;
    unsigned long long min_len = 0; double min_cov = 0.0; unsigned min_mapq = 0;
    int extract = 0;
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        auto multi = [&](std::vector<std::string> &v) { while (i + 1 < argc && argv[i + 1][0] != '-') v.push_back(argv[++i]); };
        if (a == "-i" || a == "--input") multi(in);
        else if (a == "-o" || a == "--output") multi(out);
        else if (a == "-a" || a == "--alignment") aln = val();
        else if (a == "-f" || a == "--format") fmt = val();
        else if (a == "-l" || a == "--min-len") min_len = strtoull(val().c_str(), nullptr, 10);
        else if (a == "-c" || a == "--min-cov") min_cov = atof(val().c_str());
        else if (a == "-q" || a == "--min-mapq") min_mapq = (unsigned)atoi(val().c_str());
        else if (a == "-j" || a == "--json") json = val();
        else if (a == "-r" || a == "--read-ids") ids = val();
        else if (a == "-w" || a == "--workdir") val();
        else if (a == "-e" || a == "--extract") extract = 1;
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (in.empty() || in.size() > 2 || in.size() != out.size()) { fprintf(stderr, "error: one or two inputs and as many outputs are required\n"); return 2; }
    sh_alignment_config c{};
    for (size_t k = 0; k < in.size(); ++k) { c.input[k] = in[k].c_str(); c.output[k] = out[k].c_str(); }
    c.n_files = (uint32_t)in.size(); c.extract = extract; c.alignment = aln.empty() ? nullptr : aln.c_str();
    c.format = fmt.empty() ? nullptr : fmt.c_str(); c.min_len = min_len; c.min_cov = min_cov; c.min_mapq = min_mapq;
    c.json = json.empty() ? nullptr : json.c_str(); c.read_ids = ids.empty() ? nullptr : ids.c_str(); c.command = command.c_str();
    sh_reads_result r{};
    sh_status st = sh_alignment_run(&c, &r);
    if (st != SH_OK) { fprintf(stderr, "error (%d): %s\n", st, sh_last_error()); return 1; }
    fprintf(stderr, "[scrubby-hip] read ids selected by alignment: %llu\n", (unsigned long long)r.n_depleted_ids);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && (std::string(argv[1]) == "classifier" || std::string(argv[1]) == "alignment")) {
        std::string command;
        for (int i = 0; i < argc; ++i) { if (i) command += ' '; command += argv[i]; }
        return std::string(argv[1]) == "alignment" ? main_alignment(argc, argv, command) : main_classifier(argc, argv, command);
    }
    if (argc < 2 || std::string(argv[1]) != "reads") { usage(); return 2; }
    std::vector<std::string> in, out, taxa, taxa_direct;
    std::string index, preset, json, ids, aligner, classifier, workdir, cargs, command;
    int extract = 0, threads = 0;      // 0: the CPUs this process may use (the deflate stage of .gz outputs scales with them)
    for (int i = 0; i < argc; ++i) { if (i) command += ' '; command += argv[i]; }      // terminal.rs:178
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        auto multi = [&](std::vector<std::string> &v) { while (i + 1 < argc && argv[i + 1][0] != '-') v.push_back(argv[++i]); };
        if (a == "-i" || a == "--input") multi(in);
        else if (a == "-o" || a == "--output") multi(out);
        else if (a == "-I" || a == "--index") index = val();
        else if (a == "-p" || a == "--preset") preset = val();
        else if (a == "-a" || a == "--aligner") aligner = val();
        else if (a == "-c" || a == "--classifier") classifier = val();
        else if (a == "-T" || a == "--taxa") multi(taxa);
        else if (a == "-D" || a == "--taxa-direct") multi(taxa_direct);
        else if (a == "-C" || a == "--classifier-args") cargs = val();
        else if (a == "-j" || a == "--json") json = val();
        else if (a == "-r" || a == "--read-ids") ids = val();
        else if (a == "-t" || a == "--threads") threads = atoi(val().c_str());
        else if (a == "-w" || a == "--workdir") workdir = val();
        else if (a == "-l" || a == "--log-file" || a == "-A" || a == "--aligner-args") val();
        else if (a == "-e" || a == "--extract") extract = 1;
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); usage(); return 2; }
    }
    if (in.empty() || in.size() > 2 || in.size() != out.size()) { fprintf(stderr, "error: one or two inputs and as many outputs are required\n"); return 2; }
    if (!classifier.empty()) {       // Cleaner::run_classifier (cleaner.rs:159-163): kraken2 on the GPU
        if (classifier != "kraken2") { fprintf(stderr, "error: the HIP backend replaces --classifier kraken2 only (got %s)\n", classifier.c_str()); return 2; }
        if (!aligner.empty()) { fprintf(stderr, "error: AlignerAndClassifierConfigured\n"); return 2; }
        if (index.empty()) { fprintf(stderr, "error: MissingClassifierIndex (-I)\n"); return 2; }
        sh_kraken_config k{};
        std::vector<const char *> tp, dp;
        for (auto &x : taxa) tp.push_back(x.c_str());
        for (auto &x : taxa_direct) dp.push_back(x.c_str());
        for (size_t q = 0; q < in.size(); ++q) { k.input[q] = in[q].c_str(); k.output[q] = out[q].c_str(); }
        k.n_files = (uint32_t)in.size(); k.extract = extract; k.db = index.c_str(); k.workdir = workdir.empty() ? nullptr : workdir.c_str();
        k.taxa = tp.data(); k.n_taxa = (uint32_t)tp.size(); k.taxa_direct = dp.data(); k.n_taxa_direct = (uint32_t)dp.size();
        k.confidence = -1.0; k.min_hit_groups = 0;
        {   // the two Kraken 2 thresholds out of --classifier-args
            size_t p = cargs.find("--confidence");
            if (p != std::string::npos) k.confidence = atof(cargs.c_str() + p + 12);
            p = cargs.find("--minimum-hit-groups");
            if (p != std::string::npos) k.min_hit_groups = atoi(cargs.c_str() + p + 20);
        }
        k.json = json.empty() ? nullptr : json.c_str(); k.read_ids = ids.empty() ? nullptr : ids.c_str();
        k.command = command.c_str(); k.device = 0; k.threads = threads;
        k.classifier_args = cargs.empty() ? nullptr : cargs.c_str();
        sh_reads_result r{};
        sh_status st = sh_kraken_run(&k, &r);
        if (st != SH_OK) { fprintf(stderr, "error (%d): %s\n", st, sh_last_error()); return 1; }
        fprintf(stderr, "[scrubby-hip] depleted ids: %llu | database %.0f ms, ingest %.0f ms, classify %.0f ms, write %.0f ms\n",
                (unsigned long long)r.n_depleted_ids, r.ms_index, r.ms_ingest, r.ms_classify, r.ms_write);
        return 0;
    }
    if (aligner.empty()) aligner = "minimap2-rs";
    if (aligner != "minimap2-rs") { fprintf(stderr, "error: the HIP backend replaces --aligner minimap2-rs only (got %s)\n", aligner.c_str()); return 2; }
    if (index.empty()) { fprintf(stderr, "error: MissingAlignmentIndex (-I)\n"); return 2; }
    sh_reads_config c{};
    for (size_t k = 0; k < in.size(); ++k) { c.input[k] = in[k].c_str(); c.output[k] = out[k].c_str(); }
    c.n_files = (uint32_t)in.size(); c.extract = extract; c.index = index.c_str();
    c.preset = preset.empty() ? nullptr : preset.c_str();
    c.json = json.empty() ? nullptr : json.c_str(); c.read_ids = ids.empty() ? nullptr : ids.c_str();
    c.command = command.c_str(); c.threads = threads; c.device = 0;
    sh_reads_result r{};
    sh_status st = sh_reads_run(&c, &r);
    if (st != SH_OK) { fprintf(stderr, "error (%d): %s\n", st, sh_last_error()); return 1; }
    fprintf(stderr, "[scrubby-hip] depleted ids: %llu | index %.0f ms, ingest %.0f ms, classify %.0f ms, write %.0f ms\n",
            (unsigned long long)r.n_depleted_ids, r.ms_index, r.ms_ingest, r.ms_classify, r.ms_write);
    return 0;
}
