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
            "scrubby-hip reads -i <R1> [R2] -o <O1> [O2] -I <ref.fa[.gz]|index.shidx> [-p sr|map-ont|lr:hq]\n"
            "                  [-e] [-j report.json] [-r read_ids.tsv[.gz]] [-t threads] [-a minimap2-rs] [-w workdir]\n");
}

int main(int argc, char **argv)
{
    if (argc < 2 || std::string(argv[1]) != "reads") { usage(); return 2; }
    std::vector<std::string> in, out;
    std::string index, preset, json, ids, aligner = "minimap2-rs", command;
    int extract = 0, threads = 4;
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
        else if (a == "-j" || a == "--json") json = val();
        else if (a == "-r" || a == "--read-ids") ids = val();
        else if (a == "-t" || a == "--threads") threads = atoi(val().c_str());
        else if (a == "-w" || a == "--workdir" || a == "-l" || a == "--log-file" || a == "-A" || a == "--aligner-args") val();
        else if (a == "-e" || a == "--extract") extract = 1;
        else if (a == "-c" || a == "--classifier" || a == "-T" || a == "--taxa" || a == "-D" || a == "--taxa-direct" || a == "-C" || a == "--classifier-args") {
            fprintf(stderr, "error: classifier paths are not part of the HIP backend yet (%s)\n", a.c_str()); return 2;
        } else { fprintf(stderr, "unknown argument %s\n", a.c_str()); usage(); return 2; }
    }
    if (aligner != "minimap2-rs") { fprintf(stderr, "error: the HIP backend replaces --aligner minimap2-rs only (got %s)\n", aligner.c_str()); return 2; }
    if (in.empty() || in.size() > 2 || in.size() != out.size()) { fprintf(stderr, "error: one or two inputs and as many outputs are required\n"); return 2; }
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
