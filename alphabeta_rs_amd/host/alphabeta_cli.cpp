// `alphabeta` command-line tool: same flags, console output and output files as the reference binary
// (src/cli/alphabeta.rs:8-38, src/arguments.rs:93-152), running the ABneutral path on an MI355X through
// libabneutral_hip.so.  Extra flags (not in the reference): --seed, --device(s), --lanes, --strict-order, and
// --pedigree FILE --p0uu X to start from an existing pedigree file instead of nodelist/edgelist.
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>

#include "alphabeta.hpp"

namespace fs = std::filesystem;
using namespace alphabeta;

static void usage() {
  std::puts(
      "Usage: alphabeta [OPTIONS]\n\n"
      "Options:\n"
      "  -i, --iterations <ITERATIONS>  Number of iterations to run for Nelder-Mead optimization, even 100 is enough [default: 1000]\n"
      "  -e, --edges <EDGES>            Relative or absolute path to an edgelist, see /data for an example [default: ./edgelist.txt]\n"
      "  -n, --nodes <NODES>            Relative or absolute path to a nodelist, see /data for an example [default: ./nodelist.txt]\n"
      "  -p, --posterior-max-filter <P> Minimum posterior probability for a singe basepair read to be included in the estimation [default: 0.99]\n"
      "  -o, --output <OUTPUT>          Relative or absolute path to an output directory, must exist, EXISTING FILES WILL BE OVERWRITTEN [default: .]\n"
      "      --seed <SEED>              Philox seed of start simplices, jitter and bootstrap indices [default: 20260101]\n"
      "      --device <N>               HIP device ordinal [default: 0]\n"
      "      --devices <A,B,..>         several HIP devices: the bootstraps are sharded over them, tables gathered with RCCL\n"
      "      --lanes <G>                lanes of a wavefront per Nelder-Mead chain: 0 (auto), 8, 16, 32, 64\n"
      "      --strict-order             sum every cost's residuals serially in row order, exactly as the reference does\n"
      "                                 (src/structs.rs:206-213): bit-equal to a reference-order CPU run, ~1.5x the time\n"
      "      --pedigree <FILE>          use this pedigree file (src/pedigree.rs:62-79 format) instead of building one\n"
      "      --p0uu <X>                 proportion of unmethylated sites at G0 (required with --pedigree)\n"
      "  -h, --help                     Print help\n"
      "  -V, --version                  Print version");
}

int main(int argc, char** argv) {
  Args args;
  bool edges_given = false, nodes_given = false;
  std::string ped_file;
  double p0uu_given = -1.0;
  uint64_t seed = 20260101ull;
  int device = 0, lanes = 0;
  bool strict_order = false;
  std::string devices_arg;
  auto need = [&](int& i) -> const char* {
    if (i + 1 >= argc) {
      std::fprintf(stderr, "error: a value is required for '%s' but none was supplied\n", argv[i]);
      std::exit(2);
    }
    return argv[++i];
  };
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i], v;
    auto eq = a.find('=');
    bool has_eq = a.rfind("--", 0) == 0 && eq != std::string::npos;
    if (has_eq) {
      v = a.substr(eq + 1);
      a = a.substr(0, eq);
    }
    auto val = [&]() -> std::string { return has_eq ? v : std::string(need(i)); };
    if (a == "-i" || a == "--iterations") args.iterations = (size_t)std::strtoull(val().c_str(), nullptr, 10);
    else if (a == "-e" || a == "--edges") { args.edges = val(); edges_given = true; }
    else if (a == "-n" || a == "--nodes") { args.nodes = val(); nodes_given = true; }
    else if (a == "-p" || a == "--posterior-max-filter") args.posterior_max_filter = std::strtod(val().c_str(), nullptr);
    else if (a == "-o" || a == "--output") args.output = val();
    else if (a == "--seed") seed = std::strtoull(val().c_str(), nullptr, 10);
    else if (a == "--device") device = std::atoi(val().c_str());
    else if (a == "--devices") devices_arg = val();
    else if (a == "--lanes") lanes = std::atoi(val().c_str());
    else if (a == "--strict-order") strict_order = true;
    else if (a == "--pedigree") ped_file = val();
    else if (a == "--p0uu") p0uu_given = std::strtod(val().c_str(), nullptr);
    else if (a == "-h" || a == "--help") { usage(); return 0; }
    else if (a == "-V" || a == "--version") { std::puts("alphabeta 0.2.1 (MI355X ABneutral path)"); return 0; }
    else {
      std::fprintf(stderr, "error: unexpected argument '%s' found\n\nFor more information, try '--help'.\n", argv[i]);
      return 2;
    }
  }
  (void)edges_given;
  (void)nodes_given;
  // value parsers of src/arguments.rs:116-140
  if (ped_file.empty()) {
    for (const std::string* f : {&args.edges, &args.nodes}) {
      if (fs::exists(*f)) std::printf("Using default file: %s\n", f->c_str());
      else {
        std::fprintf(stderr, "error: Please provide a valid file path. By default, we fill try %s, which does not exist.\n", f->c_str());
        return 2;
      }
    }
  }
  if (fs::exists(args.output)) std::printf("Using default output directory: %s\n", fs::canonical(args.output).c_str());
  else {
    std::fprintf(stderr, "error: Please provide a valid output directory. By default, we fill try %s, which does not exist.\n", args.output.c_str());
    return 2;
  }
  if (args.iterations == 0) {
    std::fprintf(stderr, "error: --iterations must be positive\n");
    return 2;
  }
  try {
    if (!devices_arg.empty()) {
      device_list() = parse_device_list(devices_arg);
      device = device_list()[0];
    }
    Device& dev = default_device(device);
    if (device_list().empty()) device_list().push_back(device);
    dev.options.seed = seed;
    dev.options.lanes_per_chain = lanes;
    dev.options.strict_order = strict_order ? 1 : 0;
    RunResult r;
    if (!ped_file.empty()) {
      if (!(p0uu_given > 0.0 && p0uu_given < 1.0)) {
        std::fprintf(stderr, "error: --pedigree needs --p0uu in (0,1)\n");
        return 2;
      }
      r = run_on_pedigree(Pedigree::from_file(ped_file), p0uu_given, args.iterations, args.output);
    } else {
      r = run(args);
    }
    // src/cli/alphabeta.rs:19-35
    std::printf("##########\nResults:\n\n%s\n%s\n", r.model.display().c_str(), r.analysis.display().c_str());
    std::printf("Estimated steady state %s\n", fmt_f64(steady_state(r.model.alpha, r.model.beta)).c_str());
    std::printf("Observed steady state methylation %s\n##########\n", fmt_f64(r.obs_steady_state).c_str());
    const fs::path out(args.output);
    r.pedigree.to_file((out / "pedigree.txt").string());
    r.analysis.to_file((out / "analysis.txt").string());
    r.raw_analysis.write_npy((out / "raw.npy").string());
  } catch (const Error& e) {
    std::printf("Error: %s\n", e.what());  // src/cli/alphabeta.rs:17 prints and exits 0; a status is more useful
    return 1;
  }
  return 0;
}
