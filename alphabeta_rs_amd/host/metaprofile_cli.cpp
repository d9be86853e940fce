// `metaprofile_alphabeta`: the AlphaBeta stage of the reference's `metaprofile` binary
// (src/cli/metaprofile.rs:33-114) on window directories that already exist
// (<output-dir>/{upstream,gene,downstream}/<window>/{nodelist,edgelist}.txt, as written by src/setup.rs:35-72).
// All windows are fitted by one batched plan on the MI355X.  Flags follow src/arguments.rs:6-62 where they apply.
#include <cstdlib>
#include <cstring>
#include <filesystem>

#include "metaprofile.hpp"

using namespace alphabeta;

int main(int argc, char** argv) {
  metaprofile::WindowArgs a;
  uint32_t max_gene_length = 0;
  std::string dist_file;
  uint64_t seed = 20260101ull;
  int device = 0;
  std::string devices_arg;
  for (int i = 1; i < argc; ++i) {
    const std::string f = argv[i];
    auto val = [&]() -> std::string {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "error: a value is required for '%s'\n", argv[i]);
        std::exit(2);
      }
      return argv[++i];
    };
    if (f == "-o" || f == "--output-dir") a.output_dir = val();
    else if (f == "--name") a.name = val();
    else if (f == "-s" || f == "--window-step") a.window_step = (uint32_t)std::strtoul(val().c_str(), nullptr, 10);
    else if (f == "-w" || f == "--window-size") a.window_size = (uint32_t)std::strtoul(val().c_str(), nullptr, 10);
    else if (f == "-c" || f == "--cutoff") a.cutoff = (uint32_t)std::strtoul(val().c_str(), nullptr, 10);
    else if (f == "-a" || f == "--absolute") a.absolute = true;
    else if (f == "--iterations") a.iterations = (size_t)std::strtoull(val().c_str(), nullptr, 10);
    else if (f == "--max-gene-length") max_gene_length = (uint32_t)std::strtoul(val().c_str(), nullptr, 10);
    else if (f == "--distribution") dist_file = val();
    else if (f == "--seed") seed = std::strtoull(val().c_str(), nullptr, 10);
    else if (f == "--device") device = std::atoi(val().c_str());
    else if (f == "--devices") devices_arg = val();
    else if (f == "-h" || f == "--help") {
      std::puts("Usage: metaprofile_alphabeta -o <output-dir> [--name N] [-s step] [-w size] [-c cutoff] [-a]\n"
                "       [--iterations 100] [--max-gene-length L] [--distribution FILE] [--seed S] [--device D]\n"
                "       [--devices A,B,..]   windows sharded over several HIP devices, tables gathered with RCCL");
      return 0;
    } else {
      std::fprintf(stderr, "error: unexpected argument '%s' found\n", argv[i]);
      return 2;
    }
  }
  if (!std::filesystem::exists(a.output_dir)) {
    std::fprintf(stderr, "error: output directory %s does not exist\n", a.output_dir.c_str());
    return 2;
  }
  std::printf("Starting run %s\n", a.name.c_str());
  try {
    if (!devices_arg.empty()) {
      device_list() = parse_device_list(devices_arg);
      device = device_list()[0];
    }
    Device& dev = default_device(device);
    if (device_list().empty()) device_list().push_back(device);
    dev.options.seed = seed;
    std::vector<int> distribution;
    if (!dist_file.empty()) {
      std::ifstream f(dist_file);
      int v;
      while (f >> v) distribution.push_back(v);
    } else {
      distribution.assign(100000, 0);  // cg_count column is filled by the extraction step, which is not part of this tool
    }
    const auto out = metaprofile::alphabeta_multiple(a, max_gene_length, distribution);
    std::printf("%s\n", out.results_txt.c_str());
    std::ofstream(std::filesystem::path(a.output_dir) / "results.txt") << out.results_txt;
    metaprofile::write_raw_npy(out, (std::filesystem::path(a.output_dir) / "raw.npy").string());
  } catch (const Error& e) {
    std::printf("Error: %s\n", e.what());
    return 1;
  }
  return 0;
}
