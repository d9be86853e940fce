// metaprofile.hpp — the `alphabeta_multiple` half of the reference's `metaprofile` binary
// (src/cli/metaprofile.rs:33-114; SURVEY.md §8f row 2, "next"): the SERIAL window loop that calls
// alphabeta::run once per (region, window) directory becomes ONE batched, device-resident plan
// (abn_plan_*: W windows x (S starts + B bootstraps) in three kernel launches).  The window extraction
// that fills those directories (src/extract.rs, src/windows.rs, src/setup.rs) is out of scope; this driver
// starts from the directory tree setup.rs writes: <output_dir>/<region>/<window>/{nodelist,edgelist}.txt.
//
// Outputs as the reference: results.txt (:74-99, ';'-separated, one line per successful window) and
// raw.npy, the (iterations, 7, n_windows) array of :49,68,110.  The metaplot PNG (:113) is not produced.
#pragma once

#include <filesystem>
#include <iostream>

#include "alphabeta.hpp"

namespace alphabeta {
namespace metaprofile {

struct WindowArgs {  // the fields of arguments::Windows (src/arguments.rs:6-62) this driver reads
  std::string output_dir = ".";
  std::string name = "Anonymous Run";
  uint32_t window_step = 5;      // 0 means window_size (src/cli/metaprofile.rs:18-20)
  uint32_t window_size = 5;
  uint32_t cutoff = 2048;
  bool absolute = false;
  size_t iterations = 100;
  double posterior_max_filter = 0.99;  // AlphaBeta::default (src/arguments.rs:142-152)
};

struct WindowResult {
  Model model;
  Analysis analysis;
  std::string region;
  double obs_meth_lvl;
  size_t window_index;  // position in the (region, window) enumeration
};

struct Output {
  std::vector<WindowResult> results;
  std::vector<double> raw;  // (iterations, 7, n_ok) C order
  size_t n_ok = 0, iterations = 0;
  std::string results_txt;
};

// src/cli/metaprofile.rs:33-114
inline Output alphabeta_multiple(const WindowArgs& args, uint32_t max_gene_length, const std::vector<int>& distribution) {
  namespace fs = std::filesystem;
  const uint32_t step = args.window_step == 0 ? args.window_size : args.window_step;
  const std::vector<std::pair<std::string, uint32_t>> regions = {
      {"upstream", args.cutoff}, {"gene", max_gene_length}, {"downstream", args.cutoff}};  // :34-38
  struct Win {
    std::string region;
    Pedigree ped;
    double p0uu;
    size_t index;
  };
  std::vector<Win> wins;
  size_t index = 0;
  for (const auto& region : regions) {
    const uint32_t max = args.absolute ? region.second : 100;
    for (uint32_t window = 0; window < max; window += step, ++index) {
      const fs::path dir = fs::path(args.output_dir) / region.first / std::to_string(window);
      try {  // alphabeta::run's pedigree build; a failing window is reported and skipped (:64-65)
        auto [ped, p0uu] = Pedigree::build((dir / "nodelist.txt").string(), (dir / "edgelist.txt").string(),
                                           args.posterior_max_filter, /*gpu_pairwise=*/true);
        if (ped.nrows() == 0) throw Error(ABN_ERR_BAD_PEDIGREE, "empty pedigree");
        wins.push_back(Win{region.first, std::move(ped), p0uu, index});
      } catch (const std::exception& e) {
        std::printf("Error: Error while building pedigree: %s\n", e.what());
      }
    }
  }
  Output out;
  out.iterations = args.iterations;
  if (wins.empty()) return out;

  // Windows that share the (t0, t1, t2) rows (the normal case: one nodelist/edgelist for all windows) are
  // fitted by one plan; anything else gets a plan of its own.
  auto same_topology = [](const Pedigree& a, const Pedigree& b) {
    if (a.nrows() != b.nrows()) return false;
    for (size_t i = 0; i < a.nrows(); ++i)
      for (size_t c = 0; c < 3; ++c)
        if (a.at(i, c) != b.at(i, c)) return false;
    return true;
  };
  std::vector<int> group(wins.size(), -1);
  int ngroups = 0;
  for (size_t i = 0; i < wins.size(); ++i) {
    if (group[i] >= 0) continue;
    group[i] = ngroups;
    for (size_t j = i + 1; j < wins.size(); ++j)
      if (group[j] < 0 && same_topology(wins[i].ped, wins[j].ped)) group[j] = ngroups;
    ++ngroups;
  }
  Device& dev = default_device();
  const size_t B = args.iterations, S = args.iterations;  // n_starts = n_boot = iterations (src/alphabeta.rs:33-54)
  std::vector<Model> models(wins.size());
  std::vector<RawAnalysis> raws(wins.size());
  std::vector<char> ok(wins.size(), 0);
  for (int g = 0; g < ngroups; ++g) {
    std::vector<size_t> members;
    for (size_t i = 0; i < wins.size(); ++i)
      if (group[i] == g) members.push_back(i);
    const Pedigree& p0 = wins[members[0]].ped;
    const size_t N = p0.nrows(), W = members.size();
    std::vector<double> gens(N * 3), D(W * N), p0uu(W);
    for (size_t i = 0; i < N; ++i)
      for (size_t c = 0; c < 3; ++c) gens[i * 3 + c] = p0.at(i, c);
    for (size_t w = 0; w < W; ++w) {
      for (size_t i = 0; i < N; ++i) D[w * N + i] = wins[members[w]].ped.at(i, 3);
      p0uu[w] = wins[members[w]].p0uu;
    }
    // every window draws from the Philox streams of ITS position in the (region, window) enumeration, whatever
    // windows failed before it and whichever topology group it landed in
    std::vector<uint32_t> ids(W);
    for (size_t w = 0; w < W; ++w) ids[w] = (uint32_t)wins[members[w]].index;
    std::vector<double> mod(W * 4), raw(W * B * 7);
    std::vector<int32_t> best(W);
    // one plan per device (abn_multi_*: --devices; one device = one plan), windows in contiguous blocks, the bootstrap
    // tables gathered with RCCL: the loop of src/cli/metaprofile.rs:50-72 in three launches per device
    std::vector<int32_t> devs = device_list();
    if (devs.empty()) devs.push_back(0);
    {
      MultiDevice md(devs, dev.options, gens.data(), N, W, S, B);
      md.check(abn_multi_set_window_ids(md.get(), ids.data()), "abn_multi_set_window_ids");
      md.check(abn_multi_set_windows(md.get(), D.data(), p0uu.data(), nullptr, nullptr), "abn_multi_set_windows");
      md.check(abn_multi_run(md.get()), "abn_multi_run");
      int rc = abn_multi_download(md.get(), mod.data(), nullptr, nullptr, raw.data(), nullptr, nullptr, best.data());
      if (rc == ABN_ERR_NO_FINITE_FIT) rc = ABN_OK;  // per window below: best[w] < 0 is printed and skipped (:64-65)
      md.check(rc, "metaprofile plan");
    }
    for (size_t w = 0; w < W; ++w) {
      const size_t i = members[w];
      if (best[w] < 0) {
        std::printf("Error: Model failed: %s\n", abn_status_string(ABN_ERR_NO_FINITE_FIT));
        continue;
      }
      models[i] = Model::from_ptr(&mod[w * 4]);
      raws[i].n_boot = B;
      raws[i].rows.assign(raw.begin() + (std::ptrdiff_t)(w * B * 7), raw.begin() + (std::ptrdiff_t)((w + 1) * B * 7));
      ok[i] = 1;
    }
  }
  for (size_t i = 0; i < wins.size(); ++i) {
    if (!ok[i]) continue;
    try {
      out.results.push_back(WindowResult{models[i], raws[i].analyze(), wins[i].region, 1.0 - wins[i].p0uu, wins[i].index});
    } catch (const Error& e) {  // a bootstrap table with non-finite fits: this window fails, the others go on (:64-65)
      std::printf("Error: Model failed: %s\n", e.what());
      ok[i] = 0;
    }
  }
  out.n_ok = out.results.size();
  // raw_analyses.push(Axis(2), ...) -> (iterations, 7, n_ok), C order (:49,68)
  out.raw.assign(B * 7 * out.n_ok, 0.0);
  {
    size_t k = 0;
    for (size_t i = 0; i < wins.size(); ++i) {
      if (!ok[i]) continue;
      for (size_t b = 0; b < B; ++b)
        for (size_t c = 0; c < 7; ++c) out.raw[(b * 7 + c) * out.n_ok + k] = raws[i].rows[b * 7 + c];
      ++k;
    }
  }
  // :74-96 — results zipped with `distribution`: the shorter of the two decides the number of lines
  std::string print =
      "run;window;cg_count;region;alpha;beta;1/2*(alpha+beta);pred_steady_state;obs_steady_state;sd_alpha;sd_beta;"
      "ci_alpha_0.025;ci_alpha_0.975;ci_beta_0.025;ci_beta_0.975\n";
  const size_t nl = std::min(out.results.size(), distribution.size());
  for (size_t i = 0; i < nl; ++i) {
    const auto& r = out.results[i];
    const Model& m = r.model;
    const Analysis& a = r.analysis;
    print += args.name + ";" + std::to_string(i) + ";" + std::to_string(distribution[i]) + ";" + r.region + ";" +
             fmt_f64(m.alpha) + ";" + fmt_f64(m.beta) + ";" + fmt_f64(0.5 * (m.alpha + m.beta)) + ";" +
             fmt_f64(steady_state(m.alpha, m.beta)) + ";" + fmt_f64(r.obs_meth_lvl) + ";" + fmt_f64(a.sd_alpha) + ";" +
             fmt_f64(a.sd_beta) + ";" + fmt_f64(a.ci_alpha.lo) + ";" + fmt_f64(a.ci_alpha.hi) + ";" +
             fmt_f64(a.ci_beta.lo) + ";" + fmt_f64(a.ci_beta.hi) + "\n";
  }
  out.results_txt = print;
  return out;
}

// ndarray_npy::write_npy of the (iterations, 7, n_windows) array (:110)
inline void write_raw_npy(const Output& o, const std::string& path) {
  std::string dict = "{'descr': '<f8', 'fortran_order': False, 'shape': (" + std::to_string(o.iterations) + ", 7, " +
                     std::to_string(o.n_ok) + "), }";
  const size_t total = 10 + dict.size() + 1;
  dict += std::string((64 - total % 64) % 64, ' ') + "\n";
  std::ofstream f(path, std::ios::binary);
  const char magic[8] = {'\x93', 'N', 'U', 'M', 'P', 'Y', 1, 0};
  f.write(magic, 8);
  const uint16_t hl = (uint16_t)dict.size();
  f.write(reinterpret_cast<const char*>(&hl), 2);
  f.write(dict.data(), (std::streamsize)dict.size());
  f.write(reinterpret_cast<const char*>(o.raw.data()), (std::streamsize)(o.raw.size() * sizeof(double)));
}

}  // namespace metaprofile
}  // namespace alphabeta
