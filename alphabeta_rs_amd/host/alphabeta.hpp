// alphabeta.hpp — C++ host-side mirror of the reference crate's interface for the ABneutral path, on top
// of the C-ABI (include/abneutral.h).  The reference's host language is Rust, which this image lacks
// (no rustc/cargo); INTEGRATION.md shows the Rust `extern "C"` binding.  Names, argument meaning and
// error behaviour follow the reference so that tests read like its own:
//
//   alphabeta::Pedigree            src/pedigree.rs:44-45   (N x 4 f64: t0, t1, t2, D)
//     ::from_file / ::to_file      src/pedigree.rs:62-90
//     ::build                      src/pedigree.rs:92-193  (nodelist/edgelist/methylomes -> rows, p0uu)
//   alphabeta::Model               src/structs.rs:20-26,66-75,130-169
//   alphabeta::Problem::cost       src/structs.rs:11-19,191-217       -> abn_cost_batch (strict order)
//   alphabeta::ab_neutral::run     src/ab_neutral.rs:13-142           -> abn_ab_neutral_run
//   alphabeta::boot_model::run     src/boot_model.rs:17-115           -> abn_boot_model_run + abn_analyze
//   alphabeta::RawAnalysis/Analysis/CI  src/analysis.rs:12-195
//   alphabeta::run                 src/alphabeta.rs:23-59
//   alphabeta::steady_state / p_uu_est / p_mm_est   src/alphabeta.rs:62-79
//
// Errors: the reference returns Result<_, Box<dyn Error>> (and panics in places); here every failure is
// a thrown alphabeta::Error carrying the abn_status.  All numerics run on the GPU through the C-ABI;
// there is no host fallback.
#pragma once

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <functional>
#include <limits>
#include <map>
#include <queue>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/abneutral.h"

namespace alphabeta {

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

// Rust's `{}` for f64: shortest digits that round-trip, never scientific notation.
inline std::string fmt_f64(double v) {
  if (std::isnan(v)) return "NaN";
  if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
  char buf[400];
  auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
  return std::string(buf, r.ptr);
}

using PredictedDivergence = std::vector<double>;  // src/structs.rs:28
using Residuals = std::vector<double>;            // src/structs.rs:29

// progress callback standing in for Option<&ProgressBar> (src/ab_neutral.rs:19): called with the
// number of finished fits once the batch completes (the GPU finishes them together)
using Progress = std::function<void(size_t)>;

// ------------------------------------------------------------------------------------------------
class Device {  // RAII abn_ctx
 public:
  explicit Device(int ordinal = 0) {
    int rc = abn_init(ordinal, nullptr, &ctx_);
    if (rc) throw Error(rc, std::string("abn_init: ") + abn_status_string(rc) + " (no CPU fallback exists)");
  }
  ~Device() {
    if (ctx_) abn_shutdown(ctx_);
  }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
  abn_ctx* get() const { return ctx_; }
  void check(int rc, const char* what) const {
    if (rc) throw Error(rc, std::string(what) + ": " + abn_status_string(rc) + " — " + abn_last_error(ctx_));
  }
  abn_options options;  // seed, lanes, ... (abn_default_options)
  void init_options() { abn_default_options(&options); }

 private:
  abn_ctx* ctx_ = nullptr;
};

// HIP ordinals the batched entry points spread their work over (--devices a,b,c; default: the one default device).
// More than one: abn_multi_* — a plan per device, windows (or, for a single pedigree, bootstraps) sharded in
// contiguous blocks, the bootstrap tables gathered with RCCL over xGMI; byte-identical results for any device count.
inline std::vector<int32_t>& device_list() {
  static std::vector<int32_t> devs;
  return devs;
}
inline std::vector<int32_t> parse_device_list(const std::string& arg) {
  std::vector<int32_t> out;
  std::stringstream ss(arg);
  std::string tok;
  while (std::getline(ss, tok, ',')) {
    if (tok.empty()) continue;
    char* end = nullptr;
    const long v = std::strtol(tok.c_str(), &end, 10);
    if (*end != '\0' || v < 0) throw Error(ABN_ERR_INVALID_ARG, "--devices expects a comma-separated list of HIP ordinals");
    out.push_back((int32_t)v);
  }
  if (out.empty()) throw Error(ABN_ERR_INVALID_ARG, "--devices: empty list");
  return out;
}

// RAII abn_multi
class MultiDevice {
 public:
  MultiDevice(const std::vector<int32_t>& devs, const abn_options& opts, const double* gens, size_t n_rows, size_t n_windows,
              size_t n_starts, size_t n_boot) {
    const int rc = abn_multi_create(devs.data(), (int32_t)devs.size(), &opts, gens, (int32_t)n_rows, (int32_t)n_windows,
                                    (int32_t)n_starts, (int32_t)n_boot, &m_);
    if (rc) {
      const std::string why = m_ ? abn_multi_last_error(m_) : "";
      if (m_) abn_multi_destroy(m_);
      m_ = nullptr;
      throw Error(rc, std::string("abn_multi_create: ") + abn_status_string(rc) + " — " + why);
    }
  }
  ~MultiDevice() {
    if (m_) abn_multi_destroy(m_);
  }
  MultiDevice(const MultiDevice&) = delete;
  MultiDevice& operator=(const MultiDevice&) = delete;
  abn_multi* get() const { return m_; }
  void check(int rc, const char* what) const {
    if (rc) throw Error(rc, std::string(what) + ": " + abn_status_string(rc) + " — " + abn_multi_last_error(m_));
  }

 private:
  abn_multi* m_ = nullptr;
};

inline Device& default_device(int ordinal = -1) {
  static int chosen = 0;
  if (ordinal >= 0) chosen = ordinal;
  static Device dev(chosen);
  static bool once = (dev.init_options(), true);
  (void)once;
  return dev;
}

// ------------------------------------------------------------------------------------------------
// src/alphabeta.rs:62-79, src/structs.rs:146-159 (host copies for reporting only)
inline double p_uu_est(double a, double b) {
  return (b * ((1.0 - b) * (1.0 - b) - (1.0 - a) * (1.0 - a) - 1.0)) / ((a + b) * ((a + b - 1.0) * (a + b - 1.0) - 2.0));
}
inline double p_mm_est(double a, double b) {
  return (a * ((1.0 - a) * (1.0 - a) - (1.0 - b) * (1.0 - b) - 1.0)) / ((a + b) * ((a + b - 1.0) * (a + b - 1.0) - 2.0));
}
inline double steady_state(double a, double b) {
  const double pi_2 = (4.0 * a * b * (a + b - 2.0)) / ((a + b) * ((a + b - 1.0) * (a + b - 1.0) - 2.0));
  return p_mm_est(a, b) + 0.5 * pi_2;
}

// ------------------------------------------------------------------------------------------------
struct Model {  // src/structs.rs:20-26
  double alpha = 0.0001179555, beta = 0.0001180614, weight = 0.03693534, intercept = 0.003023981;  // Default :66-75
  std::vector<double> to_vec() const { return {alpha, beta, weight, intercept}; }
  static Model from_vec(const std::vector<double>& v) { return Model{v[0], v[1], v[2], v[3]}; }
  static Model from_ptr(const double* v) { return Model{v[0], v[1], v[2], v[3]}; }
  double est_mm() const { return p_mm_est(alpha, beta); }
  double est_um() const {
    return (4.0 * alpha * beta * (alpha + beta - 2.0)) /
           ((alpha + beta) * ((alpha + beta - 1.0) * (alpha + beta - 1.0) - 2.0));
  }
  double est_uu() const { return p_uu_est(alpha, beta); }
  std::string display() const {  // impl Display, :55-63
    return "Model:\n\tAlpha: " + fmt_f64(alpha) + "\n\tBeta: " + fmt_f64(beta) + "\n\tWeight: " + fmt_f64(weight) +
           "\n\tIntercept: " + fmt_f64(intercept);
  }
  void to_file(const std::string& path) const {  // :160-169
    std::printf("Writing model to file: %s\n", path.c_str());
    std::ofstream f(path);
    f << "Alpha " << fmt_f64(alpha) << "\nBeta " << fmt_f64(beta) << "\nWeight " << fmt_f64(weight) << "\n Intercept "
      << fmt_f64(intercept) << "\n";
  }
};

// ------------------------------------------------------------------------------------------------
struct CI {  // src/analysis.rs:194-195
  double lo = 0, hi = 0;
};

struct Analysis {  // src/analysis.rs:15-47
  double alpha, beta, alphabeta, weight, intercept, pr_mm, pr_um, pr_uu;
  double sd_alpha, sd_beta, sd_alphabeta, sd_weight, sd_intercept, sd_pr_mm, sd_pr_um, sd_pr_uu;
  CI ci_alpha, ci_beta, ci_alphabeta, ci_weight, ci_intercept, ci_pr_mm, ci_pr_um, ci_pr_uu;

  std::string display() const {  // impl Display, src/analysis.rs:147-187
    std::ostringstream o;
    auto kv = [&](const char* k, double v) { o << k << "\t" << fmt_f64(v) << "\n"; };
    auto ci = [&](const char* k, const CI& c) { o << k << "\t" << fmt_f64(c.lo) << "-" << fmt_f64(c.hi) << "\n"; };
    kv("Alpha", alpha); kv("Beta", beta); kv("AlphaBeta", alphabeta); kv("Weight", weight); kv("Intercept", intercept);
    kv("PrMM", pr_mm); kv("PrUM", pr_um); kv("PrUU", pr_uu);
    kv("SDAlpha", sd_alpha); kv("SDBeta", sd_beta); kv("SDAlphaBeta", sd_alphabeta); kv("SDWeight", sd_weight);
    kv("SDIntercept", sd_intercept); kv("SDPrMM", sd_pr_mm); kv("SDPrUM", sd_pr_um); kv("SDPrUU", sd_pr_uu);
    ci("CIAlpha", ci_alpha); ci("CIBeta", ci_beta); ci("CIAlphaBeta", ci_alphabeta); ci("CIWeight", ci_weight);
    ci("CIIntercept", ci_intercept); ci("CIPrMM", ci_pr_mm); ci("CIPrUM", ci_pr_um); ci("CIPrUU", ci_pr_uu);
    return o.str();
  }
  void to_file(const std::string& path) const {  // src/analysis.rs:101-145
    std::printf("Writing model to file: %s\n", path.c_str());
    std::ofstream f(path);
    f << display();
  }
};

struct RawAnalysis {  // src/analysis.rs:12 — n_boot x 7: alpha, beta, weight, intercept, pr_mm, pr_um, pr_uu
  std::vector<double> rows;
  size_t n_boot = 0;
  Analysis analyze() const {  // src/analysis.rs:50-98
    // A bootstrap fit without a finite best vertex is a NaN row here (status ABN_FIT_NONFINITE); the reference never gets
    // this far (best_param.unwrap() panics, src/boot_model.rs:86).  Quantiles of NaN have no order — abn_analyze sorts with
    // `<` and requires NaN-free columns (include/abneutral.h) —, so the mirror refuses the table like the reference does.
    for (size_t b = 0; b < n_boot; ++b) {
      bool bad = std::isnan(rows[7 * b + 1] / rows[7 * b]);  // beta / alpha, src/analysis.rs:54
      for (size_t k = 0; k < 7; ++k) bad = bad || std::isnan(rows[7 * b + k]);
      if (bad)
        throw Error(ABN_ERR_NO_FINITE_FIT, "bootstrap " + std::to_string(b) + " has no finite fit: no analysis of this table");
    }
    double o[32];
    int rc = abn_analyze(rows.data(), (int64_t)n_boot, o);
    if (rc) throw Error(rc, "abn_analyze failed");
    Analysis a{};
    a.alpha = o[0]; a.beta = o[1]; a.alphabeta = o[2]; a.weight = o[3]; a.intercept = o[4];
    a.pr_mm = o[5]; a.pr_um = o[6]; a.pr_uu = o[7];
    a.sd_alpha = o[8]; a.sd_beta = o[9]; a.sd_alphabeta = o[10]; a.sd_weight = o[11]; a.sd_intercept = o[12];
    a.sd_pr_mm = o[13]; a.sd_pr_um = o[14]; a.sd_pr_uu = o[15];
    CI* cis[8] = {&a.ci_alpha, &a.ci_beta, &a.ci_alphabeta, &a.ci_weight, &a.ci_intercept, &a.ci_pr_mm, &a.ci_pr_um,
                  &a.ci_pr_uu};
    for (int k = 0; k < 8; ++k) *cis[k] = CI{o[16 + k], o[24 + k]};
    return a;
  }
  // ndarray_npy::write_npy (src/cli/alphabeta.rs:34): NPY v1.0, C order, <f8, shape (n_boot, 7)
  void write_npy(const std::string& path) const {
    std::string dict = "{'descr': '<f8', 'fortran_order': False, 'shape': (" + std::to_string(n_boot) + ", 7), }";
    size_t total = 10 + dict.size() + 1;
    size_t pad = (64 - total % 64) % 64;
    dict += std::string(pad, ' ') + "\n";
    std::ofstream f(path, std::ios::binary);
    const char magic[8] = {'\x93', 'N', 'U', 'M', 'P', 'Y', 1, 0};
    f.write(magic, 8);
    const uint16_t hl = (uint16_t)dict.size();
    f.write(reinterpret_cast<const char*>(&hl), 2);
    f.write(dict.data(), (std::streamsize)dict.size());
    f.write(reinterpret_cast<const char*>(rows.data()), (std::streamsize)(rows.size() * sizeof(double)));
  }
};

// ------------------------------------------------------------------------------------------------
class Pedigree {  // src/pedigree.rs:44-45
 public:
  std::vector<double> data;  // row-major N x 4
  size_t nrows() const { return data.size() / 4; }
  double at(size_t i, size_t c) const { return data[4 * i + c]; }
  void push_row(double t0, double t1, double t2, double d) {
    data.push_back(t0); data.push_back(t1); data.push_back(t2); data.push_back(d);
  }

  // src/pedigree.rs:62-79: skip the header, space-separated rows; panics (throws) on malformed input
  static Pedigree from_file(const std::string& filename) {
    std::ifstream f(filename);
    if (!f) throw Error(ABN_ERR_INVALID_ARG, "could not read pedigree file " + filename);
    Pedigree p;
    std::string line;
    std::getline(f, line);
    while (std::getline(f, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (line.empty()) continue;
      double v[4];
      size_t pos = 0;
      for (int k = 0; k < 4; ++k) {
        size_t end = line.find_first_of(" \t", pos);  // the reference splits on ' ' only; tabs accepted too
        std::string tok = line.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
        char* e = nullptr;
        v[k] = std::strtod(tok.c_str(), &e);
        if (tok.empty() || *e) throw Error(ABN_ERR_INVALID_ARG, "malformed pedigree line: " + line);
        pos = end == std::string::npos ? line.size() : end + 1;
      }
      p.push_row(v[0], v[1], v[2], v[3]);
    }
    return p;
  }

  // src/pedigree.rs:81-90
  void to_file(const std::string& path) const {
    std::printf("Writing pedigree to file: %s\n", path.c_str());
    std::ofstream f(path);
    if (!f) throw Error(ABN_ERR_INVALID_ARG, "could not write " + path);
    f << "time0\ttime1\ttime2\tD.value\n";
    for (size_t i = 0; i < nrows(); ++i)
      f << fmt_f64(at(i, 0)) << "\t" << fmt_f64(at(i, 1)) << "\t" << fmt_f64(at(i, 2)) << "\t" << fmt_f64(at(i, 3)) << "\n";
  }

  // src/pedigree.rs:92-193; defined in pedigree_build.hpp.  gpu_pairwise: the O(pairs x sites) status
  // comparison (DMatrix::from, :210-261) runs on the MI355X (abn_pairwise_divergence); same bits either way.
  static std::pair<Pedigree, double> build(const std::string& nodelist, const std::string& edgelist,
                                           double posterior_max_filter, bool gpu_pairwise = false);
};

// ------------------------------------------------------------------------------------------------
struct Problem {  // src/structs.rs:11-19
  Pedigree pedigree;
  double eqp_weight = 0.7, eqp = 0.5, p_mm = 0.25, p_um = 0.0, p_uu = 0.75;  // Default :172-189 (+ data/pedigree.txt)
  // impl CostFunction, src/structs.rs:191-217.  Serial row-order accumulation (strict_order = 1), i.e.
  // the reference's own summation order, evaluated on the GPU.
  double cost(const std::vector<double>& p) const {
    Device& dev = default_device();
    abn_options o = dev.options;
    o.strict_order = 1;
    double c = 0.0;
    dev.check(abn_cost_batch(dev.get(), &o, pedigree.data.data(), (int32_t)pedigree.nrows(), p_uu, eqp, eqp_weight,
                             p.data(), 1, nullptr, nullptr, nullptr, nullptr, 0, &c, nullptr, nullptr),
              "Problem::cost");
    return c;
  }
};

namespace divergence {
struct Divergence {  // src/divergence.rs:10-14
  std::vector<double> dt1t2;
  double p_uu;
};
// src/divergence.rs:33-94 (one candidate through abn_cost_batch; `_p_um` is unused there as well)
inline Divergence divergence(const Pedigree& pedigree, double p_mm, double /*_p_um*/, double p_uu, double alpha,
                             double beta, double weight) {
  Device& dev = default_device();
  if (p_mm != 1.0 - p_uu)  // the device path derives p_mm as the callers do (src/ab_neutral.rs:23)
    throw Error(ABN_ERR_INVALID_ARG, "divergence: p_mm must equal 1 - p_uu");
  const double x[4] = {alpha, beta, weight, 0.0};
  Divergence d;
  d.dt1t2.resize(pedigree.nrows());
  double c = 0.0;
  dev.check(abn_cost_batch(dev.get(), &dev.options, pedigree.data.data(), (int32_t)pedigree.nrows(), p_uu, p_uu, 0.0, x,
                           1, nullptr, nullptr, nullptr, nullptr, 0, &c, d.dt1t2.data(), &d.p_uu),
            "divergence");
  return d;
}
}  // namespace divergence

namespace ab_neutral {
// src/ab_neutral.rs:13-20
inline std::tuple<Model, PredictedDivergence, Residuals> run(const Pedigree& pedigree, double p0uu, double eqp,
                                                             double eqp_weight, size_t n_starts,
                                                             const Progress& pb = nullptr) {
  Device& dev = default_device();
  const double p0mm = 1.0 - p0uu, p0um = 0.0;
  if (p0mm + p0uu + p0um != 1.0)  // assert_eq!, src/ab_neutral.rs:31
    throw Error(ABN_ERR_INVALID_ARG, "p0mm + p0uu + p0um != 1");
  const size_t n = pedigree.nrows();
  double m[4];
  PredictedDivergence pred(n);
  Residuals resid(n);
  dev.check(abn_ab_neutral_run(dev.get(), &dev.options, pedigree.data.data(), (int32_t)n, p0uu, eqp, eqp_weight,
                               (int32_t)n_starts, m, pred.data(), resid.data(), nullptr, nullptr, nullptr),
            "ab_neutral::run");
  if (pb) pb(n_starts);
  return {Model::from_ptr(m), std::move(pred), std::move(resid)};
}
}  // namespace ab_neutral

namespace boot_model {
// src/boot_model.rs:17-28 (the PNG of :105-109 is not produced; output_dir is accepted for signature parity)
inline std::pair<Analysis, RawAnalysis> run(const Pedigree& pedigree, const Model& params, PredictedDivergence pred_div,
                                            Residuals residuals, double p0uu, double eqp, double eqp_weight,
                                            size_t n_boot, const Progress& pb = nullptr,
                                            const std::string& output_dir = ".") {
  (void)output_dir;
  Device& dev = default_device();
  const double p0mm = 1.0 - p0uu, p0um = 0.0;
  if (p0mm + p0uu + p0um != 1.0) throw Error(ABN_ERR_INVALID_ARG, "p0mm + p0uu + p0um != 1");  // :35
  const size_t n = pedigree.nrows();
  if (pred_div.size() != n || residuals.size() != n) throw Error(ABN_ERR_INVALID_ARG, "pred/resid length");  // :49
  RawAnalysis raw;
  raw.n_boot = n_boot;
  raw.rows.resize(n_boot * 7);
  const double m[4] = {params.alpha, params.beta, params.weight, params.intercept};
  dev.check(abn_boot_model_run(dev.get(), &dev.options, pedigree.data.data(), (int32_t)n, m, pred_div.data(),
                               residuals.data(), p0uu, eqp, eqp_weight, (int32_t)n_boot, raw.rows.data(), nullptr),
            "boot_model::run");
  if (pb) pb(n_boot);
  Analysis a = raw.analyze();
  return {a, std::move(raw)};
}
}  // namespace boot_model

struct Args {  // arguments::AlphaBeta, src/arguments.rs:93-114
  size_t iterations = 1000;
  std::string edges = "./edgelist.txt", nodes = "./nodelist.txt";
  double posterior_max_filter = 0.99;
  std::string output = ".";
};

struct RunResult {
  Model model;
  Analysis analysis;
  RawAnalysis raw_analysis;
  Pedigree pedigree;
  double obs_steady_state;
};

// src/alphabeta.rs:23-59 on several GPUs: one window, the bootstraps sharded over the devices (every device repeats the
// cheap multi-start phase: same inputs, same bits), tables gathered with RCCL.  Same bits as the one-device path.
inline RunResult run_on_pedigree_multi(Pedigree pedigree, double p0uu, size_t iterations, const std::vector<int32_t>& devs) {
  Device& dev = default_device();
  const size_t n = pedigree.nrows();
  std::vector<double> gens(n * 3), d(n);
  for (size_t i = 0; i < n; ++i) {
    for (size_t c = 0; c < 3; ++c) gens[i * 3 + c] = pedigree.at(i, c);
    d[i] = pedigree.at(i, 3);
  }
  MultiDevice md(devs, dev.options, gens.data(), n, 1, iterations, iterations);
  md.check(abn_multi_set_windows(md.get(), d.data(), &p0uu, nullptr, nullptr), "abn_multi_set_windows");  // eqp = p0uu, weight 1
  md.check(abn_multi_run(md.get()), "abn_multi_run");
  double m[4];
  RawAnalysis raw;
  raw.n_boot = iterations;
  raw.rows.resize(iterations * 7);
  md.check(abn_multi_download(md.get(), m, nullptr, nullptr, raw.rows.data(), nullptr, nullptr, nullptr), "abn_multi_download");
  Analysis a = raw.analyze();
  return RunResult{Model::from_ptr(m), a, std::move(raw), std::move(pedigree), 1.0 - p0uu};
}

// src/alphabeta.rs:23-59 with an already built pedigree
inline RunResult run_on_pedigree(Pedigree pedigree, double p0uu, size_t iterations, const std::string& output) {
  if (device_list().size() > 1) return run_on_pedigree_multi(std::move(pedigree), p0uu, iterations, device_list());
  auto [model, pred_div, residuals] = ab_neutral::run(pedigree, p0uu, p0uu, 1.0, iterations);
  auto [analysis, raw] = boot_model::run(pedigree, model, std::move(pred_div), std::move(residuals), p0uu, p0uu, 1.0,
                                         iterations, nullptr, output);
  return RunResult{model, analysis, std::move(raw), std::move(pedigree), 1.0 - p0uu};
}

}  // namespace alphabeta

#include "pedigree_build.hpp"

namespace alphabeta {
// src/alphabeta.rs:23-59
inline RunResult run(const Args& args) {
  std::printf("Building pedigree...\n");
  auto [pedigree, p0uu] = Pedigree::build(args.nodes, args.edges, args.posterior_max_filter, /*gpu_pairwise=*/true);
  return run_on_pedigree(std::move(pedigree), p0uu, args.iterations, args.output);
}
}  // namespace alphabeta
