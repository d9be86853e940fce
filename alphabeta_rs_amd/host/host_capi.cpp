// Tiny C shim over the C++ host layer so that the Python tests can call Pedigree::build / from_file /
// to_file and the number formatter without a GPU.  Not part of the product ABI (include/abneutral.h).
#include <cstring>

#include "alphabeta.hpp"

extern "C" {
// returns the number of rows (<0 on error); rows (capacity cap x 4) and *p0uu are filled
int abh_pedigree_build(const char* nodelist, const char* edgelist, double posterior_max_filter, double* rows, int cap,
                       double* p0uu, char* err, int errcap) {
  try {
    auto [ped, p0] = alphabeta::Pedigree::build(nodelist, edgelist, posterior_max_filter);
    const int n = (int)ped.nrows();
    if (n > cap) return -2;
    if (n > 0) std::memcpy(rows, ped.data.data(), sizeof(double) * 4 * (size_t)n);  // an empty pedigree has no buffer
    *p0uu = p0;
    return n;
  } catch (const std::exception& e) {
    if (err && errcap > 0) std::strncpy(err, e.what(), (size_t)errcap - 1), err[errcap - 1] = 0;
    return -1;
  }
}
int abh_pedigree_roundtrip(const char* in_path, const char* out_path) {
  try {
    auto ped = alphabeta::Pedigree::from_file(in_path);
    ped.to_file(out_path);
    return (int)ped.nrows();
  } catch (const std::exception&) {
    return -1;
  }
}
int abh_fmt_f64(double v, char* out, int cap) {
  const std::string s = alphabeta::fmt_f64(v);
  if ((int)s.size() + 1 > cap) return -1;
  std::memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}
// RawAnalysis::analyze -> 0 and the mean of alpha, or -1 and the error text (a table with non-finite fits)
int abh_analyze(const double* rows, long long n_boot, double* mean_alpha, char* err, int errcap) {
  try {
    alphabeta::RawAnalysis r;
    r.n_boot = (size_t)n_boot;
    r.rows.assign(rows, rows + 7 * n_boot);
    *mean_alpha = r.analyze().alpha;
    return 0;
  } catch (const std::exception& e) {
    if (err && errcap > 0) std::strncpy(err, e.what(), (size_t)errcap - 1), err[errcap - 1] = 0;
    return -1;
  }
}
int abh_write_npy(const char* path, const double* rows, long long n_boot) {
  alphabeta::RawAnalysis r;
  r.n_boot = (size_t)n_boot;
  r.rows.assign(rows, rows + 7 * n_boot);
  r.write_npy(path);
  return 0;
}
}
