// The reference's own enabled unit tests on the ABneutral path, restated against the C++ mirror of its API
// (alphabeta.hpp) and run on the MI355X.  One function per reference test, same names, same fixtures
// (paths relative to the working directory, like the reference's `./data/...`), same assertions.
//   src/divergence.rs:138-161   same_as_r                 (tolerance 1e-4, src/macros.rs:13-22)
//   src/structs.rs:225-240      test_cost_function        (bit-exact assert_eq)
//   src/pedigree.rs:344-358     build_pedigree            (shape 6 x 4, writes pedigree_generated.txt)
// Exit code = number of failed tests.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>

#include "alphabeta.hpp"

using namespace alphabeta;

static int failures = 0;
#define CHECK(cond, msg)                                      \
  do {                                                        \
    if (!(cond)) {                                            \
      std::printf("  FAILED: %s (%s)\n", msg, #cond);         \
      ++failures;                                             \
      return;                                                 \
    }                                                         \
  } while (0)

// src/divergence.rs:138-161
static void same_as_r() {
  const Pedigree pedigree = Pedigree::from_file("./data/pedigree.txt");
  const auto d = divergence::divergence(pedigree, 0.25, 0.0, 0.75, 3.974271e-09, 1.519045e-07, 0.06892953);
  std::vector<double> r;
  std::ifstream f("./data/divergence.txt");
  double v;
  while (f >> v) r.push_back(v);
  CHECK(d.dt1t2.size() == r.size(), "divergence.dt1t2.len() == r.len()");
  for (size_t i = 0; i < r.size(); ++i) {
    CHECK(std::isnormal(d.dt1t2[i]), "dt1t2 is normal");
    CHECK(!(std::fabs(d.dt1t2[i] - r[i]) > 1e-4), "assert_close!(dt1t2[i], r)");
  }
  std::printf("test divergence::test::same_as_r ... ok\n");
}

// src/structs.rs:225-240
static void test_cost_function() {
  Problem p;  // Problem::default, src/structs.rs:172-189
  p.pedigree = Pedigree::from_file("./data/pedigree.txt");
  const std::vector<double> param = Model().to_vec();  // Model::default, :66-75
  const double result = p.cost(param);
  CHECK(result == 0.0006700888539608879, "assert_eq!(result, 0.0006700888539608879)");
  std::printf("test structs::test::test_cost_function ... ok\n");
}

// src/pedigree.rs:344-358
static void build_pedigree() {
  auto [pedigree, p0uu] = Pedigree::build("./data/nodelist.txt", "./data/edgelist.txt", 0.99, /*gpu_pairwise=*/true);
  CHECK(pedigree.nrows() == 4 * 3 / 2, "assert_eq!(pedigree.0.shape(), &[4 * 3 / 2, 4])");
  const Pedigree want = Pedigree::from_file("./pedigree_generated.txt");
  CHECK(pedigree.data == want.data, "rows equal data/pedigree_generated.txt");
  CHECK(p0uu == 0.6554051647850447, "p0uu");
  std::printf("test pedigree::tests::build_pedigree ... ok\n");
}

int main() {
  try {
    same_as_r();
    test_cost_function();
    build_pedigree();
  } catch (const Error& e) {
    std::printf("error: %s\n", e.what());
    return 100;
  }
  std::printf("test result: %s. %d failed\n", failures ? "FAILED" : "ok", failures);
  return failures;
}
