// pedigree_build.hpp — Pedigree::build (src/pedigree.rs:92-337): nodelist / edgelist / methylome files ->
// (t0, t1, t2, D) rows and p0uu.  SURVEY.md §8f row 1 ("next"): host-side data preparation in front of
// the hot path, needed for the `alphabeta` CLI to be drop-in from raw inputs.  Plain C++ on the host;
// the O(pairs x sites) status comparison is integer work done once per run.
#pragma once

#include "alphabeta.hpp"

namespace alphabeta {
namespace detail {

struct Site {  // the fields of MethylationSite (src/methylation_site.rs:32-45) the pedigree build reads
  double posteriormax;
  uint32_t status_numeric;  // U=0, I=1, M=2 (src/methylation_site.rs:130-136)
  double meth_lvl;
};

inline std::vector<std::string> split_any(const std::string& s, const char* delims) {
  std::vector<std::string> out;
  size_t pos = 0;
  for (;;) {
    size_t e = s.find_first_of(delims, pos);
    out.push_back(s.substr(pos, e == std::string::npos ? std::string::npos : e - pos));
    if (e == std::string::npos) break;
    pos = e + 1;
  }
  return out;
}
inline bool parse_u32(const std::string& t, uint32_t& v) {  // str::parse::<u32>
  if (t.empty()) return false;
  size_t i = (t[0] == '+') ? 1 : 0;
  if (i >= t.size()) return false;
  uint64_t acc = 0;
  for (; i < t.size(); ++i) {
    if (t[i] < '0' || t[i] > '9') return false;
    acc = acc * 10 + (uint64_t)(t[i] - '0');
    if (acc > 0xffffffffull) return false;
  }
  v = (uint32_t)acc;
  return true;
}
inline bool parse_f64(const std::string& t, double& v) {  // str::parse::<f64>
  if (t.empty()) return false;
  char* e = nullptr;
  v = std::strtod(t.c_str(), &e);
  return *e == 0 && !std::isspace((unsigned char)t[0]);
}
inline bool parse_chromosome(std::string t) {  // src/methylation_site.rs:55-68
  while (t.rfind("chr", 0) == 0) t = t.substr(3);
  if (t == "M" || t == "C") return true;
  uint32_t n;
  return parse_u32(t, n) && n <= 255;
}
inline uint32_t status_from(char c) {  // src/methylation_site.rs:100-114
  if (c == 'M') return 2;
  if (c == 'I') return 1;
  if (c != 'U') std::printf("Warning: Encountered invalid methylation status: %c. Parsed as Unmethylated\n", c);
  return 0;
}

// MethylationSite::from_methylome_file_line (src/methylation_site.rs:146-362), reduced to what
// Pedigree::build consumes.  Formats are tried in the reference's order.
inline bool parse_site(const std::string& line, Site& out) {
  const auto tab = split_any(line, "\t");
  auto cg = [&](size_t chrom, size_t s0, int s1, size_t cm, size_t ct, size_t pm, size_t st, size_t ml) -> bool {
    uint32_t u;
    double pmax, lvl;
    if (!parse_chromosome(tab[chrom]) || !parse_u32(tab[s0], u)) return false;
    if (s1 >= 0 && !parse_u32(tab[(size_t)s1], u)) return false;
    if (!parse_u32(tab[cm], u) || !parse_u32(tab[ct], u) || !parse_f64(tab[pm], pmax)) return false;
    if (tab[st].empty() || !parse_f64(tab[ml], lvl)) return false;
    out = Site{pmax, status_from(tab[st][0]), lvl};
    return true;
  };
  if (tab.size() == 9 && tab[3] == "CG" && cg(0, 1, -1, 4, 5, 6, 7, 8)) return true;    // first_format
  if (tab.size() == 10 && tab[3] == "CG" && cg(0, 1, -1, 4, 5, 6, 7, 8)) return true;   // second_format
  if (tab.size() == 11 && tab[3] == "CG" && cg(0, 1, 2, 6, 7, 8, 9, 10)) return true;   // third_format
  const auto ws = split_any(line, "\t ");
  if (ws.size() == 4) {  // chromatin-state / bigwig rows: posteriormax 0, status U, level 0
    uint32_t a, b;
    if (parse_chromosome(ws[0]) && parse_u32(ws[1], a) && parse_u32(ws[2], b)) {
      out = Site{0.0, 0, 0.0};
      return true;
    }
  }
  return false;
}

struct Node {  // src/pedigree.rs:16-26
  size_t id;
  std::string file, name;
  uint32_t generation;
  bool meth;
  double rc_meth_lvl = 0.0;
  std::vector<Site> sites;
};

inline std::string read_file(const std::string& path, const char* what) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw Error(ABN_ERR_INVALID_ARG, std::string("Error while building pedigree: could not read ") + what + " " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

}  // namespace detail

inline std::pair<Pedigree, double> Pedigree::build(const std::string& nodelist, const std::string& edgelist,
                                                   double posterior_max_filter, bool gpu_pairwise) {
  using namespace detail;
  const std::string nodes_txt = read_file(nodelist, "nodelist"), edges_txt = read_file(edgelist, "edgelist");
  // :98-117
  std::vector<Node> all;
  {
    const auto lines = split_any(nodes_txt, "\n\r");
    for (size_t li = 1; li < lines.size(); ++li) {
      const auto e = split_any(lines[li], ",\t ");
      if (e.size() < 4) continue;
      uint32_t gen;
      if (!parse_u32(e[2], gen)) continue;
      all.push_back(Node{li - 1, e[0], e[1], gen, e[3] == "Y", 0.0, {}});
    }
  }
  if (all.empty()) throw Error(ABN_ERR_INVALID_ARG, "No nodes could be parsed from the nodelist");
  // :124-136
  struct EdgeRef { const Node* from; const Node* to; };
  std::vector<EdgeRef> edges;
  {
    const auto lines = split_any(edges_txt, "\n\r");
    for (size_t li = 1; li < lines.size(); ++li) {
      const auto e = split_any(lines[li], "\t ,");
      if (e.size() < 2) continue;
      const Node *f = nullptr, *t = nullptr;
      for (const auto& n : all) {
        if (!f && n.name == e[0]) f = &n;
        if (!t && n.name == e[1]) t = &n;
      }
      if (f && t) edges.push_back(EdgeRef{f, t});
    }
  }
  // :138-178 — load the methylomes of the sampled ("Y") nodes
  std::vector<Node> nodes;
  for (const auto& n : all)
    if (n.meth) nodes.push_back(n);
  for (auto& node : nodes) {
    std::ifstream f(node.file);
    if (!f) throw Error(ABN_ERR_INVALID_ARG, "Could not open node file: " + node.file);
    std::string line;
    while (std::getline(f, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      Site s;
      if (parse_site(line, s)) node.sites.push_back(s);
    }
    double sum = 0.0;
    size_t cnt = 0;
    for (const auto& s : node.sites)
      if (s.posteriormax >= posterior_max_filter) {
        sum += s.meth_lvl;
        ++cnt;
      }
    node.rc_meth_lvl = sum / (double)cnt;  // :165-166
  }
  double p0 = 0.0;  // :180-184
  for (const auto& n : nodes) p0 += 1.0 - n.rc_meth_lvl;
  p0 = p0 / (double)nodes.size();

  // DMatrix::from, :210-261 — entry [i][j - i - 1]
  const size_t nn = nodes.size();
  std::vector<double> dm(nn * nn, 0.0);
  bool same_len = true;
  for (size_t i = 1; i < nn; ++i) same_len = same_len && nodes[i].sites.size() == nodes[0].sites.size();
  if (gpu_pairwise && same_len && nn >= 2) {
    // one byte per (sample, site): status | 0x80 when the posterior is below the filter
    const size_t L = nodes[0].sites.size();
    std::vector<uint8_t> codes(nn * L);
    for (size_t i = 0; i < nn; ++i)
      for (size_t k = 0; k < L; ++k)
        codes[i * L + k] = (uint8_t)(nodes[i].sites[k].status_numeric |
                                     (nodes[i].sites[k].posteriormax < posterior_max_filter ? 0x80u : 0u));
    std::vector<double> dv(nn * (nn - 1) / 2);
    Device& dev = default_device();
    dev.check(abn_pairwise_divergence(dev.get(), codes.data(), (int32_t)nn, (int64_t)L, nullptr, nullptr, dv.data()),
              "Pedigree::build (pairwise divergence)");
    size_t p = 0;
    for (size_t i = 0; i < nn; ++i)
      for (size_t j = i + 1; j < nn; ++j) dm[i * nn + (j - i - 1)] = dv[p++];
  } else {
    for (size_t i = 0; i < nn; ++i)
      for (size_t j = i + 1; j < nn; ++j) {
        const auto &a = nodes[i].sites, &b = nodes[j].sites;
        if (a.size() != b.size()) {
          std::printf("Lengths do not match, all bets are off: %zu vs %zu\n", a.size(), b.size());
          dm[i * nn + (j - i - 1)] = 0.0;
          continue;
        }
        uint64_t div = 0, compared = 0;
        for (size_t k = 0; k < a.size(); ++k) {
          if (a[k].posteriormax < posterior_max_filter || b[k].posteriormax < posterior_max_filter) continue;
          div += a[k].status_numeric > b[k].status_numeric ? a[k].status_numeric - b[k].status_numeric
                                                           : b[k].status_numeric - a[k].status_numeric;
          ++compared;
        }
        dm[i * nn + (j - i - 1)] = (double)div / (2.0 * (double)compared);
      }
  }

  // DMatrix::convert, :263-337 — undirected graph, edge weight = |generation difference|
  size_t vmax = 0;
  for (const auto& e : edges) vmax = std::max(vmax, std::max(e.from->id, e.to->id) + 1);
  std::vector<std::vector<std::pair<size_t, size_t>>> adj(vmax);
  for (const auto& e : edges) {
    const size_t w = e.from->generation > e.to->generation ? e.from->generation - e.to->generation
                                                           : e.to->generation - e.from->generation;
    adj[e.from->id].push_back({e.to->id, w});
    adj[e.to->id].push_back({e.from->id, w});
  }
  auto generation_of = [&](size_t id) -> uint32_t {  // :298-309
    for (const auto& e : edges) {
      if (e.from->id == id) return e.from->generation;
      if (e.to->id == id) return e.to->generation;
    }
    throw Error(ABN_ERR_BAD_PEDIGREE, "node on a path is not part of any edge");
  };
  Pedigree ped;
  for (size_t i = 0; i < nn; ++i)
    for (size_t j = i + 1; j < nn; ++j) {
      const size_t src = nodes[i].id, dst = nodes[j].id;
      if (src == dst || src >= vmax || dst >= vmax) continue;
      // Dijkstra (astar with a zero heuristic, :283-289)
      const size_t INF = std::numeric_limits<size_t>::max();
      std::vector<size_t> dist(vmax, INF), prev(vmax, INF);
      using QE = std::pair<size_t, size_t>;
      std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
      dist[src] = 0;
      pq.push({0, src});
      while (!pq.empty()) {
        auto [d, u] = pq.top();
        pq.pop();
        if (d != dist[u]) continue;
        if (u == dst) break;
        for (auto [v, w] : adj[u])
          if (d + w < dist[v]) {
            dist[v] = d + w;
            prev[v] = u;
            pq.push({dist[v], v});
          }
      }
      if (dist[dst] == INF) continue;  // None => continue, :292
      uint32_t t0 = generation_of(dst);
      for (size_t v = dst; v != src; v = prev[v]) t0 = std::min(t0, generation_of(v));
      t0 = std::min(t0, generation_of(src));
      const double t1 = (double)nodes[i].generation, t2 = (double)nodes[j].generation;
      if ((double)dist[dst] != t1 - (double)t0 + t2 - (double)t0)  // assert_eq!, :327
        throw Error(ABN_ERR_BAD_PEDIGREE, "path length does not match the generation times");
      ped.push_row((double)t0, t1, t2, dm[i * nn + (j - i - 1)]);
    }
  return {std::move(ped), p0};
}

}  // namespace alphabeta
