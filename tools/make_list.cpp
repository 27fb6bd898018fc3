// tools/make_list.cpp -- timing + self-check driver with the flow of the reference's GPU harness (make_list.cu:102-201):
// allocate cuda_ptr buffers, generate particles, host2dev once, Initialize, LOOP x MakeNeighList(sync=false), one
// device sync, print "# of particles N T[ms]", then rebuild the list by brute force on the host and compare the
// per-particle sorted neighbour sets (make_list.cu:145-198).  What the reference fixes in source is a CLI here:
//
//   make_list [--n N | --lattice] [--rho R] [--rc C] [--dtype f32|f64] [--loop K] [--seed S] [--check 0|1]
//
//   --lattice   the reference's own problem: jittered FCC lattice, L = 50 (N = 119164 at rho 1, make_list.cu:17-24)
//   --n N       uniform random box of N particles at density rho (the BASELINE configs), L = cbrt(N / rho)
// Also prints one JSON line (ms/build, Mpairs/s) after the reference's line.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <numeric>
#include <string>
#include <vector>

#include "neighlist_gpu.hpp"

extern "C" {
double nl_box_length(int64_t n, double density);
int64_t nl_gen_uniform_f32(float* q, int32_t stride, int64_t n, double Lx, double Ly, double Lz, uint64_t seed);
int64_t nl_gen_uniform_f64(double* q, int32_t stride, int64_t n, double Lx, double Ly, double Lz, uint64_t seed);
int64_t nl_gen_fcc_f32(float* q, int32_t stride, int64_t cap, double density, double L);
int64_t nl_gen_fcc_f64(double* q, int32_t stride, int64_t cap, double density, double L);
}

template <typename D> struct Vec4 {
  D x, y, z, w;
};

// make_list.cu:79-98: full list, j != i, reject iff dr2 > rc2 (here restricted to `rows` particles when N is large)
template <typename Vec, typename D>
void bruteforce_rows(const Vec* q, int32_t n, D rc2, const std::vector<int32_t>& rows, std::vector<std::vector<int32_t>>& out) {
  out.assign(rows.size(), {});
  for (size_t r = 0; r < rows.size(); r++) {
    const int32_t i = rows[r];
    for (int32_t j = 0; j < n; j++) {
      if (i == j) continue;
      const D drx = q[i].x - q[j].x, dry = q[i].y - q[j].y, drz = q[i].z - q[j].z;
      const D dr2 = drx * drx + dry * dry + drz * drz;
      if (dr2 > rc2) continue;
      out[r].push_back(j);
    }
  }
}

template <typename D> int run(int64_t n_req, bool lattice, double rho, double rc, int loop, uint64_t seed, int check) {
  typedef Vec4<D> Vectype;
  double L;
  int32_t n;
  cuda_ptr<Vectype> q;
  if (lattice) {
    L = 50.0;
    n = (int32_t)(sizeof(D) == 4 ? nl_gen_fcc_f32(nullptr, 4, 0, rho, L) : nl_gen_fcc_f64(nullptr, 4, 0, rho, L));
    q.allocate(n);
    if (sizeof(D) == 4) nl_gen_fcc_f32(reinterpret_cast<float*>(&q[0]), 4, n, rho, L);
    else nl_gen_fcc_f64(reinterpret_cast<double*>(&q[0]), 4, n, rho, L);
  } else {
    n = (int32_t)n_req;
    L = nl_box_length(n, rho);
    q.allocate(n);
    if (sizeof(D) == 4) nl_gen_uniform_f32(reinterpret_cast<float*>(&q[0]), 4, n, L, L, L, seed);
    else nl_gen_uniform_f64(reinterpret_cast<double*>(&q[0]), 4, n, L, L, L, seed);
  }
  q.host2dev();  // make_list.cu:119

  NeighListGPU<Vectype, D> nlistmaker((D)rc, (D)L, (D)L, (D)L);  // make_list.cu:122
  nlistmaker.Initialize(n);                                       // make_list.cu:123
  nlistmaker.MakeNeighList(q, n, true);                           // warm-up (sizes the list)
  const auto beg = std::chrono::system_clock::now();
  for (int i = 0; i < loop; i++) nlistmaker.MakeNeighList(q, n, false);  // make_list.cu:125-127
  nlistmaker.Synchronize();                                              // make_list.cu:128
  const auto end = std::chrono::system_clock::now();
  const double ms = std::chrono::duration<double, std::milli>(end - beg).count();
  std::cout << "# of particles " << n << " " << (long long)ms << "[ms]\n";  // make_list.cu:131-132
  const int64_t half_pairs = nlistmaker.half_number_of_pairs();
  std::printf("{\"n\": %d, \"rho\": %g, \"rc\": %g, \"dtype\": \"%s\", \"loop\": %d, \"ms_per_build\": %.4f, \"half_pairs\": %lld, "
              "\"mpairs_per_s\": %.1f}\n",
              n, rho, rc, sizeof(D) == 4 ? "f32" : "f64", loop, ms / loop, (long long)half_pairs, half_pairs / (ms / loop) / 1e3);
  if (!check) return 0;

  // make_list.cu:136-198 -- copy back, brute force on the host, compare sorted neighbour sets
  cuda_ptr<int32_t>& neigh_list = nlistmaker.neigh_list();
  cuda_ptr<int32_t>& number_of_partners = nlistmaker.number_of_partners();
  neigh_list.dev2host();
  number_of_partners.dev2host();
  std::vector<int32_t> rows;
  if ((int64_t)n * n <= 40000LL * 40000LL) {
    rows.resize(n);
    std::iota(rows.begin(), rows.end(), 0);
  } else {  // O(N^2) is out of reach: check a deterministic sample of rows against an O(N) scan each
    for (int k = 0; k < 2000; k++) rows.push_back((int32_t)(((int64_t)k * 2654435761LL) % n));
  }
  std::vector<std::vector<int32_t>> ref;
  const D rc2 = (D)rc * (D)rc;  // SEARCH_LENGTH2 has type Dtype in the harness (make_list.cu:24)
  bruteforce_rows(&q[0], n, rc2, rows, ref);
  if (rows.size() == (size_t)n) {
    int64_t total = 0;
    for (auto& r : ref) total += (int64_t)r.size();
    if (total != nlistmaker.number_of_pairs()) {
      std::cerr << "TEST fail\nnumber_of_pairs " << nlistmaker.number_of_pairs() << "\nnumber_of_pairs_ref " << total << "\n";
      return 1;
    }
  }
  std::vector<int32_t> gpu_buf;
  for (size_t r = 0; r < rows.size(); r++) {
    const int32_t i = rows[r];
    if (number_of_partners[i] != (int32_t)ref[r].size()) {
      std::cerr << "TEST fail\ni " << i << "\nnumber_of_partners[i] " << number_of_partners[i] << "\nnumber_of_partners_ref[i] "
                << ref[r].size() << "\n";
      return 1;
    }
    gpu_buf.resize(ref[r].size());
    for (size_t j = 0; j < ref[r].size(); j++) gpu_buf[j] = neigh_list[(size_t)n * j + i];
    std::sort(gpu_buf.begin(), gpu_buf.end());
    std::sort(ref[r].begin(), ref[r].end());
    if (gpu_buf != ref[r]) {
      std::cerr << "TEST fail\ni " << i << "\n";
      return 1;
    }
  }
  std::cerr << "TEST is passed.\n";
  return 0;
}

int main(int argc, char* argv[]) {
  int64_t n = 1 << 20;
  bool lattice = false;
  double rho = 1.0, rc = 3.3;
  int loop = 100, check = 1;
  uint64_t seed = 12345;
  std::string dtype = "f32";
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--n") n = std::atoll(val());
    else if (a == "--lattice") lattice = true;
    else if (a == "--rho") rho = std::atof(val());
    else if (a == "--rc") rc = std::atof(val());
    else if (a == "--dtype") dtype = val();
    else if (a == "--loop") loop = std::atoi(val());
    else if (a == "--seed") seed = std::strtoull(val(), nullptr, 10);
    else if (a == "--check") check = std::atoi(val());
    else {
      std::cerr << "usage: make_list [--n N | --lattice] [--rho R] [--rc C] [--dtype f32|f64] [--loop K] [--seed S] [--check 0|1]\n";
      return 2;
    }
  }
  return dtype == "f64" ? run<double>(n, lattice, rho, rc, loop, seed, check) : run<float>(n, lattice, rho, rc, loop, seed, check);
}
