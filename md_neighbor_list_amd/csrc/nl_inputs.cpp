// nl_inputs.cpp -- deterministic synthetic particle boxes for tests, tools and bench.py.
// Host-only C++ (libstdc++ <random>), exported with a C ABI and built into libnl_inputs.so.
//
//  * nl_gen_uniform_*: the benchmark workload of SURVEY.md section 8(d) -- i.i.d. uniform positions in
//    [0,L)^3, std::mt19937_64(seed), std::uniform_real_distribution<double>(0,L), draw order x,y,z per
//    particle, cast to the position type, values that round up to L are clamped to nextafter(L,0);
//    particle index = generation order (spatially random: the worst case for an unsorted builder).
//  * nl_gen_fcc_*: the reference harness's own generator (make_list.cpp:34-77 == make_list.cu:26-66):
//    FCC lattice of spacing s = (rho/4)^(-1/3), int(L/s)^3 unit cells x 4 atoms, every coordinate jittered
//    by U(0,0.1) drawn from std::mt19937(2) in the position type.
#include <cmath>
#include <cstdint>
#include <random>

namespace {

template <class T>
int64_t gen_uniform(T* q, int32_t stride, int64_t n, double Lx, double Ly, double Lz, uint64_t seed) {
  std::mt19937_64 mt(seed);
  const double L[3] = {Lx, Ly, Lz};
  std::uniform_real_distribution<double> ud[3] = {std::uniform_real_distribution<double>(0.0, Lx),
                                                  std::uniform_real_distribution<double>(0.0, Ly),
                                                  std::uniform_real_distribution<double>(0.0, Lz)};
  for (int64_t i = 0; i < n; i++) {
    for (int d = 0; d < 3; d++) {
      T v = static_cast<T>(ud[d](mt));
      const T Ld = static_cast<T>(L[d]);
      if (v >= Ld) v = std::nextafter(Ld, static_cast<T>(0));
      q[(size_t)i * stride + d] = v;
    }
    for (int d = 3; d < stride; d++) q[(size_t)i * stride + d] = 0;
  }
  return n;
}

// Returns the number of particles the lattice holds; writes min(count, cap) of them.
template <class T>
int64_t gen_fcc(T* q, int32_t stride, int64_t cap, double density_, double L_) {
  const T density = static_cast<T>(density_), L = static_cast<T>(L_);
  std::mt19937 mt(2);                                   // make_list.cpp:40
  std::uniform_real_distribution<T> ud(0.0, 0.1);       // make_list.cpp:41
  const T s = 1.0 / std::pow(density * 0.25, 1.0 / 3.0);  // make_list.cpp:54
  const T hs = s * 0.5;
  const int sx = static_cast<int>(L / s), sy = sx, sz = sx;
  int64_t n = 0;
  auto add = [&](T x, T y, T z) {
    // draw order x, y, z (make_list.cpp:42-44)
    const T jx = ud(mt), jy = ud(mt), jz = ud(mt);
    if (n < cap) {
      q[(size_t)n * stride + 0] = x + jx;
      q[(size_t)n * stride + 1] = y + jy;
      q[(size_t)n * stride + 2] = z + jz;
      for (int d = 3; d < stride; d++) q[(size_t)n * stride + d] = 0;
    }
    n++;
  };
  for (int iz = 0; iz < sz; iz++)
    for (int iy = 0; iy < sy; iy++)
      for (int ix = 0; ix < sx; ix++) {
        const T x = ix * s, y = iy * s, z = iz * s;
        add(x, y, z);
        add(x, y + hs, z + hs);
        add(x + hs, y, z + hs);
        add(x + hs, y + hs, z);
      }
  return n;
}

}  // namespace

extern "C" {
// Box edge of the benchmark workload: L = cbrt(N / rho), evaluated in double.
double nl_box_length(int64_t n, double density) { return std::cbrt(static_cast<double>(n) / density); }

int64_t nl_gen_uniform_f32(float* q, int32_t stride, int64_t n, double Lx, double Ly, double Lz, uint64_t seed) {
  return gen_uniform<float>(q, stride, n, Lx, Ly, Lz, seed);
}
int64_t nl_gen_uniform_f64(double* q, int32_t stride, int64_t n, double Lx, double Ly, double Lz, uint64_t seed) {
  return gen_uniform<double>(q, stride, n, Lx, Ly, Lz, seed);
}
int64_t nl_gen_fcc_f32(float* q, int32_t stride, int64_t cap, double density, double L) {
  return gen_fcc<float>(q, stride, cap, density, L);
}
int64_t nl_gen_fcc_f64(double* q, int32_t stride, int64_t cap, double density, double L) {
  return gen_fcc<double>(q, stride, cap, density, L);
}
}
