// tests/native/stdsort_check.cpp — host check of lvi-exc_amd/csrc/lvx_stdsort.h against the REAL libstdc++ std::sort / std::partial_sort on inputs with
// many equal keys (the order of equal keys is exactly what the restatement exists to reproduce).  Built by tests/test_host_stdsort.py with g++.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "../../lvi-exc_amd/csrc/lvx_stdsort.h"

namespace {
struct El { float key; int id; };
struct ByKey { bool operator()(const El& x, const El& y) const { return x.key < y.key; } };
}  // namespace

extern "C" {
// returns the number of trials whose element order differs from std::sort's; kind: 0 few distinct keys, 1 mostly distinct with some ties, 2 sorted / reversed / organ-pipe
// runs with ties, 3 all equal; force_heap != 0: compare the heap-sort branch (depth limit 0) against std::partial_sort(first, last, last)
int stdsort_mismatches(int seed, int trials, int max_n, int kind, int force_heap) {
  std::mt19937 rng(seed);
  int bad = 0;
  for (int t = 0; t < trials; ++t) {
    const int n = (force_heap ? 17 : 1) + (int)(rng() % (unsigned)max_n);   // introsort reaches its heap sort only for ranges longer than 16
    std::vector<El> a(n);
    for (int i = 0; i < n; ++i) {
      float k;
      switch (kind) {
        case 0: k = (float)(rng() % 7u); break;
        case 1: k = (rng() % 5u == 0) ? (float)(rng() % 11u) : (float)(rng() % 100000u) * 1e-3f; break;
        case 2: { const int m = t % 3; k = m == 0 ? (float)(i / 3) : m == 1 ? (float)((n - i) / 4) : (float)(std::min(i, n - 1 - i) / 2); break; }
        default: k = 1.5f;
      }
      a[i] = El{k, i};
    }
    std::vector<El> want = a, got = a;
    if (force_heap) std::partial_sort(want.begin(), want.end(), want.end(), ByKey());
    else std::sort(want.begin(), want.end(), ByKey());
    lvx::libstdcxx_sort(got.data(), n, ByKey(), force_heap ? 0 : -1);
    bool same = true;
    for (int i = 0; i < n; ++i) if (want[i].id != got[i].id) { same = false; break; }
    if (!same) ++bad;
  }
  return bad;
}

// packed 64-bit elements as the kernel sorts them: (float bits << 32 | index), compared by the float only; out = index order after the sort
int stdsort_packed(int n, const float* key, int* out_ids) {
  std::vector<unsigned long long> a(n);
  for (int i = 0; i < n; ++i) { unsigned b; std::memcpy(&b, &key[i], 4); a[i] = ((unsigned long long)b << 32) | (unsigned)i; }
  auto less = [](unsigned long long x, unsigned long long y) { float fx, fy; const unsigned bx = (unsigned)(x >> 32), by = (unsigned)(y >> 32); std::memcpy(&fx, &bx, 4); std::memcpy(&fy, &by, 4); return fx < fy; };
  lvx::libstdcxx_sort(a.data(), n, less);
  std::vector<int> ref(n);
  for (int i = 0; i < n; ++i) ref[i] = i;
  std::sort(ref.begin(), ref.end(), [&](int i, int j) { return key[i] < key[j]; });   // scanRegistration.cpp:87 comp + :327
  int bad = 0;
  for (int i = 0; i < n; ++i) { out_ids[i] = (int)(a[i] & 0xffffffffu); if (out_ids[i] != ref[i]) ++bad; }
  return bad;
}
}
