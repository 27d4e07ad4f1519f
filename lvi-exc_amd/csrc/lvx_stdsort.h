// lvx_stdsort.h — libstdc++'s std::sort, restated so that ONE lane can reproduce its result for EQUAL keys.
//
// Why: A-LOAM sorts every sector's point indices by curvature with std::sort (src/aloam/src/scanRegistration.cpp:327,
// comp = cloudCurvature[i] < cloudCurvature[j]).  std::sort is not stable: where two points of a sector have the same curvature their
// order — and with it which of them the greedy pick takes first (:336-372) — is whatever libstdc++'s introsort leaves.  A parallel sort can
// only reproduce that when no keys are equal; for sectors WITH equal keys the kernel runs this restatement on one lane instead, and the
// labels / picked flags / sorted indices stay bit-exact against the reference built with libstdc++ (GCC 5 .. 13: bits/stl_algo.h
// __introsort_loop / __final_insertion_sort, bits/stl_heap.h; the algorithm has not changed in that range).
//
// Restated (not copied) from the published algorithm: introsort = median-of-three quicksort down to 16-element runs with a
// 2 * floor(log2 n) depth limit falling back to heap sort, then one insertion sort over everything (guarded for the first 16 elements,
// unguarded after).  tests/native/stdsort_check.cpp compares it against the real std::sort / std::partial_sort on the host.
#pragma once

#if defined(__HIPCC__)
#define LVX_SORT_HD __host__ __device__ __forceinline__
#else
#define LVX_SORT_HD inline
#endif

namespace lvx {

// T: trivially copyable element; less(a, b): strict weak order, called with the same argument order as libstdc++ calls comp
template <class T, class Less> struct StdSort {
  T* a; Less less;

  LVX_SORT_HD void swp(int i, int j) { const T t = a[i]; a[i] = a[j]; a[j] = t; }

  // ---- heap sort (depth limit exhausted): std::__partial_sort(first, last, last) = make_heap + sort_heap ----
  LVX_SORT_HD void push_heap(int first, int hole, int top, T value) {
    int parent = (hole - 1) / 2;
    while (hole > top && less(a[first + parent], value)) { a[first + hole] = a[first + parent]; hole = parent; parent = (hole - 1) / 2; }
    a[first + hole] = value;
  }
  LVX_SORT_HD void adjust_heap(int first, int hole, int len, T value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (less(a[first + child], a[first + child - 1])) child--;
      a[first + hole] = a[first + child];
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); a[first + hole] = a[first + child - 1]; hole = child - 1; }
    push_heap(first, hole, top, value);
  }
  LVX_SORT_HD void heap_sort(int first, int last) {
    const int len = last - first;
    if (len >= 2) for (int parent = (len - 2) / 2;; --parent) { const T v = a[first + parent]; adjust_heap(first, parent, len, v); if (parent == 0) break; }
    while (last - first > 1) { --last; const T v = a[last]; a[last] = a[first]; adjust_heap(first, 0, last - first, v); }
  }

  // ---- quicksort part ----
  LVX_SORT_HD void move_median_to_first(int result, int x, int y, int z) {
    if (less(a[x], a[y])) {
      if (less(a[y], a[z])) swp(result, y);
      else if (less(a[x], a[z])) swp(result, z);
      else swp(result, x);
    } else if (less(a[x], a[z])) swp(result, x);
    else if (less(a[y], a[z])) swp(result, z);
    else swp(result, y);
  }
  LVX_SORT_HD int unguarded_partition(int first, int last, int pivot) {
    for (;;) {
      while (less(a[first], a[pivot])) ++first;
      --last;
      while (less(a[pivot], a[last])) --last;
      if (!(first < last)) return first;
      swp(first, last);
      ++first;
    }
  }
  // ---- insertion sorts ----
  LVX_SORT_HD void unguarded_linear_insert(int last) {
    const T val = a[last];
    int next = last - 1;
    while (less(val, a[next])) { a[last] = a[next]; last = next; --next; }
    a[last] = val;
  }
  LVX_SORT_HD void insertion_sort(int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
      if (less(a[i], a[first])) { const T val = a[i]; for (int k = i; k > first; --k) a[k] = a[k - 1]; a[first] = val; }
      else unguarded_linear_insert(i);
    }
  }

  // std::sort(a, a + n, less); depth_limit < 0: libstdc++'s 2 * floor(log2 n) (tests pass 0 to force the heap-sort branch)
  LVX_SORT_HD void sort(int n, int depth_limit = -1) {
    if (n <= 0) return;
    if (depth_limit < 0) { int lg = 0; for (int m = n; m > 1; m >>= 1) ++lg; depth_limit = 2 * lg; }
    // __introsort_loop: recursion on the right part, iteration on the left — here with an explicit stack (a range is pushed at most once per level)
    int st_first[64], st_last[64], st_depth[64], sp = 0;
    int first = 0, last = n, depth = depth_limit;
    for (;;) {
      while (last - first > 16) {
        if (depth == 0) { heap_sort(first, last); break; }
        --depth;
        const int mid = first + (last - first) / 2;
        move_median_to_first(first, first + 1, mid, last - 1);
        const int cut = unguarded_partition(first + 1, last, first);
        if (sp < 64) { st_first[sp] = cut; st_last[sp] = last; st_depth[sp] = depth; ++sp; }   // the right part [cut, last), with the depth it would have been called with
        last = cut;
      }
      if (sp == 0) break;
      --sp; first = st_first[sp]; last = st_last[sp]; depth = st_depth[sp];
    }
    // __final_insertion_sort
    if (n > 16) { insertion_sort(0, 16); for (int i = 16; i != n; ++i) unguarded_linear_insert(i); }
    else insertion_sort(0, n);
  }
};

template <class T, class Less> LVX_SORT_HD void libstdcxx_sort(T* a, int n, Less less, int depth_limit = -1) {
  StdSort<T, Less> s{a, less};
  s.sort(n, depth_limit);
}

}  // namespace lvx
