// Stand-in for <rocprim/rocprim.hpp> in the wavefront-emulator build (test infrastructure): the three device-wide primitives the index
// code uses, as plain host loops with rocPRIM's calling convention (first call with a null work space returns its size).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <numeric>
#include <vector>
namespace rocprim {
template <class T> struct double_buffer {
    T *a, *b; double_buffer(T *x, T *y) : a(x), b(y) {}
    T *current() const { return a; } T *alternate() const { return b; } void swap() { std::swap(a, b); }
};
template <class T> struct plus { T operator()(const T &x, const T &y) const { return x + y; } };
template <class T> struct maximum { T operator()(const T &x, const T &y) const { return x < y ? y : x; } };
template <class K, class V> hipError_t radix_sort_pairs(void *tmp, size_t &need, double_buffer<K> &k, double_buffer<V> &v, size_t n, unsigned b0, unsigned b1, hipStream_t = nullptr) {
    if (!tmp) { need = 16; return hipSuccess; }
    const K mask = (b1 - b0 >= sizeof(K) * 8) ? ~(K)0 : (((K)1 << (b1 - b0)) - 1);
    std::vector<size_t> idx(n); std::iota(idx.begin(), idx.end(), (size_t)0);
    const K *kc = k.current(); const V *vc = v.current();
    std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return ((kc[x] >> b0) & mask) < ((kc[y] >> b0) & mask); });
    for (size_t i = 0; i < n; i++) { k.alternate()[i] = kc[idx[i]]; v.alternate()[i] = vc[idx[i]]; }
    k.swap(); v.swap();
    return hipSuccess;
}
template <class In, class Out, class T, class Op> hipError_t exclusive_scan(void *tmp, size_t &need, In in, Out out, T init, size_t n, Op op, hipStream_t = nullptr) {
    if (!tmp) { need = 16; return hipSuccess; }
    T acc = init; for (size_t i = 0; i < n; i++) { T x = in[i]; out[i] = acc; acc = op(acc, x); }
    return hipSuccess;
}
template <class In, class Out, class Op> hipError_t inclusive_scan(void *tmp, size_t &need, In in, Out out, size_t n, Op op, hipStream_t = nullptr) {
    if (!tmp) { need = 16; return hipSuccess; }
    if (n) { auto acc = in[0]; out[0] = acc; for (size_t i = 1; i < n; i++) { acc = op(acc, in[i]); out[i] = acc; } }
    return hipSuccess;
}
}  // namespace rocprim
