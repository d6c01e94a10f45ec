// Vec.hpp -- the value types the reference's callers hand across the plan boundary, written for this adapter:
//   Vec<N,T>   include/Vec.hpp:175-276 (a uBLAS fixed vector there; a plain array here)
//   Mat3<T>    include/Mat3.hpp:8-87   (StokesSphericalBEM::kernel_value_type)
// Same names and the semantics the reference's solver code relies on:
//   * the default constructor zero-initialises;
//   * Vec<N,T>(a0, ..., aN-1) sets the elements;
//   * a SINGLE arithmetic argument is a size, not a value -- Vec<3,double>(1.) is the ZERO vector (uBLAS size
//     constructor, include/Vec.hpp:209-216; examples/StokesBEM.cpp:260 starts GMRES from it);
//   * element-wise + - * / between vectors and with scalars on either side, compound forms, unary minus, == and !=,
//     dot / norm / normSq / norm_inf as free functions, operator<<, begin()/end(), value_type, dimension;
//   * N contiguous T, nothing else: std::vector<Vec<3,double>> IS the N x 3 row-major array the C ABI takes.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <iostream>
#include <type_traits>

template <std::size_t N, typename T>
class Vec {
  T a_[N];

 public:
  typedef T value_type;
  typedef T* iterator;
  typedef const T* const_iterator;
  typedef std::size_t size_type;
  static constexpr size_type dimension = N;
  static const size_type max_size = N;

  Vec() { for (size_type i = 0; i < N; ++i) a_[i] = T(); }
  explicit Vec(size_type) { for (size_type i = 0; i < N; ++i) a_[i] = T(); }       // a size, as in the reference
  template <typename A0, typename A1, typename... Rest,
            typename = typename std::enable_if<sizeof...(Rest) + 2 == N && std::is_convertible<A0, T>::value>::type>
  Vec(const A0& a0, const A1& a1, const Rest&... rest) {
    const T v[N] = {T(a0), T(a1), T(rest)...};
    for (size_type i = 0; i < N; ++i) a_[i] = v[i];
  }

  size_type size() const { return N; }
  T& operator[](size_type i) { return a_[i]; }
  const T& operator[](size_type i) const { return a_[i]; }
  iterator begin() { return a_; }
  iterator end() { return a_ + N; }
  const_iterator begin() const { return a_; }
  const_iterator end() const { return a_ + N; }
  T* data() { return a_; }
  const T* data() const { return a_; }

  Vec operator-() const { Vec r; for (size_type i = 0; i < N; ++i) r.a_[i] = -a_[i]; return r; }
  Vec& operator+=(const Vec& b) { for (size_type i = 0; i < N; ++i) a_[i] += b.a_[i]; return *this; }
  Vec& operator-=(const Vec& b) { for (size_type i = 0; i < N; ++i) a_[i] -= b.a_[i]; return *this; }
  Vec& operator*=(const Vec& b) { for (size_type i = 0; i < N; ++i) a_[i] *= b.a_[i]; return *this; }
  Vec& operator/=(const Vec& b) { for (size_type i = 0; i < N; ++i) a_[i] /= b.a_[i]; return *this; }
  Vec& operator+=(const T& s) { for (size_type i = 0; i < N; ++i) a_[i] += s; return *this; }
  Vec& operator-=(const T& s) { for (size_type i = 0; i < N; ++i) a_[i] -= s; return *this; }
  Vec& operator*=(const T& s) { for (size_type i = 0; i < N; ++i) a_[i] *= s; return *this; }
  Vec& operator/=(const T& s) { for (size_type i = 0; i < N; ++i) a_[i] /= s; return *this; }
};

template <std::size_t N, typename T> bool operator==(const Vec<N, T>& a, const Vec<N, T>& b) { return std::equal(a.begin(), a.end(), b.begin()); }
template <std::size_t N, typename T> bool operator!=(const Vec<N, T>& a, const Vec<N, T>& b) { return !(a == b); }
template <std::size_t N, typename T>
std::ostream& operator<<(std::ostream& s, const Vec<N, T>& v) {
  for (std::size_t i = 0; i < N; ++i) s << (i ? " " : "") << v[i];
  return s;
}

#define FMMBEM_VEC_OP(op)                                                                                        \
  template <std::size_t N, typename T> Vec<N, T> operator op(Vec<N, T> a, const Vec<N, T>& b) { return a op## = b; } \
  template <std::size_t N, typename T, typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type> \
  Vec<N, T> operator op(Vec<N, T> a, const S& s) { return a op## = T(s); }
FMMBEM_VEC_OP(+)
FMMBEM_VEC_OP(-)
FMMBEM_VEC_OP(*)
FMMBEM_VEC_OP(/)
#undef FMMBEM_VEC_OP
template <std::size_t N, typename T, typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
Vec<N, T> operator+(const S& s, Vec<N, T> a) { return a += T(s); }
template <std::size_t N, typename T, typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
Vec<N, T> operator*(const S& s, Vec<N, T> a) { return a *= T(s); }
template <std::size_t N, typename T, typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
Vec<N, T> operator-(const S& s, const Vec<N, T>& a) { Vec<N, T> r; for (std::size_t i = 0; i < N; ++i) r[i] = T(s) - a[i]; return r; }
template <std::size_t N, typename T, typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
Vec<N, T> operator/(const S& s, const Vec<N, T>& a) { Vec<N, T> r; for (std::size_t i = 0; i < N; ++i) r[i] = T(s) / a[i]; return r; }

template <std::size_t N, typename T> T dot(const Vec<N, T>& a, const Vec<N, T>& b) { T s = T(); for (std::size_t i = 0; i < N; ++i) s += a[i] * b[i]; return s; }
template <std::size_t N, typename T> T inner_prod(const Vec<N, T>& a, const Vec<N, T>& b) { return dot(a, b); }
template <std::size_t N, typename T> T normSq(const Vec<N, T>& a) { return dot(a, a); }
template <std::size_t N, typename T> T norm(const Vec<N, T>& a) { return std::sqrt(normSq(a)); }
template <std::size_t N, typename T> T norm_2(const Vec<N, T>& a) { return norm(a); }
template <std::size_t N, typename T> T norm_inf(const Vec<N, T>& a) { T m = T(); for (std::size_t i = 0; i < N; ++i) m = std::max(m, T(std::fabs(a[i]))); return m; }

// include/Mat3.hpp:8-87: row-major 3x3, what StokesSphericalBEM::operator() returns
template <typename T>
struct Mat3 {
  T vals_[9];
  Mat3() { for (unsigned i = 0; i < 9; ++i) vals_[i] = T(); }
  explicit Mat3(double v) { for (unsigned i = 0; i < 9; ++i) vals_[i] = v; }
  template <typename It> Mat3(It first, It last) { unsigned i = 0; for (; first != last && i < 9; ++first, ++i) vals_[i] = *first; for (; i < 9; ++i) vals_[i] = T(); }
  const T& operator()(unsigned i, unsigned j) const { return vals_[3 * i + j]; }
  T& operator()(unsigned i, unsigned j) { return vals_[3 * i + j]; }
  Mat3 operator-() const { Mat3 r; for (unsigned i = 0; i < 9; ++i) r.vals_[i] = -vals_[i]; return r; }
  Mat3& operator+=(const Mat3& m) { for (unsigned i = 0; i < 9; ++i) vals_[i] += m.vals_[i]; return *this; }
  Mat3 operator+(const Mat3& m) const { Mat3 r(*this); return r += m; }
  Mat3 operator*(double x) const { Mat3 r; for (unsigned i = 0; i < 9; ++i) r.vals_[i] = x * vals_[i]; return r; }
  Vec<3, T> operator*(const Vec<3, T>& x) const {
    return Vec<3, T>(vals_[0] * x[0] + vals_[1] * x[1] + vals_[2] * x[2], vals_[3] * x[0] + vals_[4] * x[1] + vals_[5] * x[2],
                     vals_[6] * x[0] + vals_[7] * x[1] + vals_[8] * x[2]);
  }
  Vec<3, T> multiply(const Vec<3, T>& x) const { return *this * x; }
  Mat3 multiply(double x) const { return *this * x; }
};
